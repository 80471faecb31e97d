#!/usr/bin/env python
"""does a tensor written by one kernel and read by the next one come out of the 256 MB infinity cache?  producer = fill_, consumer =
a 16-B-per-lane read-only pass (torch.max) and x *= c, timed back to back for buffer sizes below / above the cache"""
import torch
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")          # evicts the cache between runs when asked to
for mb in (32, 64, 129, 192, 258, 516, 1032):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda")
    tf = t(lambda: a.fill_(1.0))
    def fill_then_max():
        a.fill_(1.0); a.max()
    def fill_then_mul():
        a.fill_(1.0); a.mul_(1.0001)
    def evict_then_max():
        big.fill_(0); a.max()
    tb = t(lambda: big.fill_(0))
    tm = t(fill_then_max) - tf
    tmu = t(fill_then_mul) - tf
    tcold = t(evict_then_max) - tb
    print("%5d MB: fill %.3f ms (%.2f TB/s) | max right after the fill %.3f ms = %.2f TB/s | max after 1 GiB of other traffic %.3f ms = %.2f TB/s | mul_ after fill %.3f ms = %.2f TB/s (r+w)"
          % (mb, tf, mb * 1.048576e-3 / tf, tm, mb * 1.048576e-3 / tm, tcold, mb * 1.048576e-3 / tcold, tmu, 2 * mb * 1.048576e-3 / tmu))
