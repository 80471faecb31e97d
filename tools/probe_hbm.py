#!/usr/bin/env python
"""write / read / copy rates of plain torch kernels on big buffers (what a store-bound kernel of the step can hope for)"""
import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (516, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    ms = t(lambda: a.zero_());            print("fill  %5d MB: %.3f ms  %.2f TB/s (write)" % (mb, ms, mb * 1.048576e-3 / ms))
    ms = t(lambda: a.fill_(1.5));         print("fill_ %5d MB: %.3f ms  %.2f TB/s (write)" % (mb, ms, mb * 1.048576e-3 / ms))
    ms = t(lambda: b.copy_(a));           print("copy  %5d MB: %.3f ms  %.2f TB/s (read + write)" % (mb, ms, 2 * mb * 1.048576e-3 / ms))
    ms = t(lambda: a.sum());              print("sum   %5d MB: %.3f ms  %.2f TB/s (read)" % (mb, ms, mb * 1.048576e-3 / ms))
    ms = t(lambda: a.mul_(1.0001));       print("mul_  %5d MB: %.3f ms  %.2f TB/s (read + write)" % (mb, ms, 2 * mb * 1.048576e-3 / ms))
