#!/usr/bin/env python
"""Micro-benchmark of szn_adam_step on the flat parameter buffer size of the bench (134.3 M elements).
usage: tools/bench_adam.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L

n = 134_260_544 + 1_228_800
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
v.abs_()
lp = torch.empty(n, device="cuda", dtype=torch.bfloat16)
st = L.stream_ptr()
fn = lambda: L.call("szn_adam_step", n, L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), 1e-5, 0.9, 0.999, 1e-8, 0.0, 3, 1.0, L.ptr(lp), L.SZN_BF16, st)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("szn_adam_step: %.3f ms  %.0f GB/s (30 B/element)" % (ms, n * 30 / ms / 1e6))
