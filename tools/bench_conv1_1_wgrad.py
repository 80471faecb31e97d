#!/usr/bin/env python
"""conv1_1 weight gradient at the bench shape (B = 8, 512 x 512, pad 100): ms per call, alone on the device
(SZN_C11_WGRAD_GATHER=1: the round-2 form whose taps are gathered straight from memory)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
B, H, pad = 8, 512, 100
g = torch.Generator().manual_seed(3)
x = (torch.rand(B, 3, H, H, generator=g) * 255 - 110).cuda()
Ho = H + 2 * pad - 2
dt = torch.bfloat16
code = L.dtype_code(dt)
dout = torch.randn(B, Ho, Ho, 64, generator=g).cuda().to(dt)
dw = torch.empty(64, 3, 3, 3, device="cuda")
ws = torch.empty(L.load().szn_conv1_1_wgrad_workspace_bytes(code, B, H, H, pad), dtype=torch.uint8, device="cuda")
fn = lambda: L.call("szn_conv1_1_wgrad", code, B, H, H, pad, L.ptr(x), L.ptr(dout), L.ptr(dw), None, 0, L.ptr(ws), L.stream_ptr())
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print("conv1_1 wgrad: %.1f us per call (kernel + reduce), |dw| = %.6e" % (e0.elapsed_time(e1) / 50 * 1e3, float(dw.double().abs().sum())))
