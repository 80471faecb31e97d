#!/usr/bin/env python
"""conv1_1 forward at the bench shape (B = 8, 512 x 512, pad 100): ms, output write rate, and the error against torch fp32"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
B, H, pad = 8, 512, 100
g = torch.Generator().manual_seed(3)
x = (torch.rand(B, 3, H, H, generator=g) * 255 - 110).cuda()
w = (torch.randn(64, 3, 3, 3, generator=g) / 5)
bias = torch.randn(64, generator=g).cuda()
wd = w.permute(0, 2, 3, 1).contiguous().cuda()
Ho = H + 2 * pad - 2
for dt in (torch.bfloat16, torch.float16):
    out = torch.empty(B, Ho, Ho, 64, device="cuda", dtype=dt)
    fn = lambda: L.call("szn_conv1_1_fwd", L.dtype_code(dt), B, H, H, pad, L.ptr(x), L.ptr(wd), L.ptr(bias), L.ptr(out), L.stream_ptr())
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = F.relu(F.conv2d(x[:1], w.cuda(), bias, padding=pad))
    err = (out[:1].float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()
    print("%s: %.3f ms  %.2f TB/s of output  max err / max |ref| = %.2e  (%s)" % (dt, ms, out.numel() * 2 / ms / 1e9, float(err), L.last_kernel()))
