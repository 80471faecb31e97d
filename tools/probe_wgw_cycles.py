#!/usr/bin/env python
"""Where a tile of fc6's weight gradient + Adam (conv_wgrad_wide<T, true>) spends its cycles: clock64 at the phase boundaries, wave 0 of every block
(`make ABLATE=1` build, SZN_WGW_ABLATE=9; results are still correct).  SZN_LIB_PATH=.../lib_ablate/libszn_hip.so SZN_WGW_ABLATE=9 python tools/probe_wgw_cycles.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402

B, Hi, Ci, Co, k = 8, 23, 512, 4096, 7
Ho = Hi - k + 1
dt = torch.bfloat16
x = torch.relu(torch.randn(B, Hi, Hi, Ci, device="cuda")).to(dt)
dout = (torch.randn(B, Ho, Ho, Co, device="cuda") * 1e-3).to(dt)
n = Co * k * k * Ci
p, m1, m2 = (torch.zeros(n, device="cuda") for _ in range(3))
lp = torch.zeros(n, device="cuda", dtype=dt)
tiles = (Co // 256) * (Ci // 256) * k * k
dbg = torch.zeros(tiles * 16, device="cuda")
d = L.ConvDesc(L.SZN_BF16, B, Hi, Hi, Ci, Ho, Ho, Co, k, k, 0, Ci, Co, 0, 0, 0)
d.dw_lp = dbg.data_ptr()
a = L.AdamArgs()
a.param, a.exp_avg, a.exp_avg_sq, a.w_lp, a.w_lp_dtype = p.data_ptr(), m1.data_ptr(), m2.data_ptr(), lp.data_ptr(), L.SZN_BF16
a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step, a.grad_scale, a.grad_optional = 1e-5, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 1
for _ in range(3):
    L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), None, C.byref(a), L.stream_ptr())
torch.cuda.synchronize()
assert L.last_kernel() == "conv_wgrad_wide_adam", L.last_kernel()
t = dbg.view(tiles, 16).cpu()
names = ["prologue + K loop", "pass 0 whole", "pass 1: loads issued -> barrier 1", "barrier 1 -> gradient staged", "staged -> group 0 updated / stored",
         "groups 1..6", "group 7 + stores issued", "passes 2 + 3", "tile total"]
import numpy as np
v = t.numpy()
print("fc6 weight gradient + Adam, B = 8: %d tiles; cycles per tile, wave 0 (median | p10 | p90 over the tiles)" % tiles)
for i, nm in enumerate(names):
    c = v[:, i]
    print("  %-40s %9.0f | %9.0f | %9.0f" % (nm, np.median(c), np.percentile(c, 10), np.percentile(c, 90)))
first = v[:256]
print("first-round tiles (blocks 0..255) total median %.0f, later rounds %.0f" % (np.median(first[:, 8]), np.median(v[256:, 8])))
