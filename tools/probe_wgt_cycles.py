#!/usr/bin/env python
"""clock64() split of conv_wgrad_taps per block (ablation build, SZN_WGT_ABLATE=9): cycles per tile, of which in vmcnt(0) + barrier.
usage: SZN_LIB_PATH=.../lib_ablate/libszn_hip.so SZN_WGT_ABLATE=9 tools/probe_wgt_cycles.py [layer ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
from bench_conv import SHAPES

SLAB = 64 * 9 * 64
B = 8
for name in (sys.argv[1:] or ["conv4_2", "conv3_2", "conv2_2", "conv1_2"]):
    Hi, Ci, Co, K, pad = SHAPES[name]
    x = torch.randn(B, Hi, Hi, Ci, device="cuda").bfloat16()
    dout = torch.randn(B, Hi, Hi, Co, device="cuda").bfloat16()
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    ws = torch.zeros(2 * 256 * SLAB * 4, dtype=torch.uint8, device="cuda")
    d = L.ConvDesc(L.SZN_BF16, B, Hi, Hi, Ci, Hi, Hi, Co, 3, 3, 1, Ci, Co, 0, 0, 0)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    for _ in range(3):
        L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), 0, L.stream_ptr())
    torch.cuda.synchronize()
    f = ws.view(torch.float32).view(-1, SLAB)[:, :4].cpu()
    f = f[f[:, 3] > 0]
    tiles = f[:, 3]
    per_tile = f[:, 2] / tiles
    wait = f[:, 0] / tiles
    print("%-8s blocks %4d tiles/block %5.1f | cycles per tile: mean %7.0f min %7.0f max %7.0f | in vmcnt(0)+barrier: mean %6.0f (%.1f %%) max %6.0f"
          % (name, f.shape[0], float(tiles.mean()), float(per_tile.mean()), float(per_tile.min()), float(per_tile.max()),
             float(wait.mean()), 100.0 * float((f[:, 0].sum() / f[:, 2].sum())), float(wait.max())))
