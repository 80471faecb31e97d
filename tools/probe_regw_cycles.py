#!/usr/bin/env python
"""clock64() split of conv3x3_regw's tile loop per wave (ablation build, SZN_REGW_ABLATE bit 8): cycles per tile in the top part (gate / patch
DMA issue), the MFMA phase, the counted vmcnt wait, group B's barrier, the epilogue, group A's barrier.
usage: SZN_LIB_PATH=.../lib_ablate/libszn_hip.so SZN_REGW_ABLATE=8 tools/probe_regw_cycles.py [layer ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
from bench_conv import SHAPES

B = 8
NAMES = ["top", "mfma", "vmwait", "barB", "epi", "barA"]
for name in (sys.argv[1:] or ["conv1_2", "conv2_1", "conv2_2"]):
    Hi, Ci, Co, K, pad = SHAPES[name]
    x = torch.randn(B, Hi, Hi, Ci, device="cuda").bfloat16()
    w = (torch.randn(Co, 3, 3, Ci, device="cuda") / (Ci * 9) ** 0.5).bfloat16()
    bias = torch.randn(Co, device="cuda")
    out = torch.zeros(B, Hi, Hi, Co, device="cuda", dtype=torch.bfloat16)
    d = L.ConvDesc(L.SZN_BF16, B, Hi, Hi, Ci, Hi, Hi, Co, 3, 3, 1, Ci, Co, Ci, 1, 0)
    for _ in range(3):
        L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), L.stream_ptr())
    torch.cuda.synchronize()
    assert L.load().szn_last_kernel().decode() == "conv3x3_regw"
    f = out.view(-1)[:256 * 8 * 16].view(torch.float32).view(256, 8, 8).cpu()
    ok = f[:, :, 7] == 1.0
    tiles = f[:, :, 6]
    print("%-8s records %d / 2048, tiles per block %.1f" % (name, int(ok.sum()), float(tiles[ok].mean())))
    for grp, sl in (("A (waves 0-3)", slice(0, 4)), ("B (waves 4-7)", slice(4, 8))):
        m = ok[:, sl]
        per = [float((f[:, sl, i][m] / tiles[:, sl][m]).mean()) for i in range(6)]
        print("   group %s cycles per tile: %s | total %.0f" % (grp, "  ".join("%s %.0f" % (n, v) for n, v in zip(NAMES, per)), sum(per)))
