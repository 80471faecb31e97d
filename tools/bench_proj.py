#!/usr/bin/env python
"""MFMA fraction of the pixel-projection GEMM: three clearly labelled numbers.

  (1) TRUE shape      M = B*289 (the 17x17 fc7 map of a 512x512 image), K = 4096, N = 300 -- what score_fr executes,
                      in the reference and here (the projection runs BEFORE the x32 bilinear upsampling);
  (2) AGGREGATE       MFMA fraction over all MFMA-class kernels of a train step: printed by bench.py
                      (roofline.step_mfma_frac) -- the honest utilisation figure;
  (3) NOMINAL shape   M = B*H*W = 262,144*B, K = 4096, N = 300: what an "H*W x 300 projection" would cost at full
                      resolution, run as a stand-alone synthetic GEMM through the same C-ABI entry point.

usage: tools/bench_proj.py [--iters 10]   (bf16 operands, fp32 accumulate, N padded to the 304-element row stride)"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L

PEAK = 2500.0   # TFLOP/s dense bf16 (MI355X_MICROARCH.md)


def run(B, H, W, iters, N=300, K=4096, relu=True):
    dt = torch.bfloat16
    code = L.dtype_code(dt)
    ldo = (N + 7) // 8 * 8
    x = torch.randn(B, H, W, K, device="cuda")
    if relu:                       # score_fr reads relu7: non-negative, half of the elements zero
        x = torch.relu(x)
    x = x.to(dt)
    w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(B, H, W, ldo, device="cuda", dtype=dt)
    d = L.ConvDesc(code, B, H, W, K, H, W, N, 1, 1, 0, K, ldo, 0, 0, 0)
    st = L.stream_ptr()
    fn = lambda: L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), st)
    for _ in range(30 if B * H * W >= 65536 else 3):      # cold TLBs / ramping clocks under-report the first launches by 15-25 % (probe_proj_warmup.py)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * B * H * W * K * N / (ms * 1e-3) / 1e12
    return ms, tf, L.last_kernel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--nominal-only", action="store_true", help="only the 262,144-row shape (the PMC passes)")
    a = ap.parse_args()
    L.load()
    res = {"peak_TF": PEAK, "true_shape": [], "nominal_shape": []}
    for B in (() if a.nominal_only else (1, 8, 64)):
        ms, tf, kern = run(B, 17, 17, a.iters)
        res["true_shape"].append({"M": B * 289, "K": 4096, "N": 300, "ms": round(ms, 4), "TF": round(tf, 1), "frac": round(tf / PEAK, 4),
                                  "kernel": kern})
    for B, relu in ((1, True),) if a.nominal_only else ((1, True), (1, False), (4, True)):
        ms, tf, kern = run(B, 512, 512, a.iters, relu=relu)
        res["nominal_shape"].append({"M": B * 262144, "K": 4096, "N": 300, "ms": round(ms, 4), "TF": round(tf, 1), "frac": round(tf / PEAK, 4),
                                     "kernel": kern, "operand": "relu(randn)" if relu else "randn",
                                     "activation_stream_TBps": round(B * 262144 * 4096 * 2 / (ms * 1e-3) / 1e12, 2)})
    res["aggregate"] = "see bench.py roofline.step_mfma_frac"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
