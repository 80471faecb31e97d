#!/usr/bin/env python
"""fc6 / fc7 weight gradient with and without the Adam step in its epilogue (szn_conv2d_wgrad_adam), and the separate optimizer pass,
on the shapes of the bench step.  python tools/bench_wgrad_adam.py [--batch 8]   (SZN_WGW_STAGGER is read once per process)"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    L.load()
    out = {"stagger_env": os.environ.get("SZN_WGW_STAGGER"), "half": os.environ.get("SZN_WGW_HALF"), "half_stagger": os.environ.get("SZN_WGH_STAGGER"),
           "batch": args.batch, "layers": {}}
    for name, Hi, Ci, Co, k in (("fc6", 23, 512, 4096, 7), ("fc7", 17, 4096, 4096, 1)):
        B, dt = args.batch, torch.bfloat16
        Ho = Hi - k + 1
        x = torch.relu(torch.randn(B, Hi, Hi, Ci, device="cuda")).to(dt)
        dout = (torch.randn(B, Ho, Ho, Co, device="cuda") * 1e-3).to(dt)
        n = Co * k * k * Ci
        p, m1, m2, dw = (torch.zeros(n, device="cuda") for _ in range(4))
        lp = torch.zeros(n, device="cuda", dtype=dt)
        d = L.ConvDesc(L.SZN_BF16, B, Hi, Hi, Ci, Ho, Ho, Co, k, k, 0, Ci, Co, 0, 0, 0)
        a = L.AdamArgs()
        a.param, a.exp_avg, a.exp_avg_sq, a.w_lp, a.w_lp_dtype = p.data_ptr(), m1.data_ptr(), m2.data_ptr(), lp.data_ptr(), L.SZN_BF16
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step, a.grad_scale = 1e-5, 0.9, 0.999, 1e-8, 0.0, 1, 1.0
        st = L.stream_ptr()
        t_w = timeit(lambda: L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), 0, st))
        t_a = timeit(lambda: L.call("szn_adam_step", n, L.ptr(p), L.ptr(dw), L.ptr(m1), L.ptr(m2), 1e-5, 0.9, 0.999, 1e-8, 0.0, 1, 1.0,
                                    L.ptr(lp), L.SZN_BF16, st))
        t_f = timeit(lambda: L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), None, C.byref(a), st))
        t_fk = timeit(lambda: L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), C.byref(a), st))
        flop = 2.0 * B * Ho * Ho * Co * k * k * Ci
        out["layers"][name] = {"wgrad_us": round(t_w, 1), "adam_us": round(t_a, 1), "sum_us": round(t_w + t_a, 1), "fused_us": round(t_f, 1),
                               "fused_keep_grads_us": round(t_fk, 1), "wgrad_TF": round(flop / t_w / 1e6, 1),
                               "fused_GBps": round(n * 26 / t_f / 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
