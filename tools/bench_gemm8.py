#!/usr/bin/env python
"""A/B of the CDNA guide's 256^2 8-phase GEMM template (tools/gemm8/gemm_8phase.hip) against this library's 256 x 256-tile kernels
on THIS repo's GEMM shapes, same box, same operands, interleaved rounds.

    python tools/bench_gemm8.py [--rounds 5] [--iters 20] [--out profiles/r04_gemm8_ab.json]

shapes: conv3_2 forward as a GEMM (M = 8 x 178^2 = 253,472, N = 256, K = 9 x 256 = 2,304) and conv4_2 (M = 8 x 89^2 = 63,368, N = 512,
K = 4,608), plus 4096^3 / 8192^3 (the guide's own shapes).  Contenders per shape:
  gemm8[...]         the template (variants: xor8 / st16x32 / linear swizzle; no-setprio; lockstep groups)
  libszn 1x1 conv    libszn_hip's kernel for the same plain GEMM (a 1x1 convolution with Ci = K); the kernel name is recorded
  libszn 3x3 conv    the kernel the training step actually runs for that layer (3x3 convolution, same FLOPs); SZN_WIDE_8PH=0
                     selects the round 1-3 kernels (conv_igemm_wide / conv3x3_wide_rows), the default is conv_igemm_8ph
operands: randn, relu(randn) (what the layers see), uniform(-1,1), zeros (DVFS upper bound; never quoted as a result)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402

G8 = C.CDLL(os.path.join(ROOT, "tools", "_build", "libgemm8.so"))
G8.gemm8_bf16.restype = C.c_int
G8.gemm8_bf16.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]


def gemm8(A, B, Cm, variant=0, flags=0):
    M, K = A.shape
    N = B.shape[0]
    rc = G8.gemm8_bf16(M, N, K, Cm.shape[1], A.data_ptr(), B.data_ptr(), Cm.data_ptr(), variant, flags,
                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def wide_gemm(A, B, Cm, bhw=None):
    """the same GEMM through szn_conv2d_fwd: a 1x1 convolution over a (b, h, w, K) map with b h w = M (map sides stay below 32768)"""
    M, K = A.shape
    N = B.shape[0]
    b, h, w_ = bhw if bhw else (1, M // 64, 64)
    assert b * h * w_ == M
    d = L.ConvDesc(L.SZN_BF16, b, h, w_, K, h, w_, N, 1, 1, 0, K, Cm.shape[1], 0, 0, 0)
    L.call("szn_conv2d_fwd", C.byref(d), L.ptr(A), L.ptr(B), None, None, None, L.ptr(Cm), L.stream_ptr())


def conv3x3(x, w, out):
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    d = L.ConvDesc(L.SZN_BF16, B, H, W, Ci, H, W, Co, 3, 3, 1, Ci, Co, 0, 0, 0)
    L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), None, None, None, L.ptr(out), L.stream_ptr())


def fill(shape, kind, scale=1.0):
    if kind == "zeros":
        return torch.zeros(shape, device="cuda", dtype=torch.bfloat16)
    if kind == "uniform":
        return ((torch.rand(shape, device="cuda") * 2 - 1) * scale).to(torch.bfloat16)
    t = torch.randn(shape, device="cuda")
    if kind == "relu":
        t = torch.relu(t)
    return (t * scale).to(torch.bfloat16)


def check():
    """refcheck of every template variant against torch fp32 (transpose-detecting: A, B random, M != N, ragged M)"""
    worst = 0.0
    for (M, N, K) in ((256, 256, 64), (512, 256, 256), (1000, 512, 4608), (253472 // 64, 256, 2304), (300, 264, 128)):
        A, B = fill((M, K), "randn"), fill((N, K), "randn", K ** -0.5)
        want = A.float() @ B.float().t()
        for variant, flags in ((0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (0, 3)):
            out = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
            for _ in range(3):                   # repeated: a read racing its LDS-DMA would show up as a changing tile
                gemm8(A, B, out, variant, flags)
            torch.cuda.synchronize()
            err = float((out.float() - want).abs().max() / want.abs().max())
            worst = max(worst, err)
            assert err < 1e-2, (M, N, K, variant, flags, err)
    return worst


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    L.load()
    res = {"refcheck_max_rel_err": check(), "rounds": args.rounds, "iters": args.iters, "peak_TF": 2500.0, "shapes": []}
    print("refcheck ok, worst rel err %.2e" % res["refcheck_max_rel_err"])
    shapes = [("conv3_2", 253472, 256, 2304, (8, 178, 178, 256)), ("conv4_2", 63368, 512, 4608, (8, 89, 89, 512)),
              ("4096^3", 4096, 4096, 4096, None), ("8192^3", 8192, 8192, 8192, None)]
    kinds = ["relu", "randn", "uniform", "zeros"]
    if args.quick:
        shapes, kinds = shapes[:2], kinds[:2]
    for name, M, N, K, conv in shapes:
        for kind in kinds:
            A = fill((M, K), kind)
            B = fill((N, K), "zeros" if kind == "zeros" else ("uniform" if kind == "uniform" else "randn"), 1.0 if kind == "uniform" else K ** -0.5)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            cont = {"gemm8[xor8]": lambda: gemm8(A, B, out, 0, 0), "gemm8[st16x32]": lambda: gemm8(A, B, out, 1, 0),
                    "gemm8[linear]": lambda: gemm8(A, B, out, 2, 0), "gemm8[xor8,no-setprio]": lambda: gemm8(A, B, out, 0, 1),
                    "gemm8[xor8,lockstep]": lambda: gemm8(A, B, out, 0, 2), "gemm8[xor8,lockstep,no-setprio]": lambda: gemm8(A, B, out, 0, 3)}
            kernels = {}
            if N % 256 == 0 and (M * K * 2) < 0x7fff0000:
                cont["libszn 1x1 conv (same GEMM)"] = lambda: wide_gemm(A, B, out, conv[:3] if conv else None)
            if conv is not None:
                Bc, Hc, Wc, Ci = conv
                x = fill((Bc, Hc, Wc, Ci), kind)
                w = fill((N, 3, 3, Ci), "zeros" if kind == "zeros" else "randn", (9 * Ci) ** -0.5)
                oc = torch.empty(Bc, Hc, Wc, N, device="cuda", dtype=torch.bfloat16)
                cont["libszn 3x3 conv (the layer itself)"] = lambda: conv3x3(x, w, oc)
            for k, fn in cont.items():               # warm-up + which kernel the dispatcher took
                fn()
                if not k.startswith("gemm8"):
                    kernels[k] = L.last_kernel()
            torch.cuda.synchronize()
            times = {k: [] for k in cont}
            for _ in range(args.rounds):
                for k, fn in cont.items():
                    times[k].append(timeit(fn, args.iters))
            flop = 2.0 * M * N * K
            row = {"shape": name, "M": M, "N": N, "K": K, "operand": kind, "results": {}}
            for k, v in times.items():
                med, best = sorted(v)[len(v) // 2], min(v)
                row["results"][k] = {"ms_median": round(med, 4), "ms_min": round(best, 4), "TF_median": round(flop / med / 1e9, 1),
                                     "TF_max": round(flop / best / 1e9, 1), "frac_of_2.5PF": round(flop / med / 1e9 / 2500.0, 4)}
                if k in kernels:
                    row["results"][k]["kernel"] = kernels[k]
            res["shapes"].append(row)
            print("%-8s %-8s " % (name, kind) + "  ".join("%s %.0f" % (k.split(" ")[0], r["TF_median"]) for k, r in row["results"].items()))
            del A, B, out
            torch.cuda.empty_cache()
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res)[:200])


if __name__ == "__main__":
    main()
