#!/usr/bin/env python
"""TIMING-ONLY A/B (round 6): the 8-phase GEMM template with v_mfma_f32_16x16x32_bf16 (variant 0, correct) against the same schedule issuing
v_mfma_f32_32x32x16_bf16 on the same operand registers (variant 3: wrong results by construction, same LDS traffic / registers / waits).
python tools/bench_mma32.py [--rounds 5]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_gemm8 as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    out = []
    for name, M, N, K in (("conv3_2", 8 * 154 * 154, 256, 2304), ("conv4_2", 63368, 512, 4608), ("fc6", 2312, 4096, 25088), ("8192^3", 8192, 8192, 8192)):
        for kind in ("relu", "zeros"):
            A = G.fill((M, K), kind)
            B = G.fill((N, K), "zeros" if kind == "zeros" else "randn", K ** -0.5)
            C_ = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fns = {"mfma16x16x32": lambda: G.gemm8(A, B, C_, 0, 0), "mfma32x32x16 (timing only)": lambda: G.gemm8(A, B, C_, 3, 0)}
            for f in fns.values():
                f()
            torch.cuda.synchronize()
            t = {k: [] for k in fns}
            for _ in range(args.rounds):
                for k, f in fns.items():
                    t[k].append(G.timeit(f, args.iters))
            flop = 2.0 * M * N * K
            row = {"shape": name, "operand": kind}
            for k, v in t.items():
                row[k] = round(flop / sorted(v)[len(v) // 2] / 1e9, 1)
            out.append(row)
            print(row)
            del A, B, C_
    print(json.dumps(out))


if __name__ == "__main__":
    main()
