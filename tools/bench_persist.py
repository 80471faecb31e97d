#!/usr/bin/env python
"""Round 6 experiment: the 8-phase GEMM template (tools/gemm8) one tile per block (variant 0) against a PERSISTENT block that walks its tiles
(variant 4: the next tile's prologue loads issued before the finished tile's epilogue stores; variant 5: behind them).  All three are checked against
torch first.  python tools/bench_persist.py [--rounds 5]"""
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
    worst = 0.0
    for (M, N, K) in ((256, 256, 64), (1000, 512, 4608), (70000, 512, 256), (66000, 256, 128), (300, 264, 128)):
        A, B = G.fill((M, K), "randn"), G.fill((N, K), "randn", K ** -0.5)
        want = A.float() @ B.float().t()
        ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        G.gemm8(A, B, ref, 0, 0)
        for variant in (4, 5):
            out = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
            for _ in range(3):
                G.gemm8(A, B, out, variant, 0)
            torch.cuda.synchronize()
            err = float((out.float() - want).abs().max() / want.abs().max())
            assert err < 1e-2, (M, N, K, variant, err)
            assert torch.equal(out, ref), (M, N, K, variant, "not bit-identical to the one-tile-per-block kernel")
            worst = max(worst, err)
    print("refcheck ok (persistent variants bit-identical to variant 0), worst rel err %.2e" % worst)
    out = []
    for name, M, N, K in (("conv4_2", 63368, 512, 4608), ("conv4 half K", 63368, 512, 2304), ("conv3-like", 189728, 256, 1152),
                          ("conv3-like K=2304 (874 MB of A)", 189728, 256, 2304), ("fc6", 2312, 4096, 25088), ("8192^3", 8192, 8192, 8192)):
        A = G.fill((M, K), "relu")
        B = G.fill((N, K), "randn", K ** -0.5)
        C_ = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fns = {"one tile / block": lambda: G.gemm8(A, B, C_, 0, 0), "persistent, prologue before epilogue": lambda: G.gemm8(A, B, C_, 4, 0),
               "persistent, prologue behind epilogue": lambda: G.gemm8(A, B, C_, 5, 0)}
        for f in fns.values():
            f()
        torch.cuda.synchronize()
        t = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, f in fns.items():
                t[k].append(G.timeit(f, args.iters))
        flop = 2.0 * M * N * K
        row = {"shape": name, "tiles": ((M + 255) // 256) * ((N + 255) // 256), "K_tiles": K // 64}
        for k, v in t.items():
            med = sorted(v)[len(v) // 2]
            row[k] = {"us": round(med * 1e3, 1), "TF": round(flop / med / 1e9, 1)}
        out.append(row)
        print(row)
        del A, B, C_
    print(json.dumps(out))


if __name__ == "__main__":
    main()
