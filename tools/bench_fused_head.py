#!/usr/bin/env python
"""the fused phase-1 head (szn_fused_head) at the bench shape: ms per call with HIP events around it"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L, synth
B, H, E, K, h = 8, 512, 300, 59, 17
CP = 320
emb = torch.from_numpy(synth.make_embeddings(K, E)).cuda()
coarse = torch.randn(B, h, h, CP, device="cuda")
tgt = torch.from_numpy(synth.make_labels(B, H, H, K, seed=2, block=32)).cuda()
loss = torch.zeros(1, device="cuda"); stats = torch.empty(B, 2, device="cuda")
pred = torch.empty(B, H, H, dtype=torch.int64, device="cuda")
dco = torch.zeros(B, h, h, CP, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(L.load().szn_fused_head_workspace_bytes(B, h, h, E, K), dtype=torch.uint8, device="cuda")
fn = lambda: L.call("szn_fused_head", B, h, h, E, CP, 0, H, H, 19, K, L.ptr(coarse), L.ptr(emb), L.ptr(tgt), L.ptr(loss), L.ptr(stats),
                    L.ptr(pred), L.dtype_code(dco.dtype), L.ptr(dco), L.ptr(ws), L.stream_ptr())
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print("szn_fused_head: %.4f ms per call; loss %.6f, pred checksum %d, dcoarse checksum %.6f"
      % (e0.elapsed_time(e1) / 50, float(loss), int(pred.sum()), float(dco.float().abs().sum())))
