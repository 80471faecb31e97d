#!/usr/bin/env python
"""per-step GPU / host times of the phase-2 step (engine.SeenmaskStep) as bench.py builds it: looks for the source of a slow repeat.
usage: tools/diag_phase2.py [--gc 0|1] [--reps 5] [--steps 10] [--after-phase1 1]"""
import argparse, gc, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import engine, models, synth

ap = argparse.ArgumentParser()
ap.add_argument("--gc", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--after-phase1", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
E, K, B, H = 300, 59, 8, 512
emb = synth.make_embeddings(K, E) if hasattr(synth, "make_embeddings") else np.random.RandomState(0).randn(K, E).astype(np.float32)
model = models.FCN32s(E).load_synthetic(1337).to(dev).train()
x = torch.from_numpy(synth.make_images(B, H, H, seed=1)).to(dev)
tgt = torch.from_numpy(synth.make_labels(B, H, H, K, seed=2, block=32)).to(dev)
if a.after_phase1:
    ts = engine.TrainStep(model, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True)
    for _ in range(5):
        ts.step(x, tgt)
    torch.cuda.synchronize()
    del ts
p2 = engine.SeenmaskStep(model, K, [50, 51], lr=1e-3, precision=torch.bfloat16)
for _ in range(3):
    p2.step(x, tgt)
torch.cuda.synchronize()
if not a.gc:
    gc.disable()
for rep in range(a.reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host = []
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(a.steps):
        t0 = time.perf_counter()
        p2.step(x, tgt)
        host.append((time.perf_counter() - t0) * 1e3)
        ev[i + 1].record()
    torch.cuda.synchronize()
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
    print("rep %d mean %.3f  gpu/step: %s  host enqueue: %s" % (rep, sum(gpu) / len(gpu), " ".join("%.2f" % g for g in gpu),
                                                              " ".join("%.2f" % h for h in host)), flush=True)
