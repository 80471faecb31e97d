#!/usr/bin/env python
"""Host pacing of engine.TrainStep: seconds the Python thread needs to ENQUEUE one step (no synchronisation inside the loop) next to
the GPU time of the step.  The step is GPU-bound while enqueue < GPU time; the margin is what N ranks sharing the host cores can lose.
usage: tools/host_pace.py [--arch fcn32s|fcn8s] [--batch 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeroshotsemanticsegmentation_amd import engine, models, synth

ap = argparse.ArgumentParser(); ap.add_argument("--arch", default="fcn32s"); ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
E, K, H, B = 300, 59, 512, a.batch
emb = synth.make_embeddings(K, E)
m = (models.FCN8s if a.arch == "fcn8s" else models.FCN32s)(E); m.load_synthetic(1337, device=torch.device("cuda")); m.train()
ts = engine.TrainStep(m, emb, precision=torch.bfloat16)
x = torch.from_numpy(synth.make_images(B, H, H)).cuda(); t = torch.from_numpy(synth.make_labels(B, H, H, K)).cuda()
for _ in range(5): ts.step(x, t)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): ts.step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s B=%d: enqueue %.2f ms/step (host), wall %.2f ms/step" % (a.arch, B, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# the same with the GPU idle-free: enqueue time alone (steps queue up behind the GPU, so the loop above may have been throttled by
# the launch queue depth); measure one step enqueued on an idle GPU
torch.cuda.synchronize(); t0 = time.perf_counter(); ts.step(x, t); t1 = time.perf_counter(); torch.cuda.synchronize()
print("single step on an idle GPU: enqueue %.2f ms" % ((t1 - t0) * 1e3))
