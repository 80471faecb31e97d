#!/usr/bin/env python
"""One-image step with fc6's weight gradient + Adam on a CU-masked stream (SZN_FC6_CUMASK), timed on the default stream and on a
non-blocking torch stream (a stream made by hipExtStreamCreateWithCUMask is a BLOCKING stream: it synchronises with the null stream).
usage: tools/probe_cumask.py [--dtype bf16] [--stream default|side] [--steps 30]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
from zeroshotsemanticsegmentation_amd import engine, models, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stream", default="default")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L.load()
    E, H, K = 300, 512, 59
    rng = np.random.RandomState(7)
    emb = rng.randn(K, E).astype(np.float32)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(1337)
    m = models.FCN32s(n_class=E)
    m.load_synthetic(1337, device=dev)
    m.train()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=dt, fused_head=True, keep_grads=False)
    x = torch.from_numpy(synth.make_images(a.batch, H, H, seed=1337)).to(dev)
    t = torch.from_numpy(synth.make_labels(a.batch, H, H, K, seed=1337)).to(dev)
    side = torch.cuda.Stream() if a.stream == "side" else torch.cuda.current_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):
            ts.step(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            ts.step(x, t)
        e1.record()
        torch.cuda.synchronize()
    print("mask=%s stream=%s %s B=%d: %.3f ms/step  loss %.5f" % (os.environ.get("SZN_FC6_CUMASK", "-"), a.stream, a.dtype, a.batch,
                                                                 e0.elapsed_time(e1) / a.steps, float(ts.loss.item())), flush=True)


if __name__ == "__main__":
    main()
