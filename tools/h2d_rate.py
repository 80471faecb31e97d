#!/usr/bin/env python
"""The PCIe-inclusive rate of the training step (the tier contract: `value` is measured with inputs resident in HBM; the rate with the hand-over
from host memory belongs in DESIGN.md).  Same step as bench.py's headline (bf16, B = 8, 512 x 512, E = 300, K = 59), three ways to feed it:
  resident      the batch already in HBM (= bench.py)
  native serial uint8 RGB (B,H,W,3) + int64 labels from PINNED host memory on the compute stream every step, BGR / mean on the GPU
                (utils.image_to_device = szn_image_u8_to_bgr_f32), then the step: nothing overlapped
  native copy   the same, double-buffered: batch n + 1 is copied on a second stream while step n runs (what a DataLoader with pin_memory does)
  reference     the reference's tuple form (trainer_fcn.py:93-95): f32 (B,3,H,W) image + int64 label + the dense target embedding (B,E,H,W) f32
                = 315 MB per image, copied every step (B = 2 here: 8 images of it are 2.5 GB of pinned memory)
python tools/h2d_rate.py [--steps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import engine, models, synth, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    E, K, H, B = 300, 59, 512, args.batch
    emb = synth.make_embeddings(K, E)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=dev)
    m.train()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True, keep_grads=False)
    rng = np.random.RandomState(0)
    host_img = [torch.from_numpy(rng.randint(0, 256, (B, H, H, 3), dtype=np.uint8)).pin_memory() for _ in range(2)]
    host_lbl = [torch.from_numpy(synth.make_labels(B, H, H, K, seed=7 + i, classes=list(range(49)))).pin_memory() for i in range(2)]
    x_res = utils.image_to_device(host_img[0], dev)
    t_res = host_lbl[0].to(dev)

    def timed(fn, steps):
        for _ in range(3):
            fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    out = {"workload": "bench.py headline step (bf16, B=%d, 512x512, E=300, K=59)" % B, "steps": args.steps}
    out["resident_ms"] = round(timed(lambda i: ts.step(x_res, t_res), args.steps), 3)

    def serial(i):
        x = utils.image_to_device(host_img[i & 1], dev)
        t = host_lbl[i & 1].to(dev, non_blocking=True)
        ts.step(x, t)
    out["native_serial_ms"] = round(timed(serial, args.steps), 3)

    copy = torch.cuda.Stream()
    bufs = [None, None]

    def stage(i):
        with torch.cuda.stream(copy):
            u8 = host_img[i & 1].to(dev, non_blocking=True)
            t = host_lbl[i & 1].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        bufs[i & 1] = (u8, t, ev)
    stage(0)

    def overlapped(i):
        u8, t, ev = bufs[i & 1]
        torch.cuda.current_stream().wait_event(ev)
        stage(i + 1)                                   # next batch crosses PCIe while this step runs
        x = utils.image_to_device(u8, dev)
        u8.record_stream(torch.cuda.current_stream()); t.record_stream(torch.cuda.current_stream())
        ts.step(x, t)
    out["native_overlapped_ms"] = round(timed(overlapped, args.steps), 3)
    out["native_bytes_per_step"] = int(B * H * H * 3 + B * H * H * 8)

    # the reference's hand-over: dense f32 image + label + dense target embedding, B = 2 (315 MB of lbl_vec per image)
    Br = 2
    m2 = models.FCN32s(E)
    m2.load_synthetic(1337, device=dev)
    m2.train()
    ts2 = engine.TrainStep(m2, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True, keep_grads=False)
    h_x = torch.from_numpy(synth.make_images(Br, H, H, seed=1)).pin_memory()
    h_t = torch.from_numpy(synth.make_labels(Br, H, H, K, seed=2, classes=list(range(49)))).pin_memory()
    h_vec = torch.empty(Br, E, H, H, dtype=torch.float32).pin_memory()
    h_vec.normal_()

    def res2(i):
        ts2.step(x2, t2)
    x2, t2 = h_x.to(dev), h_t.to(dev)
    r2 = timed(res2, 10)

    def ref_form(i):
        x = h_x.to(dev, non_blocking=True)
        t = h_t.to(dev, non_blocking=True)
        v = h_vec.to(dev, non_blocking=True)          # the tuple's lbl_vec: accepted and ignored by the native step (gathered on the GPU instead)
        ts2.step(x, t)
        del v
    f2 = timed(ref_form, 10)
    out["reference_tuple_form"] = {"batch": Br, "resident_ms": round(r2, 3), "with_dense_lbl_vec_copy_ms": round(f2, 3),
                                   "bytes_per_step": int(Br * (3 * 4 + 8 + E * 4) * H * H)}
    for k in ("resident_ms", "native_serial_ms", "native_overlapped_ms"):
        out[k.replace("_ms", "_Mpx_s")] = round(B * H * H / (out[k] * 1e-3) / 1e6, 1)
    out["reference_tuple_form"]["Mpx_s"] = round(Br * H * H / (f2 * 1e-3) / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
