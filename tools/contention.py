#!/usr/bin/env python
"""What does a kernel of another queue holding CUs cost the train step?  (stand-in for the RCCL all-reduce of the fc6
gradient bucket, which runs under the backward pass at N > 1 and cannot be measured on a 1-GPU box)

A spin kernel of `--hog-blocks` workgroups is launched on a second stream right after the forward pass and keeps its CUs
for `--hog-ms` milliseconds while the backward pass runs on the main stream.  Prints ms/step with and without the hog,
for the library's default (one block per CU for the persistent kernels) and for SZN_WGT_OVERSUB=2 (run the script twice:
the knob is read once per process).

usage: tools/contention.py [--hog-blocks 32] [--hog-ms 3.0] [--steps 10]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
from zeroshotsemanticsegmentation_amd import engine, models, synth


def debug_lib():
    """hipcc-compile tools/szn_debug_spin.hip into tools/_build/ on first use (the spin kernel is not in libszn_hip.so)"""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libszn_debug.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-I", here,
                               os.path.join(here, "szn_debug_spin.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.szn_debug_spin.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hog-blocks", type=int, default=32)
    ap.add_argument("--hog-ms", type=float, default=3.0)
    ap.add_argument("--heavy", action="store_true", help="~100-VGPR spinner: cannot co-reside with the persistent kernels")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    L.load()
    dbg = debug_lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emb = np.load(os.path.join(root, "tests", "golden", "embeddings_pascal_300.npy"))
    m = models.FCN32s(300)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.train()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16)
    x = torch.from_numpy(synth.make_images(8, 512, 512)).cuda()
    t = torch.from_numpy(synth.make_labels(8, 512, 512, 21)).cuda()
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    cycles = int(a.hog_ms * 1e-3 * 2.1e9)

    # hook the hog in front of the backward pass: TrainStep._backward is the first thing after the fused head
    orig_backward = ts._backward
    hog = [False]

    def backward_with_hog(ctx, dcoarse, layer_done):
        if hog[0]:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rc = dbg.szn_debug_spin(-a.hog_blocks if a.heavy else a.hog_blocks, cycles, L.ptr(sink), L.stream_ptr())
                assert rc == 0, rc
        return orig_backward(ctx, dcoarse, layer_done)
    ts._backward = backward_with_hog

    def run(flag):
        hog[0] = flag
        for _ in range(3):
            ts.step(x, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            ts.step(x, t)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    base = run(False)
    hogged = run(True)
    print("oversub=%s heavy=%d hog=%d blocks x %.1f ms: %.3f ms/step without, %.3f ms/step with the hog (+%.3f)"
          % (os.environ.get("SZN_WGT_OVERSUB", "1"), int(a.heavy), a.hog_blocks, a.hog_ms, base, hogged, hogged - base))


if __name__ == "__main__":
    main()
