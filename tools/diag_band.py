#!/usr/bin/env python
"""bf16 step with the band removed from {no, conv3, conv2 + conv3} blocks: per-layer relative L2 difference of the weight gradients"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import engine, models, synth

size, B = int(sys.argv[1]), int(sys.argv[2])
prec = torch.bfloat16 if len(sys.argv) < 4 or sys.argv[3] == "bf16" else torch.float32
res = {}
for mode in ("none", "conv3_1", "conv2_1", "conv2_1,conv3_1"):
    models._BAND_CROP = mode != "none"
    models._BAND_16BIT = mode.split(",")
    m = models.FCN32s(20); m.load_synthetic(1337, device=torch.device("cuda", 0)); m.eval()
    ts = engine.TrainStep(m, synth.make_embeddings(33, 20), optimizer="adam", lr=1e-5, precision=prec, fused_head=True, keep_grads=True, fused_adam=False)
    x = torch.from_numpy(synth.make_images(B, size, size, seed=9)).cuda()
    t = torch.from_numpy(synth.make_labels(B, size, size, 33, seed=10, block=16)).cuda()
    ts.keep_ctx = False
    loss, pred = ts.step(x, t)
    torch.cuda.synchronize()
    res[mode] = (float(loss), ts.flat_gw.double().clone(), ts.flat_gb.double().clone(), ts)
l0, g0, b0, ts0 = res["none"]
for mode in list(res)[1:]:
    l, g, b, _ = res[mode]
    print("mode", mode, "loss", l, "vs", l0)
    for n in ts0.layers:
        o, c = ts0.woff[n]; bo, bc = ts0.boff[n]
        print("   %-10s w %.3e  b %.3e" % (n, float((g[o:o+c]-g0[o:o+c]).norm()/g0[o:o+c].norm()), float((b[bo:bo+bc]-b0[bo:bo+bc]).norm()/(b0[bo:bo+bc].norm()+1e-300))))
