#!/usr/bin/env python
"""bench.py -- train-step throughput of the SZN pixel-embedding path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16|fp32] [--size 512]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = the reference's hot-loop body (trainer_fcn.py:149-180) on one synthetic batch already resident
in HBM: FCN32s forward (train mode, Dropout2d on) -> cosine loss -> train-time infer_lbl -> backward ->
gradient all-reduce (N > 1) -> Adam step -> confusion histogram.  Workload = BASELINE.json configs[1]:
512x512, E = 300 (pascal 21 x 300 embedding matrix), bf16 operands / fp32 accumulate / fp32 master weights.
(The reference has no FCN8s -- SURVEY.md D1 -- so the backbone is its FCN32s.)

Prints ONE JSON line: metric train_Mpixels_per_sec (whole job), plus
  roofline     -- the dominant kernel family (every conv / fc forward and dgrad launch behind szn_conv2d_fwd /
                  szn_conv2d_dgrad: conv_igemm_v2 | conv_igemm_wide | conv3x3_regw): algorithmic FLOPs of its launches
                  / their HIP-event-measured duration, against the dense MFMA peak,
  cpu_baseline -- the CPU oracle (oracle/, C + OpenMP "port" of the reference algorithm) timed on this
                  host's cores on ONE 512x512 image of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (weak scaling)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--embed-dim", type=int, default=300)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--unfused-head", action="store_true", help="materialise the (B,E,H,W) score like the reference")
    ap.add_argument("--phase", choices=["fcn", "seenmask"], default="fcn",
                    help="fcn = phase 1 (the headline train step); seenmask = phase 2 (BASELINE configs[2]: frozen backbone, "
                         "seen-mask head + 2-class cross entropy, trainer_seenmask.py:72-102)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    return ap.parse_args()


def conv_flops(d):
    """algorithmic FLOPs of one conv launch described by a ConvDesc (forward or dgrad-as-forward)"""
    return 2.0 * d.B * d.Ho * d.Wo * d.Co * d.Ci * d.KH * d.KW


def cpu_baseline(E, K, H, emb):
    """time the CPU oracle on ONE image of the same workload (forward + loss + infer + backward + Adam)"""
    from oracle import szn_oracle as O
    from zeroshotsemanticsegmentation_amd import synth
    rng = np.random.default_rng(1337)
    params = {}
    for name, co, ci, k in synth.layer_table(E):
        b = np.sqrt(6.0 / (ci * k * k))
        params[name + ".weight"] = ((rng.random((co, ci, k, k), dtype=np.float32) * 2 - 1) * b).astype(np.float32)
        params[name + ".bias"] = ((rng.random((co,), dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32)
    m = O.FCN32sOracle(params, E)
    x = synth.make_images(1, H, H)
    tgt = synth.make_labels(1, H, H, K)
    opt = O.Adam(1e-5)
    t0 = time.time()
    f = m.forward(x, "fcn", keep=True)
    t1 = time.time()
    loss, df, _ = O.cosine_loss(f, tgt, embed=emb)
    t2 = time.time()
    O.infer_lbl(f, emb)
    t3 = time.time()
    g = m.backward(df=df)
    t4 = time.time()
    g = {k: v for k, v in g.items() if k.split(".")[0] in O.WEIGHT_GROUP}
    opt.step(m.p, g, lambda k: 1e-5 * (2 if k.endswith(".bias") else 1))
    t5 = time.time()
    total = t5 - t0
    return {"value": round(H * H / total / 1e6, 6), "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "1 image %dx%d, E=%d, K=%d, fp32: fwd %.1fs loss %.1fs infer %.1fs bwd %.1fs adam %.1fs (C+OpenMP oracle)"
                      % (H, H, E, K, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torchrun with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # backend "nccl" is RCCL on ROCm
    L.load()

    E, H, B = args.embed_dim, args.size, args.batch
    emb_np = np.load(os.path.join(ROOT, "tests", "golden", "embeddings_pascal_300.npy")) if E == 300 else \
        synth.make_embeddings(21, E)
    K = emb_np.shape[0]
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32

    torch.manual_seed(1337)                                   # identical initial weights on every rank
    model = models.FCN32s(n_class=E)
    model.load_synthetic(1337, device=dev)
    model.train()
    x = torch.from_numpy(synth.make_images(B, H, H, seed=1337 + rank)).to(dev)
    target = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337 + rank)).to(dev)
    if args.phase == "fcn":
        ts = engine.TrainStep(model, emb_np, optimizer="adam", lr=1e-5, precision=dtype, fused_head=not args.unfused_head)
    else:
        # phase 2 (train.py:164-175): everything frozen except seenmask_score (w, b) and seenmask_upscore (w); binary target
        # "label is a seen class" with unlabelled pixels = 0 (trainer_seenmask.py:55-56); 2-class CE, size_average=True
        from zeroshotsemanticsegmentation_amd import optim as szn_optim, utils as szn_utils
        if world > 1:
            raise SystemExit("--phase seenmask is a single-GPU line (98 KB of gradients)")
        model.set_precision(dtype)
        for p in model.parameters():
            p.requires_grad = False
        head = [model.seenmask_score.weight, model.seenmask_score.bias, model.seenmask_upscore.weight]
        for p in head:
            p.requires_grad = True
        opt2 = szn_optim.FusedAdam(head, lr=1e-5)
        unseen = (16, 18)                                             # cfg 18 split of the pascal classes
        lut = torch.ones(K + 1, dtype=torch.int64, device=dev)
        lut[list(unseen)] = 0
        lut[K] = 0
        bin_target = lut[torch.where(target >= 0, target, torch.full_like(target, K))]

        class _Phase2(object):
            def step(self, xx, tt):
                score = model(xx, mode="seenmask")
                loss = szn_utils.cross_entropy2d(score, bin_target, size_average=True)
                opt2.zero_grad()
                loss.backward()
                opt2.step()
                return loss.detach(), szn_utils.channel_argmax(score)
        ts = _Phase2()

    # ---- kernel-level timing of the dominant kernel family (conv fwd/dgrad) with HIP events on the launch stream ----
    events, flops_per_step = [], [0.0]
    record = [False]
    if not args.no_kernel_events:
        orig_call = L.call

        def timed_call(name, *a):
            if record[0] and name in ("szn_conv2d_fwd", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm"):
                d = a[0]._obj
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig_call(name, *a)
                e1.record()
                events.append((e0, e1, conv_flops(d)))
            else:
                orig_call(name, *a)
        L.call = timed_call
        models.L.call = timed_call

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, pred = ts.step(x, target)
    sync()
    record[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, pred = ts.step(x, target)
    sync()
    dt = time.perf_counter() - t0
    record[0] = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lossv = float(loss.item())
    if not np.isfinite(lossv):
        raise SystemExit("loss is not finite: %r" % lossv)

    if rank == 0:
        mpx = world * B * H * H * args.steps / dt / 1e6
        out = {
            "metric": "train_Mpixels_per_sec", "value": round(mpx, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if dtype == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: FCN32s (the reference has no FCN8s) + %d-d pixel projection, "
                                    "%dx%d, K=%d, Adam lr 1e-5, train step fwd+cosine loss+infer_lbl+bwd+optimizer" % (E, H, H, K))
                       if args.phase == "fcn" else
                       ("BASELINE configs[2] (phase 2): seen-mask head on the frozen FCN32s backbone, %dx%d, K=%d, 2-class CE, "
                        "train step fwd+CE+argmax+head bwd+Adam" % (H, H, K)),
                       "per_gpu_batch": B, "global_batch": B * world, "head": ("unfused" if args.unfused_head else "fused-from-coarse") if args.phase == "fcn" else "seenmask_score + learned 64x64 s32 deconv",
                       "parallelism": "dp%d" % world, "final_loss": round(lossv, 5)},
        }
        if events:
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in events)
            fl = sum(f for _, _, f in events)
            peak = 2500.0 if dtype == torch.bfloat16 else 157.3
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None      # HBM bytes per launch from a committed PMC run of this same command (tools/pmc_bench.sh)
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("per_gpu_batch") == B and tj.get("precision") == args.precision and H == 512 and E == 300 and args.phase == "fcn":
                    traffic = round(tj["hbm_bytes_per_launch"])
            out["roofline"] = {"bound": "mfma", "kernel": "conv fwd + dgrad launches (conv_igemm_v2 | conv_igemm_wide | conv3x3_regw; fc6 dgrad = wide GEMM + col2im)",
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": traffic, "launches_per_step": len(events) // args.steps,
                               "avg_launch_ms": round(ms / len(events), 4),
                               "gflop_per_launch": round(fl / len(events) / 1e9, 2),
                               "share_of_step": round(ms / (dt * 1e3), 3)}
            # whole-step MFMA-class algorithmic FLOPs (4.342 MFLOP/px at 512^2, E=300)
            if H == 512 and E == 300 and args.phase == "fcn":
                out["roofline"]["step_mfma_frac"] = round(4.342e6 * B * H * H * args.steps / dt / 1e12 / peak, 4)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(E, K, H, emb_np)
            except Exception as ex:      # the baseline is a reported extra: never lose the measured line
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
