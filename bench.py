#!/usr/bin/env python
"""bench.py -- train-step throughput of the SZN pixel-embedding path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16|fp32] [--size 512] [--classes 59]
    (N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run, one rank per GPU; the driver's own
     `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` form works as well)

One "step" = the reference's hot-loop body (trainer_fcn.py:149-180) on one synthetic batch already resident in HBM:
FCN32s forward (train mode, Dropout2d on) -> cosine loss -> train-time infer_lbl -> backward -> gradient all-reduce
(N > 1) -> Adam step -> confusion histogram.  Default workload = the configuration BASELINE.json's metric is quoted on:
PASCAL-Context 512x512, K = 59 classes (49 seen / 10 unseen; synthetic 59 x 300 embedding matrix of SURVEY 8-d, labels
drawn from the seen classes), E = 300, bf16 operands / fp32 accumulate / fp32 master weights, 8 images per GPU.
(`--classes 21` = BASELINE configs[1], the PASCAL-VOC matrix.  The reference has no FCN8s -- SURVEY D1 -- the backbone
is its FCN32s.)

Prints ONE JSON line: metric train_Mpixels_per_sec (whole job) plus
  roofline      the dominant kernel (conv_igemm_8ph: conv3_x / conv4_x forward + dgrad, fc6 forward): algorithmic FLOPs of
                its launches / their HIP-event-measured duration INSIDE the timed region, against the dense MFMA peak;
                `traffic` = HBM bytes per launch from the committed PMC passes (source file named; null if none matches)
  kernels       per kernel family, MFMA-class and HBM-class, from 3 extra instrumented steps after the timed region
                (HIP events around every C-ABI call): ms per step, algorithmic FLOPs or bytes, fraction of the peak
  projection    the three labelled MFMA numbers SURVEY 8-d asks for (true shape, step aggregate, nominal shape)
  phase2        BASELINE configs[2]: the seen-mask step (engine.SeenmaskStep: frozen backbone forward, fused-from-coarse
                2-class head, head-only backward + Adam), three repeats, its own roofline record
  comm          N = 1: BASELINE configs[3]'s gradient exchange forced through a one-rank RCCL communicator on this GPU (child process):
                step time with the real buckets going through librccl on its own high-priority stream vs without, fp32 / bf16 wire
                (written by the weight-gradient kernels / through staging copies) / sharded optimizer.  N > 1: the same variants
                measured across the ranks, exposed-comm ms and per-bucket issue / wait times (the line explains its own scaling)
  fp32, b1      measured in a child process after the headline (a crash there cannot lose the line): the same step at the
                reference's arithmetic (fp32, B = 8, against the 157.3 TF fp32 MFMA peak) and at the reference's batch size
                (B = 1, bf16 and fp32, eager and replayed from a captured hipGraph)
  cpu_baseline  the same train step on this host's cores (rank 0, N = 1 only): torch-CPU restatement (what the reference
                executes; primary) and the C + OpenMP oracle beside it; one 512x512 image each.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16, PEAK_F32, PEAK_HBM = 2500.0, 157.3, 8000.0       # TFLOP/s dense MFMA, GB/s (MI355X_MICROARCH.md)
STEP_MFLOP_PER_PX = {512: 4.342, 768: 3.678}                # SURVEY 8-d, E = 300, phase-1 train step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (weak scaling)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--embed-dim", type=int, default=300)
    ap.add_argument("--classes", type=int, default=59, help="59 = PASCAL-Context (synthetic matrix), 21 = PASCAL-VOC matrix")
    ap.add_argument("--precision", choices=["bf16", "fp16", "fp32"], default="bf16",
                    help="fp16 = IEEE-half activations / weight images with static loss scaling (BASELINE configs[4])")
    ap.add_argument("--head-fp8", action="store_true", help="projection head forward on the fp8 (e4m3) matrix cores (configs[4])")
    ap.add_argument("--unfused-head", action="store_true", help="materialise the (B,E,H,W) score like the reference")
    ap.add_argument("--arch", choices=["fcn32s", "fcn8s"], default="fcn32s",
                    help="fcn8s: the public FCN8s skip head BASELINE's north_star names (not in the reference; autograd path, "
                         "per-tensor fused Adam) -- a secondary line, the default stays the reference's FCN32s")
    ap.add_argument("--phase", choices=["fcn", "seenmask"], default="fcn",
                    help="fcn = phase 1 (headline; phase 2 is reported as a sub-record); seenmask = phase 2 as the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernels / projection / phase2 / fp32 / b1 sub-records")
    ap.add_argument("--grad-comm", choices=["auto", "fp32", "bf16", "bf16-sharded", "off"], default="auto",
                    help="N > 1: wire format of the gradient exchange.  auto = bf16 written by the weight-gradient kernels themselves on "
                         "the 16-bit paths (271 MB per step instead of 542), fp32 on the fp32 path; bf16-sharded = reduce-scatter + "
                         "rank-sharded Adam + all-gather of the weight image; the `comm` record times the alternatives beside it")
    ap.add_argument("--sub-record", choices=["fp32", "b1", "comm", "c1_fcn8s", "c4_768"], default=None,
                    help="(internal) measure one sub-record and print it as JSON; run by the main process as a child")
    return ap.parse_args()


# ---- algorithmic work of one C-ABI call (what `achieved` is computed from; formulas in DESIGN.md section 4) -------------
def _esize(code):
    return 4 if code == 0 else 2


def _conv_flops(d):
    return 2.0 * d.B * d.Ho * d.Wo * d.Co * d.Ci * d.KH * d.KW


def call_work(name, a):
    """-> (bound, work) with work in FLOP (mfma) or bytes (hbm); None for calls that are not accounted"""
    if name in ("szn_conv2d_fwd", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm", "szn_conv2d_dgrad_gemm_native", "szn_conv2d_wgrad",
                "szn_conv2d_wgrad_adam"):      # (the last one also moves 26 B per weight: fc6's Adam step rides in its epilogue)
        return "mfma", _conv_flops(a[0]._obj)
    if name == "szn_conv1_1_fwd":          # reads the f32 image once, writes B x (H+198)^2 x 64 activations
        code, B, H, W, pad = a[:5]
        return "hbm", B * 3 * H * W * 4.0 + B * (H + 2 * pad - 2) * (W + 2 * pad - 2) * 64.0 * _esize(code)
    if name == "szn_conv1_1_wgrad":        # reads dout once + the image
        code, B, H, W, pad = a[:5]
        return "hbm", B * 3 * H * W * 4.0 + B * (H + 2 * pad - 2) * (W + 2 * pad - 2) * 64.0 * _esize(code)
    if name == "szn_maxpool2x2_ceil_fwd":
        code, B, Hi, Wi, Cc = a[:5]
        return "hbm", B * Cc * _esize(code) * (Hi * Wi + ((Hi + 1) // 2) * ((Wi + 1) // 2))
    if name == "szn_maxpool2x2_ceil_bwd":  # pool input + pooled + d(pooled) in, d(input) out
        code, B, Hi, Wi, Cc = a[:5]
        return "hbm", B * Cc * _esize(code) * (2.0 * Hi * Wi + 2.0 * ((Hi + 1) // 2) * ((Wi + 1) // 2))
    if name in ("szn_maxpool2x2_ceil_bwd_code", "szn_maxpool2x2_ceil_bwd_code_cb"):   # d(pooled) + one code byte per pooled element in, d(input) out
        code, B, Hi, Wi, Cc = a[:5]
        return "hbm", B * Cc * (_esize(code) * (1.0 * Hi * Wi + ((Hi + 1) // 2) * ((Wi + 1) // 2)) + ((Hi + 1) // 2) * ((Wi + 1) // 2))
    if name == "szn_adam_step":            # p, g, m, v in; p, m, v out (+ the bf16 weight image when asked for)
        return "hbm", a[0] * (28.0 + (2.0 if a[12] is not None else 0.0))
    if name == "szn_sgd_momentum_step":
        return "hbm", a[0] * (20.0 + (2.0 if a[9] is not None else 0.0))
    if name == "szn_fused_head":           # label in, prediction out, coarse map in, dcoarse out
        B, h, w, E, ldc, c0, H, W = a[:8]
        return "hbm", B * H * W * 16.0 + 2.0 * B * h * w * E * 4.0
    if name == "szn_fused_head_strided":
        B, h, w, E, ldc, c0, H, W = a[1:9]
        return "hbm", B * H * W * 16.0 + 2.0 * B * h * w * E * 4.0
    if name == "szn_pack_weight_dgrad_batch":
        code, n, _, _, co, k, ci = a[:7]
        return "hbm", sum(2.0 * co[i] * k[i] * k[i] * ci[i] * _esize(code) for i in range(n))
    if name == "szn_pack_weight_dgrad":
        code, co, kh, kw, ci = a[:5]
        return "hbm", 2.0 * co * kh * kw * ci * _esize(code)
    if name in ("szn_confusion_hist", "szn_confusion_hist_k"):
        return "hbm", a[0] * 16.0
    if name in ("szn_embed_argmax", "szn_embed_argmax_k"):         # E*s read + 8 B written per pixel
        B, E, H, W, K = a[:5]
        return "hbm", B * H * W * (E * 4.0 + 8.0)
    if name in ("szn_cosine_loss_fwd", "szn_mse_loss_fwd"):
        B, E, H, W, K = a[:5]
        return "hbm", B * H * W * (E * 4.0 + 8.0)
    if name in ("szn_cosine_loss_bwd", "szn_mse_loss_bwd"):
        B, E, H, W, K = a[:5]
        return "hbm", B * H * W * (2.0 * E * 4.0 + 8.0)
    if name in ("szn_bilinear_up32_crop_fwd", "szn_bilinear_up32_crop_bwd"):
        B, h, w, E, ldc, c0, H, W = a[:8]
        return "hbm", B * H * W * E * 4.0
    if name in ("szn_bilinear_up_crop_fwd", "szn_bilinear_up_crop_bwd"):
        B, h, w, E, ldc, c0, H, W = a[1:9]
        return "hbm", B * H * W * E * 4.0
    return None


def cpu_baseline(E, K, H, emb, arch="fcn32s"):
    """the train step on ONE image on this host's cores: torch-CPU restatement (primary) and the C + OpenMP oracle"""
    from oracle import szn_oracle as O
    from oracle import torch_ref as T
    from zeroshotsemanticsegmentation_amd import synth
    import torch
    x = synth.make_images(1, H, H)
    tgt = synth.make_labels(1, H, H, K)
    cores = os.cpu_count()
    out = {"unit": "Mpixels/s", "cores": cores, "kind": "port"}
    # torch's default thread count = the physical cores of the host (128 on the 256-thread GPU boxes).  One thread per LOGICAL
    # core (BASELINE.md section 3 reads "all host cores") was measured too: 0.0048 Mpx/s against 0.038-0.048 -- oversubscribed
    # SMT siblings make mkldnn ~10x slower -- so the faster setting is the baseline; `threads` states what was used
    t = T.timed_train_step(E, K, H, emb, x, tgt, steps=2, arch=arch)  # second step timed (first pays allocation / mkldnn setup)
    out["value"] = round(H * H / t["total"] / 1e6, 6)
    out["threads"] = torch.get_num_threads()
    out["sample"] = ("torch-CPU restatement of the %s step (oracle/torch_ref.py; depthwise upscore), 1 image %dx%d, "
                     "E=%d, K=%d, fp32, 2nd of 2 steps: fwd %.2fs loss %.2fs infer %.2fs bwd %.2fs adam %.2fs"
                     % ("reference" if arch == "fcn32s" else "FCN8s", H, H, E, K, t["fwd"], t["loss"], t["infer"], t["bwd"], t["adam"]))
    if arch != "fcn32s":
        return out                      # the C oracle restates the reference's FCN32s only
    try:
        rng = np.random.default_rng(1337)
        params = {}
        for name, co, ci, k in synth.layer_table(E):
            b = np.sqrt(6.0 / (ci * k * k))
            params[name + ".weight"] = ((rng.random((co, ci, k, k), dtype=np.float32) * 2 - 1) * b).astype(np.float32)
            params[name + ".bias"] = ((rng.random((co,), dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32)
        m = O.FCN32sOracle(params, E)
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
        out["c_oracle"] = {"value": round(H * H / (t5 - t0) / 1e6, 6), "unit": "Mpixels/s", "cores": cores,
                           "sample": "C+OpenMP oracle, same image: fwd %.1fs loss %.1fs infer %.1fs bwd %.1fs adam %.1fs"
                                     % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)}
    except Exception as ex:
        out["c_oracle"] = {"value": None, "sample": "failed: %r" % (ex,)}
    return out


def projection_report(L, torch, step_frac, iters=10):
    """(1) true shape M = B*289, (2) step aggregate, (3) nominal full-resolution shape; bf16, K=4096, N=300"""
    import ctypes as C

    def run(B, H, W, N=300, K=4096, relu=True):
        dt = torch.bfloat16
        ldo = (N + 7) // 8 * 8
        x = torch.randn(B, H, W, K, device="cuda")
        if relu:                                   # score_fr reads relu7: non-negative, half of the elements zero
            x = torch.relu(x)
        x = x.to(dt)
        w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).to(dt)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(B, H, W, ldo, device="cuda", dtype=dt)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        d = L.ConvDesc(L.SZN_BF16, B, H, W, K, H, W, N, 1, 1, 0, K, ldo, 0, 0, 0)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        st = L.stream_ptr()
        fn = lambda: L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), st)
        # Warm-up: the first launches on a freshly allocated 2.1 GB operand (and right behind the microsecond-sized true shapes) run on cold
        # TLBs and ramping clocks -- three warm-up launches (rounds 2-3) under-reported the kernel by 15-25 % (tools/probe_proj_warmup.py:
        # 0.31-0.39 of peak after 3 warm-up launches, 0.40-0.41 after 30, same kernel, same box).  `frac` = a window of `iters` launches behind
        # `warm` warm-up launches; `frac_sustained` = the following 3 x iters launches (the part settles ~2 % lower under sustained load).
        warm = 30                                   # the same count for every row
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()

        def window(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        ms = window(iters)
        tf = 2.0 * B * H * W * K * N / (ms * 1e-3) / 1e12
        r = {"M": B * H * W, "K": K, "N": N, "ms": round(ms, 4), "TF": round(tf, 1), "frac": round(tf / PEAK_BF16, 4),
             "kernel": L.last_kernel(), "operand": "relu(randn)" if relu else "randn", "warmup_launches": warm, "timed_launches": iters,
             "activation_stream_TBps": round(B * H * W * K * 2 / (ms * 1e-3) / 1e12, 2)}
        if B * H * W >= 65536:
            ms2 = window(3 * iters)
            r["frac_sustained"] = round(2.0 * B * H * W * K * N / (ms2 * 1e-3) / 1e12 / PEAK_BF16, 4)
        return r
    return {"peak_TF": PEAK_BF16,
            "true_shape": [run(B, 17, 17) for B in (1, 8, 64)],
            "step_aggregate_frac": step_frac,
            "nominal_shape": run(1, 512, 512),
            "nominal_shape_randn": run(1, 512, 512, relu=False),
            "note": "true = what score_fr executes (17x17 map, before the x32 upsampling); nominal = a full-resolution "
                    "H*W x 4096 x 300 projection the path never runs (proj_gemm_stream: the 2.1 GB activation operand is read "
                    "once from HBM, AI = 300 FLOP/B; operands relu(randn) like fc7's output, randn beside it)"}


def _workload(args, torch, rank=0):
    """(embedding matrix, seen, unseen, images, phase-1 labels, all-class labels) of the configured workload"""
    from zeroshotsemanticsegmentation_amd import synth, trainer_fcn
    E, H, B, K = args.embed_dim, args.size, args.batch, args.classes
    if K == 21 and E in (20, 21, 300):
        emb_np, seen, unseen = trainer_fcn.load_embeddings("pascal", E), list(range(21)), [16, 18]
    elif K == 33 and E in (20, 300):
        emb_np, seen, unseen = trainer_fcn.load_embeddings("context", E), list(range(33)), [16, 18]
    else:
        emb_np = synth.make_embeddings(K, E)
        n_unseen = 10 if K == 59 else max(K // 6, 1)
        seen, unseen = list(range(K - n_unseen)), list(range(K - n_unseen, K))     # K = 59: seen 0..48, unseen 49..58
    return emb_np, seen, unseen


def _time_steps(torch, fn, steps, warmup):
    """ms per call of fn(): HIP events on the current stream around `steps` calls"""
    import gc
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gc.collect()
    gc.disable()                     # a generation-2 pass over the bench's event lists is a 10-20 ms host stall inside a 30 ms region
    try:
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    finally:
        gc.enable()
    return e0.elapsed_time(e1) / steps


def _conv_family(torch, L, mods, fn):
    """one call of fn() with HIP events around every conv forward / dgrad launch -> (TFLOP/s, ms, launches)"""
    orig = L.call
    ev = []

    def timed(name, *a):
        if name not in ("szn_conv2d_fwd", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm", "szn_conv2d_dgrad_gemm_native"):
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record()
        ev.append((e0, e1, _conv_flops(a[0]._obj)))
    for mod in mods:
        mod.L.call = timed
    L.call = timed
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        L.call = orig
        for mod in mods:
            mod.L.call = orig
    ms = sum(a.elapsed_time(b) for a, b, _ in ev)
    fl = sum(f for _, _, f in ev)
    return fl / (ms * 1e-3) / 1e12, ms, len(ev)


def _skipped_flops_one_step(L, mods, step_fn, nominal=None):
    """FLOPs of one step that are NOT executed: what the constant-border hint replaces by a broadcast (forward) or a rank-one term (weight
    gradient) -- szn_conv_desc_t.result->work_fraction of every szn_conv2d_fwd / szn_conv2d_wgrad / szn_conv2d_dgrad call -- and, with `nominal` (the
    algorithmic FLOPs of the step without conv1_1, which has its own entry points), also what the band removed from the conv3 block never
    reaches a kernel (round 5: those launches carry smaller descriptors): nominal - sum of the executed FLOPs of every conv call"""
    tot = [0.0, 0.0]
    orig = L.call
    conv_entries = ("szn_conv2d_fwd", "szn_conv2d_wgrad", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm", "szn_conv2d_dgrad_gemm_native",
                    "szn_conv2d_wgrad_adam")

    def hooked(name, *a):
        r = orig(name, *a)
        if name in conv_entries:
            fr = a[0]._obj.res.work_fraction if name in ("szn_conv2d_fwd", "szn_conv2d_wgrad", "szn_conv2d_dgrad") else 1.0
            fl = _conv_flops(a[0]._obj)
            tot[0] += fl * (1.0 - fr)
            tot[1] += fl * fr
        return r
    for mod in mods:
        mod.L.call = hooked
    L.call = hooked
    try:
        step_fn()
    finally:
        L.call = orig
        for mod in mods:
            mod.L.call = orig
    if nominal is not None:
        return max(nominal - tot[1], tot[0])
    return tot[0]


def _init_pg(dist, backend, dev=None, **kw):
    """engine.init_process_group: RCCL's kernels on a high-priority HIP stream unless SZN_RCCL_HIPRI=0"""
    from zeroshotsemanticsegmentation_amd import engine
    return engine.init_process_group(backend, dev, **kw)


def _comm_kwargs(kind):
    import torch
    return {"off": dict(exchange=False),
            "fp32": dict(grad_comm_dtype=torch.float32),
            "bf16": dict(grad_comm_dtype=torch.bfloat16, direct_wire=True),
            "bf16-staged": dict(grad_comm_dtype=torch.bfloat16, direct_wire=False),
            "bf16-sharded": dict(grad_comm_dtype=torch.bfloat16, direct_wire=True, sharded=True),
            "fp32-sharded": dict(grad_comm_dtype=torch.float32, sharded=True)}[kind]


def comm_record_world(args, torch, dist, dev, emb_np, seen, rank, world, headline_kind):
    """N > 1, every rank: the step with the exchange off / fp32 wire / bf16 wire (direct) / bf16 reduce-scatter + sharded Adam, timed
    like the headline (barrier + synchronize on both sides, max over ranks), plus per-bucket issue / wait times from HIP events on the
    compute stream -- so that a SCALE record explains itself: exposed_comm_ms = step(variant) - step(off)."""
    from zeroshotsemanticsegmentation_amd import engine, models, synth
    E, H, B, K = args.embed_dim, args.size, args.batch, args.classes
    x = torch.from_numpy(synth.make_images(B, H, H, seed=1337 + rank)).to(dev)
    t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337 + rank, classes=seen)).to(dev)
    n = max(min(args.steps, 20), 5)
    out = {"world": world, "backend": dist.get_backend(), "steps": n, "headline_wire": headline_kind,
           "rccl_high_priority_stream": os.environ.get("SZN_RCCL_HIPRI", "1") == "1", "ms_per_step": {}, "buckets": {}}
    kinds = ["off", "fp32", "bf16", "bf16-sharded"] if args.precision != "fp32" else ["off", "fp32", "fp32-sharded"]
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.precision]
    for kind in kinds:
        try:
            torch.manual_seed(1337)
            m = models.FCN32s(n_class=E)
            m.load_synthetic(1337, device=dev)
            m.train()
            m._engine.dropout_seed = 1337 + 7919 * rank
            ts = engine.TrainStep(m, emb_np, optimizer="adam", lr=1e-5, precision=dtype, fused_head=True, keep_grads=False,
                                  **_comm_kwargs(kind))
            for _ in range(2):
                ts.step(x, t)
            ts.buckets.timing = kind != "off"
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ts.step(x, t)
            dist.barrier(); torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            out["ms_per_step"][kind] = round(float(dt.item()) / n * 1e3, 3)
            rep = ts.buckets.timing_report()
            if rep:
                out["buckets"][kind] = rep
            if not np.isfinite(float(ts.loss.item())):
                out["ms_per_step"][kind] = "non-finite loss"
            del m, ts
            torch.cuda.empty_cache()
        except Exception as ex:
            out["ms_per_step"][kind] = "failed: %r" % (ex,)
    off = out["ms_per_step"].get("off")
    if isinstance(off, float):
        out["exposed_comm_ms"] = {k: round(v - off, 3) for k, v in out["ms_per_step"].items() if k != "off" and isinstance(v, float)}
        out["cost_frac"] = {k: round(v / off - 1.0, 4) for k, v in out["ms_per_step"].items() if k != "off" and isinstance(v, float)}
    out["note"] = ("rank-0 view of the bucket events (issued_at_ms since the start of the backward pass; wait_ms = how long the compute "
                   "stream stood still for that bucket in front of the optimizer); `off` = no exchange at all (ranks diverge: timing only)")
    return out


def sub_record(args):
    """child process: `fp32` = the headline workload at the reference's arithmetic; `b1` = at the reference's batch size
    (train.py:82-84), eager and replayed from a captured hipGraph (host pacing out).  Prints one JSON object."""
    import torch
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    L.load()
    E, H, K = args.embed_dim, args.size, args.classes
    emb_np, seen, unseen = _workload(args, torch)
    mflop_px = STEP_MFLOP_PER_PX.get(H) if E == 300 else None

    def build(B, dtype):
        torch.manual_seed(1337)
        m = models.FCN32s(n_class=E)
        m.load_synthetic(1337, device=dev)
        m.train()
        ts = engine.TrainStep(m, emb_np, optimizer="adam", lr=1e-5, precision=dtype, fused_head=True, keep_grads=False)
        x = torch.from_numpy(synth.make_images(B, H, H, seed=1337)).to(dev)
        t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337, classes=seen)).to(dev)
        return m, ts, x, t

    def record(B, dtype, ms, extra=None, skipped=0.0):
        peak = PEAK_F32 if dtype == torch.float32 else PEAK_BF16
        r = {"per_gpu_batch": B, "dtype": "f32" if dtype == torch.float32 else "bf16", "ms_per_step": round(ms, 3),
             "value": round(B * H * H / (ms * 1e-3) / 1e6, 3), "unit": "Mpixels/s", "peak_TF": peak}
        if mflop_px:      # executed FLOPs (the constant-border hint skips tiles on the 16-bit path)
            r["step_mfma_frac"] = round((mflop_px * 1e6 * B * H * H - skipped) / (ms * 1e-3) / 1e12 / peak, 4)
        r.update(extra or {})
        return r

    out = {}
    if args.sub_record in ("c1_fcn8s", "c4_768"):
        # BASELINE configs[1] ("PASCAL-VOC FCN8s + 300-d word2vec pixel projection, 512x512, bf16": K = 21 PASCAL matrix, FCN8s skip head --
        # not in the reference, parity unpinned) and configs[4] ("PASCAL-Context 768x768, fp16 activations + fp8 MFMA projection GEMM"), one
        # GPU's share each, through the same engine.TrainStep as the headline
        c1 = args.sub_record == "c1_fcn8s"
        import argparse
        a2 = argparse.Namespace(**vars(args))
        a2.classes, a2.size = (21, 512) if c1 else (59, 768)
        H, K, B = a2.size, a2.classes, args.batch
        emb_np, seen, unseen = _workload(a2, torch)
        dtype = torch.bfloat16 if c1 else torch.float16
        torch.manual_seed(1337)
        m = (models.FCN8s if c1 else models.FCN32s)(n_class=E)
        m.load_synthetic(1337, device=dev)
        m.train()
        if not c1:
            m.set_head_precision("fp8")
        ts = engine.TrainStep(m, emb_np, optimizer="adam", lr=1e-5, precision=dtype, fused_head=True, keep_grads=False)
        x = torch.from_numpy(synth.make_images(B, H, H, seed=1337)).to(dev)
        t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337, classes=seen)).to(dev)
        ms = _time_steps(torch, lambda: ts.step(x, t), max(args.steps, 10), 3)
        # executed MFMA-class FLOPs of one step (every conv entry's descriptor x the fraction of its tiles that ran; conv1_1's two entry
        # points counted dense) and the launches of the band plan
        calls = []
        tot = [0.0]
        orig = L.call
        conv_entries = ("szn_conv2d_fwd", "szn_conv2d_wgrad", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm", "szn_conv2d_dgrad_gemm_native",
                        "szn_conv2d_wgrad_adam")

        def hooked(name, *a):
            r = orig(name, *a)
            calls.append(name)
            if name in conv_entries:
                fr = a[0]._obj.res.work_fraction if name in ("szn_conv2d_fwd", "szn_conv2d_wgrad", "szn_conv2d_dgrad") else 1.0
                tot[0] += _conv_flops(a[0]._obj) * fr
            return r
        L.call = models.L.call = engine.L.call = hooked
        try:
            ts.step(x, t)
            torch.cuda.synchronize()
        finally:
            L.call = models.L.call = engine.L.call = orig
        c11 = 2 * 2.0 * B * (H + 198) ** 2 * 64 * 27
        executed = tot[0] + c11
        out = {"workload": ("BASELINE configs[1]: FCN8s skip head (public definition, not in the reference: parity unpinned) + 300-d pixel "
                            "projection, PASCAL-VOC K=21 matrix, 512x512, bf16, fused stride-8 head" if c1 else
                            "BASELINE configs[4] (one GPU's share): PASCAL-Context K=59 768x768, fp16 activations + dynamic loss scale, fp8 "
                            "(e4m3) projection GEMM, FCN32s, fused-from-coarse head") + ", B=%d, Adam lr 1e-5, train step" % B,
               "per_gpu_batch": B, "dtype": "bf16" if c1 else "f16 (+ fp8 e4m3 projection head)", "ms_per_step": round(ms, 3),
               "value": round(B * H * H / (ms * 1e-3) / 1e6, 3), "unit": "Mpixels/s", "peak_TF": PEAK_BF16,
               "step_gflop_executed": round(executed / 1e9, 1),
               "step_mfma_frac": round(executed / (ms * 1e-3) / 1e12 / PEAK_BF16, 4),
               "band_plan": {"band_remap_launches_per_step": calls.count("szn_band_remap"),
                             "conv1_1_writes_cropped_map": calls.count("szn_conv1_1_fwd_c") > 0,
                             "pool_backward_through_band_map": calls.count("szn_maxpool2x2_ceil_bwd_code_gather")},
               "c_abi_calls_per_step": len(calls), "final_loss": round(float(ts.loss.item()), 5)}
        if not c1 and H in STEP_MFLOP_PER_PX:
            out["step_gflop_algorithmic"] = round(STEP_MFLOP_PER_PX[H] * 1e6 * B * H * H / 1e9, 1)
        print("SUBRECORD " + json.dumps(out))
        return
    if args.sub_record == "comm":
        # BASELINE configs[3]'s exchange on ONE GPU: a one-rank RCCL communicator (RCCL refuses two ranks per device), the step's
        # real gradient buckets forced through librccl on ProcessGroupNCCL's stream (engine.GradBuckets(force=True): pre-multiplied
        # sum, factor 1.0 -> librccl's one-rank reduce kernel reads + writes every bucket while dgrad / wgrad continue).  What a ring
        # all-reduce adds on top of this at N > 1 is the xGMI transfer itself; the stream plumbing, the bucket order, the waits in
        # front of the optimizer and the competition for CUs / HBM with the backward kernels are the same code.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        _init_pg(dist, "nccl", dev, rank=0, world_size=1)
        B = args.batch
        variants = [("comm_off", dict(force_comm=False)),
                    ("fp32_wire", dict(force_comm=True, grad_comm_dtype=torch.float32)),
                    ("bf16_wire", dict(force_comm=True, grad_comm_dtype=torch.bfloat16, direct_wire=True)),
                    ("bf16_wire_staged", dict(force_comm=True, grad_comm_dtype=torch.bfloat16, direct_wire=False)),
                    ("bf16_sharded", dict(force_comm=True, grad_comm_dtype=torch.bfloat16, direct_wire=True, sharded=True))]
        x = torch.from_numpy(synth.make_images(B, H, H, seed=1337)).to(dev)
        t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337, classes=seen)).to(dev)
        steps = {}
        for name, kw in variants:
            torch.manual_seed(1337)
            m = models.FCN32s(n_class=E)
            m.load_synthetic(1337, device=dev)
            m.train()
            steps[name] = engine.TrainStep(m, emb_np, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True, keep_grads=False, **kw)
        n = max(args.steps, 10)
        times = {name: [] for name, _ in variants}
        for rnd in range(3):                                   # interleaved rounds: box drift hits every variant alike
            for name, _ in variants:
                ts = steps[name]
                times[name].append(_time_steps(torch, lambda: ts.step(x, t), n, 2))
        bk = steps["fp32_wire"].buckets
        out = {"workload": "configs[3] per-rank step (bf16, B=%d, %dx%d, E=%d, K=%d) with its gradient buckets exchanged through a "
                           "one-rank RCCL communicator on this GPU (forced; see DESIGN.md section 5)" % (B, H, H, E, K),
               "backend": dist.get_backend(), "world": 1, "steps_per_round": n, "rounds": 3,
               "bucket_mib": [round((e - o) * 4 / 2 ** 20, 1) for o, e, _ in bk.buckets],
               "allreduce_calls_per_step": len(bk.buckets) + 1,
               "rccl_kernel": "one-rank reduce (pre-multiplied sum x 1.0) over every bucket on RCCL's stream",
               "rccl_high_priority_stream": os.environ.get("SZN_RCCL_HIPRI", "1") == "1",
               "variants": "bf16_wire = the weight-gradient kernels write the bf16 wire image, RCCL sums it in place, Adam reads it "
                           "(szn_adam_step_g16); bf16_wire_staged = round 4's two staging copies; bf16_sharded = the same wire with the "
                           "rank-sharded optimizer (one rank owns every slice here)",
               "ms_per_step": {k: round(sorted(v)[1], 3) for k, v in times.items()},
               "rounds_ms": {k: [round(a, 3) for a in v] for k, v in times.items()}}
        off = out["ms_per_step"]["comm_off"]
        out["overlap_cost_frac"] = {k: round(v / off - 1.0, 4) for k, v in out["ms_per_step"].items() if k != "comm_off"}
        dist.destroy_process_group()
        print("SUBRECORD " + json.dumps(out))
        return
    if args.sub_record == "fp32":
        B = args.batch
        m, ts, x, t = build(B, torch.float32)
        ms = _time_steps(torch, lambda: ts.step(x, t), max(args.steps // 3, 3), 2)
        tf, fam_ms, n = _conv_family(torch, L, (models, engine), lambda: ts.step(x, t))
        nominal = (mflop_px * 1e6 * B * H * H - 2 * 2.0 * B * (H + 198) ** 2 * 64 * 27) if mflop_px else None
        sk = _skipped_flops_one_step(L, (models, engine), lambda: ts.step(x, t), nominal) if nominal else 0.0
        out = record(B, torch.float32, ms, {
            "workload": "the headline step at the reference's arithmetic: fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 "
                        "products, the parity-gated path), B=%d, %dx%d, E=%d, K=%d" % (B, H, H, E, K),
            "roofline": {"bound": "mfma", "kernels": "conv forward + dgrad launches of one step (HIP events)", "achieved": round(tf, 2),
                         "peak": PEAK_F32, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32, 4), "launches": n,
                         "ms": round(fam_ms, 3)},
            "final_loss": round(float(ts.loss.item()), 5), "step_gflop_not_executed": round(sk / 1e9, 1)}, skipped=sk)
    else:
        for dtype in (torch.bfloat16, torch.float32):
            key = "bf16" if dtype == torch.bfloat16 else "fp32"
            m, ts, x, t = build(1, dtype)
            steps = 20 if dtype == torch.bfloat16 else 8
            ms = _time_steps(torch, lambda: ts.step(x, t), steps, 3)
            nominal = (mflop_px * 1e6 * H * H - 2 * 2.0 * (H + 198) ** 2 * 64 * 27) if mflop_px else None
            skipped = _skipped_flops_one_step(L, (models, engine), lambda: ts.step(x, t), nominal)
            rec = record(1, dtype, ms, skipped=skipped)
            rec["eager_ms_per_step"] = rec.pop("ms_per_step")
            rec["eager_value"] = rec.pop("value")
            eager_frac = rec.pop("step_mfma_frac", None)
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ts.step(x, t)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ts.step(x, t)
                gms = _time_steps(torch, g.replay, steps, 3)
                grec = record(1, dtype, gms, skipped=skipped)
                rec.update({"graph": True, "graph_ms_per_step": grec["ms_per_step"]})
                # the record's ms_per_step = the faster of the two forms (eager overlaps the weight gradients of a small step with the
                # dgrad chain on a second stream, which a captured graph does not: DESIGN.md section 8)
                best = grec if gms <= ms else record(1, dtype, ms, skipped=skipped)
                rec.update({"ms_per_step": best["ms_per_step"], "value": best["value"], "form": "graph replay" if gms <= ms else "eager"})
                if "step_mfma_frac" in best:
                    rec["step_mfma_frac"] = best["step_mfma_frac"]
                rec["graph_note"] = ("one train step captured with torch.cuda.graph (every launch goes through the C-ABI on the "
                                     "capture stream) and replayed: all kernels of the step run, with the capture-time scalars "
                                     "(Adam step count, Dropout2d counter)")
                if not np.isfinite(float(ts.loss.item())):
                    rec["graph"] = "non-finite loss after replay"
            except Exception as ex:
                rec.update({"graph": False, "graph_error": repr(ex)[:300], "ms_per_step": rec["eager_ms_per_step"],
                            "value": rec["eager_value"]})
                if eager_frac is not None:
                    rec["step_mfma_frac"] = eager_frac
            if eager_frac is not None:
                rec["eager_step_mfma_frac"] = eager_frac
            out[key] = rec
            del m, ts, x, t
            torch.cuda.empty_cache()
        out["workload"] = ("the reference's batch size (train.py:82-84: one image per step), %dx%d, E=%d, K=%d; at B=1 fc6 / fc7 "
                           "stream their weights for 289 pixels and the host enqueues as fast as the GPU executes, hence the "
                           "hipGraph replay beside the eager number" % (H, H, E, K))
    print("SUBRECORD " + json.dumps(out))


def run_sub_record(kind, args, timeout=240):
    """run `bench.py --sub-record kind` as a child (the GPU is idle here: the parent has finished measuring)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sub-record", kind, "--batch", str(args.batch), "--size", str(args.size),
           "--embed-dim", str(args.embed_dim), "--classes", str(args.classes), "--steps", str(args.steps)]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        for line in p.stdout.splitlines():
            if line.startswith("SUBRECORD "):
                return json.loads(line[len("SUBRECORD "):])
        return {"error": "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])}
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def main():
    args = parse()
    if args.sub_record:
        return sub_record(args)
    import torch
    import torch.distributed as dist
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth, trainer_fcn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: SZN_TEST_ONE_GPU=1 runs every rank on device 0 over gloo, so that the N > 1 code path can be exercised on a
    # 1-GPU box (tests/test_gpu_bench_contract.py); the measured line of a real run always uses RCCL, one GPU per rank
    one_gpu = os.environ.get("SZN_TEST_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: re-exec under torch.distributed.run, one rank per GPU on this node
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d: the rank count must match" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            _init_pg(dist, "nccl", dev)                       # backend "nccl" is RCCL on ROCm
    L.load()

    E, H, B, K = args.embed_dim, args.size, args.batch, args.classes
    emb_np, seen, unseen = _workload(args, torch)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.precision]
    peak = PEAK_F32 if dtype == torch.float32 else PEAK_BF16          # fp16 and bf16 MFMA share the dense peak

    torch.manual_seed(1337)                                   # identical initial weights on every rank
    model = (models.FCN8s if args.arch == "fcn8s" else models.FCN32s)(n_class=E)
    model.load_synthetic(1337, device=dev)
    model.train()
    if args.head_fp8:
        model.set_head_precision("fp8")
    model._engine.dropout_seed = 1337 + 7919 * rank           # ranks draw different Dropout2d masks
    x = torch.from_numpy(synth.make_images(B, H, H, seed=1337 + rank)).to(dev)
    # phase-1 batches contain seen classes only (the reference's train_seen split, context_dataset.py:75-94)
    target = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337 + rank, classes=seen)).to(dev)
    target_all = torch.from_numpy(synth.make_labels(B, H, H, K, seed=4337 + rank)).to(dev)

    # N > 1 headline: the direct bf16 wire (per-rank gradients rounded once to bf16, summed by RCCL in bf16; 271 MB instead of 542 MB on the links).
    # The trainer's default is the fp32 wire (SZN_GRAD_COMM=fp32: exact sums, the reference-equivalent numerics) -- `config.grad_wire` on the line says
    # which one was measured, `comm` carries both, and `--grad-comm fp32` / SZN_GRAD_COMM=bf16 move either side.
    wire = args.grad_comm
    if wire == "auto":
        wire = "bf16" if (dtype == torch.bfloat16 and args.arch == "fcn32s") else "fp32"
    if dtype == torch.float32 and wire.startswith("bf16"):
        wire = "fp32"

    def make_phase1():
        # keep_grads=False like trainer_fcn.Trainer: the loop calls zero_grad() next (train.py:170-175), so fc6 / fc7's gradients --
        # consumed by the Adam step inside their weight-gradient kernel on one rank -- are not also written out
        return engine.TrainStep(model, emb_np, optimizer="adam", lr=1e-5, precision=dtype, keep_grads=False,
                                fused_head=not (args.unfused_head and args.arch == "fcn32s"), **_comm_kwargs(wire))

    def make_phase1_fcn8s():
        # autograd path: forward (skip head, materialised score) -> cosine loss -> infer_lbl -> backward -> two-group Adam
        # (train.py:126-133 wiring: Conv2d weights at lr, biases at 2 lr; the bilinear ConvTranspose2d kernels stay fixed)
        from zeroshotsemanticsegmentation_amd import optim as szn_optim, utils as szn_utils
        model.set_precision(dtype)
        emb = torch.from_numpy(emb_np).to(dev)
        ws = [m_.weight for n_, m_ in model.named_modules() if isinstance(m_, torch.nn.Conv2d) and not n_.startswith("seenmask")]
        bs = [m_.bias for n_, m_ in model.named_modules() if isinstance(m_, torch.nn.Conv2d) and not n_.startswith("seenmask")]
        for n_, p_ in model.named_parameters():
            p_.requires_grad = not (n_.startswith("seenmask") or n_.startswith("upscore"))
        opt = szn_optim.FusedAdam([{"params": ws}, {"params": bs, "lr": 2e-5}], lr=1e-5)

        class _Phase1(object):
            def step(self, xx, tt):
                score = model(xx)
                loss = szn_utils.cosine_loss(score, tt, emb)
                pred = szn_utils.infer_lbl_device(score.detach(), emb)
                opt.zero_grad()
                loss.backward()
                if world > 1:
                    engine.allreduce_param_grads(ws + bs)
                opt.step()
                return loss.detach(), pred
        return _Phase1()

    def make_phase2():
        # train.py:164-175: everything frozen except seenmask_score (w, b) and seenmask_upscore (w); binary target
        # "label is a seen class" with unlabelled pixels = 0 (trainer_seenmask.py:55-56); 2-class CE, size_average=True;
        # phase 2 trains on train_unseen (two of the unseen classes) vs everything else.  engine.SeenmaskStep: no autograd,
        # no (B,2,H,W) score, head-only flat Adam
        return engine.SeenmaskStep(model, K, unseen[:2], lr=1e-3, precision=dtype)

    if args.phase == "seenmask" and world > 1:
        raise SystemExit("--phase seenmask is a single-GPU line (98 KB of gradients)")
    if args.arch == "fcn8s" and args.head_fp8:
        raise SystemExit("--arch fcn8s has no fp8 head")
    # FCN8s: engine.TrainStep runs its fused stride-8 head; --unfused-head = the autograd path with the materialised score
    ts = (make_phase1_fcn8s() if (args.arch == "fcn8s" and args.unfused_head) else make_phase1()) if args.phase == "fcn" else make_phase2()
    wire_note = None
    if world > 1 and args.phase == "fcn" and wire != "fp32" and not (args.arch == "fcn8s" and args.unfused_head):
        # insurance for the first run on a real node: if the 16-bit / sharded exchange fails where this build could not test it (one-GPU
        # boxes only), the headline falls back to the plain fp32 all-reduce instead of losing the line; every rank sees the same error
        err = None
        try:
            ts.step(x, target)
            torch.cuda.synchronize()
        except Exception as ex:
            err = ex
        # every rank takes the same decision (a fallback decided per rank leaves the others inside a collective of the old
        # configuration): the failure flags are summed over a fresh fp32 all-reduce before anybody rebuilds its TrainStep
        flag = torch.tensor([1.0 if err is not None else 0.0], device=dev)
        try:
            dist.all_reduce(flag)
        except Exception:
            flag.fill_(1.0)
        if float(flag.item()) > 0:
            wire_note = "grad wire %s failed on %d rank(s) (%r): fell back to fp32" % (wire, int(flag.item()), err)
            sys.stderr.write(wire_note + "\n")
            wire = "fp32"
            ts = make_phase1()

    # ---- HIP events on the launch stream around C-ABI calls (torch's current stream IS the stream handed to the C-ABI) ----
    CONV_ENTRIES = ("szn_conv2d_fwd", "szn_conv2d_dgrad", "szn_conv2d_dgrad_gemm", "szn_conv2d_dgrad_gemm_native")
    events = []                      # (e0, e1, entry, kernel, bound, work)
    mode = ["off"]                   # off | dominant (timed region: conv fwd/dgrad calls only) | all (instrumented pass)
    orig_call = L.call

    # Every event pair costs GPU time (instrumenting all ~35 conv launches of a step measured 2.5-5 % of `value`), so the timed
    # region instruments the launches of the DOMINANT kernel only: the last warmup step times every conv forward / dgrad call
    # ("learn"), the kernel with the largest total is the dominant one, and the ordinals of its calls within a step are kept.
    skipped = [0.0]                  # FLOPs per step the constant-border hint does not execute
    ordn = [0]                       # ordinal of the next conv forward / dgrad call within the current step
    dom_ord = [None]                 # ordinals of the dominant kernel's calls (None: instrument every conv call)

    def timed_call(name, *a):
        k = -1
        if name in CONV_ENTRIES:
            k = ordn[0]; ordn[0] += 1
        if mode[0] == "off" or (mode[0] in ("dominant", "learn") and name not in CONV_ENTRIES) or \
                (mode[0] == "dominant" and dom_ord[0] is not None and k not in dom_ord[0]):
            return orig_call(name, *a)
        wk = call_work(name, a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        kern = L.last_kernel()
        if name in ("szn_conv2d_fwd", "szn_conv2d_wgrad", "szn_conv2d_dgrad") and wk:      # constant-border hint: only the executed tiles count as FLOPs
            wk = (wk[0], wk[1] * a[0]._obj.res.work_fraction)
        helper = {"splitk_epilogue": "+splitk", "col2im_kernel": "+col2im", "maxpool_fwd_kernel": "+maxpool",
                  "wgrad_taps_reduce": "+reduce", "conv1_1_wgrad_reduce": "+reduce"}.get(kern)
        if kern == "wgrad_taps_reduce":                            # (under the constant-border hint two column-sum launches sit in between)
            kern = "conv_wgrad_taps+reduce"
        elif kern == "slab_rows_sum_kernel":                       # szn_maxpool2x2_ceil_bwd_code_cb: the pool's backward + its skipped-tile sums
            kern = "maxpool_bwd_code_kernel"
        elif helper:
            kern = L.prev_kernel() + helper
        elif name in ("szn_fused_head", "szn_fused_head_strided"):
            kern = "fused_head (fh_prep + fh_cell + fh_finalize + fh_gather)"
        events.append((e0, e1, name, kern, wk[0] if wk else None, wk[1] if wk else 0.0, k))
    if not args.no_kernel_events:
        L.call = timed_call
        for mod in (models, engine):
            mod.L.call = timed_call

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    learn_events = []
    if args.phase == "fcn":          # one untimed step that adds up what the hint skips (every szn_conv2d_fwd call reports its fraction)
        nominal = None
        if H in STEP_MFLOP_PER_PX and E == 300 and args.arch == "fcn32s":
            c11 = 2 * 2.0 * B * (H + 198) ** 2 * 64 * 27                 # conv1_1 forward + weight gradient (own entry points)
            nominal = STEP_MFLOP_PER_PX[H] * 1e6 * B * H * H - c11
        skipped[0] = _skipped_flops_one_step(L, (models, engine), lambda: ts.step(x, target), nominal)
    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_kernel_events:
            torch.cuda.synchronize()
            mode[0] = "learn"
        ordn[0] = 0
        loss, pred = ts.step(x, target)
    if mode[0] == "learn":
        torch.cuda.synchronize()
        mode[0] = "off"
        learn_events, events = events, []
        tot = {}
        for e0, e1, name, kern, bound, work, k in learn_events:
            tot[kern] = tot.get(kern, 0.0) + e0.elapsed_time(e1)
        if tot:
            dom_k = max(tot, key=tot.get)
            dom_ord[0] = {k for (_, _, _, kern, _, _, k) in learn_events if kern == dom_k}
    import gc
    gc.collect()
    gc.disable()                         # no collector pause inside the timed region (re-enabled right behind it)
    sync()
    mode[0] = "dominant"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ordn[0] = 0
        loss, pred = ts.step(x, target)
    sync()
    dt = time.perf_counter() - t0
    mode[0] = "off"
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lossv = float(loss.item())
    if not np.isfinite(lossv):
        raise SystemExit("loss is not finite: %r" % lossv)
    timed_events, events = events, []

    out = None
    if rank == 0:
        mpx = world * B * H * H * args.steps / dt / 1e6
        workload = ("PASCAL-Context 512x512 phase 1" if (K == 59 and H == 512) else "BASELINE configs[1]" if K == 21 else "custom")
        out = {
            "metric": "train_Mpixels_per_sec", "value": round(mpx, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16": "f16 (+ fp8 e4m3 projection head)" if args.head_fp8 else "f16", "fp32": "f32"}[args.precision]
            if not (args.head_fp8 and args.precision == "bf16") else "bf16 (+ fp8 e4m3 projection head)", "data": "synthetic",
            "config": {"workload": ("%s: %s + %d-d pixel projection, %dx%d, K=%d (%d seen / %d "
                                    "unseen), Adam lr 1e-5, train step fwd+cosine loss+infer_lbl+bwd+optimizer"
                                    % (workload, "FCN32s (the reference has no FCN8s)" if args.arch == "fcn32s" else
                                       "FCN8s skip head (public definition, not in the reference: parity unpinned)", E, H, H, K,
                                       len(seen), len(unseen))) if args.phase == "fcn" else
                       ("BASELINE configs[2] (phase 2): seen-mask head on the frozen FCN32s backbone, %dx%d, K=%d, 2-class CE, "
                        "train step fwd+CE+argmax+head bwd+Adam" % (H, H, K)),
                       "per_gpu_batch": B, "global_batch": B * world,
                       "head": ("unfused" if args.unfused_head else "fused-from-coarse") if args.phase == "fcn"
                       else "seenmask_score + learned 64x64 s32 deconv",
                       "parallelism": "dp%d" % world, "final_loss": round(lossv, 5)},
        }
        if timed_events:
            by = {}
            for e0, e1, name, kern, bound, work, _k in timed_events:
                r = by.setdefault(kern, [0.0, 0.0, 0])
                r[0] += e0.elapsed_time(e1); r[1] += work; r[2] += 1
            dom = max(by, key=lambda k: by[k][0])
            ms, fl, n = by[dom]
            ach = fl / (ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if fn.endswith("_traffic.json"):
                    tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    rec = tj.get("kernels", {}).get(dom)
                    if rec and tj.get("per_gpu_batch") == B and tj.get("precision") == args.precision and tj.get("size") == H \
                            and tj.get("classes", K) == K:
                        traffic, tsrc = round(rec["hbm_bytes_per_launch"]), "profiles/" + fn
                        break
            # all conv forward / dgrad launches: from the learn step when the timed region held the dominant kernel only
            fam = learn_events if (learn_events and dom_ord[0] is not None) else timed_events
            fam_steps = 1 if fam is learn_events else args.steps
            fam_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in fam)
            fam_fl = sum(ev[5] for ev in fam)
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                               "launches_per_step": n // args.steps, "avg_launch_ms": round(ms / n, 4),
                               "gflop_per_launch": round(fl / n / 1e9, 2), "share_of_step": round(ms / (dt * 1e3), 3),
                               "conv_fwd_dgrad_family": {"achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 2),
                                                         "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / peak, 4),
                                                         "launches_per_step": len(fam) // fam_steps,
                                                         "share_of_step": round(fam_ms / fam_steps / (dt / args.steps * 1e3), 3),
                                                         "source": "last warmup step (events on every conv launch)"
                                                         if fam is learn_events else "timed region"},
                               "events": "HIP events around every launch of the dominant kernel inside the timed region "
                                         "(%d per step)" % (n // args.steps)}
            if H in STEP_MFLOP_PER_PX and E == 300 and args.phase == "fcn" and args.arch == "fcn32s":
                # executed FLOPs: the dense algorithmic count minus the tiles the constant-border hint replaced by a broadcast
                step_fl = STEP_MFLOP_PER_PX[H] * 1e6 * B * H * H - skipped[0]
                out["roofline"]["step_mfma_frac"] = round(step_fl * args.steps / dt / 1e12 / peak, 4)
                out["roofline"]["step_gflop_executed"] = round(step_fl / 1e9, 1)
                out["roofline"]["step_gflop_skipped_constant_border"] = round(skipped[0] / 1e9, 1)   # hints + the band removed from conv3_x

    if world > 1 and args.phase == "fcn" and args.arch == "fcn32s" and not args.unfused_head and not args.no_extras:
        if out is not None:
            out["config"]["grad_wire"] = wire
            if wire_note:
                out["config"]["grad_wire_note"] = wire_note[:300]
        L.call = orig_call
        for mod in (models, engine):
            mod.L.call = orig_call
        del ts
        torch.cuda.empty_cache()
        try:
            rec = comm_record_world(args, torch, dist, dev, emb_np, seen, rank, world, wire)
        except Exception as ex:
            rec = {"error": repr(ex)[:400]}
        if out is not None:
            out["comm"] = rec

    # ---- instrumented pass: every C-ABI call of 3 more steps (same state, not part of `value`) ----
    if not args.no_kernel_events and not args.no_extras and world == 1:
        NI = 3
        mode[0] = "all"
        torch.cuda.synchronize()
        for _ in range(NI):
            ts.step(x, target)
        torch.cuda.synchronize()
        mode[0] = "off"
        by = {}
        for e0, e1, name, kern, bound, work, _k in events:
            key = (kern or name, bound)
            r = by.setdefault(key, [0.0, 0.0, 0, name])
            r[0] += e0.elapsed_time(e1); r[1] += work; r[2] += 1
        rows = []
        for (kern, bound), (ms, work, n, entry) in sorted(by.items(), key=lambda kv: -kv[1][0]):
            row = {"kernel": kern, "entry": entry, "calls_per_step": round(n / NI, 2), "ms_per_step": round(ms / NI, 4)}
            if bound == "mfma":
                a = work / (ms * 1e-3) / 1e12
                row.update({"bound": "mfma", "gflop_per_step": round(work / NI / 1e9, 1), "achieved": round(a, 1),
                            "unit": "TFLOP/s", "frac": round(a / peak, 4)})
            elif bound == "hbm":
                a = work / (ms * 1e-3) / 1e9
                row.update({"bound": "hbm", "mbytes_per_step": round(work / NI / 1e6, 1), "achieved": round(a, 1),
                            "unit": "GB/s", "frac": round(a / PEAK_HBM, 4)})
            rows.append(row)
        tot = sum(r["ms_per_step"] for r in rows)
        out["kernels"] = {"note": "HIP events around every C-ABI call of %d extra steps after the timed region (host-paced: the sum of "
                                  "the rows, not the wall time, is meaningful); kernels launched by torch itself (fills, "
                                  "copies: ~0.2 ms/step) are in profiles/, not here" % NI,
                          "c_abi_ms_per_step": round(tot, 3),
                          "mfma_class_ms": round(sum(r["ms_per_step"] for r in rows if r.get("bound") == "mfma"), 3),
                          "hbm_class_ms": round(sum(r["ms_per_step"] for r in rows if r.get("bound") == "hbm"), 3),
                          "rows": rows}
        events = []

    if rank == 0 and world == 1 and not args.no_extras:
        if dtype != torch.float32:
            try:
                out["projection"] = projection_report(L, torch, out.get("roofline", {}).get("step_mfma_frac"))
            except Exception as ex:
                out["projection"] = {"error": repr(ex)}
        if args.phase == "fcn":
            try:
                del ts                                         # phase 1 is over: its flat buffers keep the parameters alive
                p2 = make_phase2()
                eng = model._engine
                n2 = max(args.steps, 10)
                for _ in range(3):
                    l2, _ = p2.step(x, target_all)
                l_first = float(l2.item())
                reps = [_time_steps(torch, lambda: p2.step(x, target_all), n2, 1) for _rep in range(3)]
                l_last = float(p2.loss.item())
                d2 = sorted(reps)[1]                           # median of three repeats
                fwd_ms = _time_steps(torch, lambda: eng.forward(x, train=True, keep=False), n2, 2)
                # head kernels: events around the calls behind the backbone (one extra step)
                hev = []
                orig2 = L.call

                def head_timed(name, *a):
                    if not (name.startswith("szn_seenmask") or name == "szn_adam_step"):
                        return orig2(name, *a)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); orig2(name, *a); e1.record()
                    hev.append((name, e0, e1))
                for mod in (models, engine):
                    mod.L.call = head_timed
                try:
                    p2.step(x, target_all)
                    torch.cuda.synchronize()
                finally:
                    for mod in (models, engine):
                        mod.L.call = orig2
                head = {}
                for name, e0, e1 in hev:
                    head[name] = round(head.get(name, 0.0) + e0.elapsed_time(e1), 4)
                fwd_gf = {512: 380.01, 768: 724.28}.get(H) if E == 300 else None
                out["phase2"] = {"workload": "BASELINE configs[2]: seen-mask head on the frozen backbone, %dx%d, K=%d (train_unseen "
                                             "= 2 classes), engine.SeenmaskStep: backbone forward (train mode, nothing kept) + fused "
                                             "deconv / 2-class CE / argmax / head backward from the 1/32 map + Adam lr 1e-3" % (H, H, K),
                                 "value": round(B * H * H / (d2 * 1e-3) / 1e6, 3), "unit": "Mpixels/s", "steps": n2,
                                 "ms_per_step": round(d2, 3), "repeats_ms_per_step": [round(r, 3) for r in reps],
                                 "repeat_spread": round((max(reps) - min(reps)) / d2, 4),
                                 "backbone_forward_only_ms": round(fwd_ms, 3), "head_ms": head,
                                 "loss_first_to_last": [round(l_first, 5), round(l_last, 5)]}
                if fwd_gf:
                    tf2 = fwd_gf * 1e9 * B / (d2 * 1e-3) / 1e12
                    out["phase2"]["roofline"] = {"bound": "mfma", "achieved": round(tf2, 1), "peak": peak, "unit": "TFLOP/s",
                                                 "frac": round(tf2 / peak, 4),
                                                 "note": "forward MFMA FLOPs (%.2f GF/image, SURVEY 8-d) over the whole step time; "
                                                         "the head is 16 B/px of labels in / predictions out" % fwd_gf}
            except Exception as ex:
                out["phase2"] = {"error": repr(ex)}
        if args.phase == "fcn" and args.arch == "fcn32s" and not args.unfused_head:
            out["fp32"] = run_sub_record("fp32", args)
            out["b1"] = run_sub_record("b1", args)
            if E == 300:
                out["c1_fcn8s"] = run_sub_record("c1_fcn8s", args)
                out["c4_768"] = run_sub_record("c4_768", args)
            if dtype == torch.bfloat16:
                out["comm"] = run_sub_record("comm", args)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(E, K, H, emb_np, args.arch)
            except Exception as ex:      # the baseline is a reported extra: never lose the measured line
                out["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
