"""engine.py -- the fused training step of the SZN path (what bench.py measures and the trainers run).

One `TrainStep.step(x, target)` is the reference's hot loop body (trainer_fcn.py:149-180):
    forward (models.py:114-160, train mode)  ->  cosine loss (utils.py:75-102)  ->  train-time
    infer_lbl (utils.py:159-185)  ->  backward  ->  [gradient all-reduce over RCCL]  ->  Adam / SGD step
    (train.py:126-133)  ->  confusion histogram for the running metrics (utils.py:104-154)
without autograd bookkeeping: parameters, gradients and optimizer moments live in flat fp32 buffers (the
module's Parameters are channels_last views into them), the head runs fused from the 1/32 map
(szn_fused_head) and the per-layer gradient buckets are all-reduced on RCCL's stream while the rest of the
backward pass is still running.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import synth
from . import models as _models
from .models import CROP, CROP_POOL3, CROP_POOL4, CROP_UP8, opt_layers


def init_process_group(backend="nccl", device=None, **kw):
    """torch.distributed.init_process_group for the data-parallel step.  On the `nccl` backend (RCCL on ROCm) the collectives go
    to a HIGH-PRIORITY HIP stream (ProcessGroupNCCL.Options.is_high_priority_stream; SZN_RCCL_HIPRI=0 turns it off): the tile
    kernels of the backward pass hold 128-156 KiB of every CU's LDS and keep thousands of workgroups pending, and at normal
    priority RCCL's workgroups only got CUs when a compute kernel had drained -- measured with the one-rank communicator
    (profiles/r05_comm_hipri.json): fp32 wire +10.7 % per step at normal priority, +6.1 % at high priority; bf16 wire +10.9 % ->
    +2.4 %.  The reference is single-GPU (train.py:58,82-84) and has no counterpart."""
    if backend == "nccl" and os.environ.get("SZN_RCCL_HIPRI", "1") == "1":
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            return dist.init_process_group(backend, device_id=device, pg_options=opts, **kw)
        except (AttributeError, TypeError):
            pass
    if backend == "nccl":
        return dist.init_process_group(backend, device_id=device, **kw)
    return dist.init_process_group(backend, **kw)


class GradBuckets(object):
    """Data-parallel exchange of the flat weight gradient: contiguous buckets in BACKWARD completion order.

    `layers` = [(name, offset, count)] in FORWARD order inside `flat`; backward finishes them last-to-first, so the
    tail of `flat` becomes final first.  A bucket is closed once it holds >= bucket_elems elements (or at the first
    layer) and its sum all-reduce is issued asynchronously (RCCL's own stream; gloo in the CPU tests) as soon as the
    bucket's earliest layer reports `layer_done`; `finish` waits for everything.  The 1/world scaling is applied by the
    optimizer kernel (grad_scale), not here.

    comm_dtype=torch.bfloat16 halves the bytes on the xGMI links (271 MB instead of 542 MB per step at E = 300).  Two forms:
      staged (direct=False)  the fp32 bucket is rounded into a 16-bit staging buffer, summed there and widened back into `flat`
                             by `finish` (two extra passes over the gradient; `.grad` stays fp32 and holds the sum);
      direct (direct=True)   the weight-gradient kernels write the 16-bit image themselves (szn_conv_desc_t.dw_lp) into
                             `self.stage`, the all-reduce sums it in place and the optimizer kernel reads it
                             (szn_adam_step_g16): no copy in, no copy out, `flat` is never touched.
    sharded=True: every bucket is reduce-scattered instead of all-reduced -- rank r then owns the summed slice
    `shard(bucket)` of it, runs the optimizer over that slice only (1/world of the pass) and `gather_weights` all-gathers the
    updated weight image (16-bit on the 16-bit paths: the same bytes on the links as a 16-bit all-reduce, 25 % fewer than an
    fp32 one).  enabled=False: no exchange at all (bench.py's comm-off timing at world > 1)."""

    def __init__(self, flat, layers, bucket_elems, extra=(), group=None, comm_dtype=torch.float32, force=None, enabled=True,
                 direct=False, sharded=False, tail=0):
        """force (default: SZN_FORCE_COMM=1): issue the bucket all-reduces even in a process group of ONE rank -- the whole
        exchange path (bucket slicing, RCCL's stream, the waits in front of the optimizer, the bf16 staging) then runs on a
        single GPU.  RCCL returns from an in-place sum over one rank without touching the device, so on the `nccl` backend the
        forced single-rank exchange uses the pre-multiplied sum (factor 1.0; AVG for 16-bit wire buffers): librccl's one-rank
        reduce kernel reads and writes every bucket on RCCL's stream while dgrad / wgrad keep running on the compute stream (same
        bits as no exchange).  tail: elements behind `flat` the staging buffer also needs (the fused head's padding rows)."""
        self.flat, self.extra, self.group = flat, list(extra), group
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.rank = dist.get_rank(group) if ready else 0
        if force is None:
            force = os.environ.get("SZN_FORCE_COMM", "0") == "1"
        self.active = bool(enabled) and ready and (self.world > 1 or bool(force))
        self.op = dist.ReduceOp.SUM
        if self.active and self.world == 1 and dist.get_backend(group) == "nccl":
            # one rank: sum == average == x * 1.0.  RCCL skips an in-place SUM entirely; the pre-multiplied sum (fp32 buckets) and
            # AVG (16-bit staging buffers: torch 2.10 hands RCCL a zero factor for a bf16 pre-multiplied sum) make it launch its
            # one-rank reduce kernel.
            which = "premul" if comm_dtype == torch.float32 else "avg"
            if which == "premul" and hasattr(dist, "_make_nccl_premul_sum"):
                self.op = dist._make_nccl_premul_sum(1.0)
            elif which == "avg":
                self.op = dist.ReduceOp.AVG
        self.buckets = []            # (start, end, name of the layer whose completion closes the bucket)
        end = None
        for name, off, cnt in reversed(layers):
            if end is None:
                end = off + cnt
            if end - off >= bucket_elems or name == layers[0][0]:
                self.buckets.append((off, end, name))
                end = None
        self.ready_after = {name: (o, e) for o, e, name in self.buckets}
        self.works = []
        self.issued = 0              # collective calls handed to the backend so far (tests / bench)
        self.comm_dtype = comm_dtype
        self.stage = None
        self.direct = bool(direct) and self.active and comm_dtype != torch.float32
        self.sharded = bool(sharded) and self.active
        if self.sharded and any((e - o) % (4 * self.world) for o, e, _ in self.buckets):
            raise L.SznError("sharded optimizer: every bucket must split into %d slices of whole 16-byte groups" % self.world)
        if self.active and comm_dtype != torch.float32:
            self.stage = torch.zeros(flat.numel() + tail, dtype=comm_dtype, device=flat.device)
        # self-diagnosis (bench.py): per step, when each bucket was handed to the backend and how long the compute stream then
        # stood still for it in finish() -- HIP events on the compute stream, read back by `timing_report`
        self.timing = False
        self._t0 = None
        self._tlog = []

    # ---- what the exchange works on ------------------------------------------------------------------
    def wire(self):
        """the buffer the collectives run on (the 16-bit staging buffer, or `flat` itself)"""
        return self.stage if self.stage is not None else self.flat

    def shard(self, o, e):
        """the slice [lo, hi) of bucket [o, e) whose sum this rank owns after a reduce-scatter"""
        n = (e - o) // self.world
        return o + self.rank * n, o + (self.rank + 1) * n

    def _emulate(self, t):
        """gloo has no reduce-scatter / all-gather-into-tensor for device memory (the one-GPU test harness runs CUDA tensors over
        gloo): an all-reduce of the whole bucket leaves the right sum in this rank's slice, a list all-gather moves the slices"""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _collective(self, t, o, e):
        self.issued += 1
        if self.sharded and self.world > 1 and not self._emulate(t):
            lo, hi = self.shard(o, e)
            return dist.reduce_scatter_tensor(t[lo:hi], t[o:e], op=self.op, group=self.group, async_op=True)
        # (sharded, one rank: it owns everything -- the forced single-rank exchange keeps the all-reduce kernel)
        return dist.all_reduce(t[o:e], op=self.op, group=self.group, async_op=True)

    def begin_step(self):
        if self.timing and self.active:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()
            self._cur = []

    def layer_done(self, name):
        if self.active and name in self.ready_after:
            o, e = self.ready_after[name]
            ev = None
            if self.timing and self._t0 is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
            if self.stage is None:
                self.works.append((self._collective(self.flat, o, e), None, name, ev))
            else:
                if not self.direct:
                    self.stage[o:e].copy_(self.flat[o:e])
                self.works.append((self._collective(self.stage, o, e), (o, e), name, ev))

    def finish(self):
        if self.active:
            for t in self.extra:
                self.issued += 1
                self.works.append((dist.all_reduce(t, op=self.op, group=self.group, async_op=True), None, "bias", None))
            for wk, span, name, ev in self.works:
                tm = self.timing and self._t0 is not None
                if tm:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                wk.wait()
                if tm:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._cur.append((name, ev, e0, e1))
                if span is not None and not self.direct:
                    o, e = self.shard(*span) if (self.sharded and self.world > 1) else span
                    self.flat[o:e].copy_(self.stage[o:e])
            if self.timing and self._t0 is not None:
                self._tlog.append((self._t0, self._cur))
                self._t0 = None
        self.works = []

    def gather_weights(self, image, first_only=False, spans=False):
        """sharded optimizer: all-gather the slices of `image` (a tensor laid out like `flat`: the 16-bit weight image, or the
        fp32 masters) that the ranks just updated; returns the async works (wait before anything reads the gathered range: the next
        forward pass waits bucket by bucket, TrainStep.wait_weights).  first_only: the bucket of the first layers only (conv1_1 reads
        its fp32 master in the forward pass).  spans=True: [((start, end) of the bucket, work)] instead of the bare works."""
        works = []
        if self.sharded and self.world > 1:
            for o, e, _ in list(reversed(self.buckets))[:1 if first_only else None]:       # forward order: conv1_1's bucket first
                lo, hi = self.shard(o, e)
                self.issued += 1
                if self._emulate(image):
                    n = (e - o) // self.world
                    wk = dist.all_gather([image[o + r * n:o + (r + 1) * n] for r in range(self.world)], image[lo:hi].clone(),
                                         group=self.group, async_op=True)
                else:
                    wk = dist.all_gather_into_tensor(image[o:e], image[lo:hi], group=self.group, async_op=True)
                works.append(((o, e), wk) if spans else wk)
        return works

    def timing_report(self):
        """[{bucket, mib, issued_at_ms (since begin_step), wait_ms (compute stream stalled in finish)}] averaged over the logged
        steps + the exposed total; call after a device synchronize.  Clears the log."""
        log, self._tlog = self._tlog, []
        if not log:
            return None
        acc = {}
        order = []
        for t0, cur in log:
            for name, ev, e0, e1 in cur:
                a = acc.setdefault(name, [0.0, 0.0, 0])
                if name not in order:
                    order.append(name)
                a[0] += t0.elapsed_time(ev) if ev is not None else t0.elapsed_time(e0)
                a[1] += e0.elapsed_time(e1)
                a[2] += 1
        esz = 2 if self.stage is not None else 4
        size = {name: (e - o) * esz / 2.0 ** 20 for o, e, name in self.buckets}
        rows = [{"bucket": n, "mib": round(size.get(n, sum(t.numel() * 4 for t in self.extra) / 2.0 ** 20), 2),
                 "issued_at_ms": round(acc[n][0] / acc[n][2], 3), "wait_ms": round(acc[n][1] / acc[n][2], 3)} for n in order]
        return {"buckets": rows, "exposed_wait_ms": round(sum(r["wait_ms"] for r in rows), 3), "steps": len(log)}


def allreduce_param_grads(params, group=None):
    """mean of the per-rank gradients of an explicit parameter list (the autograd trainer paths: softmax / mse losses,
    seen-mask phase): one flat all-reduce.  No-op in a single process."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, group=group)
    flat /= world
    off = 0
    for p in ps:
        p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
        off += p.numel()


class SeenmaskStep(object):
    """Phase 2 (BASELINE configs[2]) as one fused step: the body of trainer_seenmask.Trainer.train_epoch
    (trainer_seenmask.py:72-102) with the optimizer wiring of train.py:164-175, without autograd:

        backbone forward (train mode: Dropout2d on, like the reference's model.train(); no activation is kept -- the
        backbone is frozen)  ->  szn_seenmask_head: learned 64x64 stride-32 deconv + crop + 2-class cross entropy
        (size_average) + channel argmax + d(coarse) + d(seenmask_upscore.weight) straight from the 1/32 map, the
        (B,2,H,W) score never exists  ->  szn_seenmask_score_wgrad (seenmask_score weight / bias)  ->
        [one 98 KB all-reduce under data parallelism]  ->  Adam on the 24,578 head parameters.

    The three trainable tensors live in one flat fp32 buffer (the module's Parameters and .grads are views of it and of
    the flat gradient); the engine's fused head image (rows E, E+1 of [CP][4096]) and bias vector are rewritten in place
    after every update, so no weight image is rebuilt between steps.  Every kernel on this path reduces in a fixed order:
    two runs give bit-identical losses and weights."""

    def __init__(self, model, n_class, unseen, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, precision=None,
                 process_group=None):
        """n_class / unseen: the binary target of a pixel is "its label is one of the n_class classes and not in `unseen`"
        (trainer_seenmask.py:53-56); n_class = 0: step() is handed {0,1} targets already (other values are ignored)"""
        if n_class > L.MAX_CLASSES:
            raise L.SznError("SeenmaskStep: at most %d classes (szn_class_set), got %d" % (L.MAX_CLASSES, n_class))
        self.model, self.eng = model, model._engine
        if precision is not None:
            model.set_precision(precision)
        self.dev = model.conv1_1.weight.device
        if self.dev.type != "cuda":
            raise L.SznError("SeenmaskStep needs the model on the GPU")
        self.n_class = int(n_class)
        self.seen = L.class_set(k for k in range(self.n_class) if k not in set(unseen))
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.nstep = 0
        m = model
        F = m.fc7.out_channels
        self.F = F
        ps = (m.seenmask_score.weight, m.seenmask_upscore.weight, m.seenmask_score.bias)
        n = sum(p.numel() for p in ps)
        self.flat_p = torch.empty(n, device=self.dev)
        self.flat_g = torch.zeros(n, device=self.dev)
        self.m1, self.m2 = torch.zeros(n, device=self.dev), torch.zeros(n, device=self.dev)
        off = 0
        self.seg = {}
        for key, p in zip(("score_w", "up_w", "score_b"), ps):
            cnt = p.numel()
            if key == "score_w":             # (2,F,1,1): OHWI order == (2, F) rows
                src = p.detach().permute(0, 2, 3, 1).reshape(-1)
                view = lambda t, o=off, c=cnt: t[o:o + c].view(2, 1, 1, F).permute(0, 3, 1, 2)
            else:
                src = p.detach().reshape(-1)
                view = lambda t, o=off, c=cnt, sh=tuple(p.shape): t[o:o + c].view(sh)
            self.flat_p[off:off + cnt].copy_(src)
            p.data = view(self.flat_p)
            p.grad = view(self.flat_g)
            self.seg[key] = (off, cnt)
            off += cnt
        self.eng.mark_dirty()
        self.loss = torch.zeros(1, device=self.dev)
        self.stats = torch.zeros(2, device=self.dev)
        self.conf = torch.zeros(4, dtype=torch.int64, device=self.dev)     # [target][prediction] counts since the last reset
        self._ws = self._ws2 = None

    def _head_images(self):
        """(rows E, E+1 of the fused head's weight image in the compute dtype, the same slice of its bias vector)"""
        m, img = self.model, self.eng._images
        E, CP, F = m.n_class, m.head_width, self.F
        return img["head.w"].view(CP, F)[E:E + 2], img["head.b"][E:E + 2]

    def step(self, x, target, dropout_masks=None):
        """x (B,3,H,W) f32 NCHW, target (B,H,W) int64 class labels (-1 = unlabelled), both on the GPU.
        -> (loss 0-dim device tensor, pred (B,H,W) int64 device tensor: 1 = "seen")"""
        m, eng = self.model, self.eng
        B, _, H, W = x.shape
        st = L.stream_ptr()
        ctx = eng.forward(x, train=m.training, masks=dropout_masks, keep=False)
        E, CP, F = m.n_class, m.head_width, self.F
        lib = L.load()
        nb = lib.szn_seenmask_head_workspace_bytes(B, ctx.h, ctx.w, H, W, CROP)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        M = B * ctx.h * ctx.w
        nb2 = lib.szn_seenmask_score_wgrad_workspace_bytes(M, F)
        if self._ws2 is None or self._ws2.numel() < nb2 + M * 8:
            self._ws2 = torch.empty(nb2 + M * 8, dtype=torch.uint8, device=self.dev)
        dsc = self._ws2[nb2:nb2 + M * 8].view(torch.float32)
        pred = torch.empty(B, H, W, dtype=torch.int64, device=self.dev)
        (ow, nw), (ou, nu), (ob, nbias) = self.seg["score_w"], self.seg["up_w"], self.seg["score_b"]
        g = self.flat_g
        L.call("szn_seenmask_head_k", B, ctx.h, ctx.w, CP, E, H, W, CROP, L.ptr(ctx.coarse), L.ptr(self.flat_p[ou:ou + nu]),
               L.ptr(target), self.n_class, self.seen, L.ptr(self.loss), L.ptr(self.stats), L.ptr(self.conf), L.ptr(pred),
               L.ptr(dsc), L.ptr(g[ou:ou + nu]), L.ptr(self._ws), st)
        feat = ctx.relu7
        L.call("szn_seenmask_score_wgrad", L.dtype_code(feat.dtype), M, F, F, L.ptr(feat), L.ptr(dsc), L.ptr(g[ow:ow + nw]),
               L.ptr(g[ob:ob + nbias]), L.ptr(self._ws2), st)
        if self.world > 1:
            dist.all_reduce(g, group=self.pg)
        self._optimizer_step()
        return self.loss.reshape(()), pred

    def _optimizer_step(self):
        self.nstep += 1
        st = L.stream_ptr()
        gs = 1.0 / self.world
        rows, bias = self._head_images()
        (ow, nw) = self.seg["score_w"]
        n = self.flat_p.numel()
        lp16 = rows.dtype != torch.float32
        for o, cnt, lp in ((ow, nw, rows if lp16 else None), (ow + nw, n - ow - nw, None)):
            L.call("szn_adam_step", cnt, L.ptr(self.flat_p[o:o + cnt]), L.ptr(self.flat_g[o:o + cnt]), L.ptr(self.m1[o:o + cnt]),
                   L.ptr(self.m2[o:o + cnt]), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                   float(self.wd), self.nstep, gs, L.ptr(lp), L.dtype_code(lp.dtype) if lp is not None else 0, st)
        if not lp16:
            rows.copy_(self.flat_p[ow:ow + nw].view(2, self.F))
        ob, nbias = self.seg["score_b"]
        bias.copy_(self.flat_p[ob:ob + nbias])
        # seenmask_upscore.weight is read by the kernels in place (fp32 parameter storage): nothing to refresh

    def metrics(self, reset=True):
        """running train metrics (trainer_seenmask.py:87: label_accuracy_score on the binary maps with n_class classes)"""
        import numpy as np
        from .utils import _hist_to_metrics
        c = self.conf.cpu().numpy().reshape(2, 2)
        if reset:
            self.conf.zero_()
        h = np.zeros((max(self.n_class, 2), max(self.n_class, 2)))
        h[:2, :2] = c
        return _hist_to_metrics(h)

    def export_optimizer_state(self, optim):
        for key, p in (("score_w", self.model.seenmask_score.weight), ("score_b", self.model.seenmask_score.bias),
                       ("up_w", self.model.seenmask_upscore.weight)):
            o, cnt = self.seg[key]
            shape = lambda t: (t[o:o + cnt].view(2, 1, 1, self.F).permute(0, 3, 1, 2) if key == "score_w" else t[o:o + cnt].view(p.shape))
            stt = optim.state[p]
            stt['step'] = torch.tensor(float(self.nstep))
            stt['exp_avg'], stt['exp_avg_sq'] = shape(self.m1), shape(self.m2)


class TrainStep(object):
    def __init__(self, model, embeddings, optimizer="adam", lr=1e-5, momentum=0.99, weight_decay=0.0005,
                 precision=torch.bfloat16, fused_head=True, loss="cos", process_group=None, bucket_mb=25,
                 train_metrics=True, betas=(0.9, 0.999), eps=1e-8, bias_lr=None, bias_weight_decay=0.0,
                 adam_weight_decay=0.0, grad_comm_dtype=None, loss_scale=None, dynamic_loss_scale=None,
                 scale_growth=2.0, scale_backoff=0.5, scale_growth_interval=2000, force_comm=None, reserved_cus=None,
                 fused_adam=None, keep_grads=True, exchange=None, direct_wire=None, sharded=None):
        """Data-parallel knobs (the reference is single-GPU): grad_comm_dtype (SZN_GRAD_COMM = fp32 | bf16) = the
        wire format of the gradient buckets; exchange=False (SZN_GRAD_COMM=off) = no exchange at all (bench.py's comm-off timing);
        direct_wire (default: on for a 16-bit wire with keep_grads=False; SZN_WIRE_DIRECT=0 turns it off) = the weight-gradient
        kernels write the 16-bit wire image themselves and the optimizer kernel reads it (no staging copies; `.grad` of the
        weights is then None); sharded (SZN_SHARDED_OPT=1) = reduce-scatter + rank-sharded optimizer + all-gather of the weight
        image instead of all-reduce + replicated optimizer."""
        if loss not in ("cos", "mse") or (fused_head and loss != "cos"):
            raise L.SznError("TrainStep: fused head supports the cosine loss; use fused_head=False for mse")
        self.model = model
        self.eng = model._engine
        # FCN8s (models.FCN8s: public skip head, not in the reference): two more Conv2d layers in the flat buffers and the head
        # chain upscore2 -> +score_pool4 -> upscore_pool4 -> +score_pool3 -> fused head over 8x8 cells, run without autograd
        self.layers = opt_layers(model)
        self.is8 = len(self.layers) > 17
        if self.is8 and not fused_head:
            raise L.SznError("TrainStep(FCN8s) runs the fused stride-8 head only; the materialised score goes through autograd")
        model.set_precision(precision)
        self.dev = model.conv1_1.weight.device
        if self.dev.type != "cuda":
            raise L.SznError("TrainStep needs the model on the GPU")
        self._ws_prep = None             # what the head of the fused head's workspace was prepared for (szn_fused_head_prepare)
        self.emb = torch.as_tensor(embeddings).to(self.dev, torch.float32).contiguous()
        self.K, self.E = self.emb.shape
        if self.E != model.n_class:
            raise L.SznError("embedding dimension %d != model n_class %d" % (self.E, model.n_class))
        self.opt, self.lr, self.momentum, self.wd = optimizer, lr, momentum, weight_decay
        # the reference wiring (train.py:126-133): biases at 2 x lr without weight decay; Adam has no weight decay at all
        self.betas, self.eps = betas, eps
        self.adam_wd = adam_weight_decay
        self.bias_lr = 2 * lr if bias_lr is None else bias_lr
        self.bias_wd = bias_weight_decay
        env_comm = os.environ.get("SZN_GRAD_COMM", "fp32")
        if grad_comm_dtype is None:
            grad_comm_dtype = torch.bfloat16 if env_comm == "bf16" else torch.float32
        self.grad_comm_dtype = grad_comm_dtype
        self.exchange = (env_comm != "off") if exchange is None else bool(exchange)
        self.keep_grads = bool(keep_grads)
        if direct_wire is None:
            direct_wire = os.environ.get("SZN_WIRE_DIRECT", "1") == "1" and not self.keep_grads
        self._want_direct = bool(direct_wire) and grad_comm_dtype != torch.float32
        self._want_sharded = (os.environ.get("SZN_SHARDED_OPT", "0") == "1") if sharded is None else bool(sharded)
        self.fused_head, self.loss_kind = fused_head, loss
        # static loss scaling for the fp16 path: gradients below 6e-8 vanish in IEEE half, so d(loss)/d(coarse) is multiplied
        # by loss_scale in fp32 before it enters the 16-bit backward pass and the optimizer kernel divides it out again
        # (grad_scale).  The .grad views then hold loss_scale x gradient.  bf16 / fp32 need none.
        self._loss_scale0 = float(loss_scale) if loss_scale is not None else (4096.0 if precision == torch.float16 else 1.0)
        # dynamic loss scaling (default for fp16): the scale, the overflow flag and the count of APPLIED optimizer steps live
        # on the device (include/szn.h, szn_grad_check_finite / szn_*_step_scaled / szn_loss_scale_update); a step whose
        # gradients contain inf / NaN leaves masters, moments and weight images untouched and halves the scale -- no host
        # synchronisation, identical decisions on every data-parallel rank (the flag is read from the all-reduced gradient)
        self.dynamic = (precision == torch.float16) if dynamic_loss_scale is None else bool(dynamic_loss_scale)
        self.scale_cfg = (float(scale_growth), float(scale_backoff), int(scale_growth_interval), 1.0, 2.0 ** 32)
        self.scale_state = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.force_comm = force_comm
        self.train_metrics = train_metrics
        self.nstep = 0
        self._flatten()
        self._buckets(bucket_mb)
        # CUs the persistent one-block-per-CU backward kernels (conv3x3_regw dgrad, conv_wgrad_taps) leave to RCCL's workgroups
        # while an exchange is in flight (default SZN_RESERVED_CUS, 0 = none; only when buckets are actually exchanged)
        if reserved_cus is None:
            reserved_cus = int(os.environ.get("SZN_RESERVED_CUS", "0"))
        self.eng.reserved_cus = int(reserved_cus) if self.buckets.active else 0
        if self.dynamic:
            self.scale_state = torch.tensor([self._loss_scale0, 0.0, 0.0, 0.0], device=self.dev)
        self.hist = torch.zeros(3, self.K, self.K, dtype=torch.int64, device=self.dev)
        self.loss = torch.zeros(1, device=self.dev)
        self.stats = None
        self._ws = None
        # Adam for fc6 / fc7 (89 % of the weights) applied in the epilogue of their weight-gradient kernels (szn_conv2d_wgrad_adam):
        # only where a gradient is final when its kernel ends -- one rank (no exchange), no dynamic loss scale -- and on the 16-bit
        # paths (the kernel with that epilogue is a 16-bit MFMA kernel).  SZN_FUSED_ADAM=0 / fused_adam=False: the separate pass.
        # keep_grads=False: the fused layers' gradients are not written to the flat gradient buffer at all (nothing reads them in
        # a training loop that calls zero_grad() next, train.py:170-175); the default keeps .grad meaningful.
        if fused_adam is None:
            fused_adam = os.environ.get("SZN_FUSED_ADAM", "1") == "1"
        self._own_stream = None          # see step(): the non-blocking stream of a small step
        self.fused_adam = bool(fused_adam and optimizer == "adam" and not self.dynamic and not self.buckets.active
                               and self.flat_w_lp is not None and os.environ.get("SZN_EARLY_ADAM", "auto") != "1")
        if self.fused_adam and not self.keep_grads:
            # the fused layers' gradients are consumed inside their weight-gradient kernel and never stored: `.grad` must not
            # look like a gradient (the reference keeps .grad valid until zero_grad(); keep_grads=True restores that)
            for n in os.environ.get("SZN_FUSED_ADAM_LAYERS", "fc6").split(","):
                if n in self.woff and n in ("fc6", "fc7"):
                    getattr(model, n).weight.grad = None
        self._pending_gather = []    # sharded optimizer: [((start, end), work)] all-gathers of the weight image nobody has waited for yet
        self.gather_wait_log = []    # (tests) (layer that asked, bucket start, bucket end) in the order the waits were issued
        self.keep_ctx = False        # tests: keep the forward state of the last step (activations stay alive one step longer)
        self.last_ctx = None

    # the class-embedding matrix: assigning a new tensor (or calling invalidate_head_prep() after writing into it behind torch's back --
    # a raw kernel does not bump _version, and a new tensor may reuse the address of the old one) makes the fused head rebuild its tables
    @property
    def emb(self):
        return self._emb

    @emb.setter
    def emb(self, t):
        self._emb = t
        self.invalidate_head_prep()

    def invalidate_head_prep(self):
        self._emb_serial = getattr(self, "_emb_serial", 0) + 1
        self._ws_prep = None

    @property
    def loss_scale(self):
        """the factor the .grad views currently carry (dynamic scaling: read back from the device -- a host sync; tests only)"""
        return float(self.scale_state[0].item()) if self.dynamic else self._loss_scale0

    def _scale(self, t):
        """t (fp32 d loss / d head input) x loss scale, as the compute dtype"""
        if self.dynamic:
            return (t.float() * self.scale_state[0]).to(self.eng.dtype)
        return (t.float() * self._loss_scale0).to(self.eng.dtype)

    @property
    def applied_steps(self):
        """optimizer steps that were actually applied (dynamic loss scaling skips overflowed ones)"""
        return int(self.scale_state[2].item()) if self.dynamic else self.nstep

    # ---- flat parameter / gradient / moment storage ----------------------------------------------------
    def _flatten(self):
        m = self.model
        ws = [getattr(m, n).weight for n in self.layers]
        bs = [getattr(m, n).bias for n in self.layers]
        nw, nb = sum(p.numel() for p in ws), sum(p.numel() for p in bs)
        CP, E, F = m.head_width, m.n_class, m.fc7.out_channels
        self.flat_w = torch.empty(nw, device=self.dev)
        # score_fr is the last layer of the flat layout: the (CP - E) rows behind its bias slot complete the fused head's
        # bias vector (seenmask_score + zero padding), so the engine can use one contiguous view of it
        self._flat_b_store = torch.zeros(nb + CP - E, device=self.dev)
        self.flat_b = self._flat_b_store[:nb]
        # gradients: the fused head's wgrad (CP rows: score_fr | seenmask_score | padding) writes straight into score_fr's
        # slot at the END of the flat buffers; the (CP - E) rows behind it land in a tail that nothing reads
        self._flat_gw_store = torch.zeros(nw + (CP - E) * F, device=self.dev)
        self._flat_gb_store = torch.zeros(nb + CP - E, device=self.dev)
        self.flat_gw = self._flat_gw_store[:nw]
        self.flat_gb = self._flat_gb_store[:nb]
        self.woff, self.boff = {}, {}
        off = 0
        for n, p in zip(self.layers, ws):
            co, ci, kh, kw = p.shape
            seg = self.flat_w[off:off + p.numel()].view(co, kh, kw, ci)
            seg.copy_(p.detach().permute(0, 2, 3, 1))
            p.data = seg.permute(0, 3, 1, 2)                       # channels_last view of the flat master
            p.grad = self.flat_gw[off:off + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            self.woff[n] = (off, p.numel())
            off += p.numel()
        off = 0
        for n, p in zip(self.layers, bs):
            seg = self.flat_b[off:off + p.numel()]
            seg.copy_(p.detach())
            p.data = seg
            p.grad = self.flat_gb[off:off + p.numel()]
            self.boff[n] = (off, p.numel())
            off += p.numel()
        self.state = {}
        for key, flat in (("w", self.flat_w), ("b", self.flat_b)):
            if self.opt == "adam":
                self.state[key] = (torch.zeros_like(flat), torch.zeros_like(flat))
            else:
                self.state[key] = (torch.zeros_like(flat),)
        # low-precision weight image written by the optimizer kernel itself (no separate cast pass per step)
        self.flat_w_lp = None
        self.eng.lp_views = {}
        if self.eng.dtype in (torch.bfloat16, torch.float16):
            # (CP - E) x F extra elements behind score_fr's slot: the fused head's weight image [CP][F] is then a view
            store = torch.zeros(nw + (CP - E) * F, device=self.dev, dtype=self.eng.dtype)
            store[:nw].copy_(self.flat_w)
            self._flat_w_lp_store = store
            self.flat_w_lp = store[:nw]
            for n in self.layers[:-1]:
                co, ci, kh, kw = getattr(m, n).weight.shape
                o, cnt = self.woff[n]
                if n.startswith("score_pool"):
                    continue                                  # padded images are assembled per step (_skip_images)
                self.eng.lp_views[n] = self.flat_w_lp[o:o + cnt].view(co, kh, kw, ci)
            assert self.layers[-1] == "score_fr"
            o, cnt = self.woff["score_fr"]
            bo, bc = self.boff["score_fr"]
            assert o + cnt == nw and bo + bc == nb
            self.eng.lp_views["head"] = (store[o:o + CP * F].view(CP, F), self._flat_b_store[bo:bo + CP])
            self.eng._seen_versions = None
        self.eng.mark_dirty()
        # per-layer gradient targets handed to the engine (OHWI views of the flat gradient)
        self.grads = {}
        self.skip = {}               # FCN8s: per skip layer (padded weight image, dgrad image, bias image, gradient scratch)
        for n in self.layers[:-1]:
            if n.startswith("score_pool"):
                ci = getattr(m, n).in_channels
                dt = self.eng.dtype
                self.skip[n] = dict(w=torch.zeros(CP, 1, 1, ci, device=self.dev, dtype=dt),
                                    wT=torch.zeros(ci, 1, 1, CP, device=self.dev, dtype=dt),
                                    b=torch.zeros(CP, device=self.dev), gw=torch.empty(CP, 1, 1, ci, device=self.dev),
                                    gb=torch.empty(CP, device=self.dev))
                continue
            p = getattr(m, n).weight
            co, ci, kh, kw = p.shape
            o, cnt = self.woff[n]
            bo, bc = self.boff[n]
            self.grads[n] = (self.flat_gw[o:o + cnt].view(co, kh, kw, ci), self.flat_gb[bo:bo + bc])
        o, cnt = self.woff["score_fr"]
        bo, bc = self.boff["score_fr"]
        assert o + cnt == nw and bo + bc == nb       # score_fr is the last layer of both flat layouts
        self.head_gw = self._flat_gw_store[o:o + CP * F].view(CP, 1, 1, F)
        self.head_gb = self._flat_gb_store[bo:bo + CP]
        self.grads["head"] = (self.head_gw, self.head_gb)
        self.grads["_flat_bias"] = self._flat_gb_store      # lets the engine zero all bias gradients with one fill

    def _buckets(self, bucket_mb):
        layers = [(n,) + self.woff[n] for n in self.layers]
        m = self.model
        CP, E, F = m.head_width, m.n_class, m.fc7.out_channels
        # direct 16-bit wire: bf16 compute path only (the kernels that write the image are the 16-bit weight-gradient kernels; the
        # dynamic loss scale's overflow check reads fp32 gradients; the FCN8s skip layers copy theirs in from scratch buffers)
        direct = self._want_direct and not self.dynamic and not self.is8 and self.eng.dtype != torch.float32
        # sharded optimizer: not with the dynamic loss scale either -- after a reduce-scatter a rank holds summed gradients for its own slices
        # only, so the overflow check (szn_grad_check_finite over the whole buffer) would see different data on different ranks and the ranks
        # would disagree on skipping the step / backing the scale off
        self.buckets = GradBuckets(self.flat_gw, layers, bucket_mb * (1 << 20) // 4, extra=[self.flat_gb], group=self.pg,
                                   comm_dtype=self.grad_comm_dtype, force=self.force_comm, enabled=self.exchange, direct=direct,
                                   sharded=self._want_sharded and not self.dynamic, tail=(CP - E) * F)
        self._gather = []
        if self.buckets.direct:
            stage = self.buckets.stage
            lp = {}
            for n in self.layers[:-1]:
                o, cnt = self.woff[n]
                lp[n] = stage[o:o + cnt]
                getattr(m, n).weight.grad = None         # never written on this path (the wire image is the gradient)
            o, cnt = self.woff["score_fr"]
            lp["head"] = stage[o:o + CP * F]
            m.score_fr.weight.grad = None
            self.grads["_lp"] = lp

    # ---- one training step -----------------------------------------------------------------------------
    def step(self, x, target):
        """x (B,3,H,W) f32 NCHW, target (B,H,W) int64 (-1 ignore), both on the GPU.
        Returns (loss 0-dim device tensor, pred (B,H,W) int64 device tensor)."""
        work = self._small_step_stream(x)
        if work is None:
            return self._step(x, target)
        # A small step (the reference's one image per step, train.py:82-84) whose fc6 update rides in its weight-gradient kernel: that
        # kernel goes to a stream confined to part of the CUs (models._Engine._masked_stream).  Such a stream is a BLOCKING one -- it
        # synchronises with the null stream launch by launch (measured: 2.57 -> 3.5 ms) -- so when the caller is on the null stream the
        # whole step moves to a non-blocking stream of its own, forked from and joined to the caller's.
        cur = torch.cuda.current_stream(self.dev)
        work.wait_stream(cur)
        with torch.cuda.stream(work):
            out = self._step(x, target)
        cur.wait_stream(work)
        for t in out:
            t.record_stream(cur)
        return out

    def _small_step_stream(self, x):
        if not self.fused_adam or self.eng.dtype == torch.float32 or torch.cuda.is_current_stream_capturing():
            return None
        if not _models.masked_stream_wanted(x.shape[0] * x.shape[2] * x.shape[3]):
            return None
        if torch.cuda.current_stream(self.dev) != torch.cuda.default_stream(self.dev):
            return None                  # already on a non-blocking stream of the caller's
        if self._own_stream is None:
            self._own_stream = torch.cuda.Stream(device=self.dev)
        return self._own_stream

    def _step(self, x, target):
        m, eng = self.model, self.eng
        B, _, H, W = x.shape
        st = L.stream_ptr()
        eng.keep_prepool = bool(self.keep_ctx)          # tests that read the forward state need the un-pooled tensors too
        ctx = eng.forward(x, train=m.training)
        self.last_ctx = ctx if self.keep_ctx else None
        if self.is8:
            return self._step8(ctx, target, B, H, W)
        CP, E, K = m.head_width, self.E, self.K
        code = L.dtype_code(eng.dtype)
        pred = torch.empty(B, H, W, dtype=torch.int64, device=self.dev)
        stats = torch.empty(B, 2, device=self.dev)
        scaled = self.dynamic or self._loss_scale0 != 1.0
        # d(loss)/d(coarse): the fused head writes channels [0, E) and the padding channels stay zero, so the buffer of the previous
        # step is reused without a fill (everything that read it was queued on this stream -- or joined -- before this step)
        key = (B, ctx.h, ctx.w, CP, torch.float32 if scaled else eng.dtype)
        if self.fused_head and getattr(self, "_dcoarse_key", None) == key:
            dcoarse = self._dcoarse
        else:
            dcoarse = torch.zeros(B, ctx.h, ctx.w, CP, device=self.dev, dtype=key[4])
            self._dcoarse, self._dcoarse_key = (dcoarse, key) if self.fused_head else (None, None)
        code = L.dtype_code(dcoarse.dtype)
        if self.fused_head:
            nbytes = L.load().szn_fused_head_workspace_bytes(B, ctx.h, ctx.w, E, K)
            if self._ws is None or self._ws.numel() < nbytes:
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            # the embeddings are constants of a run: their transpose + norms (fh_prep_kernel: 23 us of dependent loads) are written to the head of
            # the workspace once -- again whenever the workspace or the embedding tensor (an in-place edit bumps _version) changes
            prep = (self._ws.data_ptr(), self.emb.data_ptr(), self.emb._version, self._emb_serial, E, K)
            if self._ws_prep != prep:
                L.call("szn_fused_head_prepare", E, K, L.ptr(self.emb), L.ptr(self._ws), st)
                self._ws_prep = prep
            L.call("szn_fused_head_prepared", 32, B, ctx.h, ctx.w, E, CP, 0, H, W, CROP, K, L.ptr(ctx.coarse), L.ptr(self.emb),
                   L.ptr(target), L.ptr(self.loss), L.ptr(stats), L.ptr(pred), code, L.ptr(dcoarse), L.ptr(self._ws), st)
        else:
            f = eng.upscore(ctx)
            ws = torch.empty(L.load().szn_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=self.dev)
            fwd = "szn_cosine_loss_fwd" if self.loss_kind == "cos" else "szn_mse_loss_fwd"
            bwd = "szn_cosine_loss_bwd" if self.loss_kind == "cos" else "szn_mse_loss_bwd"
            L.call(fwd, B, E, H, W, K, L.ptr(f), L.ptr(target), L.ptr(self.emb), None, L.ptr(self.loss), L.ptr(stats),
                   L.ptr(ws), st)
            L.call("szn_embed_argmax_k", B, E, H, W, K, L.ptr(f), L.ptr(self.emb), 0, None, None, None, L.ptr(pred), st)
            df = torch.empty_like(f)
            L.call(bwd, B, E, H, W, K, L.ptr(f), L.ptr(target), L.ptr(self.emb), None, L.ptr(stats), None, L.ptr(df), st)
            dc32, _ = eng.head_backward(ctx, df=df)
            dcoarse = dc32
        if scaled:
            dcoarse = self._scale(dcoarse)
        self.stats = stats
        self._fused_begin()
        try:
            self.buckets.begin_step()
            self._backward(ctx, dcoarse, self._layer_done_hook(B * H * W))
            self.buckets.finish()
            self._optimizer_step()
        finally:                         # an error in between must not leave the engine armed with this step's Adam arguments
            eng.fused_opt, eng.fused_done = None, set()
        if self.train_metrics:
            L.call("szn_confusion_hist_k", target.numel(), K, L.ptr(target), L.ptr(pred), None, L.ptr(self.hist), st)
        return self.loss.reshape(()), pred

    # ---- FCN8s head chain (forward and backward by hand: no autograd objects on the step path) ----------------------------
    def _skip_images(self):
        """padded [CP][Ci] images of score_pool3 / score_pool4 (rows >= E stay zero) + their dgrad transposes, from the flat
        masters the optimizer kernel just wrote"""
        m, E = self.model, self.E
        code = L.dtype_code(self.eng.dtype)
        for n, s in self.skip.items():
            o, cnt = self.woff[n]
            ci = s["w"].shape[3]
            src = self.flat_w_lp if (self.flat_w_lp is not None) else self.flat_w
            s["w"][:E].copy_(src[o:o + cnt].view(E, 1, 1, ci))
            bo, bc = self.boff[n]
            s["b"][:E].copy_(self.flat_b[bo:bo + bc])
            L.call("szn_pack_weight_dgrad", code, s["w"].shape[0], 1, 1, ci, L.ptr(s["w"]), L.ptr(s["wT"]), L.stream_ptr())

    @staticmethod
    def _up2(x, fwd=True, shape=None):
        x = x.contiguous()
        if fwd:
            B, h, w, ld = x.shape
            out = torch.empty(B, 2 * h + 2, 2 * w + 2, ld, device=x.device, dtype=torch.float32)
            L.call("szn_bilinear_up2_nhwc_fwd", B, h, w, ld, ld, L.ptr(x), L.ptr(out), L.stream_ptr())
            return out
        B, h, w, ld = shape
        out = torch.empty(B, h, w, ld, device=x.device, dtype=torch.float32)
        L.call("szn_bilinear_up2_nhwc_bwd", B, h, w, ld, ld, L.ptr(x), L.ptr(out), L.stream_ptr())
        return out

    def _step8(self, ctx, target, B, H, W):
        m, eng = self.model, self.eng
        st = L.stream_ptr()
        CP, E, K = m.head_width, self.E, self.K
        dt = eng.dtype
        self._skip_images()
        pool3, pool4 = ctx.pools[2][1], ctx.pools[3][1]
        s3, s4 = self.skip["score_pool3"], self.skip["score_pool4"]
        # forward: upscore2(score_fr) + score_pool4c -> upscore_pool4 -> + score_pool3c
        up2 = self._up2(ctx.coarse)
        sp4 = eng._conv(pool4, None, 0, relu=False, out_f32=True, w=s4["w"], b=s4["b"])
        n4, m4 = up2.shape[1:3]
        fuse4 = up2 + sp4[:, CROP_POOL4:CROP_POOL4 + n4, CROP_POOL4:CROP_POOL4 + m4]
        up4 = self._up2(fuse4)
        sp3 = eng._conv(pool3, None, 0, relu=False, out_f32=True, w=s3["w"], b=s3["b"])
        n3, m3 = up4.shape[1:3]
        fuse3 = (up4 + sp3[:, CROP_POOL3:CROP_POOL3 + n3, CROP_POOL3:CROP_POOL3 + m3]).contiguous()
        # fused head over 8x8 cells: loss, prediction, d(fuse3)
        pred = torch.empty(B, H, W, dtype=torch.int64, device=self.dev)
        stats = torch.empty(B, 2, device=self.dev)
        dfuse3 = torch.zeros(B, n3, m3, CP, device=self.dev, dtype=torch.float32)
        nbytes = L.load().szn_fused_head_workspace_bytes(B, n3, m3, E, K)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        # (the embedding tables at the head of the workspace are prepared once per workspace / embedding tensor, as in _step)
        prep = (self._ws.data_ptr(), self.emb.data_ptr(), self.emb._version, self._emb_serial, E, K)
        if self._ws_prep != prep:
            L.call("szn_fused_head_prepare", E, K, L.ptr(self.emb), L.ptr(self._ws), st)
            self._ws_prep = prep
        L.call("szn_fused_head_prepared", 8, B, n3, m3, E, CP, 0, H, W, CROP_UP8, K, L.ptr(fuse3), L.ptr(self.emb), L.ptr(target),
               L.ptr(self.loss), L.ptr(stats), L.ptr(pred), L.SZN_F32, L.ptr(dfuse3), L.ptr(self._ws), st)
        self.stats = stats
        if self.dynamic:
            dfuse3 = dfuse3 * self.scale_state[0]
        elif self._loss_scale0 != 1.0:
            dfuse3 = dfuse3 * self._loss_scale0
        # backward of the head chain; the skip gradients join the backbone chain at the pool3 / pool4 outputs
        skips = {}
        dmap = dfuse3
        for (name, s, pool, crop, pi, up_shape) in (("score_pool3", s3, pool3, CROP_POOL3, 2, fuse4.shape),
                                                    ("score_pool4", s4, pool4, CROP_POOL4, 3, ctx.coarse.shape)):
            Bp, hp, wp, ci = pool.shape
            nn, mm = dmap.shape[1:3]
            dsp = torch.zeros(Bp, hp, wp, CP, device=self.dev, dtype=dt)
            dsp[:, crop:crop + nn, crop:crop + mm] = dmap
            o, cnt = self.woff[name]
            # the copy into the flat gradient runs as the wgrad's `after` hook: on the weight-gradient stream when
            # SZN_WGRAD_STREAM=1 moves these launches off the main stream (a copy queued on the main stream would race them)
            eng._wgrad(pool, dsp, s["gw"], s["gb"], ci, CP, 1, 0,
                       after=lambda o=o, cnt=cnt, s=s, ci=ci: self.flat_gw[o:o + cnt].view(E, 1, 1, ci).copy_(s["gw"][:E]))
            skips[pi] = eng._dgrad(dsp, None, pool.shape, 0, wT=s["wT"])
            dmap = self._up2(dmap, fwd=False, shape=tuple(up_shape))
        eng._join_wgrad()
        dcoarse = dmap if dmap.dtype == dt else dmap.to(dt)
        done = self.buckets.layer_done

        def head_first():                 # flat order is (..., score_pool3, score_pool4, score_fr): report them last-to-first
            done("score_fr"); done("score_pool4"); done("score_pool3")
        # the bias gradients of the skip layers live in the flat bias buffer the engine zeroes first: re-apply them after
        saved = [(self.boff[n], self.skip[n]["gb"]) for n in ("score_pool3", "score_pool4")]
        self._fused_begin()
        try:
            self.buckets.begin_step()
            eng.backward(ctx, dcoarse, self.grads, backbone=True, layer_done=done, head_first=head_first, skips=skips)
            for (bo, bc), gb in saved:
                self.flat_gb[bo:bo + bc].copy_(gb[:E])
            self.buckets.finish()
            self._optimizer_step()
        finally:
            eng.fused_opt, eng.fused_done = None, set()
        if self.train_metrics:
            L.call("szn_confusion_hist_k", target.numel(), K, L.ptr(target), L.ptr(pred), None, L.ptr(self.hist), st)
        return self.loss.reshape(()), pred

    def _backward(self, ctx, dcoarse, layer_done):
        eng, m = self.eng, self.model
        eng.backward(ctx, dcoarse, self.grads, backbone=True, layer_done=layer_done, head_first=self._head_copy(layer_done))

    def _head_copy(self, layer_done):
        # rows [0,E) of the fused head gradient ARE score_fr's slot of the flat gradient (seenmask_score is frozen in
        # phase 1): nothing to copy, the first bucket can go as soon as the head wgrad has been queued
        return lambda: layer_done("score_fr")

    # ---- optimizer --------------------------------------------------------------------------------------
    def _layer_done_hook(self, pixels):
        """layer_done callback of one backward pass.  Small steps (the reference's batch size of one image: train.py:82-84) leave most
        of the chip idle during conv5_3 .. conv1_1's backward kernels (32 .. 124 tiles on 256 CUs) while the HBM-bound optimizer pass
        over fc6 / fc7 / score_fr -- 90 % of the parameters, final as soon as fc6's dgrad has been queued -- waits for the end of the
        step: on one rank without dynamic loss scaling that slice is updated on a second stream from layer_done("conv5_3") on (fc6's
        dgrad, the last reader of fc6's weight image, is queued before it).  Same kernels, same values.  At B = 8 the MFMA-bound backward
        kernels slow down by what the overlapped pass takes (profiles/r03_ablations.txt section 17), and at B = 1 the 16-bit step does not gain
        either (3.09 -> 3.12 ms: the pass takes CUs from the few-tile kernels it was meant to hide behind), but the fp32 step does (15.41
        -> 14.90 ms, profiles/r04_ablations.txt section 7): SZN_EARLY_ADAM = 0 / 1 forces it off / on, default = fp32 steps of at most
        2 x 512 x 512 pixels."""
        self._early = False
        base = self.buckets.layer_done
        mode = os.environ.get("SZN_EARLY_ADAM", "auto")
        on = mode == "1" or (mode == "auto" and pixels <= 2 * 512 * 512 and self.eng.dtype == torch.float32)
        if not on or self.buckets.active or self.dynamic or self.is8 or "fc6" not in self.woff:
            return base

        def done(name):
            base(name)
            if name == "conv5_3":
                if getattr(self, "_side", None) is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                cur = torch.cuda.current_stream()
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._opt_launch("w", self.woff["fc6"][0], self.flat_w.numel(), self.nstep + 1)
                self._early = True
        return done

    def _fused_begin(self):
        """hand the engine the Adam arguments of the layers whose update rides in their weight-gradient kernel (this step = nstep + 1)"""
        eng = self.eng
        eng.fused_done = set()
        eng.fused_opt = None
        if not self.fused_adam:
            return
        m1, m2 = self.state["w"]
        gs = 1.0 / (self.world * self._loss_scale0)
        out = {}
        # fc6 only by default: its 1,568 tiles run 6 rounds, so update traffic and K loops of different CUs overlap (852 us against
        # 447 + 523 separately at B = 8); fc7's 256 tiles are one round -- K loop, then everybody's update: 155 us against 70 + 75
        for n in os.environ.get("SZN_FUSED_ADAM_LAYERS", "fc6").split(","):
            if n not in self.woff or n not in ("fc6", "fc7"):
                continue
            o, cnt = self.woff[n]
            a = L.AdamArgs()
            a.param, a.exp_avg, a.exp_avg_sq = self.flat_w[o:o + cnt].data_ptr(), m1[o:o + cnt].data_ptr(), m2[o:o + cnt].data_ptr()
            a.w_lp, a.w_lp_dtype = self.flat_w_lp[o:o + cnt].data_ptr(), L.dtype_code(self.flat_w_lp.dtype)
            a.lr, a.beta1, a.beta2, a.eps = float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps)
            a.weight_decay, a.step, a.grad_scale = float(self.adam_wd), self.nstep + 1, gs
            out[n] = (a, self.keep_grads)
        eng.fused_opt = out

    def _opt_launch(self, key, lo, hi, nstep):
        """one optimizer launch over elements [lo, hi) of the flat weight ("w") or bias ("b") buffers, on the current stream"""
        if hi <= lo:
            return
        st = L.stream_ptr()
        dyn = self.scale_state
        gs = 1.0 / self.world if self.dynamic else 1.0 / (self.world * self._loss_scale0)
        flat, grad = (self.flat_w, self.flat_gw) if key == "w" else (self.flat_b, self.flat_gb)
        g16 = key == "w" and self.buckets.direct      # the summed 16-bit wire image IS the gradient (szn_*_step_g16)
        if g16:
            grad = self.buckets.stage
        lr, wd = (self.lr, self.wd) if key == "w" else (self.bias_lr, self.bias_wd)
        lp_on = key == "w" and self.flat_w_lp is not None
        lp_code = L.dtype_code(self.flat_w_lp.dtype) if self.flat_w_lp is not None else 0
        lp = L.ptr(self.flat_w_lp[lo:hi]) if lp_on else None
        n = hi - lo
        if self.opt == "adam":           # train.py:130-133 (Adam has no weight decay in the reference wiring)
            m1, m2 = self.state[key]
            awd = float(self.bias_wd if key == "b" else self.adam_wd)
            if g16:
                L.call("szn_adam_step_g16", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.dtype_code(grad.dtype), L.ptr(m1[lo:hi]),
                       L.ptr(m2[lo:hi]), float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), awd, nstep, gs, lp,
                       lp_code, st)
            elif self.dynamic:
                L.call("szn_adam_step_scaled", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.ptr(m1[lo:hi]), L.ptr(m2[lo:hi]), float(lr),
                       float(self.betas[0]), float(self.betas[1]), float(self.eps), awd, L.ptr(dyn), gs, lp, lp_code, st)
            else:
                L.call("szn_adam_step", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.ptr(m1[lo:hi]), L.ptr(m2[lo:hi]), float(lr),
                       float(self.betas[0]), float(self.betas[1]), float(self.eps), awd, nstep, gs, lp, lp_code, st)
        else:                            # train.py:126-129
            (buf,) = self.state[key]
            if g16:
                L.call("szn_sgd_momentum_step_g16", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.dtype_code(grad.dtype), L.ptr(buf[lo:hi]),
                       float(lr), float(self.momentum), float(wd), int(nstep == 1), gs, lp, lp_code, st)
            elif self.dynamic:
                L.call("szn_sgd_momentum_step_scaled", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.ptr(buf[lo:hi]), float(lr),
                       float(self.momentum), float(wd), L.ptr(dyn), gs, lp, lp_code, st)
            else:
                L.call("szn_sgd_momentum_step", n, L.ptr(flat[lo:hi]), L.ptr(grad[lo:hi]), L.ptr(buf[lo:hi]), float(lr),
                       float(self.momentum), float(wd), int(nstep == 1), gs, lp, lp_code, st)

    def _optimizer_step(self):
        self.nstep += 1
        st = L.stream_ptr()
        dyn = self.scale_state
        if self.dynamic:                 # raise the overflow flag from the (already all-reduced) gradients
            L.call("szn_grad_check_finite", self.flat_gw.numel(), L.ptr(self.flat_gw), L.ptr(dyn), st)
            L.call("szn_grad_check_finite", self.flat_gb.numel(), L.ptr(self.flat_gb), L.ptr(dyn), st)
        whi = self.flat_w.numel()
        if getattr(self, "_early", False):           # fc6 .. score_fr were updated under the backward pass (_layer_done_hook)
            torch.cuda.current_stream().wait_stream(self._side)
            whi = self.woff["fc6"][0]
            self._early = False
        if self.buckets.sharded:
            # reduce-scattered buckets: this rank holds the sum of one slice per bucket and updates only that (1/world of the pass);
            # the updated slices of the weight image are all-gathered (16-bit on the 16-bit paths; the fp32 masters of the other
            # ranks' slices go stale here -- gather_masters() before anything reads them: checkpoints)
            for o, e, _ in self.buckets.buckets:
                lo, hi = self.buckets.shard(o, e)
                self._opt_launch("w", lo, hi, self.nstep)
            image = self.flat_w_lp if self.flat_w_lp is not None else self.flat_w
            works = self.buckets.gather_weights(image, spans=True)
            if self.flat_w_lp is not None:       # conv1_1's kernel reads the fp32 master (3 input channels: no 16-bit image): its bucket's too
                works += self.buckets.gather_weights(self.flat_w, first_only=True, spans=True)
            # Nobody waits here: the all-gathers (forward order, conv1_1's 2 MiB bucket first, fc6's 196 MB fourth) run on RCCL's stream
            # under the tail of this step and the head of the next forward pass, whose layers wait for THEIR bucket only
            # (_Engine.weight_gate -> wait_weights(layer)): conv1_1 .. conv5_3 of the next step run while fc6's image is still on the links
            self._pending_gather = works
            self.eng.weight_gate = self.wait_weights if works else None       # (one rank owns every slice: nothing to wait for)
            # masters (16-bit paths) and moments (every path) of the other ranks' slices are stale from here on
            self._masters_stale = self.buckets.world > 1
        else:
            # the weight ranges not already updated inside their weight-gradient kernels (_fused_begin)
            lo = 0
            for n in sorted(self.eng.fused_done, key=lambda n: self.woff[n][0]):
                o, cnt = self.woff[n]
                if o < whi:
                    self._opt_launch("w", lo, min(o, whi), self.nstep)
                    lo = o + cnt
            self._opt_launch("w", lo, whi, self.nstep)
        self.eng.fused_opt = None
        self.eng.fused_done = set()
        self._opt_launch("b", 0, self.flat_b.numel(), self.nstep)
        if self.dynamic:
            g, b, iv, lo, hi = self.scale_cfg
            L.call("szn_loss_scale_update", L.ptr(dyn), g, b, iv, lo, hi, st)
        self.eng.mark_dirty()

    def wait_weights(self, layer=None):
        """sharded optimizer: make the current stream wait for the pending all-gathers of the weight image -- of the bucket that holds
        `layer` (and of every bucket in front of it: forward order), or of all of them (None).  Called by the engine in front of every
        layer's first use of its weights in the next forward pass, and by everything else that reads parameters (gather_masters)."""
        pend = getattr(self, "_pending_gather", None)
        if not pend:
            return
        if layer is None:
            upto = self.flat_w.numel()
        else:
            o, cnt = self.woff["score_fr" if layer == "head" else layer]
            upto = o + cnt
        keep = []
        for (o, e), wk in pend:
            if o < upto:
                wk.wait()                # (the compute stream waits, not the host)
                self.gather_wait_log.append((layer, o, e))
            else:
                keep.append(((o, e), wk))
        self._pending_gather = keep
        if not keep:
            self.eng.weight_gate = None

    def gather_masters(self):
        """sharded optimizer: bring the optimizer moments -- and on a 16-bit path the fp32 masters (the fp32 path all-gathers them as the
        weight image in every step) -- of the other ranks' slices up to date on this rank (all-gather over the bucket slices).  Call before
        reading model parameters / optimizer state: checkpoints, export."""
        self.wait_weights()
        if not getattr(self, "_masters_stale", False):
            return
        ts = ([self.flat_w] if self.flat_w_lp is not None else []) + list(self.state["w"])
        for t in ts:
            for wk in self.buckets.gather_weights(t):
                wk.wait()
        self._masters_stale = False

    # ---- checkpoint compatibility (reference dict keys: trainer_fcn.py:281-288) ------------------------------
    def _param_slots(self):
        m = self.model
        for n in self.layers:
            p = getattr(m, n).weight
            co, ci, kh, kw = p.shape
            o, cnt = self.woff[n]
            yield p, "w", lambda t, o=o, cnt=cnt, co=co, ci=ci, kh=kh, kw=kw: t[o:o + cnt].view(co, kh, kw, ci).permute(0, 3, 1, 2)
            b = getattr(m, n).bias
            bo, bc = self.boff[n]
            yield b, "b", lambda t, bo=bo, bc=bc: t[bo:bo + bc]

    def export_optimizer_state(self, optim):
        """expose the flat moments as per-parameter state of a torch-style optimizer object (views, no copies), so that
        `optim.state_dict()` written into a checkpoint has the layout torch.optim.Adam / SGD would have produced"""
        self.gather_masters()
        for p, key, view in self._param_slots():
            st = optim.state[p]
            if self.opt == "adam":
                st['step'] = torch.tensor(float(self.applied_steps))
                st['exp_avg'] = view(self.state[key][0])
                st['exp_avg_sq'] = view(self.state[key][1])
            else:
                st['momentum_buffer'] = view(self.state[key][0]) if self.applied_steps > 0 else None

    def import_optimizer_state(self, optim):
        """inverse of export_optimizer_state (resume: train.py:135-136 loads `optim_state_dict` into the optimizer)"""
        steps = []
        for p, key, view in self._param_slots():
            st = optim.state.get(p, {})
            if self.opt == "adam" and 'exp_avg' in st:
                view(self.state[key][0]).copy_(st['exp_avg'])
                view(self.state[key][1]).copy_(st['exp_avg_sq'])
                steps.append(int(st.get('step', 0)))
            elif self.opt != "adam" and st.get('momentum_buffer') is not None:
                view(self.state[key][0]).copy_(st['momentum_buffer'])
                steps.append(1)
        if steps:
            self.nstep = max(steps)
            if self.dynamic:
                self.scale_state[2] = float(self.nstep)

    def metrics(self, reset=True):
        """running train metrics from the device histogram (trainer_fcn.py:164)"""
        from .utils import _hist_to_metrics
        h = self.hist[0].cpu().numpy()
        if reset:
            self.hist.zero_()
        return _hist_to_metrics(h)
