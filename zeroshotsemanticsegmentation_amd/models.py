"""models.py -- MI355X-native FCN32s with the reference's module surface.

Mirrors /root/reference/models.py: `get_upsampling_weight` (:11-24), `FCN32s` (:27-193) with the same layer
attribute names / parameter shapes / state_dict keys, `forward(x, mode)` (:114-160) and
`copy_params_from_vgg16` (:162-193), `VGG16` (:195-203).  Underneath, every operation is a hand-written
HIP kernel reached through the C-ABI of include/szn.h (no torch.nn.functional compute, no CPU fallback):

  * activations live in HBM as NHWC in the compute dtype (fp32 parity path or bf16 throughput path),
  * conv weights are kept as fp32 masters in OHWI memory order (torch channels_last of the (O,I,KH,KW)
    parameter), with a compute-dtype image and a flipped/transposed image for dgrad refreshed whenever a
    parameter changes,
  * score_fr and seenmask_score run as ONE projection GEMM with N = n_class + 2 (padded to 64).

The layer objects subclass torch.nn.Conv2d / ConvTranspose2d / ReLU / MaxPool2d / Dropout2d purely as typed
parameter containers so that `train.get_parameters` (reference train.py:302-331) classifies them the same way.
"""
import atexit
import contextlib
import ctypes as C
import math
import os
import os.path as osp

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import synth

CROP = 19            # models.py:147
PAD1 = 100           # models.py:43


def get_upsampling_weight(in_channels, out_channels, kernel_size):
    """2-D bilinear kernel on the channel diagonal, float64 -> float32 (reference models.py:11-24)."""
    return torch.from_numpy(synth.bilinear_weight(in_channels, out_channels, kernel_size))


def _round_up(a, b):
    return (a + b - 1) // b * b


# (name, pad) in forward order; "P" = MaxPool2d(2,2,ceil)   (reference models.py:116-137)
_BACKBONE = [("conv1_1", PAD1), ("conv1_2", 1), "P", ("conv2_1", 1), ("conv2_2", 1), "P",
             ("conv3_1", 1), ("conv3_2", 1), ("conv3_3", 1), "P", ("conv4_1", 1), ("conv4_2", 1), ("conv4_3", 1), "P",
             ("conv5_1", 1), ("conv5_2", 1), ("conv5_3", 1), "P"]
_TRUNK = [n for n, _, _, _ in synth.CONV_LAYERS]          # conv1_1 .. fc7
_CONST_BORDER = os.environ.get("SZN_CONST_BORDER", "1") != "0"  # 0: no constant-border hint to the 710^2 / 355^2 forward convs
_DGRAD_SPLIT = True         # few-tile dgrads may split their K range (the library decides; its split-K epilogue delivers the column sums)
_WGRAD_CB_FUSED = True      # the pools' backward pass sums dn over the tiles the producer's weight gradient skips (False: that call sums them itself)
_DGRAD_BORDER = os.environ.get("SZN_DGRAD_BORDER", "1") != "0"        # 0: conv1_2's dgrad runs the tiles nobody reads (for their column sums)
_FC6_NATIVE = True          # fc6's dgrad GEMM on the forward weight image (False: on the packed transpose, rounds 1-2)
_OPT_LAYERS = _TRUNK + ["score_fr"]                        # the layers train.get_parameters yields (train.py:302-331)
_OPT_LAYERS8 = _TRUNK + ["score_pool3", "score_pool4", "score_fr"]       # ... for FCN8s (score_fr stays last: engine.TrainStep)


def opt_layers(model):
    """names of the Conv2d layers the phase-1 optimizer updates, in the order of TrainStep's flat buffers"""
    return _OPT_LAYERS8 if hasattr(model, "score_pool3") else _OPT_LAYERS


# ---- constant-border bookkeeping (szn_conv_desc_t.cb_on) -------------------------------------------------------------------------
# conv1_1 pads by 100 (models.py:43): outside the rows / columns the image can reach, its output is relu(bias) everywhere, and the
# layers behind it keep one value per channel there until their own zero padding is felt.  A region is, per axis, (r0, r1) = the
# interval the image can influence and (c0, c1) = the interval the zero padding of the layers so far cannot influence.
def _cb_conv1_1(n_in, pad):
    n_out = n_in + 2 * pad - 2
    return (max(pad - 2, 0), min(pad + n_in, n_out), 0, n_out)


def _cb_conv3x3(reg, n):
    r0, r1, c0, c1 = reg
    return (max(r0 - 1, 0), min(r1 + 1, n), c0 + 1, c1 - 1)


def _cb_pool(reg, n):
    r0, r1, c0, c1 = reg
    npool = (n + 1) // 2
    return (r0 // 2, min((r1 + 1) // 2, npool), (c0 + 1) // 2, npool if c1 >= n else c1 // 2)


# ---- the constant band inside the conv3 block (round 5) -------------------------------------------------------------------------------
# At 1/4 resolution (conv3_1 .. conv3_3: 178 x 178 for a 512 x 512 image) the rows / columns between the tensor edge and the image's
# reach -- 21 per side -- hold one value per channel (the regions above).  A 3x3 convolution maps equal neighbourhoods to equal outputs,
# so an interval [a, e) of them can be REMOVED before conv3_1 as long as, through all L = 3 layers, the rows on both sides of the cut are
# still "pure" (neither the zero padding of the layers so far nor the image has reached them): the kept rows then see the same values
# they would see in the full map and come out bit for bit the same.  a and e are even, so the 2x2 pooling windows keep their partners;
# behind pool3 the removed pooled rows are copies of a pure pooled row next to the cut.  12 rows / columns per side at 512 x 512:
# 154^2 instead of 178^2 pixels through conv3_x forward, dgrad and wgrad (-25 %).  Backward: the gradient of the copies is summed into
# the representative row / column (every parameter gradient upstream depends on the band only through such sums: the band pixels are
# produced by identical computations on identical values), the removed input rows get zero -- equal to the full computation up to the
# order of fp32 additions, like the other constant-border hints.  SZN_BAND_CROP=0 turns it off.
# The same holds for the conv2 block (conv2_1, conv2_2 at 355 x 355: 47 band rows per side, 40 removable: 275^2 pixels, -40 %; its kernels
# then run without the constant-border hint, which skipped 30 % of the forward tiles and none of the dgrad tiles) and, on the fp32 path
# -- whose kernels take no hints at all -- for conv1_2 (710 -> 526 rows / columns, -45 %).  On the 16-bit paths conv1_2 keeps its hints:
# cropping conv1_1's 516 MB output would cost what it saves.
_BAND_CROP = os.environ.get("SZN_BAND_CROP", "1") != "0"
_BAND_BLOCKS = {"conv1_2": ("conv1_2",), "conv2_1": ("conv2_1", "conv2_2"), "conv3_1": ("conv3_1", "conv3_2", "conv3_3")}   # first layer -> block
# (module constants, not environment knobs since round 6: tools/diag_band.py and the tests patch them)
_BAND_16BIT = ["conv2_1", "conv3_1"]      # blocks cropped through the generic szn_band_remap on the 16-bit paths (fp32: all three)
_BAND_C11 = True       # 16-bit paths: conv1_1 writes its output cropped (szn_conv1_1_fwd_c) and conv1_2's block runs on it
_BAND_FUSE = True      # adjacent cropped blocks hand over in one composed index map (False: copy the pooled rows back, crop again)
# default of SZN_FC6_CUMASK (see _Engine._masked_stream): CUs of the stream fc6's weight gradient + Adam runs on in a small step.  Off:
# measured on three boxes, "128:low" moved the one-image step by -0.13 / -0.04 ms for a caller on a non-blocking stream and by -0.05 / +0.02 ms
# for a caller on the null stream (profiles/r05_ablations.txt 15) -- inside the box-to-box spread
_POOL_GATHER = True   # the pools' backward pass applies the transposed band map while reading (szn_maxpool2x2_ceil_bwd_code_gather) instead of two szn_band_remap passes in front of it
_FC6_CUMASK = "0"
_SMALL_STEP_PX = 2 * 512 * 512


def _cumask_spec():
    """SZN_FC6_CUMASK = "<n>[:low|:even]" -> (n, how); n = 0: off"""
    n, _, how = os.environ.get("SZN_FC6_CUMASK", _FC6_CUMASK).partition(":")
    try:
        return max(int(n or 0), 0), (how or "low")
    except ValueError:
        raise L.SznError("SZN_FC6_CUMASK: expected <n>[:low|:even], got %r" % os.environ.get("SZN_FC6_CUMASK"))


_MASKED_HANDLES = []             # hipStream_t of every masked stream made (torch's ExternalStream does not own its handle)


@atexit.register
def _destroy_masked_streams():
    # before the HIP runtime's own teardown: a profiler's finalizer otherwise walks a queue the runtime has half taken down
    while _MASKED_HANDLES:
        h = _MASKED_HANDLES.pop()
        try:
            torch.cuda.synchronize()
            L.call("szn_stream_destroy", h)
        except Exception:
            pass


def masked_stream_wanted(pixels):
    """does a training step over `pixels` input pixels put fc6's fused weight gradient + Adam on a CU-masked stream?  (The same rule
    as the second stream of the weight gradients: small steps only, SZN_WGRAD_STREAM not 0.)"""
    return _cumask_spec()[0] > 0 and pixels <= _SMALL_STEP_PX and os.environ.get("SZN_WGRAD_STREAM", "auto") != "0"


def _band_cut(reg, n, L=3):
    """one axis: region (r0, r1, c0, c1) of the block's input map of size n -> [(a, e, rep_pooled, first_pooled_source), ...] for the
    top / left and bottom / right band (intervals of full rows to remove), possibly empty"""
    r0, r1, c0, c1 = reg
    cuts = []
    a = c0 + L + 2
    a += a & 1
    e = (r0 - L) & ~1
    if e - a >= 4:
        cuts.append((a, e, a // 2 - 1, a // 2 - 1))                    # representative pooled row = the one in front of the cut
    a2 = r1 + L
    a2 += a2 & 1
    e2 = (min(c1, n) - L - 2) & ~1
    if e2 - a2 >= 4 and c1 <= n:
        cuts.append((a2, e2, e2 // 2, a2 // 2))                        # ... = the one behind the cut
    return cuts


def _gather_form(dev, ty, tx):
    """a transposed band map (host tables [[start, count], ...] per axis) for the pools' backward pass: the runs with more than one source, to be
    folded in place first (szn_band_fold: rows, then columns), and the one-source tables szn_maxpool2x2_ceil_bwd_code_gather then reads through"""
    out = {}
    for key, t in (("y", ty), ("x", tx)):
        runs = [[s0, c] for s0, c in t if c > 1]
        out["runs_" + key] = dev(runs) if runs else None
    out["tabs"] = (dev([[s0, min(c, 1)] for s0, c in ty]), dev([[s0, min(c, 1)] for s0, c in tx]))
    return out


class _BandPlan(object):
    """index tables (device int32 [n][2] = start, count) of the four maps of szn_band_remap for one (regions, size) geometry"""

    def __init__(self, regy, regx, H, W, device, L=3):
        self.H, self.W = H, W
        ty, tx = self._axis(regy, H, L), self._axis(regx, W, L)
        cut8 = []
        for reg, n in ((regy, H), (regx, W)):       # {a, e, a2, e2} per axis for the kernels that take a cut (szn_conv1_1_fwd_c / _wgrad_c)
            top, bot = (0, 0), (n, n)
            for a, e, _, _ in _band_cut(reg, n, L):
                if e <= reg[0]:
                    top = (a, e)
                else:
                    bot = (a, e)
            cut8 += [top[0], top[1], bot[0], bot[1]]
        self.cut8 = (C.c_int * 8)(*cut8)
        self.ok = ty is not None and tx is not None and (ty["n"] < H or tx["n"] < W)
        if not self.ok:
            return
        self.Hc, self.Wc = ty["n"], tx["n"]                             # cropped size
        self.Hp, self.Wp = (H + 1) // 2, (W + 1) // 2                   # pooled, full
        self.Hpc, self.Wpc = (self.Hc + 1) // 2, (self.Wc + 1) // 2    # pooled, cropped
        dev = lambda t: torch.tensor(t, dtype=torch.int32, device=device).contiguous()
        self.tabs = {k: (dev(ty[k]), dev(tx[k])) for k in ("crop", "crop_bwd", "uncrop", "uncrop_bwd")}
        self.host = {k: (ty[k], tx[k]) for k in ("crop", "crop_bwd", "uncrop", "uncrop_bwd")}
        self._dev, self._fused = dev, {}
        # the gradient of the copied pooled rows in two separable passes (rows, then columns): a corner representative has
        # (n/2 + 1)^2 sources (441 in the conv2 block) -- one thread adding them all made that launch 132 us; 21 + 21 take 2 x 15
        ident = lambda n: dev([[i, 1] for i in range(n)])
        self.tabs["uncrop_bwd_y"] = (self.tabs["uncrop_bwd"][0], ident(self.Wp))
        self.tabs["uncrop_bwd_x"] = (ident(self.Hpc), self.tabs["uncrop_bwd"][1])
        self.gather_uncrop = _gather_form(dev, *self.host["uncrop_bwd"])
        self.x = None

    def fused_with(self, nxt):
        """this block's pooled, cropped output -> the NEXT block's cropped input in one map (and back), instead of copying the pooled rows
        back into the full map only to remove most of them again: tables {"fwd": one source per element, "bwd_y" / "bwd_x": the two
        separable passes of the transposed map}"""
        key = id(nxt)
        if key not in self._fused:
            dev = self._dev
            out = {}
            per_axis = []
            for ax in (0, 1):
                un, ub = self.host["uncrop"][ax], self.host["uncrop_bwd"][ax]
                cr, cb = nxt.host["crop"][ax], nxt.host["crop_bwd"][ax]
                fwd = [[un[y][0], 1] for y, _ in cr]
                bwd = []
                for s0, cnt in ub:
                    kept = [p for p in range(s0, s0 + cnt) if cb[p][1] == 1]
                    assert all(cb[kept[i + 1]][0] == cb[kept[i]][0] + 1 for i in range(len(kept) - 1))
                    bwd.append([cb[kept[0]][0], len(kept)] if kept else [0, 0])
                per_axis.append((fwd, bwd))
            ident = lambda n: [[i, 1] for i in range(n)]
            out["fwd"] = (dev(per_axis[0][0]), dev(per_axis[1][0]))
            out["bwd_y"] = (dev(per_axis[0][1]), dev(ident(nxt.Wc)))
            out["bwd_x"] = (dev(ident(self.Hpc)), dev(per_axis[1][1]))
            out["bwd_gather"] = _gather_form(dev, per_axis[0][1], per_axis[1][1])      # (the form the pool's backward pass takes)
            self._fused[key] = out
        return self._fused[key]

    @staticmethod
    def _axis(reg, n, L=3):
        cuts = _band_cut(reg, n, L)
        removed = set()
        for a, e, _, _ in cuts:
            removed.update(range(a, e))
        kept = [y for y in range(n) if y not in removed]
        pos = {y: i for i, y in enumerate(kept)}
        npool = (n + 1) // 2
        removed_p, rep_of = set(), {}
        for a, e, rep, _ in cuts:
            for p in range(a // 2, e // 2):
                removed_p.add(p)
                rep_of[p] = rep
        kept_p = [p for p in range(npool) if p not in removed_p]
        pos_p = {p: i for i, p in enumerate(kept_p)}
        if len(kept_p) != (len(kept) + 1) // 2:
            return None
        t = {"n": len(kept)}
        t["crop"] = [[y, 1] for y in kept]
        t["crop_bwd"] = [[pos[y], 1] if y in pos else [0, 0] for y in range(n)]
        t["uncrop"] = [[pos_p[p] if p in pos_p else pos_p[rep_of[p]], 1] for p in range(npool)]
        ub = [[p, 1] for p in kept_p]
        for a, e, rep, first in cuts:
            ub[pos_p[rep]] = [first, (e - a) // 2 + 1]
        t["uncrop_bwd"] = ub
        return t


class _Ctx(object):
    """what one forward pass leaves behind for its backward"""
    __slots__ = ("x", "acts", "pools", "relu6", "relu7", "masks", "coarse", "B", "H", "W", "h", "w", "train", "cb_in", "crop")


class _Engine(object):
    """Owns device-side weight images and runs the kernel sequence for one FCN32s instance."""

    def __init__(self, model):
        self.model = model
        self.dtype = torch.float32
        self._versions = None
        self._images = {}
        self._layer_versions = {}     # layer -> parameter versions its images were built from
        self.dropout_seed = 1337
        self.dropout_calls = 0
        self._splitk_ws = None
        self._gemm_ws = None          # fp32 Y of the GEMM + col2im dgrad (fc6)
        self._wg_ws = None            # slab workspace of the wgrad kernels (they run on their own stream)
        self._wg_stream = None        # torch.cuda.Stream, or False when disabled (SZN_WGRAD_STREAM=0)
        self._wg_masked = None        # torch.cuda.ExternalStream on part of the CUs (see _masked_stream); False = off
        # The un-pooled outputs of conv1_2 .. conv5_3 are read by nobody but the pool's backward pass: the forward pass writes a
        # one-byte winner code per pooled element instead (szn_conv_desc_t.pool_code) and tells the conv kernel it may skip the
        # un-pooled store (pool_only; conv3x3_regw does: 516 + 258 MB per step neither written nor read back).  keep_prepool = True
        # (tests that inspect the forward state) keeps those tensors valid; SZN_POOL_CODES=0: the round-1/2 backward from the tensor.
        self.keep_prepool = False
        # TrainStep, one rank: {layer: (AdamArgs, store the gradient too?)} -- the layer's weight update is applied in the epilogue of
        # its weight-gradient kernel (szn_conv2d_wgrad_adam); fused_done collects the layers for which that happened
        self.fused_opt = None
        self.fused_done = set()
        self.reserved_cus = 0         # CUs the persistent backward kernels leave to the RCCL queue (szn_conv_desc_t.reserved_cus; TrainStep)
        self.pool_codes = True        # False (tests / A-B): the pools' backward pass reads the un-pooled tensors (rounds 1-2) instead of winner codes
        self.head_fp8 = False         # forward of the projection head on the fp8 matrix cores (set_head_precision)
        self.head_fp8_bwd = False     # ... and its dgrad / wgrad (e5m2 gradient x e4m3 operands)
        self._fp8_ws = None
        self._fp8_bws = None
        self.lp_views = {}          # layer -> compute-dtype OHWI weight image maintained by the optimizer kernel (TrainStep)
        # bias gradients = column sums of the tensors the backward kernels write.  deterministic (default, SZN_DETERMINISTIC=0
        # turns it off): every producing kernel writes per-tile partial rows into a slab of this pool and ONE
        # szn_colsum_reduce_batch launch at the end of the backward pass adds them in a fixed order; otherwise fp32 atomics
        self.deterministic = os.environ.get("SZN_DETERMINISTIC", "1") != "0"
        self._cs_pool, self._cs_off, self._cs_jobs = None, 0, []
        self._seen_versions = None    # versions of seenmask_score mirrored into the TrainStep-owned head image
        # TrainStep with the rank-sharded optimizer: callable(layer) that makes the stream wait for the all-gather of the bucket holding that
        # layer's weight image (engine.TrainStep.wait_weights); None = nothing pending.  While it is set the dgrad images (_pack: one batched
        # launch that reads EVERY layer's image) are built at the end of the forward pass instead of at its start.
        self.weight_gate = None
        self._pack_pending = None

    def _workspace(self, desc, nbytes, device):
        """split-K scratch handed to the conv kernels (they use it only for few-tile / long-K shapes: fc6, fc7)"""
        if nbytes > (64 << 20):          # one fp32 slab of the output; layers this large never split
            return
        nbytes = min(16 * nbytes, 256 << 20)     # room for up to 16 K-splits (the library picks the count)
        if self._splitk_ws is None or self._splitk_ws.numel() < nbytes or self._splitk_ws.device != device:
            self._splitk_ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        desc.workspace, desc.workspace_bytes = self._splitk_ws.data_ptr(), self._splitk_ws.numel()

    # ---- weights ---------------------------------------------------------------------------------
    def set_precision(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise L.SznError("compute dtype must be float32, bfloat16 or float16")
        if dtype != self.dtype:
            self.dtype = dtype
            self._versions = None
            self._layer_versions = {}

    def mark_dirty(self):
        """parameters were rewritten behind torch's back (the optimizer kernel of TrainStep works on the flat buffers): every
        image is rebuilt by the next sync_weights"""
        self._versions = None
        self._layer_versions = {}

    def _param_versions(self):
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def _ohwi(self, p):
        """fp32 OHWI-contiguous storage of an (O,I,KH,KW) parameter (zero-copy when already channels_last)"""
        t = p.detach()
        if t.dtype != torch.float32:
            t = t.float()
        return t.permute(0, 2, 3, 1).contiguous()

    def sync_weights(self):
        """refresh compute-dtype / dgrad weight images if any parameter changed since the last call"""
        v = self._param_versions()
        if v == self._versions:
            return
        m = self.model
        dev = m.conv1_1.weight.device
        if dev.type != "cuda":
            raise L.SznError("FCN32s parameters live on %s: the HIP path needs them on the GPU (model.cuda())" % dev)
        dt, code = self.dtype, L.dtype_code(self.dtype)
        # per-layer refresh: only the layers whose parameters changed since the images were built are repacked (phase 2 and
        # fine-tuning touch the heads only; a full TrainStep touches everything)
        img = dict(self._images) if (self._images and self._layer_versions.get("_dtype") == (dt, dev)) else {}
        if not img:
            self._layer_versions = {"_dtype": (dt, dev)}
        lv = self._layer_versions
        st = L.stream_ptr()
        jobs = []                     # (image, flipped / transposed image, Co, K, Ci) to refresh: one batched launch at the end
        for name, co, ci, k in synth.CONV_LAYERS:
            layer = getattr(m, name)
            ver = (layer.weight.data_ptr(), layer.weight._version, layer.bias.data_ptr(), layer.bias._version,
                   id(self.lp_views.get(name)))
            if lv.get(name) == ver and (name + ".w") in img:
                continue
            lv[name] = ver
            w32 = self._ohwi(layer.weight)
            img[name + ".b"] = layer.bias.detach().float().contiguous()
            if name == "conv1_1":
                img[name + ".w"] = w32                       # conv1_1 kernel reads fp32 weights
                continue
            if dt == torch.float32:
                wc = w32
            elif name in self.lp_views and self.lp_views[name].dtype == dt:
                wc = self.lp_views[name]                     # already written by szn_adam_step / szn_sgd_momentum_step
            else:
                self._gate(name)                             # (a copy made NOW: the layer's master must be complete)
                wc = w32.to(dt)
            img[name + ".w"] = wc
            if k >= 5:      # fc6: dgrad runs as GEMM + col2im (szn_conv2d_dgrad_gemm*).  On the 16-bit paths the GEMM takes the
                # forward image itself (szn_conv2d_dgrad_gemm_native: no transposed copy of the 205 MB filter bank per step); the plain
                # transpose [k*k*ci][co] the other form needs is built on demand (_dgrad) and stamped with the weight version
                img[name + ".wG"] = img.get(name + ".wG")           # (buffer kept; its content is stale now)
                img[name + ".wG.stale"] = True
                continue
            wt = torch.empty(ci, k, k, co, device=dev, dtype=dt)
            jobs.append((wc, wt, co, k, ci))
            img[name + ".wT"] = wt
        # fused projection head: rows [0,E) = score_fr, [E,E+2) = seenmask_score, zero rows up to CP
        E, CP, F = m.n_class, m.head_width, m.fc7.out_channels
        hver = tuple((p.data_ptr(), p._version) for p in (m.score_fr.weight, m.score_fr.bias, m.seenmask_score.weight,
                                                          m.seenmask_score.bias)) + (id(self.lp_views.get("head")),)
        uver = (m.seenmask_upscore.weight.data_ptr(), m.seenmask_upscore.weight._version)
        if lv.get("up") != uver or "up.w" not in img:
            img["up.w"] = m.seenmask_upscore.weight.detach().float().contiguous()
            lv["up"] = uver
        if lv.get("head") == hver and "head.w" in img:
            self._pack_or_defer(jobs, code, st)
            self._images = img
            self._versions = v
            return
        lv["head"] = hver
        hv = self.lp_views.get("head")
        if hv is not None and hv[0].dtype == dt:
            # TrainStep keeps the image itself: score_fr's rows are written by the optimizer kernel, the rows behind them
            # hold seenmask_score (refreshed only when that parameter changes) and zeros -- nothing to assemble per step
            whc, bh = hv
            sv = (m.seenmask_score.weight._version, m.seenmask_score.bias._version)
            if sv != self._seen_versions:
                whc[E:E + 2].copy_(m.seenmask_score.weight.detach().reshape(2, F))
                bh[E:E + 2].copy_(m.seenmask_score.bias.detach())
                self._seen_versions = sv
        else:
            # assembled by copy (fp32 path / no TrainStep-owned image): with a sharded optimizer's all-gathers still pending this needs all of
            # them (score_fr is the last layer of the flat layout) -- the per-layer waits pay on the 16-bit paths, whose images are views
            self._gate("head")
            wh = torch.zeros(CP, F, device=dev, dtype=torch.float32)
            bh = torch.zeros(CP, device=dev, dtype=torch.float32)
            wh[:E] = m.score_fr.weight.detach().float().reshape(E, F)
            wh[E:E + 2] = m.seenmask_score.weight.detach().float().reshape(2, F)
            bh[:E] = m.score_fr.bias.detach().float()
            bh[E:E + 2] = m.seenmask_score.bias.detach().float()
            whc = wh if dt == torch.float32 else wh.to(dt)
        wht = torch.empty(F, CP, device=dev, dtype=dt)
        jobs.append((whc, wht, CP, 1, F))
        self._pack_or_defer(jobs, code, st)
        img["head.w"], img["head.b"], img["head.wT"] = whc.view(CP, 1, 1, F), bh, wht.view(F, 1, 1, CP)
        self._images = img
        self._versions = v

    def _gate(self, name):
        if self.weight_gate is not None:
            self.weight_gate(name)

    def _pack_or_defer(self, jobs, code, st):
        if self.weight_gate is not None:
            self._pack_pending = (self._pack_pending[0] if self._pack_pending else []) + list(jobs), code
        else:
            self._pack(jobs, code, st)

    def _flush_pack(self):
        """the deferred dgrad images (see weight_gate): every all-gather has been waited for by now"""
        if self._pack_pending:
            if self.weight_gate is not None:
                self.weight_gate(None)
            jobs, code = self._pack_pending
            self._pack_pending = None
            self._pack(jobs, code, L.stream_ptr())

    @staticmethod
    def _pack(jobs, code, st):
        """dgrad images of `jobs`: one szn_pack_weight_dgrad_batch launch for the 16-bit images it accepts (Co, Ci multiples of
        64), szn_pack_weight_dgrad for the rest"""
        batch = [j for j in jobs if code != L.SZN_F32 and j[2] % 64 == 0 and j[4] % 64 == 0]
        if len(batch) > 1:
            n = len(batch)
            VP, IA = C.c_void_p * n, C.c_int * n
            L.call("szn_pack_weight_dgrad_batch", code, n, VP(*[j[0].data_ptr() for j in batch]), VP(*[j[1].data_ptr() for j in batch]),
                   IA(*[j[2] for j in batch]), IA(*[j[3] for j in batch]), IA(*[j[4] for j in batch]), st)
        else:
            batch = []
        for wc, wt, co, k, ci in jobs:
            if not any(wt is b[1] for b in batch):
                L.call("szn_pack_weight_dgrad", code, co, k, k, ci, L.ptr(wc), L.ptr(wt), st)

    # ---- deterministic column sums (bias gradients) ------------------------------------------------
    def _cs_slab(self, M, Cc, device):
        """-> (fp32 slab tensor, rows it can hold) for a kernel that writes an (M, Cc) tensor, or (None, 0)"""
        if not self.deterministic:
            return None, 0
        rows = max((M + 255) // 256, min(2048, (M + 31) // 32), 512)
        n = (rows * Cc + 63) // 64 * 64
        if self._cs_pool is None or self._cs_pool.device != device or self._cs_off + n > self._cs_pool.numel():
            self._cs_pool = torch.empty(max(32 << 20, 2 * n), dtype=torch.float32, device=device)   # (jobs keep the old pool alive)
            self._cs_off = 0
        slab = self._cs_pool[self._cs_off:self._cs_off + rows * Cc]
        self._cs_off += n
        return slab, rows

    def _cs_register(self, slab, Cc, out, rows):
        """after a call that was handed `slab`: remember how many rows it wrote (rows: the call's own out-parameter --
        szn_conv_desc_t.result->colsum_rows / colsum_rows_out)"""
        if slab is None:
            return
        rows = int(getattr(rows, "value", rows))
        if rows > 0:
            self._cs_jobs.append((slab, rows, Cc, out))

    def _flush_colsum(self):
        """out[c] += the partial rows of every pending job, one launch, fixed order"""
        jobs, self._cs_jobs, self._cs_off = self._cs_jobs, [], 0
        if not jobs:
            return
        n = len(jobs)
        VP, IA = C.c_void_p * n, C.c_int * n
        L.call("szn_colsum_reduce_batch", n, VP(*[j[0].data_ptr() for j in jobs]), IA(*[j[1] for j in jobs]),
               IA(*[j[2] for j in jobs]), VP(*[j[3].data_ptr() for j in jobs]), L.stream_ptr())

    def _band_remap(self, x, plan, which, Ho, Wo):
        B, Hi, Wi, Cc = x.shape
        x = x.contiguous()
        out = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
        ty, tx = which if isinstance(which, tuple) else plan.tabs[which]
        L.call("szn_band_remap", L.dtype_code(x.dtype), B, Hi, Wi, Ho, Wo, Cc, L.ptr(x), L.ptr(out), L.ptr(ty), L.ptr(tx), L.stream_ptr())
        return out

    def _band_close(self, ctx, a, band, name, items, i, regy, regx):
        """behind a cropped block's pool: copy the pooled rows back into the full map -- unless the next block is cropped too: then the
        pooled map stays as it is and the next block's entry maps it straight into its own cropped coordinates (_BandPlan.fused_with).
        -> (tensor, pending)"""
        nxt = items[i + 2][0] if (i + 2 < len(items) and items[i + 2] != "P") else None
        if nxt in _BAND_BLOCKS and _BAND_FUSE and (self.dtype == torch.float32 or nxt in _BAND_16BIT):
            ny, nx = _cb_pool(regy, band.H), _cb_pool(regx, band.W)
            nplan = self._band_plan(ny, nx, band.Hp, band.Wp, a.device, len(_BAND_BLOCKS[nxt]))
            if nplan is not None:
                ctx.crop[("fusedout", name)] = True
                ctx.crop[("pool", name)] = band       # (pool_output() rebuilds the full map for callers that want it)
                return a, (band, nplan)
        return self._band_remap(a, band, "uncrop", band.Hp, band.Wp), None

    def pool_output(self, ctx, i):
        """the i-th pooling layer's output (NHWC, full map) of a forward pass that kept its state: ctx.pools[i][1], with the rows / columns a
        fused band map left out (conv2 -> conv3 blocks) copied back"""
        t = ctx.pools[i][1]
        producers = [it[0] for k, it in enumerate(_BACKBONE[:-1]) if _BACKBONE[k + 1] == "P"]
        band = (ctx.crop or {}).get(("pool", producers[i]))
        return t if band is None else self._band_remap(t, band, "uncrop", band.Hp, band.Wp)

    def _band_plan(self, regy, regx, H, W, device, L):
        key = (regy, regx, H, W, str(device), L)
        cache = self.__dict__.setdefault("_band_plans", {})
        if key not in cache:
            cache[key] = _BandPlan(regy, regx, H, W, device, L)
        return cache[key] if cache[key].ok else None

    # ---- kernels ---------------------------------------------------------------------------------
    def _conv(self, x, name, pad, relu=True, scale=None, out_f32=False, w=None, b=None, co=None, k=None, pool=False, codes=False,
              pool_only=False, cb=None):
        """conv (+ bias, ReLU, dropout factor) through szn_conv2d_fwd; pool=True also returns MaxPool2d(2,2,ceil) of the
        output (the descriptor's pool_out: fused into the epilogue of the kernels that support it)"""
        B, Hi, Wi, Ci = x.shape
        img = self._images
        if w is None:
            self._gate(name)
        w = img[name + ".w"] if w is None else w
        b = img[name + ".b"] if b is None else b
        co = w.shape[0] if co is None else co
        k = w.shape[1] if k is None else k
        Ho, Wo = Hi + 2 * pad - k + 1, Wi + 2 * pad - k + 1
        out = torch.empty(B, Ho, Wo, co, device=x.device, dtype=torch.float32 if out_f32 else self.dtype)
        d = L.ConvDesc(L.dtype_code(self.dtype), B, Hi, Wi, Ci, Ho, Wo, co, k, k, pad, Ci, co, 0, int(relu), int(out_f32))
        self._workspace(d, B * Ho * Wo * co * 4, x.device)
        pooled = code = None
        if pool:
            pooled = torch.empty(B, (Ho + 1) // 2, (Wo + 1) // 2, co, device=x.device, dtype=out.dtype)
            d.pool_out = pooled.data_ptr()
            if codes:
                code = torch.empty(B, (Ho + 1) // 2, (Wo + 1) // 2, co, device=x.device, dtype=torch.uint8)
                d.pool_code = code.data_ptr()
            d.pool_only = int(pool_only)
        if cb is not None and _CONST_BORDER and self.dtype != torch.float32:
            (ry, rx) = cb                                       # per-axis regions of THIS conv's output
            d.cb_on = 1
            d.cb_rect[0], d.cb_rect[1], d.cb_rect[2], d.cb_rect[3] = ry[0], ry[1], rx[0], rx[1]
            d.cb_const[0], d.cb_const[1], d.cb_const[2], d.cb_const[3] = ry[2], ry[3], rx[2], rx[3]
        L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(b), None, L.ptr(scale), L.ptr(out), L.stream_ptr())
        if pool and codes:
            return out, pooled, code
        return (out, pooled) if pool else out

    def make_masks(self, B, F, device):
        """Dropout2d factors (B,F) in {0, 2} for drop6 / drop7 (models.py:86,91; p = 0.5)"""
        masks = []
        for _ in range(2):
            mk = torch.empty(B, F, device=device, dtype=torch.float32)
            L.call("szn_dropout2d_mask", B * F, 0.5, self.dropout_seed, self.dropout_calls * (1 << 24), L.ptr(mk),
                   L.stream_ptr())
            self.dropout_calls += 1
            masks.append(mk)
        return masks

    def forward(self, x, train=False, masks=None, keep=True):
        """x (B,3,H,W) f32 NCHW on the GPU -> ctx with ctx.coarse (B,h,w,CP) f32: projection-head output at 1/32.
        keep=False (frozen backbone: phase 2, inference): the per-layer activations are not kept for a backward pass -- each
        one goes back to the allocator as soon as the next layer has consumed it; ctx then holds relu7 and coarse only"""
        if x.dim() != 4 or x.shape[1] != 3:
            raise L.SznError("FCN32s expects (B,3,H,W) input, got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise L.SznError("FCN32s input must live on the GPU (no CPU fallback in the product path)")
        self.sync_weights()
        x = x.contiguous().float()
        B, _, H, W = x.shape
        m = self.model
        code = L.dtype_code(self.dtype)
        ctx = _Ctx()
        ctx.x, ctx.B, ctx.H, ctx.W, ctx.train = x, B, H, W, train
        H1, W1 = H + 2 * PAD1 - 2, W + 2 * PAD1 - 2
        regy, regx = _cb_conv1_1(H, PAD1), _cb_conv1_1(W, PAD1)         # constant-border regions of the current tensor, per axis
        self._gate("conv1_1")
        ctx.crop = {}
        a = c11 = None
        # (the cropped pair has no fallback in the backward pass -- szn_conv1_1_wgrad_c is the fused kernel or nothing -- so the forward pass takes
        #  it only where that kernel's own admission rule holds: full-map gradient and image below 2 GiB, i.e. B < 34 at 512 x 512)
        c11_fits = B * H1 * W1 * 128 < 0x7fff0000 and B * 3 * H * W * 4 < 0x7fff0000
        if _BAND_CROP and _BAND_C11 and c11_fits and self.dtype != torch.float32 and not self.keep_prepool and (self.pool_codes or not keep):
            # 16-bit paths: conv1_1 never stores the rows / columns of the constant band that conv1_2's block does without (92 per side
            # at 512 x 512: 526^2 instead of 710^2 pixels written here, read by conv1_2, pooled, and walked by the backward pass)
            c11 = self._band_plan(regy, regx, H1, W1, x.device, len(_BAND_BLOCKS["conv1_2"]))
            if c11 is not None:
                a = torch.empty(B, c11.Hc, c11.Wc, 64, device=x.device, dtype=self.dtype)
                try:
                    L.call("szn_conv1_1_fwd_c", code, B, H, W, PAD1, L.ptr(x), L.ptr(self._images["conv1_1.w"]),
                           L.ptr(self._images["conv1_1.b"]), L.ptr(a), c11.cut8, L.stream_ptr())
                except L.SznError:                                      # (a kernel variant that writes full maps only)
                    a = c11 = None
        if a is None:
            a = torch.empty(B, H1, W1, 64, device=x.device, dtype=self.dtype)
            L.call("szn_conv1_1_fwd", code, B, H, W, PAD1, L.ptr(x), L.ptr(self._images["conv1_1.w"]),
                   L.ptr(self._images["conv1_1.b"]), L.ptr(a), L.stream_ptr())
        acts, pools = ({"conv1_1": a} if keep else {}), []
        items = _BACKBONE[1:]
        cb_in = ctx.cb_in = {}
        band = pending = None
        for i, item in enumerate(items):
            if item == "P":
                regy, regx = _cb_pool(regy, a_hw[0]), _cb_pool(regx, a_hw[1])
                continue                                  # pooled by the conv in front of it (pool_out)
            name, pad = item
            if name in _BAND_BLOCKS and _BAND_CROP and not self.keep_prepool and (self.pool_codes or not keep) and \
                    (self.dtype == torch.float32 or name in _BAND_16BIT or (i == 0 and c11 is not None)):
                # the constant band inside a conv block: remove most of it (see _band_cut), put the pooled rows back behind the block's pool
                blk = _BAND_BLOCKS[name]
                if i == 0 and c11 is not None:            # conv1_1 wrote the cropped map itself
                    band = c11
                    ctx.crop["c11cut"] = c11.cut8
                elif pending is not None:                 # the previous block's pooled output is still in ITS cropped coordinates
                    prev_band, band = pending
                    a = self._band_remap(a, band, prev_band.fused_with(band)["fwd"], band.Hc, band.Wc)
                    ctx.crop[("fused", name)] = prev_band
                    pending = None
                else:
                    band = self._band_plan(regy, regx, a.shape[1], a.shape[2], a.device, len(blk))
                    if band is not None:
                        a = self._band_remap(a, band, "crop", band.Hc, band.Wc)
                if band is not None:
                    ctx.crop[("in", name)] = (band, a)    # (the block's cropped input: its first layer's weight gradient reads it)
                    ctx.crop[("out", blk[-1])] = band
                    if i == 0 and keep:
                        acts["conv1_1"] = None            # conv1_1's full output: nothing reads it any more (fp32: 1 GB at B = 8)
            a_hw = (a.shape[1], a.shape[2]) if band is None else (band.H, band.W)       # 3x3 / pad 1: the conv's output size (full map)
            cb_in[name] = (regy, regx) if band is None else None    # regions of this conv's INPUT (its weight gradient can use them)
            regy, regx = _cb_conv3x3(regy, a_hw[0]), _cb_conv3x3(regx, a_hw[1])
            cb = (regy, regx) if band is None else None
            if i + 1 < len(items) and items[i + 1] == "P":
                if keep and self.pool_codes:
                    pin, a, code = self._conv(a, name, pad, pool=True, codes=True, pool_only=not self.keep_prepool, cb=cb)
                    acts[name] = pin if self.keep_prepool else None       # (may be unwritten: the backward pass takes the codes)
                    if band is not None:
                        a, pending = self._band_close(ctx, a, band, name, items, i, regy, regx)
                    pools.append((acts[name], a, code, tuple(pin.shape)))
                else:
                    pin, a = self._conv(a, name, pad, pool=True, pool_only=not keep, cb=cb)
                    if band is not None:
                        a, pending = self._band_close(ctx, a, band, name, items, i, regy, regx)
                    if keep:
                        acts[name] = pin
                        pools.append((pin, a))
                del pin
                band = None
            else:
                a = self._conv(a, name, pad, cb=cb)
                if keep:
                    acts[name] = a
        if train and masks is None:
            masks = self.make_masks(B, m.fc6.out_channels, x.device)
        ctx.masks = masks if train else None
        s6 = ctx.masks[0] if ctx.masks is not None else None
        s7 = ctx.masks[1] if ctx.masks is not None else None
        ctx.relu6 = self._conv(a, "fc6", 0, scale=s6)            # relu -> dropout factor, fused epilogue
        del a
        ctx.relu7 = self._conv(ctx.relu6, "fc7", 0, scale=s7)
        if not keep:
            ctx.relu6 = None
        ctx.acts, ctx.pools = acts, pools
        if self.head_fp8:
            self._gate("head")
        ctx.coarse = self._head_fp8(ctx.relu7) if self.head_fp8 else self._conv(ctx.relu7, "head", 0, relu=False, out_f32=True)
        ctx.h, ctx.w = ctx.coarse.shape[1:3]
        self._flush_pack()
        return ctx

    def _head_fp8(self, feat):
        """score_fr || seenmask_score as ONE fp8 (e4m3) GEMM with per-tensor scales (szn_proj_fp8_fwd; BASELINE configs[4]).
        The backward pass differentiates the 16-bit / fp32 head (straight-through) unless head_fp8_bwd is set."""
        B, h, w, F = feat.shape
        CP = self.model.head_width
        wimg, bimg = self._images["head.w"], self._images["head.b"]
        nb = L.load().szn_proj_fp8_workspace_bytes(B * h * w, F, CP)
        if self._fp8_ws is None or self._fp8_ws.numel() < nb or self._fp8_ws.device != feat.device:
            self._fp8_ws = torch.empty(nb, dtype=torch.uint8, device=feat.device)
        out = torch.empty(B, h, w, CP, device=feat.device, dtype=torch.float32)
        L.call("szn_proj_fp8_fwd", L.dtype_code(feat.dtype), L.dtype_code(wimg.dtype), B * h * w, F, CP, CP, L.ptr(feat),
               L.ptr(wimg), L.ptr(bimg), L.ptr(out), L.ptr(self._fp8_ws), L.stream_ptr())
        return out

    def _fp8_bwd_ws(self, M, F, CP, dev):
        nb = L.load().szn_proj_fp8_bwd_workspace_bytes(M, F, CP)
        if self._fp8_bws is None or self._fp8_bws.numel() < nb or self._fp8_bws.device != dev:
            self._fp8_bws = torch.empty(nb, dtype=torch.uint8, device=dev)
        return self._fp8_bws

    def _head_wgrad_fp8(self, feat, dc, dwh, dbh):
        """d(score_fr || seenmask_score weight) = dc^T . feat with the gradient in e5m2 and the features in e4m3
        (szn_proj_fp8_wgrad); the bias gradient is the plain column sum of dc"""
        B, h, w, F = feat.shape
        CP, M = self.model.head_width, B * h * w
        ws = self._fp8_bwd_ws(M, F, CP, feat.device)
        st = L.stream_ptr()
        L.call("szn_proj_fp8_wgrad", L.dtype_code(dc.dtype), L.dtype_code(feat.dtype), M, F, CP, CP, F, L.ptr(dc), L.ptr(feat),
               L.ptr(dwh), L.ptr(ws), st)
        if dbh is not None:
            slab, rows = self._cs_slab(M, CP, dc.device)
            ro = L.rows_out()
            L.call("szn_bias_grad_slab", L.dtype_code(dc.dtype), M, CP, CP, L.ptr(dc), L.ptr(dbh), 0, L.ptr(slab), rows, C.byref(ro), st)
            self._cs_register(slab, CP, dbh, ro)

    def _head_dgrad_fp8(self, dc, feat, scale, colsum):
        """d(fc7 output) = dc . W_head on the fp8 matrix cores (szn_proj_fp8_dgrad) with the ReLU gate / Dropout2d factor of fc7
        in the epilogue; `colsum` (fc7's bias gradient, pre-zeroed) receives the column sums of the result"""
        B, h, w, F = feat.shape
        CP, M = self.model.head_width, B * h * w
        ws = self._fp8_bwd_ws(M, F, CP, feat.device)
        wimg = self._images["head.w"]
        code = L.dtype_code(self.dtype)
        d = torch.empty(B, h, w, F, device=feat.device, dtype=self.dtype)
        st = L.stream_ptr()
        L.call("szn_proj_fp8_dgrad", L.dtype_code(dc.dtype), L.dtype_code(wimg.dtype), M, F, CP, CP, L.ptr(dc), L.ptr(wimg),
               L.ptr(feat), L.dtype_code(feat.dtype), F, L.ptr(scale), h * w, code, L.ptr(d), F, L.ptr(ws), st)
        if colsum is not None:
            slab, rows = self._cs_slab(M, F, d.device)
            ro = L.rows_out()
            L.call("szn_bias_grad_slab", code, M, F, F, L.ptr(d), L.ptr(colsum), 1, L.ptr(slab), rows, C.byref(ro), st)
            self._cs_register(slab, F, colsum, ro)
        return d

    def upscore(self, ctx):
        """coarse -> f (B,E,H,W) f32 NCHW: fixed bilinear ConvTranspose2d + crop (models.py:146-147)"""
        m = self.model
        f = torch.empty(ctx.B, m.n_class, ctx.H, ctx.W, device=ctx.coarse.device, dtype=torch.float32)
        L.call("szn_bilinear_up32_crop_fwd", ctx.B, ctx.h, ctx.w, m.n_class, m.head_width, 0, ctx.H, ctx.W, CROP,
               L.ptr(ctx.coarse), L.ptr(f), L.stream_ptr())
        return f

    def seenmask_upscore(self, ctx):
        """coarse -> s (B,2,H,W): learned dense ConvTranspose2d + crop (models.py:150-151)"""
        m = self.model
        s = torch.empty(ctx.B, 2, ctx.H, ctx.W, device=ctx.coarse.device, dtype=torch.float32)
        L.call("szn_deconv64s32_fwd", ctx.B, ctx.h, ctx.w, 2, m.head_width, m.n_class, ctx.H, ctx.W, CROP,
               L.ptr(ctx.coarse), L.ptr(self._images["up.w"]), L.ptr(s), L.stream_ptr())
        return s

    # ---- backward --------------------------------------------------------------------------------
    def head_backward(self, ctx, df=None, ds=None, dcoarse=None):
        """(df, ds) gradients of the two full-resolution outputs -> dcoarse (B,h,w,CP) f32 (+ d seenmask_upscore.weight)"""
        m = self.model
        dev = ctx.coarse.device
        if dcoarse is None:
            dcoarse = torch.zeros(ctx.B, ctx.h, ctx.w, m.head_width, device=dev, dtype=torch.float32)
        dup = None
        if df is not None:
            df = df.contiguous().float()
            L.call("szn_bilinear_up32_crop_bwd", ctx.B, ctx.h, ctx.w, m.n_class, m.head_width, 0, ctx.H, ctx.W, CROP,
                   L.ptr(df), L.ptr(dcoarse), L.stream_ptr())
        if ds is not None:
            ds = ds.contiguous().float()
            L.call("szn_deconv64s32_dgrad", ctx.B, ctx.h, ctx.w, 2, m.head_width, m.n_class, ctx.H, ctx.W, CROP,
                   L.ptr(ds), L.ptr(self._images["up.w"]), L.ptr(dcoarse), L.stream_ptr())
            dup = torch.empty(2, 2, 64, 64, device=dev, dtype=torch.float32)
            L.call("szn_deconv64s32_wgrad", ctx.B, ctx.h, ctx.w, 2, m.head_width, m.n_class, ctx.H, ctx.W, CROP,
                   L.ptr(ctx.coarse), L.ptr(ds), L.ptr(dup), 0, L.stream_ptr())
        return dcoarse, dup

    @contextlib.contextmanager
    def _wgrad_stream(self, *tensors, heavy=False):
        """Weight gradients are leaves of the backward chain: with SZN_WGRAD_STREAM=1 they run on a second HIP stream, so that the tail
        of a dgrad launch and the head of the weight-gradient launch beside it overlap.  Round 1 measured no gain at B=8 (158.1 vs
        158.8 Mpx/s); at the end of round 4 -- kernels 40 % faster, so the tails weigh more -- it is worth 0.07 ms per step (8.91 ->
        8.84, four alternating runs) and 0.09 ms at B = 1 (profiles/r04_ablations.txt 19).  Off by default all the same: two
        kernels sharing the chip make every per-kernel time (the bench line's `roofline`, the rocprof tables) a measurement of the
        pair, not of the kernel -- conv_igemm_8ph reads 0.34 of peak instead of 0.51 with it on.  Same kernels, same values.
        The side stream first waits for everything queued on the current stream (dout is produced there); `tensors`
        are the operands whose memory must not be recycled before the side stream is done with them."""
        if not getattr(self, "_wg_on", False):
            yield
            return
        if self._wg_stream is None:
            self._wg_stream = torch.cuda.Stream(device=tensors[0].device)
        side = self._wg_stream
        dev = tensors[0].device
        if heavy and torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev):      # (never next to the null stream: see TrainStep.step)
            side = self._masked_stream(dev) or side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            yield
        for t in tensors:
            t.record_stream(side)

    def _masked_stream(self, dev):
        """The stream of the ONE launch of a small step that would otherwise take every CU for half a millisecond: fc6's weight gradient
        + Adam step (2.7 GB of optimizer state streamed once, HBM-bound; two 72-KiB blocks per CU, so that the 120-KiB dgrad kernels of
        conv5_x .. conv3_x -- 8-124 tiles each -- found no CU to start on and the main stream stood still behind it: #48 / #49 of
        profiles/r05_g_b1_timeline.md).  On a stream confined to SZN_FC6_CUMASK = "<n>[:low|:even]" compute units (szn_stream_create_cu_mask)
        it leaves the rest of the chip to the backward chain.  0 = the plain side stream.  None when masking is off / unavailable."""
        if self._wg_masked is None:
            self._wg_masked = False
            n, how = _cumask_spec()
            info = L.DeviceInfo()
            L.call("szn_device_info", dev.index or 0, C.byref(info))
            ncu = info.compute_units
            if 0 < n < ncu:
                words = (ncu + 31) // 32
                mask = (C.c_uint32 * words)()
                picked = range(n) if how == "low" else sorted({(i * ncu) // n for i in range(n)})
                for cu in picked:
                    mask[cu // 32] |= 1 << (cu % 32)
                h = C.c_void_p()
                L.call("szn_stream_create_cu_mask", words, mask, C.byref(h))
                self._wg_masked = torch.cuda.ExternalStream(h.value, device=dev)
                _MASKED_HANDLES.append(h)
        return self._wg_masked or None

    def _join_wgrad(self):
        """end of a backward pass: the weight-gradient streams (if any) rejoin, the pending bias-gradient rows are reduced"""
        if self._wg_stream and getattr(self, "_wg_on", False):
            torch.cuda.current_stream().wait_stream(self._wg_stream)
            if self._wg_masked:
                torch.cuda.current_stream().wait_stream(self._wg_masked)
        self._flush_colsum()

    def _conv1_1_dgrad_cb(self, ctx):
        """the constant-border hint of the dgrad that feeds conv1_1 (conv1_2's): (where the gate -- conv1_1's output -- varies, what
        szn_conv1_1_wgrad reads of the result); None when that kernel reads everything"""
        ry, rx = _cb_conv1_1(ctx.H, PAD1), _cb_conv1_1(ctx.W, PAD1)
        # (returns 1 and a proper sub-rectangle only when szn_conv1_1_wgrad takes its fused kernel for these arguments -- otherwise
        # `reads` is the whole map and nothing may be skipped; the library refuses to fall back to a kernel that reads more than it
        # reported)
        reads = (C.c_int * 4)()
        sub = L.load().szn_conv1_1_wgrad_reads(L.dtype_code(self.dtype), ctx.B, ctx.H, ctx.W, PAD1, reads)
        if sub not in (0, 1):
            raise L.SznError("szn_conv1_1_wgrad_reads failed (%d)" % sub)
        return ((ry[0], ry[1], rx[0], rx[1]), tuple(reads))

    def _wgrad_desc(self, x, dout_shape, ci, co, k, pad, ldo=None, cb=None):
        """the descriptor of one layer's szn_conv2d_wgrad call (with the slab workspace and, on the 16-bit paths, the constant-border
        hint of the layer's INPUT x: cb = its per-axis regions)"""
        B, Hi, Wi, _ = x.shape
        Ho, Wo = dout_shape[1:3]
        ldo = dout_shape[3] if ldo is None else ldo
        code = L.dtype_code(self.dtype)
        d = L.ConvDesc(code, B, Hi, Wi, ci, Ho, Wo, co, k, k, pad, ci, ldo, 0, 0, 0)
        d.reserved_cus = self.reserved_cus
        # slab workspace of the weight-gradient kernels: the all-taps kernel (szn_conv_wgrad_taps.hip: <= 256 blocks x 64*9*64
        # fp32, room for SZN_WGT_OVERSUB=2) and the pixel splits of conv_wgrad_v2 (head / skip layers, every f32 layer: fixed-order
        # slabs instead of fp32 atomics -- it takes as many splits as fit)
        nb = 2 * 256 * 64 * 9 * 64 * 4
        if self._wg_ws is None or self._wg_ws.device != x.device:
            self._wg_ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        d.workspace, d.workspace_bytes = self._wg_ws.data_ptr(), nb
        if cb is not None and _CONST_BORDER and self.dtype != torch.float32:
            (ry, rx) = cb                                       # per-axis regions of this conv's INPUT x
            d.cb_on = 1
            d.cb_rect[0], d.cb_rect[1], d.cb_rect[2], d.cb_rect[3] = ry[0], ry[1], rx[0], rx[1]
            d.cb_const[0], d.cb_const[1], d.cb_const[2], d.cb_const[3] = ry[2], ry[3], rx[2], rx[3]
        return d

    def _wgrad(self, x, dout, dw, db, ci, co, k, pad, ldo=None, after=None, fuse=None, cb=None, csum=None, dw_lp=None):
        """dw (OHWI f32) = wgrad of one layer, on the wgrad stream; `after` (e.g. the DDP bucket hook) runs there too.  fuse: layer
        name under which self.fused_opt may hold the Adam step to apply in the kernel's epilogue.  cb / csum: constant-border regions
        of x and, optionally, the column sums of dout over the tiles the hint skips (from the producer of dout).  dw_lp: deliver the
        gradient as a 16-bit image there instead (the wire buffer of the data-parallel exchange; dw is then scratch)."""
        B, Hi, Wi, _ = x.shape
        Ho, Wo = dout.shape[1:3]
        code = L.dtype_code(self.dtype)
        d = self._wgrad_desc(x, dout.shape, ci, co, k, pad, ldo=ldo, cb=cb)
        ldo = d.ldo
        if dw_lp is not None:
            d.dw_lp, d.dw_lp_dtype = dw_lp.data_ptr(), L.dtype_code(dw_lp.dtype)
        if csum is not None and d.cb_on:
            d.colsum = csum.data_ptr()
        opt = self.fused_opt.get(fuse) if (self.fused_opt and fuse) else None
        if opt is not None and L.load().szn_conv2d_wgrad_adam_supported(C.byref(d)) != 1:
            opt = None
        with self._wgrad_stream(*([x, dout] + ([csum] if csum is not None else [])), heavy=opt is not None):
            st = L.stream_ptr()
            if opt is not None:
                opt[0].grad_optional = 0 if opt[1] else 1          # (dw is handed over either way: the follower form needs it)
                L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), C.byref(opt[0]), st)
                self.fused_done.add(fuse)
            else:
                L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), 0, st)
            if db is not None:
                slab, rows = self._cs_slab(B * Ho * Wo, co, dout.device)
                ro = L.rows_out()
                L.call("szn_bias_grad_slab", code, B * Ho * Wo, co, ldo, L.ptr(dout), L.ptr(db), 0, L.ptr(slab), rows, C.byref(ro), st)
                self._cs_register(slab, co, db, ro)
            if after is not None:
                after()

    def _dgrad(self, dout, name, in_shape, pad, gate=None, scale=None, colsum=None, wT=None, cb=None, border_sum=None):
        """din = conv(dout, flipped weights) with the ReLU gate / dropout factor of the producing layer fused; colsum
        (f32 [Ci], pre-zeroed) receives the column sums of din = that layer's bias gradient"""
        B, Hi, Wi, Ci = in_shape
        Ho, Wo, Co = dout.shape[1:]
        din = torch.empty(B, Hi, Wi, Ci, device=dout.device, dtype=self.dtype)
        if wT is None and (name + ".wG") in self._images:
            # large window (fc6): GEMM + col2im, no gate / dropout factor / column sums on this edge
            if gate is not None or scale is not None or colsum is not None:
                raise L.SznError("dgrad of %s: the GEMM form has no gate / scale / colsum epilogue" % name)
            img = self._images
            wf = img[name + ".w"]                                  # forward image [Co][k][k][Ci]
            k = wf.shape[1]
            code = L.dtype_code(self.dtype)
            d = L.ConvDesc(code, B, Hi, Wi, Ci, Ho, Wo, Co, k, k, pad, Ci, Co, 0, 0, 0)
            lib = L.load()
            native = _FC6_NATIVE and dout.is_contiguous() and lib.szn_conv2d_dgrad_gemm_native_supported(C.byref(d)) == 1
            nb = (lib.szn_conv2d_dgrad_gemm_native_workspace_bytes if native else lib.szn_conv2d_dgrad_gemm_workspace_bytes)(C.byref(d))
            if self._gemm_ws is None or self._gemm_ws.numel() < nb or self._gemm_ws.device != dout.device:
                self._gemm_ws = torch.empty(nb, dtype=torch.uint8, device=dout.device)
            d.workspace, d.workspace_bytes = self._gemm_ws.data_ptr(), self._gemm_ws.numel()
            if native:
                L.call("szn_conv2d_dgrad_gemm_native", C.byref(d), L.ptr(dout), L.ptr(wf), L.ptr(din), L.stream_ptr())
                return din
            if img.get(name + ".wG.stale", True) or img[name + ".wG"] is None:
                if img[name + ".wG"] is None:
                    img[name + ".wG"] = torch.empty(k * k * Ci, Co, device=dout.device, dtype=self.dtype)
                L.call("szn_pack_weight_dgrad", code, Co, 1, 1, k * k * Ci, L.ptr(wf), L.ptr(img[name + ".wG"]), L.stream_ptr())
                img[name + ".wG.stale"] = False
            L.call("szn_conv2d_dgrad_gemm", C.byref(d), L.ptr(dout), L.ptr(img[name + ".wG"]), L.ptr(din), L.stream_ptr())
            return din
        wT = self._images[name + ".wT"] if wT is None else wT
        k = wT.shape[1]
        d = L.ConvDesc(L.dtype_code(self.dtype), B, Hi, Wi, Ci, Ho, Wo, Co, k, k, pad, Ci, Co, Ci, 0, 0)
        d.reserved_cus = self.reserved_cus
        slab = None
        M = B * Hi * Wi
        if colsum is not None:
            d.colsum = colsum.data_ptr()
            slab, rows = self._cs_slab(M, Ci, dout.device)
            if slab is not None:
                d.colsum_slab, d.colsum_slab_rows = slab.data_ptr(), rows
        # Few output tiles and a long reduction (conv4_x / conv5_x at the reference's batch size of one image: 32 .. 124 tiles of
        # 256 x 128 on 256 CUs): the library may split the K range (deterministic slabs); since round 5 its split-K epilogue also
        # produces the column sums (= the producer layer's bias gradient) from the fp32 values (splitk_epilogue_cs), so the
        # scratch is handed over whether or not column sums are asked for and the decision is the library's alone
        if colsum is None or (cb is None and _DGRAD_SPLIT):
            self._workspace(d, M * Ci * 4, dout.device)
        if cb is not None and gate is not None and _CONST_BORDER and self.dtype != torch.float32:
            grect, srect = cb                                   # (r0, r1, c0, c1) each: where the gate varies / what the consumer reads
            # border_sum: the sum of dout over the part of the map this call need not run (szn_conv2d_dgrad_border_region), from the
            # producer of dout -- the tiles outside are then skipped and their column sums added by szn_conv2d_dgrad_border_finish
            d.cb_on = 2 if (border_sum is not None and colsum is not None) else 1
            for i in range(4):
                d.cb_rect[i], d.cb_const[i] = grect[i], srect[i]
        L.call("szn_conv2d_dgrad", C.byref(d), L.ptr(dout), L.ptr(wT), L.ptr(gate), L.ptr(scale), L.ptr(din), L.stream_ptr())
        cs_rows = d.res.colsum_rows                    # (the call's own report: szn_conv_desc_t.result)
        if d.cb_on == 2 and d.res.work_fraction < 1.0:
            bws = torch.empty(2 * 24 * B * Co, device=dout.device)
            L.call("szn_conv2d_dgrad_border_finish", C.byref(d), L.ptr(dout), L.ptr(wT), L.ptr(gate), L.ptr(border_sum), L.ptr(colsum),
                   L.ptr(bws), L.stream_ptr())
        self._cs_register(slab, Ci, colsum, cs_rows)
        return din

    def backward(self, ctx, dcoarse, grads, backbone=True, layer_done=None, head_first=None, skips=None):
        """dcoarse (B,h,w,CP) f32 or compute dtype -> fills grads[name] = (dw OHWI f32, db f32) for every layer
        present in `grads` ('head' holds the fused score_fr||seenmask_score gradient, CP rows).  skips: {pool index: gradient
        arriving at that pool's OUTPUT from a side branch} (the FCN8s score_pool3 / score_pool4 taps), added where the chain
        reaches it."""
        m = self.model
        dt = self.dtype
        code = L.dtype_code(dt)
        st = L.stream_ptr()
        # weight gradients on a second stream (see _wgrad_stream): SZN_WGRAD_STREAM = 1 always, 0 never, auto (default) = small steps
        # only (at most two 512 x 512 images: the reference's one-image step, train.py:82-84) -- there the backward kernels run on
        # 8-124 tiles and leave most of the chip idle, so a dgrad and the weight gradient beside it really overlap (B = 1: -0.09 ms);
        # at B = 8 every kernel fills the chip, the gain is 0.07 ms and the per-kernel timings of bench.py / rocprof would describe
        # pairs of kernels
        mode = os.environ.get("SZN_WGRAD_STREAM", "auto")
        # (not while a hipGraph is being captured: replayed with the cross-stream edges the one-image step measured 2.90 ms against
        #  2.78 without them and 2.71 eager with them -- profiles/r05_ablations.txt 8)
        self._wg_on = (mode == "1" or (mode == "auto" and ctx.B * ctx.H * ctx.W <= _SMALL_STEP_PX)) and \
            not (mode != "1" and torch.cuda.is_current_stream_capturing())
        dc = dcoarse if dcoarse.dtype == dt else dcoarse.to(dt)      # tiny (B*h*w*CP)
        feat = ctx.relu7
        F = m.fc7.out_channels
        flat_bias = grads.get("_flat_bias")          # TrainStep: every bias gradient is a view of one flat buffer:
        if flat_bias is not None and backbone:       # one fill, BEFORE the head hook copies score_fr's gradient into it
            flat_bias.zero_()
        fp8b = self.head_fp8_bwd and self.head_fp8
        lp = grads.get("_lp") or {}                  # TrainStep, direct 16-bit wire: {layer: slice of the staging buffer}
        if "head" in grads:
            dwh, dbh = grads["head"]
            if fp8b:
                self._head_wgrad_fp8(feat, dc, dwh, dbh)
                if "head" in lp:
                    L.call("szn_cast", L.SZN_F32, L.dtype_code(lp["head"].dtype), dwh.numel(), L.ptr(dwh), L.ptr(lp["head"]), st)
                if head_first is not None:
                    head_first()
            else:
                self._wgrad(feat, dc, dwh, dbh, F, m.head_width, 1, 0, after=head_first, dw_lp=lp.get("head"))
        if not backbone:
            self._join_wgrad()
            return
        done = layer_done if layer_done is not None else (lambda name: None)
        s6 = ctx.masks[0] if ctx.masks is not None else None
        s7 = ctx.masks[1] if ctx.masks is not None else None
        # bias gradients are produced as column sums by whichever kernel WRITES the pre-activation gradient (the dgrad
        # of the next layer or the pool backward), so they start from zero here
        if flat_bias is None:
            for name in grads:
                if name != "head" and not name.startswith("_"):
                    grads[name][1].zero_()
        # d(fc7 pre-activation): ReLU gate (feat > 0) and dropout factor fused into the dgrad epilogue
        if fp8b:
            d = self._head_dgrad_fp8(dc, feat, s7, grads["fc7"][1])
        else:
            d = self._dgrad(dc, "head", feat.shape, 0, gate=feat, scale=s7, colsum=grads["fc7"][1])
        self._wgrad(ctx.relu6, d, grads["fc7"][0], None, F, F, 1, 0, after=lambda: done("fc7"), fuse="fc7", dw_lp=lp.get("fc7"))
        d = self._dgrad(d, "fc7", ctx.relu6.shape, 0, gate=ctx.relu6, scale=s6, colsum=grads["fc6"][1])
        pool5 = ctx.pools[4][1]
        if self.fused_opt and "fc6" in self.fused_opt:
            # the fused update rewrites fc6's 16-bit weight image, which fc6's dgrad reads in place: dgrad first
            d6 = d
            d = self._dgrad(d6, "fc6", pool5.shape, 0)
            self._wgrad(pool5, d6, grads["fc6"][0], None, pool5.shape[3], F, 7, 0, after=lambda: done("fc6"), fuse="fc6")
        else:
            self._wgrad(pool5, d, grads["fc6"][0], None, pool5.shape[3], F, 7, 0, after=lambda: done("fc6"), dw_lp=lp.get("fc6"))
            d = self._dgrad(d, "fc6", pool5.shape, 0)
        pi = 4
        prev_out = None
        pending_gather = None
        cb_sums, border_sums = {}, {}
        items = _BACKBONE
        for idx in range(len(items) - 1, -1, -1):
            item = items[idx]
            if item == "P":
                pin, pout = ctx.pools[pi][:2]
                pcode = ctx.pools[pi][2] if len(ctx.pools[pi]) > 2 else None
                pi -= 1
                producer = items[idx - 1][0]                   # the conv whose (ReLU'd) output this pool reads
                gather = None          # (ytab, xtab): d is still in the consumer's coordinates, this pool's backward pass reads it through the map
                if ctx.crop and ("out", producer) in ctx.crop and not ctx.crop.get(("fusedout", producer)):
                    band = ctx.crop[("out", producer)]         # the pooled rows that were copies: their gradients are summed
                    if _POOL_GATHER and pcode is not None:
                        gather = band.gather_uncrop
                    else:
                        d = self._band_remap(d, band, "uncrop_bwd_y", band.Hpc, band.Wp)
                        d = self._band_remap(d, band, "uncrop_bwd_x", band.Hpc, band.Wpc)
                elif pending_gather is not None:
                    if pcode is not None:
                        gather = pending_gather[0]
                    else:
                        for tab, ho, wo in pending_gather[1]:
                            d = self._band_remap(d, pending_gather[2], tab, ho, wo)
                pending_gather = None
                B, Hi, Wi, Cc = ctx.pools[pi + 1][3] if pcode is not None else pin.shape
                dn = torch.empty(B, Hi, Wi, Cc, device=d.device, dtype=pout.dtype)
                slab, rows = self._cs_slab(B * Hi * Wi, Cc, d.device)
                ro = L.rows_out()
                # constant-border hint of the producer's weight gradient (which reads dn next): let this pass sum dn over the tiles
                # that call is going to skip, instead of a second pass over them (szn_conv2d_wgrad_cb_tiles)
                regions, want = [], []
                cbp = (getattr(ctx, "cb_in", None) or {}).get(producer)
                if pcode is not None and slab is not None and _WGRAD_CB_FUSED and idx >= 2 and self.dtype != torch.float32 and _CONST_BORDER:
                    prev2 = items[idx - 2]
                    xin2 = ctx.pools[pi][1] if prev2 == "P" else ctx.acts[prev2[0]]
                    lay = getattr(m, producer)
                    if cbp is not None:
                        dd = self._wgrad_desc(xin2, (B, Hi, Wi, Cc), lay.in_channels, lay.out_channels, 3, items[idx - 1][1], cb=cbp)
                        r8 = (C.c_int * 8)()
                        if dd.cb_on and L.load().szn_conv2d_wgrad_cb_region(C.byref(dd), r8) == 1:
                            regions += list(r8)
                            want.append("w")
                    # (from two 512 x 512 images on: the three small launches that replace the tiles cost what they save on one image)
                    if prev2 != "P" and prev2[0] == "conv1_1" and _DGRAD_BORDER and B * Hi * Wi >= 1000000 and \
                            ("in", producer) not in (ctx.crop or {}):
                        # the producer's dgrad feeds conv1_1 only: the part of the map it can replace by region sums of dn
                        grect, srect = self._conv1_1_dgrad_cb(ctx)
                        ci2 = lay.in_channels
                        dq = L.ConvDesc(code, B, Hi, Wi, ci2, Hi, Wi, Cc, 3, 3, items[idx - 1][1], ci2, Cc, ci2, 0, 0)
                        dq.cb_on = 2
                        for i in range(4):
                            dq.cb_rect[i], dq.cb_const[i] = grect[i], srect[i]
                        r8 = (C.c_int * 8)()
                        if L.load().szn_conv2d_dgrad_border_region(C.byref(dq), r8) == 1 and all(v % 2 == 0 for v in r8):
                            regions += list(r8)
                            want.append("d")
                if regions and gather is not None:             # (never together in practice: hints belong to un-cropped producers)
                    raise L.SznError("pool backward of %s: a band map and constant-border regions at once" % producer)
                if regions:
                    n = len(want)
                    ssum = torch.empty(n, Cc, device=d.device)
                    slab2 = torch.empty(n * rows * Cc, device=d.device)
                    L.call("szn_maxpool2x2_ceil_bwd_code_cb", code, B, Hi, Wi, Cc, L.ptr(pcode), L.ptr(d), L.ptr(dn),
                           L.ptr(grads[producer][1]), L.ptr(slab), rows, C.byref(ro), (C.c_int * len(regions))(*regions), n, L.ptr(ssum),
                           L.ptr(slab2), st)
                    for i, kind in enumerate(want):
                        (cb_sums if kind == "w" else border_sums)[producer] = ssum[i]
                elif gather is not None:
                    # the few rows / columns that sum several sources are folded in place (rows, then columns: ~10 % of the tensor), the
                    # rest of the transposed map is a shift this pool's backward pass applies while it reads d
                    d = d.contiguous()
                    for axis, runs in ((0, gather["runs_y"]), (1, gather["runs_x"])):
                        if runs is not None:
                            L.call("szn_band_fold", code, B, d.shape[1], d.shape[2], Cc, L.ptr(d), axis, L.ptr(runs), runs.shape[0], st)
                    ty, tx = gather["tabs"]
                    L.call("szn_maxpool2x2_ceil_bwd_code_gather", code, B, Hi, Wi, Cc, L.ptr(pcode), L.ptr(d), d.shape[1], d.shape[2],
                           L.ptr(ty), L.ptr(tx), L.ptr(dn), L.ptr(grads[producer][1]), L.ptr(slab), rows, C.byref(ro), st)
                elif pcode is not None:
                    L.call("szn_maxpool2x2_ceil_bwd_code", code, B, Hi, Wi, Cc, L.ptr(pcode), L.ptr(d), L.ptr(dn),
                           L.ptr(grads[producer][1]), L.ptr(slab), rows, C.byref(ro), st)
                else:
                    L.call("szn_maxpool2x2_ceil_bwd", code, B, Hi, Wi, Cc, L.ptr(pin), L.ptr(pout), L.ptr(d), L.ptr(dn),
                           L.ptr(grads[producer][1]), L.ptr(slab), rows, C.byref(ro), st)
                self._cs_register(slab, Cc, grads[producer][1], ro)
                d = dn
                continue
            name, pad = item
            if name == "conv1_1":
                dw, db = grads[name]
                nb = L.load().szn_conv1_1_wgrad_workspace_bytes(code, ctx.B, ctx.H, ctx.W, PAD1)
                ws = torch.empty(nb, dtype=torch.uint8, device=d.device)
                with self._wgrad_stream(d, ws):
                    if ctx.crop and "c11cut" in ctx.crop:      # d is the cropped map conv1_2's dgrad wrote
                        L.call("szn_conv1_1_wgrad_c", code, ctx.B, ctx.H, ctx.W, PAD1, L.ptr(ctx.x), L.ptr(d), L.ptr(dw), 0,
                               L.ptr(ws), ctx.crop["c11cut"], L.stream_ptr())
                    else:
                        L.call("szn_conv1_1_wgrad", code, ctx.B, ctx.H, ctx.W, PAD1, L.ptr(ctx.x), L.ptr(d), L.ptr(dw), None, 0,
                               L.ptr(ws), L.stream_ptr())      # db came from conv1_2's dgrad (colsum)
                    if name in lp:                             # 1,728 elements: converted, not produced, as 16-bit
                        L.call("szn_cast", L.SZN_F32, L.dtype_code(lp[name].dtype), dw.numel(), L.ptr(dw), L.ptr(lp[name]), L.stream_ptr())
                    done(name)
                break
            prev = items[idx - 1]
            xin = ctx.pools[pi][1] if prev == "P" else ctx.acts[prev[0]]
            cropped_in = bool(ctx.crop) and ("in", name) in ctx.crop
            if cropped_in:
                xin = ctx.crop[("in", name)][1]
            layer = getattr(m, name)
            self._wgrad(xin, d, grads[name][0], None, layer.in_channels, layer.out_channels, 3, pad,
                        after=lambda name=name: done(name), cb=(getattr(ctx, "cb_in", None) or {}).get(name),
                        csum=cb_sums.pop(name, None), dw_lp=lp.get(name))
            # next d: wrt this conv's input; gate by the ReLU of the producing conv unless a pool sits in between
            if prev == "P":
                d = self._dgrad(d, name, xin.shape, pad)
                if cropped_in:                                 # the removed rows / columns of the block's input get no gradient
                    band = ctx.crop[("in", name)][0]
                    prev_band = ctx.crop.get(("fused", name))
                    if prev_band is not None:                  # ... straight into the previous block's pooled, cropped coordinates
                        ft = prev_band.fused_with(band)
                        two = [(ft["bwd_y"], prev_band.Hpc, band.Wc), (ft["bwd_x"], prev_band.Hpc, prev_band.Wpc)]
                        if _POOL_GATHER and not (skips and skips.get(pi) is not None):
                            pending_gather = (ft["bwd_gather"], two, band)      # applied by the pool's backward pass while it reads d
                        else:
                            for tab, ho, wo in two:
                                d = self._band_remap(d, band, tab, ho, wo)
                    else:
                        d = self._band_remap(d, band, "crop_bwd", band.H, band.W)
                side = skips.get(pi) if skips else None
                if side is not None:
                    d = d + side.to(d.dtype)
            else:
                cb = self._conv1_1_dgrad_cb(ctx) if (prev[0] == "conv1_1" and not cropped_in) else None
                d = self._dgrad(d, name, xin.shape, pad, gate=xin, colsum=grads[prev[0]][1], cb=cb, border_sum=border_sums.pop(name, None))
                if cropped_in and "c11cut" not in ctx.crop:    # (fp32: conv1_2's block; conv1_1's weight gradient reads the full map)
                    band = ctx.crop[("in", name)][0]
                    d = self._band_remap(d, band, "crop_bwd", band.H, band.W)
        self._join_wgrad()


def _backbone_backward(ctx, dcoarse, skips=None):
    """shared by _Backbone / _Backbone8: run the dgrad / wgrad chain and hand every parameter its gradient"""
    model, c = ctx.model, ctx.c
    eng = model._engine
    dev = dcoarse.device
    need = ctx.needs_input_grad[4:]
    names = [n for n, _ in model.named_parameters()]
    need_of = dict(zip(names, need))
    backbone = any(need_of.get(n + ".weight", False) for n in _TRUNK)
    grads = {}
    for name, co, ci, k in synth.CONV_LAYERS:
        if backbone:
            grads[name] = (torch.empty(co, k, k, ci, device=dev), torch.empty(co, device=dev))
    CP, F, E = model.head_width, model.fc7.out_channels, model.n_class
    grads["head"] = (torch.empty(CP, 1, 1, F, device=dev), torch.empty(CP, device=dev))
    eng.backward(c, dcoarse.contiguous(), grads, backbone=backbone, skips=skips)
    out = []
    for n in names:
        layer, kind = n.rsplit(".", 1)
        if not need_of[n] or layer in ("upscore", "seenmask_upscore"):
            out.append(None)
        elif layer == "score_fr":
            g = grads["head"][0][:E].reshape(E, F, 1, 1) if kind == "weight" else grads["head"][1][:E]
            out.append(g.clone())
        elif layer == "seenmask_score":
            g = grads["head"][0][E:E + 2].reshape(2, F, 1, 1) if kind == "weight" else grads["head"][1][E:E + 2]
            out.append(g.clone())
        elif layer in grads:
            out.append(grads[layer][0].permute(0, 3, 1, 2) if kind == "weight" else grads[layer][1])
        else:
            out.append(None)
    return (None, None, None, None) + tuple(out)


class _Backbone(torch.autograd.Function):
    """autograd bridge: parameters -> coarse projection map; backward runs the HIP dgrad/wgrad chain"""

    @staticmethod
    def forward(ctx, model, x, train, masks, *params):
        eng = model._engine
        c = eng.forward(x, train=train, masks=masks)
        ctx.model, ctx.c = model, c
        model._last_ctx = c
        return c.coarse

    @staticmethod
    def backward(ctx, dcoarse):
        return _backbone_backward(ctx, dcoarse)


class _Upscore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, coarse):
        c = model._last_ctx
        ctx.model, ctx.c = model, c
        return model._engine.upscore(c)

    @staticmethod
    def backward(ctx, df):
        dcoarse, _ = ctx.model._engine.head_backward(ctx.c, df=df)
        return None, dcoarse


class _SeenmaskUpscore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, coarse, weight):
        c = model._last_ctx
        ctx.model, ctx.c = model, c
        return model._engine.seenmask_upscore(c)

    @staticmethod
    def backward(ctx, ds):
        dcoarse, dup = ctx.model._engine.head_backward(ctx.c, ds=ds)
        return None, dcoarse, dup


class FCN32s(nn.Module):
    """Reference models.py:27-193.  n_class = embedding dimension E (train.py:102-105)."""

    pretrained_model = 'data/fcn32s_from_caffe.pth'

    def __init__(self, n_class=21):
        super(FCN32s, self).__init__()
        self.n_class = n_class
        self.head_width = _round_up(n_class + 2, 64)
        chans = {n: (ci, co, k) for n, co, ci, k in synth.CONV_LAYERS}

        def conv(name, pad):
            ci, co, k = chans[name]
            return nn.Conv2d(ci, co, k, padding=pad)

        # conv1
        self.conv1_1 = conv("conv1_1", PAD1); self.relu1_1 = nn.ReLU(inplace=True)
        self.conv1_2 = conv("conv1_2", 1); self.relu1_2 = nn.ReLU(inplace=True)
        self.pool1 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        # conv2
        self.conv2_1 = conv("conv2_1", 1); self.relu2_1 = nn.ReLU(inplace=True)
        self.conv2_2 = conv("conv2_2", 1); self.relu2_2 = nn.ReLU(inplace=True)
        self.pool2 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        # conv3
        self.conv3_1 = conv("conv3_1", 1); self.relu3_1 = nn.ReLU(inplace=True)
        self.conv3_2 = conv("conv3_2", 1); self.relu3_2 = nn.ReLU(inplace=True)
        self.conv3_3 = conv("conv3_3", 1); self.relu3_3 = nn.ReLU(inplace=True)
        self.pool3 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        # conv4
        self.conv4_1 = conv("conv4_1", 1); self.relu4_1 = nn.ReLU(inplace=True)
        self.conv4_2 = conv("conv4_2", 1); self.relu4_2 = nn.ReLU(inplace=True)
        self.conv4_3 = conv("conv4_3", 1); self.relu4_3 = nn.ReLU(inplace=True)
        self.pool4 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        # conv5
        self.conv5_1 = conv("conv5_1", 1); self.relu5_1 = nn.ReLU(inplace=True)
        self.conv5_2 = conv("conv5_2", 1); self.relu5_2 = nn.ReLU(inplace=True)
        self.conv5_3 = conv("conv5_3", 1); self.relu5_3 = nn.ReLU(inplace=True)
        self.pool5 = nn.MaxPool2d(2, stride=2, ceil_mode=True)
        # fc6 / fc7
        self.fc6 = nn.Conv2d(512, 4096, 7); self.relu6 = nn.ReLU(inplace=True); self.drop6 = nn.Dropout2d()
        self.fc7 = nn.Conv2d(4096, 4096, 1); self.relu7 = nn.ReLU(inplace=True); self.drop7 = nn.Dropout2d()
        # heads
        self.score_fr = nn.Conv2d(4096, n_class, 1)
        self.upscore = nn.ConvTranspose2d(n_class, n_class, 64, stride=32, bias=False)
        self.seenmask_score = nn.Conv2d(4096, 2, 1)
        self.seenmask_upscore = nn.ConvTranspose2d(2, 2, 64, stride=32, bias=False)
        self._initialize_weights()
        # OHWI memory order for every k > 1 conv weight (== channels_last): the kernels read the masters in place
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)
        object.__setattr__(self, "_engine", _Engine(self))
        object.__setattr__(self, "_last_ctx", None)
        object.__setattr__(self, "_last_pred", None)

    def _initialize_weights(self):
        # only the transposed convolutions are (re)initialised -- bilinear kernels (models.py:102-112)
        for mod in self.modules():
            if isinstance(mod, nn.ConvTranspose2d):
                assert mod.kernel_size[0] == mod.kernel_size[1]
                mod.weight.data.copy_(get_upsampling_weight(mod.in_channels, mod.out_channels, mod.kernel_size[0]))

    # ---- precision / synthetic init ----------------------------------------------------------------
    def set_precision(self, dtype):
        """compute dtype of the HIP path: torch.float32 (parity), torch.bfloat16 (throughput) or torch.float16 (IEEE half
        activations / weight images, BASELINE configs[4]; training needs TrainStep's loss scaling)"""
        self._engine.set_precision(dtype)
        return self

    def set_head_precision(self, kind):
        """'native' (the compute dtype); 'fp8': projection head forward on the fp8 matrix cores (e4m3 operands, BASELINE
        configs[4]), backward on the 16-bit head (straight-through); 'fp8_bwd': forward as 'fp8' AND the layer's dgrad / wgrad
        on the fp8 matrix cores (gradient in e5m2, weights / features in e4m3)"""
        if kind not in ("native", "fp8", "fp8_bwd"):
            raise L.SznError("head precision must be 'native', 'fp8' or 'fp8_bwd'")
        self._engine.head_fp8 = kind in ("fp8", "fp8_bwd")
        self._engine.head_fp8_bwd = kind == "fp8_bwd"
        return self

    def load_synthetic(self, seed=1337, device=None):
        """deterministic He-uniform weights (no VGG16 download available).
        device=None: the exact counter-based values of synth.make_params (what the oracle / golden use);
        device given: moves the model there and draws the same distribution with a seeded torch generator on the
        GPU (fast path for benchmarks; identical on every rank for a given seed)."""
        if device is None:
            sd = self.state_dict()
            for k, v in synth.make_params(self.n_class, seed).items():
                sd[k].copy_(torch.from_numpy(v))
        else:
            self.to(device)
            g = torch.Generator(device=device)
            g.manual_seed(seed)
            with torch.no_grad():
                for name, mod in self.named_modules():
                    if isinstance(mod, nn.Conv2d):
                        fan_in = mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]
                        b = math.sqrt(6.0 / fan_in)
                        mod.weight.copy_((torch.rand(mod.weight.shape, device=device, generator=g) * 2 - 1) * b)
                        mod.bias.copy_((torch.rand(mod.bias.shape, device=device, generator=g) * 2 - 1) * 0.1)
        self._engine.mark_dirty()
        return self

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x, mode='fcn', dropout_masks=None):
        if mode not in ('fcn', 'seenmask', 'both'):
            raise Exception('model given unexpected forward mode')
        params = [p for _, p in self.named_parameters()]
        coarse = _Backbone.apply(self, x, self.training, dropout_masks, *params)
        # only the requested head is evaluated (the reference computes both and discards one, models.py:145-160)
        f = _Upscore.apply(self, coarse) if mode in ('fcn', 'both') else None
        s = _SeenmaskUpscore.apply(self, coarse, self.seenmask_upscore.weight) if mode in ('seenmask', 'both') else None
        if mode == 'fcn':
            return f
        if mode == 'seenmask':
            return s
        return f, s

    def embed_predict(self, x, embeddings, target=None):
        """forward pass + nearest-class-embedding prediction (+ cosine loss when `target` is given) WITHOUT materialising the
        (B,E,H,W) score: the fused-from-coarse head (szn_fused_head) evaluates upscore + crop (models.py:146-147), cosine_loss
        (utils.py:75-102) and infer_lbl (utils.py:159-185) per 32x32 cell of the 1/32 map.  -> (loss 0-dim tensor or None,
        pred (B,H,W) int64 device tensor).  Same numbers as `forward` + utils up to rounding order (class assignment differs only
        on pixels whose top-2 cosine margin is < 1e-5).  Used by Trainer.validate."""
        eng = self._engine
        with torch.no_grad():
            ctx = eng.forward(x.detach() if isinstance(x, torch.Tensor) else x, train=False, keep=False)
            emb = torch.as_tensor(embeddings).to(ctx.coarse.device, torch.float32).contiguous()
            K, E = emb.shape
            if E != self.n_class:
                raise L.SznError("embedding dimension %d != model n_class %d" % (E, self.n_class))
            B, H, W = ctx.B, ctx.H, ctx.W
            dev = ctx.coarse.device
            pred = torch.empty(B, H, W, dtype=torch.int64, device=dev)
            ws = torch.empty(L.load().szn_fused_head_workspace_bytes(B, ctx.h, ctx.w, E, K), dtype=torch.uint8, device=dev)
            loss = stats = tgt = None
            if target is not None:
                tgt = target.to(device=dev, dtype=torch.int64).contiguous()
                loss, stats = torch.empty(1, device=dev), torch.empty(B, 2, device=dev)
            L.call("szn_fused_head", B, ctx.h, ctx.w, E, self.head_width, 0, H, W, CROP, K, L.ptr(ctx.coarse), L.ptr(emb),
                   L.ptr(tgt), L.ptr(loss), L.ptr(stats), L.ptr(pred), L.SZN_F32, None, L.ptr(ws), L.stream_ptr())
        return (loss.reshape(()) if loss is not None else None), pred

    def seenmask_predict(self, x, target, n_class, unseen):
        """forward pass + seen-mask loss and prediction WITHOUT the (B,2,H,W) score: szn_seenmask_head evaluates the learned
        stride-32 deconv + crop (models.py:150-151), the binary target `label is a seen class` (trainer_seenmask.py:53-56), the
        2-class cross entropy (size_average) and the channel argmax per 32x32 pixel cell of the 1/32 map.  target: (B,H,W)
        int64 class labels.  -> (loss 0-dim tensor, pred (B,H,W) int64 device tensor).
        Same numbers as forward(mode='seenmask') + utils.cross_entropy2d / channel_argmax (bit-identical score arithmetic).
        Used by trainer_seenmask.Trainer.validate."""
        if n_class > L.MAX_CLASSES:
            raise L.SznError("seenmask_predict: at most %d classes (szn_class_set), got %d" % (L.MAX_CLASSES, n_class))
        eng = self._engine
        with torch.no_grad():
            ctx = eng.forward(x.detach(), train=False, keep=False)
            dev = ctx.coarse.device
            B, H, W = ctx.B, ctx.H, ctx.W
            tgt = target.to(device=dev, dtype=torch.int64).contiguous()
            seen = L.class_set(k for k in range(n_class) if k not in set(unseen))
            pred = torch.empty(B, H, W, dtype=torch.int64, device=dev)
            loss = torch.empty(1, device=dev)
            ws = torch.empty(L.load().szn_seenmask_head_workspace_bytes(B, ctx.h, ctx.w, H, W, CROP), dtype=torch.uint8, device=dev)
            L.call("szn_seenmask_head_k", B, ctx.h, ctx.w, self.head_width, self.n_class, H, W, CROP, L.ptr(ctx.coarse),
                   L.ptr(eng._images["up.w"]), L.ptr(tgt), n_class, seen, L.ptr(loss), None, None, L.ptr(pred), None, None,
                   L.ptr(ws), L.stream_ptr())
        return loss.reshape(()), pred

    def copy_params_from_vgg16(self, vgg16):
        """reference models.py:162-193: zip vgg16.features with our conv list; fc6/fc7 from classifier[0], [3]"""
        features = []
        for item in _BACKBONE:
            if item == "P":
                features.append(None)
            else:
                features += [getattr(self, item[0]), None]
        for l1, l2 in zip(vgg16.features, features):
            if isinstance(l1, nn.Conv2d) and isinstance(l2, nn.Conv2d):
                assert l1.weight.size() == l2.weight.size()
                assert l1.bias.size() == l2.bias.size()
                l2.weight.data.copy_(l1.weight.data)
                l2.bias.data.copy_(l1.bias.data)
        for i, name in zip([0, 3], ['fc6', 'fc7']):
            l1 = vgg16.classifier[i]
            l2 = getattr(self, name)
            l2.weight.data.copy_(l1.weight.data.view(l2.weight.size()))
            l2.bias.data.copy_(l1.bias.data.view(l2.bias.size()))
        self._engine.mark_dirty()


# ---- FCN8s: the skip head BASELINE's north_star names -----------------------------------------------------------
# NOT in /root/reference (models.py:27 is FCN32s only): this follows the public pytorch-fcn FCN8s head on top of the same
# backbone -- score_pool3 Conv2d(256,E,1), score_pool4 Conv2d(512,E,1), upscore2 / upscore_pool4 ConvTranspose2d(E,E,4,stride 2),
# upscore8 ConvTranspose2d(E,E,16,stride 8), all transposed convolutions bias-free with the fixed bilinear kernel
# (models.py:11-24,109-112 initialise them that way and train.py:324-327 never updates them), crops 5 / 9 / 31.  PARITY UNPINNED:
# the checker the tests use is a torch-CPU restatement of that public definition (FCN8sTorch), not reference output.
CROP_POOL4, CROP_POOL3, CROP_UP8 = 5, 9, 31


class _Backbone8(torch.autograd.Function):
    """_Backbone that also exposes the pool3 / pool4 outputs (NHWC, compute dtype) and takes their gradients back"""

    @staticmethod
    def forward(ctx, model, x, train, masks, *params):
        c = model._engine.forward(x, train=train, masks=masks)
        ctx.model, ctx.c = model, c
        model._last_ctx = c
        ctx.set_materialize_grads(False)
        return c.coarse, c.pools[2][1], c.pools[3][1]

    @staticmethod
    def backward(ctx, dcoarse, dpool3, dpool4):
        if dcoarse is None:
            dcoarse = torch.zeros_like(ctx.c.coarse)
        return _backbone_backward(ctx, dcoarse, skips={2: dpool3, 3: dpool4})


class _SkipScore(torch.autograd.Function):
    """score_pool3 / score_pool4: 1x1 conv of a pooled NHWC map into the (padded) projection width, fp32 output"""

    @staticmethod
    def forward(ctx, model, x, weight, bias):
        eng = model._engine
        E, Ci = weight.shape[0], weight.shape[1]
        CP = model.head_width
        wimg = torch.zeros(CP, 1, 1, Ci, device=x.device, dtype=eng.dtype)
        wimg[:E] = weight.detach().reshape(E, 1, 1, Ci).to(eng.dtype)
        bimg = torch.zeros(CP, device=x.device, dtype=torch.float32)
        bimg[:E] = bias.detach().float()
        ctx.model, ctx.x, ctx.wimg, ctx.E = model, x, wimg, E
        return eng._conv(x, None, 0, relu=False, out_f32=True, w=wimg, b=bimg)

    @staticmethod
    def backward(ctx, dout):
        eng, x, wimg, E = ctx.model._engine, ctx.x, ctx.wimg, ctx.E
        CP, Ci = wimg.shape[0], wimg.shape[3]
        dc = dout.contiguous().to(eng.dtype)
        dw = torch.empty(CP, 1, 1, Ci, device=x.device)
        db = torch.empty(CP, device=x.device)
        eng._wgrad(x, dc, dw, db, Ci, CP, 1, 0)
        eng._join_wgrad()
        wT = torch.empty(Ci, 1, 1, CP, device=x.device, dtype=eng.dtype)
        L.call("szn_pack_weight_dgrad", L.dtype_code(eng.dtype), CP, 1, 1, Ci, L.ptr(wimg), L.ptr(wT), L.stream_ptr())
        dx = eng._dgrad(dc, None, x.shape, 0, wT=wT)
        return None, dx, dw[:E].permute(0, 3, 1, 2), db[:E]


class _Up2(torch.autograd.Function):
    """upscore2 / upscore_pool4 (fixed bilinear x2) between NHWC fp32 maps"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, h, w, ld = x.shape
        ctx.shape = (B, h, w, ld)
        out = torch.empty(B, 2 * h + 2, 2 * w + 2, ld, device=x.device, dtype=torch.float32)
        L.call("szn_bilinear_up2_nhwc_fwd", B, h, w, ld, ld, L.ptr(x), L.ptr(out), L.stream_ptr())
        return out

    @staticmethod
    def backward(ctx, dout):
        B, h, w, ld = ctx.shape
        dout = dout.contiguous().float()
        din = torch.empty(B, h, w, ld, device=dout.device, dtype=torch.float32)
        L.call("szn_bilinear_up2_nhwc_bwd", B, h, w, ld, ld, L.ptr(dout), L.ptr(din), L.stream_ptr())
        return din


class _Up8Crop(torch.autograd.Function):
    """upscore8 + crop 31: NHWC fp32 (B,h,w,ld) at 1/8 -> the (B,E,H,W) NCHW score"""

    @staticmethod
    def forward(ctx, x, E, H, W):
        x = x.contiguous()
        B, h, w, ld = x.shape
        ctx.args = (B, h, w, E, ld, H, W)
        f = torch.empty(B, E, H, W, device=x.device, dtype=torch.float32)
        L.call("szn_bilinear_up_crop_fwd", 8, B, h, w, E, ld, 0, H, W, CROP_UP8, L.ptr(x), L.ptr(f), L.stream_ptr())
        return f

    @staticmethod
    def backward(ctx, df):
        B, h, w, E, ld, H, W = ctx.args
        df = df.contiguous().float()
        dx = torch.zeros(B, h, w, ld, device=df.device, dtype=torch.float32)
        L.call("szn_bilinear_up_crop_bwd", 8, B, h, w, E, ld, 0, H, W, CROP_UP8, L.ptr(df), L.ptr(dx), L.stream_ptr())
        return dx, None, None, None


class _FusedHead8(torch.autograd.Function):
    """upscore8 + crop + cosine loss + nearest-embedding prediction straight from the 1/8 fused map (szn_fused_head_strided,
    8x8 cells): the (B,E,H,W) score and its gradient never exist in HBM.  forward -> loss (0-dim); the prediction is left in
    model._last_pred; backward hands back d loss / d map, which the kernel produced in the same pass."""

    @staticmethod
    def forward(ctx, model, x, emb, target, H, W, want_grad):
        x = x.contiguous()
        B, h, w, ld = x.shape
        K, E = emb.shape
        dev = x.device
        pred = torch.empty(B, H, W, dtype=torch.int64, device=dev)
        ws = torch.empty(L.load().szn_fused_head_workspace_bytes(B, h, w, E, K), dtype=torch.uint8, device=dev)
        loss = stats = dx = None
        if target is not None:
            loss, stats = torch.empty(1, device=dev), torch.empty(B, 2, device=dev)
            if want_grad:
                dx = torch.zeros(B, h, w, ld, device=dev, dtype=torch.float32)
        L.call("szn_fused_head_strided", 8, B, h, w, E, ld, 0, H, W, CROP_UP8, K, L.ptr(x), L.ptr(emb), L.ptr(target), L.ptr(loss),
               L.ptr(stats), L.ptr(pred), L.SZN_F32, L.ptr(dx), L.ptr(ws), L.stream_ptr())
        ctx.dx = dx
        model._last_pred = pred
        return loss.reshape(()) if loss is not None else x.new_zeros(())

    @staticmethod
    def backward(ctx, g):
        return None, ctx.dx * g, None, None, None, None, None


class FCN8s(FCN32s):
    """FCN8s skip architecture on the FCN32s trunk (see the note above: public pytorch-fcn definition, parity unpinned).
    state_dict keys follow pytorch-fcn's FCN8s: the trunk + score_fr, score_pool3, score_pool4, upscore2, upscore8,
    upscore_pool4 (+ this repo's seenmask_score / seenmask_upscore, which keep the reference's x32 seen-mask head,
    models.py:97-98,149-151).  forward() / embed_loss() run through the autograd bridge; engine.TrainStep (what the
    trainer and bench.py use for the cosine-loss configurations) runs the same chain by hand on its flat buffers."""

    pretrained_model = 'data/fcn8s_from_caffe.pth'

    def __init__(self, n_class=21):
        super(FCN8s, self).__init__(n_class)
        del self.upscore
        self.score_pool3 = nn.Conv2d(256, n_class, 1)
        self.score_pool4 = nn.Conv2d(512, n_class, 1)
        self.upscore2 = nn.ConvTranspose2d(n_class, n_class, 4, stride=2, bias=False)
        self.upscore8 = nn.ConvTranspose2d(n_class, n_class, 16, stride=8, bias=False)
        self.upscore_pool4 = nn.ConvTranspose2d(n_class, n_class, 4, stride=2, bias=False)
        self._initialize_weights()

    def load_synthetic(self, seed=1337, device=None):
        if device is not None:
            return super(FCN8s, self).load_synthetic(seed, device)
        sd = self.state_dict()
        for k, v in synth.make_params(self.n_class, seed).items():
            if k in sd:
                sd[k].copy_(torch.from_numpy(v))
        rs = np.random.RandomState(seed + 8)
        with torch.no_grad():
            for mod in (self.score_pool3, self.score_pool4):
                b = math.sqrt(6.0 / mod.in_channels)
                mod.weight.copy_(torch.from_numpy(rs.uniform(-b, b, size=tuple(mod.weight.shape)).astype(np.float32)))
                mod.bias.copy_(torch.from_numpy(rs.uniform(-0.1, 0.1, size=tuple(mod.bias.shape)).astype(np.float32)))
        self._engine.mark_dirty()
        return self

    def _fuse(self, x, train, masks):
        """-> (coarse (B,h,w,CP) at 1/32, fused map (B,h8,w8,CP) fp32 at 1/8: upscore_pool4(upscore2(score_fr) + score_pool4c)
        + score_pool3c)"""
        params = [p for _, p in self.named_parameters()]
        coarse, pool3, pool4 = _Backbone8.apply(self, x, train, masks, *params)
        up2 = _Up2.apply(coarse)                                                           # upscore2
        sp4 = _SkipScore.apply(self, pool4, self.score_pool4.weight, self.score_pool4.bias)
        n, m_ = up2.shape[1:3]
        fuse4 = up2 + sp4[:, CROP_POOL4:CROP_POOL4 + n, CROP_POOL4:CROP_POOL4 + m_, :]
        up4 = _Up2.apply(fuse4)                                                            # upscore_pool4
        sp3 = _SkipScore.apply(self, pool3, self.score_pool3.weight, self.score_pool3.bias)
        n, m_ = up4.shape[1:3]
        return coarse, up4 + sp3[:, CROP_POOL3:CROP_POOL3 + n, CROP_POOL3:CROP_POOL3 + m_, :]

    def _run(self, x, mode, train, masks):
        f = s = None
        if mode in ('fcn', 'both'):
            coarse, fuse3 = self._fuse(x, train, masks)
            f = _Up8Crop.apply(fuse3, self.n_class, x.shape[2], x.shape[3])                # upscore8 + crop
        else:
            params = [p for _, p in self.named_parameters()]
            coarse, _, _ = _Backbone8.apply(self, x, train, masks, *params)
        if mode in ('seenmask', 'both'):
            s = _SeenmaskUpscore.apply(self, coarse, self.seenmask_upscore.weight)
        return f, s

    def forward(self, x, mode='fcn', dropout_masks=None):
        if mode not in ('fcn', 'seenmask', 'both'):
            raise Exception('model given unexpected forward mode')
        f, s = self._run(x, mode, self.training, dropout_masks)
        return f if mode == 'fcn' else (s if mode == 'seenmask' else (f, s))

    def _emb(self, embeddings, dev):
        emb = torch.as_tensor(embeddings).to(dev, torch.float32).contiguous()
        if emb.shape[1] != self.n_class:
            raise L.SznError("embedding dimension %d != model n_class %d" % (emb.shape[1], self.n_class))
        if emb.shape[0] > L.MAX_CLASSES:
            raise L.SznError("the fused head holds at most %d classes, got %d: use forward() + utils" % (L.MAX_CLASSES, emb.shape[0]))
        return emb

    def embed_loss(self, x, embeddings, target, dropout_masks=None):
        """training-time fused head: -> (cosine loss with autograd history, pred (B,H,W) int64).  Same numbers as
        utils.cosine_loss(self(x), target, embeddings) / utils.infer_lbl_device up to rounding order, without the
        (B,E,H,W) score or its gradient in HBM."""
        emb = self._emb(embeddings, x.device)
        _, fuse3 = self._fuse(x, self.training, dropout_masks)
        tgt = target.to(device=x.device, dtype=torch.int64).contiguous()
        loss = _FusedHead8.apply(self, fuse3, emb, tgt, x.shape[2], x.shape[3], torch.is_grad_enabled())
        return loss, self._last_pred

    def embed_predict(self, x, embeddings, target=None):
        """inference-time fused head -> (loss 0-dim tensor or None, pred (B,H,W) int64 device tensor); see FCN32s.embed_predict"""
        with torch.no_grad():
            emb = self._emb(embeddings, x.device)
            _, fuse3 = self._fuse(x.detach(), False, None)
            tgt = None if target is None else target.to(device=x.device, dtype=torch.int64).contiguous()
            loss = _FusedHead8.apply(self, fuse3, emb, tgt, x.shape[2], x.shape[3], False)
            return (loss if target is not None else None), self._last_pred


def VGG16(pretrained=False, data_dir='data'):
    """reference models.py:195-210 builds torchvision's VGG16 and downloads caffe weights (no network here).
    Accepts a local `<data_dir>/models/vgg16_from_caffe.pth` state dict; otherwise raises."""
    class _VGG(nn.Module):
        def __init__(self):
            super(_VGG, self).__init__()
            layers, cin = [], 3
            for v in [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']:
                if v == 'M':
                    layers.append(nn.MaxPool2d(2, 2))
                else:
                    layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                    cin = v
            self.features = nn.Sequential(*layers)
            self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))
    model = _VGG()
    if not pretrained:
        return model
    model_path = osp.join(data_dir, 'models/vgg16_from_caffe.pth')
    if not osp.exists(model_path):
        raise IOError("pretrained VGG16 weights not found at %s (no network access to download them); "
                      "use FCN32s.load_synthetic() or place the file there" % model_path)
    model.load_state_dict(torch.load(model_path))
    return model
