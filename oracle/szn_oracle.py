"""CPU oracle of the SZN pixel-embedding training path -- TEST INFRASTRUCTURE ONLY.

A numpy + C (oracle/szn_oracle_conv.c, oracle/szn_oracle_head.c) restatement of the reference's algorithm,
in the reference's own tensor layouts (NCHW activations, OIHW weights).  It is the checker that
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use; the product path
(zeroshotsemanticsegmentation_amd/) never imports it.

Pinning: every function here is checked against golden vectors captured from the reference itself
(tests/golden/*.npz, written by tools/capture_golden.py) in tests/test_oracle_golden.py.
A6 (pretrained VGG16 weights via torchvision / fcn) is absent from the reference tree: parity unpinned
for that row only (weights here are the deterministic synthetic ones of synth.make_params).

Reference sites restated:
  FCN32s.forward            /root/reference/models.py:114-160
  get_upsampling_weight     models.py:11-24
  cosine_loss / mse_loss / cross_entropy2d   utils.py:19-102
  infer_lbl*                utils.py:159-205
  label_accuracy_score      utils.py:104-154
  get_parameters + optimizers  train.py:126-133,174-175,302-331 (torch.optim.Adam / SGD update rules)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libszn_oracle.so")
_lib = None

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """gcc-compile the C part into oracle/_build/ (also called by __graft_entry__.build())."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.szo_cosine_loss.restype = C.c_double
        _lib.szo_mse_loss.restype = C.c_double
        _lib.szo_ce2d.restype = C.c_double
        _lib.szo_fused_head.restype = C.c_double
        _lib.szo_fused_head_s.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------------------------------------ conv ops
def conv2d_fwd(x, w, b, pad, relu=False):
    x, w = _c(x), _c(w)
    B, Ci, Hi, Wi = x.shape
    Co, _, K, _ = w.shape
    if K == 1 and pad == 0:      # 1x1: one long row vectorises better; same arithmetic
        out = np.empty((B, Co, 1, Hi * Wi), np.float32)
        lib().szo_conv2d_fwd(_p(x), _p(w), _p(None if b is None else _c(b)), _p(out), B, Ci, 1, Hi * Wi, Co, 1, 0, int(relu))
        return out.reshape(B, Co, Hi, Wi)
    out = np.empty((B, Co, Hi + 2 * pad - K + 1, Wi + 2 * pad - K + 1), np.float32)
    lib().szo_conv2d_fwd(_p(x), _p(w), _p(None if b is None else _c(b)), _p(out), B, Ci, Hi, Wi, Co, K, pad, int(relu))
    return out


def conv2d_dgrad(dout, w, in_shape, pad):
    dout, w = _c(dout), _c(w)
    B, Ci, Hi, Wi = in_shape
    Co, _, K, _ = w.shape
    din = np.empty(in_shape, np.float32)
    if K == 1 and pad == 0:
        lib().szo_conv2d_dgrad(_p(dout), _p(w), _p(din), B, Ci, 1, Hi * Wi, Co, 1, 0)
    else:
        lib().szo_conv2d_dgrad(_p(dout), _p(w), _p(din), B, Ci, Hi, Wi, Co, K, pad)
    return din


def conv2d_wgrad(x, dout, K, pad):
    x, dout = _c(x), _c(dout)
    B, Ci, Hi, Wi = x.shape
    Co = dout.shape[1]
    dw = np.empty((Co, Ci, K, K), np.float32)
    db = np.empty((Co,), np.float32)
    if K == 1 and pad == 0:
        lib().szo_conv2d_wgrad(_p(x), _p(dout), _p(dw), _p(db), B, Ci, 1, Hi * Wi, Co, 1, 0)
    else:
        lib().szo_conv2d_wgrad(_p(x), _p(dout), _p(dw), _p(db), B, Ci, Hi, Wi, Co, K, pad)
    return dw, db


def maxpool_fwd(x):
    x = _c(x)
    B, Cc, Hi, Wi = x.shape
    out = np.empty((B, Cc, (Hi + 1) // 2, (Wi + 1) // 2), np.float32)
    idx = np.empty(out.shape, np.int32)
    lib().szo_maxpool_fwd(_p(x), _p(out), _p(idx), B, Cc, Hi, Wi)
    return out, idx


def maxpool_bwd(dout, idx, in_shape):
    dout = _c(dout)
    din = np.empty(in_shape, np.float32)
    B, Cc, Hi, Wi = in_shape
    lib().szo_maxpool_bwd(_p(dout), _p(idx), _p(din), B, Cc, Hi, Wi)
    return din


def get_upsampling_weight(cin, cout, k):
    """models.py:11-24 (float64 outer product, then float32)."""
    factor = (k + 1) // 2
    center = factor - 1 if k % 2 == 1 else factor - 0.5
    og = np.ogrid[:k, :k]
    filt = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
    w = np.zeros((cin, cout, k, k), dtype=np.float64)
    w[range(cin), range(cout), :, :] = filt
    return w.astype(np.float32)


def deconv_fwd(x, wt, H, W, crop=19, diag=False):
    x, wt = _c(x), _c(wt)
    B, Cin, h, w = x.shape
    Cout = Cin if diag else wt.shape[1]
    out = np.empty((B, Cout, H, W), np.float32)
    lib().szo_deconv64s32_fwd(_p(x), _p(wt), _p(out), B, Cin, Cout, h, w, H, W, crop, int(diag))
    return out


def deconv_dgrad(dout, wt, in_shape, crop=19, diag=False):
    dout, wt = _c(dout), _c(wt)
    B, Cin, h, w = in_shape
    Cout, H, W = dout.shape[1:]
    din = np.empty(in_shape, np.float32)
    lib().szo_deconv64s32_dgrad(_p(dout), _p(wt), _p(din), B, Cin, Cout, h, w, H, W, crop, int(diag))
    return din


def deconv_wgrad(x, dout, crop=19):
    x, dout = _c(x), _c(dout)
    B, Cin, h, w = x.shape
    Cout, H, W = dout.shape[1:]
    dwt = np.empty((Cin, Cout, 64, 64), np.float32)
    lib().szo_deconv64s32_wgrad(_p(x), _p(dout), _p(dwt), B, Cin, Cout, h, w, H, W, crop)
    return dwt


# ------------------------------------------------------------------------------------------------ the network
BACKBONE = [  # (name, pad) conv+relu ... 'P' = pool  -- models.py:116-137
    ("conv1_1", 100), ("conv1_2", 1), "P",
    ("conv2_1", 1), ("conv2_2", 1), "P",
    ("conv3_1", 1), ("conv3_2", 1), ("conv3_3", 1), "P",
    ("conv4_1", 1), ("conv4_2", 1), ("conv4_3", 1), "P",
    ("conv5_1", 1), ("conv5_2", 1), ("conv5_3", 1), "P",
]


class FCN32sOracle(object):
    """models.FCN32s restated on numpy arrays.  params: dict 'layer.weight' / 'layer.bias' in torch layout;
    upscore is the fixed bilinear kernel (models.py:109-112), seenmask_upscore.weight (2,2,64,64) learnable."""

    def __init__(self, params, n_class):
        self.p = {k: _c(v) for k, v in params.items()}
        self.n_class = n_class
        self.up_filt = get_upsampling_weight(1, 1, 64)[0, 0]                       # (64,64)
        if "seenmask_upscore.weight" not in self.p:
            self.p["seenmask_upscore.weight"] = get_upsampling_weight(2, 2, 64)   # models.py:98,109-112
        self.saved = None

    def forward(self, x, mode="fcn", masks=None, keep=False):
        """masks: None (eval) or (mask6, mask7) Dropout2d factors of shape (B,4096) with values {0, 2}."""
        p = self.p
        x = _c(x)
        H, W = x.shape[2:]
        saved = {"x": x}
        h = x
        pool_i = 0
        for item in BACKBONE:
            if item == "P":
                pool_i += 1
                saved["pool%d_in" % pool_i] = h
                h, idx = maxpool_fwd(h)
                saved["pool%d_idx" % pool_i] = idx
                saved["pool%d" % pool_i] = h
            else:
                name, pad = item
                saved[name + "_in"] = h
                h = conv2d_fwd(h, p[name + ".weight"], p[name + ".bias"], pad, relu=True)
        saved["fc6_in"] = h
        h = conv2d_fwd(h, p["fc6.weight"], p["fc6.bias"], 0, relu=True)
        saved["relu6"] = h
        if masks is not None:
            h = h * masks[0][:, :, None, None].astype(np.float32)
        saved["fc7_in"] = h
        h = conv2d_fwd(h, p["fc7.weight"], p["fc7.bias"], 0, relu=True)
        saved["relu7"] = h
        if masks is not None:
            h = h * masks[1][:, :, None, None].astype(np.float32)
        saved["feat"] = h
        saved["masks"] = masks
        coarse_f = conv2d_fwd(h, p["score_fr.weight"], p["score_fr.bias"], 0)
        coarse_s = conv2d_fwd(h, p["seenmask_score.weight"], p["seenmask_score.bias"], 0)
        saved["coarse_f"], saved["coarse_s"] = coarse_f, coarse_s
        f = deconv_fwd(coarse_f, np.broadcast_to(self.up_filt, (self.n_class, 64, 64)), H, W, diag=True)
        s = deconv_fwd(coarse_s, p["seenmask_upscore.weight"], H, W)
        if keep:
            self.saved = saved
        self.last = saved
        if mode == "fcn":
            return f
        if mode == "seenmask":
            return s
        if mode == "both":
            return f, s
        raise Exception("model given unexpected forward mode")

    def backward(self, df=None, ds=None, backbone=True):
        """grads of every parameter given d(loss)/df and/or d(loss)/ds (autograd of forward)."""
        p, sv = self.p, self.saved
        g = {}
        feat = sv["feat"]
        dfeat = np.zeros_like(feat)
        if df is not None:
            dcf = deconv_dgrad(df, np.broadcast_to(self.up_filt, (self.n_class, 64, 64)), sv["coarse_f"].shape, diag=True)
            g["score_fr.weight"], g["score_fr.bias"] = conv2d_wgrad(feat, dcf, 1, 0)
            dfeat += conv2d_dgrad(dcf, p["score_fr.weight"], feat.shape, 0)
        if ds is not None:
            dcs = deconv_dgrad(ds, p["seenmask_upscore.weight"], sv["coarse_s"].shape)
            g["seenmask_upscore.weight"] = deconv_wgrad(sv["coarse_s"], ds)
            g["seenmask_score.weight"], g["seenmask_score.bias"] = conv2d_wgrad(feat, dcs, 1, 0)
            dfeat += conv2d_dgrad(dcs, p["seenmask_score.weight"], feat.shape, 0)
        if not backbone:
            return g
        masks = sv["masks"]
        d = dfeat
        if masks is not None:
            d = d * masks[1][:, :, None, None].astype(np.float32)
        d = np.where(sv["relu7"] > 0, d, 0).astype(np.float32)
        g["fc7.weight"], g["fc7.bias"] = conv2d_wgrad(sv["fc7_in"], d, 1, 0)
        d = conv2d_dgrad(d, p["fc7.weight"], sv["fc7_in"].shape, 0)
        if masks is not None:
            d = d * masks[0][:, :, None, None].astype(np.float32)
        d = np.where(sv["relu6"] > 0, d, 0).astype(np.float32)
        g["fc6.weight"], g["fc6.bias"] = conv2d_wgrad(sv["fc6_in"], d, 7, 0)
        d = conv2d_dgrad(d, p["fc6.weight"], sv["fc6_in"].shape, 0)
        pool_i = 5
        for item in reversed(BACKBONE):
            if item == "P":
                d = maxpool_bwd(d, sv["pool%d_idx" % pool_i], sv["pool%d_in" % pool_i].shape)
                act = sv["pool%d_in" % pool_i]
                pool_i -= 1
            else:
                name, pad = item
                d = np.where(act > 0, d, 0).astype(np.float32)          # ReLU backward on this conv's output
                xin = sv[name + "_in"]
                K = p[name + ".weight"].shape[2]
                g[name + ".weight"], g[name + ".bias"] = conv2d_wgrad(xin, d, K, pad)
                if name != "conv1_1":
                    d = conv2d_dgrad(d, p[name + ".weight"], xin.shape, pad)
                    act = xin
        return g


# ------------------------------------------------------------------------------------------------ losses / inference
def _loss(fn, score, target, embed, target_embed, want_grad):
    score = _c(score)
    target = _c(target, np.int64)
    B, E, H, W = score.shape
    stats = np.zeros((B, 2), np.float32)
    dscore = np.empty_like(score) if want_grad else None
    K = 0 if embed is None else embed.shape[0]
    loss = fn(B, E, H * W, K, _p(score), _p(target), _p(None if embed is None else _c(embed)),
              _p(None if target_embed is None else _c(target_embed)), _p(stats), _p(dscore))
    return np.float32(loss), dscore, stats


def cosine_loss(score, target, embed=None, target_embed=None, want_grad=True):
    """utils.py:75-102 (+ d loss / d score); target embedding gathered from embed[K][E] or given densely."""
    return _loss(lib().szo_cosine_loss, score, target, embed, target_embed, want_grad)


def mse_loss(score, target, embed=None, target_embed=None, want_grad=True):
    """utils.py:50-73"""
    return _loss(lib().szo_mse_loss, score, target, embed, target_embed, want_grad)


def cross_entropy2d(score, target, size_average=False, want_grad=True, weight=None):
    """utils.py:19-48 (weight = the optional (C,) class weights of utils.py:46); also returns the channel argmax
    (trainer_fcn.py:117)."""
    score = _c(score)
    target = _c(target, np.int64)
    B, Cc, H, W = score.shape
    stats = np.zeros((B, 2), np.float32)
    dscore = np.empty_like(score) if want_grad else None
    pred = np.empty((B, H, W), np.int64)
    weight = None if weight is None else _c(weight)
    loss = lib().szo_ce2d(B, Cc, H * W, _p(score), _p(target), _p(weight), int(size_average), _p(stats), _p(dscore), _p(pred))
    return np.float32(loss), dscore, pred


def bits(unseen):
    b = 0
    for k in unseen:
        b |= 1 << int(k)
    return b


def words(unseen, K):
    """class set -> (K + 63) // 64 uint64 words (bit k % 64 of word k // 64), or None for the empty set"""
    b = bits(unseen or [])
    if not b:
        return None
    assert b >> K == 0, "class index >= K"
    n = (K + 63) // 64
    return (C.c_uint64 * n)(*[(b >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)])


def infer_lbl(score, embed):
    """utils.py:159-185 -> int64 (B,H,W)"""
    score, embed = _c(score), _c(embed)
    B, E, H, W = score.shape
    pred = np.empty((B, H, W), np.int64)
    lib().szo_embed_argmax(B, E, H * W, embed.shape[0], _p(score), _p(embed), 0, None, None, None, _p(pred))
    return pred


def infer_lbl_szn(score, seenmask_score, embed, unseen):
    """utils.py:195-205 with seen/unseen matrices = embed with the other group's rows zeroed (trainer_fcn.py:56-64)"""
    score, embed, sm = _c(score), _c(embed), _c(seenmask_score)
    B, E, H, W = score.shape
    pred = np.empty((B, H, W), np.int64)
    lib().szo_embed_argmax(B, E, H * W, embed.shape[0], _p(score), _p(embed), 1, words(unseen, embed.shape[0]), _p(sm), None, _p(pred))
    return pred


def infer_lbl_forced_unseen(score, target, embed, unseen):
    """utils.py:188-192"""
    score, embed, t = _c(score), _c(embed), _c(target, np.int64)
    B, E, H, W = score.shape
    pred = np.empty((B, H, W), np.int64)
    lib().szo_embed_argmax(B, E, H * W, embed.shape[0], _p(score), _p(embed), 1, words(unseen, embed.shape[0]), None, _p(t), _p(pred))
    return pred


def fused_head(coarse, embed, target, H, W, n_class=None, c0=0, crop=19, want_grad=True, want_pred=True, stride=32):
    """The fused-from-coarse head of the training step (csrc/szn_fused_head.hip) restated: coarse (B,h,w,ldc) f32
    NHWC projection map -> (loss, stats (B,2), pred (B,H,W) int64, dcoarse (B,h,w,ldc) f32 with channels [c0,c0+E) filled).
    Same math as deconv_fwd(diag) -> cosine_loss -> infer_lbl (models.py:146-147, utils.py:75-102,159-185), different
    rounding order; pred follows the kernel's arithmetic contract bit for bit (see szn_oracle_head.c)."""
    coarse, embed = _c(coarse), _c(embed)
    B, h, w, ldc = coarse.shape
    K, E = embed.shape
    if n_class is not None:
        assert n_class == E
    t = None if target is None else _c(target, np.int64)
    stats = np.zeros((B, 2), np.float32)
    pred = np.empty((B, H, W), np.int64) if want_pred else None
    dc = np.zeros_like(coarse) if (want_grad and t is not None) else None
    loss = lib().szo_fused_head_s(stride, B, h, w, E, ldc, c0, H, W, crop, K, _p(coarse), _p(embed), _p(t), _p(stats), _p(pred),
                                  _p(dc))
    return np.float32(loss), stats, pred, dc


def e4m3_round(x):
    """round float32 values to the nearest OCP e4m3 (fn) value, ties to even, saturating at +-448 (what gfx950's
    v_cvt_pk_fp8_f32 produces for |x| <= 448): 3 mantissa bits, normal exponents 2^-6 .. 2^8, subnormal step 2^-9"""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)
    step = np.exp2(e - 3.0)
    q = np.minimum(np.rint(a / step) * step, 448.0)
    return (np.sign(x) * q).astype(np.float32)


def proj_fp8(x, w, bias=None):
    """csrc/szn_proj_fp8.hip restated: per-tensor scales amax/448, e4m3 operands, exact products, rescale + bias.
    x (M,K), w (N,K): the values the kernel reads (already rounded to their storage dtype) -> (M,N) float32."""
    x, w = np.asarray(x, np.float32), np.asarray(w, np.float32)
    ax, aw = np.float32(np.abs(x).max()), np.float32(np.abs(w).max())
    ix = np.float32(448.0) / ax if ax > 0 else np.float32(1.0)
    iw = np.float32(448.0) / aw if aw > 0 else np.float32(1.0)
    xq, wq = e4m3_round(x * ix), e4m3_round(w * iw)
    s = (ax / np.float32(448.0) if ax > 0 else np.float32(1.0)) * (aw / np.float32(448.0) if aw > 0 else np.float32(1.0))
    out = (xq.astype(np.float64) @ wq.astype(np.float64).T) * np.float64(s)
    if bias is not None:
        out = out + np.asarray(bias, np.float64)[None, :]
    return out.astype(np.float32)


def e5m2_round(x):
    """round float32 values to the nearest OCP e5m2 value, ties to even, saturating at +-57344 (gfx950's v_cvt_pk_bf8_f32 for
    in-range inputs): 2 mantissa bits, normal exponents 2^-14 .. 2^15, subnormal step 2^-16"""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -14.0)
    step = np.exp2(e - 2.0)
    q = np.minimum(np.rint(a / step) * step, 57344.0)
    return (np.sign(x) * q).astype(np.float32)


def _fp8_scale(t, fmax):
    a = np.float32(np.abs(t).max()) if t.size else np.float32(0)
    inv = np.float32(fmax) / a if a > 0 else np.float32(1.0)
    sc = a / np.float32(fmax) if a > 0 else np.float32(1.0)
    return inv, sc


def proj_fp8_dgrad(g, w):
    """csrc/szn_proj_fp8.hip backward restated: dx = g . w with the gradient g (M,N) in e5m2 and the weights w (N,K) in e4m3,
    per-tensor scales amax / 57344 and amax / 448, exact products, fp32 rescale -> (M,K) float32 (before gate / dropout)"""
    g, w = np.asarray(g, np.float32), np.asarray(w, np.float32)
    ig, sg = _fp8_scale(g, 57344.0)
    iw, sw = _fp8_scale(w, 448.0)
    gq, wq = e5m2_round(g * ig), e4m3_round(w * iw)
    return ((gq.astype(np.float64) @ wq.astype(np.float64)) * np.float64(sg * sw)).astype(np.float32)


def proj_fp8_wgrad(g, x):
    """dw = g^T . x with g (M,N) in e5m2 and the activations x (M,K) in e4m3 -> (N,K) float32"""
    g, x = np.asarray(g, np.float32), np.asarray(x, np.float32)
    ig, sg = _fp8_scale(g, 57344.0)
    ix, sx = _fp8_scale(x, 448.0)
    gq, xq = e5m2_round(g * ig), e4m3_round(x * ix)
    return ((gq.astype(np.float64).T @ xq.astype(np.float64)) * np.float64(sg * sx)).astype(np.float32)


def confusion_hist(label_trues, label_preds, n_class, unseen=None):
    lt = _c(np.asarray(label_trues).reshape(-1), np.int64)
    lp = _c(np.asarray(label_preds).reshape(-1), np.int64)
    hist = np.zeros((3, n_class, n_class), np.int64)
    lib().szo_confusion_hist(C.c_long(lt.size), n_class, _p(lt), _p(lp), words(unseen, n_class), _p(hist))
    return hist


def hist_to_metrics(hist):
    """utils.py:121-129"""
    hist = hist.astype(np.float64)
    with np.errstate(all="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return acc, acc_cls, mean_iu, fwavacc


def label_accuracy_score(label_trues, label_preds, n_class, unseen=None):
    """utils.py:131-154"""
    hist = confusion_hist(np.concatenate([np.asarray(a).reshape(-1) for a in label_trues]),
                          np.concatenate([np.asarray(a).reshape(-1) for a in label_preds]), n_class, unseen)
    m = hist_to_metrics(hist[0])
    if unseen:
        return m, hist_to_metrics(hist[1]), hist_to_metrics(hist[2])
    return m


# ------------------------------------------------------------------------------------------------ optimizers
WEIGHT_GROUP = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2",
                "conv4_3", "conv5_1", "conv5_2", "conv5_3", "fc6", "fc7", "score_fr"]      # train.py:302-331


class Adam(object):
    """torch.optim.Adam update rule (train.py:130-133: group 0 conv weights lr, group 1 conv biases 2*lr)."""

    def __init__(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t, self.m, self.v = lr, b1, b2, eps, 0, {}, {}

    def step(self, params, grads, lr_of):
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        for k, g in grads.items():
            m = self.m.setdefault(k, np.zeros_like(g))
            v = self.v.setdefault(k, np.zeros_like(g))
            m *= self.b1; m += (1 - self.b1) * g
            v *= self.b2; v += (1 - self.b2) * g * g
            denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(self.eps)
            params[k] -= np.float32(lr_of(k) / bc1) * (m / denom)


class SGD(object):
    """torch.optim.SGD with momentum (train.py:126-129: momentum .99, wd 5e-4 on weights, 0 on biases)."""

    def __init__(self, lr, momentum=0.99):
        self.lr, self.mom, self.buf = lr, momentum, {}

    def step(self, params, grads, lr_of, wd_of):
        for k, g in grads.items():
            g = g + np.float32(wd_of(k)) * params[k]
            if k not in self.buf:
                self.buf[k] = g.copy()
            else:
                self.buf[k] = np.float32(self.mom) * self.buf[k] + g
            params[k] -= np.float32(lr_of(k)) * self.buf[k]
