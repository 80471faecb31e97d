"""utils.py -- losses, nearest-class-embedding inference and segmentation metrics of the SZN path.

Mirrors the function surface of /root/reference/utils.py (same names, argument meaning and return types);
the arithmetic runs in hand-written HIP kernels (csrc/szn_head.hip) through the C-ABI of include/szn.h.
There is no CPU fallback: tensors must live on the GPU.

  load_obj / save_obj                      utils.py:11-17
  cross_entropy2d(score, target, ...)      utils.py:19-48
  mse_loss / cosine_loss                   utils.py:50-102
  label_accuracy_score                     utils.py:104-154
  infer_lbl / infer_lbl_forced_unseen / infer_lbl_szn / stich_seen_unseen_with_mask   utils.py:159-205

Batched semantics (the reference raises for n > 1 in cosine_loss / infer_lbl): per-image loss, mean over
images; inference is applied per image and returns (B,H,W).
"""
import ctypes as C
import pickle

import numpy as np
import torch

from . import _lib as L


def load_obj(name):
    with open(name + '.pkl', 'rb') as f:
        return pickle.load(f, encoding='latin-1')


def save_obj(obj, name):
    with open(name + '.pkl', 'wb') as f:
        pickle.dump(obj, f, pickle.HIGHEST_PROTOCOL)


MEAN_BGR = (104.00698793, 116.66876762, 122.67891434)          # context_dataset.py:51, pascal_dataset.py:41


def image_to_device(img_u8, device=None, mean_bgr=MEAN_BGR):
    """dataset transform on the GPU (context_dataset.py:143-150): uint8 RGB image(s) (H,W,3) or (B,H,W,3) -- host or device --
    -> (B,3,H,W) f32 BGR minus mean_bgr.  The host sends 3 bytes per pixel instead of 12; results are bit-identical to the
    reference's float64 subtraction + .float()."""
    t = img_u8 if isinstance(img_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_u8))
    if t.dtype != torch.uint8 or t.shape[-1] != 3 or t.dim() not in (3, 4):
        raise L.SznError("image_to_device expects uint8 (H,W,3) or (B,H,W,3), got %s %s" % (t.dtype, tuple(t.shape)))
    if t.dim() == 3:
        t = t.unsqueeze(0)
    dev = torch.device(device) if device is not None else (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    t = t.to(dev, non_blocking=True).contiguous()
    B, H, W, _ = t.shape
    out = torch.empty(B, 3, H, W, device=dev, dtype=torch.float32)
    mean = (C.c_double * 3)(*[float(m) for m in mean_bgr])
    L.call("szn_image_u8_to_bgr_f32", B, H, W, L.ptr(t), mean, L.ptr(out), L.stream_ptr())
    return out


def _need_cuda(t, what):
    if not t.is_cuda:
        raise L.SznError("%s must be a GPU tensor (the HIP path has no CPU fallback)" % what)


def _prep(score, target):
    _need_cuda(score, "score")
    score = score.contiguous().float()
    target = target.to(device=score.device, dtype=torch.int64).contiguous()
    return score, target


class _EmbedLoss(torch.autograd.Function):
    """cosine (mode 'cos') / mse (mode 'mse') loss against a per-pixel target embedding"""

    @staticmethod
    def forward(ctx, score, target, table, dense, mode):
        B, E, H, W = score.shape
        dev = score.device
        loss = torch.empty(1, device=dev)
        stats = torch.empty(B, 2, device=dev)
        ws = torch.empty(L.load().szn_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
        K = 0 if table is None else table.shape[0]
        fn = "szn_cosine_loss_fwd" if mode == "cos" else "szn_mse_loss_fwd"
        L.call(fn, B, E, H, W, K, L.ptr(score), L.ptr(target), L.ptr(table), L.ptr(dense), L.ptr(loss), L.ptr(stats),
               L.ptr(ws), L.stream_ptr())
        ctx.save_for_backward(score, target, stats)
        ctx.table, ctx.dense, ctx.mode = table, dense, mode
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        score, target, stats = ctx.saved_tensors
        B, E, H, W = score.shape
        dscore = torch.empty_like(score)
        g = gout.contiguous().float().reshape(1)
        table, dense = ctx.table, ctx.dense
        K = 0 if table is None else table.shape[0]
        fn = "szn_cosine_loss_bwd" if ctx.mode == "cos" else "szn_mse_loss_bwd"
        L.call(fn, B, E, H, W, K, L.ptr(score), L.ptr(target), L.ptr(table), L.ptr(dense), L.ptr(stats), L.ptr(g),
               L.ptr(dscore), L.stream_ptr())
        return dscore, None, None, None, None


def _embed_loss(score, target, target_embed, mode):
    score, target = _prep(score, target)
    te = target_embed.detach() if isinstance(target_embed, torch.Tensor) else torch.as_tensor(target_embed)
    te = te.to(device=score.device, dtype=torch.float32).contiguous()
    if te.dim() == 2:            # (K,E) class-embedding matrix: gather rows by label on the device
        if te.shape[1] != score.shape[1]:
            raise L.SznError("embedding matrix has E=%d, score has %d channels" % (te.shape[1], score.shape[1]))
        return _EmbedLoss.apply(score, target, te, None, mode)
    if te.shape != score.shape:
        raise L.SznError("target_embed shape %s != score shape %s" % (tuple(te.shape), tuple(score.shape)))
    return _EmbedLoss.apply(score, target, None, te, mode)


def cosine_loss(score, target, target_embed):
    """Negative cosine similarity loss (reference utils.py:75-102).

    score (n,c,h,w); target (n,h,w) with -1 = ignore; target_embed (n,c,h,w) as in the reference, OR the
    (K,c) class-embedding matrix, in which case the per-pixel target vector is gathered by label on the GPU
    (ignore pixels are masked either way)."""
    return _embed_loss(score, target, target_embed, "cos")


def mse_loss(score, target, target_embed):
    """Masked sum of squared differences / number of valid pixels (reference utils.py:50-73)."""
    return _embed_loss(score, target, target_embed, "mse")


class _CE2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, target, size_average, weight):
        B, Cc, H, W = score.shape
        dev = score.device
        loss = torch.empty(1, device=dev)
        stats = torch.empty(B, 2, device=dev)
        ws = torch.empty(L.load().szn_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
        L.call("szn_ce2d_fwd", B, Cc, H, W, L.ptr(score), L.ptr(target), L.ptr(weight), int(size_average), L.ptr(loss),
               L.ptr(stats), None, L.ptr(ws), L.stream_ptr())
        ctx.save_for_backward(score, target, stats)
        ctx.size_average, ctx.weight = size_average, weight
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        score, target, stats = ctx.saved_tensors
        B, Cc, H, W = score.shape
        dscore = torch.empty_like(score)
        g = gout.contiguous().float().reshape(1)
        L.call("szn_ce2d_bwd", B, Cc, H, W, L.ptr(score), L.ptr(target), L.ptr(ctx.weight), int(ctx.size_average), L.ptr(stats),
               L.ptr(g), L.ptr(dscore), L.stream_ptr())
        return dscore, None, None, None


def cross_entropy2d(score, target, weight=None, size_average=False):
    """log-softmax over channels + masked, optionally class-weighted NLL sum, optionally / #valid pixels (reference
    utils.py:19-48; `weight` = the (C,) tensor handed to F.nll_loss at utils.py:46; size_average divides by the number of
    valid pixels, not by the weight sum, utils.py:47-48)."""
    score, target = _prep(score, target)
    if weight is not None:
        weight = torch.as_tensor(weight).detach().to(device=score.device, dtype=torch.float32).contiguous()
        if weight.dim() != 1 or weight.numel() != score.shape[1]:
            raise L.SznError("cross_entropy2d: weight must have one entry per class (%d), got %s"
                             % (score.shape[1], tuple(weight.shape)))
    return _CE2d.apply(score, target, bool(size_average), weight)


def channel_argmax(score):
    """score.data.max(1)[1] (trainer_fcn.py:117, trainer_seenmask.py:67) -> int64 (B,H,W) device tensor"""
    _need_cuda(score, "score")
    score = score.detach().contiguous().float()
    B, Cc, H, W = score.shape
    dev = score.device
    pred = torch.empty(B, H, W, dtype=torch.int64, device=dev)
    dummy_t = torch.full((B, H, W), -1, dtype=torch.int64, device=dev)
    loss = torch.empty(1, device=dev); stats = torch.empty(B, 2, device=dev)
    ws = torch.empty(L.load().szn_loss_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    L.call("szn_ce2d_fwd", B, Cc, H, W, L.ptr(score), L.ptr(dummy_t), None, 0, L.ptr(loss), L.ptr(stats), L.ptr(pred), L.ptr(ws),
           L.stream_ptr())
    return pred


# ---- nearest-neighbouring-embedding inference --------------------------------------------------------------
def _data(t):
    return t.data if isinstance(t, torch.Tensor) else torch.as_tensor(t)


def infer_lbl_device(score, embed_arr, mode=0, unseen=None, seenmask=None, target=None):
    """int64 (B,H,W) DEVICE tensor; see szn_embed_argmax in include/szn.h"""
    score = _data(score)
    _need_cuda(score, "score")
    score = score.contiguous().float()
    emb = _data(embed_arr).to(device=score.device, dtype=torch.float32).contiguous()
    B, E, H, W = score.shape
    if emb.dim() != 2 or emb.shape[1] != E:
        raise L.SznError("embed_arr must be (K,%d), got %s" % (E, tuple(emb.shape)))
    pred = torch.empty(B, H, W, dtype=torch.int64, device=score.device)
    sm = None if seenmask is None else _data(seenmask).to(score.device).contiguous().float()
    tg = None if target is None else _data(target).to(device=score.device, dtype=torch.int64).contiguous()
    L.call("szn_embed_argmax_k", B, E, H, W, emb.shape[0], L.ptr(score), L.ptr(emb), mode,
           L.class_set(unseen), L.ptr(sm), L.ptr(tg), L.ptr(pred), L.stream_ptr())
    return pred


def infer_lbl(score, embed_arr, cuda=False):
    """argmax_k cos(score_px, embed_k), zero-norm rows score 0 and still compete (reference utils.py:159-185).
    Returns numpy int64 (n,h,w)."""
    return infer_lbl_device(score, embed_arr).cpu().numpy()


def _split_unseen(seen_embed_arr, unseen_embed_arr):
    """recover (full matrix, unseen class list) from the zero-masked pair built at trainer_fcn.py:56-64"""
    se, ue = _data(seen_embed_arr).float(), _data(unseen_embed_arr).float()
    unseen = torch.nonzero(ue.abs().sum(1) > 0).flatten().tolist()
    return se + ue, unseen


def stich_seen_unseen_with_mask(score, seen_embed_arr, unseen_embed_arr, unseen_mask, cuda):
    """reference utils.py:201-205 with an explicit boolean mask (numpy or tensor, (n,h,w))"""
    full, unseen = _split_unseen(seen_embed_arr, unseen_embed_arr)
    dev = _data(score).device
    m = torch.as_tensor(np.asarray(unseen_mask)).to(dev)
    # encode the mask as a 2-channel "seenmask score": channel 1 > channel 0 <=> seen
    sm = torch.stack([m.float(), 1.0 - m.float()], 1)
    return infer_lbl_device(score, full, mode=1, unseen=unseen, seenmask=sm).cpu().numpy()


def infer_lbl_forced_unseen(score, target, seen_embed_arr, unseen_embed_arr, unseen, cuda=False):
    """seen pixels choose among seen classes, GT-unseen pixels among unseen classes (reference utils.py:188-192)"""
    full, _ = _split_unseen(seen_embed_arr, unseen_embed_arr)
    return infer_lbl_device(score, full, mode=1, unseen=unseen, target=target).cpu().numpy()


def infer_lbl_szn(score, seen_mask_score, seen_embed_arr, unseen_embed_arr, cuda=False):
    """full SZN inference: the 2-channel seen-mask argmax picks the group (reference utils.py:195-199)"""
    full, unseen = _split_unseen(seen_embed_arr, unseen_embed_arr)
    return infer_lbl_device(score, full, mode=1, unseen=unseen, seenmask=seen_mask_score).cpu().numpy()


# ---- metrics -----------------------------------------------------------------------------------------------
def confusion_hist_device(label_true, label_pred, n_class, unseen=None, hist=None):
    """accumulate the {all, seen, unseen} K x K histograms on the GPU (reference utils.py:104-119,138-152)"""
    lt = _data(label_true)
    _need_cuda(lt, "label_true")
    lt = lt.to(torch.int64).contiguous()
    lp = _data(label_pred).to(device=lt.device, dtype=torch.int64).contiguous()
    if hist is None:
        hist = torch.zeros(3, n_class, n_class, dtype=torch.int64, device=lt.device)
    L.call("szn_confusion_hist_k", lt.numel(), n_class, L.ptr(lt), L.ptr(lp), L.class_set(unseen), L.ptr(hist),
           L.stream_ptr())
    return hist


def _hist_to_metrics(hist):
    """acc, mean class acc, mean IU, fw IU with the reference's nan-mean rules (utils.py:121-129)"""
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(all="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return acc, acc_cls, mean_iu, fwavacc


def _fast_hist(label_true, label_pred, n_class, target='all', unseen=None):
    """host restatement of utils.py:104-119 for numpy label maps (validation logs on the host)"""
    label_true = np.asarray(label_true)
    label_pred = np.asarray(label_pred)
    mask = (label_true >= 0) & (label_true < n_class)
    if target == 'unseen':
        mask = mask & np.isin(label_true, list(unseen))
    elif target == 'seen':
        mask = mask & np.isin(label_true, [x for x in range(n_class) if x not in unseen])
    return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask],
                       minlength=n_class ** 2).reshape(n_class, n_class)


def label_accuracy_score(label_trues, label_preds, n_class, unseen=None):
    """overall acc, mean acc, mean IU, fwavacc -- and the same for seen / unseen GT pixels when `unseen` is given
    (reference utils.py:131-154).  Accepts numpy label maps (host) or GPU tensors (histogram on the device)."""
    first = label_trues[0] if isinstance(label_trues, (list, tuple)) else label_trues
    if isinstance(first, torch.Tensor) and first.is_cuda:
        hist = torch.zeros(3, n_class, n_class, dtype=torch.int64, device=first.device)
        lts = label_trues if isinstance(label_trues, (list, tuple)) else [label_trues]
        lps = label_preds if isinstance(label_preds, (list, tuple)) else [label_preds]
        for lt, lp in zip(lts, lps):
            confusion_hist_device(lt, lp, n_class, unseen, hist)
        h = hist.cpu().numpy()
    else:
        h = np.zeros((3, n_class, n_class))
        for lt, lp in zip(label_trues, label_preds):
            lt, lp = np.asarray(lt).flatten(), np.asarray(lp).flatten()
            h[0] += _fast_hist(lt, lp, n_class)
            if unseen:
                h[1] += _fast_hist(lt, lp, n_class, 'seen', unseen)
                h[2] += _fast_hist(lt, lp, n_class, 'unseen', unseen)
    metrics = _hist_to_metrics(h[0])
    if unseen:
        return metrics, _hist_to_metrics(h[1]), _hist_to_metrics(h[2])
    return metrics
