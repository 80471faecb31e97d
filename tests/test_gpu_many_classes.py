"""More than 64 classes (the szn_class_set forms of the masked entry points: szn_embed_argmax_k, szn_confusion_hist_k,
szn_seenmask_head_k; szn_fused_head with K up to SZN_MAX_CLASSES = 256) against the CPU oracle at 128 x 128.

The reference's seen / unseen lists are Python lists of any length (trainer_fcn.py:56-64, utils.py:104-154,188-205); its two datasets
have 21 and 33 / 59 classes, so nothing above 64 classes exists as a golden vector -- the oracle (pinned on the reference's vectors at
K = 21 / 33 / 59) is the checker here, and the K <= 64 results through the class-set entry points are compared bit for bit with the
uint64_t entry points the rest of the suite pins."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import szn_oracle as O  # noqa: E402
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402
from zeroshotsemanticsegmentation_amd import engine, models, synth, utils  # noqa: E402
from helpers_parity import adopt_forward  # noqa: E402

H = W = 128


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / (np.abs(b).max() + 1e-30))


def unseen_for(K):
    """members in every 64-class word that exists, on both sides of each word boundary, first and last class"""
    return sorted(set(k for k in (0, 3, 62, 63, 64, 65, 70, 127, 128, 129, 149, 191, 192, 200, 255) if k < K) | {K - 1})


@pytest.mark.parametrize("K,E", [(150, 300), (150, 20), (65, 20), (128, 21), (129, 50), (256, 20), (59, 300)])
def test_embed_argmax_class_sets_vs_oracle(K, E):
    B = 2
    emb = synth.make_embeddings(K, E, seed=7)
    emb[min(5, K - 1)] = 0.0                                      # a zero row: norm 0 -> 1 (utils.py:175)
    score = synth.uniform(40 + K, (B, E, H, W), -1, 1)
    score[0, :, 3, 4] = emb[K - 1] * 2.5                         # the last class wins somewhere (its chunk's tail)
    sm = synth.uniform(41 + K, (B, 2, H, W), -1, 1)
    target = synth.make_labels(B, H, W, K, seed=42 + K, block=8, ignore_frac=0.1)
    unseen = unseen_for(K)
    s, e = cu(score), cu(emb)
    got0 = utils.infer_lbl_device(s, e)
    assert np.array_equal(got0.cpu().numpy(), O.infer_lbl(score, emb))
    assert int(got0[0, 3, 4]) == K - 1
    got1 = utils.infer_lbl_device(s, e, mode=1, unseen=unseen, seenmask=cu(sm))
    assert np.array_equal(got1.cpu().numpy(), O.infer_lbl_szn(score, sm, emb, unseen))
    got2 = utils.infer_lbl_device(s, e, mode=1, unseen=unseen, target=cu(target))
    assert np.array_equal(got2.cpu().numpy(), O.infer_lbl_forced_unseen(score, target, emb, unseen))
    assert L.last_kernel() == ("embed_argmax_kernel_chunks" if K > 64 else "embed_argmax_kernel")
    if K <= 64:                                                   # the class-set form == the uint64_t form, bit for bit
        old = torch.empty_like(got1)
        smd = cu(sm)
        L.call("szn_embed_argmax", B, E, H, W, K, L.ptr(s), L.ptr(e), 1, synth.unseen_bits(unseen), L.ptr(smd), None, L.ptr(old),
               L.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(old, got1)
    else:                                                         # ... which refuses what it cannot express
        with pytest.raises(L.SznError):
            L.call("szn_embed_argmax", B, E, H, W, K, L.ptr(s), L.ptr(e), 0, 0, None, None, L.ptr(got0), L.stream_ptr())
    # a set naming a class >= K is an argument error (the reference would raise IndexError zeroing that embedding row)
    if K < L.MAX_CLASSES:
        with pytest.raises(L.SznError):
            utils.infer_lbl_device(s, e, mode=1, unseen=[K], seenmask=cu(sm))


@pytest.mark.parametrize("K", [150, 65, 256, 33])
def test_confusion_hist_class_sets_vs_oracle(K):
    lt = synth.make_labels(3, H, W, K, seed=50 + K, block=4, ignore_frac=0.1)
    lp = synth.make_labels(3, H, W, K, seed=51 + K, block=2, ignore_frac=0.0)
    lp[lp < 0] = 0
    lt[0, 0, :7] = K + 3                                          # out of range: dropped (utils.py:108 mask)
    unseen = unseen_for(K)
    for us in (unseen, None):
        hist = utils.confusion_hist_device(cu(lt), cu(lp), K, unseen=us)
        torch.cuda.synchronize()
        want = O.confusion_hist(lt, lp, K, unseen=us)
        assert np.array_equal(hist.cpu().numpy(), want)
        assert L.last_kernel() == ("hist_kernel_global" if K > 64 else "hist_kernel")
    assert int(want[0].sum()) == int(((lt >= 0) & (lt < K)).sum())


@pytest.mark.parametrize("K,E", [(150, 300), (150, 20), (65, 21), (128, 20), (256, 50)])
def test_fused_head_many_classes_vs_oracle(K, E):
    B, h, w = 2, 5, 5
    CP = (E + 2 + 63) // 64 * 64
    emb = synth.make_embeddings(K, E, seed=9)
    coarse = np.zeros((B, h, w, CP), np.float32)
    coarse[..., :E + 2] = synth.uniform(60 + K, (B, h, w, E + 2), -2, 2)
    target = synth.make_labels(B, H, W, K, seed=61 + K, block=8, ignore_frac=0.1)
    target[0, 8:16, 8:16] = K - 1                                 # the last class (the tail of the last 64-class group) is a label
    oloss, ostats, opred, odc = O.fused_head(coarse, emb, target, H, W)
    c, e, t = cu(coarse), cu(emb), cu(target)
    ws = torch.empty(L.load().szn_fused_head_workspace_bytes(B, h, w, E, K), dtype=torch.uint8, device="cuda")
    loss = torch.empty(1, device="cuda"); stats = torch.empty(B, 2, device="cuda")
    pred = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
    dc = torch.zeros(B, h, w, CP, device="cuda")
    L.call("szn_fused_head", B, h, w, E, CP, 0, H, W, 19, K, L.ptr(c), L.ptr(e), L.ptr(t), L.ptr(loss), L.ptr(stats),
           L.ptr(pred), L.SZN_F32, L.ptr(dc), L.ptr(ws), L.stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(pred.cpu().numpy(), opred)              # the kernel's arithmetic contract (szn_oracle_head.c): bit for bit
    assert abs(loss.item() - float(oloss)) < 2e-6 * max(1.0, abs(float(oloss)))
    assert np.array_equal(stats[:, 1].cpu().numpy(), ostats[:, 1])
    assert rel(dc[..., :E], odc[..., :E]) < 1e-4
    assert float(dc[..., E:].abs().max()) == 0.0
    # and against the unfused sequence of the reference (bilinear upsampling -> cosine loss -> infer_lbl): near-ties only
    f = O.deconv_fwd(np.ascontiguousarray(coarse[..., :E].transpose(0, 3, 1, 2)),
                     np.broadcast_to(O.get_upsampling_weight(1, 1, 64)[0, 0], (E, 64, 64)), H, W, diag=True)
    uloss, _, _ = O.cosine_loss(f, target, embed=emb)
    assert abs(loss.item() - float(uloss)) < 1e-5 * max(1.0, abs(float(uloss)))
    upred = O.infer_lbl(f, emb)
    assert (upred != opred).mean() < 2e-3


def test_seenmask_head_150_classes_vs_oracle_sequence():
    B, h, w = 2, 5, 5
    K = 150
    unseen = unseen_for(K)
    seen = [k for k in range(K) if k not in unseen]
    ldc, c0 = 24, 20
    coarse = np.zeros((B, h, w, ldc), np.float32)
    coarse[..., c0:c0 + 2] = synth.uniform(930, (B, h, w, 2), -2, 2)
    wt = (synth.uniform(931, (2, 2, 64, 64), -1, 1) * 0.05).astype(np.float32)
    target = synth.make_labels(B, H, W, K, seed=932, block=8)
    lib = L.load()
    ws = torch.empty(lib.szn_seenmask_head_workspace_bytes(B, h, w, H, W, 19), dtype=torch.uint8, device="cuda")
    loss, st = torch.zeros(1, device="cuda"), torch.zeros(2, device="cuda")
    conf = torch.zeros(4, dtype=torch.int64, device="cuda")
    pred = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
    dsc = torch.full((B * h * w, 2), 7.0, device="cuda")
    dw = torch.full((2, 2, 64, 64), 7.0, device="cuda")
    c, wd, t = cu(coarse), cu(wt), cu(target)
    L.call("szn_seenmask_head_k", B, h, w, ldc, c0, H, W, 19, L.ptr(c), L.ptr(wd), L.ptr(t), K, L.class_set(seen),
           L.ptr(loss), L.ptr(st), L.ptr(conf), L.ptr(pred), L.ptr(dsc), L.ptr(dw), L.ptr(ws), L.stream_ptr())
    torch.cuda.synchronize()
    cs = np.ascontiguousarray(coarse[..., c0:c0 + 2].transpose(0, 3, 1, 2))
    s = O.deconv_fwd(cs, wt, H, W)
    bin_t = np.isin(target, seen).astype(np.int64)                # trainer_seenmask.py:55-56
    assert 0 < bin_t.mean() < 1 and np.isin(target, [k for k in unseen if k >= 64]).any()
    oloss, ods, opred = O.cross_entropy2d(s, bin_t, size_average=True)
    assert np.array_equal(pred.cpu().numpy(), opred)
    assert abs(loss.item() - float(oloss)) < 1e-5 * max(1.0, abs(float(oloss)))
    assert np.array_equal(conf.cpu().numpy(), np.bincount((2 * bin_t + opred).ravel(), minlength=4))
    assert rel(dsc.view(B, h, w, 2).permute(0, 3, 1, 2), O.deconv_dgrad(ods, wt, (B, 2, h, w))) < 1e-4
    assert rel(dw, O.deconv_wgrad(cs, ods)) < 1e-4
    with pytest.raises(L.SznError):                               # the 64-bit form cannot express 150 classes
        L.call("szn_seenmask_head", B, h, w, ldc, c0, H, W, 19, L.ptr(c), L.ptr(wd), L.ptr(t), K, 1,
               L.ptr(loss), L.ptr(st), L.ptr(conf), L.ptr(pred), L.ptr(dsc), L.ptr(dw), L.ptr(ws), L.stream_ptr())


KEYS = ["conv1_1.weight", "conv3_2.weight", "conv5_3.bias", "fc6.weight", "fc7.bias", "score_fr.weight", "score_fr.bias"]


@pytest.mark.parametrize("fused", [True, False])
def test_trainstep_150_classes_vs_oracle(fused):
    """the whole training step (engine.TrainStep, fp32) with a 150-class label set at 128 x 128, fused and unfused head, vs the oracle"""
    E, K = 20, 150
    emb = synth.make_embeddings(K, E, seed=11)
    x = synth.make_images(1, H, W, seed=700)
    t = synth.make_labels(1, H, W, K, seed=701, block=8, ignore_frac=0.05)
    assert t.max() > 128
    params = synth.make_params(E, 1337)
    om = O.FCN32sOracle(params, E)
    of = om.forward(x, "fcn", keep=True)
    oloss, _, _ = O.cosine_loss(of, t, embed=emb)
    opred = O.infer_lbl(of, emb)
    m = models.FCN32s(E)
    sd = m.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))
    m._engine.mark_dirty()
    m = m.cuda().eval()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.float32, fused_head=fused)
    ts.keep_ctx = True
    loss, pred = ts.step(cu(x), cu(t))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(oloss)) < 1e-5 * max(1.0, abs(float(oloss)))
    s = of[0].reshape(E, -1).T.astype(np.float64)
    e64 = emb.astype(np.float64)
    sim = (s @ e64.T) / (np.linalg.norm(s, axis=1, keepdims=True) * np.linalg.norm(e64, axis=1)[None])
    top = np.sort(sim, axis=1)
    clear = ((top[:, -1] - top[:, -2]) > 1e-5).reshape(1, H, W)
    assert clear.mean() > 0.9
    assert np.array_equal(pred.cpu().numpy()[clear], opred[clear])
    # the running confusion matrix of the step (K x K int64 on the device)
    want = O.confusion_hist(t, pred.cpu().numpy(), K)[0]
    assert np.array_equal(ts.hist.cpu().numpy().reshape(-1, K, K)[0], want)
    om2 = O.FCN32sOracle(params, E)
    assert adopt_forward(om2, ts.last_ctx, x, None, E) == 0.0
    f_hip = O.deconv_fwd(om2.saved["coarse_f"], np.broadcast_to(O.get_upsampling_weight(1, 1, 64)[0, 0], (E, 64, 64)), H, W, diag=True)
    _, odf_hip, _ = O.cosine_loss(f_hip, t, embed=emb)
    og = om2.backward(df=odf_hip)
    ts.last_ctx = None
    named = dict(m.named_parameters())
    for k in KEYS:
        g = named[k].grad.detach().cpu().numpy().astype(np.float64)
        r = og[k].astype(np.float64)
        err = np.abs(g - r).max() / (np.abs(r).max() + 1e-30)
        assert err < (1e-3 if k.endswith(".bias") else 1e-4), (k, err)
