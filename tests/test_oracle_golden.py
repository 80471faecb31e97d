"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference
(tests/golden/*.npz, written by tools/capture_golden.py from /root/reference).  CPU only.

Tolerances: 1e-3 relative fp32 (north_star) -- measured deviations are ~1e-6; integer outputs (argmax,
histograms) must be exact wherever the reference's own top-2 margin exceeds 1e-5.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import szn_oracle as O  # noqa: E402
from zeroshotsemanticsegmentation_amd import synth  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(G, name + ".npz"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def stats(a):
    a = np.asarray(a, np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()])


def probe(a):
    a = a[0]
    return a[:: max(a.shape[0] // 4, 1), ::7, ::7]


def test_g1_upsampling_weight():
    g = gold("g1_upsampling_weight")
    w = O.get_upsampling_weight(2, 2, 64)
    assert np.array_equal(w[0, 0], g["filt64"]) and np.array_equal(w[0, 1], g["offdiag64"])
    assert np.array_equal(O.get_upsampling_weight(3, 3, 4), g["k4"])
    assert np.array_equal(O.get_upsampling_weight(2, 2, 5), g["k5"])
    assert np.array_equal(synth.bilinear_weight(2, 2, 64), w)
    f = synth.bilinear_filter_1d(64)
    assert np.array_equal((f[:, None] * f[None, :]).astype(np.float32), g["filt64"])


def test_g9_embeddings_hashes():
    import hashlib
    want = {"pascal_20": "5c3d5060c9ec4cc4", "pascal_300": "ed66371966f5cb65", "context_20": "fea01c46f656bb92",
            "context_300": "c019e9c535709c4d"}
    for k, h in want.items():
        a = np.load(os.path.join(G, "embeddings_%s.npy" % k))
        assert hashlib.sha256(a.tobytes()).hexdigest()[:16] == h


@pytest.fixture(scope="module")
def model20():
    return O.FCN32sOracle(synth.make_params(20), 20)


@pytest.mark.parametrize("hw", [(1, 1), (32, 32), (33, 47)])
def test_g2_forward_eval(model20, hw):
    g = gold("g2_forward_eval_%dx%d" % hw)
    f, s = model20.forward(g["x"], mode="both")
    sv = model20.last
    for st in ["pool1", "pool2", "pool3", "pool4", "pool5", "relu6", "relu7"]:
        assert list(sv[st].shape) == list(g[st + "_shape"]), st
        assert rel(stats(sv[st]), g[st + "_stats"]) < 1e-4, st
        assert rel(probe(sv[st]), g[st + "_probe"]) < 1e-4, st
    assert rel(sv["coarse_f"], g["score_fr"]) < 1e-4
    assert rel(sv["coarse_s"], g["seenmask_score"]) < 1e-4
    assert f.shape == g["f"].shape and s.shape == g["s"].shape
    assert rel(f, g["f"]) < 1e-4 and rel(s, g["s"]) < 1e-4


def test_g3_forward_train_dropout(model20):
    g = gold("g3_forward_train_32x32")
    f, s = model20.forward(g["x"], mode="both", masks=(g["mask6"], g["mask7"]))
    assert rel(model20.last["coarse_f"], g["score_fr"]) < 1e-4
    assert rel(stats(model20.last["relu7"]), g["relu7_stats"]) < 1e-4
    assert rel(f, g["f"]) < 1e-4 and rel(s, g["s"]) < 1e-4


@pytest.mark.parametrize("name", ["pascal_E20", "context_E20", "context_E300", "pascal_E300"])
def test_g4_embed_losses(name):
    g = gold("g4_embed_losses_" + name)
    for key, fn in (("cos", O.cosine_loss), ("mse", O.mse_loss)):
        loss, ds, _ = fn(g["score"], g["target"], embed=g["embed"])
        assert abs(float(loss) - float(g[key + "_loss"])) < 1e-5 * max(1.0, abs(float(g[key + "_loss"])))
        assert rel(ds, g[key + "_dscore"]) < 1e-4
        # dense target_embed form (the reference's calling convention, trainer_fcn.py:101)
        lbl0 = np.where(g["target"] < 0, 0, g["target"])
        te = np.ascontiguousarray(g["embed"][lbl0[0]].transpose(2, 0, 1)[None])
        loss2, ds2, _ = fn(g["score"], g["target"], target_embed=te)
        assert loss2 == loss and np.array_equal(ds2, ds)


@pytest.mark.parametrize("name", ["C21_n1", "C2_n1", "C2_n3"])
def test_g4_ce2d(name):
    g = gold("g4_ce2d_" + name)
    loss, ds, pred = O.cross_entropy2d(g["score"], g["target"], size_average=bool(g["size_average"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    assert rel(ds, g["dscore"]) < 1e-4
    assert np.array_equal(pred, g["pred"])


@pytest.mark.parametrize("name", ["C21_n1", "C2_n3"])
def test_g4_ce2d_weighted(name):
    """cross_entropy2d(weight=...) (reference utils.py:19,46), fixture captured from the reference"""
    g = gold("g4_ce2d_weighted_" + name)
    loss, ds, _ = O.cross_entropy2d(g["score"], g["target"], size_average=bool(g["size_average"]), weight=g["weight"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    assert rel(ds, g["dscore"]) < 1e-4
    # the weights matter: the unweighted loss is a different number
    l0, _, _ = O.cross_entropy2d(g["score"], g["target"], size_average=bool(g["size_average"]))
    assert abs(float(l0) - float(g["loss"])) > 1e-3 * abs(float(g["loss"]))


@pytest.mark.parametrize("name", ["context_E20", "pascal_E20", "context_E300"])
def test_g5_infer(name):
    g = gold("g5_infer_" + name)
    emb, unseen, seen = g["embed"], list(g["unseen"]), list(g["seen"])

    def masked(rows):
        m = np.zeros_like(emb)
        m[rows] = emb[rows]
        return m

    checks = [
        ("pred_all", O.infer_lbl(g["score"], emb), g["margin_all"]),
        ("pred_seen_only", O.infer_lbl(g["score"], masked(seen)), g["margin_seen_only"]),
        ("pred_unseen_only", O.infer_lbl(g["score"], masked(unseen)), g["margin_unseen_only"]),
        ("pred_szn", O.infer_lbl_szn(g["score"], g["seenmask"], emb, unseen),
         np.minimum(g["margin_seen_only"], g["margin_unseen_only"])),
        ("pred_forced", O.infer_lbl_forced_unseen(g["score"], g["target"], emb, unseen),
         np.minimum(g["margin_seen_only"], g["margin_unseen_only"])),
    ]
    for key, got, margin in checks:
        want = g[key]
        assert got.shape == want.shape and got.dtype == np.int64
        safe = margin[None] > 1e-5
        assert np.array_equal(got[safe], want[safe]), key
        assert (got != want).mean() < 2e-3, key          # only near-ties may differ
    # the zero-row quirk is exercised: some pixels pick a zeroed class in the unseen-only pass
    assert np.isin(g["pred_unseen_only"], seen).any()


def test_g6_metrics():
    g = gold("g6_metrics")
    lt, lp = list(g["lt"]), list(g["lp"])
    assert np.array_equal(O.confusion_hist(np.stack(lt), np.stack(lp), 33)[0], g["hist"])
    np.testing.assert_allclose(O.label_accuracy_score(lt, lp, 33), g["metrics"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(O.label_accuracy_score(lt, lp, 33, unseen=[16, 18])), g["metrics3"], rtol=1e-12,
                               equal_nan=True)
    lt2, lp2 = [g["lt_adv0"], g["lt_adv1"]], [g["lp_adv0"], g["lp_adv1"]]
    np.testing.assert_allclose(O.label_accuracy_score(lt2, lp2, 33), g["metrics_adv"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(O.label_accuracy_score(lt2, lp2, 33, unseen=[7, 9])), g["metrics_adv3"],
                               rtol=1e-12, equal_nan=True)


PROBE_PARAMS = ["conv1_1.weight", "conv1_1.bias", "conv1_2.weight", "conv3_2.weight", "conv5_3.bias", "fc6.weight",
                "fc7.weight", "fc7.bias", "score_fr.weight", "score_fr.bias"]


def probe_idx(n, cnt=64):
    return (np.arange(cnt, dtype=np.int64) * 2654435761 % n).astype(np.int64)


@pytest.mark.parametrize("optname", ["adam", "sgd"])
def test_g7_train_step(optname):
    g = gold("g7_train_step_" + optname)
    params = synth.make_params(20)
    m = O.FCN32sOracle(params, 20)
    before = {k: m.p[k].copy() for k in PROBE_PARAMS}
    lr = float(g["lr"])
    opt = O.Adam(lr) if optname == "adam" else O.SGD(lr)
    lr_of = lambda k: lr * (2 if k.endswith(".bias") else 1)
    wd_of = lambda k: 0.0 if k.endswith(".bias") else 0.0005
    for it in range(2):
        score = m.forward(g["x"], mode="fcn", keep=True)
        loss, dscore, _ = O.cosine_loss(score, g["target"], embed=g["embed"])
        grads = m.backward(df=dscore)
        grads = {k: v for k, v in grads.items() if k.split(".")[0] in O.WEIGHT_GROUP}
        if it == 0:
            assert rel(score, g["score0"]) < 1e-4
            assert abs(float(loss) - float(g["loss0"])) < 1e-5
            pred = O.infer_lbl(score, g["embed"])
            safe = g["margin0"][None] > 1e-5
            assert np.array_equal(pred[safe], g["pred0"][safe])
            assert abs(grads["score_fr.weight"].astype(np.float64).sum() - float(g["score_fr_wgrad_sum"])) < 1e-3 * (
                np.abs(grads["score_fr.weight"]).sum() + 1e-30)
            for k in PROBE_PARAMS:
                assert rel(stats(grads[k]), g["grad_stats/" + k]) < 1e-3, k
                # conv1_1.bias sums 52,900 mixed-sign terms: the reference reduces in fp32 (cancellation
                # noise ~3e-3 of the largest element), the oracle in double
                tol = 1e-2 if k == "conv1_1.bias" else 1e-3
                assert rel(grads[k].reshape(-1)[probe_idx(grads[k].size)], g["grad_probe/" + k]) < tol, k
        else:
            assert abs(float(loss) - float(g["loss1"])) < 1e-5
        if optname == "adam":
            opt.step(m.p, grads, lr_of)
        else:
            opt.step(m.p, grads, lr_of, wd_of)
        key = "delta_probe/" if it == 0 else "delta2_probe/"
        for k in PROBE_PARAMS:
            idx = probe_idx(m.p[k].size)
            d = m.p[k].reshape(-1)[idx].astype(np.float64) - before[k].reshape(-1)[idx].astype(np.float64)
            want = g[key + k]
            # the update is ~lr in size: compare relative to the largest update; float32 storage quantises the
            # difference itself, so allow one ulp of the parameter magnitude
            ulp = np.abs(before[k].reshape(-1)[idx]).max() * 2.0 ** -23
            rtol = 2e-2 if k == "conv1_1.bias" else 2e-3      # see the fp32-reduction note above
            assert np.abs(d - want).max() < rtol * np.abs(want).max() + 2 * ulp, (k, it)


def test_g8_seenmask_step():
    g = gold("g8_seenmask_step")
    m = O.FCN32sOracle(synth.make_params(20), 20)
    seen = [k for k in range(33) if k not in list(g["unseen"])]
    bin_target = np.isin(g["target"], seen).astype(np.int64)          # trainer_seenmask.py:55-56 (-1 -> 0)
    assert np.array_equal(bin_target, g["bin_target"])
    score = m.forward(g["x"], mode="seenmask", keep=True)
    assert rel(score, g["score"]) < 1e-4
    loss, ds, pred = O.cross_entropy2d(score, bin_target, size_average=True)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))     # loss ~ 23.5 with random weights
    assert np.array_equal(pred, g["pred"])
    grads = m.backward(ds=ds, backbone=False)
    assert rel(grads["seenmask_score.weight"], g["dW_score"]) < 1e-3
    assert rel(grads["seenmask_score.bias"], g["db_score"]) < 1e-3
    assert rel(stats(grads["seenmask_upscore.weight"]), g["dW_up_stats"]) < 1e-3
    assert rel(grads["seenmask_upscore.weight"][:, :, ::9, ::9], g["dW_up_probe"]) < 1e-3
    opt = O.Adam(1e-3)
    sub = {k: grads[k] for k in ("seenmask_score.weight", "seenmask_score.bias", "seenmask_upscore.weight")}
    w0 = m.p["seenmask_score.weight"].copy()
    u0 = m.p["seenmask_upscore.weight"].copy()
    opt.step(m.p, sub, lambda k: 1e-3)
    d = (m.p["seenmask_score.weight"].astype(np.float64) - w0).reshape(-1)[:256]
    assert np.abs(d - g["delta_W_score_probe"]).max() < 2e-3 * np.abs(g["delta_W_score_probe"]).max() + 1e-9
    du = (m.p["seenmask_upscore.weight"].astype(np.float64) - u0)[:, :, ::9, ::9]
    assert np.abs(du - g["delta_W_up_probe"]).max() < 2e-3 * np.abs(g["delta_W_up_probe"]).max() + 1e-7


@pytest.mark.parametrize("case", [(2, 3, 4, 20, 33, 70, 101), (1, 2, 2, 300, 59, 40, 37), (1, 1, 1, 20, 21, 1, 1)])
def test_fused_head_restatement_matches_unfused_oracle(case):
    """oracle.fused_head (the algebraic per-cell evaluation the training step's kernel uses) against the golden-pinned
    sequence deconv_fwd(bilinear) -> cosine_loss -> infer_lbl -> deconv_dgrad: same loss / gradient to rounding, same
    class assignment except on pixels whose top-2 cosine margin is below 1e-5"""
    B, h, w, E, K, H, W = case
    CP = (E + 2 + 63) // 64 * 64
    emb = synth.make_embeddings(K, E, seed=5)
    coarse = np.zeros((B, h, w, CP), np.float32)
    coarse[..., :E + 2] = synth.uniform(31 + E, (B, h, w, E + 2), -2, 2)
    target = synth.make_labels(B, H, W, K, seed=32 + K, block=8, ignore_frac=0.1)
    loss, st, pred, dc = O.fused_head(coarse, emb, target, H, W)
    cf = np.ascontiguousarray(coarse[..., :E].transpose(0, 3, 1, 2))
    filt = np.broadcast_to(O.get_upsampling_weight(1, 1, 64)[0, 0], (E, 64, 64))
    f = O.deconv_fwd(cf, filt, H, W, diag=True)
    loss_u, df, st_u = O.cosine_loss(f, target, embed=emb)
    dc_u = O.deconv_dgrad(df, filt, cf.shape, diag=True).transpose(0, 2, 3, 1)
    pred_u = O.infer_lbl(f, emb)
    assert abs(float(loss) - float(loss_u)) < 2e-6 * max(1.0, abs(float(loss_u)))
    assert np.array_equal(st[:, 1], st_u[:, 1])
    assert rel(dc[..., :E], dc_u) < 1e-4
    assert not dc[..., E:].any()
    bad = pred != pred_u
    assert bad.mean() < 2e-3
    if bad.any():
        fs = f.transpose(0, 2, 3, 1).reshape(-1, E).astype(np.float64)
        en = np.linalg.norm(emb.astype(np.float64), axis=1); en[en == 0] = 1
        sim = fs @ emb.astype(np.float64).T / (np.linalg.norm(fs, axis=1, keepdims=True) * en[None])
        top = np.sort(sim, axis=1)
        assert (top[:, -1] - top[:, -2]).reshape(bad.shape)[bad].max() < 1e-5
    # pred-only form
    _, _, p2, _ = O.fused_head(coarse, emb, None, H, W)
    assert np.array_equal(p2, pred)


def test_torch_cpu_restatement_matches_golden_g7():
    """oracle/torch_ref.py (bench.py's torch-CPU baseline) reproduces the reference's captured train step: loss,
    class assignment and gradient probes of g7 (depthwise upscore == the reference's dense diagonal upscore)"""
    import torch
    from oracle import torch_ref as T
    g = gold("g7_train_step_adam")
    m = T.FCN32sTorch(20).load_numpy(synth.make_params(20))
    x, t, e = torch.from_numpy(g["x"]), torch.from_numpy(g["target"]), torch.from_numpy(g["embed"])
    f = m(x, "fcn")
    assert rel(f.detach().numpy(), g["score0"]) < 1e-4
    loss = T.cosine_loss(f, t, e)
    assert abs(float(loss) - float(g["loss0"])) < 1e-5
    pred = T.infer_lbl(f.detach(), e).numpy()
    safe = g["margin0"][None] > 1e-5
    assert np.array_equal(pred[safe], g["pred0"][safe])
    loss.backward()
    named = dict(m.named_parameters())
    for k in ["conv1_2.weight", "conv3_2.weight", "fc6.weight", "fc7.bias", "score_fr.weight", "score_fr.bias"]:
        gr = named[k].grad.flatten().numpy()
        idx = (np.arange(64, dtype=np.int64) * 2654435761 % gr.size).astype(np.int64)
        assert rel(gr[idx], g["grad_probe/" + k]) < 1e-3, k


def test_e4m3_restatement_matches_torch_float8_cast():
    """oracle.e4m3_round (referee of the fp8 projection kernel) against torch's own float8_e4m3fn conversion: normal,
    subnormal, tie and saturation cases"""
    import torch
    rs = np.random.RandomState(0)
    x = (rs.randn(200000) * np.exp(rs.uniform(-12, 6, 200000))).astype(np.float32)
    x = np.clip(x, -448, 448)
    x[:8] = [0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 17.0, 0.4375]       # ties go to even
    t = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(O.e4m3_round(x), t)
    # proj_fp8 is exact when every value is representable after scaling
    xs = np.array([[448.0, -224.0, 0.0, 1.75] * 32], np.float32)
    ws = np.array([[1.0, 0.5, -0.25, 2.0] * 32, [0.0] * 128], np.float32)
    want = (xs.astype(np.float64) @ ws.astype(np.float64).T + np.array([0.5, -1.0])).astype(np.float32)
    assert np.allclose(O.proj_fp8(xs, ws, np.array([0.5, -1.0], np.float32)), want, rtol=1e-6)


def test_strided_fused_head_restatement_equals_torch_deconv_plus_cosine_loss():
    """szo_fused_head_s (stride 8, crop 31: the FCN8s upscore8 geometry; stride 32 is pinned by the goldens above) against a
    direct torch evaluation: depthwise bilinear ConvTranspose2d -> crop -> cosine loss / argmax.  FCN8s itself is not in the
    reference (parity unpinned); this pins the restatement the GPU kernel is checked with to the public definition."""
    import torch
    import torch.nn.functional as F
    from oracle import torch_ref as T
    rs = np.random.RandomState(0)
    for (B, h, w, E, K, H, W) in [(1, 5, 5, 12, 6, 17, 17), (2, 6, 9, 20, 21, 25, 49)]:
        coarse = rs.randn(B, h, w, 64).astype(np.float32)
        emb = rs.randn(K, E).astype(np.float32)
        t = rs.randint(-1, K, (B, H, W)).astype(np.int64)
        loss, stats, pred, dc = O.fused_head(coarse, emb, t, H, W, crop=31, stride=8)
        f1 = T._bilinear_1d(16)
        filt = torch.from_numpy((f1[:, None] * f1[None, :]).astype(np.float32)).expand(E, 1, 16, 16).contiguous()
        x = torch.from_numpy(coarse[..., :E]).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        s = F.conv_transpose2d(x, filt, stride=8, groups=E)[:, :, 31:31 + H, 31:31 + W]
        lt = T.cosine_loss(s, torch.from_numpy(t), torch.from_numpy(emb))
        lt.backward()
        assert abs(float(loss) - float(lt.detach())) < 1e-6
        assert np.abs(dc[..., :E] - x.grad.permute(0, 2, 3, 1).numpy()).max() < 1e-6
        assert (pred == T.infer_lbl(s.detach(), torch.from_numpy(emb)).numpy()).mean() > 0.999
        assert stats[:, 1].sum() == (t >= 0).sum()


def test_fcn8s_checker_shapes_and_fixed_upsampling():
    import torch
    from oracle import torch_ref as T
    m = T.FCN8sTorch(8)
    x = torch.randn(1, 3, 40, 56)
    f, s = m(x, "both")
    assert tuple(f.shape) == (1, 8, 40, 56) and tuple(s.shape) == (1, 2, 40, 56)
    names = {n for n, _ in m.named_parameters()}
    assert {"score_pool3.weight", "score_pool4.bias", "score_fr.weight"} <= names
    assert not any(n.startswith("up") for n in names)            # bilinear kernels are buffers: never trained (train.py:324-327)
