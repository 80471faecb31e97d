"""Every embedding width the reference CLI accepts (`-e {2,5,10,20,21,50,100,200,300}`, /root/reference/train.py:31) through the
real training step: engine.TrainStep in fp32, fused and unfused head, 64x64, against the CPU oracle, with the REFERENCE's own
K x E matrices (datasets/<ds>/embeddings/norm_embed_arr_<E>.pkl re-saved as tests/golden/embeddings_<ds>_<E>.npy by
tools/capture_golden.py G9).  E pads to 64-wide head rows on the device (E = 2 -> 4 real columns of 64, E = 300 -> 302 of 320), so
every width exercises its own padding / tail-fragment handling in the projection GEMM, the head and the optimizer's flat layout."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import szn_oracle as O  # noqa: E402
from zeroshotsemanticsegmentation_amd import engine, models, synth  # noqa: E402
from helpers_parity import adopt_forward  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
WIDTHS = [2, 5, 10, 20, 21, 50, 100, 200, 300]
H = W = 64
KEYS = ["conv1_1.weight", "conv1_2.weight", "conv3_2.weight", "conv5_3.weight", "conv5_3.bias", "fc6.weight", "fc7.weight",
        "fc7.bias", "score_fr.weight", "score_fr.bias"]


_BASE = {}


def params_for(E, seed=1337):
    """synth.make_params(E, seed) without regenerating the 134 M E-independent values per case: every layer draws from its own
    counter stream (synth.make_params: stream = seed * 1000 + 2 * layer index), only score_fr's shape depends on E"""
    if seed not in _BASE:
        _BASE[seed] = synth.make_params(2, seed)
    out = dict(_BASE[seed])
    li = [n for n, _, _, _ in synth.layer_table(E)].index("score_fr")
    b = np.sqrt(6.0 / 4096)
    out["score_fr.weight"] = synth.uniform(seed * 1000 + 2 * li, (E, 4096, 1, 1), -b, b)
    out["score_fr.bias"] = synth.uniform(seed * 1000 + 2 * li + 1, (E,), -0.1, 0.1)
    return out


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def margins(f, emb):
    """top-2 cosine margin per pixel of an (1,E,H,W) score (float64)"""
    s = f[0].reshape(f.shape[1], -1).T.astype(np.float64)
    e = emb.astype(np.float64)
    en = np.linalg.norm(e, axis=1)
    en[en == 0] = 1.0
    sim = (s @ e.T) / (np.linalg.norm(s, axis=1, keepdims=True) * en[None])
    top = np.sort(sim, axis=1)
    return (top[:, -1] - top[:, -2]).reshape(f.shape[2], f.shape[3])


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("E", WIDTHS)
def test_trainstep_every_cli_embedding_width_vs_oracle(E, fused):
    ds, K = ("pascal", 21) if E in (2, 10, 21, 100, 300) else ("context", 33)       # both datasets' matrices get used
    emb = np.load(os.path.join(G, "embeddings_%s_%d.npy" % (ds, E)))
    assert emb.shape == (K, E)
    x = synth.make_images(1, H, W, seed=500 + E)
    t = synth.make_labels(1, H, W, K, seed=600 + E, block=8, ignore_frac=0.05)
    # the oracle: forward, cosine loss, argmax, backward (eval mode: no dropout)
    params = params_for(E)
    om = O.FCN32sOracle(params, E)
    of = om.forward(x, "fcn", keep=True)
    oloss, odf, _ = O.cosine_loss(of, t, embed=emb)
    opred = O.infer_lbl(of, emb)
    m = models.FCN32s(E)
    sd = m.state_dict()
    for k, v in params.items():
        sd[k].copy_(torch.from_numpy(v))
    m._engine.mark_dirty()
    m = m.cuda().eval()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.float32, fused_head=fused)
    ts.keep_ctx = True
    before = m.score_fr.weight.detach().clone()
    loss, pred = ts.step(cu(x), cu(t))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(oloss)) < 1e-5 * max(1.0, abs(float(oloss))), (E, float(loss), float(oloss))
    clear = margins(of, emb)[None] > 1e-5
    assert clear.mean() > 0.9                     # (E = 2: many near-ties on a 2-d circle; still the bulk of the pixels)
    assert np.array_equal(pred.cpu().numpy()[clear], opred[clear])
    # backward: every element of the probed gradients against the oracle's backward pass run on the HIP pass's forward state (ReLU
    # gates, pooling winners: two correct fp32 forwards differ in a few of them, and each flip moves a whole gradient element --
    # tests/helpers_parity.py); the forward state itself was compared by value above (loss, class assignment)
    om2 = O.FCN32sOracle(params, E)
    assert adopt_forward(om2, ts.last_ctx, x, None, E) == 0.0
    assert np.abs(om2.saved["coarse_f"] - om.saved["coarse_f"]).max() < 1e-3 * np.abs(om.saved["coarse_f"]).max()
    f_hip = O.deconv_fwd(om2.saved["coarse_f"], np.broadcast_to(O.get_upsampling_weight(1, 1, 64)[0, 0], (E, 64, 64)), H, W, diag=True)
    _, odf_hip, _ = O.cosine_loss(f_hip, t, embed=emb)
    og = om2.backward(df=odf_hip)
    ts.last_ctx = None
    named = dict(m.named_parameters())
    for k in KEYS:
        g = named[k].grad.detach().cpu().numpy().astype(np.float64)
        r = og[k].astype(np.float64)
        err = np.abs(g - r).max() / (np.abs(r).max() + 1e-30)
        assert err < (1e-3 if k.endswith(".bias") else 1e-4), (E, k, err)       # north star: 1e-3 relative fp32
    # Adam's first step from zero moments moves every score_fr weight with a clear gradient by lr * sign(g)
    g = named["score_fr.weight"].grad
    d = (m.score_fr.weight.detach() - before)
    big = g.abs() > 1e-6
    if bool(big.any()):
        want = -1e-5 * g / (g.abs() + 1e-8)
        assert float((d - want)[big].abs().max()) < 2e-7
