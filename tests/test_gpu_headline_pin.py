"""The headline configuration of bench.py, pinned: bf16 operands, B = 8, 512 x 512, E = 300, K = 59 (49 seen),
train mode, fused-from-coarse head, every shortcut of the throughput path on (the constant band removed from the conv1_2 / conv2 / conv3 blocks, cin-chunk-major K order of
conv_igemm_8ph, fc6's Adam in its weight-gradient epilogue).

Reference = the SAME step in fp32 through the HIP path, which tests/test_gpu_parity_full.py pins element by element to the oracle
(the restatement of trainer_fcn.py:149-180 / models.py:114-160 / utils.py:75-102,159-185).  For every optimizer-visible tensor the
relative L2 error and the cosine of the bf16 gradient are bounded (bounds = 1.5 x what was measured on MI355X, written below with
the reason they grow towards the input), the loss agrees to 2e-2 and the class map to >= 0.99.  Two more bf16 runs in child processes -- full maps with the tile-skipping hints (SZN_BAND_CROP=0) and fully dense
(also SZN_CONST_BORDER=0 SZN_WGT_CB=0 SZN_DGRAD_BORDER=0: environment variables are read once per process) -- show that the band removal
and the hints move nothing beyond fp32 re-ordering."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

E, K, H, B = 300, 59, 512, 8


def run_step(precision):
    """one bench-configuration step -> (loss, pred (B,H,W) int64 cpu, {tensor name: gradient as float32 cpu tensor}, kernels seen)"""
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth
    dev = torch.device("cuda", 0)
    emb = synth.make_embeddings(K, E)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=dev)
    m.train()                                       # Dropout2d on: masks come from the engine's counter RNG, equal in every run
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=precision, fused_head=True, keep_grads=True)
    x = torch.from_numpy(synth.make_images(B, H, H, seed=1337)).to(dev)
    t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=1337, classes=list(range(49)))).to(dev)
    kernels = set()
    orig = L.call

    def spy(name, *a):
        orig(name, *a)
        kernels.add(L.last_kernel())
    engine.L.call = models.L.call = spy
    try:
        loss, pred = ts.step(x, t)
    finally:
        engine.L.call = models.L.call = orig
    torch.cuda.synchronize()
    grads = {}
    for n in ts.layers:
        o, cnt = ts.woff[n]
        grads[n + ".weight"] = ts.flat_gw[o:o + cnt].float().cpu()
        bo, bc = ts.boff[n]
        grads[n + ".bias"] = ts.flat_gb[bo:bo + bc].float().cpu()
    return float(loss), pred.cpu(), grads, kernels


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


# Relative L2 error of the bf16 gradient against the fp32 HIP step, per layer (weights and biases behave alike), as MEASURED on MI355X
# in round 5 (the test prints the table), and the bound asserted = ~1.5 x that.  The error is small where the backward pass starts
# (score_fr 3e-3, fc7 1.3e-2, fc6 2e-2) and grows towards the input (conv4_x 5-7e-2, conv3_x 8-12e-2, conv2_x 15-17e-2, conv1_2 0.27,
# conv1_1 0.39).  That growth is not rounding noise adding up -- fp32 accumulation keeps each layer's own arithmetic at 2-3e-3
# (tests/test_gpu_fullsize.py: every layer against torch fp32 on the same operands) -- it is the forward state: bf16 activations
# differ from fp32 ones by ~4e-3, so ~0.3 % of the 10^8 ReLU gates / pooling winners per layer fall on the other side, each flip
# moves whole gradient elements (relative L2 ~ sqrt(flip fraction) per layer), and the flips of all layers behind a tensor add up.
# DESIGN.md section 2 ("Full-size gradients") measures the same mechanism between two CORRECT fp32 passes (forward difference 4e-6 ->
# weight gradients differ by up to 7e-3 at 512 x 512); a forward difference 1000 x larger gives sqrt(1000) ~ 30 x that.  Synthetic
# kaiming-uniform weights make it worse than a trained VGG would (dense, sign-symmetric pre-activations).  What the test pins is
# therefore: (1) the error of every tensor stays inside the measured envelope, (2) the gradient DIRECTION is kept (cosine >= 0.9
# everywhere, >= 0.995 from conv4_1 on), (3) loss and class map agree, (4) the shortcuts of the throughput path are NOT a source of
# error: the same step without the hints agrees to fp32 re-ordering (<= 5e-5), three to four orders below the envelope.
MEASURED = {"score_fr": 3.2e-3, "fc7": 1.35e-2, "fc6": 1.97e-2, "conv5_3": 2.43e-2, "conv5_2": 3.13e-2, "conv5_1": 4.16e-2,
            "conv4_3": 4.88e-2, "conv4_2": 5.94e-2, "conv4_1": 6.76e-2, "conv3_3": 7.87e-2, "conv3_2": 9.41e-2, "conv3_1": 0.1245,
            "conv2_2": 0.1474, "conv2_1": 0.1731, "conv1_2": 0.269, "conv1_1": 0.3875}
BOUND = {k: 1.5 * v for k, v in MEASURED.items()}
COS_MIN = {k: (0.995 if v < 7e-2 else 0.9) for k, v in MEASURED.items()}


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_headline_bf16_step_gradients_track_the_fp32_step(fast_tmp):
    loss32, pred32, g32, k32 = run_step(torch.float32)
    loss16, pred16, g16, k16 = run_step(torch.bfloat16)
    # the kernels of the throughput path really ran (a silent fall-back to the generic kernels would pass the bounds too)
    for k in ("conv_igemm_8ph", "conv3x3_regw", "wgrad_taps_reduce", "conv_wgrad_wide_adam", "maxpool_bwd_code_kernel"):
        assert any(k in name for name in k16), (k, sorted(k16))
    assert abs(loss16 - loss32) < 2e-2 * abs(loss32), (loss16, loss32)
    agree = float((pred16 == pred32).float().mean())
    assert agree >= 0.99, agree
    rows = []
    for name in g32:
        layer, kind = name.rsplit(".", 1)
        rows.append((name, rel_l2(g16[name], g32[name]), BOUND[layer], cosine(g16[name], g32[name]), COS_MIN[layer]))
    print("\n".join("%-22s rel L2 %.3e  (bound %.2e)  cosine %.5f (>= %.3f)" % r for r in rows))
    print("loss fp32 %.6f bf16 %.6f, class-map agreement %.5f" % (loss32, loss16, agree))
    bad = [r for r in rows if not (r[1] <= r[2] and r[3] >= r[4])]
    assert not bad, bad
    # ---- the same bf16 step without the shortcuts, in child processes (the switches are read once per process):
    #      "hints": the full maps with the tile-skipping hints of rounds 3-4 (SZN_BAND_CROP=0), "dense": neither band removal nor hints
    assert any("band_remap" in k for k in k16)                 # the default path removes the band (round 5)
    for tag, env_extra in (("hints", dict(SZN_BAND_CROP="0")),
                           ("dense", dict(SZN_BAND_CROP="0", SZN_CONST_BORDER="0", SZN_WGT_CB="0", SZN_DGRAD_BORDER="0"))):
        out = os.path.join(fast_tmp, tag + ".pt")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), out], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                           timeout=900, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        d = torch.load(out)
        assert d["loss"] == loss16                             # forward: a kept pixel sees the very values it would see in the full map
        assert torch.equal(d["pred"], pred16)
        assert not any("band_remap" in k for k in d["kernels"])
        if tag == "hints":
            assert any("border" in k or "cb" in k for k in d["kernels"])
        else:
            assert not any("border" in k for k in d["kernels"])
        worst_w = max((rel_l2(g16[name], d["grads"][name]), name) for name in g32 if name.endswith(".weight"))
        worst_b = max((rel_l2(g16[name], d["grads"][name]), name) for name in g32 if name.endswith(".bias"))
        print("default vs %s: worst relative L2 weights %.3e (%s), biases %.3e (%s)" % ((tag,) + worst_w + worst_b))
        # rank-one terms / region sums / summed band gradients replace dense fp32 sums in another order (weights: 1e-6 class); the summed band
        # gradient passes through one more bf16 rounding before it reaches the bias sums (1e-4 class) -- orders below the bf16 envelope above
        assert worst_w[0] < 5e-5 and worst_b[0] < 2e-3, (worst_w, worst_b)


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


# Stored gradient tensors (bf16) between the loss and a layer's weight gradient: d(coarse) is the first, every dgrad output one more.
# fc6's and fc7's dgrads, five pools (the pooled gradient is re-stored) and 13 conv dgrads lie on the way to conv1_1.
CHAIN = ["score_fr", "fc7", "fc6", "conv5_3", "conv5_2", "conv5_1", "conv4_3", "conv4_2", "conv4_1", "conv3_3", "conv3_2", "conv3_1",
         "conv2_2", "conv2_1", "conv1_2", "conv1_1"]
POOLS_BEHIND = {"score_fr": 0, "fc7": 0, "fc6": 0, "conv5_3": 1, "conv5_2": 1, "conv5_1": 1, "conv4_3": 2, "conv4_2": 2, "conv4_1": 2,
                "conv3_3": 3, "conv3_2": 3, "conv3_1": 3, "conv2_2": 4, "conv2_1": 4, "conv1_2": 5, "conv1_1": 5}


def derived_bound(layer):
    """Relative L2 bound of a bf16-path gradient against fp32 arithmetic on the SAME forward state and the SAME (bf16-valued) operands.
    The only difference left is that every gradient tensor the backward pass stores is rounded to bf16: relative error <= 2^-8 per
    element (8 significand bits, round to nearest), independent from tensor to tensor, carried on linearly by the layers behind it.
    n roundings in a chain -> sqrt(n) * 2^-8 (each term at its worst-case magnitude, signs independent); + 2^-8 for the 16-bit wire / image
    operand of the layer itself (conv1_1 rounds the fp32 image, the others read stored bf16 activations exactly)."""
    n = 1 + CHAIN.index(layer) + POOLS_BEHIND[layer] + (1 if layer == "conv1_1" else 0)
    return float(np.sqrt(n)) * 2.0 ** -8


def test_bf16_backward_against_the_oracle_on_its_own_forward_state():
    """the referee of the bf16 path's BACKWARD is the oracle (the restatement of trainer_fcn.py:149-158 -- zero_grad /
    backward -- over models.py:114-160 and utils.py:75-102), not the fp32 HIP step.  One bf16 step at 512 x 512, E = 300, K = 59, B = 1,
    train mode; the oracle's fp32 backward runs on THAT pass's forward state (bf16 activations up-cast, their ReLU gates, the pools'
    winners, the Dropout2d factors) with the weights the kernels read (the bf16 image), so no gate / winner flip separates the two --
    and every optimizer-visible gradient must then agree within derived_bound(): ~4e-3 at the head, ~1.8e-2 at conv1_1, instead of the
    0.58 envelope of the test above.  That test's growth towards the input is thereby MEASURED to be the forward state (flips), not the
    backward kernels.  The state is read from a pass that keeps its full maps; the default pass (constant band removed from the conv1_2 /
    conv2 / conv3 blocks) of the same inputs is then held to it within fp32 re-ordering."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers_parity import adopt_forward
    from oracle import szn_oracle as O
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth
    dev = torch.device("cuda", 0)
    emb = synth.make_embeddings(K, E)
    x = synth.make_images(1, H, H, seed=1337)
    t = synth.make_labels(1, H, H, K, seed=1337, classes=list(range(49)))

    def one(keep):
        m = models.FCN32s(E)
        m.load_synthetic(1337, device=dev)
        m.train()
        ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True, keep_grads=True)
        ts.keep_ctx = keep
        eng = m._engine
        eng.dropout_seed, eng.dropout_calls = 1337, 0
        masks = [k.cpu().numpy() for k in eng.make_masks(1, 4096, dev)]
        eng.dropout_calls = 0                                        # the step draws exactly these
        # the operands of the step: the 16-bit weight image (conv1_1 reads its fp32 master and rounds in registers), fp32 biases
        params = {}
        for n in ts.layers:
            o, cnt = ts.woff[n]
            co, ci, kh, kw = getattr(m, n).weight.shape
            w = ts.flat_w_lp[o:o + cnt].float() if n != "conv1_1" else ts.flat_w[o:o + cnt].to(torch.bfloat16).float()
            params[n + ".weight"] = w.view(co, kh, kw, ci).permute(0, 3, 1, 2).contiguous().cpu().numpy()
            bo, bc = ts.boff[n]
            params[n + ".bias"] = ts.flat_b[bo:bo + bc].clone().cpu().numpy()
        for n in ("seenmask_score.weight", "seenmask_score.bias", "seenmask_upscore.weight"):
            params[n] = m.state_dict()[n].detach().float().cpu().numpy()
        kernels = set()
        orig = L.call

        def spy(name, *a):
            orig(name, *a)
            kernels.add(L.last_kernel())
        engine.L.call = models.L.call = spy
        try:
            loss, pred = ts.step(torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev))
        finally:
            engine.L.call = models.L.call = orig
        torch.cuda.synchronize()
        grads = {}
        for n in ts.layers:
            o, cnt = ts.woff[n]
            co, ci, kh, kw = getattr(m, n).weight.shape
            grads[n + ".weight"] = ts.flat_gw[o:o + cnt].view(co, kh, kw, ci).permute(0, 3, 1, 2).float().cpu()
            bo, bc = ts.boff[n]
            grads[n + ".bias"] = ts.flat_gb[bo:bo + bc].float().cpu()
        return float(loss), pred.cpu(), grads, kernels, ts, masks, params

    lossA, predA, gA, kA, ts, masks, params = one(True)
    assert not any("band_remap" in k for k in kA)                    # this pass kept its full maps
    ctx = ts.last_ctx
    om = O.FCN32sOracle(params, E)
    dpool = adopt_forward(om, ctx, x, masks, E)
    assert dpool == 0.0                                              # the oracle's pooling of the bf16 maps == the HIP pools, bit for bit
    # the pools' winner codes are what the backward pass reads: they must name the oracle's winners (first maximum in scan order; 4 = gated)
    for i in range(5):
        code = ctx.pools[i][2].cpu().numpy().transpose(0, 3, 1, 2)
        pin = om.saved["pool%d_in" % (i + 1)]
        idx = om.saved["pool%d_idx" % (i + 1)]
        Wi = pin.shape[3]
        oh, ow = np.meshgrid(np.arange(code.shape[2]), np.arange(code.shape[3]), indexing="ij")
        want = (idx // Wi - 2 * oh[None, None]) * 2 + (idx % Wi - 2 * ow[None, None])
        want = np.where(om.saved["pool%d" % (i + 1)] > 0, want, 4)
        assert np.array_equal(code, want.astype(np.uint8)), "pool%d winner codes" % (i + 1)
    om.saved["conv1_1_in"] = _bf16_round(x)                          # conv1_1's weight gradient rounds the image taps to bf16 in registers
    ts.last_ctx = ctx = None
    f_hip = O.deconv_fwd(om.saved["coarse_f"], np.broadcast_to(O.get_upsampling_weight(1, 1, 64)[0, 0], (E, 64, 64)), H, H, diag=True)
    oloss, odf, _ = O.cosine_loss(f_hip, t, embed=emb)
    assert abs(lossA - float(oloss)) < 1e-5 * max(1.0, abs(float(oloss)))      # fused head (fp32 arithmetic on the fp32 coarse map)
    del f_hip
    og = om.backward(df=odf)
    rows, bad = [], []
    for name in gA:
        layer, kind = name.rsplit(".", 1)
        ref = torch.from_numpy(np.ascontiguousarray(og[name]))
        e2 = rel_l2(gA[name], ref)
        emax = float((gA[name].double() - ref.double()).abs().max() / ref.double().abs().max())
        bound = derived_bound(layer) * (2.0 if kind == "bias" else 1.0)         # (column sums of 10^5 .. 10^6 mixed-sign terms)
        rows.append("%-18s rel L2 %.3e (derived bound %.2e)  max-norm %.3e  cosine %.6f" % (name, e2, bound, emax, cosine(gA[name], ref)))
        if not e2 <= bound:
            bad.append(rows[-1])
    print("bf16 backward vs the oracle's fp32 backward on the bf16 pass's own forward state:")
    print("\n".join(rows))
    assert not bad, bad
    del om, og
    # ---- the default pass (band removed) against the pass that was just pinned
    lossB, predB, gB, kB, _, _, _ = one(False)
    assert any("band" in k for k in kB), sorted(kB)
    print("loss full maps %.8f, band removed %.8f, class maps agree on %.6f" % (lossA, lossB, float((predA == predB).float().mean())))
    assert lossB == lossA and torch.equal(predB, predA)              # a kept pixel sees the very values it would see in the full map
    worst_w = max((rel_l2(gB[n], gA[n]), n) for n in gA if n.endswith(".weight"))
    worst_b = max((rel_l2(gB[n], gA[n]), n) for n in gA if n.endswith(".bias"))
    print("default (band removed) vs full maps: worst relative L2 weights %.3e (%s), biases %.3e (%s)" % (worst_w + worst_b))
    assert worst_w[0] < 5e-5 and worst_b[0] < 2e-3, (worst_w, worst_b)


if __name__ == "__main__":          # child of the test above: the bf16 step under the caller's environment -> torch.save
    loss, pred, grads, kernels = run_step(torch.bfloat16)
    torch.save({"loss": loss, "pred": pred, "grads": grads, "kernels": sorted(kernels)}, sys.argv[1])
