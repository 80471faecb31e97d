"""GPU: the low-precision variants of BASELINE configs[4] -- fp8 (OCP e4m3) projection GEMM and fp16 activations --
validated against the fp32 path with stated tolerances and the class-assignment agreement rate.
There is no reference code for these variants (parity unpinned by the reference); the referee is the oracle's restatement
of the quantisation (oracle.e4m3_round, itself checked against torch's float8_e4m3fn cast) and the fp32 HIP path."""
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


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16, torch.float16])
def test_proj_fp8_kernel_matches_oracle_restatement(xdt):
    M, K, N, ldo = 2 * 25 * 25 + 3, 4096, 302, 320
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(M, K, generator=g)) * 3.0                   # post-ReLU / dropout-scaled features
    x[5] = 0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    xd, wd = x.to(xdt).cuda(), w.to(xdt).cuda()
    out = torch.full((M, ldo), 7.0, device="cuda")
    ws = torch.empty(L.load().szn_proj_fp8_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    L.call("szn_proj_fp8_fwd", L.dtype_code(xdt), L.dtype_code(xdt), M, K, N, ldo, L.ptr(xd), L.ptr(wd), L.ptr(bias.cuda()),
           L.ptr(out), L.ptr(ws), L.stream_ptr())
    torch.cuda.synchronize()
    want = O.proj_fp8(xd.float().cpu().numpy(), wd.float().cpu().numpy(), bias.numpy())
    got = out.cpu().numpy()
    assert np.abs(got[:, :N] - want).max() < 2e-5 * np.abs(want).max()     # same quantised operands, fp32 accumulation order only
    assert (got[:, N:] == 7.0).all()                                       # columns behind N untouched
    # and how far the fp8 product is from the fp32 one (3 mantissa bits per operand, K = 4096 terms)
    ref = xd.float().cpu().numpy().astype(np.float64) @ wd.float().cpu().numpy().astype(np.float64).T + bias.numpy()
    err = np.abs(got[:, :N] - ref).max() / np.abs(ref).max()
    print("fp8 projection vs fp32 product (%s operands): max error %.3e of the output scale" % (xdt, err))
    assert err < 6e-2                                    # per-tensor e4m3 on half-normal data whose amax sits at ~4.5 sigma


def test_fp8_head_forward_agreement_768():
    """BASELINE configs[4] geometry: 768x768, E = 300, K = 59: class assignment with the fp8 head vs the same bf16 network
    with its native head, and vs the fp32 path"""
    E, K, H = 300, 59, 768
    emb = cu(synth.make_embeddings(K, E))
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.eval()
    x = cu(synth.make_images(1, H, H, seed=91))
    with torch.no_grad():
        f32 = m(x, mode="fcn")
        p32 = utils.infer_lbl_device(f32, emb)
        m.set_precision(torch.bfloat16)
        f16 = m(x, mode="fcn")
        p16 = utils.infer_lbl_device(f16, emb)
        m.set_head_precision("fp8")
        f8 = m(x, mode="fcn")
        p8 = utils.infer_lbl_device(f8, emb)
        m.set_head_precision("native")
    scale = float(f32.abs().max())
    e16 = float((f16 - f32).abs().max()) / scale
    e8 = float((f8 - f32).abs().max()) / scale
    a16 = float((p16 == p32).float().mean())
    a8 = float((p8 == p32).float().mean())
    a8_16 = float((p8 == p16).float().mean())
    print("768x768 score error vs fp32: bf16 %.3e, bf16 + fp8 head %.3e; argmax agreement vs fp32: bf16 %.4f, fp8 head %.4f; "
          "fp8 head vs bf16 head %.4f" % (e16, e8, a16, a8, a8_16))
    assert e16 < 5e-2 and e8 < 8e-2                    # stated tolerance: fp8 head within 8 % of the output scale after 16 layers
    assert a8 > 0.99 and a8_16 > 0.99                  # class-assignment agreement rate (measured 1.0000 on this input)


def test_fp16_forward_and_train_step():
    """SZN_F16 end to end: IEEE-half activations / weight images through every kernel of the step, static loss scaling
    (TrainStep.loss_scale, default 4096) so that the 1e-7-sized activation gradients survive the 16-bit backward pass"""
    E, K, H = 20, 33, 96
    emb = synth.make_embeddings(K, E)
    x = cu(synth.make_images(2, H, H, seed=71))
    t = cu(synth.make_labels(2, H, H, K, seed=72, block=16))
    sd = None
    grads, losses = {}, {}
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        m = models.FCN32s(E)
        m.load_synthetic(1337, device=torch.device("cuda"))
        if sd is None:
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.eval()
        ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=dt, fused_head=True)
        assert ts.loss_scale == (4096.0 if dt == torch.float16 else 1.0)
        loss, pred = ts.step(x, t)
        torch.cuda.synchronize()
        losses[dt] = float(loss)
        grads[dt] = {n: (getattr(m, n).weight.grad.detach().float() / ts.loss_scale).cpu() for n in ("conv1_2", "conv3_2", "fc6", "score_fr")}
        assert all(torch.isfinite(g).all() for g in grads[dt].values())
        if dt == torch.float16:
            l2 = [float(ts.step(x, t)[0]) for _ in range(3)]
            assert all(np.isfinite(l2)) and l2[-1] < losses[dt]
    assert abs(losses[torch.float16] - losses[torch.float32]) < 5e-3       # 10 mantissa bits: closer to fp32 than bf16 is
    assert abs(losses[torch.bfloat16] - losses[torch.float32]) < 2e-2
    for n in grads[torch.float32]:
        ref = grads[torch.float32][n]
        e16 = float((grads[torch.float16][n] - ref).norm() / ref.norm())
        eb = float((grads[torch.bfloat16][n] - ref).norm() / ref.norm())
        print("%s gradient, relative l2 error vs fp32: fp16 %.3e, bf16 %.3e" % (n, e16, eb))
        # 16-bit activations flip ReLU gates / pooling winners against the fp32 pass, so these are coarse: the point is
        # that the unscaled fp16 gradients are finite, correctly scaled and closer to fp32 than the bf16 ones
        assert e16 < 0.2 and e16 < eb


def test_configs4_train_step_as_configured():
    """BASELINE configs[4] as one train step: 768x768, E = 300, K = 59, IEEE-half activations / weight images with loss scaling
    + the fp8 (e4m3) projection head, B = 2.  Referees: the fp32 oracle's forward + cosine loss on the same images (first-step
    loss), the fp32 HIP path (class-assignment agreement rate), and the run itself (finite, decreasing loss, finite gradients).
    Stated tolerances: |loss - oracle| < 3e-2 (fp16 activations through 16 layers + per-tensor e4m3 head operands, measured
    ~5e-3), class agreement > 0.97 of the pixels."""
    E, K, H, B = 300, 59, 768, 2
    emb = synth.make_embeddings(K, E)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.eval()
    x = synth.make_images(B, H, H, seed=93)
    t = synth.make_labels(B, H, H, K, seed=94, classes=list(range(49)))
    om = O.FCN32sOracle({k: v.detach().cpu().numpy() for k, v in m.named_parameters() if k.split(".")[0] != "upscore"}, E)
    ol = []
    for b in range(B):                                    # the reference's loss is per image (n = 1); batched = mean over images
        of = om.forward(x[b:b + 1], "fcn")
        ol.append(float(O.cosine_loss(of, t[b:b + 1], embed=emb, want_grad=False)[0]))
        del of
    oloss = float(np.mean(ol))
    xd, td = cu(x), cu(t)
    loss32, pred32 = m.embed_predict(xd, emb, td)         # fp32 HIP path, fused head
    assert abs(float(loss32) - oloss) < 1e-4
    m.set_head_precision("fp8")
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.float16, fused_head=True)
    assert ts.loss_scale > 1.0 and m._engine.head_fp8
    losses = []
    for i in range(3):
        loss, pred = ts.step(xd, td)
        losses.append(float(loss))
        if i == 0:
            agree = float((pred == pred32).float().mean())
            assert L.last_kernel() is not None
            for n in ("conv1_1", "conv3_2", "fc6", "fc7", "score_fr"):
                g = getattr(m, n).weight.grad
                assert torch.isfinite(g).all() and float(g.abs().max()) > 0, n
    print("configs[4] step: loss %s (fp32 oracle %.5f, fp32 HIP %.5f), class agreement with the fp32 path %.4f"
          % (["%.5f" % l for l in losses], oloss, float(loss32), agree))
    assert abs(losses[0] - oloss) < 3e-2
    assert agree > 0.97
    assert all(np.isfinite(losses)) and losses[2] < losses[0]
    m.set_head_precision("native")


def _fp16_step(loss_scale=None, **kw):
    E, K, H = 20, 33, 96
    emb = synth.make_embeddings(K, E)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.eval()
    ts = engine.TrainStep(m, emb, optimizer=kw.pop("optimizer", "adam"), lr=1e-5, precision=torch.float16, fused_head=True,
                          loss_scale=loss_scale, **kw)
    x = cu(synth.make_images(2, H, H, seed=71))
    t = cu(synth.make_labels(2, H, H, K, seed=72, block=16))
    return m, ts, x, t


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_dynamic_loss_scale_skips_overflowed_steps_and_recovers(optimizer):
    """fp16 path: a loss scale far too large makes the 16-bit activation gradients overflow -> the whole optimizer step is
    skipped on the device (masters, moments, 16-bit weight image untouched), the scale is halved, and once the gradients are
    finite again steps are applied and counted; after `scale_growth_interval` clean steps the scale doubles"""
    m, ts, x, t = _fp16_step(loss_scale=2.0 ** 40, optimizer=optimizer, scale_growth_interval=3)
    assert ts.dynamic and ts.loss_scale == 2.0 ** 40
    w0, lp0 = ts.flat_w.clone(), ts.flat_w_lp.clone()
    st0 = [s.clone() for s in ts.state["w"]]
    loss, _ = ts.step(x, t)
    assert np.isfinite(float(loss))                                   # the forward pass is unaffected by the scale
    assert not torch.isfinite(ts.flat_gw).all()                       # the scaled 16-bit backward overflowed
    assert torch.equal(ts.flat_w, w0) and torch.equal(ts.flat_w_lp, lp0)
    assert all(torch.equal(a, b) for a, b in zip(ts.state["w"], st0))
    assert ts.loss_scale == 2.0 ** 39 and ts.applied_steps == 0
    scales = []
    for _ in range(40):
        ts.step(x, t)
        scales.append(ts.loss_scale)
        if ts.applied_steps >= 4:
            break
    assert ts.applied_steps >= 4, scales
    assert torch.isfinite(ts.flat_w).all() and not torch.equal(ts.flat_w, w0)
    assert torch.equal(ts.flat_w_lp.float(), ts.flat_w.half().float())           # the weight image follows the masters
    k = next(i for i in range(1, len(scales)) if scales[i] == scales[i - 1])    # first clean step
    assert all(scales[i] == scales[i - 1] / 2 for i in range(1, k))              # halved on every overflow before it
    assert max(scales[k:]) == 2 * scales[k] or len(scales) - k < 3               # grown after 3 clean steps
    # checkpoints carry the number of APPLIED steps
    class _Opt(object):
        state = __import__("collections").defaultdict(dict)
    ts.export_optimizer_state(_Opt)
    if optimizer == "adam":
        assert all(int(v["step"]) == ts.applied_steps for v in _Opt.state.values())


def test_dynamic_loss_scale_equals_static_when_nothing_overflows():
    ma, tsa, x, t = _fp16_step(dynamic_loss_scale=True)
    mb, tsb, _, _ = _fp16_step(dynamic_loss_scale=False)
    assert tsa.dynamic and not tsb.dynamic and tsa.loss_scale == tsb.loss_scale == 4096.0
    for _ in range(3):
        la, _ = tsa.step(x, t)
        lb, _ = tsb.step(x, t)
        assert float(la) == float(lb)
    assert tsa.applied_steps == 3
    assert torch.equal(tsa.flat_gw, tsb.flat_gw)
    d = (tsa.flat_w - tsb.flat_w).abs().max().item()
    assert d <= 1e-6 * 3e-5 + 1e-12, d                  # in-kernel bias correction (double pow) vs the host's: rounding only


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_proj_fp8_backward_kernels_match_oracle_restatement(dt):
    """the e5m2-gradient x e4m3-operand backward GEMMs of the projection layer (szn_proj_fp8_dgrad / _wgrad) against the
    oracle's restatement of the same quantisation, and against the fp32 products"""
    M, K, N, ldg = 2 * 25 * 25 + 3, 4096, 302, 320
    gen = torch.Generator().manual_seed(11)
    g = torch.zeros(M, ldg)
    g[:, :N] = torch.randn(M, N, generator=gen) * torch.exp(torch.randn(M, 1, generator=gen) * 2.0) * 1e-4    # wide dynamic range
    x = torch.relu(torch.randn(M, K, generator=gen)) * 3.0
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    gd, xd, wd = g.to(dt).cuda(), x.to(dt).cuda(), w.to(dt).cuda()
    gs, xs, ws_ = gd.float().cpu().numpy()[:, :N], xd.float().cpu().numpy(), wd.float().cpu().numpy()
    code = L.dtype_code(dt)
    wsb = torch.empty(L.load().szn_proj_fp8_bwd_workspace_bytes(M, K, N), dtype=torch.uint8, device="cuda")
    # dgrad, no epilogue, fp32 output
    dx = torch.full((M, K), 7.0, device="cuda")
    L.call("szn_proj_fp8_dgrad", code, code, M, K, N, ldg, L.ptr(gd), L.ptr(wd), None, 0, 0, None, 1, L.SZN_F32, L.ptr(dx), K,
           L.ptr(wsb), L.stream_ptr())
    want = O.proj_fp8_dgrad(gs, ws_)
    got = dx.cpu().numpy()
    assert np.abs(got - want).max() < 1e-4 * np.abs(want).max()          # same quantised operands: fp32 summation order only
    ref = gs.astype(np.float64) @ ws_.astype(np.float64)
    e_d = np.abs(got - ref).max() / np.abs(ref).max()
    # with the epilogue of the training step: ReLU gate of the producing layer, Dropout2d factor per (image, channel), 16-bit output
    scale = (torch.rand(3, K, generator=gen) > 0.5).float().cuda() * 2.0
    rows = (M + 2) // 3
    odt = dt if dt != torch.float32 else torch.bfloat16
    dx2 = torch.empty(M, K, device="cuda", dtype=odt)
    L.call("szn_proj_fp8_dgrad", code, code, M, K, N, ldg, L.ptr(gd), L.ptr(wd), L.ptr(xd), code, K, L.ptr(scale), rows,
           L.dtype_code(odt), L.ptr(dx2), K, L.ptr(wsb), L.stream_ptr())
    img = np.minimum(np.arange(M) // rows, 2)
    want2 = torch.from_numpy(np.where(xs > 0, want, 0.0) * scale.cpu().numpy()[img]).to(odt).float().numpy()
    ulp = 2.0 ** -7 if odt == torch.bfloat16 else 2.0 ** -10           # one unit of the 16-bit output format (relative)
    diff = np.abs(dx2.float().cpu().numpy() - want2)
    assert (diff <= ulp * np.abs(want2) + 4e-5 * np.abs(want2).max()).all()
    # wgrad
    dw = torch.full((N, K), 7.0, device="cuda")
    L.call("szn_proj_fp8_wgrad", code, code, M, K, N, ldg, K, L.ptr(gd), L.ptr(xd), L.ptr(dw), L.ptr(wsb), L.stream_ptr())
    wantw = O.proj_fp8_wgrad(gs, xs)
    gotw = dw.cpu().numpy()
    assert np.abs(gotw - wantw).max() < 5e-4 * np.abs(wantw).max()       # 1,253-term fp32 sums of mixed-sign products vs float64
    refw = gs.astype(np.float64).T @ xs.astype(np.float64)
    e_w = np.abs(gotw - refw).max() / np.abs(refw).max()
    print("fp8 backward vs fp32 products (%s operands): dgrad %.3e, wgrad %.3e of the output scale" % (dt, e_d, e_w))
    assert e_d < 0.15 and e_w < 0.15                       # e5m2 keeps 2 mantissa bits of a gradient spanning ~6 decades


def test_fp8_backward_train_step_tracks_the_16bit_head():
    """TrainStep with set_head_precision('fp8_bwd'): score_fr / fc7 gradients against the straight-through ('fp8') step on the
    same batch, and a decreasing loss"""
    E, K, H = 300, 59, 256
    emb = synth.make_embeddings(K, E)
    x = cu(synth.make_images(2, H, H, seed=75))
    t = cu(synth.make_labels(2, H, H, K, seed=76, classes=list(range(49))))
    grads = {}
    for kind in ("fp8", "fp8_bwd"):
        m = models.FCN32s(E)
        m.load_synthetic(1337, device=torch.device("cuda"))
        m.eval()
        m.set_head_precision(kind)
        ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True)
        losses = [float(ts.step(x, t)[0])]
        grads[kind] = {n: getattr(m, n).weight.grad.detach().float().clone() for n in ("score_fr", "fc7", "fc6", "conv3_2")}
        losses += [float(ts.step(x, t)[0]) for _ in range(2)]
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], (kind, losses)
        assert L.last_kernel() is not None
    for n, ref in grads["fp8"].items():
        err = float((grads["fp8_bwd"][n] - ref).norm() / ref.norm())
        print("%s weight gradient, fp8 backward vs 16-bit backward of the head: relative l2 error %.3e" % (n, err))
        assert torch.isfinite(grads["fp8_bwd"][n]).all() and err < 0.25, (n, err)
