"""The constant band inside the conv3 block (round 5, models._band_cut / szn_band_remap).

models.py:43 pads conv1_1 by 100, so at 1/4 resolution 21 rows / columns per side hold one value per channel; the engine removes 12 of them
per side before conv3_1 (models.py:56-62,123-128 run on 154^2 instead of 178^2 pixels at 512 x 512) and copies a pure pooled row back
behind pool3.  Checked against the SAME step without the shortcut (which the rest of the suite pins to the oracle):
forward bit for bit (a kept pixel sees the values it would see in the full map), backward up to the order of fp32 additions (the gradient of
the copied rows is summed into their representative; every parameter gradient depends on the band only through such sums)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402
from zeroshotsemanticsegmentation_amd import engine, models, synth  # noqa: E402


def test_band_remap_kernel_against_index_arithmetic():
    """szn_band_remap with the four tables of a plan == torch index_select / index_add on the same maps"""
    reg = (23, 51, 2, 72)
    plan = models._BandPlan(reg, reg, 74, 74, torch.device("cuda", 0))
    assert plan.ok and plan.Hc == 50 and plan.Hp == 37 and plan.Hpc == 25
    g = torch.Generator(device="cuda").manual_seed(3)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(2, 74, 74, 64, device="cuda", generator=g).to(dt)

        def remap(t, which, Ho, Wo):
            out = torch.full((t.shape[0], Ho, Wo, t.shape[3]), 7.0, device="cuda", dtype=dt)
            ty, tx = plan.tabs[which]
            L.call("szn_band_remap", L.dtype_code(dt), t.shape[0], t.shape[1], t.shape[2], Ho, Wo, t.shape[3], L.ptr(t), L.ptr(out), L.ptr(ty),
                   L.ptr(tx), L.stream_ptr())
            return out

        def ref(t, which, Ho, Wo):
            ty, tx = (q.cpu().numpy() for q in plan.tabs[which])
            tt = t.float().cpu().numpy()
            out = np.zeros((t.shape[0], Ho, Wo, t.shape[3]), np.float32)
            for y in range(Ho):
                for xx in range(Wo):
                    acc = np.zeros((t.shape[0], t.shape[3]), np.float32)
                    for sy in range(ty[y, 0], ty[y, 0] + ty[y, 1]):
                        for sx in range(tx[xx, 0], tx[xx, 0] + tx[xx, 1]):
                            acc = acc + tt[:, sy, sx]
                    out[:, y, xx] = acc
            return torch.from_numpy(out).to(dt)
        xc = remap(x, "crop", plan.Hc, plan.Wc)
        assert torch.equal(xc.cpu(), ref(x, "crop", plan.Hc, plan.Wc))
        p = torch.randn(2, plan.Hpc, plan.Wpc, 64, device="cuda", generator=g).to(dt)
        assert torch.equal(remap(p, "uncrop", plan.Hp, plan.Wp).cpu(), ref(p, "uncrop", plan.Hp, plan.Wp))
        d = torch.randn(2, plan.Hp, plan.Wp, 64, device="cuda", generator=g).to(dt)
        got, want = remap(d, "uncrop_bwd", plan.Hpc, plan.Wpc).float().cpu(), ref(d, "uncrop_bwd", plan.Hpc, plan.Wpc).float()
        assert float((got - want).abs().max()) <= (0.0 if dt == torch.float32 else 0.07)      # sums of 49 bf16 terms: one rounding
        assert torch.equal(remap(xc, "crop_bwd", 74, 74).cpu(), ref(xc, "crop_bwd", 74, 74))
        # uncrop_bwd is the transpose of uncrop, crop_bwd of crop: <A x, y> == <x, A^T y> (fp32)
        if dt == torch.float32:
            u = remap(p, "uncrop", plan.Hp, plan.Wp)
            assert abs(float((u * d).sum()) - float((p * remap(d, "uncrop_bwd", plan.Hpc, plan.Wpc)).sum())) < 1e-2


def _step(crop, precision, size, B, monkeypatch, train=True):
    monkeypatch.setattr(models, "_BAND_CROP", crop)
    E, K = 20, 33
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda", 0))
    m.train(train)
    emb = synth.make_embeddings(K, E)
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=precision, fused_head=True, keep_grads=True, fused_adam=False)
    x = torch.from_numpy(synth.make_images(B, size, size + 16, seed=9)).cuda()
    t = torch.from_numpy(synth.make_labels(B, size, size + 16, K, seed=10, block=16)).cuda()
    seen = []
    orig = L.call

    def spy(name, *a):
        orig(name, *a)
        seen.append(name)
    models.L.call = spy
    try:
        loss, pred = ts.step(x, t)
    finally:
        models.L.call = orig
    torch.cuda.synchronize()
    return float(loss), pred.cpu(), ts.flat_gw.clone(), ts.flat_gb.clone(), ts, (seen.count("szn_band_remap"),
                                                                                 seen.count("szn_maxpool2x2_ceil_bwd_code_gather"))


@pytest.mark.parametrize("precision,size,B,exact", [(torch.float32, 64, 2, False), (torch.float32, 150, 1, False), (torch.float32, 300, 1, False),
                                                    (torch.bfloat16, 512, 3, True), (torch.bfloat16, 512, 8, True)])
def test_train_step_with_the_band_removed_equals_the_full_step(precision, size, B, exact, monkeypatch):
    """exact: at the bench configuration the conv2_x / conv3_x launches take the same kernels with and without the band (conv3x3_regw;
    conv_igemm_8ph without split-K: 741 vs 990 tiles), so a kept pixel's K terms are added in the same order and the forward pass is
    bit-identical.  At small sizes the smaller maps change the dispatcher's choices (tile kernel, split-K count): equal values, another
    fp32 summation order -- compared in fp32, where that is a 1e-6 effect (on the 16-bit paths a last-bit difference of an activation
    flips ReLU gates downstream: the mechanism tests/test_gpu_headline_pin.py measures; tools/diag_band.py prints both)."""
    l0, p0, gw0, gb0, ts0, n0 = _step(False, precision, size, B, monkeypatch, train=exact)
    l1, p1, gw1, gb1, ts1, n1 = _step(True, precision, size, B, monkeypatch, train=exact)
    # bf16: conv1_1 writes its map cropped (no launch), the fused maps conv1_2 -> conv2 -> conv3 blocks, copy back behind pool3; backward: the
    # transposed maps are applied by the three pools' backward passes while they read (szn_maxpool2x2_ceil_bwd_code_gather: no launch of their
    # own; conv1_1's weight gradient reads the cropped gradient).  fp32: a crop in front and a zero-fill behind instead (its conv1_1 kernels
    # take no cut)
    assert n0 == (0, 0) and n1 == ((5, 3) if precision == torch.float32 else (3, 3))
    if exact:
        assert l1 == l0 and torch.equal(p1, p0)              # forward: bit for bit
        o = ts0.woff["conv4_1"][0]                           # ... and so is everything behind the blocks (conv4_1 .. score_fr)
        assert torch.equal(gw1[o:], gw0[o:])
    else:
        assert abs(l1 - l0) < 2e-6 * abs(l0)
        assert float((p1 == p0).float().mean()) > 0.9999
    # fp32: re-ordering only.  bf16: one more rounding of the summed band gradient -- and, where the dispatcher changed kernels, activations
    # that differ in their last bf16 bit flip a few ReLU gates (tests/test_gpu_headline_pin.py explains the mechanism and its size)
    tol = 3e-5 if precision == torch.float32 else 6e-3
    for n in ts0.layers:
        o, cnt = ts0.woff[n]
        a, b = gw1[o:o + cnt].double(), gw0[o:o + cnt].double()
        err = float((a - b).norm() / (b.norm() + 1e-300))
        assert err < tol, (n, err)
        bo, bc = ts0.boff[n]
        a, b = gb1[bo:bo + bc].double(), gb0[bo:bo + bc].double()
        assert float((a - b).norm() / (b.norm() + 1e-300)) < 5 * tol, n


def _step_cfg(crop, arch, precision, size, B, K, head_fp8, monkeypatch):
    """one train-mode step of a BASELINE configuration (E = 300) with / without the band -> loss, pred, flat gradients, TrainStep, launches"""
    monkeypatch.setattr(models, "_BAND_CROP", crop)
    E = 300
    m = (models.FCN8s if arch == "fcn8s" else models.FCN32s)(E)
    m.load_synthetic(1337, device=torch.device("cuda", 0))
    m.train()
    if head_fp8:
        m.set_head_precision("fp8")
    m._engine.dropout_seed, m._engine.dropout_calls = 1337, 0
    emb = synth.make_embeddings(K, E)
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=precision, fused_head=True, keep_grads=True, fused_adam=False)
    x = torch.from_numpy(synth.make_images(B, size, size, seed=9)).cuda()
    t = torch.from_numpy(synth.make_labels(B, size, size, K, seed=10)).cuda()
    seen = []
    orig = L.call

    def spy(name, *a):
        orig(name, *a)
        seen.append(name)
    models.L.call = engine.L.call = spy
    try:
        loss, pred = ts.step(x, t)
    finally:
        models.L.call = engine.L.call = orig
    torch.cuda.synchronize()
    return float(loss), pred.cpu(), ts.flat_gw.clone(), ts.flat_gb.clone(), ts, (seen.count("szn_band_remap"), seen.count("szn_conv1_1_fwd_c"),
                                                                                 seen.count("szn_maxpool2x2_ceil_bwd_code_gather"))


@pytest.mark.parametrize("tag,arch,precision,size,B,K,head_fp8", [
    ("configs[1] FCN8s bf16 512", "fcn8s", torch.bfloat16, 512, 3, 21, False),
    ("configs[4] 768 fp16 + fp8 head", "fcn32s", torch.float16, 768, 2, 59, True)])
def test_configs1_and_configs4_steps_with_the_band_removed_equal_their_full_steps(tag, arch, precision, size, B, K, head_fp8, monkeypatch):
    """the band plan also engages on BASELINE configs[1] (FCN8s: the skip heads read pool3 / pool4, which the engine hands
    over as full maps -- the copy back behind pool3 -- and whose gradients join the full-map chain in front of the pools' backward pass) and on
    configs[4] (768 x 768: 966^2 maps, other cut positions; fp16 with the dynamic loss scale and the e4m3 projection).  Same statement as for the
    headline shape: forward bit for bit, everything behind the cropped blocks bit for bit, the cropped blocks' gradients to the order of fp32
    additions + one more 16-bit rounding of the summed band gradient."""
    l0, p0, gw0, gb0, ts0, n0 = _step_cfg(False, arch, precision, size, B, K, head_fp8, monkeypatch)
    l1, p1, gw1, gb1, ts1, n1 = _step_cfg(True, arch, precision, size, B, K, head_fp8, monkeypatch)
    print("%s: launches (band_remap, conv1_1_fwd_c, pool bwd gather) full %s, band removed %s" % (tag, n0, n1))
    assert n0 == (0, 0, 0)
    assert n1[0] >= 3 and n1[1] == 1 and n1[2] >= (2 if arch == "fcn8s" else 3)    # (FCN8s: pool3's backward also adds the skip gradient: two passes)
    assert l1 == l0 and torch.equal(p1, p0)
    o = ts0.woff["conv4_1"][0]
    assert torch.equal(gw1[o:], gw0[o:])
    for n in ts0.layers:
        o, cnt = ts0.woff[n]
        a, b = gw1[o:o + cnt].double(), gw0[o:o + cnt].double()
        err = float((a - b).norm() / (b.norm() + 1e-300))
        assert err < 6e-3, (n, err)
        bo, bc = ts0.boff[n]
        a, b = gb1[bo:bo + bc].double(), gb0[bo:bo + bc].double()
        assert float((a - b).norm() / (b.norm() + 1e-300)) < 3e-2, n


def test_inference_forward_with_the_band_removed(monkeypatch):
    out = []
    for crop in (False, True):
        monkeypatch.setattr(models, "_BAND_CROP", crop)
        m = models.FCN32s(20)
        m.load_synthetic(1337, device=torch.device("cuda", 0))
        m.eval()
        x = torch.from_numpy(synth.make_images(1, 200, 136, seed=4)).cuda()
        with torch.no_grad():
            out.append(m(x, mode="both"))
    # (fp32, one small image: the smaller conv3 maps change the dispatcher's split-K counts -- same values, another summation order)
    for a, b in zip(out[0], out[1]):
        assert float((a - b).abs().max()) < 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(2, 96, 80), (1, 512, 512), (3, 33, 47)])
def test_conv1_1_cropped_forms_equal_the_full_ones(dt, geom):
    """szn_conv1_1_fwd_c == the kept rows / columns of szn_conv1_1_fwd, bit for bit; szn_conv1_1_wgrad_c on the cropped gradient == szn_conv1_1_wgrad
    on the full one (the removed pixels see no image pixel: their products are exact zeros), bit for bit"""
    B, H, W = geom
    code = L.dtype_code(dt)
    g = torch.Generator(device="cuda").manual_seed(H)
    x = (torch.rand(B, 3, H, W, device="cuda", generator=g) * 255 - 120).contiguous()
    w = torch.randn(64, 3, 3, 3, device="cuda", generator=g) * 0.1
    bias = torch.randn(64, device="cuda", generator=g) * 0.1
    Ho, Wo = H + 198, W + 198
    plan = models._BandPlan(models._cb_conv1_1(H, 100), models._cb_conv1_1(W, 100), Ho, Wo, torch.device("cuda", 0), 1)
    assert plan.ok
    st = L.stream_ptr()
    full = torch.empty(B, Ho, Wo, 64, device="cuda", dtype=dt)
    L.call("szn_conv1_1_fwd", code, B, H, W, 100, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(full), st)
    crop = torch.full((B, plan.Hc, plan.Wc, 64), 9.0, device="cuda", dtype=dt)
    L.call("szn_conv1_1_fwd_c", code, B, H, W, 100, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(crop), plan.cut8, st)
    ky = plan.tabs["crop"][0][:, 0].long()
    kx = plan.tabs["crop"][1][:, 0].long()
    assert torch.equal(crop, full[:, ky][:, :, kx])
    # weight gradient
    dfull = (torch.randn(B, Ho, Wo, 64, device="cuda", generator=g) * 0.01).to(dt)
    dcrop = dfull[:, ky][:, :, kx].contiguous()
    nb = L.load().szn_conv1_1_wgrad_workspace_bytes(code, B, H, W, 100)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    dw0, dw1 = torch.zeros(64, 3, 3, 3, device="cuda"), torch.zeros(64, 3, 3, 3, device="cuda")
    L.call("szn_conv1_1_wgrad", code, B, H, W, 100, L.ptr(x), L.ptr(dfull), L.ptr(dw0), None, 0, L.ptr(ws), st)
    L.call("szn_conv1_1_wgrad_c", code, B, H, W, 100, L.ptr(x), L.ptr(dcrop), L.ptr(dw1), 0, L.ptr(ws), plan.cut8, st)
    torch.cuda.synchronize()
    assert float(dw0.abs().max()) > 0 and torch.equal(dw0, dw1)
    # fp32 has no cropped form: refused, nothing written
    out32 = torch.full((B, plan.Hc, plan.Wc, 64), 5.0, device="cuda")
    assert L.load().szn_conv1_1_fwd_c(L.SZN_F32, B, H, W, 100, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(out32), plan.cut8, st) != 0
    torch.cuda.synchronize()
    assert float(out32.min()) == 5.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pool_backward_reads_through_the_transposed_band_map(dt):
    """szn_band_fold (rows, then columns, in place) + szn_maxpool2x2_ceil_bwd_code_gather == the two separable szn_band_remap passes followed by
    szn_maxpool2x2_ceil_bwd_code, bit for bit -- din and the column sums (the producer's bias gradient); the multi-source form of the gather
    (no fold) gives the same sums up to fp32 association / one bf16 rounding instead of two"""
    B, H, W, Cc = 2, 178, 170, 64
    plan = models._BandPlan((23, 155, 2, 176), (21, 151, 2, 168), H, W, torch.device("cuda", 0), 3)
    assert plan.ok and plan.gather_uncrop["runs_y"] is not None and plan.gather_uncrop["runs_x"] is not None
    g = torch.Generator(device="cuda").manual_seed(2)
    d = torch.randn(B, plan.Hp, plan.Wp, Cc, device="cuda", generator=g).to(dt)          # gradient of the FULL pooled map
    codes = torch.randint(0, 5, (B, plan.Hpc, plan.Wpc, Cc), device="cuda", generator=g, dtype=torch.uint8)
    code = L.dtype_code(dt)
    Hi, Wi = plan.Hc, plan.Wc                                                            # the pool's (cropped) input

    def run(form):
        dn = torch.full((B, Hi, Wi, Cc), 3.0, device="cuda", dtype=dt)
        ro = L.rows_out()
        db = torch.zeros(Cc, device="cuda")
        slab = torch.zeros(1024, Cc, device="cuda")
        if form == "two passes":
            t = d
            for which, ho, wo in (("uncrop_bwd_y", plan.Hpc, plan.Wp), ("uncrop_bwd_x", plan.Hpc, plan.Wpc)):
                ty, tx = plan.tabs[which]
                out = torch.empty(B, ho, wo, Cc, device="cuda", dtype=dt)
                L.call("szn_band_remap", code, B, t.shape[1], t.shape[2], ho, wo, Cc, L.ptr(t), L.ptr(out), L.ptr(ty), L.ptr(tx), L.stream_ptr())
                t = out
            L.call("szn_maxpool2x2_ceil_bwd_code", code, B, Hi, Wi, Cc, L.ptr(codes), L.ptr(t), L.ptr(dn), L.ptr(db), L.ptr(slab), 1024,
                   C.byref(ro), L.stream_ptr())
        else:
            t = d.clone()
            ty, tx = plan.tabs["uncrop_bwd"]
            if form == "fold":
                gf = plan.gather_uncrop
                for axis, runs in ((0, gf["runs_y"]), (1, gf["runs_x"])):
                    L.call("szn_band_fold", code, B, plan.Hp, plan.Wp, Cc, L.ptr(t), axis, L.ptr(runs), runs.shape[0], L.stream_ptr())
                ty, tx = gf["tabs"]
            L.call("szn_maxpool2x2_ceil_bwd_code_gather", code, B, Hi, Wi, Cc, L.ptr(codes), L.ptr(t), plan.Hp, plan.Wp, L.ptr(ty), L.ptr(tx),
                   L.ptr(dn), L.ptr(db), L.ptr(slab), 1024, C.byref(ro), L.stream_ptr())
        rows = ro.value
        torch.cuda.synchronize()
        return dn, slab[:rows].clone()
    (a, ca), (b, cb), (m, cm) = run("fold"), run("two passes"), run("multi-source")
    assert torch.equal(a, b) and torch.equal(ca, cb)
    assert not (a == 3.0).all(dim=-1).any()
    ty, tx = (torch.tensor(t) for t in plan.host["uncrop_bwd"])
    single = (ty[:, 1] == 1)[:, None] & (tx[:, 1] == 1)[None, :]                         # pooled pixels with one source
    single = single.repeat_interleave(2, 0)[:Hi].repeat_interleave(2, 1)[:, :Wi].cuda()
    assert 0.8 < float(single.float().mean()) < 1.0
    assert torch.equal(m[:, single], b[:, single])
    tol = 1e-6 if dt == torch.float32 else 1.6e-2
    assert float((m.float() - b.float()).abs().max()) <= tol * float(b.float().abs().max())
    with pytest.raises(L.SznError):
        L.call("szn_band_fold", code, B, plan.Hp, plan.Wp, Cc, L.ptr(d), 2, L.ptr(plan.gather_uncrop["runs_y"]), 1, L.stream_ptr())
