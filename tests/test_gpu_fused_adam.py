"""szn_conv2d_wgrad_adam: the Adam step of fc6 / fc7 applied in the epilogue of their weight-gradient kernel (conv_wgrad_wide<T, true>).

The reference runs `loss.backward(); optim.step()` (train.py:170-175, torch.optim.Adam): a weight's update depends only on its own
gradient, so applying it where the gradient tile is produced is the same computation.  Checked here bit for bit against the separate
sequence szn_conv2d_wgrad -> szn_adam_step (which the rest of the suite pins to the oracle / golden g7), at the C-ABI and through
engine.TrainStep."""
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


def _operands(dt, B, Hi, Ci, Co, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Ho = Hi - k + 1
    x = torch.relu(torch.randn(B, Hi, Hi, Ci, device="cuda", generator=g)).to(dt)
    dout = (torch.randn(B, Ho, Ho, Co, device="cuda", generator=g) * 1e-3).to(dt)
    n = Co * k * k * Ci
    p = torch.randn(n, device="cuda", generator=g) * 0.02
    m1 = torch.randn(n, device="cuda", generator=g) * 1e-4
    m2 = torch.rand(n, device="cuda", generator=g) * 1e-6
    return x, dout, p, m1, m2


@pytest.mark.parametrize("case", [(torch.bfloat16, 2, 9, 512, 4096, 7, 3), (torch.float16, 1, 8, 512, 4096, 7, 1),
                                  (torch.bfloat16, 3, 5, 4096, 4096, 1, 2), (torch.bfloat16, 1, 7, 640, 4096, 7, 5),
                                  (torch.bfloat16, 8, 23, 512, 4096, 7, 2), (torch.bfloat16, 2, 9, 512, 4032, 7, 4)])
@pytest.mark.parametrize("keep", [True, False])
@pytest.mark.parametrize("half", ["1", "0"])
def test_wgrad_adam_equals_wgrad_then_adam(case, keep, half, monkeypatch):
    # half = "1": conv_wgrad_half (round 5: 128 x 256 tiles, two four-wave blocks per CU), "0": conv_wgrad_wide<T, true> (round 4)
    monkeypatch.setenv("SZN_WGW_HALF", half)
    dt, B, Hi, Ci, Co, k, step = case
    code = L.dtype_code(dt)
    x, dout, p0, m10, m20 = _operands(dt, B, Hi, Ci, Co, k, seed=11 + step)
    Ho = Hi - k + 1
    d = L.ConvDesc(code, B, Hi, Hi, Ci, Ho, Ho, Co, k, k, 0, Ci, Co, 0, 0, 0)
    assert L.load().szn_conv2d_wgrad_adam_supported(C.byref(d)) == 1
    hyp = dict(lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.01, gs=1.0 / 64)
    n = p0.numel()
    st = L.stream_ptr()
    # the separate sequence
    dw = torch.empty(n, device="cuda")
    p, m1, m2 = p0.clone(), m10.clone(), m20.clone()
    lp = torch.zeros(n, device="cuda", dtype=dt)
    L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), 0, st)
    assert L.last_kernel() == "conv_wgrad_wide"
    L.call("szn_adam_step", n, L.ptr(p), L.ptr(dw), L.ptr(m1), L.ptr(m2), hyp["lr"], hyp["b1"], hyp["b2"], hyp["eps"], hyp["wd"], step,
           hyp["gs"], L.ptr(lp), code, st)
    # one launch
    pf, m1f, m2f = p0.clone(), m10.clone(), m20.clone()
    lpf = torch.zeros(n, device="cuda", dtype=dt)
    dwf = torch.full((n,), 7.0, device="cuda")
    a = L.AdamArgs()
    a.param, a.exp_avg, a.exp_avg_sq, a.w_lp, a.w_lp_dtype = pf.data_ptr(), m1f.data_ptr(), m2f.data_ptr(), lpf.data_ptr(), code
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.step, a.grad_scale = hyp["lr"], hyp["b1"], hyp["b2"], hyp["eps"], hyp["wd"], step, hyp["gs"]
    L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dwf) if keep else None, C.byref(a), st)
    assert L.last_kernel() == ("conv_wgrad_half_adam" if half == "1" else "conv_wgrad_wide_adam")
    torch.cuda.synchronize()
    assert float((p - p0).abs().max()) > 0
    assert torch.equal(pf, p) and torch.equal(m1f, m1) and torch.equal(m2f, m2) and torch.equal(lpf, lp)
    if keep:
        assert torch.equal(dwf, dw)
    else:
        assert float((dwf - 7.0).abs().max()) == 0.0              # untouched
    # dw handed over but declared optional: not written
    ph, m1h, m2h = p0.clone(), m10.clone(), m20.clone()
    dwh = torch.full((n,), 7.0, device="cuda")
    a.param, a.exp_avg, a.exp_avg_sq, a.grad_optional = ph.data_ptr(), m1h.data_ptr(), m2h.data_ptr(), 1
    lph = torch.zeros(n, device="cuda", dtype=dt)
    a.w_lp = lph.data_ptr()
    L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dwh), C.byref(a), st)
    torch.cuda.synchronize()
    assert torch.equal(ph, p) and torch.equal(lph, lp) and float((dwh - 7.0).abs().max()) == 0.0
    a.grad_optional = 0
    # no weight image: masters and moments only
    pg, m1g, m2g = p0.clone(), m10.clone(), m20.clone()
    a.param, a.exp_avg, a.exp_avg_sq, a.w_lp = pg.data_ptr(), m1g.data_ptr(), m2g.data_ptr(), None
    L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), None, C.byref(a), st)
    torch.cuda.synchronize()
    assert torch.equal(pg, p) and torch.equal(m1g, m1) and torch.equal(m2g, m2)


def test_wgrad_adam_refuses_other_layers():
    lib = L.load()
    x = torch.zeros(1, 20, 20, 256, device="cuda", dtype=torch.bfloat16)
    dout = torch.zeros(1, 20, 20, 256, device="cuda", dtype=torch.bfloat16)
    n = 256 * 9 * 256
    buf = torch.zeros(4, n, device="cuda")
    a = L.AdamArgs()
    a.param, a.exp_avg, a.exp_avg_sq, a.step = buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), 1
    a.lr, a.beta1, a.beta2, a.eps = 1e-3, 0.9, 0.999, 1e-8
    d = L.ConvDesc(L.SZN_BF16, 1, 20, 20, 256, 20, 20, 256, 3, 3, 1, 256, 256, 0, 0, 0)        # 9 x 1 x 1 tiles: below the kernel's range
    assert lib.szn_conv2d_wgrad_adam_supported(C.byref(d)) == 0
    with pytest.raises(L.SznError):
        L.call("szn_conv2d_wgrad_adam", C.byref(d), L.ptr(x), L.ptr(dout), None, C.byref(a), L.stream_ptr())
    torch.cuda.synchronize()
    assert float(buf.abs().max()) == 0.0                          # nothing was launched
    d32 = L.ConvDesc(L.SZN_F32, 1, 7, 7, 512, 1, 1, 4096, 7, 7, 0, 512, 4096, 0, 0, 0)
    assert lib.szn_conv2d_wgrad_adam_supported(C.byref(d32)) == 0
    a.step = 0
    dfc = L.ConvDesc(L.SZN_BF16, 1, 7, 7, 512, 1, 1, 4096, 7, 7, 0, 512, 4096, 0, 0, 0)
    big = torch.zeros(3, 4096 * 49 * 512, device="cuda")
    a.param, a.exp_avg, a.exp_avg_sq = big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr()
    xx = torch.zeros(1, 7, 7, 512, device="cuda", dtype=torch.bfloat16)
    dd = torch.zeros(1, 1, 1, 4096, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(L.SznError):                               # step < 1
        L.call("szn_conv2d_wgrad_adam", C.byref(dfc), L.ptr(xx), L.ptr(dd), None, C.byref(a), L.stream_ptr())


def test_trainstep_fuses_fc7_too_when_asked(monkeypatch):
    monkeypatch.setenv("SZN_FUSED_ADAM_LAYERS", "fc6,fc7")
    ref, l0 = _run_steps(False, True, torch.bfloat16, 2)
    fus, l1 = _run_steps(True, False, torch.bfloat16, 2)
    assert l0 == l1 and torch.equal(fus.flat_w, ref.flat_w) and torch.equal(fus.flat_w_lp, ref.flat_w_lp)
    for n in ("fc6", "fc7"):
        o, cnt = fus.woff[n]
        assert float(fus.flat_gw[o:o + cnt].abs().max()) == 0.0


def _run_steps(fused, keep, precision, nsteps, B=2, size=96, **kw):
    E, K = 20, 21
    m = models.FCN32s(E).load_synthetic(1337).cuda().eval()
    emb = synth.make_embeddings(K, E)
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-4, precision=precision, fused_head=True, fused_adam=fused, keep_grads=keep, **kw)
    assert ts.fused_adam == fused
    x = torch.from_numpy(synth.make_images(B, size, size, seed=5)).cuda()
    t = torch.from_numpy(synth.make_labels(B, size, size, K, seed=6, block=8)).cuda()
    losses = []
    for _ in range(nsteps):
        loss, _ = ts.step(x, t)
        losses.append(float(loss))
    torch.cuda.synchronize()
    return ts, losses


@pytest.mark.parametrize("precision,kw", [(torch.bfloat16, {}), (torch.float16, dict(dynamic_loss_scale=False, loss_scale=1024.0))])
def test_trainstep_fused_adam_is_bit_identical(precision, kw):
    ref, l0 = _run_steps(False, True, precision, 3, **kw)
    fus, l1 = _run_steps(True, True, precision, 3, **kw)
    assert l0 == l1
    for name in ("flat_w", "flat_b", "flat_w_lp", "flat_gw", "flat_gb"):
        assert torch.equal(getattr(fus, name), getattr(ref, name)), name
    for key in ("w", "b"):
        for u, v in zip(fus.state[key], ref.state[key]):
            assert torch.equal(u, v)
    # the weights really moved, in fc6 too
    o, cnt = ref.woff["fc6"]
    fresh = models.FCN32s(20).load_synthetic(1337)
    w0 = fresh.fc6.weight.detach().permute(0, 2, 3, 1).reshape(-1).cuda()
    assert float((ref.flat_w[o:o + cnt] - w0).abs().max()) > 0
    # gradients of the fused layers not stored: same weights, and their slots of the flat gradient are never written
    nog, l2 = _run_steps(True, False, precision, 3, **kw)
    assert l2 == l0
    assert torch.equal(nog.flat_w, ref.flat_w) and torch.equal(nog.flat_w_lp, ref.flat_w_lp)
    o, cnt = nog.woff["fc6"]                                      # (fc6 is the layer TrainStep fuses by default)
    assert float(nog.flat_gw[o:o + cnt].abs().max()) == 0.0
    o, cnt = nog.woff["conv5_3"]
    assert torch.equal(nog.flat_gw[o:o + cnt], ref.flat_gw[o:o + cnt])


def test_fused_adam_is_off_where_the_gradient_is_not_final():
    """dynamic loss scaling (the update may have to be skipped), fp32 (no 16-bit kernel), SGD: the separate pass"""
    E, K = 20, 21
    emb = synth.make_embeddings(K, E)
    for kw in (dict(precision=torch.float16), dict(precision=torch.float32), dict(precision=torch.bfloat16, optimizer="sgd"),
               dict(precision=torch.bfloat16, force_comm=False, fused_adam=False)):
        m = models.FCN32s(E).load_synthetic(1337).cuda().eval()
        ts = engine.TrainStep(m, emb, **kw)
        assert not ts.fused_adam


def test_cu_masked_stream_entry_points():
    """szn_stream_create_cu_mask / szn_stream_destroy: a kernel runs on the masked stream; bad arguments are refused"""
    info = L.DeviceInfo()
    L.call("szn_device_info", 0, C.byref(info))
    words = (info.compute_units + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(info.compute_units // 2):
        mask[cu // 32] |= 1 << (cu % 32)
    h = C.c_void_p()
    L.call("szn_stream_create_cu_mask", words, mask, C.byref(h))
    assert h.value
    x = torch.randn(1 << 20, device="cuda")
    y = torch.empty(1 << 20, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    L.call("szn_cast", L.SZN_F32, L.SZN_BF16, x.numel(), L.ptr(x), L.ptr(y), h)
    s = torch.cuda.ExternalStream(h.value)
    s.synchronize()
    assert torch.equal(y, x.to(torch.bfloat16))
    del s
    L.call("szn_stream_destroy", h)
    empty = (C.c_uint32 * words)()
    with pytest.raises(L.SznError):
        L.call("szn_stream_create_cu_mask", words, empty, C.byref(h))
    with pytest.raises(L.SznError):
        L.call("szn_stream_create_cu_mask", 0, mask, C.byref(h))
    with pytest.raises(L.SznError):
        L.call("szn_stream_destroy", None)


@pytest.mark.parametrize("caller", ["null", "own"])
def test_small_step_on_a_cu_masked_stream_is_bit_identical(caller, monkeypatch):
    """SZN_FC6_CUMASK: fc6's weight gradient + Adam of a small step on a stream confined to half of the CUs -- same kernels, same values,
    whether the caller sits on the null stream (the step moves to a stream of its own) or on a non-blocking one"""
    monkeypatch.setenv("SZN_FC6_CUMASK", "0")
    ref, l0 = _run_steps(True, False, torch.bfloat16, 3)
    monkeypatch.setenv("SZN_FC6_CUMASK", "128:low")
    assert models.masked_stream_wanted(2 * 96 * 96) and not models.masked_stream_wanted(8 * 512 * 512)
    if caller == "own":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            msk, l1 = _run_steps(True, False, torch.bfloat16, 3)
        torch.cuda.current_stream().wait_stream(side)
        assert msk._own_stream is None
    else:
        msk, l1 = _run_steps(True, False, torch.bfloat16, 3)
        assert msk._own_stream is not None
    assert msk.eng._wg_masked                                     # the masked stream was made and used
    assert l0 == l1
    for name in ("flat_w", "flat_b", "flat_w_lp"):
        assert torch.equal(getattr(msk, name), getattr(ref, name)), name
    for key in ("w", "b"):
        for u, v in zip(msk.state[key], ref.state[key]):
            assert torch.equal(u, v)
