"""Parity at BASELINE's full sizes (512x512, E = 300) on the MI355X.

The oracle finishes a full-size forward in seconds, so (a) compares the fp32 HIP forward against it directly.  The
rest are size-independent properties that must hold bit-exactly, run on the real BASELINE layer shapes so that every
specialised kernel (conv3x3_regw, conv_igemm_wide, conv3x3_wide_rows, conv_wgrad_taps, conv_wgrad_wide, split-K) is exercised where the
bench exercises it:
  (b) scaling by a power of two commutes with every conv kernel (bf16 operands, fp32 accumulation, bf16 rounding:
      conv(2x) == 2 conv(x), dgrad(2 dout) == 2 dgrad(dout), wgrad(x, 2 dout) == 2 wgrad(x, dout)), and the result of a
      second identical launch is bit-identical (determinism);
  (c) batch-permutation equivariance of the whole network forward (tiles straddle image boundaries);
  (d) three full train steps (bf16, B = 2) on a fixed batch: finite, decreasing loss; the confusion histogram counts
      exactly the labelled pixels.
"""
import ctypes as C
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

G = os.path.join(ROOT, "tests", "golden")
H = 512


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_fullsize_fp32_forward_matches_oracle():
    E = 300
    emb = np.load(os.path.join(G, "embeddings_pascal_300.npy"))
    m = models.FCN32s(E).load_synthetic(1337).cuda().eval()
    x = synth.make_images(1, H, H, seed=7)
    with torch.no_grad():
        f = m(cu(x), mode="fcn")
    params = {k: v.detach().cpu().numpy() for k, v in m.named_parameters()
              if k.split(".")[0] not in ("upscore", "seenmask_upscore")}
    ref = O.FCN32sOracle(params, E).forward(x, "fcn")
    got = f.cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-3, err                                   # north-star tolerance (fp32 relative)
    # per-pixel class assignment: bit-exact against the oracle run on the SAME score; equal to the oracle's own
    # assignment wherever its top-2 margin exceeds 1e-5
    pred = utils.infer_lbl(f, emb, cuda=True)
    assert np.array_equal(pred, O.infer_lbl(got, emb))
    # ... and equal to the oracle's own assignment wherever the top-2 cosine margin exceeds 1e-5 (numpy, float64)
    pref = O.infer_lbl(ref, emb)
    sc = ref[0].reshape(E, -1).T.astype(np.float64)
    en = np.linalg.norm(emb.astype(np.float64), axis=1); en[en == 0] = 1.0
    sim = sc @ emb.astype(np.float64).T / (np.linalg.norm(sc, axis=1, keepdims=True) * en[None, :])
    top2 = np.sort(sim, axis=1)[:, -2:]
    clear = ((top2[:, 1] - top2[:, 0]) > 1e-5).reshape(pref.shape[1:])[None]
    assert clear.mean() > 0.99
    assert np.array_equal(pred[clear], pref[clear])


# name: (Hi, Ci, Co, K, pad) at 512x512 input (the bench's layers), expected forward kernel in bf16
LAYERS = {   # (shape), batch, expected forward kernel (None: a split-K epilogue kernel runs last at this batch)
    "conv1_2": ((710, 64, 64, 3, 1), 2, "conv3x3_regw"), "conv2_1": ((355, 64, 128, 3, 1), 2, "conv3x3_regw"),
    "conv2_2": ((355, 128, 128, 3, 1), 2, "conv3x3_regw"), "conv3_2": ((178, 256, 256, 3, 1), 4, "conv_igemm_8ph"),
    "conv4_2": ((89, 512, 512, 3, 1), 8, "conv_igemm_8ph"), "conv5_1": ((45, 512, 512, 3, 1), 8, "conv_igemm_v2"),
    "fc6": ((23, 512, 4096, 7, 0), 8, None), "fc7": ((17, 4096, 4096, 1, 0), 8, "conv_igemm_wide"),      # 256 x 192 tiles: one round of 220
}
WGRAD_KERNEL = {"conv1_2": "wgrad_taps_reduce", "conv2_1": "wgrad_taps_reduce", "conv2_2": "wgrad_taps_reduce",
                "conv3_2": "wgrad_taps_reduce", "conv4_2": "wgrad_taps_reduce", "conv5_1": "wgrad_taps_reduce",
                "fc6": "conv_wgrad_wide", "fc7": "conv_wgrad_wide"}


@pytest.mark.parametrize("name", list(LAYERS))
def test_fullsize_conv_scaling_and_determinism(name):
    (Hi, Ci, Co, K, pad), B, fwd_kernel = LAYERS[name]
    dt = torch.bfloat16
    code = L.dtype_code(dt)
    Ho = Hi + 2 * pad - K + 1
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(B, Hi, Hi, Ci, device="cuda", generator=g).to(dt)
    w = (torch.randn(Co, K, K, Ci, device="cuda", generator=g) / (Ci * K * K) ** 0.5).to(dt)
    bias = torch.randn(Co, device="cuda", generator=g)
    dout = torch.randn(B, Ho, Ho, Co, device="cuda", generator=g).to(dt)
    wT = torch.empty(Ci, K, K, Co, device="cuda", dtype=dt)
    st = L.stream_ptr()
    L.call("szn_pack_weight_dgrad", code, Co, K, K, Ci, L.ptr(w), L.ptr(wT), st)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    d = L.ConvDesc(code, B, Hi, Hi, Ci, Ho, Ho, Co, K, K, pad, Ci, Co, Ci, 1, 0)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()

    def fwd(xx, bb):
        out = torch.empty(B, Ho, Ho, Co, device="cuda", dtype=dt)
        L.call("szn_conv2d_fwd", C.byref(d), L.ptr(xx), L.ptr(w), L.ptr(bb), None, None, L.ptr(out), st)
        return out

    def dgrad(dd):
        din = torch.empty(B, Hi, Hi, Ci, device="cuda", dtype=dt)
        L.call("szn_conv2d_dgrad", C.byref(d), L.ptr(dd), L.ptr(wT), L.ptr(x), None, L.ptr(din), st)
        return din

    def wgrad(dd):
        dw = torch.empty(Co, K, K, Ci, device="cuda")
        L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dd), L.ptr(dw), 0, st)
        return dw

    o1 = fwd(x, bias)
    if fwd_kernel:
        assert L.last_kernel() == fwd_kernel, L.last_kernel()
    assert torch.equal(o1, fwd(x, bias)), "forward not deterministic"
    assert torch.equal(fwd(x * 2, bias * 2).float(), o1.float() * 2), "forward does not commute with x2"
    assert float(o1.float().abs().max()) > 0

    g1 = dgrad(dout)
    assert torch.equal(g1, dgrad(dout)), "dgrad not deterministic"
    assert torch.equal(dgrad(dout * 2).float(), g1.float() * 2), "dgrad does not commute with x2"

    w1 = wgrad(dout)
    assert L.last_kernel() == WGRAD_KERNEL[name], L.last_kernel()
    assert torch.equal(w1, wgrad(dout)), "wgrad not deterministic"
    assert torch.equal(wgrad(dout * 2), w1 * 2), "wgrad does not commute with x2"
    assert torch.isfinite(w1).all() and float(w1.abs().max()) > 0


def test_fullsize_fc6_forward_parity_split_k():
    """fc6 at the bench's batch (2312 pixels x 4096 couts x 25088): 1.25 waves of tiles unsplit, so the library runs it
    as a deterministic split-K on the 256 x 256 tile kernel; parity against torch's fp32 convolution of the same
    bf16-rounded operands"""
    import torch.nn.functional as F
    B, Hi, Ci, Co, K = 8, 23, 512, 4096, 7
    dt = torch.bfloat16
    code = L.dtype_code(dt)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, Hi, Hi, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, K, K, generator=g) / (Ci * K * K) ** 0.5).bfloat16().float()
    bias = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w, bias))
    Ho = Hi - K + 1
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    bd = bias.cuda()
    out = torch.empty(B, Ho, Ho, Co, device="cuda", dtype=dt)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    d = L.ConvDesc(code, B, Hi, Hi, Ci, Ho, Ho, Co, K, K, 0, Ci, Co, 0, 1, 0)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    L.call("szn_conv2d_fwd", C.byref(d), L.ptr(xd), L.ptr(wd), L.ptr(bd), None, None, L.ptr(out), L.stream_ptr())
    assert L.last_kernel() == "splitk_epilogue" and L.prev_kernel() == "conv_igemm_8ph", (L.prev_kernel(), L.last_kernel())
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err


def test_fullsize_forward_batch_permutation():
    E = 300
    m = models.FCN32s(E).load_synthetic(1337).cuda().eval()
    m.set_precision(torch.bfloat16)
    x = cu(synth.make_images(3, H, H, seed=5))
    perm = [2, 0, 1]
    with torch.no_grad():
        m(x, mode="fcn")
        c0 = m._last_ctx.coarse.clone()
        m(x[perm].contiguous(), mode="fcn")
        c1 = m._last_ctx.coarse.clone()
    assert torch.equal(c0[perm], c1)


def test_fullsize_train_steps_bf16():
    E = 300
    emb = np.load(os.path.join(G, "embeddings_pascal_300.npy"))
    K = emb.shape[0]
    m = models.FCN32s(E).load_synthetic(1337).cuda().train()
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.bfloat16, fused_head=True)
    B = 2
    x = cu(synth.make_images(B, H, H, seed=21))
    t = cu(synth.make_labels(B, H, H, K, seed=22))
    ts.hist.zero_()
    losses = [float(ts.step(x, t)[0]) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    assert losses[2] < losses[0], losses
    labelled = int((t >= 0).sum().item())
    assert int(ts.hist[0].sum().item()) == 3 * labelled      # every labelled pixel counted once per step
