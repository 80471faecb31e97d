"""The 16-bit wire image of the data-parallel gradient exchange at the C-ABI (round 5): szn_conv_desc_t.dw_lp and szn_*_step_g16.

The reference is single-GPU (trainer_fcn.py:157-158: loss.backward(); optim.step()); under data parallelism the build sums the
weight gradients of the ranks over RCCL.  With a bf16 wire the kernel that owns the final store of a gradient element rounds it
ONCE from its fp32 sum and writes 2 B (no fp32 gradient, no staging copy), and the optimizer kernel widens it again.  Checked here:
the image equals the fp32 gradient of the same call rounded by torch (bit for bit: same fp32 sums, same round-to-nearest-even), per
kernel family, and the g16 optimizer steps equal the fp32 steps on the widened gradient (bit for bit)."""
import ctypes as C
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import _lib as L  # noqa: E402

# (B, Hi, Ci, Co, k, pad, workspace?, kernel that must take it)
CASES = [(2, 128, 128, 128, 3, 1, True, "wgrad_taps_reduce"),         # every 3x3 layer: conv_wgrad_taps + its fixed-order reduce
         (1, 262, 64, 64, 3, 1, True, "wgrad_taps_reduce"),
         (3, 5, 4096, 4096, 1, 0, False, "conv_wgrad_wide"),           # fc7
         (2, 9, 512, 4096, 7, 0, False, "conv_wgrad_wide"),            # fc6
         (2, 17, 4096, 320, 1, 0, True, None),                         # the projection head: conv_wgrad_v2 + slabs, converted afterwards
         (1, 12, 64, 24, 3, 1, False, None)]                           # a layer no specialised kernel takes


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("wire", [torch.bfloat16, torch.float16])
def test_wgrad_wire_image_equals_rounded_fp32_gradient(case, wire):
    B, Hi, Ci, Co, k, pad, with_ws, kern = case
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5 + Hi)
    Ho = Hi + 2 * pad - k + 1
    x = torch.relu(torch.randn(B, Hi, Hi, Ci, device="cuda", generator=g)).to(dt)
    dout = (torch.randn(B, Ho, Ho, Co, device="cuda", generator=g) * 1e-2).to(dt)
    n = Co * k * k * Ci
    ws = torch.empty(2 * 256 * 64 * 9 * 64 * 4, dtype=torch.uint8, device="cuda") if with_ws else None

    def desc():
        d = L.ConvDesc(L.SZN_BF16, B, Hi, Hi, Ci, Ho, Ho, Co, k, k, pad, Ci, Co, 0, 0, 0)
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        return d
    st = L.stream_ptr()
    dw = torch.empty(n, device="cuda")
    d = desc()
    L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(dw), 0, st)
    if kern:
        assert L.last_kernel() == kern
    img = torch.full((n,), 3.0, device="cuda", dtype=wire)
    scratch = torch.full((n,), 7.0, device="cuda")
    d = desc()
    d.dw_lp, d.dw_lp_dtype = img.data_ptr(), L.dtype_code(wire)
    L.call("szn_conv2d_wgrad", C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(scratch), 0, st)
    torch.cuda.synchronize()
    assert torch.equal(img.view(torch.int16), dw.to(wire).view(torch.int16))
    if kern:        # the specialised kernels never touch the fp32 scratch on this path
        assert float(scratch.min()) == 7.0 and float(scratch.max()) == 7.0
    # accumulate + wire image is refused, nothing launched
    d = desc()
    d.dw_lp, d.dw_lp_dtype = img.data_ptr(), L.dtype_code(wire)
    assert L.load().szn_conv2d_wgrad(C.byref(d), L.ptr(x), L.ptr(dout), L.ptr(scratch), 1, st) != 0


@pytest.mark.parametrize("n", [4 * 1000 + 3, 1 << 20])
@pytest.mark.parametrize("wire", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lp", [None, torch.bfloat16])
def test_optimizer_steps_on_a_16bit_gradient_image(n, wire, lp):
    g = torch.Generator(device="cuda").manual_seed(n % 1000)
    p0 = torch.randn(n, device="cuda", generator=g) * 0.02
    g16 = (torch.randn(n, device="cuda", generator=g) * 1e-3).to(wire)
    g32 = g16.float()
    m0 = torch.randn(n, device="cuda", generator=g) * 1e-4
    v0 = torch.rand(n, device="cuda", generator=g) * 1e-6
    st = L.stream_ptr()
    lpc = L.dtype_code(lp) if lp is not None else 0
    # Adam
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    ia = torch.zeros(n, device="cuda", dtype=lp) if lp is not None else None
    L.call("szn_adam_step", n, L.ptr(pa), L.ptr(g32), L.ptr(ma), L.ptr(va), 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, 0.5, L.ptr(ia), lpc, st)
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    ib = torch.zeros(n, device="cuda", dtype=lp) if lp is not None else None
    L.call("szn_adam_step_g16", n, L.ptr(pb), L.ptr(g16), L.dtype_code(wire), L.ptr(mb), L.ptr(vb), 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, 0.5,
           L.ptr(ib), lpc, st)
    assert L.last_kernel() == "adam_kernel_g16"
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    if lp is not None:
        assert torch.equal(ia.view(torch.int16), ib.view(torch.int16))
    # SGD with momentum (first and later steps)
    for first in (1, 0):
        pa, ba = p0.clone(), m0.clone()
        L.call("szn_sgd_momentum_step", n, L.ptr(pa), L.ptr(g32), L.ptr(ba), 1e-2, 0.99, 5e-4, first, 0.5, L.ptr(ia), lpc, st)
        pb, bb = p0.clone(), m0.clone()
        L.call("szn_sgd_momentum_step_g16", n, L.ptr(pb), L.ptr(g16), L.dtype_code(wire), L.ptr(bb), 1e-2, 0.99, 5e-4, first, 0.5,
               L.ptr(ib), lpc, st)
        assert L.last_kernel() == "sgd_kernel_g16"
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ba, bb)
    # an fp32 "image" is refused
    assert L.load().szn_adam_step_g16(n, L.ptr(pb), L.ptr(g32), L.SZN_F32, L.ptr(mb), L.ptr(vb), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None, 0,
                                      st) != 0
