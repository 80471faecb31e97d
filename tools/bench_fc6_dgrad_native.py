#!/usr/bin/env python
"""fc6's input gradient on the forward weight layout (szn_conv2d_dgrad_gemm_native) at the bench shape: ms per call, alone on the
device, and a checksum (SZN_WGW_XCD=0: tiles in the old order)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
B, Hi, Ci, Co, K = 8, 23, 512, 4096, 7
Ho = Hi - K + 1
code = L.dtype_code(torch.bfloat16)
g = torch.Generator().manual_seed(5)
dout = torch.randn(B, Ho, Ho, Co, generator=g).cuda().bfloat16()
w = (torch.randn(Co, K, K, Ci, generator=g) / (K * K * Ci) ** 0.5).cuda().bfloat16()
din = torch.empty(B, Hi, Hi, Ci, device="cuda", dtype=torch.bfloat16)
d = L.ConvDesc(code, B, Hi, Hi, Ci, Ho, Ho, Co, K, K, 0, Ci, Co, 0, 0, 0)
lib = L.load()
assert lib.szn_conv2d_dgrad_gemm_native_supported(C.byref(d)) == 1
ws = torch.empty(lib.szn_conv2d_dgrad_gemm_native_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
fn = lambda: L.call("szn_conv2d_dgrad_gemm_native", C.byref(d), L.ptr(dout), L.ptr(w), L.ptr(din), L.stream_ptr())
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print("fc6 dgrad (pack + GEMM + col2im): %.1f us per call, |din| = %.6e" % (e0.elapsed_time(e1) / 20 * 1e3, float(din.double().abs().sum())))
