#!/usr/bin/env python
"""fc6 dgrad as Y = dout x W with W in its NATIVE [co][kh*kw*ci] layout: conv_wgrad_wide computes D[a][b] = sum_k A[k][a] B[k][b] with
both operands K-major, so Y[m][n] = sum_co doutT[co][m] W[co][n] is a '1x1 wgrad' with 4096 'pixels' (= couts of fc6), A = dout^T,
B = the forward weight image.  No transposed copy of the 205 MB filter bank per step (pack_dgrad16_batch), only of the 19 MB dout.
Compares with the shipped path (szn_conv2d_dgrad_gemm on the packed wG) and times both."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
B, Hi, Ci, Co, K = 8, 23, 512, 4096, 7
Ho = Hi - K + 1
M, N = B * Ho * Ho, K * K * Ci
dt = L.dtype_code(torch.bfloat16)
st = L.stream_ptr()
torch.manual_seed(1)
w = (torch.randn(Co, K, K, Ci, device="cuda") / (Ci * K * K) ** 0.5).bfloat16()
dout = torch.randn(B, Ho, Ho, Co, device="cuda").bfloat16()
# shipped: wG = plain transpose, GEMM (conv_igemm_wide, fp32 Y in the workspace) + col2im
wG = torch.empty(N, Co, device="cuda", dtype=torch.bfloat16)
L.call("szn_pack_weight_dgrad", dt, Co, 1, 1, N, L.ptr(w), L.ptr(wG), st)
d = L.ConvDesc(dt, B, Hi, Hi, Ci, Ho, Ho, Co, K, K, 0, Ci, Co, 0, 0, 0)
ws = torch.empty(L.load().szn_conv2d_dgrad_gemm_workspace_bytes(C.byref(d)), dtype=torch.uint8, device="cuda")
d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
din = torch.empty(B, Hi, Hi, Ci, device="cuda", dtype=torch.bfloat16)
ref = lambda: L.call("szn_conv2d_dgrad_gemm", C.byref(d), L.ptr(dout), L.ptr(wG), L.ptr(din), st)
ref(); torch.cuda.synchronize()
Yref = ws[:M * N * 4].view(torch.float32).view(M, N).clone()
# prototype: Y via szn_conv2d_wgrad on a 1x1 'layer' with Co_fc6 pixels
Y = torch.empty(M, N, device="cuda")
doutT = torch.empty(Co, M, device="cuda", dtype=torch.bfloat16)
dw_desc = L.ConvDesc(dt, 1, 1, Co, N, 1, Co, M, 1, 1, 0, N, M, 0, 0, 0)     # x: [Co px][N ch], dout: [Co px][M ch] -> dw [M][N]
wsg = torch.empty(2 * 256 * 64 * 9 * 64 * 4, dtype=torch.uint8, device="cuda")
dw_desc.workspace, dw_desc.workspace_bytes = wsg.data_ptr(), wsg.numel()
def proto():
    doutT.copy_(dout.view(M, Co).t())
    L.call("szn_conv2d_wgrad", C.byref(dw_desc), L.ptr(w), L.ptr(doutT), L.ptr(Y), 0, st)
proto(); torch.cuda.synchronize()
print("kernel:", L.last_kernel())
err = (Y - Yref).abs().max() / Yref.abs().max()
print("max |Y - Yref| / max |Yref| = %.3e   bit-identical: %s" % (float(err), bool(torch.equal(Y, Yref))))
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flops = 2.0 * M * N * Co
for rep in range(2):
    t_ref = t(ref); t_pro = t(proto)
    t_tr = t(lambda: doutT.copy_(dout.view(M, Co).t()))
    t_pack = t(lambda: L.call("szn_pack_weight_dgrad", dt, Co, 1, 1, N, L.ptr(w), L.ptr(wG), st))
    print("shipped GEMM + col2im %.3f ms | prototype (dout^T %.3f + wgrad-form GEMM, no col2im) %.3f ms = %.0f TF/s | pack of the filter bank %.3f ms"
          % (t_ref, t_tr, t_pro, flops / ((t_pro - t_tr) * 1e-3) / 1e12, t_pack))
