import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotsemanticsegmentation_amd import _lib as L
dt = L.dtype_code(torch.bfloat16)
B, Hi, Ci, Co = 8, 64, 64, 64
torch.manual_seed(0)
x = torch.relu(torch.randn(B, Hi, Hi, Ci, device="cuda")).bfloat16()
w = (torch.randn(Co, 3, 3, Ci, device="cuda") / (Ci * 9) ** 0.5).bfloat16()
bias = torch.randn(Co, device="cuda")
out = torch.empty(B, Hi, Hi, Co, device="cuda", dtype=torch.bfloat16)
Hp = (Hi + 1) // 2
pool = torch.zeros(B, Hp, Hp, Co, device="cuda", dtype=torch.bfloat16)
d = L.ConvDesc(dt, B, Hi, Hi, Ci, Hi, Hi, Co, 3, 3, 1, Ci, Co, 0, 1, 0)
d.pool_out = pool.data_ptr()
L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), L.stream_ptr())
torch.cuda.synchronize()
print(L.last_kernel())
ref = torch.nn.functional.max_pool2d(out.float().permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
df = (pool.float() != ref)
print("mismatches", int(df.sum()), "of", df.numel())
idx = df.nonzero()[:6]
for i in idx:
    b, ph, pw, c = [int(t) for t in i]
    win = out[b, 2*ph:2*ph+2, 2*pw:2*pw+2, c].float().flatten().tolist()
    print((b, ph, pw, c), "window", win, "got", float(pool[b, ph, pw, c]), "ref", float(ref[b, ph, pw, c]))
