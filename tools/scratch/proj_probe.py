import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zeroshotsemanticsegmentation_amd import _lib as L
L.load()
def run(H, W, relu, iters=10, N=300, K=4096, out32=False):
    M = H * W
    dt = torch.bfloat16; code = L.dtype_code(dt); ldo = 304
    x = torch.randn(1, H, W, K, device="cuda")
    if relu: x = torch.relu(x)
    x = x.to(dt)
    w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(1, H, W, ldo, device="cuda", dtype=torch.float32 if out32 else dt)
    d = L.ConvDesc(code, 1, H, W, K, H, W, N, 1, 1, 0, K, ldo, 0, 0, int(out32))
    st = L.stream_ptr()
    fn = lambda: L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), st)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * K * N / ms / 1e9, L.last_kernel()
print("env", {k: v for k, v in os.environ.items() if k.startswith("SZN_")})
for H, W in ((256, 256), (512, 512), (1024, 512)):
    for relu in (0, 1):
        ms, tf, k = run(H, W, relu)
        print("M=%d relu=%d: %.4f ms %.1f TF/s (%.3f of 2.5 PF) A-stream %.2f TB/s %s" % (H * W, relu, ms, tf, tf / 2500, H * W * 4096 * 2 / ms / 1e9, k), flush=True)
