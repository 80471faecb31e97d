import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from zeroshotsemanticsegmentation_amd import _lib as L
L.load()
def run(relu, warm, iters, M=262144, N=300, K=4096):
    dt = torch.bfloat16
    ldo = (N + 7) // 8 * 8
    x = torch.randn(1, 512, 512, K, device="cuda")
    if relu: x = torch.relu(x)
    x = x.to(dt)
    w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(1, 512, 512, ldo, device="cuda", dtype=dt)
    d = L.ConvDesc(L.SZN_BF16, 1, 512, 512, K, 512, 512, N, 1, 1, 0, K, ldo, 0, 0, 0)
    st = L.stream_ptr()
    fn = lambda: L.call("szn_conv2d_fwd", C.byref(d), L.ptr(x), L.ptr(w), L.ptr(bias), None, None, L.ptr(out), st)
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * K * N / ms / 1e9
for rep in range(2):
    for relu in (True, False):
        for warm, iters in ((3, 10), (30, 10), (100, 30)):
            ms, tf = run(relu, warm, iters)
            print("relu" if relu else "randn", "warm", warm, "iters", iters, "%.4f ms %.1f TF %.4f" % (ms, tf, tf / 2500), L.last_kernel())
