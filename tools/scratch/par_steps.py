"""per-step time of the PAR propagation against the plane count (cosa_par_forward, b = 16, 224^2): (T = 20) - (T = 10) over 10"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cosa_amd import _C
L = _C.lib()
dev = torch.device("cuda:0")
B, h, w = 16, 224, 224
import ctypes
dil = (ctypes.c_int * 6)(1, 2, 4, 8, 12, 24)
img = torch.rand(B, 3, h, w, device=dev)
def run(K, T):
    m = torch.rand(B, K, h, w, device=dev)
    out = torch.empty_like(m)
    ws = torch.empty(L.cosa_par_workspace_bytes(B, K, h, w, 6), dtype=torch.uint8, device=dev)
    f = lambda: _C.check(L.cosa_par_forward(_C.ptr(img), _C.ptr(m), _C.ptr(out), B, K, h, w, dil, 6, T, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "par")
    f(); f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / 10 * 1000
for K in (2, 4, 8, 12, 16):
    t10, t20 = run(K, 10), run(K, 20)
    print(f"K={K:2d}: T=10 {t10:7.1f} us  T=20 {t20:7.1f} us  per step {(t20 - t10) / 10:6.2f} us  affinity+rest {t10 - (t20 - t10):6.1f} us", flush=True)
