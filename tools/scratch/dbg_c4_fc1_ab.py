"""where do two library builds differ on the fp16c4 fc1 launch (c4 rows + scales out)?"""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from cosa_amd import _C, nn_ops
new, old = _C.lib(), ctypes.CDLL(os.path.join(os.path.dirname(_C.LIB_PATH), "libcosa_hip_old.so"))
P = ctypes.c_void_p
old.cosa_gemm_f16c4.argtypes = [P, P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
dev = torch.device("cuda", 0)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 87904, 3072, 768
z = torch.zeros(8192, device=dev, dtype=torch.float16)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
xs, xsc = nn_ops.c4_rows(x, ones=True)
ws, wsc = nn_ops.c4_rows(w, bias=b, weight=True)
ldy = nn_ops.split_ld(N)
yo, yn = (torch.zeros((M, ldy), device=dev, dtype=torch.float16) for _ in range(2))
so, sn = nn_ops.c4_scales(M, N, dev), nn_ops.c4_scales(M, N, dev)
for rep in range(3):
    old.cosa_gemm_f16c4(ptr(xs), ptr(xsc), ptr(ws), ptr(wsc), ptr(z), None, ptr(yo), ptr(so), M, N, K, 1, ldy, _C.stream_ptr())
    new.cosa_gemm_f16c4(ptr(xs), ptr(xsc), ptr(ws), ptr(wsc), ptr(z), None, ptr(yn), ptr(sn), M, N, K, 1, ldy, _C.stream_ptr())
    torch.cuda.synchronize()
    d = (yo.view(torch.int16) != yn.view(torch.int16))
    ds = so != sn
    print("rep", rep, "rows differ:", int(d.any(1).sum()), "cols differ:", int(d.any(0).sum()), "scale bytes differ:", int(ds.sum()))
    if d.any():
        r = d.any(1).nonzero().flatten(); c = d.any(0).nonzero().flatten()
        print(" first rows", r[:8].tolist(), "last", r[-4:].tolist(), " cols range", int(c.min()), int(c.max()), "n", len(c))
    if ds.any():
        i = ds.nonzero().flatten()
        print(" scale idx", i[:8].tolist(), "...", i[-4:].tolist(), "old", so[i[:8]].tolist(), "new", sn[i[:8]].tolist())
    # run-to-run determinism of each arm
    y2 = torch.zeros_like(yn); s2 = torch.zeros_like(sn)
    new.cosa_gemm_f16c4(ptr(xs), ptr(xsc), ptr(ws), ptr(wsc), ptr(z), None, ptr(y2), ptr(s2), M, N, K, 1, ldy, _C.stream_ptr())
    torch.cuda.synchronize()
    print(" new vs new:", bool(torch.equal(y2, yn)), bool(torch.equal(s2, sn)))
