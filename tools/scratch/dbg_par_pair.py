"""debug: cosa_par_forward (plane-pair step kernel) against the C oracle, small sizes, mismatch pattern per plane"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cosa_amd import _C
from oracle import c_oracle
L = _C.lib()
dev = torch.device("cuda:0")
dil = [1, 2, 4, 8, 12, 24]
cd = (ctypes.c_int * 6)(*dil)
rng = np.random.default_rng(0)
for (B, K, h, w, T) in [(1, 1, 40, 56, 1), (1, 2, 40, 56, 1), (1, 3, 64, 64, 1), (2, 5, 56, 72, 2), (1, 4, 224, 224, 10)]:
    img = rng.random((B, 3, h, w), dtype=np.float32)
    m = rng.random((B, K, h, w), dtype=np.float32)
    ref = np.stack([c_oracle.par_forward(img[b], m[b], dil, T) for b in range(B)])
    ti, tm = torch.from_numpy(img).to(dev), torch.from_numpy(m).to(dev)
    out = torch.empty_like(tm)
    ws = torch.empty(L.cosa_par_workspace_bytes(B, K, h, w, 6), dtype=torch.uint8, device=dev)
    _C.check(L.cosa_par_forward(_C.ptr(ti), _C.ptr(tm), _C.ptr(out), B, K, h, w, cd, 6, T, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "par")
    o = out.cpu().numpy()
    bad = o != ref
    print(f"B={B} K={K} {h}x{w} T={T}: mismatches {bad.sum()} of {bad.size}; per plane {[int(bad[b, k].sum()) for b in range(B) for k in range(K)]}; max abs diff {np.abs(o - ref).max():.3e}", flush=True)
    if bad.any() and T == 1:
        b, k = np.argwhere(bad.reshape(B, K, -1).any(-1))[0]
        ys, xs = np.nonzero(bad[b, k])
        print("   first plane with errors:", b, k, "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "sample", o[b, k, ys[0], xs[0]], ref[b, k, ys[0], xs[0]])
print("---- identical planes / zero second plane (T = 1)")
B, K, h, w, T = 1, 2, 40, 56, 1
img = rng.random((B, 3, h, w), dtype=np.float32)
for mode in ("same", "zero1", "zero0"):
    m = rng.random((B, K, h, w), dtype=np.float32)
    if mode == "same": m[:, 1] = m[:, 0]
    if mode == "zero1": m[:, 1] = 0
    if mode == "zero0": m[:, 0] = 0
    ref = np.stack([c_oracle.par_forward(img[b], m[b], dil, T) for b in range(B)])
    ti, tm = torch.from_numpy(img).to(dev), torch.from_numpy(m).to(dev)
    out = torch.empty_like(tm)
    ws = torch.empty(L.cosa_par_workspace_bytes(B, K, h, w, 6), dtype=torch.uint8, device=dev)
    _C.check(L.cosa_par_forward(_C.ptr(ti), _C.ptr(tm), _C.ptr(out), B, K, h, w, cd, 6, T, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "par")
    o = out.cpu().numpy()
    print(mode, "plane0 ok", np.array_equal(o[:, 0], ref[:, 0]), "plane1 ok", np.array_equal(o[:, 1], ref[:, 1]), "plane1 == plane0 out", np.array_equal(o[:, 1], o[:, 0]),
          "plane1 sample", o[0, 1, 20, 20:23], "ref", ref[0, 1, 20, 20:23], "ref0", ref[0, 0, 20, 20:23])
