"""bit-level comparison of the device bilateral filter with the golden vectors and the C oracle (sorted splat vs the atomic one via
COSA_LATTICE_ATOMIC_SPLAT=1), and run-to-run determinism"""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from cosa_amd import _C
from oracle import c_oracle
c_oracle.build()
g = dict(np.load('tests/golden/bilateral.npz'))
L = _C.lib()
for tag in ("smooth", "noise", "odd"):
    img, seg, ref = torch.from_numpy(g[f"{tag}_img"]).cuda(), torch.from_numpy(g[f"{tag}_seg"]).cuda(), g[f"{tag}_out"]
    N, K, H, W = seg.shape
    outs = []
    for rep in range(2):
        out = torch.empty_like(seg)
        Ms = torch.zeros(N, dtype=torch.int32, device="cuda")
        ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), "cuda", "t")
        _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(img), _C.ptr(seg), _C.ptr(out), N, K, H, W, 15.0, 50.0, _C.ptr(Ms),
                                                  _C.ptr(ws), ws.numel(), _C.stream_ptr()))
        outs.append(out.cpu().numpy())
    o_ref, M_ref = c_oracle.bilateralfilter_batch(g[f"{tag}_img"], g[f"{tag}_seg"], N, K, H, W, 15.0, 50.0)
    o = outs[0]
    print(tag, (N, K, H, W), "M ok", np.array_equal(Ms.cpu().numpy(), M_ref), "| vs golden: equal", np.array_equal(o, ref), "max rel", float(np.max(np.abs(o - ref) / (np.abs(ref) + 1e-6))),
          "| vs oracle: equal", np.array_equal(o, o_ref.reshape(o.shape)), "| run-to-run equal", np.array_equal(outs[0], outs[1]))
