"""A/B of builds of libcosa_hip.so on the teacher's attention launches (cosa_attn_fwd_f16c8: fp16 q, k, v, c8 rows out; B = 32, 12 heads,
N = 1765 / 785 / 197), INTERLEAVED in one process; the outputs of every build are compared byte for byte with the first one's.
usage (GPU box): python tools/ab_attn_libs.py a.so b.so [c.so ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd import _C
libs = [(os.path.basename(p).replace("libcosa_hip_", "").replace(".so", ""), ctypes.CDLL(os.path.abspath(p))) for p in sys.argv[1:]]
P, I = ctypes.c_void_p, ctypes.c_int
for _, L in libs:
    L.cosa_attn_fwd_f16c8.argtypes = [P, P, P, I, I, I, I, ctypes.c_float, P, P]
    L.cosa_attn_fwd_f16c8.restype = I
dev = torch.device("cuda", 0)
st = _C.stream_ptr
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
B, H = 32, 12
for N in (1765, 785, 197):
    g = torch.Generator(device="cpu").manual_seed(N)
    qkv = (torch.randn(B, N, 3 * H * 64, generator=g) * 1.5).to(dev).half()
    outs = [torch.zeros(B * N, 4 * H * 64 + 128, device=dev, dtype=torch.uint8) for _ in libs]
    timing = os.environ.get("ATTN_TIMING") == "1"          # builds of tools/attn_variants.py `timing`: per-section cycle sums in stamps[128 ..]
    stamp = [torch.zeros(256, device=dev, dtype=torch.int64) for _ in libs]
    fs = [(lambda L=L, o=o, sp=sp: L.cosa_attn_fwd_f16c8(ptr(qkv), ptr(o), None, B, N, H, 64, 0.125, ptr(sp) if timing else None, st())) for (_, L), o, sp in zip(libs, outs, stamp)]
    for f in fs:
        for _ in range(3):
            assert f() == 0
    torch.cuda.synchronize()
    same = [torch.equal(outs[0], o) for o in outs]
    # run-to-run determinism of every build, and where the builds differ
    for (name, L), f, o in zip(libs, fs, outs):
        keep = o.clone()
        o.zero_()
        f()
        torch.cuda.synchronize()
        if not torch.equal(keep, o):
            print(f"N={N:5d} {name}: NOT deterministic run to run ({int((keep != o).sum())} bytes differ)")
        if not torch.equal(outs[0], o):
            d = (outs[0] != o)
            rows = d.any(dim=1).nonzero().flatten()
            cols = d.any(dim=0).nonzero().flatten()
            print(f"N={N:5d} {name}: {int(d.sum())} bytes differ from the first build in {rows.numel()} rows (first {rows[:6].tolist()}, last {rows[-3:].tolist()}); columns {cols[:8].tolist()} .. {cols[-4:].tolist()}")
    ts = [[] for _ in libs]
    for _ in range(6):
        for f, acc in zip(fs, ts):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                f()
            e.record()
            torch.cuda.synchronize()
            acc.append(a.elapsed_time(e) / 10 * 1e3)
    if timing:
        for (name, _), f, sp in zip(libs, fs, stamp):
            sp.zero_()
            sp[:128:2] = torch.iinfo(torch.int64).max          # (the launch-span slots: min start / max end)
            f()
            torch.cuda.synchronize()
            v = sp[128:134].cpu().tolist()
            if v[5]:
                print(f"N={N:5d} {name}: cycles per wave and tile: dma issue {v[0] / v[5]:.0f} | K reads + QK MFMAs {v[1] / v[5]:.0f} | softmax (incl. waiting for the scores) {v[2] / v[5]:.0f} | V reads + PV MFMAs {v[3] / v[5]:.0f} | vmcnt + barrier {v[4] / v[5]:.0f} | total {sum(v[:5]) / v[5]:.0f}  ({v[5]} wave-tiles)")
    fl = 4.0 * N * N * 64 * B * H
    base = sorted(ts[0])[3]
    for (name, _), t, s in zip(libs, ts, same):
        med = sorted(t)[3]
        print(f"N={N:5d} {name:14s} {med:8.1f} us (min {min(t):8.1f}, {fl / med / 1e6:5.0f} TF)  vs first {med / base:.3f}  bits {'equal' if s else 'DIFFER'}", flush=True)
