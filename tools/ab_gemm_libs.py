"""A/B of two builds of libcosa_hip.so on the projection GEMMs of a training step, INTERLEAVED in one process (guide rule 24): the tree's
library against another build (default cosa_amd/lib/libcosa_hip_old.so, e.g. `git worktree add _old <commit> && (cd _old && python -m
cosa_amd.build) && cp _old/cosa_amd/lib/libcosa_hip.so cosa_amd/lib/libcosa_hip_old.so`).  Also checks that both give the same bits.
usage (GPU box): python tools/ab_gemm_libs.py [old.so] > gpurun_out/ab_gemm.txt"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd import _C, nn_ops

old_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(_C.LIB_PATH), "libcosa_hip_old.so")
new, old = _C.lib(), ctypes.CDLL(old_path)
P = ctypes.c_void_p
for L in (old,):
    for fn in ("cosa_gemm_bf16", "cosa_gemm_f16"):
        getattr(L, fn).argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    L.cosa_gemm_f16c4.argtypes = [P, P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    L.cosa_gemm_f16c8.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    L.cosa_gemm_bf16_dual_gelu.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    L.cosa_gemm_bf16x3.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
dev = torch.device("cuda", 0)
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timed_pair(fa, fb, rounds=6, n=10):
    """interleaved rounds: median and min of each arm in us"""
    ta, tb = [], []
    for f in (fa, fb):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for f, acc in ((fa, ta), (fb, tb)):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                f()
            e.record()
            torch.cuda.synchronize()
            acc.append(a.elapsed_time(e) / n * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    return med(ta), min(ta), med(tb), min(tb)


def report(name, M, N, K, fo, fn, outs_o, outs_n, tiles=None):
    for t in outs_o + outs_n:
        t.zero_()
    fo(); fn()
    torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(outs_o, outs_n))          # (bytes: c4 block planes read as fp16 hold NaN patterns)
    mo, no_, mn, nn_ = timed_pair(fo, fn)
    fl = 2.0 * M * N * K
    print(f"{name:26s} M={M:6d} N={N:4d} K={K:4d}  old {mo:7.1f} us (min {no_:7.1f}, {fl / mo / 1e6:5.0f} TF)   new {mn:7.1f} us (min {nn_:7.1f}, {fl / mn / 1e6:5.0f} TF)"
          f"   new/old {mn / mo:.3f}   bits {'equal' if same else 'DIFFER'}", flush=True)


st = _C.stream_ptr
for M in (87904, 12560):
    for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        for epi, en in ((0, "bias"), (1, "gelu"), (2, "res")):
            if (epi == 1) != (name == "fc1") or (epi == 2) != (name in ("proj", "fc2")):
                continue
            od = torch.float32 if epi == 2 else torch.bfloat16
            yo, yn = torch.zeros(M, N, device=dev, dtype=od), torch.zeros(M, N, device=dev, dtype=od)
            r = torch.randn(M, N, device=dev) if epi == 2 else None
            report(f"bf16 {name} {en}", M, N, K,
                   lambda: old.cosa_gemm_bf16(ptr(x), ptr(w), ptr(b), ptr(r), ptr(yo), M, N, K, epi, st()),
                   lambda: new.cosa_gemm_bf16(ptr(x), ptr(w), ptr(b), ptr(r), ptr(yn), M, N, K, epi, st()), [yo], [yn])
        if name == "fc1" and M == 12560:
            ho, ao, hn, an = (torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(4))
            report("bf16 fc1 dual", M, N, K,
                   lambda: old.cosa_gemm_bf16_dual_gelu(ptr(x), ptr(w), ptr(b), ptr(ho), ptr(ao), M, N, K, st()),
                   lambda: new.cosa_gemm_bf16_dual_gelu(ptr(x), ptr(w), ptr(b), ptr(hn), ptr(an), M, N, K, st()), [ho, ao], [hn, an])
# the teacher's corrected projections: fp16c4 qkv (fp16 out), fc1 (GELU -> c4 rows), fc2 (fp32 residual, in place), fp16c8 proj
M = 87904
z = torch.zeros(8192, device=dev, dtype=torch.float16)
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    xs, xsc = nn_ops.c4_rows(x, ones=True)
    ws, wsc = nn_ops.c4_rows(w, bias=b, weight=True)
    if epi == 2:
        r = torch.randn(M, N, device=dev)
        yo, yn = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
        so = sn = None
        ldy = N
    elif epi == 1:
        ldy = nn_ops.split_ld(N)
        yo, yn = (torch.zeros((M, ldy), device=dev, dtype=torch.float16) for _ in range(2))
        so, sn = nn_ops.c4_scales(M, N, dev), nn_ops.c4_scales(M, N, dev)
        r = None
    else:
        ldy = N
        yo, yn = (torch.zeros((M, N), device=dev, dtype=torch.float16) for _ in range(2))
        so = sn = r = None
    report(f"fp16c4 {name}", M, N, K,
           lambda: old.cosa_gemm_f16c4(ptr(xs), ptr(xsc), ptr(ws), ptr(wsc), ptr(z), ptr(r), ptr(yo), ptr(so), M, N, K, epi, ldy, st()),
           lambda: new.cosa_gemm_f16c4(ptr(xs), ptr(xsc), ptr(ws), ptr(wsc), ptr(z), ptr(r), ptr(yn), ptr(sn), M, N, K, epi, ldy, st()),
           [yo] + ([so] if so is not None else []), [yn] + ([sn] if sn is not None else []))
x = torch.randn(M, 768, device=dev); w = torch.randn(768, 768, device=dev) * 0.03; b = torch.randn(768, device=dev)
xs, ws = nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b)
r = torch.randn(M, 768, device=dev)
yo, yn = torch.zeros(M, 768, device=dev), torch.zeros(M, 768, device=dev)
report("fp16c8 proj res", M, 768, 768,
       lambda: old.cosa_gemm_f16c8(ptr(xs), ptr(ws), ptr(z), ptr(r), ptr(yo), M, 768, 768, 2, 768, st()),
       lambda: new.cosa_gemm_f16c8(ptr(xs), ptr(ws), ptr(z), ptr(r), ptr(yn), M, 768, 768, 2, 768, st()), [yo], [yn])

# fp16c8 (blocks 2-11 of the default teacher): qkv (fp16 out), fc1 + GELU (c8 rows out), fc2 + fp32 residual
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    xs, ws = nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b)
    if epi == 2:
        r, ldy = torch.randn(M, N, device=dev), N
        yo, yn = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
    else:
        r, ldy = None, (nn_ops.split_ld(N) if epi == 1 else N)
        yo, yn = (torch.zeros((M, ldy), device=dev, dtype=torch.float16) for _ in range(2))
    report(f"fp16c8 {name}", M, N, K,
           lambda: old.cosa_gemm_f16c8(ptr(xs), ptr(ws), ptr(z), ptr(r), ptr(yo), M, N, K, epi, ldy, st()),
           lambda: new.cosa_gemm_f16c8(ptr(xs), ptr(ws), ptr(z), ptr(r), ptr(yn), M, N, K, epi, ldy, st()), [yo], [yn])

# bf16x3 (the first two blocks of the default teacher since round 5): qkv (split rows out), fc1 + GELU (split rows out), fc2 + fp32 residual
zb = torch.zeros(8192, device=dev, dtype=torch.bfloat16)
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    xs, ws = nn_ops.split_rows(x, ones=True), nn_ops.split_rows(w, bias=b)
    if epi == 2:
        r, ldy = torch.randn(M, N, device=dev), N
        yo, yn = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
    else:
        r, ldy = None, 2 * N + (64 if epi == 1 else 0)
        yo, yn = (torch.zeros((M, ldy), device=dev, dtype=torch.bfloat16) for _ in range(2))
    report(f"bf16x3 {name}", M, N, K,
           lambda: old.cosa_gemm_bf16x3(ptr(xs), ptr(ws), ptr(zb), ptr(r), ptr(yo), M, N, K, epi, ldy, st()),
           lambda: new.cosa_gemm_bf16x3(ptr(xs), ptr(ws), ptr(zb), ptr(r), ptr(yn), M, N, K, epi, ldy, st()), [yo], [yn])
