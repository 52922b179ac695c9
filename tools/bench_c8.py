"""Times the teacher's four projection shapes (M = 87 904 token rows of a step's six passes) in the plain fp16 kernel, the bf16x3 kernel and the
fp16c8 kernel (HIP events, 20 launches each after 3 warm-ups): algorithmic TFLOP/s (2 M N K / time) and the cost ratios."""
import sys
import torch
sys.path.insert(0, ".")
from cosa_amd import nn_ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 87904
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e-3


for name, N, K, epi in (("qkv", 2304, 768, 0), ("proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    x16, w16, b16 = x.half(), w.half(), b.half()
    o16 = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == 2 else torch.float16)
    t16 = timeit(lambda: nn_ops.gemm_bf16(x16, w16, b16, epi, residual=r, out=o16))
    xs, ws = nn_ops.split_rows(x, ones=True), nn_ops.split_rows(w, bias=b)
    o3 = torch.empty((M, N if epi == 2 else 2 * N + 64), device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    t3 = timeit(lambda: nn_ops.gemm_x3(xs, ws, M, N, K, epi, residual=r, out=o3, ldy=None if epi == 2 else 2 * N + 64))
    xc, wc = nn_ops.c8_rows(x, ones=True), nn_ops.c8_rows(w, bias=b)
    o8 = torch.empty((M, N if epi != 1 else 2 * N + 64), device="cuda", dtype=torch.float32 if epi == 2 else torch.float16)
    t8 = timeit(lambda: nn_ops.gemm_c8(xc, wc, M, N, K, epi, residual=r, out=o8))
    fl = 2.0 * M * N * K / 1e12
    print(f"{name:5s} M={M} N={N} K={K}: fp16 {t16 * 1e6:7.1f} us ({fl / t16:6.0f} TF/s)  bf16x3 {t3 * 1e6:7.1f} us ({fl / t3:5.0f}, x{t3 / t16:.2f})  "
          f"fp16c8 {t8 * 1e6:7.1f} us ({fl / t8:5.0f}, x{t8 / t16:.2f})", flush=True)
