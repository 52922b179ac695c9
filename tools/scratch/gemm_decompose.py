"""Where does a job of the persistent GEMM spend its time?  (a) K sweep at the qkv shape: time per job = nk * t + s; (b) the timing ablations of
the epilogue (COSA_GEMM_VARIANT 61..65: no stores / no epilogue / L2-resident store window / stores dropped by the range check); run it
under COSA_GEMM_STAGGER=<ticks | 1 << 20> for the start-stagger experiment.  Needs the experiments build of the library:
COSA_EXTRA_FLAGS_GEMM_KERNELS=-DCOSA_GEMM_EXPERIMENTS=1 python -m cosa_amd.build --force"""
import os, sys, torch
sys.path.insert(0, ".")
from cosa_amd import nn_ops, _C

M, N = 87904, 2304
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
    return a.elapsed_time(e) / n * 1e3


jobs = ((M + 255) // 256) * (N // 256)
rounds = (jobs + 255) // 256
print(f"stagger env {os.environ.get('COSA_GEMM_STAGGER')}  jobs {jobs} rounds {rounds}")
res = {}
for K in (768, 1536, 3072):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: nn_ops.gemm_bf16(x, w, b, 0, out=o))
    res[K] = us
    print(f"K={K}: {us:.1f} us  per job {us / rounds:.2f} us  {2.0 * M * N * K / us / 1e6:.0f} TF/s", flush=True)
t = (res[3072] - res[768]) / rounds / 36
print(f"per K-tile t = {t:.3f} us, per-job fixed s = {res[768] / rounds - 12 * t:.2f} us (K=768), {res[3072] / rounds - 48 * t:.2f} (K=3072)")
K = 768
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for v, name in ((0, "default"), (61, "epilogue without stores"), (62, "no epilogue work"), (64, "L2-resident store window"), (65, "stores dropped (range check)"),
                (7, "nt stores"), (8, "nt+sc0+sc1 stores")):
    _C.lib().cosa_gemm_set_variant(v)
    us = timeit(lambda: nn_ops.gemm_bf16(x, w, b, 0, out=o))
    print(f"variant {v:2d} ({name}): {us:.1f} us  per job {us / rounds:.2f}", flush=True)
_C.lib().cosa_gemm_set_variant(0)
