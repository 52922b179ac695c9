import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cosa_amd import nn_ops
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
torch.manual_seed(0)
M = 12560
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
for nblk in (1, 3, 6, 12):
    pairs = [(torch.randn(M, N, device='cuda').bfloat16(), torch.randn(M, K, device='cuda').bfloat16(), True) for _ in range(nblk) for (N, K) in shapes]
    fl = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in pairs) / 1e12
    tb = timeit(lambda: nn_ops.gemm_wgrad_batched(pairs), 5)
    ts = timeit(lambda: [nn_ops.gemm_wgrad(p[0], p[1], want_bias=True) for p in pairs], 5)
    print(f"blocks {nblk:2d}: batched {tb*1e3:8.1f} us ({fl/tb*1e3:5.0f} TF/s)   per-linear launches {ts*1e3:8.1f} us ({fl/ts*1e3:5.0f} TF/s)", flush=True)
