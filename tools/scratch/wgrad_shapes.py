import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cosa_amd import nn_ops
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
torch.manual_seed(0)
M = 12560
for name, (N, K), cnt in (("fc1", (3072, 768), 36), ("fc2", (768, 3072), 36), ("qkv", (2304, 768), 48), ("proj", (768, 768), 142)):
    pairs = [(torch.randn(M, N, device='cuda').bfloat16(), torch.randn(M, K, device='cuda').bfloat16(), False) for _ in range(cnt)]
    tiles = cnt * ((N + 255) // 256) * (K // 128)
    fl = cnt * 2.0 * M * N * K / 1e12
    t = timeit(lambda: nn_ops.gemm_wgrad_batched(pairs))
    print(f"{name:5s} x{cnt}: {tiles} tiles = {tiles/256:.2f} rounds  {t*1e3:8.1f} us  {fl/t*1e3:5.0f} TF/s  {t*1e3/(tiles/256):6.1f} us per round-equivalent", flush=True)
