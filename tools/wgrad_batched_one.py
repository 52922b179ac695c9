import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosa_amd import nn_ops
torch.manual_seed(0)
M = 12560
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
pairs = [(torch.randn(M, N, device='cuda').bfloat16(), torch.randn(M, K, device='cuda').bfloat16(), True) for _ in range(12) for (N, K) in shapes]
for _ in range(13):
    nn_ops.gemm_wgrad_batched(pairs)
torch.cuda.synchronize()
print(json.dumps({"iters": 13}))
