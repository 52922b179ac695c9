"""one launch shape of the weight-gradient GEMM for rocprofv3 --pmc passes (student fc1: M=12560 tokens, N=3072, K=768)"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
M, N, K = 12560, 3072, 768
dy = torch.randn(M, N, device='cuda').bfloat16(); x = torch.randn(M, K, device='cuda').bfloat16()
for _ in range(5): nn_ops.gemm_wgrad(dy, x)
torch.cuda.synchronize()
