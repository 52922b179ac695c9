"""one launch shape of the projection GEMM for rocprofv3 --pmc passes: the teacher's fc1 + GELU (M = all tokens of a step)"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
M, N, K, epi = 87904, 3072, 768, 1
x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') * 0.03).bfloat16(); b = torch.randn(N, device='cuda').bfloat16()
for _ in range(5): y = nn_ops.gemm_bf16(x, w, b, epi)
torch.cuda.synchronize()
