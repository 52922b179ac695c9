"""one launch shape of the fp16c8 projection GEMM for rocprofv3 --pmc passes.  default: the teacher's fc1 + GELU (M = all tokens of a step), c8 rows
in and out;  `fc2`: mlp.fc2 + fp32 residual (N = 768, K = 3072, in place)"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
fc2 = len(sys.argv) > 1 and sys.argv[1] == "fc2"
M, N, K = (87904, 768, 3072) if fc2 else (87904, 3072, 768)
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.03; b = torch.randn(N, device='cuda')
xs = nn_ops.c8_rows(x, ones=True)
ws = nn_ops.c8_rows(w, bias=b)
if fc2:
    res = torch.randn(M, N, device='cuda')
    for _ in range(5):
        nn_ops.gemm_c8(xs, ws, M, N, K, nn_ops.EPI_RESIDUAL, residual=res, out=res)
else:
    out = torch.zeros((M, nn_ops.split_ld(N)), device='cuda', dtype=torch.float16)
    for _ in range(5):
        nn_ops.gemm_c8(xs, ws, M, N, K, nn_ops.EPI_GELU, out=out, ldy=nn_ops.split_ld(N))
torch.cuda.synchronize()
