"""one launch shape of the three-term (fp16x3 / bf16x3) projection GEMM for rocprofv3 --pmc passes.  default: the teacher's fc1 + GELU (M = all tokens of a
step), split rows in and out;  `fc2`: mlp.fc2 + fp32 residual (N = 768, K = 3072, in place);  second argument: bf16 (default fp16 halves)"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
fc2 = len(sys.argv) > 1 and sys.argv[1] == "fc2"
hdt = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float16
M, N, K = (87904, 768, 3072) if fc2 else (87904, 3072, 768)
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.03; b = torch.randn(N, device='cuda')
xs = nn_ops.split_rows(x, ones=True, dtype=hdt)
ws = nn_ops.split_rows(w, bias=b, dtype=hdt)
if fc2:
    res = torch.randn(M, N, device='cuda')
    for _ in range(5):
        nn_ops.gemm_x3(xs, ws, M, N, K, nn_ops.EPI_RESIDUAL, residual=res, out=res)
else:
    out = torch.zeros((M, nn_ops.split_ld(N)), device='cuda', dtype=hdt)
    for _ in range(5):
        nn_ops.gemm_x3(xs, ws, M, N, K, nn_ops.EPI_GELU, out=out, ldy=nn_ops.split_ld(N))
torch.cuda.synchronize()
