import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
v=int(sys.argv[1]) if len(sys.argv)>1 else 1
M,N,K=25120,2304,768
x=torch.randn(M,K,device='cuda').bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
_C.lib().cosa_gemm_set_variant(v)
for _ in range(5): y=nn_ops.gemm_bf16(x,w,b,0)
torch.cuda.synchronize()
