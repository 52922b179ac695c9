"""timing ablations of the v5 GEMM (results are wrong by construction): 51 no DMA, 52 no fragment reads, 53 neither, 54 no barriers, 56/57 combos"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (M,N,K) in [(87904,2304,768),(87904,3072,768)]:
    x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
    fl=2.0*M*N*K/1e12; out=f"M={M} N={N} K={K}"
    for v in (6,61,66,67,68,6):
        _C.lib().cosa_gemm_set_variant(v)
        t=timeit(lambda: nn_ops.gemm_bf16(x,w,b,0))
        out+=f" | v{v} {t*1e3:.0f}us {fl/t*1e3:.0f}TF"
    print(out, flush=True)
