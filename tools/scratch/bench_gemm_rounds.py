import sys, torch, os
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (N,K,epi) in [(768,3072,2),(768,768,2),(2304,768,0)]:
    out=f"N={N} K={K} epi={epi}:"
    for M in (87904, 87296, 87040, 65536, 608):
        x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
        r=torch.randn(M,N,device='cuda') if epi==2 else None
        t=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r))
        out+=f"  M={M} {t*1e3:.0f}us ({2.0*M*N*K/t/1e9:.0f}TF)"
    print(out, flush=True)
