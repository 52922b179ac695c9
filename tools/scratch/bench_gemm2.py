import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
torch.manual_seed(0)
M=87904
for (N,K,epi) in [(2304,768,0),(768,768,2),(3072,768,1),(768,3072,2)]:
    x=torch.randn(M,K,device='cuda').bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16(); r=torch.randn(M,N,device='cuda')
    res=[]
    for v in (1,3,4):
        _C.lib().cosa_gemm_set_variant(v)
        t=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r if epi==2 else None))
        res.append(f"v{v} {2.0*M*N*K/1e12/t*1e3:.0f} TF ({t*1e3:.0f}us)")
    t=timeit(lambda: torch.nn.functional.linear(x,w,b))
    print(f"M={M} N={N} K={K} epi={epi}: "+" | ".join(res)+f" | hipBLASLt bare {2.0*M*N*K/1e12/t*1e3:.0f} TF")
