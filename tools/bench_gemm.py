"""micro-benchmark + correctness of cosa_gemm_bf16 variants against torch (hipBLASLt) at the step's shapes
usage: bench_gemm.py [variants, e.g. 1,3,5] [quick]"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
torch.manual_seed(0)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,3,5").split(",")]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
shapes = [(87904,2304,768,0),(87904,768,768,2),(87904,3072,768,1),(87904,768,3072,2),
          (12560,2304,768,0),(12560,768,768,0),(12560,3072,768,1),(12560,768,3072,0),
          (25120,2304,768,0),(6304,2304,768,0),(8192,8192,8192,0),(300,768,768,2),(257,256,64,1)]
if len(sys.argv) > 2: shapes = shapes[:4] + shapes[-2:]
for (M,N,K,epi) in shapes:
    x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
    r=torch.randn(M,N,device='cuda') if epi==2 else None
    fl=2.0*M*N*K/1e12
    if M*N <= 87904*3072:
        ref=(x.float()@w.float().t()+b.float())
        if epi==1: ref=torch.nn.functional.gelu(ref)
        if epi==2: ref=ref+r
    else: ref=None
    out=f"M={M} N={N} K={K} epi={epi}"
    for v in variants:
        _C.lib().cosa_gemm_set_variant(v)
        y=nn_ops.gemm_bf16(x,w,b,epi,residual=r)
        err=((y.float()-ref).abs().max().item()/ref.abs().max().item()) if ref is not None else float('nan')
        t=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r))
        out+=f" | v{v} {t*1e3:.0f}us {fl/t*1e3:.0f}TF err {err:.1e}"
        del y
    _C.lib().cosa_gemm_set_variant(0)
    if epi==0: f=lambda: torch.nn.functional.linear(x,w,b)
    elif epi==1: f=lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x,w,b))
    else: f=lambda: r+torch.nn.functional.linear(x,w,b)
    t_ref=timeit(f)
    print(out+f" | torch {t_ref*1e3:.0f}us {fl/t_ref*1e3:.0f}TF", flush=True)
    del x,w,b,r,ref
