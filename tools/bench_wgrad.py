import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
torch.manual_seed(0)
for (M,N,K) in [(12560,3072,768),(12560,768,3072),(12560,2304,768),(12560,768,768),(1000,128,128),(64,128,256),(130,256,128)]:
    dy=torch.randn(M,N,device='cuda').bfloat16(); x=torch.randn(M,K,device='cuda').bfloat16()
    ref=dy.float().t()@x.float()
    out,db=nn_ops.gemm_wgrad(dy,x,want_bias=True)
    err=(out-ref).abs().max().item()/ref.abs().max().item()
    berr=(db-dy.float().sum(0)).abs().max().item()/dy.float().sum(0).abs().max().item()
    print("   bias-grad relerr %.2e"%berr)
    t=timeit(lambda: nn_ops.gemm_wgrad(dy,x)); t2=timeit(lambda: nn_ops._mm_f32(dy.t(),x))
    fl=2.0*M*N*K/1e12
    print(f"wgrad M={M} N={N} K={K} relerr={err:.2e} mine {t*1e3:.1f}us {fl/t*1e3:.0f} TF | torch {t2*1e3:.1f}us {fl/t2*1e3:.0f} TF")
