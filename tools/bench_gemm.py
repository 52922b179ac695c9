"""micro-benchmark + correctness of cosa_gemm_bf16 / cosa_layernorm against torch (hipBLASLt) at the step's shapes"""
import sys, torch, time
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
torch.manual_seed(0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (M,N,K,epi) in [(25120,2304,768,0),(25120,768,768,2),(25120,3072,768,1),(25120,768,3072,2),(6304,2304,768,0),(56480,2304,768,0),(56480,3072,768,1),(56480,768,3072,2),(12560,2304,768,0),(300,768,768,2)]:
    x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
    r=torch.randn(M,N,device='cuda')
    ref=(x.float()@w.float().t()+b.float())
    if epi==1: ref=torch.nn.functional.gelu(ref)
    if epi==2: ref=ref+r
    _C.lib().cosa_gemm_set_variant(1)
    t_v1=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r if epi==2 else None))
    _C.lib().cosa_gemm_set_variant(3)
    t_v3=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r if epi==2 else None))
    _C.lib().cosa_gemm_set_variant(4)
    y=nn_ops.gemm_bf16(x,w,b,epi,residual=r if epi==2 else None)
    err=(y.float()-ref).abs().max().item()/ref.abs().max().item()
    t_mine=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi,residual=r if epi==2 else None))
    if epi==0: f=lambda: torch.nn.functional.linear(x,w,b)
    elif epi==1: f=lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x,w,b))
    else: f=lambda: r+torch.nn.functional.linear(x,w,b)
    t_ref=timeit(f)
    fl=2.0*M*N*K/1e12
    print(f"M={M} N={N} K={K} epi={epi} relerr={err:.2e} v1 {fl/t_v1*1e3:.0f} TF | v3 {fl/t_v3*1e3:.0f} TF | v4 {t_mine*1e3:.1f}us {fl/t_mine*1e3:.0f} TF | torch {t_ref*1e3:.1f}us {fl/t_ref*1e3:.0f} TF")
x=torch.randn(25120,768,device='cuda')*2+0.5; g=torch.randn(768,device='cuda').bfloat16(); bb=torch.randn(768,device='cuda').bfloat16()
y16,y32=nn_ops.layernorm_f32(x,g,bb,1e-6,True,True)
ref=torch.nn.functional.layer_norm(x,(768,),g.float(),bb.float(),1e-6)
print("LN err", (y32-ref).abs().max().item(), (y16.float()-ref).abs().max().item(), "time us", timeit(lambda: nn_ops.layernorm_f32(x,g,bb,1e-6))*1e3, "torch", timeit(lambda: torch.nn.functional.layer_norm(x,(768,),g.float(),bb.float(),1e-6).bfloat16())*1e3)
