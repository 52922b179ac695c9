import sys, os, subprocess
# one process per stagger value (the launcher reads the env var once)
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, '.')
    from cosa_amd import nn_ops, _C
    def timeit(f, n=20):
        for _ in range(5): f()
        torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): f()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
    out = f"stagger={os.environ.get('COSA_GEMM_STAGGER')}"
    for (M,N,K,epi) in [(87904,2304,768,0),(87904,3072,768,1),(12560,2304,768,0)]:
        x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
        _C.lib().cosa_gemm_set_variant(6)
        timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi))
        t=timeit(lambda: nn_ops.gemm_bf16(x,w,b,epi))
        out += f" | N={N} epi={epi} M={M} {t*1e3:.0f}us {2.0*M*N*K/t/1e9:.0f}TF"
    print(out, flush=True)
else:
    for s in ("0", "1200", "2340", "3500", str((1<<20)+1200), str((1<<20)+2340), str((1<<20)+4680)):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, COSA_GEMM_STAGGER=s))
