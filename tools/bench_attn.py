import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
H=12
for B,N in [(32,785),(32,197),(32,1765),(16,785)]:
    qkv=torch.randn(B,N,3*H*64,device='cuda').bfloat16()
    L=_C.lib(); ws=_C.workspace(L.cosa_attn_workspace_bytes(B,N,H),'cuda','attn')
    out=torch.empty(B,N,H*64,device='cuda',dtype=torch.bfloat16); lse=torch.empty(B,H,N,device='cuda')
    L.cosa_attn_prepare_vt(_C.ptr(qkv),B,N,H,_C.ptr(ws),ws.numel(),_C.stream_ptr())
    t_old=timeit(lambda: L.cosa_attn_fwd(_C.ptr(qkv),_C.ptr(out),_C.ptr(lse),B,N,H,64,0.125,3,None,_C.ptr(ws),ws.numel(),_C.stream_ptr()))
    ref=out.clone()
    t=timeit(lambda: L.cosa_attn_fwd(_C.ptr(qkv),_C.ptr(out),_C.ptr(lse),B,N,H,64,0.125,5,None,_C.ptr(ws),ws.numel(),_C.stream_ptr()))
    fl=4.0*B*H*N*N*64/1e12
    print("   4x32 kernel: %.0f us %.0f TF; new-vs-old maxdiff %.3e" % (t_old*1e3, fl/t_old*1e3, (out.float()-ref.float()).abs().max().item()))
    tp=timeit(lambda: L.cosa_attn_prepare_vt(_C.ptr(qkv),B,N,H,_C.ptr(ws),ws.numel(),_C.stream_ptr()))
    print(f"fwd B={B} N={N}: {t*1e3:.0f} us  {fl/t*1e3:.0f} TF   (vt prep {tp*1e3:.0f} us)")
    if B==16:
        do=torch.randn(B,N,H*64,device='cuda').bfloat16(); dq=torch.empty_like(qkv)
        wb=_C.workspace(L.cosa_attn_bwd_workspace_bytes(B,N,H),'cuda','attn_bwd')
        t=timeit(lambda: L.cosa_attn_bwd(_C.ptr(qkv),_C.ptr(out),_C.ptr(do),_C.ptr(lse),_C.ptr(dq),B,N,H,64,0.125,_C.ptr(wb),wb.numel(),_C.stream_ptr()))
        print(f"bwd B={B} N={N}: {t*1e3:.0f} us  {2.5*fl/t*1e3:.0f} TF(5-matmul flops)")
