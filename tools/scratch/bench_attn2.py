import sys, torch
sys.path.insert(0, '.')
from cosa_amd import _C
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
L=_C.lib()
for (B,N,H) in [(32,785,12),(16,785,12),(32,197,12),(32,1765,12),(16,3601,12)]:
    qkv=torch.randn(B,N,3*H*64,device='cuda').bfloat16()
    ws=_C.workspace(L.cosa_attn_workspace_bytes(B,N,H),'cuda','attn')
    out=torch.empty(B,N,H*64,device='cuda',dtype=torch.bfloat16); lse=torch.empty(B,H,N,device='cuda')
    res=[]
    for fl in (2,4):
        t=timeit(lambda: L.cosa_attn_fwd(_C.ptr(qkv),_C.ptr(out),_C.ptr(lse),B,N,H,64,0.125,fl,None,_C.ptr(ws),ws.numel(),_C.stream_ptr()))
        res.append(f"{'4x32' if fl==2 else '2x64'} {t*1e3:.0f}us {4.0*B*H*N*N*64/t/1e9:.0f}TF")
    print(f"B={B} N={N}: "+" | ".join(res), flush=True)
