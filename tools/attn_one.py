import sys, torch
sys.path.insert(0, '.')
from cosa_amd import _C
B,N,H=32,1765,12
qkv=torch.randn(B,N,3*H*64,device='cuda').bfloat16()
L=_C.lib(); ws=_C.workspace(L.cosa_attn_workspace_bytes(B,N,H),'cuda','attn')
out=torch.empty(B,N,H*64,device='cuda',dtype=torch.bfloat16); lse=torch.empty(B,H,N,device='cuda')
L.cosa_attn_prepare_vt(_C.ptr(qkv),B,N,H,_C.ptr(ws),ws.numel(),_C.stream_ptr())
for _ in range(3):
    L.cosa_attn_fwd(_C.ptr(qkv),_C.ptr(out),_C.ptr(lse),B,N,H,64,0.125,1,None,_C.ptr(ws),ws.numel(),_C.stream_ptr())
torch.cuda.synchronize()
