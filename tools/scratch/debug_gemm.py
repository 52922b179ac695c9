import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops, _C
torch.manual_seed(0)
for v in (1,3):
  for (M,N,K) in [(6304,2304,768),(12560,2304,768),(25120,2304,768)]:
    x=(torch.randn(M,K,device='cuda')).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.03).bfloat16(); b=torch.randn(N,device='cuda').bfloat16()
    ref=(x.float()@w.float().t()+b.float())
    _C.lib().cosa_gemm_set_variant(v)
    worst=0
    for rep in range(3):
        y=nn_ops.gemm_bf16(x,w,b,0).float()
        err=(y-ref).abs()
        rel=(err/(ref.abs()+1.0))
        idx=rel.argmax().item(); m,n=idx//N, idx%N
        bad=(rel>8e-3).sum().item()
        print(f"v{v} M={M} rep{rep} max rel(err/(|ref|+1))={rel.max().item():.3e} at m={m} (tile {m//256}, in-tile {m%256}) n={n} (tile {n//256}); count>8e-3: {bad}; y={y[m,n].item():.4f} ref={ref[m,n].item():.4f}")
