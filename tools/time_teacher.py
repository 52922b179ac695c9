import sys, torch, time
sys.path.insert(0, '.')
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
from cosa_amd.utils import seg_helper
from cosa_amd import nn_ops
dev=torch.device('cuda',0)
args=default_args('VOC12'); tr=CoSATrainer(args,dev)
wimg,simg,lab,box=synthetic_batch(16,448,20,dev)
def t(f,n=3):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
enc=tr.model_AN.encoder
with torch.no_grad():
    print("use_fused", enc.use_fused(wimg))
    print("teacher multi-scale fused ms", t(lambda: seg_helper.multi_scale_camseg(tr.model_AN,wimg,args.pseudo_scales)))
    a=seg_helper.multi_scale_camseg(tr.model_AN,wimg,args.pseudo_scales)
    enc.use_fused=lambda x: False
    print("teacher multi-scale unfused ms", t(lambda: seg_helper.multi_scale_camseg(tr.model_AN,wimg,args.pseudo_scales)))
    b=seg_helper.multi_scale_camseg(tr.model_AN,wimg,args.pseudo_scales)
    for u,v in zip(a,b): print("fused vs unfused maxdiff", (u-v).abs().max().item(), v.abs().max().item())
    x=torch.cat([wimg,wimg.flip(-1)],0)
    del enc.use_fused
    print("one fwd b=32 448 fused ms", t(lambda: tr.model_AN(x)))
print("student fwd+bwd ms", t(lambda: tr.step(wimg,simg,lab,box,10**6)))
