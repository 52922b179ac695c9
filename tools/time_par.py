import sys, torch, time
sys.path.insert(0, '.')
from cosa_amd.train_step import default_args, synthetic_batch
from cosa_amd.utils import seg_helper, torch_helper
from cosa_amd.models.PAR import PAR
dev=torch.device('cuda',0)
wimg,simg,lab,box=synthetic_batch(16,448,20,dev)
cams=torch.rand(16,20,56,56,device=dev); cams=torch.nn.functional.interpolate(cams,size=(448,448),mode='bilinear')
den=torch_helper.denormalize_img(simg)
par=PAR(num_iter=10,dilations=[1,2,4,8,12,24])
def t(f,n=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
a=t(lambda: seg_helper.cam2mask(den,box,cams,lab,0.7,0.25,_fold_validation=True))
b=t(lambda: seg_helper.cam2mask(den,box,cams,lab,0.7,0.25,refine_model=par,_fold_validation=True))
K=float((lab.sum(1)+1).mean())
print(f"cam2mask no-PAR {a:.3f} ms/batch16 ({a/16*1e3:.1f} us/img); with PAR {b:.3f} ms ({b/16:.4f} ms/img per call; x2 calls per step = {2*b/16:.4f} ms/img); mean K={K:.2f}")
alg=4*224*224*(3+2*K*10)*2*16   # hi+lo per image
print(f"PAR algorithmic bytes per call batch: {alg/1e6:.1f} MB -> {alg/((b-a)*1e-3)/1e12:.3f} TB/s = {alg/((b-a)*1e-3)/8e12*100:.1f}% of 8 TB/s")
