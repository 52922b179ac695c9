import sys, torch, time
sys.path.insert(0, '.')
import torch.nn.functional as F
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
from cosa_amd.utils import seg_helper, torch_helper
dev=torch.device('cuda',0)
args=default_args('VOC12'); tr=CoSATrainer(args,dev)
wimg,simg,lab,box=synthetic_batch(16,448,20,dev)
for _ in range(2): tr.step(wimg,simg,lab,box,10**6)
T={}
class R:
    def __init__(s,n): s.n=n
    def __enter__(s): torch.cuda.synchronize(); s.t=time.perf_counter()
    def __exit__(s,*a): torch.cuda.synchronize(); T[s.n]=T.get(s.n,0)+(time.perf_counter()-s.t)*1e3
n_iter=10**6
for it in range(3):
    with R("teacher"):
        cam_ps, cam_aux_ps, seg_ps = seg_helper.multi_scale_camseg(tr.model_AN, wimg, args.pseudo_scales)
    with R("student_fwd"):
        cls_final, cls_aux, _f, seg_pred, cam_pred, cam_aux_pred = tr.model_ON(simg)
    with R("cls_loss"):
        cls_loss = F.multilabel_soft_margin_loss(cls_final, lab); cls_loss_aux = F.multilabel_soft_margin_loss(cls_aux, lab)
    with R("cam2mask x2"):
        m1 = seg_helper.cam2mask(simg, box, cam_ps, lab, 0.7, 0.25, _fold_validation=True)
        m2 = seg_helper.cam2mask(simg, box, cam_aux_ps, lab, 0.7, 0.25, _fold_validation=True)
    with R("seg_up+seg_loss x2"):
        sp = F.interpolate(seg_pred, size=m1.shape[1:], mode='bilinear', align_corners=False)
        sl = 0.5*seg_helper.seg_loss(sp, m1)+0.5*seg_helper.seg_loss(sp, m2)
    with R("energy_loss"):
        rl = seg_helper.get_energy_loss(img=simg, logit=sp, label=m1, img_box=box, loss_layer=tr.reg_layer)
    with R("seg_refine+cam_loss"):
        vs = seg_helper.seg_refine_by_label(seg_ps, lab, softmaxtemp=0.01)
        cl = seg_helper.cam_loss(cam_pred, vs)
    loss = cls_loss+cls_loss_aux+0.1*sl+0.05*cl+0.05*rl
    with R("backward"):
        tr.optimizer.zero_grad(set_to_none=True); loss.backward()
    with R("optimizer"):
        tr.optimizer.step()
    with R("ema"):
        torch_helper.ema_update(tr._ema_pairs[0], tr._ema_pairs[1], args.momentum)
tot=sum(T.values())
for k,v in T.items(): print(f"{k:24s} {v/3:8.2f} ms")
print("sum", tot/3)
