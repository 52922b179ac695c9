"""PAR refinement passes alone for rocprofv3 --pmc: cam2mask_multi with PAR(T=10, 6 dilations) on main + aux CAM sets, b = 16, 448^2 (one pass = the
four PAR calls per image of the metric); prints the number of passes"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd.models.PAR import PAR
from cosa_amd.utils import seg_helper
from cosa_amd.train_step import synthetic_batch
from cosa_amd.utils import torch_helper
dev = torch.device("cuda:0")
b, C, S = 16, 20, 448
wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=1234)
g = torch.Generator(device="cpu").manual_seed(7)
up = lambda t: torch.nn.functional.interpolate(t.to(dev), size=(S, S), mode="bilinear")
cams, cams_aux = up(torch.rand(b, C, S // 8, S // 8, generator=g)), up(torch.rand(b, C, S // 8, S // 8, generator=g))
den = torch_helper.denormalize_img(simg)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24])
n = 6
for _ in range(n):
    seg_helper.cam2mask_multi(den, box, [cams, cams_aux], lab, [0.7, 0.7], [0.25, 0.25], refine_model=par, _fold_validation=True)
torch.cuda.synchronize()
K = float((lab.sum(1) + 1).mean())
print(json.dumps({"passes": n, "mean_K": K, "algorithmic_bytes_per_pass": 4.0 * (S // 2) ** 2 * (3 + 2 * K * 10) * 4 * b}))
