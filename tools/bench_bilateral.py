"""Dense-energy regulariser (permutohedral bilateral filter) forward + backward alone at the bench configuration: b=16, 448^2 crops (filtered
at 224^2), K = 21 planes, synthetic images of SURVEY d-2.  Prints ms per image; run under tools/prof_kernels.sh / pmc passes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd.train_step import synthetic_batch
from cosa_amd.utils import seg_helper

dev = torch.device("cuda:0")
b, C, S = 16, int(os.environ.get("C", "20")), int(os.environ.get("S", "448"))
wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=1234)
g = torch.Generator(device="cpu").manual_seed(7)
layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
mask = torch.randint(0, C + 1, (b, S, S), generator=g).to(dev).float()
logit = torch.randn(b, C + 1, S, S, generator=g).to(dev).requires_grad_(True)

def f():
    logit.grad = None                 # (as zero_grad(set_to_none=True) leaves it)
    seg_helper.get_energy_loss(simg, logit, mask, box, layer).backward()

for _ in range(3):
    f()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    f()
e.record(); torch.cuda.synchronize()
print(json.dumps({"bilateral_fwd_bwd_ms_per_img": round(a.elapsed_time(e) / 10 / b, 5), "b": b, "S": S, "K": C + 1}))
