import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cosa_amd.utils import seg_helper
for (B, K, H, W) in [(2, 21, 64, 96), (1, 5, 448, 448), (16, 21, 448, 448)]:
    torch.manual_seed(5)
    x = (torch.randn(B, K, H, W, device="cuda") * 3).requires_grad_(True)
    g = torch.randn(B, K, H // 2, W // 2, device="cuda")
    ref = F.interpolate(F.softmax(x, dim=1), scale_factor=0.5, mode="bilinear", align_corners=False, recompute_scale_factor=True)
    (gref,) = torch.autograd.grad(ref, x, g)
    x2 = x.detach().clone().requires_grad_(True)
    out = seg_helper.SoftmaxHalfRes.apply(x2)
    (gout,) = torch.autograd.grad(out, x2, g)
    gd = torch.autograd.grad(F.interpolate(F.softmax(x.double(), dim=1), scale_factor=0.5, mode="bilinear", align_corners=False, recompute_scale_factor=True), x, g.double())[0] if False else None
    print(B, K, H, W, "fwd max abs", float((out - ref).abs().max()), "equal", bool(torch.equal(out, ref)), "| bwd max abs", float((gout - gref).abs().max()), "max |g|", float(gref.abs().max()),
          "equal frac", float((gout == gref).float().mean()))
