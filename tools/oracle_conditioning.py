"""How well-posed is "1e-3 relative on the min-max NORMALISED CAMs" on a given weight / batch draw?  The fp32 CPU oracle (the reference's own
arithmetic) against the same oracle in FLOAT64, on the draw of tests/test_precision_gpu.py's sweep: per active class plane the normalised-CAM
difference, next to the conditioning of the plane -- max |raw class logit| over the relu'd, scale-summed CAM's range (what the normalisation
divides by).   usage: python tools/oracle_conditioning.py SEED [S]      (CPU only; test infrastructure: imports oracle/)"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from oracle import torch_oracle as to          # noqa: E402
from cosa_amd.models import build_model        # noqa: E402
from cosa_amd.train_step import default_args, synthetic_batch          # noqa: E402

seed, S = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 448
torch.manual_seed(seed)
net = build_model(default_args("VOC12", crop_size=S, compute_dtype=torch.float32))
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
wimg, _, lab, box = synthetic_batch(2, S, 20, torch.device("cpu"), seed=seed + 2)


def run(dt):
    m = to.OracleViT(num_classes=21, aux_layer=-4)
    m.load_named(sd)
    m = m.to(dt)
    raw = {}

    def hook(cam):          # the raw class logits of the scale-1.0 pass (before relu / flip-max / normalisation)
        raw.setdefault("cam", cam)
    fwd = m.forward

    def spy(x, **kw):
        out = fwd(x, **kw)
        if x.shape[-1] == S:
            hook(out[4])
        return out
    m.forward = spy
    with torch.no_grad():
        cam, aux, _ = to.multi_scale_camseg(m, wimg.to(dt), [1.0, 0.5, 1.5])
    return cam, aux, raw["cam"]


c32, a32, r32 = run(torch.float32)
c64, a64, r64 = run(torch.float64)
act = lab.bool()
print(f"# seed {seed} S={S}: fp32 oracle vs float64 oracle, per active class plane (image, class)")
worst = 0.0
for b in range(2):
    for c in range(20):
        if not act[b, c]:
            continue
        d = float((c32[b, c].double() - c64[b, c]).abs().max())
        rawmax = float(r64[b, c].abs().max())
        # what the normalisation divides by: the range of the relu'd flip-max CAM summed over the scales = its max (min is 0 or above)
        worst = max(worst, d)
        pos = float((r64[b, c] > 0).double().mean())
        print(f"image {b} class {c:2d}: normalised-CAM |fp32 - fp64| max {d:.3e}   raw logit max|.| {rawmax:.3e}  share of positive pixels (scale 1.0) {pos:.4f}  "
              f"raw positive peak {float(r64[b, c].clamp_min(0).max()):.3e}")
print(f"worst {worst:.3e}")
