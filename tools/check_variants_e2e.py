"""teacher pass at the bench configuration (b=16, 448^2: M = 87904 token rows -> the persistent v6 GEMM) against the same pass on the
128x128 kernel (variant 1): CAM difference and pseudo-label agreement.  The two differ only in summation/rounding order of the epilogues."""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import _C
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
from cosa_amd.utils import seg_helper
dev = torch.device("cuda")
args = default_args("VOC12", crop_size=448, batch_size=16, teacher_graph=False)
tr = CoSATrainer(args, dev, seed=0)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234)
outs = {}
for v in (1, 0):
    _C.lib().cosa_gemm_set_variant(v)
    with torch.no_grad():
        cam, aux, seg = seg_helper.multi_scale_camseg(tr.model_AN, wimg, args.pseudo_scales)
    m = seg_helper.cam2mask(simg, box, seg_helper.cam_validation(cam, lab), lab, args.high_thre, args.low_thre)
    outs[v] = (cam.clone(), aux.clone(), seg.clone(), m.clone())
_C.lib().cosa_gemm_set_variant(0)
a, b = outs[1], outs[0]
act = lab[:, :, None, None] > 0
for name, x, y in (("cam", a[0], b[0]), ("cam_aux", a[1], b[1])):
    d = ((x - y).abs() * act).max().item()
    print(f"{name}: max abs diff on present classes {d:.3e} (values in [0,1])")
print(f"seg: max abs diff {(a[2]-b[2]).abs().max().item():.3e} of max |seg| {a[2].abs().max().item():.3f}")
print(f"pseudo-label agreement v6 vs 128x128 kernel: {(a[3] == b[3]).float().mean().item():.6f}")
