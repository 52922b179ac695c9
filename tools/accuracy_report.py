"""bf16 (throughput mode) and fp32 (parity mode) GPU step vs the CPU oracle step on identical weights/inputs:
label-map agreement / mean IoU, CAM relative error, loss differences.  Writes profiles/r01_accuracy.txt when run on the GPU box."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
from cosa_amd.utils import seg_helper
from oracle.cpu_step import CpuStep

def miou(a, b, n=21):
    ious = []
    for c in list(range(n)) + [255]:
        A, B = a == c, b == c
        u = (A | B).sum()
        if u: ious.append((A & B).sum() / u)
    return float(np.mean(ious))

dev = torch.device('cuda', 0)
S, b, C = int(sys.argv[1]) if len(sys.argv) > 1 else 224, 2, 20
out = []
sd = None
for name, dt in (("fp32 parity mode", torch.float32), ("bf16 throughput mode", torch.bfloat16)):
    args = default_args("VOC12", crop_size=S, compute_dtype=dt, teacher_graph=False)
    tr = CoSATrainer(args, dev, seed=3)
    if sd is None:
        sd = {k: v.detach().cpu().clone() for k, v in tr.student.state_dict().items()}
        cpu = CpuStep(sd, num_classes=21, aux_layer=-4)
        wimg, simg, lab, box = synthetic_batch(b, S, C, torch.device('cpu'), seed=5)
        closs, cl = cpu.losses(wimg, simg, lab, box.numpy(), args.warmup_iters + 1)
    else:
        tr.student.load_state_dict(sd); tr.model_AN.load_state_dict(sd)
        if tr._shadows is not None: tr._shadows.refresh()
    with torch.no_grad():
        cam, cam_aux, _ = seg_helper.multi_scale_camseg(tr.model_AN, wimg.to(dev), args.pseudo_scales)
    loss, lg = tr.forward_losses(wimg.to(dev), simg.to(dev), lab.to(dev), box, args.warmup_iters + 1)
    m_g, m_c = lg["mask"].cpu().numpy(), cl["mask"].numpy()
    act = lab.bool()
    rel = ((cam.cpu() - cl["cam_ps"]).abs().amax(dim=(2, 3)) / cl["cam_ps"].abs().amax(dim=(2, 3)).clamp_min(1e-6))[act].max().item()
    line = (f"{name}: S={S} b={b}: label agreement {np.mean(m_g == m_c):.5f}, mIoU(GPU vs CPU masks) {miou(m_g, m_c):.5f}, "
            f"max rel err of normalised CAMs (present classes) {rel:.3e}; losses GPU/CPU: " +
            ", ".join(f"{k} {float(lg[k]):.5f}/{float(cl[k]):.5f}" for k in ("cls_loss", "seg_loss", "cam_loss", "reg_loss", "overall_loss")))
    print(line); out.append(line)
open("gpurun_out/accuracy.txt", "w").write("\n".join(out) + "\n")
