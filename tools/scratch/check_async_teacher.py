"""the side-stream teacher replay must not change results: same seeds, async on/off, losses of the first steps side by side"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda")
res = {}
for mode in (False, True, False):
    args = default_args("VOC12", crop_size=448, batch_size=8, teacher_async=mode)
    tr = CoSATrainer(args, dev, seed=0)
    wimg, simg, lab, box = synthetic_batch(8, 448, 20, dev, seed=1234)
    ls = []
    for i in range(8):
        logs = tr.step(wimg, simg, lab, box, args.warmup_iters + 1)
        ls.append([float(logs[k]) for k in ("overall_loss", "seg_loss", "cam_loss", "reg_loss")])
    res.setdefault(mode, []).append(ls)
    print("async" if mode else "sync ", " ".join(f"{v[0]:.5f}" for v in ls), flush=True)
a, b, c = res[False][0], res[True][0], res[False][1]
d_ab = max(abs(x - y) for u, v in zip(a, b) for x, y in zip(u, v))
d_ac = max(abs(x - y) for u, v in zip(a, c) for x, y in zip(u, v))
print(f"max |sync - async| over 8 steps x 4 losses: {d_ab:.3e};  sync vs sync (run-to-run noise of the atomics): {d_ac:.3e}")
