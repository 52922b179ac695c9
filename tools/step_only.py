"""N training steps of the bench configuration and nothing else (for rocprofv3 --kernel-trace --stats: per-step kernel shares without bench.py's
extra legs).  usage: python tools/step_only.py [steps=10] [teacher_precision=bf16] [dataset=VOC12] [nodefer]; the first 4 steps are set-up (graph capture);
"nodefer": the student's weight gradients as one launch per linear instead of the batched deferred launch (A/B); "groupsN": N batched launches;
a fifth argument picks the student's residual stream (fp32 | bf16)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dataset = sys.argv[3] if len(sys.argv) > 3 else "VOC12"          # VOC12 (20 classes) | COCO (80)
dev = torch.device("cuda", 0)
kw = dict(crop_size=448, batch_size=16, teacher_async=os.environ.get("COSA_TEACHER_SYNC") is None)
try:
    args = default_args(dataset, teacher_precision=prec, **kw)
except TypeError:        # a round-1 checkout (same-box comparisons): no precision switch
    args = default_args(dataset, **kw)
tr = CoSATrainer(args, dev, seed=0)
if len(sys.argv) > 4 and sys.argv[4] == "nodefer":
    tr.student.encoder.defer_wgrad = False
if len(sys.argv) > 4 and sys.argv[4].startswith("groups"):          # groupsN: the blocks' weight gradients in N batched launches
    tr.student.encoder.defer_groups = int(sys.argv[4][6:])
if len(sys.argv) > 5:                                                 # fp32 | bf16 residual stream of the student
    tr.student.encoder.residual_stream = sys.argv[5]
wimg, simg, lab, box = synthetic_batch(16, 448, 80 if dataset == "COCO" else 20, dev, seed=1234)
n_iter = args.warmup_iters + 1
for _ in range(4):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"ms_per_step": round(dt * 1e3, 3), "images_per_s": round(16 / dt, 2), "steps": steps, "setup_steps": 4, "teacher": prec, "wgrad": sys.argv[4] if len(sys.argv) > 4 else "batched", "stream": tr.student.encoder.residual_stream}))
