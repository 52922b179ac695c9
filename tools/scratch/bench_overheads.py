"""cost of bench.py's own instrumentation inside the timed loop: device stamps (per-workgroup atomics) and HIP event pairs"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cosa_amd import nn_ops, _C
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
mode = sys.argv[1]
dev = torch.device("cuda", 0)
args = default_args("VOC12", teacher_precision="bf16", crop_size=448, batch_size=16, teacher_async=True)
if "stamps" in mode:
    nn_ops.stamps = nn_ops.KernelStamps(dev)
    nn_ops.gemm_stamps = nn_ops.KernelStamps(dev)
tr = CoSATrainer(args, dev, seed=0)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234)
n_iter = args.warmup_iters + 1
for _ in range(5):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
if "events" in mode:
    _C.profile_start()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(json.dumps({"mode": mode, "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(16 / dt, 2)}), flush=True)
