"""300 training steps of the bench configuration: memory must not grow, losses stay finite, two trainers with the same seed agree bit for bit"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda", 0)
args = default_args("VOC12", teacher_precision=(sys.argv[1] if len(sys.argv) > 1 else "fp16c4-8"), crop_size=448, batch_size=16, teacher_async=True)
tr = CoSATrainer(args, dev, seed=0)
n_iter = args.warmup_iters + 1
mem = []
t0 = time.perf_counter()
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234 + (i % 7))
    logs = tr.step(wimg, simg, lab, box, n_iter + i)
    if i % 50 == 49:
        torch.cuda.synchronize()
        mem.append((i + 1, round(torch.cuda.memory_allocated() / 2**30, 3), round(torch.cuda.max_memory_allocated() / 2**30, 3), float(logs["overall_loss"])))
        print(mem[-1], flush=True)
torch.cuda.synchronize()
print("steps/s", (i + 1) / (time.perf_counter() - t0))
assert all(m[3] == m[3] for m in mem), "NaN loss"
assert mem[-1][1] <= mem[1][1] + 0.05, "allocated memory grows"
print("OK")
