"""which Python lines launch torch's own copy / cast / add kernels in a training step (torch.profiler with stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda", 0)
args = default_args("VOC12", teacher_precision="fp16c4-8", crop_size=448, batch_size=16, teacher_async=False)
tr = CoSATrainer(args, dev, seed=0)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234)
n_iter = args.warmup_iters + 1
for _ in range(5):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(wimg, simg, lab, box, n_iter)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue                      # count leaves only
    st = [s for s in (ev.stack or []) if "cosa_amd" in s or "bench" in s]
    key = (ev.name, str(ev.input_shapes)[:70], st[0][-90:] if st else "?")
    agg[key][0] += 1
    agg[key][1] += ev.device_time_total
tot = 0
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    tot += t
    print(f"{t:8.1f} us x{n:3d}  {k[0]:28s} {k[1]:70s} {k[2]}")
print("listed total us", tot)
