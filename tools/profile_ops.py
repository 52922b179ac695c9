"""torch.profiler view of one training step: ATen ops by device time with input shapes (finds stray copies / casts / cats).
usage: python tools/profile_ops.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch

dev = torch.device("cuda:0")
args = default_args("VOC12", batch_size=16, crop_size=448, teacher_async=False)
tr = CoSATrainer(args, dev)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev)
for i in range(6):
    tr.step(wimg, simg, lab, box, n_iter=args.warmup_iters + 1 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    tr.step(wimg, simg, lab, box, n_iter=args.warmup_iters + 10)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
sel = [r for r in rows if r.key.startswith("aten::") and r.key.split("::")[1] in
       ("copy_", "cat", "_to_copy", "contiguous", "clone", "add", "add_", "mul", "amax", "max", "sum", "fill_", "zero_", "where", "stack",
        "gelu", "gelu_backward", "index", "flip", "sub", "div", "neg", "to", "select_backward", "slice_backward", "cat_backward",
        "mean", "sigmoid", "log_sigmoid_forward", "upsample_bilinear2d", "permute", "reshape", "view", "expand", "sum_to_size")]
sel.sort(key=lambda r: -getattr(r, "device_time_total", getattr(r, "cuda_time_total", 0)))
for r in sel[:45]:
    t = getattr(r, "device_time_total", getattr(r, "cuda_time_total", 0))
    if t < 15:
        break
    print(f"{t:9.1f} us  x{r.count:<3d} {r.key:28s} {str(r.input_shapes)[:150]}")
