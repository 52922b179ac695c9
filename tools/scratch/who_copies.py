"""which Python lines of the training step's FORWARD launch torch's copy / cat / cast kernels on large tensors (TorchDispatchMode + traceback)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda", 0)
args = default_args("VOC12", crop_size=448, batch_size=16, teacher_async=False, teacher_graph=False)
tr = CoSATrainer(args, dev, seed=0)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234)
n_iter = args.warmup_iters + 1
for _ in range(2):
    tr.step(wimg, simg, lab, box, n_iter)
agg = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        big = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor) and a.numel() >= 1 << 20]
        if big and any(k in name for k in ("copy", "cat", "clone", "_to_copy", "add", "mul", "fill", "zero", "where", "amax", "eq", "div", "contiguous")):
            fr = [f for f in traceback.extract_stack() if "cosa_amd" in f.filename and "who_copies" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            agg[(name, tuple(big[0].shape), str(big[0].dtype), where)] += 1
        return out


with Spy():
    loss, logs = tr.forward_losses(wimg, simg, lab, box, n_iter)
for k, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(n, k)
