"""which Python lines of a whole training step (teacher pass, student forward, backward, optimizer) launch torch's (ATen) kernels, with the bytes
they touch (TorchDispatchMode + traceback; autograd's engine-side adds show up as "?")"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
agg, byt = collections.Counter(), collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        ts = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []) if isinstance(a, torch.Tensor) and a.is_cuda]
        view = any(k in name for k in ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "detach", "alias", "unsqueeze", "squeeze", "as_strided", "t.default", "unbind", "split", "_unsafe_view", "empty", "stride", "size", "is_", "_local_scalar", "lift"))
        if ts and not view:
            nbytes = sum(a.numel() * a.element_size() for a in ts)
            fr = [f for f in traceback.extract_stack() if "cosa_amd" in f.filename and "aten_calls_in_step" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            key = (name, tuple(ts[0].shape), str(ts[0].dtype).replace("torch.", ""), where)
            agg[key] += 1
            byt[key] += nbytes
        return out


with Spy():
    tr.step(wimg, simg, lab, box, n_iter)
tot = sum(byt.values())
print(f"# {sum(agg.values())} ATen calls, {tot / 1e6:.0f} MB touched (at 5 TB/s: {tot / 5e12 * 1e3:.2f} ms)")
for k, n in sorted(agg.items(), key=lambda kv: -byt[kv[0]])[:70]:
    print(f"{byt[k] / 1e6:9.1f} MB  x{n:<3d} {k}")
