"""which torch (non-cosa) ops does a training step run, and from where: a TorchDispatchMode logs every aten op of one step with the innermost
cosa_amd call site and the bytes it moves (output elements x element size)"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from cosa_amd.train_step import CoSATrainer, default_args, synthetic_batch
dev = torch.device("cuda", 0)
args = default_args("VOC12", crop_size=448, batch_size=16, teacher_async=False, teacher_graph=False)
tr = CoSATrainer(args, dev, seed=0)
wimg, simg, lab, box = synthetic_batch(16, 448, 20, dev, seed=1234)
n_iter = args.warmup_iters + 1
for _ in range(3):
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
log = collections.defaultdict(lambda: [0, 0])
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::slice", "aten::select", "aten::t", "aten::transpose", "aten::permute", "aten::expand",
        "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::_local_scalar_dense", "aten::item", "aten::is_nonzero", "aten::narrow", "aten::unbind", "aten::split", "aten::chunk", "aten::view_as", "aten::stride", "aten::sym")

class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "cosa_amd" in fr.filename:
                    site = "%s:%d" % (fr.filename.split("cosa_amd/")[-1], fr.lineno)
                    break
            o = out[0] if isinstance(out, (tuple, list)) and len(out) else out
            nbytes = o.numel() * o.element_size() if isinstance(o, torch.Tensor) else 0
            e = log[(name, site)]
            e[0] += 1
            e[1] += nbytes
        return out

with Log():
    tr.step(wimg, simg, lab, box, n_iter)
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -(kv[1][1] + 4e5 * kv[1][0]))      # ~ bytes + 0.4 MB-equivalent per launch
for (name, site), (n, b) in rows[:90]:
    print("%4d x %9.2f MB  %-26s %s" % (n, b / 1e6, name, site))
print("ops total", sum(v[0] for v in log.values()))
