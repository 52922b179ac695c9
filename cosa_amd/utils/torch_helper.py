"""utils.torch_helper -- the hot-path subset of the reference module, MI355X-native.

Reference: utils/torch_helper.py:32-42 (setup_seed), :261-293 (PolyWarmupAdamW), :354-367
(denormalize_img_/denormalize_img).
"""
import random

import numpy as np
import torch

from .. import _C


def setup_seed(seed):
    """utils/torch_helper.py:32-42"""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def denormalize_img(imgs):
    """utils/torch_helper.py:354-367: (x*std+mean) -> uint8 truncation -> /255, one fused kernel."""
    _C.require_cuda(imgs)
    imgs = imgs.contiguous().float()
    B, C, H, W = imgs.shape
    if C != 3:
        raise ValueError("denormalize_img expects [B,3,H,W]")
    out = torch.empty_like(imgs)
    _C.check(_C.lib().cosa_denormalize_img(_C.ptr(imgs), _C.ptr(out), B, H, W, _C.stream_ptr()), "cosa_denormalize_img")
    return out


def denormalize_img_(imgs, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
    """uint8 image (utils/torch_helper.py:354-361); derived from the fused kernel's output."""
    if tuple(mean) != (123.675, 116.28, 103.53) or tuple(std) != (58.395, 57.12, 57.375):
        raise NotImplementedError("only the ImageNet mean/std of the reference are compiled in")
    return (denormalize_img(imgs) * 255.0).to(torch.uint8)


def poly_warmup_lr_mult(step, warmup_iter, max_iter, warmup_ratio, power, min_mult):
    """LR multiplier of PolyWarmupAdamW.step (utils/torch_helper.py:275-289); None = keep previous LR."""
    if step < warmup_iter:
        return 1 - (1 - step / warmup_iter) * (1 - warmup_ratio)
    if step < max_iter:
        return max((1 - step / max_iter) ** power, min_mult)
    return None


class PolyWarmupAdamW(torch.optim.AdamW):
    """utils/torch_helper.py:261-293, same constructor and schedule.

    The update itself is torch's fused multi-tensor AdamW (one launch per dtype group, no
    per-parameter Python loop); `ema_update()` applies the teacher EMA (main.py:250-252) with
    one foreach launch.
    """

    def __init__(self, params, lr, weight_decay, betas, warmup_iter, max_iter, warmup_ratio, power, min_mult=0, **kwargs):
        fused = kwargs.pop("fused", None)
        params = list(params)
        if fused is None:
            first = params[0]["params"][0] if isinstance(params[0], dict) else params[0]
            fused = bool(first.is_cuda)
        super().__init__(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=1e-8, fused=fused)
        self.global_step = 0
        self.warmup_iter = warmup_iter
        self.warmup_ratio = warmup_ratio
        self.max_iter = max_iter
        self.power = power
        self.min_mult = min_mult
        self._init_lr = [group["lr"] for group in self.param_groups]

    def step(self, closure=None):
        mult = poly_warmup_lr_mult(self.global_step, self.warmup_iter, self.max_iter, self.warmup_ratio, self.power,
                                   self.min_mult)
        if mult is not None:
            for i, g in enumerate(self.param_groups):
                g["lr"] = self._init_lr[i] * mult
        super().step(closure)
        self.global_step += 1


@torch.no_grad()
def ema_update(teacher_params, student_params, momentum):
    """main.py:250-252: a <- m*a + (1-m)*o over parameters, as two foreach launches."""
    teacher_params, student_params = list(teacher_params), list(student_params)
    torch._foreach_mul_(teacher_params, momentum)
    torch._foreach_add_(teacher_params, student_params, alpha=1 - momentum)


class FusedAdamWEMAStep:
    """optimizer.step() + the teacher EMA (main.py:250-252) + refresh of the bf16 shadow weights as ONE HIP kernel.

    State lives where torch keeps it (optimizer.state[p]['exp_avg'|'exp_avg_sq'], optimizer.param_groups[i]['lr']) so the
    PolyWarmupAdamW object stays the source of truth (state_dict compatible); this class only replaces the sweeps over
    memory.  The LR schedule is PolyWarmupAdamW's (utils/torch_helper.py:275-289)."""

    def __init__(self, optimizer, student_params, teacher_params, momentum, shadow_of=None):
        import numpy as np
        self.opt = optimizer
        self.momentum = float(momentum)
        self.student = list(student_params)
        self.teacher = list(teacher_params)
        assert len(self.student) == len(self.teacher)
        dev = self.student[0].device
        L = _C.lib()
        self.rec_dtype = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("tp", "u8"), ("p16", "u8"), ("t16", "u8"),
                                   ("lr", "f4"), ("wd", "f4"), ("n", "i8"), ("t16_f16", "i4"), ("p16_f16", "i4")])
        assert self.rec_dtype.itemsize == L.cosa_optim_record_bytes()
        group_of = {}
        for gi, g in enumerate(optimizer.param_groups):
            for p in g["params"]:
                group_of[id(p)] = gi
        self.group_idx = [group_of.get(id(p), -1) for p in self.student]
        shadow_of = shadow_of or (lambda p: None)
        n = len(self.student)
        # The trainer has no per-step host sync, so the host may run steps ahead of the GPU: the record table (gradient pointers,
        # scheduled lr / wd) is therefore kept in a ring of kRing pinned host tables + device tables.  Slot s is rewritten only after
        # the event recorded behind the H2D copy that last read it has completed, and every step's kernel reads its own device table.
        self.kRing = 4
        self.hosts = [torch.zeros(n * self.rec_dtype.itemsize, dtype=torch.uint8).pin_memory() for _ in range(self.kRing)]
        self.recs = [h.numpy().view(self.rec_dtype) for h in self.hosts]
        self.copied = [None] * self.kRing
        self.slot = 0
        self.rec = self.recs[0]
        self._step_t = torch.tensor(0.0)
        chunk = L.cosa_optim_chunk_elems()
        chunks = []
        for i, (p, tp) in enumerate(zip(self.student, self.teacher)):
            assert p.is_contiguous() and tp.is_contiguous() and p.dtype == torch.float32 and tp.dtype == torch.float32
            r = self.rec[i]
            r["p"], r["tp"], r["n"] = p.data_ptr(), tp.data_ptr(), p.numel()
            sp, st = shadow_of(p), shadow_of(tp)
            r["p16"] = sp.data_ptr() if sp is not None else 0
            r["t16"] = st.data_ptr() if st is not None else 0
            r["p16_f16"] = int(sp is not None and sp.dtype == torch.float16)
            r["t16_f16"] = int(st is not None and st.dtype == torch.float16)
            if self.group_idx[i] >= 0:
                stt = optimizer.state[p]
                if "exp_avg" not in stt:
                    stt["step"] = self._step_t
                    stt["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    stt["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                r["m"], r["v"] = stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr()
            chunks += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        self.n_chunks = len(chunks)
        self.d_chunks = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous()
        for r in self.recs[1:]:
            r[:] = self.recs[0]
        self.d_recs = [torch.empty(n * self.rec_dtype.itemsize, dtype=torch.uint8, device=dev) for _ in range(self.kRing)]

    def step(self):
        opt = self.opt
        mult = poly_warmup_lr_mult(opt.global_step, opt.warmup_iter, opt.max_iter, opt.warmup_ratio, opt.power, opt.min_mult)
        if mult is not None:
            for i, g in enumerate(opt.param_groups):
                g["lr"] = opt._init_lr[i] * mult
        groups = opt.param_groups
        slot = self.slot
        self.slot = (slot + 1) % self.kRing
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()            # the copy issued kRing steps ago; never waits in practice
        rec, d_rec = self.recs[slot], self.d_recs[slot]
        for i, p in enumerate(self.student):
            gi = self.group_idx[i]
            if gi >= 0 and p.grad is not None:
                rec[i]["g"] = p.grad.data_ptr()
                rec[i]["lr"] = groups[gi]["lr"]
                rec[i]["wd"] = groups[gi]["weight_decay"]
            else:
                rec[i]["g"] = 0
        d_rec.copy_(self.hosts[slot], non_blocking=True)
        if self.copied[slot] is None:
            self.copied[slot] = torch.cuda.Event()
        self.copied[slot].record()
        b1, b2 = groups[0]["betas"]
        opt.global_step += 1
        self._step_t.fill_(float(opt.global_step))
        _C.check(_C.lib().cosa_fused_adamw_ema(_C.ptr(d_rec), _C.ptr(self.d_chunks), self.n_chunks, float(b1), float(b2),
                                               float(groups[0]["eps"]), int(opt.global_step), self.momentum, _C.stream_ptr()),
                 "cosa_fused_adamw_ema")


# --------------------------------------------------------------------------------------------
# evaluation helpers  (utils/torch_helper.py:12-30 format_tabs, :61-90 AverageMeter, :140-148 compute_mAP)
# --------------------------------------------------------------------------------------------
def average_precision(labels, outputs):
    """Per-sample average precision over the class axis, on the device: [b,C] {0,1} labels and scores -> ([b] AP, [b] valid).

    What sklearn's average_precision_score returns for each row (utils/torch_helper.py:146): sum over the distinct score
    thresholds, descending, of (R_n - R_{n-1}) * P_n; tied scores form ONE threshold.  Rows without a positive are invalid
    (the reference skips them)."""
    y = labels.double()
    s, order = torch.sort(outputs.double(), dim=1, descending=True, stable=True)
    yt = torch.gather(y, 1, order)
    C = y.shape[1]
    tps = yt.cumsum(1)
    last = torch.ones_like(s, dtype=torch.bool)
    last[:, :-1] = s[:, 1:] != s[:, :-1]                                  # last element of every run of equal scores
    npos = y.sum(1, keepdim=True)
    prec = tps / torch.arange(1, C + 1, device=y.device, dtype=torch.float64)
    rec = tps / npos.clamp_min(1.0)
    # recall at the previous threshold: running max of the recall at `last` positions strictly before i
    rec_at = torch.where(last, rec, torch.zeros_like(rec))
    prev = torch.cat([torch.zeros_like(rec[:, :1]), torch.cummax(rec_at, dim=1).values[:, :-1]], dim=1)
    ap = (torch.where(last, (rec - prev) * prec, torch.zeros_like(rec))).sum(1)
    return ap, npos[:, 0] > 0


def compute_mAP(labels, outputs):
    """utils/torch_helper.py:140-148: list of per-sample APs (samples without positives skipped).  One device->host copy."""
    ap, valid = average_precision(labels, outputs)
    return [float(a) for a, v in zip(ap.tolist(), valid.tolist()) if v]


class AverageMeter:
    """utils/torch_helper.py:61-90"""

    def __init__(self, *keys):
        self._data = {k: [0.0, 0] for k in keys}

    def add(self, d):
        for k, v in d.items():
            e = self._data.setdefault(k, [0.0, 0])
            e[0] += v
            e[1] += 1

    def get(self, *keys):
        vals = [self._data[k][0] / self._data[k][1] for k in keys]
        return vals[0] if len(vals) == 1 else tuple(vals)

    def pop(self, key=None):
        if key is None:
            for k in self._data:
                self._data[k] = [0.0, 0]
            return None
        v = self.get(key)
        self._data[key] = [0.0, 0]
        return v


def format_tabs(scores, name_list, cat_list=None, getmIoU_list=True):
    """utils/torch_helper.py:12-30 without the texttable dependency: (table text, last column's mIoU, list of mIoUs), values in
    per cent rounded to 2 decimals; the mIoU row is the plain mean over ALL classes of the rounded per-class values, as there."""
    import numpy as np
    keys = list(scores[0]["iou"].keys())
    vals = np.round(np.array([list(sc["iou"].values()) for sc in scores]) * 100, 2)
    names = [str(cat_list[i]) if cat_list is not None else str(k) for i, k in enumerate(keys)]
    wid = max([len(n) for n in names] + [5])
    lines = ["| " + "Class".ljust(wid) + " | " + " | ".join(n.rjust(8) for n in name_list) + " |"]
    for i, n in enumerate(names):
        lines.append("| " + n.ljust(wid) + " | " + " | ".join(f"{v:8.2f}" for v in vals[:, i]) + " |")
    means = vals.mean(1)
    lines.append("| " + "mIoU".ljust(wid) + " | " + " | ".join(f"{v:8.2f}" for v in means) + " |")
    return "\n".join(lines), means[-1], list(means)


# --------------------------------------------------------------------------------------------
# checkpoints  (utils/torch_helper.py:101-117 save_best; main.py:401-412 finaleval's load)
# --------------------------------------------------------------------------------------------
class EMAtracker:
    """utils/torch_helper.py:90-99.  `update` also takes a device scalar (the adaptive thresholds never visit the host); a
    non-finite new value (a fit that left an outer component empty -- the reference would have raised) leaves X as it is."""

    def __init__(self, initial_value=0, decay=0.9):
        self.X = initial_value
        self.decay = decay

    def update(self, value):
        keep, mix = self.decay, 1 - self.decay
        if torch.is_tensor(value):
            x = torch.as_tensor(self.X, dtype=value.dtype, device=value.device)
            self.X = torch.where(torch.isfinite(value), x * keep + value * mix, x)
        else:
            self.X = self.X * keep + value * mix

    def get(self):
        return self.X


def save_best(output_dir, model, finish_epoch, result, args, s_or_t, comment=''):
    """Same file name and dict layout as the reference ({'s_or_t','model','epoch','args','result'}), written by rank 0 only
    (utils.save_on_master).  `model.state_dict()` has the reference's key names, so either code base reads the other's files."""
    import os
    import torch.distributed as dist
    path = os.path.join(str(output_dir), f'best_{comment}.pth')
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return path
    os.makedirs(str(output_dir), exist_ok=True)
    model = getattr(model, "module", model)                      # unwrap DistributedDataParallel
    torch.save({'s_or_t': s_or_t, 'model': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'epoch': finish_epoch,
                'args': args, 'result': result}, path)
    return path


def load_best(model, path, strict=True):
    """main.py:410-412: `ckpt["model"]` into the network (strict by default, as there); returns the checkpoint dict."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    getattr(model, "module", model).load_state_dict(ckpt["model"], strict=strict)
    return ckpt
