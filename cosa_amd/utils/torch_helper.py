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
