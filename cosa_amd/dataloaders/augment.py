"""Training-time augmentation with the reference's contract, split where the hardware wants it.

Reference: dataloaders/voc.py:219-305 (`VOC12ClsDatasetNew`: `__transforms` -> (weak, strong, img_box)),
dataloaders/transforms.py:10-28,52-77,104-120,150-202, dataloaders/randaug.py:58-130.

  host    `draw_params(h, w)`: the random draws of one image, from Python's `random` and numpy's global generator in the
          reference's call order (so a seeded run picks the same scale / flip / pads / crop / blur / op / magnitude), plus the
          integer consequences (sizes, img_box).  JPEG decode stays with the caller.
  device  `DeviceAugmenter(images, params)`: one H2D copy of the decoded uint8 images + parameter records, then
          cosa_augment_batch (csrc/aug_kernels.hip): Pillow-exact resize restricted to the crop window, flip / pad / crop as
          index arithmetic, Pillow-exact GaussianBlur, the nine strong ops, ToTensor + Normalize -> wimg, simg on the GPU.

Returns exactly what the reference's loader yields per batch after collation: wimg [b,3,S,S], simg [b,3,S,S] float32 and
img_box [b,4] int16 (rows [h0,h1), columns [w0,w1) of the crop that hold image).
"""
import math
import random

import numpy as np
import torch

from .. import _C

OPS = ["Identity", "AutoContrast", "RandEqualize", "RandSolarize", "RandColor", "RandContrast", "RandBrightness", "RandSharpness",
       "RandPosterize"]                                                     # voc.py:252-261 (order = np.random.choice index)
_MAX_TAPS = 16                                                              # kAugMaxTaps in aug_kernels.hip
_REC_INTS = 26


def draw_params(h, w, crop_size=448, scale_range=(0.5, 2.0), blur_p=0.5, radius_range=(0.1, 2.0)):
    """The draws of `__transforms` for one (h, w) image, in order: random.uniform (scale), random.random (flip),
    np.random.randint x2 (padding offsets), random.randrange x2 (crop window), random.random [+ random.uniform] (blur),
    np.random.choice (op), np.random.random, np.random.randint(1, 10) (magnitude)."""
    ratio = random.uniform(scale_range[0], scale_range[1])
    new_w, new_h = int(ratio * w), int(ratio * h)
    if new_w < 1 or new_h < 1:
        raise ValueError("draw_params: image vanishes at this scale")
    flip = random.random() > 0.5
    H, W = max(crop_size, new_h), max(crop_size, new_w)
    H_pad = int(np.random.randint(H - new_h + 1))
    W_pad = int(np.random.randint(W - new_w + 1))
    H_start = random.randrange(0, H - crop_size + 1, 1)
    W_start = random.randrange(0, W - crop_size + 1, 1)
    blur = random.random() <= blur_p
    radius = random.uniform(radius_range[0], radius_range[1]) if blur else 0.0
    op = int(np.random.choice(len(OPS)))
    np.random.random()                                                      # RandAug.__call__: `np.random.random() < prob`, prob = 1
    magnitude = int(np.random.randint(1, 10))
    box = (max(H_pad - H_start, 0), min(crop_size, new_h + H_pad - H_start), max(W_pad - W_start, 0),
           min(crop_size, new_w + W_pad - W_start))
    return dict(h=h, w=w, new_h=new_h, new_w=new_w, flip=bool(flip), H_pad=H_pad, W_pad=W_pad, H_start=H_start, W_start=W_start,
                blur=bool(blur), radius=radius, op=op, magnitude=magnitude, img_box=np.asarray(box, np.int16))


def _box_weights(radius):
    """Pillow's GaussianBlur -> (integer box radius, centre weight, outer weight) in its float32 arithmetic"""
    f = np.float32
    r = f(radius)
    sigma2 = f(r * r / f(3))
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = f(f(2 * l + 1) * f(f(l * f(l + 1)) - f(3 * sigma2)))
    a = f(a / f(6 * f(sigma2 - f(f(l + 1) * f(l + 1)))))
    fr = f(l + a)
    br = int(fr)
    ww = int(f(f(1 << 24) / f(fr * f(2) + f(1))))
    return br, ww, ((1 << 24) - (br * 2 + 1) * ww) // 2


def _first_source_index(out_index, in_size, out_size):
    scale = in_size / out_size
    support = max(scale, 1.0)
    center = (out_index + 0.5) * scale
    return max(int(center - support + 0.5), 0), min(int(center + support + 0.5), in_size)


def _record(p, raw_off, crop_size):
    h, w, nh, nw = p["h"], p["w"], p["new_h"], p["new_w"]
    for a, b in ((h, nh), (w, nw)):
        if int(math.ceil(max(a / b, 1.0))) * 2 + 1 > _MAX_TAPS:
            raise ValueError("DeviceAugmenter: down-scaling beyond 7x is not supported")
    b0, b1, b2, b3 = (int(v) for v in p["img_box"])
    if b1 > b0:                                                             # source rows the vertical pass reads (one row of slack)
        ys0, ys1 = b0 + p["H_start"] - p["H_pad"], b1 - 1 + p["H_start"] - p["H_pad"]
        lo = ys0 if nh == h else _first_source_index(ys0, h, nh)[0]
        hi = ys1 + 1 if nh == h else _first_source_index(ys1, h, nh)[1]
        lo, hi = max(lo - 1, 0), min(hi + 1, h)
    else:
        lo, hi = 0, 1
    br, ww, fw = _box_weights(p["radius"]) if p["blur"] else (0, 1 << 24, 0)
    alpha = np.float32(float(p["magnitude"]) * 1.8 / 10 + 0.1)             # randaug.py:81-87
    rec = np.zeros(_REC_INTS, np.int32)
    rec[0:2] = np.array([raw_off], np.int64).view(np.int32)
    rec[2:6] = (h, w, nh, nw)
    rec[6] = int(p["flip"])
    rec[7:11] = (p["H_pad"], p["W_pad"], p["H_start"], p["W_start"])
    rec[11:15] = (b0, b1, b2, b3)
    rec[15:17] = (lo, hi - lo)
    rec[17:21] = (int(p["blur"]), br, ww, fw)
    rec[21:23] = (p["op"], p["magnitude"])
    rec[23] = alpha.view(np.int32)
    return rec, hi - lo


class DeviceAugmenter:
    """`aug(images, params) -> (wimg, simg, img_box)`: images = list of decoded uint8 [h,w,3] arrays, params = list of
    draw_params(...) dicts.  `debug=True` also returns the uint8 stages (crop, weak, strong) for parity tests."""

    def __init__(self, crop_size=448, device="cuda"):
        self.crop_size = crop_size
        self.device = torch.device(device)
        L = _C.lib()
        if L.cosa_augment_record_bytes() != _REC_INTS * 4:
            raise _C.CosaError("DeviceAugmenter: record layout mismatch with libcosa_hip")

    def __call__(self, images, params, debug=False):
        B, S, dev = len(images), self.crop_size, self.device
        if B == 0 or len(params) != B:
            raise ValueError("DeviceAugmenter: one parameter dict per image")
        offs, total = [], 0
        for im, p in zip(images, params):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 or im.shape[:2] != (p["h"], p["w"]):
                raise ValueError("DeviceAugmenter: images must be uint8 [h,w,3] matching their parameters")
            offs.append(total)
            total += (im.size + 255) // 256 * 256
        raw = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        rawn = raw.numpy()
        recs = torch.empty((B, _REC_INTS), dtype=torch.int32, pin_memory=True)
        max_rows = 1
        for i, (im, p) in enumerate(zip(images, params)):
            rawn[offs[i]:offs[i] + im.size] = np.ascontiguousarray(im).reshape(-1)
            rec, rows = _record(p, offs[i], S)
            recs[i] = torch.from_numpy(rec)
            max_rows = max(max_rows, rows)
        raw_d = raw.to(dev, non_blocking=True)
        recs_d = recs.to(dev, non_blocking=True)
        wimg = torch.empty((B, 3, S, S), device=dev, dtype=torch.float32)
        simg = torch.empty((B, 3, S, S), device=dev, dtype=torch.float32)
        stages = [torch.empty((B, S, S, 3), device=dev, dtype=torch.uint8) for _ in range(3)] if debug else [None] * 3
        L = _C.lib()
        ws = _C.workspace(L.cosa_augment_workspace_bytes(B, S, max_rows), dev, "augment")
        _C.check(L.cosa_augment_batch(_C.ptr(raw_d), _C.ptr(recs_d), B, S, max_rows, _C.ptr(wimg), _C.ptr(simg), _C.ptr(stages[0]),
                                      _C.ptr(stages[1]), _C.ptr(stages[2]), _C.ptr(ws), ws.numel(), _C.stream_ptr()),
                 "cosa_augment_batch")
        img_box = torch.from_numpy(np.stack([p["img_box"] for p in params]))
        if debug:
            return wimg, simg, img_box, stages
        return wimg, simg, img_box
