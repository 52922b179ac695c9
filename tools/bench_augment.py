"""Input pipeline throughput (SURVEY f-2): the device augmenter vs the same pipeline on host cores.
usage: python tools/bench_augment.py [batch=16] [crop=448]
  device   : DeviceAugmenter on decoded 375x500 images (H2D of the raw bytes + 13 launches), images/s incl. host packing
  cpu      : oracle's restatement is not the fair baseline here -- the reference's path IS Pillow -- so the CPU number is
             Pillow itself driven in the reference's order (resize, flip, pad/crop, GaussianBlur, strong op, normalise), 1 core."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

from cosa_amd.dataloaders import DeviceAugmenter, draw_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 448
rng = np.random.default_rng(0)
images = []
for i in range(B):
    h, w = [(375, 500), (500, 375), (333, 500), (500, 500)][i % 4]
    small = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    images.append(np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC)))
random.seed(0)
np.random.seed(0)
params = [draw_params(im.shape[0], im.shape[1], crop_size=S) for im in images]
aug = DeviceAugmenter(S)
for _ in range(3):
    aug(images, params)
torch.cuda.synchronize()
t = time.perf_counter()
n = 20
for _ in range(n):
    out = aug(images, params)
torch.cuda.synchronize()
dev_s = (time.perf_counter() - t) / n
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    out = aug(images, params)
e.record()
torch.cuda.synchronize()
gpu_ms = a.elapsed_time(e) / n


def pillow_path(img, p):
    im = np.asarray(Image.fromarray(img).resize((p["new_w"], p["new_h"]), resample=Image.BILINEAR)).astype(np.float32)
    if p["flip"]:
        im = np.fliplr(im)
    H, W = max(S, p["new_h"]), max(S, p["new_w"])
    pad = np.zeros((H, W, 3), np.uint8)
    pad[p["H_pad"]:p["H_pad"] + p["new_h"], p["W_pad"]:p["W_pad"] + p["new_w"]] = im
    pil = Image.fromarray(pad[p["H_start"]:p["H_start"] + S, p["W_start"]:p["W_start"] + S])
    if p["blur"]:
        pil = pil.filter(ImageFilter.GaussianBlur(radius=p["radius"]))
    v = float(p["magnitude"]) * 1.8 / 10 + 0.1
    st = [lambda q: q, ImageOps.autocontrast, ImageOps.equalize, lambda q: ImageOps.solarize(q, min(int(p["magnitude"] * 256 / 10), 255)),
          lambda q: ImageEnhance.Color(q).enhance(v), lambda q: ImageEnhance.Contrast(q).enhance(v),
          lambda q: ImageEnhance.Brightness(q).enhance(v), lambda q: ImageEnhance.Sharpness(q).enhance(v),
          lambda q: ImageOps.posterize(q, 4 - int(p["magnitude"] * 4 / 10))][p["op"]](pil)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    return [((np.asarray(x).astype(np.float32) / 255 - mean) / std).transpose(2, 0, 1) for x in (pil, st)]


t = time.perf_counter()
reps = 3
for _ in range(reps):
    for im, p in zip(images, params):
        pillow_path(im, p)
cpu_s = (time.perf_counter() - t) / reps
print(json.dumps({"batch": B, "crop": S, "device_images_per_s": round(B / dev_s, 1), "device_gpu_ms_per_batch": round(gpu_ms, 3),
                  "device_wall_ms_per_batch": round(dev_s * 1e3, 3), "pillow_images_per_s_1core": round(B / cpu_s, 1)}))
