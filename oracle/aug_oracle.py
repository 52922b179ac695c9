"""CPU restatement of the training input pipeline (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py may import it).

Reference: dataloaders/voc.py:262-275 `VOC12ClsDatasetNew.__transforms` -> dataloaders/transforms.py:52-77 (random_scaling,
_img_rescaling), :104-120 (random_fliplr), :150-202 (random_crop), :10-28 (GaussianBlur), dataloaders/randaug.py (OneOf of nine
ops), then torchvision ToTensor + Normalize (voc.py:247-250).  The pixel arithmetic lives in third-party libraries that are not
part of the reference tree: Pillow (resize BILINEAR, GaussianBlur, ImageOps, ImageEnhance; this image has 12.2.0), mmcv
(`solarize`: where(img < thr, img, 255 - img)) and torchvision (ToTensor: x/255 in float32; Normalize: (x - mean)/std in float32).
Their published algorithms are restated here in integer / float32 numpy:

  resize      two passes (horizontal, then vertical, 8-bit intermediate); per output index: centre = (i + .5)*scale, triangle
              weights over [centre - support, centre + support) normalised in double, quantised to 22-bit fixed point,
              sum + 2^21 >> 22, clipped
  blur        box radius from sigma (Gwosdek et al., float32 arithmetic), three horizontal then three vertical box passes, each
              (sum(2r+1 taps)*ww + (two outer taps)*fw + 2^23) >> 24 with edge replication and 8-bit rounding per pass
  autocontrast / equalize / posterize / solarize    per-channel look-up tables from the histogram
  Color / Contrast / Brightness   blend(degenerate, image, v) = trunc(d + v*(x - d)) in float32, clipped when v is outside [0,1];
              degenerate = ITU-R 601 luma (19595, 38470, 7471; +2^15 >> 16) / its rounded mean / black
  Sharpness   degenerate = 3x3 SMOOTH (1,1,1;1,5,1;1,1,1)/13 in float32 with +0.5 and truncation, border pixels copied

Pinned by tests/golden/augment.npz (outputs of the reference's own transforms.py / randaug.py functions under fixed seeds; the
solarize cases go through this file's restatement of mmcv.solarize and are marked) and, in tests, against Pillow itself.
The random draws replicate the reference's call order on Python's `random` and numpy's global generator.
"""
import math
import random

import numpy as np

f32 = np.float32
PRECISION_BITS = 32 - 8 - 2
MEAN = np.array([0.485, 0.456, 0.406], f32)
STD = np.array([0.229, 0.224, 0.225], f32)
OPS = ["identity", "autocontrast", "equalize", "solarize", "color", "contrast", "brightness", "sharpness", "posterize"]


# ---- Pillow: Image.resize(size, BILINEAR) ------------------------------------------------------------------------------
def resize_coeffs(in_size, out_size):
    """bounds [out,2] (first source index, count) and fixed-point weights [out, ksize]"""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    bounds, kk = resize_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += img[xmin + x].astype(np.int64) * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_bilinear(img, new_w, new_h):
    h, w, _ = img.shape
    t = _resample_axis0(img.transpose(1, 0, 2), new_w).transpose(1, 0, 2) if new_w != w else img
    return _resample_axis0(t, new_h) if new_h != h else t


# ---- Pillow: ImageFilter.GaussianBlur(radius) --------------------------------------------------------------------------
def box_radius(radius, passes=3):
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(2 * l + 1) * f32(f32(l * f32(l + 1)) - f32(3 * sigma2)))
    a = f32(a / f32(6 * f32(sigma2 - f32(f32(l + 1) * f32(l + 1)))))
    return f32(l + a)


def box_weights(fr):
    """(integer radius, ww, fw) of one box pass with float32 radius fr"""
    fr = f32(fr)
    radius = int(fr)
    ww = int(f32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def _box_pass_axis1(img, radius, ww, fw):
    W = img.shape[1]
    x = np.arange(W)
    acc = np.zeros(img.shape, np.int64)
    for d in range(-radius, radius + 1):
        acc += img[:, np.clip(x + d, 0, W - 1)]
    far = img[:, np.clip(x - radius - 1, 0, W - 1)].astype(np.int64) + img[:, np.clip(x + radius + 1, 0, W - 1)]
    bulk = (acc * ww + far * fw) & 0xffffffff
    return (((bulk + (1 << 23)) & 0xffffffff) >> 24).astype(np.uint8)


def gaussian_blur(img, radius):
    fr = box_radius(radius)
    if fr == 0:
        return img.copy()
    r, ww, fw = box_weights(fr)
    out = img
    for _ in range(3):
        out = _box_pass_axis1(out, r, ww, fw)
    t = out.transpose(1, 0, 2)
    for _ in range(3):
        t = _box_pass_axis1(t, r, ww, fw)
    return np.ascontiguousarray(t.transpose(1, 0, 2))


# ---- the nine strong ops (randaug.py:62-121) -----------------------------------------------------------------------------
def _hist(ch):
    return np.bincount(ch.ravel(), minlength=256)


def autocontrast_lut(h):
    nz = np.nonzero(h)[0]
    lo, hi = int(nz[0]), int(nz[-1])
    if hi <= lo:
        return np.arange(256, dtype=np.uint8)
    scale = 255.0 / (hi - lo)
    offset = -lo * scale
    return np.array([min(max(int(ix * scale + offset), 0), 255) for ix in range(256)], np.uint8)


def equalize_lut(h):
    histo = h[h > 0]
    if len(histo) <= 1:
        return np.arange(256, dtype=np.uint8)
    step = (int(histo.sum()) - int(histo[-1])) // 255
    if not step:
        return np.arange(256, dtype=np.uint8)
    n = step // 2
    lut = []
    for i in range(256):
        lut.append(min(n // step, 255))
        n += int(h[i])
    return np.array(lut, np.uint8)


def luma(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(a, b, alpha):
    alpha = f32(alpha)
    a = a.astype(np.int32)
    b = b.astype(np.int32)
    t = (a.astype(f32) + alpha * (b - a).astype(f32)).astype(f32)
    if 0 <= alpha <= 1:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def smooth3x3(img):
    k = (np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], f32) / f32(13)).astype(f32)
    out = img.copy()
    if img.shape[0] < 3 or img.shape[1] < 3:
        return out
    x = img.astype(f32)
    acc = np.full((img.shape[0] - 2, img.shape[1] - 2, 3), f32(0.5), f32)
    for r, kk in zip((x[2:], x[1:-1], x[:-2]), (k[0:3], k[3:6], k[6:9])):
        s = (r[:, :-2] * kk[0]).astype(f32)
        s = (s + r[:, 1:-1] * kk[1]).astype(f32)
        s = (s + r[:, 2:] * kk[2]).astype(f32)
        acc = (acc + s).astype(f32)
    out[1:-1, 1:-1] = np.where(acc <= 0, 0, np.where(acc >= 255, 255, acc.astype(np.int32)))
    return out


def enhance_factor(magnitude):
    return float(magnitude) * 1.8 / 10 + 0.1          # randaug.py:81-87 (_enhancer_impl, PARAMETER_MAX = 10)


def strong_op(img, op, magnitude):
    """img uint8 [H,W,3]; op index into OPS; magnitude 1..9 (np.random.randint(1, 10))"""
    name = OPS[op]
    if name == "identity":
        return img.copy()
    if name == "autocontrast":
        return np.stack([autocontrast_lut(_hist(img[..., c]))[img[..., c]] for c in range(3)], -1)
    if name == "equalize":
        return np.stack([equalize_lut(_hist(img[..., c]))[img[..., c]] for c in range(3)], -1)
    if name == "solarize":                               # mmcv.solarize(img, thr), thr = min(int(magnitude*256/10), 255)
        thr = min(int(magnitude * 256 / 10), 255)
        return np.where(img < thr, img, 255 - img).astype(np.uint8)
    if name == "posterize":                              # ImageOps.posterize(img, 4 - int(magnitude*4/10))
        bits = 4 - int(magnitude * 4 / 10)
        return img & np.uint8((~(2 ** (8 - bits) - 1)) & 0xff)
    v = enhance_factor(magnitude)
    if name == "color":
        return blend(np.repeat(luma(img)[..., None], 3, 2), img, v)
    if name == "contrast":
        h = _hist(luma(img))
        mean = int((np.arange(256) * h).sum() / h.sum() + 0.5)
        return blend(np.full_like(img, mean), img, v)
    if name == "brightness":
        return blend(np.zeros_like(img), img, v)
    if name == "sharpness":
        return blend(smooth3x3(img), img, v)
    raise ValueError(name)


# ---- torchvision ToTensor + Normalize ---------------------------------------------------------------------------------------
def normalize(u8):
    x = (u8.astype(f32) / f32(255)).astype(f32)
    return np.ascontiguousarray(((x - MEAN) / STD).astype(f32).transpose(2, 0, 1))


# ---- the random draws, in the reference's order ----------------------------------------------------------------------------
def draw_params(h, w, crop_size=448, scale_range=(0.5, 2.0), blur_p=0.5, radius_range=(0.1, 2.0)):
    """One image's draws from Python's `random` and numpy's global generator in the order of voc.py:262-275."""
    p = {}
    ratio = random.uniform(*scale_range)                                   # transforms.py:57
    p["new_w"], p["new_h"] = int(ratio * w), int(ratio * h)                 # transforms.py:66
    p["flip"] = bool(random.random() > 0.5)                                # transforms.py:105-109
    nh, nw = p["new_h"], p["new_w"]
    H, W = max(crop_size, nh), max(crop_size, nw)
    p["H_pad"] = int(np.random.randint(H - nh + 1))                        # transforms.py:163-164
    p["W_pad"] = int(np.random.randint(W - nw + 1))
    p["H_start"] = random.randrange(0, H - crop_size + 1, 1)               # transforms.py:172-175
    p["W_start"] = random.randrange(0, W - crop_size + 1, 1)
    p["blur"] = bool(random.random() <= blur_p)                            # transforms.py:20
    p["radius"] = random.uniform(*radius_range) if p["blur"] else 0.0      # transforms.py:26
    p["op"] = int(np.random.choice(len(OPS)))                              # randaug.py:129 (choice over the nine transforms)
    np.random.random()                                                     # randaug.py:46 (prob = 1.0: drawn, always true)
    p["magnitude"] = int(np.random.randint(1, 10))                         # randaug.py:49
    p["img_box"] = np.asarray([max(p["H_pad"] - p["H_start"], 0), min(crop_size, nh + p["H_pad"] - p["H_start"]),
                               max(p["W_pad"] - p["W_start"], 0), min(crop_size, nw + p["W_pad"] - p["W_start"])], np.int16)
    return p


def apply(image, p, crop_size=448):
    """image uint8 [h,w,3] -> (crop before blur, weak image = after blur, strong image), all uint8 [crop,crop,3]"""
    im = resize_bilinear(image, p["new_w"], p["new_h"])
    if p["flip"]:
        im = im[:, ::-1]
    nh, nw = im.shape[:2]
    H, W = max(crop_size, nh), max(crop_size, nw)
    pad = np.zeros((H, W, 3), np.uint8)
    pad[p["H_pad"]:p["H_pad"] + nh, p["W_pad"]:p["W_pad"] + nw] = im
    crop = np.ascontiguousarray(pad[p["H_start"]:p["H_start"] + crop_size, p["W_start"]:p["W_start"] + crop_size])
    weak = gaussian_blur(crop, p["radius"]) if p["blur"] else crop
    return crop, weak, strong_op(weak, p["op"], p["magnitude"])
