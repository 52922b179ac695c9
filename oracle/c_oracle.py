"""ctypes binding to the plain-C oracle (oracle/cosa_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "cosa_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_bilateral.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_expf_export.restype = ctypes.c_float
        L.orc_expf_export.argtypes = [ctypes.c_float]
        L.orc_denormalize_img.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_cam_minmax_norm.argtypes = [_f32p, ctypes.c_int, ctypes.c_int]
        L.orc_par_pos_weights.argtypes = [_i32p, ctypes.c_int, _f32p]
        L.orc_par_affinity.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_int, _f32p]
        L.orc_par_propagate.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                        ctypes.c_int, ctypes.c_int]
        L.orc_par_forward.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                      ctypes.c_int, ctypes.c_int]
        L.orc_cam2mask.argtypes = [_f32p, _i32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, _i32p,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_float, _f32p]
        L.orc_lowres_softmax_image.restype = ctypes.c_int
        L.orc_lowres_softmax_image.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, _f32p, _i32p]
        L.orc_bilateralfilter_batch.restype = ctypes.c_int
        L.orc_bilateralfilter_batch.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_float, ctypes.c_float, _i32p]
        L.orc_dense_energy_forward.restype = ctypes.c_float
        L.orc_dense_energy_forward.argtypes = [_f32p, _f32p, _f32p, _u8p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _f32p]
        _LIB = L
    return _LIB


def _c(a, dt=np.float32):
    return np.ascontiguousarray(np.asarray(a), dtype=dt)


def expf(x):
    return lib().orc_expf_export(float(x))


def denormalize_img(img):
    img = _c(img)
    out = np.empty_like(img)
    B, _, H, W = img.shape
    lib().orc_denormalize_img(img, out, B, H, W)
    return out


def cam_minmax_norm(cam):
    cam = _c(cam).copy()
    b, c, h, w = cam.shape
    lib().orc_cam_minmax_norm(cam, b * c, h * w)
    return cam


def par_pos_weights(dilations):
    d = _c(dilations, np.int32)
    out = np.empty(len(d) * 8, np.float32)
    lib().orc_par_pos_weights(d, len(d), out)
    return out


def par_affinity(img, dilations):
    img = _c(img)
    d = _c(dilations, np.int32)
    _, h, w = img.shape
    aff = np.empty((len(d) * 8, h, w), np.float32)
    lib().orc_par_affinity(img, h, w, d, len(d), aff)
    return aff


def par_forward(img, masks, dilations, num_iter):
    """img [3,h,w], masks [K,h,w] -> refined [K,h,w]"""
    img = _c(img)
    masks = _c(masks).copy()
    d = _c(dilations, np.int32)
    K, h, w = masks.shape
    lib().orc_par_forward(img, masks, K, h, w, d, len(d), int(num_iter))
    return masks


def cam2mask(images, boxes, cams, labels, thr_hi, thr_lo, downscale=2, par=None, ignore_index=255):
    """par = None or (dilations, num_iter).  cams are RAW (cam_validation folded in)."""
    cams = _c(cams)
    labels = _c(labels)
    boxes = _c(boxes, np.int32)
    B, C, S, _ = cams.shape
    images = _c(images) if images is not None else np.zeros((B, 3, S, S), np.float32)
    mask = np.empty((B, S, S), np.float32)
    if par is None:
        d = np.zeros(1, np.int32)
        nd, T, use = 0, 0, 0
    else:
        d = _c(par[0], np.int32)
        nd, T, use = len(d), int(par[1]), 1
    lib().orc_cam2mask(images, boxes, cams, labels, B, C, S, float(thr_hi), float(thr_lo), int(downscale),
                       use, d, nd, T, float(ignore_index), mask)
    return mask


def lowres_softmax_image(cam, label, downscale, thr):
    cam = _c(cam)
    label = _c(label)
    C, S, _ = cam.shape
    s = S // downscale if downscale else S
    out = np.zeros((C + 1, s, s), np.float32)
    act = np.zeros(C + 1, np.int32)
    K = lib().orc_lowres_softmax_image(cam, label, C, S, int(downscale), float(thr), out, act)
    return out[:K].copy(), act[:K].copy()


def bilateralfilter_batch(images, ins, N, K, H, W, sigmargb, sigmaxy):
    images = _c(images).reshape(-1)
    ins = _c(ins).reshape(-1)
    outs = np.zeros_like(ins)
    M = np.zeros(N, np.int32)
    rc = lib().orc_bilateralfilter_batch(images, ins, outs, N, K, H, W, float(sigmargb), float(sigmaxy), M)
    if rc:
        raise RuntimeError("oracle: lattice key out of packable range")
    return outs.reshape(N, K, H, W), M


def dense_energy_forward(images, seg, roi, unlabel, sigmargb, sigmaxy):
    images = _c(images)
    seg = _c(seg)
    roi = _c(roi)
    unlabel = _c(unlabel, np.uint8)
    N, K, H, W = seg.shape
    AS = np.zeros_like(seg)
    loss = lib().orc_dense_energy_forward(images, seg, roi, unlabel, N, K, H, W, float(sigmargb), float(sigmaxy), AS)
    return loss, AS


# --------------------------------------------------------------------------------------------
# oracle/_ref: the reference's own C++ (compiled from /root/reference by oracle/Makefile).
# The C++ functions are not extern "C"; bind their Itanium-mangled names directly.
# --------------------------------------------------------------------------------------------
def ref_lib():
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_bilateral.so")
        if not os.path.exists(p):
            return None
        R = ctypes.CDLL(p)
        fn = getattr(R, "_Z21bilateralfilter_batchPfiS_iS_iiiiiff")
        fn.restype = None
        fn.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                       ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
        R.bilateralfilter_batch = fn
        _REF = R
    return _REF


def ref_bilateralfilter_batch(images, ins, outs, N, K, H, W, sigmargb, sigmaxy):
    """Same 9-argument call as the reference's SWIG module (utils/seg_helper.py:887)."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libref_bilateral.so not built")
    images = _c(images).reshape(-1)
    ins = _c(ins).reshape(-1)
    assert outs.dtype == np.float32 and outs.flags.c_contiguous
    R.bilateralfilter_batch(images, images.size, ins, ins.size, outs.reshape(-1), outs.size, N, K, H, W,
                            float(sigmargb), float(sigmaxy))
