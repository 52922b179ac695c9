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
    so2 = os.path.join(_HERE, "liboracle_d2.so")
    if force or not os.path.exists(so2) or os.path.getmtime(so2) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_d2.so"], stdout=subprocess.DEVNULL)
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
        vp = ctypes.c_void_p      # evaluation-path entry points take optional (NULL-able) arrays: raw addresses
        L.orc_resize_bilinear.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        L.orc_cam_to_label.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp,
                                       ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp]
        L.orc_eval_labels.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp]
        L.orc_confusion.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp]
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


_LIB2 = None


def gaussian_filter_d2(ins, H, W, sigmaxy):
    """position-only permutohedral filter (2-D lattice: the same C code as the bilateral filter, built with -DORC_PD=2): ins [K, H, W]"""
    global _LIB2
    if _LIB2 is None:
        build()
        _LIB2 = ctypes.CDLL(os.path.join(_HERE, "liboracle_d2.so"))
        _LIB2.orc_bilateralfilter_batch.restype = ctypes.c_int
        _LIB2.orc_bilateralfilter_batch.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_float, ctypes.c_float, _i32p]
    ins = _c(ins)
    K = ins.shape[0]
    outs = np.zeros_like(ins).reshape(-1)
    M = np.zeros(1, np.int32)
    dummy = np.zeros(3 * H * W, np.float32)
    if _LIB2.orc_bilateralfilter_batch(dummy, ins.reshape(-1), outs, 1, K, H, W, 1.0, float(sigmaxy), M):
        raise RuntimeError("oracle: lattice key out of packable range")
    return outs.reshape(K, H, W), int(M[0])


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


# ---- evaluation path (SURVEY f-1) ---------------------------------------------------------------------------------
def _a(x):
    return None if x is None else x.ctypes.data


def resize_bilinear(p, H, W):
    p = _c(p)
    C, h, w = p.shape
    out = np.empty((C, H, W), np.float32)
    lib().orc_resize_bilinear(_a(p), C, h, w, H, W, _a(out))
    return out


def cam_to_label(cam, cls_label, img_box=None, bkg_thre=0.5, high_thre=0.7, low_thre=0.25, ignore_mid=False, ignore_index=255):
    """utils/seg_helper.py:515-546: label map when img_box is None, else (valid_cam, label)."""
    cam = _c(cam)
    B, C, H, W = cam.shape
    lab = np.empty((B, H, W), np.int64)
    cl = _c(cls_label) if cls_label is not None else None
    bx = _c(img_box, np.int32) if img_box is not None else None
    vc = np.empty_like(cam) if img_box is not None else None
    lib().orc_cam_to_label(_a(cam), _a(cl), B, C, H, W, bkg_thre, _a(bx), int(ignore_mid), high_thre, low_thre, ignore_index,
                           _a(lab), _a(vc))
    return lab if img_box is None else (vc, lab)


def eval_labels(cam, seg, cls_label, H, W, bkg_thre=0.5):
    """one image: (S,S) CAM [C,S,S] + logits [C+1,S,S] -> uint8 label maps (cam, seg raw, seg validated) at (H,W)"""
    cam, seg, cl = _c(cam), _c(seg), _c(cls_label)
    C, S, _ = cam.shape
    outs = [np.empty((H, W), np.uint8) for _ in range(3)]
    lib().orc_eval_labels(_a(cam), _a(seg), _a(cl), C, S, H, W, bkg_thre, *[_a(o) for o in outs])
    return outs


def confusion(gts, preds, nc, pseudo=False):
    hist = np.zeros((nc, nc), np.int64)
    for g, p in zip(gts, preds):
        g, p = _c(g, np.uint8).ravel(), _c(p, np.uint8).ravel()
        lib().orc_confusion(_a(g), _a(p), g.size, nc, int(pseudo), _a(hist))
    return hist


def scores_from_hist(hist):
    """utils/evaluation.py:21-35 from an accumulated confusion matrix"""
    hist = hist.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
    valid = hist.sum(axis=1) > 0
    return {"pAcc": acc, "mAcc": acc_cls, "miou": np.nanmean(iu[valid]), "iou": dict(zip(range(hist.shape[0]), iu))}


def average_precision(y_true, y_score):
    """sklearn.metrics.average_precision_score for one sample (binary relevance over classes): AP = sum_n (R_n - R_{n-1}) P_n over
    the distinct score thresholds, descending (tied scores form one threshold).  torch_helper.py:140-148 calls it per image."""
    y_true, y_score = np.asarray(y_true, np.float64), np.asarray(y_score, np.float64)
    order = np.argsort(-y_score, kind="mergesort")
    ys, yt = y_score[order], y_true[order]
    last = np.r_[np.where(np.diff(ys))[0], ys.size - 1]          # last index of every run of equal scores
    tps = np.cumsum(yt)[last]
    prec = tps / (last + 1.0)
    rec = tps / yt.sum()
    return float(np.sum(np.diff(np.r_[0.0, rec]) * prec))
