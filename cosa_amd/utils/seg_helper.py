"""utils.seg_helper -- hot-path functions of the reference module, backed by the HIP kernels.

Same names, arguments and error behaviour as the reference (utils/seg_helper.py); every function
cites the lines it replaces.  All tensors live on the GPU; nothing here falls back to the CPU.
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _C, nn_ops
from ..models.PAR import PAR


def _refresh_once(model):
    """bring the network's 16-bit weight shadows up to date once per multi-scale pass (see nn_ops.ensure_shadows)"""
    f = getattr(getattr(model, "module", model), "refresh_shadows", None)
    if f is not None:
        f()


# --------------------------------------------------------------------------------------------
# multi_scale_camseg  (utils/seg_helper.py:232-275)
# --------------------------------------------------------------------------------------------
def resize_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for NCHW fp32 on the GPU (own kernel: the teacher's input rescale,
    utils/seg_helper.py:247-250, was the last ATen kernel inside the captured teacher pass)"""
    x = x.contiguous().float()
    b, c, h, w = x.shape
    out = torch.empty((b, c, int(size[0]), int(size[1])), device=x.device, dtype=torch.float32)
    _C.check(_C.lib().cosa_resize_bilinear(_C.ptr(x), _C.ptr(out), b * c, h, w, int(size[0]), int(size[1]), _C.stream_ptr()), "cosa_resize_bilinear")
    return out


def _flip_merge_upsample(src, dst, B, S, mode, accumulate, active=None, prev_active=None):
    src = src.contiguous().float()
    _, C, h, w = src.shape
    if prev_active is not None:
        _C.check(_C.lib().cosa_cam_flip_merge_upsample_reuse(_C.ptr(src), _C.ptr(dst), B, C, h, w, S, mode, int(accumulate), _C.ptr(active),
                                                             _C.ptr(prev_active), _C.stream_ptr()), "cosa_cam_flip_merge_upsample_reuse")
        return
    _C.check(_C.lib().cosa_cam_flip_merge_upsample(_C.ptr(src), _C.ptr(dst), B, C, h, w, S, mode, int(accumulate),
                                                   _C.ptr(active), _C.stream_ptr()), "cosa_cam_flip_merge_upsample")


# CAM buffers of a training loop's teacher passes, kept from step to step: a plane that is absent from the image now and was absent
# last time is already zero, so only the planes that were live last time are cleared (80 COCO planes per image, ~3 live: 1 GB of zero
# stores per call otherwise).  The buffers and the `prev` map belong to the CALLER (CoSATrainer passes its own dict as `_buffers`):
# nothing is shared between trainers or with other callers in the process.
def _persistent_cams(store, b, C, h, w, device):
    key = (str(device), b, C, h, w)
    ent = store.get(key)
    if ent is None:
        ent = store[key] = {"cam": torch.zeros((b, C, h, w), device=device, dtype=torch.float32),
                            "aux": torch.zeros((b, C, h, w), device=device, dtype=torch.float32),
                            "prev": torch.zeros((b, C), device=device, dtype=torch.float32)}
    return ent


def cam_minmax_norm_(cam, active=None):
    """In place x -= min; x /= max + 1e-5 per (b,c) plane (utils/seg_helper.py:265-266,269-270)."""
    b, c, h, w = cam.shape
    ws = torch.empty(2 * b * c, device=cam.device, dtype=torch.int32)      # per-plane min / max keys (csrc/label_kernels.hip)
    _C.check(_C.lib().cosa_cam_minmax_norm_ws(_C.ptr(cam), b * c, h * w, _C.ptr(active), _C.ptr(ws), _C.stream_ptr()), "cosa_cam_minmax_norm_ws")
    return cam


def multi_scale_camseg(model, imgs, scales, _active_labels=None, _seg_scales=False, _buffers=None):
    """Teacher forward over scales x {orig, flip}; returns (cam, cam_aux, seg) at input size.

    utils/seg_helper.py:232-275.  Per scale one fused kernel does bilinear-up + un-flip + max/sum
    (+ReLU) + accumulation; the reference's quirk that cam_aux keeps ONLY the last scale
    (`cam_aux_list = [...]`, :258) is reproduced.  `_active_labels` ([b,C] image-level labels, optional): the CAM
    planes of absent classes are returned as zeros -- exactly what cam_validation makes of them in the very next
    call of the training loop (main.py:137) -- so the tail only touches the 1-4 live planes per image.
    `_seg_scales=True` returns the per-scale low-res seg outputs (a list of [2b,K,h_s,w_s]) in place of the summed
    full-resolution seg: the only consumer in the loop (cam_loss's targets) reads 4 pixels per 16x16 block of it
    (see cam_loss_targets), so the [b,K,S,S] tensor need not exist.
    `_buffers` (a dict owned by the caller, with `_active_labels` only): the returned cam / cam_aux then live in buffers kept in that dict
    from call to call (planes absent now and last time are not re-zeroed) -- they are VALID ONLY UNTIL THE NEXT CALL with the same dict
    and shapes, and must not be written in place by the caller.  Without it every call returns fresh tensors.
    """
    b, c, h, w = imgs.shape
    assert 1.0 in scales, 'scale 1.0 must be in scales'
    assert h == w, "square crops only"
    _C.require_cuda(imgs)
    cam = cam_aux = seg = None
    seg_list = []
    act = _active_labels.contiguous().float() if _active_labels is not None else None
    with torch.no_grad():
        scaled = [imgs if s == 1.0 else resize_bilinear(imgs, (int(s * h), int(s * w))) for s in scales]
        # cosa_amd networks can take all scales in one go (shared GEMM/LayerNorm launches across scales), the mirror images only as
        # im2col rows of the patch projection (flip_pairs)
        if getattr(model, "can_forward_multi", lambda _x: False)(scaled[0]):
            multi, inputs = model.forward_multi(scaled, flip_pairs=True, need_cls=False), None
        else:
            multi, inputs = None, [torch.cat([x_, x_.flip(-1)], dim=0) for x_ in scaled]
            _refresh_once(model)
        for si, s in enumerate(scales):
            with nn_ops.shadows_fresh():
                _, _, _, _seg, _cam, _cam_aux = multi[si] if multi is not None else model(inputs[si], cam_only=False)
            if cam is None:
                if act is not None and _buffers is not None:
                    keep = _persistent_cams(_buffers, b, _cam.shape[1], h, w, imgs.device)
                    cam, cam_aux, prev = keep["cam"], keep["aux"], keep["prev"]
                elif act is not None:
                    cam = torch.zeros((b, _cam.shape[1], h, w), device=imgs.device, dtype=torch.float32)
                    cam_aux = torch.zeros_like(cam)
                    prev = torch.zeros((b, _cam.shape[1]), device=imgs.device, dtype=torch.float32)
                else:
                    cam = torch.empty((b, _cam.shape[1], h, w), device=imgs.device, dtype=torch.float32)
                    cam_aux = torch.empty_like(cam)
                    prev = None
                seg = None if _seg_scales else torch.empty((b, _seg.shape[1], h, w), device=imgs.device, dtype=torch.float32)
            _flip_merge_upsample(_cam, cam, b, h, 0, si > 0, act, prev if si == 0 else None)
            if si == len(scales) - 1:                                   # only the last scale survives (:258)
                _flip_merge_upsample(_cam_aux, cam_aux, b, h, 0, False, act, prev)
            if _seg_scales:
                seg_list.append(_seg.contiguous().float())
            else:
                _flip_merge_upsample(_seg, seg, b, h, 1, si > 0)
        cam_minmax_norm_(cam, act)
        cam_minmax_norm_(cam_aux, act)
        if act is not None:
            prev.copy_(act.view_as(prev))
    return cam, cam_aux, (seg_list if _seg_scales else seg)


def multi_scale_camsegv3(model, imgs, scales, getcls=False, _per_image_cls=False):
    """Evaluation-time variant (utils/seg_helper.py:399-450; evaluation_engine.py:82-85 calls it with five scales x two flips):
    same fused tail as multi_scale_camseg, plus the classification logits summed over scales and over {orig, flip}
    (`cls_f_ += sum(cls_f, dim=0)`, :432-434).  cam_aux again keeps only the LAST scale (:426)."""
    b, c, h, w = imgs.shape
    assert 1.0 in scales, 'scale 1.0 must be in scales'
    assert h == w, "square inputs only (evaluation resizes to crop_size x crop_size first)"
    _C.require_cuda(imgs)
    cam = cam_aux = seg = None
    cls_f_ = cls_a_ = None
    with torch.no_grad():
        scaled = [imgs if s == 1.0 else resize_bilinear(imgs, (int(s * h), int(s * w))) for s in scales]
        if getattr(model, "can_forward_multi", lambda _x: False)(scaled[0]):
            multi, inputs = model.forward_multi(scaled, flip_pairs=True, need_cls=bool(getcls)), None
        else:
            multi, inputs = None, [torch.cat([x_, x_.flip(-1)], dim=0) for x_ in scaled]
            _refresh_once(model)
        for si, s in enumerate(scales):
            with nn_ops.shadows_fresh():
                cls_f, cls_a, _, _seg, _cam, _cam_aux = multi[si] if multi is not None else model(inputs[si], cam_only=False)
            if cam is None:
                cam = torch.empty((b, _cam.shape[1], h, w), device=imgs.device, dtype=torch.float32)
                cam_aux = torch.empty_like(cam)
                seg = torch.empty((b, _seg.shape[1], h, w), device=imgs.device, dtype=torch.float32)
            _flip_merge_upsample(_cam, cam, b, h, 0, si > 0)
            if si == len(scales) - 1:
                _flip_merge_upsample(_cam_aux, cam_aux, b, h, 0, False)
            _flip_merge_upsample(_seg, seg, b, h, 1, si > 0)
            if getcls and _per_image_cls:
                # several images per pass (evaluate's grouping): the reference's sum over {orig, flip} kept per image -> [b, C]
                cf, ca = cls_f.float().reshape(2, b, -1).sum(0), cls_a.float().reshape(2, b, -1).sum(0)
                cls_f_ = cf if cls_f_ is None else cls_f_ + cf
                cls_a_ = ca if cls_a_ is None else cls_a_ + ca
            elif getcls:
                cf, ca = cls_f.float().sum(0, keepdim=True), cls_a.float().sum(0, keepdim=True)
                cls_f_ = cf if cls_f_ is None else cls_f_ + cf
                cls_a_ = ca if cls_a_ is None else cls_a_ + ca
        cam_minmax_norm_(cam)
        cam_minmax_norm_(cam_aux)
    if getcls:
        return cam, cam_aux, seg, cls_f_, cls_a_
    return cam, cam_aux, seg


# --------------------------------------------------------------------------------------------
# cam_to_label / seg_validation / evaluation label maps  (utils/seg_helper.py:515-546, 581-591; evaluation_engine.py:96-126,198-200)
# --------------------------------------------------------------------------------------------
def cam_to_label(cam, cls_label, img_box=None, bkg_thre=None, high_thre=None, low_thre=None, ignore_mid=False, ignore_index=None):
    """utils/seg_helper.py:515-546: argmax of the class-validated CAM (+1), background where the max <= bkg_thre.
    Returns the int64 label map when `img_box` is None, else `(valid_cam, pseudo_label)` with the label confined to the boxes
    (ignore_index outside) and, if `ignore_mid`, the high/low threshold band marked ignore_index."""
    _C.require_cuda(cam)
    if bkg_thre is None:
        raise TypeError("cam_to_label: bkg_thre is required (the reference compares against it unconditionally)")
    cam = cam.contiguous().float()
    b, c, h, w = cam.shape
    cls = cls_label.contiguous().float() if cls_label is not None else None
    label = torch.empty((b, h, w), device=cam.device, dtype=torch.int64)
    boxes = valid = None
    if img_box is not None:
        boxes = _boxes_to_device(img_box, cam.device)
        if boxes.shape != (b, 4):
            raise ValueError("cam_to_label: img_box must be [b,4]")
        if ignore_index is None or (ignore_mid and (high_thre is None or low_thre is None)):
            raise TypeError("cam_to_label: ignore_index (and high_thre/low_thre with ignore_mid) are required with img_box")
        valid = torch.empty_like(cam)
    _C.check(_C.lib().cosa_cam_to_label(_C.ptr(cam), _C.ptr(cls), b, c, h, w, float(bkg_thre), _C.ptr(boxes), int(bool(ignore_mid)),
                                        float(high_thre or 0.0), float(low_thre or 0.0), int(ignore_index if ignore_index is not None else 255),
                                        _C.ptr(label), _C.ptr(valid), _C.stream_ptr()), "cosa_cam_to_label")
    return label if img_box is None else (valid, label)


def seg_validation(seg, cls_label):
    """utils/seg_helper.py:581-591: logits of the classes absent from the image-level label set to -1e5 (background kept)."""
    if cls_label is None:
        return seg
    b = seg.shape[0]
    present = torch.cat([torch.ones(b, 1, device=seg.device, dtype=torch.bool), cls_label != 0], dim=1)
    return torch.where(present[:, :, None, None], seg, torch.full((), -1e5, device=seg.device, dtype=seg.dtype))


def eval_label_maps(cam, seg, cls_label, size, bkg_thre):
    """One launch for evaluation_engine.py:96-126,198-200: `F.interpolate(cam, size)` -> cam_to_label(bkg_thre),
    `F.interpolate(seg, size)` -> argmax, seg_validation -> argmax, without the resized tensors.
    cam [b,C,S,S] and/or seg [b,C+1,S,S] -> uint8 maps [b,H,W]: (cam_label, pred_ps, pred_vd) (None for an absent input)."""
    H, W = int(size[0]), int(size[1])
    ref = cam if cam is not None else seg
    _C.require_cuda(ref, cls_label)
    b, S = ref.shape[0], ref.shape[-1]
    C = cls_label.shape[1]
    cam = cam.contiguous().float() if cam is not None else None
    seg = seg.contiguous().float() if seg is not None else None
    if (cam is not None and cam.shape != (b, C, S, S)) or (seg is not None and seg.shape != (b, C + 1, S, S)):
        raise ValueError("eval_label_maps: cam must be [b,C,S,S] and seg [b,C+1,S,S]")
    mk = lambda: torch.empty((b, H, W), device=ref.device, dtype=torch.uint8)
    lc = mk() if cam is not None else None
    lp, lv = (mk(), mk()) if seg is not None else (None, None)
    _C.check(_C.lib().cosa_eval_labels(_C.ptr(cam), _C.ptr(seg), _C.ptr(cls_label.contiguous().float()), b, C, S, H, W, float(bkg_thre),
                                       _C.ptr(lc), _C.ptr(lp), _C.ptr(lv), _C.stream_ptr()), "cosa_eval_labels")
    return lc, lp, lv


# --------------------------------------------------------------------------------------------
# cam_validation / cam2mask  (utils/seg_helper.py:547-551, 721-797)
# --------------------------------------------------------------------------------------------
def cam_validation(cam, cls_label):
    """utils/seg_helper.py:547-551 (broadcast multiply; no materialised repeat)."""
    return cam * cls_label[:, :, None, None]


def _boxes_to_device(img_boxes, device):
    if not torch.is_tensor(img_boxes):
        img_boxes = torch.as_tensor(img_boxes)
    return img_boxes.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


def cam2mask(images, img_boxes, cams, cls_labels, threshold_high, threshold_low, refine_model=None, ignore_index=255,
             downscale=2, _fold_validation=False):
    """CAM -> {0..C, 255} label mask [b,h,w] (float32), one fused launch sequence for the whole batch.

    utils/seg_helper.py:721-785 (+ _refine_cams :787-797).  `refine_model` may be None (reference
    default) or a cosa_amd.models.PAR.PAR instance (the reference's only refine model).
    `cams` are the validated CAMs as in the reference call order (main.py:137,158-166);
    `_fold_validation=True` lets the training loop pass raw CAMs and skip cam_validation's pass.
    """
    return cam2mask_multi(images, img_boxes, [cams], cls_labels, [threshold_high], [threshold_low], refine_model=refine_model,
                          ignore_index=ignore_index, downscale=downscale, _fold_validation=_fold_validation)[0]


def _cam2mask_generic(images, img_boxes, cams, cls_labels, threshold_high, threshold_low, refine_model, ignore_index, downscale):
    """utils/seg_helper.py:721-797 as written there (per-image loop, torch ops on the device), for what the fused kernels do not
    cover: an arbitrary callable `refine_model`, non-square crops, other downscale factors.  `cams` are the validated CAMs."""
    b, _, h, w = images.shape
    dev = cams.device
    size = [h // downscale, w // downscale] if downscale else None
    rs = (lambda t: F.interpolate(t, size=size, mode="bilinear", align_corners=False)) if downscale else (lambda t: t)
    _images = rs(images.float())
    plane = torch.ones((b, 1, h, w), device=dev)
    hi = torch.as_tensor(threshold_high, device=dev, dtype=torch.float32)
    lo = torch.as_tensor(threshold_low, device=dev, dtype=torch.float32)
    cams_h, cams_l = rs(torch.cat([plane * hi, cams], dim=1)), rs(torch.cat([plane * lo, cams], dim=1))
    with_bkg = torch.cat([torch.ones((b, 1), device=dev), cls_labels.float()], dim=1)
    out_h = torch.full((b, h, w), float(ignore_index), device=dev)
    out_l = out_h.clone()

    def refine(img, act, keys):
        r = refine_model(img, act) if refine_model is not None else act
        r = F.interpolate(r, size=(h, w), mode="bilinear", align_corners=False)
        return keys[r.argmax(dim=1)]

    boxes = torch.as_tensor(img_boxes).tolist()
    for i, (y0, y1, x0, x1) in enumerate(boxes):
        keys = torch.nonzero(with_bkg[i])[:, 0]
        out_h[i, y0:y1, x0:x1] = refine(_images[[i]], cams_h[i, keys].unsqueeze(0).softmax(dim=1), keys)[0, y0:y1, x0:x1].float()
        out_l[i, y0:y1, x0:x1] = refine(_images[[i]], cams_l[i, keys].unsqueeze(0).softmax(dim=1), keys)[0, y0:y1, x0:x1].float()
    mask = out_h.clone()
    mask[out_h == 0] = ignore_index
    mask[(out_h + out_l) == 0] = 0
    return mask


_BOX_SIZE_CACHE = {}       # (h, w, device, dtype) -> [h, h, w, w] on that device (cam2mask_multi: negative box bounds)


def cam2mask_multi(images, img_boxes, cams_list, cls_labels, thresholds_high, thresholds_low, refine_model=None, ignore_index=255,
                   downscale=2, _fold_validation=False):
    """cam2mask for several CAM sets of the SAME images (the training step's main and auxiliary CAMs, main.py:137-166) in
    one pass: the refine model's affinities are built once and streamed once per propagation step for all sets.  Returns
    a list of label maps, each bit-identical to a separate cam2mask call."""
    G = len(cams_list)
    if not (G >= 1 and len(thresholds_high) == G and len(thresholds_low) == G):
        raise ValueError("cam2mask_multi: one (high, low) threshold pair per CAM set")
    _C.require_cuda(cls_labels, *cams_list)
    b, _, h, w = images.shape
    generic = (refine_model is not None and not isinstance(refine_model, PAR)) or h != w or downscale not in (0, 2, None, False)
    if generic:
        # any other callable refine model (the reference accepts whatever `refine_model(images, cams)` returns,
        # utils/seg_helper.py:787-792), non-square crops and other downscale factors: the reference's per-image loop on the GPU
        if refine_model is not None and not callable(refine_model):
            raise TypeError("cam2mask: refine_model must be None or a callable (images[1,3,h,w], cams[1,K,h,w]) -> [1,K,h',w']")
        return [_cam2mask_generic(images, img_boxes, c.float() * cls_labels[:, :, None, None] if _fold_validation else c.float(),
                                  cls_labels, th, tl, refine_model, ignore_index, downscale)
                for c, th, tl in zip(cams_list, thresholds_high, thresholds_low)]
    downscale = 2 if downscale == 2 else 0
    cams_list = [c.contiguous().float() for c in cams_list]
    cls_labels = cls_labels.contiguous().float()
    C = cams_list[0].shape[1]
    for c in cams_list:
        if c.shape != (b, C, h, w) or cls_labels.shape != (b, C):
            raise ValueError("cam2mask: cams must be [b,C,h,w] at image size and cls_labels [b,C]")
    dev = cams_list[0].device
    # the reference slices with the box (`mask[y0:y1, x0:x1]`, seg_helper.py:776-777), so negative bounds count from the end -- evaluation passes
    # [0, -1, 0, -1] (evaluation_engine.py:136): bring them to the kernels' absolute form (checked on the host when the boxes live there)
    if not torch.is_tensor(img_boxes):
        img_boxes = torch.as_tensor(img_boxes)
    if img_boxes.is_cuda or bool((img_boxes < 0).any()):
        key = (h, w, img_boxes.device, img_boxes.dtype)
        size = _BOX_SIZE_CACHE.get(key)
        if size is None:           # (built once per geometry: a fresh torch.tensor(..., device=cuda) is a pageable host-to-device copy on every call)
            size = _BOX_SIZE_CACHE[key] = torch.tensor([h, h, w, w], device=img_boxes.device, dtype=img_boxes.dtype)
        img_boxes = torch.where(img_boxes < 0, img_boxes + size, img_boxes)
    boxes = _boxes_to_device(img_boxes, dev)
    if boxes.shape != (b, 4):
        raise ValueError("cam2mask: img_boxes must be [b,4]")
    masks = [torch.empty((b, h, w), device=dev, dtype=torch.float32) for _ in range(G)]
    L = _C.lib()
    if refine_model is not None:
        _C.require_cuda(images)
        images = images.contiguous().float()
        dil, nd, iters = _C.int_array(refine_model.dilations), len(refine_model.dilations), refine_model.num_iter
    else:
        dil, nd, iters = _C.int_array([1]), 0, 0
    ws = _C.workspace(L.cosa_cam2mask_multi_workspace_bytes(G, b, C, h, downscale, nd if iters > 0 else 0), dev, "cam2mask")
    cam_ptrs = (ctypes.c_void_p * G)(*[c.data_ptr() for c in cams_list])
    mask_ptrs = (ctypes.c_void_p * G)(*[m.data_ptr() for m in masks])
    on_device = any(torch.is_tensor(t) for t in list(thresholds_high) + list(thresholds_low))
    if on_device:
        # thresholds that live on the device (adaptive thresholds): [G][hi, lo] float32, read by the kernels -- no host sync
        tdev = torch.stack([torch.stack([torch.as_tensor(h, device=dev).reshape(()).to(torch.float32),
                                         torch.as_tensor(l, device=dev).reshape(()).to(torch.float32)])
                            for h, l in zip(thresholds_high, thresholds_low)]).contiguous()
        hi = lo = (ctypes.c_float * G)(*([0.0] * G))
    else:
        tdev = None
        hi = (ctypes.c_float * G)(*[float(t) for t in thresholds_high])
        lo = (ctypes.c_float * G)(*[float(t) for t in thresholds_low])
    _C.check(L.cosa_cam2mask_multi(_C.ptr(images if iters > 0 else None), _C.ptr(boxes), cam_ptrs, _C.ptr(cls_labels), mask_ptrs,
                                   hi, lo, _C.ptr(tdev), G, b, C, h, downscale, int(bool(_fold_validation)), dil, nd, iters, float(ignore_index),
                                   _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_cam2mask_multi")
    return masks


# --------------------------------------------------------------------------------------------
# adaptive thresholds  (utils/seg_helper.py:924-959, main.py:94-103,138-151,174-184)
# --------------------------------------------------------------------------------------------
class DynamicQueue(object):
    """utils/seg_helper.py:946-959 with the storage on the device: a ring of `max_size` rows x `dim` float64 values, created
    with uniform noise exactly as the reference's is (np.random.random) and overwritten `batch_size` rows at a time."""

    def __init__(self, max_size, dim, batch_size, device="cuda"):
        self.max_size = max_size
        self.queue = torch.from_numpy(np.random.random((max_size, dim))).to(device)
        self.ptr = 0
        self.batch_size = batch_size

    def update(self, income):
        # income -> batchsize,dim (device tensor, any float dtype; stored as float64 like the reference's numpy queue)
        self.queue[self.ptr:self.ptr + self.batch_size, :] = income.reshape(self.batch_size, -1).to(self.queue.dtype)
        self.ptr = (self.ptr + self.batch_size) % self.max_size

    def getqueue(self):
        return self.queue


def cell_bilinear(x, g):
    """F.interpolate(x, size=(g, g), mode='bilinear', align_corners=False) for an integer, even reduction factor (main.py:140:
    448 -> 28): every output is 0.5*(0.5*a + 0.5*b) + 0.5*(0.5*c + 0.5*d) of the four pixels around the cell centre, evaluated
    in ATen's order, so the values are bit-identical -- but as four strided gathers instead of ATen's kernel, which gives a
    whole [b*C] column to each of only g*g threads (0.74 ms per call at b=16, C=20; this: ~30 us)."""
    S = x.shape[-1]
    s = S // g if g > 0 else 0
    if x.shape[-2] != S or g <= 0 or s * g != S or s % 2:
        return F.interpolate(x, size=(g, g), mode='bilinear', align_corners=False)
    a = s // 2 - 1
    r0, r1 = x[:, :, a::s, :], x[:, :, a + 1::s, :]
    top = 0.5 * r0[..., a::s] + 0.5 * r0[..., a + 1::s]
    bot = 0.5 * r1[..., a::s] + 0.5 * r1[..., a + 1::s]
    return 0.5 * top + 0.5 * bot


def rungmm_device(queue, modal, filter_thre=0.05, tol=1e-3, reg_covar=1e-6, max_iter=100):
    """The fit of `rungmm` without leaving the device: returns a float64 tensor [13] -- [0] low threshold (largest sample of
    component 0), [1] high threshold (smallest sample of component 2; NaN for modal=2), [2] EM iterations, [3] status bits
    (see include/cosa_hip.h), then means / weights / inverse sigmas.  No host synchronisation."""
    assert modal in [2, 3]
    if filter_thre < 0:
        raise ValueError("rungmm: filter_thre must be >= 0")
    _C.require_cuda(queue)
    q = queue.to(torch.float64).flatten()
    keep = q > filter_thre
    xs, _ = torch.sort(torch.where(keep, q, torch.full_like(q, float("inf"))))
    n = keep.sum()                                                   # int64, stays on the device
    out = torch.empty(13, device=q.device, dtype=torch.float64)
    L = _C.lib()
    ws = _C.workspace(L.cosa_gmm_workspace_bytes(), q.device, "gmm")
    _C.check(L.cosa_gmm_fit_thresholds(_C.ptr(xs), _C.ptr(n), xs.numel(), modal, tol, reg_covar, max_iter, _C.ptr(out), _C.ptr(ws),
                                       ws.numel(), _C.stream_ptr()), "cosa_gmm_fit_thresholds")
    return out


def rungmm(queue, modal, filter_thre=0.05):
    """utils/seg_helper.py:924-943: (low, high) for modal=3, low for modal=2, as Python floats (this form synchronises; the
    training step uses rungmm_device).  An empty outer component raises ValueError like the reference's max() / min()."""
    if isinstance(queue, np.ndarray):
        queue = torch.from_numpy(queue).cuda()
    out = rungmm_device(queue, modal, filter_thre).tolist()
    status = int(out[3])
    if status & 8:
        raise _C.CosaError("rungmm: device barrier expired")
    if status & 4:
        raise ValueError("rungmm: fewer samples above filter_thre than mixture components")
    if status & 1 or (modal == 3 and status & 2):
        raise ValueError("rungmm: max()/min() of an empty component")
    return out[0] if modal == 2 else (out[0], out[1])


# --------------------------------------------------------------------------------------------
# seg_loss  (utils/seg_helper.py:800-813)
# --------------------------------------------------------------------------------------------
def seg_loss(seg_pred, mask_label, fg_alpha=0.5, ignore_index=255):
    assert fg_alpha >= 0 and fg_alpha <= 1, "fg_alpha should be in [0,1]"
    lab = mask_label.long()
    bg_label = torch.where(lab != 0, torch.full_like(lab, ignore_index), lab)
    fg_label = torch.where(lab == 0, torch.full_like(lab, ignore_index), lab)
    # one log-softmax pass shared by the two class-balanced terms
    logp = F.log_softmax(seg_pred.float(), dim=1)
    bg_loss = F.nll_loss(logp, bg_label, ignore_index=ignore_index, reduction='sum') / ((bg_label != ignore_index).sum() + 1e-6)
    fg_loss = F.nll_loss(logp, fg_label, ignore_index=ignore_index, reduction='sum') / ((fg_label != ignore_index).sum() + 1e-6)
    return (1 - fg_alpha) * bg_loss + fg_alpha * fg_loss


# --------------------------------------------------------------------------------------------
# DenseEnergyLoss  (utils/seg_helper.py:191-230, 864-903)
# --------------------------------------------------------------------------------------------
class DenseEnergyLossFunction(Function):
    """utils/seg_helper.py:864-903 with the bilateral filter, gate, dot product and the backward
    scaling all on the device (no D2H/H2D hops, no host-side AS)."""

    @staticmethod
    def forward(ctx, images, segmentations, sigma_rgb, sigma_xy, ROIs, unlabel_region):
        _C.require_cuda(images, segmentations, ROIs, unlabel_region)
        N, K, H, W = segmentations.shape
        images = images.contiguous().float()
        seg = segmentations.contiguous().float()
        roi = ROIs.contiguous().float()
        unl = unlabel_region.contiguous().to(torch.uint8)
        AS = torch.empty_like(seg)
        loss = torch.empty(1, device=seg.device, dtype=torch.float32)
        L = _C.lib()
        ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), seg.device, "bilateral")
        _C.check(L.cosa_dense_energy_forward(_C.ptr(images), _C.ptr(seg), _C.ptr(roi), _C.ptr(unl), _C.ptr(AS), _C.ptr(loss),
                                             N, K, H, W, float(sigma_rgb), float(sigma_xy), _C.ptr(ws), ws.numel(),
                                             _C.stream_ptr()), "cosa_dense_energy_forward")
        ctx.save_for_backward(AS, roi)
        ctx.shape = (N, K, H, W)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        AS, roi = ctx.saved_tensors
        N, K, H, W = ctx.shape
        g = grad_output.contiguous().float()
        grad_seg = torch.empty_like(AS)
        _C.check(_C.lib().cosa_dense_energy_backward(_C.ptr(AS), _C.ptr(roi), _C.ptr(g), _C.ptr(grad_seg), N, K, H, W,
                                                     _C.stream_ptr()), "cosa_dense_energy_backward")
        return None, grad_seg, None, None, None, None


class DenseEnergyLoss(torch.nn.Module):
    """utils/seg_helper.py:191-208"""

    def __init__(self, weight, sigma_rgb, sigma_xy, scale_factor):
        super().__init__()
        self.weight = weight
        self.sigma_rgb = sigma_rgb
        self.sigma_xy = sigma_xy
        self.scale_factor = scale_factor

    def forward(self, images, segmentations, ROIs, seg_label):
        sf = self.scale_factor
        scaled_images = F.interpolate(images, scale_factor=sf, recompute_scale_factor=True)
        scaled_segs = F.interpolate(segmentations, scale_factor=sf, mode='bilinear', align_corners=False,
                                    recompute_scale_factor=True)
        scaled_ROIs = F.interpolate(ROIs.unsqueeze(1), scale_factor=sf, recompute_scale_factor=True).squeeze(1)
        scaled_seg_label = F.interpolate(seg_label.float(), scale_factor=sf, mode='nearest', recompute_scale_factor=True)
        unlabel_region = (scaled_seg_label.long() == 255).squeeze(1)
        return self.weight * DenseEnergyLossFunction.apply(scaled_images, scaled_segs, self.sigma_rgb,
                                                           self.sigma_xy * self.scale_factor, scaled_ROIs, unlabel_region)

    def extra_repr(self):
        return 'sigma_rgb={}, sigma_xy={}, weight={}, scale_factor={}'.format(
            self.sigma_rgb, self.sigma_xy, self.weight, self.scale_factor)


class FusedSegRegLoss(Function):
    """seg_loss(main) / seg_loss(aux) blend + dense-energy regulariser of the SAME low-res logits in two launches.

    Equivalent to main.py:167-212 (F.interpolate -> seg_loss x2 -> get_energy_loss) with fg_alpha = 0.5 and
    aux_cam2seg_alpha = 0.5 (the reference defaults), but nothing of size [b,K,S,S] is ever written: the kernels
    re-derive the per-pixel softmax from an LDS tile of the low-res logits.  Returns (seg_loss, reg_loss)."""

    @staticmethod
    def forward(ctx, seg_lr, maskA, maskB, simg, boxes, weight, sigma_rgb, sigma_xy, prepared=None):
        _C.require_cuda(seg_lr, maskA, maskB, simg, boxes)
        seg_lr = seg_lr.contiguous().float()
        B, K, hs, ws = seg_lr.shape
        S = maskA.shape[-1]
        Sq = S // 2
        dev = seg_lr.device
        sums = torch.empty(8, device=dev)
        s_seg = torch.empty((B, K, Sq, Sq), device=dev)
        s_img = torch.empty((B, 3, Sq, Sq), device=dev)
        roi = torch.empty((B, Sq, Sq), device=dev)
        unl = torch.empty((B, Sq, Sq), device=dev, dtype=torch.uint8)
        L = _C.lib()
        maskA, maskB, simg = maskA.contiguous().float(), maskB.contiguous().float(), simg.contiguous().float()
        wsl = _C.workspace(L.cosa_seg_loss_workspace_bytes(B, K, hs, ws), dev, "seg_loss")
        _C.check(L.cosa_seg_loss_forward(_C.ptr(seg_lr), _C.ptr(maskA), _C.ptr(maskB), _C.ptr(simg), _C.ptr(boxes), _C.ptr(sums),
                                         _C.ptr(s_seg), _C.ptr(s_img), _C.ptr(roi), _C.ptr(unl), B, K, hs, ws, S, _C.ptr(wsl), wsl.numel(),
                                         _C.stream_ptr()), "cosa_seg_loss_forward")
        AS = torch.empty_like(s_seg)
        energy = torch.empty(1, device=dev)
        if prepared is not None and prepared.matches(B, K, Sq, sigma_rgb, sigma_xy):
            # the lattice of this strong image was built on a side stream while the networks ran (PreparedLattice)
            prepared.join()
            _C.check(L.cosa_dense_energy_forward_prepared(_C.ptr(s_seg), _C.ptr(roi), _C.ptr(unl), _C.ptr(AS), _C.ptr(energy), B, K, Sq, Sq,
                                                          float(sigma_rgb), float(sigma_xy), _C.ptr(prepared.ws), prepared.ws.numel(),
                                                          _C.stream_ptr()), "cosa_dense_energy_forward_prepared")
        else:
            wsb = _C.workspace(L.cosa_bilateral_workspace_bytes(B, K, Sq, Sq), dev, "bilateral")
            _C.check(L.cosa_dense_energy_forward(_C.ptr(s_img), _C.ptr(s_seg), _C.ptr(roi), _C.ptr(unl), _C.ptr(AS), _C.ptr(energy), B, K,
                                                 Sq, Sq, float(sigma_rgb), float(sigma_xy), _C.ptr(wsb), wsb.numel(), _C.stream_ptr()),
                     "cosa_dense_energy_forward")
        # 0.5 * (0.5 bgA + 0.5 fgA) + 0.5 * (0.5 bgB + 0.5 fgB), each term sum / (count + 1e-6)  (seg_helper.py:800-813, main.py:200-203):
        # four vector ops instead of eighteen scalar ones
        pairs = sums.view(4, 2)
        seg_l = (pairs[:, 0] / (pairs[:, 1] + 1e-6)).sum() * 0.25
        ctx.save_for_backward(seg_lr, maskA, maskB, sums, AS, roi)
        ctx.weight = float(weight)
        ctx.S = S
        return seg_l, energy * float(weight)

    @staticmethod
    def backward(ctx, g_seg, g_reg):
        seg_lr, maskA, maskB, sums, AS, roi = ctx.saved_tensors
        B, K, hs, ws = seg_lr.shape
        grad = torch.empty_like(seg_lr)
        gs = g_seg.reshape(1).float().contiguous()
        gr = (g_reg.reshape(1).float() * ctx.weight).contiguous()
        wsl = _C.workspace(_C.lib().cosa_seg_loss_workspace_bytes(B, K, hs, ws), seg_lr.device, "seg_loss")
        _C.check(_C.lib().cosa_seg_loss_backward(_C.ptr(seg_lr), _C.ptr(maskA), _C.ptr(maskB), _C.ptr(sums), _C.ptr(AS), _C.ptr(roi),
                                                 _C.ptr(gs), _C.ptr(gr), _C.ptr(grad), B, K, hs, ws, ctx.S, _C.ptr(wsl), wsl.numel(),
                                                 _C.stream_ptr()), "cosa_seg_loss_backward")
        return grad, None, None, None, None, None, None, None, None


class PreparedLattice:
    """The image-only half of the dense-energy regulariser (half-resolution de-normalised strong image + permutohedral lattice:
    hash table, per-pixel offsets / weights, blur neighbours) built on a side stream at the start of a step, so that only splat /
    blur / slice wait for the student's logits.  One instance per trainer; `start(simg, K)` then pass it to
    fused_seg_and_energy_loss(..., prepared=...)."""

    def __init__(self, sigma_rgb, sigma_xy):
        self.sigma = (float(sigma_rgb), float(sigma_xy))
        self.stream = None
        self.event = None
        self.ws = self.s_img = None
        self.key = None

    def start(self, simg, K):
        _C.require_cuda(simg)
        B, _, S, _ = simg.shape
        dev = simg.device
        if self.stream is None:
            self.stream, self.event = torch.cuda.Stream(device=dev), torch.cuda.Event()
        L = _C.lib()
        need = L.cosa_bilateral_workspace_bytes(B, K, S // 2, S // 2)
        if self.ws is None or self.ws.numel() < need or self.ws.device != dev:
            self.ws = torch.empty(need, device=dev, dtype=torch.uint8)
        if self.s_img is None or self.s_img.shape != (B, 3, S // 2, S // 2):
            self.s_img = torch.empty((B, 3, S // 2, S // 2), device=dev, dtype=torch.float32)
        simg = simg.contiguous().float()
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            _C.check(L.cosa_dense_energy_prepare(_C.ptr(simg), _C.ptr(self.s_img), B, K, S, self.sigma[0], self.sigma[1], _C.ptr(self.ws),
                                                 self.ws.numel(), _C.stream_ptr()), "cosa_dense_energy_prepare")
            self.event.record(self.stream)
        simg.record_stream(self.stream)
        self.key = (B, K, S // 2)

    def matches(self, B, K, Sq, sigma_rgb, sigma_xy):
        return self.key == (B, K, Sq) and self.sigma == (float(sigma_rgb), float(sigma_xy))

    def join(self):
        torch.cuda.current_stream().wait_event(self.event)
        self.key = None                                  # one filter pass per prepared lattice in the training step


def fused_seg_and_energy_loss(seg_pred_lr, mask_main, mask_aux, img, img_box, loss_layer, prepared=None):
    """(seg_loss, reg_loss) of main.py:167-212 for the reference defaults (fg_alpha 0.5, aux blend 0.5, scale_factor 0.5).
    `prepared`: a PreparedLattice started on this step's `img` (optional; otherwise the lattice is built here)."""
    if loss_layer.scale_factor != 0.5:
        raise NotImplementedError("fused losses are built for DenseEnergyLoss(scale_factor=0.5) (main.py:77)")
    boxes = _boxes_to_device(img_box, seg_pred_lr.device)
    return FusedSegRegLoss.apply(seg_pred_lr, mask_main, mask_aux, img, boxes, loss_layer.weight, loss_layer.sigma_rgb,
                                 loss_layer.sigma_xy * loss_layer.scale_factor, prepared)


def _crop_mask_from_boxes(img_box, b, h, w, device):
    boxes = _boxes_to_device(img_box, device)
    ys = torch.arange(h, device=device, dtype=torch.int32)[None, :, None]
    xs = torch.arange(w, device=device, dtype=torch.int32)[None, None, :]
    inside = (ys >= boxes[:, 0, None, None]) & (ys < boxes[:, 1, None, None]) & \
             (xs >= boxes[:, 2, None, None]) & (xs < boxes[:, 3, None, None])
    return inside.float()


class SoftmaxHalfRes(Function):
    """F.softmax(logit, dim=1) followed by DenseEnergyLoss.forward's resize of the probabilities by 0.5 (utils/seg_helper.py:199-203, 224),
    forward and backward in one kernel each: nothing of the logits' size is written forward."""

    @staticmethod
    def forward(ctx, logit):
        logit = logit.contiguous().float()
        B, K, H, W = logit.shape
        out = torch.empty((B, K, H // 2, W // 2), device=logit.device, dtype=torch.float32)
        _C.check(_C.lib().cosa_softmax_halfres_forward(_C.ptr(logit), _C.ptr(out), B, K, H, W, _C.stream_ptr()), "cosa_softmax_halfres_forward")
        ctx.save_for_backward(logit)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (logit,) = ctx.saved_tensors
        B, K, H, W = logit.shape
        g = grad_out.contiguous().float()
        grad = torch.empty_like(logit)
        _C.check(_C.lib().cosa_softmax_halfres_backward(_C.ptr(logit), _C.ptr(g), _C.ptr(grad), B, K, H, W, _C.stream_ptr()),
                 "cosa_softmax_halfres_backward")
        return grad


def get_energy_loss(img, logit, label, img_box, loss_layer, mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]):
    """utils/seg_helper.py:210-230 (box mask built with one broadcast compare instead of a Python loop).  With the reference's
    scale_factor = 0.5 on the GPU the softmax and the layer's resize of the probabilities are one kernel (SoftmaxHalfRes)."""
    b, _, h, w = logit.shape
    crop_mask = _crop_mask_from_boxes(img_box, b, h, w, logit.device)
    mean_t = torch.tensor(mean, device=img.device, dtype=torch.float32)[None, :, None, None]
    std_t = torch.tensor(std, device=img.device, dtype=torch.float32)[None, :, None, None]
    _img = img * std_t + mean_t
    seg_label = label.type(torch.uint8).unsqueeze(1)
    if logit.is_cuda and loss_layer.scale_factor == 0.5 and h % 2 == 0 and w % 2 == 0:
        sf = loss_layer.scale_factor
        scaled_segs = SoftmaxHalfRes.apply(logit)
        scaled_images = F.interpolate(_img, scale_factor=sf, recompute_scale_factor=True)
        scaled_ROIs = F.interpolate(crop_mask.unsqueeze(1), scale_factor=sf, recompute_scale_factor=True).squeeze(1)
        scaled_seg_label = F.interpolate(seg_label.float(), scale_factor=sf, mode='nearest', recompute_scale_factor=True)
        unlabel_region = (scaled_seg_label.long() == 255).squeeze(1)
        return loss_layer.weight * DenseEnergyLossFunction.apply(scaled_images, scaled_segs, loss_layer.sigma_rgb,
                                                                 loss_layer.sigma_xy * sf, scaled_ROIs, unlabel_region)
    pred_prob = F.softmax(logit.float(), dim=1)
    return loss_layer(_img, pred_prob, crop_mask, seg_label)


# --------------------------------------------------------------------------------------------
# seg_refine_by_label / cam_loss  (utils/seg_helper.py:553-568, 593-602)
# --------------------------------------------------------------------------------------------
def seg_refine_by_label(seg, cls_label, softmaxtemp, after_softmax=False):
    b, c, h, w = seg.shape
    cls_label_bk = torch.cat([torch.ones(b, 1, device=cls_label.device, dtype=cls_label.dtype), cls_label], dim=1)
    if after_softmax:
        seg = F.softmax(seg / softmaxtemp, dim=1)
        return cls_label_bk[:, :, None, None] * seg
    valid_seg = torch.where((cls_label_bk == 0)[:, :, None, None], torch.full_like(seg, -1e5), seg)
    return F.softmax(valid_seg / softmaxtemp, dim=1)


def cam_loss_targets(seg_scales, cls_label, S, out_hw, softmaxtemp):
    """seg_refine_by_label(sum_scales seg, T)[:, 1:] bilinearly resized to `out_hw` (main.py:227-228, seg_helper.py:553-568,
    595-597), computed from the per-scale low-res teacher segs without building the [b,K,S,S] tensor."""
    b2, K = seg_scales[0].shape[:2]
    B = b2 // 2
    oh, ow = out_hw
    out = torch.empty((B, K - 1, oh, ow), device=seg_scales[0].device, dtype=torch.float32)
    n = len(seg_scales)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in seg_scales])
    hs = _C.int_array([t.shape[2] for t in seg_scales])
    ws = _C.int_array([t.shape[3] for t in seg_scales])
    lab = cls_label.contiguous().float()
    _C.check(_C.lib().cosa_cam_loss_targets(ptrs, hs, ws, n, _C.ptr(lab), _C.ptr(out), B, K, int(S), oh, ow, float(softmaxtemp),
                                            _C.stream_ptr()), "cosa_cam_loss_targets")
    return out


class _MultilabelSoftMarginFn(torch.autograd.Function):
    """F.multilabel_soft_margin_loss(v, y), v = x or relu(x), value and gradient from one kernel pass (cosa_msm_loss)"""

    @staticmethod
    def forward(ctx, x, y, relu):
        x = x.contiguous()
        y = y.contiguous().float()
        if x.dim() == 4:
            B, C, H, W = x.shape
            R, HW = B * H * W, H * W
        else:
            R, C = x.shape
            HW = 1
        grad = torch.empty_like(x)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        ws = torch.empty((R + 255) // 256, device=x.device, dtype=torch.float64)
        _C.check(_C.lib().cosa_msm_loss(_C.ptr(x), _C.ptr(y), _C.ptr(grad), _C.ptr(loss), _C.ptr(ws), R, C, HW, int(relu), _C.stream_ptr()),
                 "cosa_msm_loss")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def multilabel_soft_margin(x, y, relu=False):
    """F.multilabel_soft_margin_loss(relu(x) if relu else x, y) over the class dimension (dim 1 of [R,C] or [B,C,H,W] logits: every pixel a
    row, as cam_loss flattens them): the fused kernel for fp32 CUDA logits, torch otherwise"""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 4) and y.shape == x.shape:
        return _MultilabelSoftMarginFn.apply(x, y, relu)
    v = F.relu(x) if relu else x
    if x.dim() == 4:
        C = x.shape[1]
        v, y = v.float().permute(0, 2, 3, 1).reshape(-1, C), y.permute(0, 2, 3, 1).reshape(-1, C)
    return F.multilabel_soft_margin_loss(v.float(), y)


def cam_loss_from_targets(cam, targets, is_relu=True):
    """cam_loss (seg_helper.py:593-602) given pre-resized targets [B,C,H,W]"""
    if cam.is_cuda and cam.dtype == torch.float32 and targets.shape == cam.shape:
        return multilabel_soft_margin(cam, targets, relu=is_relu)
    B, C, H, W = cam.shape
    if is_relu:
        cam = F.relu(cam)
    cam_flat = cam.float().permute(0, 2, 3, 1).reshape(B * H * W, C)
    tgt_flat = targets.permute(0, 2, 3, 1).reshape(B * H * W, C)
    return F.multilabel_soft_margin_loss(cam_flat, tgt_flat)


def cam_loss(cam, seg_ps, is_relu=True):
    B, C, H, W = cam.shape
    seg_ps_fg = seg_ps[:, 1:, ...]
    seg_ps_fg = F.interpolate(seg_ps_fg, size=[H, W], mode='bilinear', align_corners=False)
    seg_ps_fg_flat = seg_ps_fg.permute(0, 2, 3, 1).reshape(B * H * W, C)
    if is_relu:
        cam = F.relu(cam)
    cam_flat = cam.float().permute(0, 2, 3, 1).reshape(B * H * W, C)
    return F.multilabel_soft_margin_loss(cam_flat, seg_ps_fg_flat)


# --------------------------------------------------------------------------------------------
# dense-CRF post-processing of the final evaluation  (utils/seg_helper.py:961-996)
# --------------------------------------------------------------------------------------------
class DenseCRF(object):
    """utils/seg_helper.py:961-987, same constructor and call: `DenseCRF(...)(image [H,W,3] uint8, probmap [C,H,W])` -> Q [C,H,W].

    The reference delegates to pydensecrf (DenseCRF2D: setUnaryEnergy(-log p), addPairwiseGaussian, addPairwiseBilateral, `iter_max`
    mean-field steps).  That library is not in this image; its published algorithm (Kraehenbuehl & Koltun 2011) is two
    permutohedral-lattice filters per step -- exactly what the bilateral regulariser of the training loop already runs on the device --
    around a softmax:   Q <- softmax(-U + pos_w K_g(Q) + bi_w K_b(Q)),   K(v) = n F(n v),  n = 1 / sqrt(F(1) + 1e-20)  (symmetric
    normalisation), F_g on the 2-D lattice of (x, y) / pos_xy_std, F_b on the 5-D lattice of (x, y) / bi_xy_std, rgb / bi_rgb_std.
    Checked in the test-suite against a CPU restatement of the same algorithm (parity with pydensecrf itself is unpinned: no fixture
    of its output exists).
    numpy in -> numpy out (the reference's types); CUDA tensors in -> CUDA tensor out."""

    def __init__(self, iter_max, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std):
        self.iter_max = iter_max
        self.pos_w = pos_w
        self.pos_xy_std = pos_xy_std
        self.bi_w = bi_w
        self.bi_xy_std = bi_xy_std
        self.bi_rgb_std = bi_rgb_std

    @staticmethod
    def _filter_gauss(v, sxy):
        K, H, W = v.shape
        L = _C.lib()
        ws = _C.workspace(L.cosa_lattice_filter_d2_workspace_bytes(1, K, H, W), v.device, "crf_d2")
        out = torch.empty_like(v)
        _C.check(L.cosa_lattice_filter_d2(_C.ptr(v), _C.ptr(out), 1, K, H, W, float(sxy), _C.ptr(ws), ws.numel(), _C.stream_ptr()),
                 "cosa_lattice_filter_d2")
        return out

    @staticmethod
    def _filter_bilateral(img, v, srgb, sxy):
        K, H, W = v.shape
        L = _C.lib()
        ws = _C.workspace(L.cosa_bilateral_workspace_bytes(1, K, H, W), v.device, "bilateral")
        out = torch.empty_like(v)
        _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(img), _C.ptr(v), _C.ptr(out), 1, K, H, W, float(srgb), float(sxy), None, _C.ptr(ws),
                                                  ws.numel(), _C.stream_ptr()), "cosa_bilateralfilter_batch_dev")
        return out

    def __call__(self, image, probmap):
        as_numpy = not torch.is_tensor(probmap)
        dev = probmap.device if torch.is_tensor(probmap) and probmap.is_cuda else torch.device("cuda", torch.cuda.current_device())
        prob = torch.as_tensor(np.ascontiguousarray(probmap) if as_numpy else probmap).to(dev).float().contiguous()
        img = torch.as_tensor(np.ascontiguousarray(image) if not torch.is_tensor(image) else image).to(dev)
        C, H, W = prob.shape
        if img.shape != (H, W, 3):
            raise ValueError("DenseCRF: image must be [H, W, 3] (the reference passes the de-normalised uint8 image, HWC)")
        img = img.permute(2, 0, 1).float().contiguous()                                   # CHW planes 0..255, as the lattice kernels read them
        U = -torch.log(prob.clamp(1e-5, 1.0))                                             # pydensecrf.utils.unary_from_softmax
        one = torch.ones((1, H, W), device=dev)
        n_g = torch.rsqrt(self._filter_gauss(one, self.pos_xy_std) + 1e-20)
        n_b = torch.rsqrt(self._filter_bilateral(img, one, self.bi_rgb_std, self.bi_xy_std) + 1e-20)
        Q = torch.softmax(-U, dim=0)
        for _ in range(int(self.iter_max)):
            t = -U + float(self.pos_w) * (n_g * self._filter_gauss((Q * n_g).contiguous(), self.pos_xy_std)) \
                + float(self.bi_w) * (n_b * self._filter_bilateral(img, (Q * n_b).contiguous(), self.bi_rgb_std, self.bi_xy_std))
            Q = torch.softmax(t, dim=0)
        return Q.cpu().numpy() if as_numpy else Q


crf_inference_infv2 = DenseCRF(       # utils/seg_helper.py:989-996
    iter_max=1,
    pos_xy_std=1,
    pos_w=1,
    bi_xy_std=121,
    bi_rgb_std=5,
    bi_w=4,
)
