"""Generate tests/golden/*.npz from the REFERENCE itself (authoring container only).

    python -m oracle.gen_golden

Every vector is produced by the reference's own code: its Python modules loaded by path
(oracle/ref_loader.py) and its C++ bilateral filter compiled from its sources into oracle/_ref/.
Only inputs and expected outputs are stored -- no reference source text.  TEST INFRASTRUCTURE.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle, ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DIL = [1, 2, 4, 8, 12, 24]


def smooth_field(rng, n, h, w, lo=0.0, hi=1.0, waves=4):
    """sum of a few low-frequency sinusoids, rescaled to [lo,hi] per plane"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((n, h, w))
    for i in range(n):
        acc = np.zeros((h, w))
        for _ in range(waves):
            fx, fy = rng.uniform(0.02, 0.25, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            acc += rng.uniform(0.3, 1.0) * np.sin(xx * fx + ph[0]) * np.cos(yy * fy + ph[1])
        acc = (acc - acc.min()) / (acc.max() - acc.min() + 1e-12)
        out[i] = lo + (hi - lo) * acc
    return out.astype(np.float32)


def synth_image255(rng, n, h, w, noise=2.0):
    base = smooth_field(rng, n * 3, h, w, 0, 255).reshape(n, 3, h, w)
    img = np.clip(np.floor(base + rng.normal(0, noise, base.shape)), 0, 255)
    return img.astype(np.float32)


def gen_par(rng):
    par_mod = ref_loader.par_module()
    out = {}
    for tag, (K, h, w) in {"k2": (2, 48, 40), "k4": (4, 40, 56)}.items():
        img = torch.from_numpy(synth_image255(rng, 1, h, w) / 255.0)
        masks = torch.from_numpy(smooth_field(rng, K, h, w))[None].softmax(1)
        par = par_mod.PAR(num_iter=10, dilations=DIL)
        with torch.no_grad():
            ref = par(img, masks)
        out[f"{tag}_img"] = img.numpy()
        out[f"{tag}_masks"] = masks.numpy()
        out[f"{tag}_out"] = ref.numpy()
    # short propagation / other dilation set
    img = torch.from_numpy(synth_image255(rng, 2, 24, 24) / 255.0)
    masks = torch.from_numpy(smooth_field(rng, 6, 24, 24)).reshape(2, 3, 24, 24).softmax(1)
    par = par_mod.PAR(num_iter=3, dilations=[1, 2, 4])
    with torch.no_grad():
        ref = torch.cat([par(img[i:i + 1], masks[i:i + 1]) for i in range(2)])
    out.update(b2_img=img.numpy(), b2_masks=masks.numpy(), b2_out=ref.numpy(), b2_dil=np.array([1, 2, 4]), b2_iter=3)
    np.savez_compressed(os.path.join(OUT, "par.npz"), **out)


def gen_cam2mask(rng):
    sh = ref_loader.seg_helper()
    par_mod = ref_loader.par_module()
    B, C, S = 4, 6, 64
    cams = smooth_field(rng, B * C, S, S).reshape(B, C, S, S)
    cams = np.maximum(cams * 1.3 - 0.15, 0).astype(np.float32)
    labels = np.zeros((B, C), np.float32)
    labels[0, [1, 4]] = 1
    labels[1, [0]] = 1
    labels[2, [2, 3, 5]] = 1
    labels[3, [0, 1, 2, 3, 4, 5]] = 1
    boxes = np.array([[0, S, 0, S], [3, 60, 5, 50], [0, S, 10, S], [7, 33, 0, 64]], np.int16)
    images = synth_image255(rng, B, S, S) / 255.0
    t = torch.from_numpy
    vc = sh.cam_validation(t(cams), t(labels))
    out = dict(cams=cams, labels=labels, boxes=boxes, images=images.astype(np.float32), valid_cams=vc.numpy(),
               thr=np.array([0.7, 0.25], np.float32))
    with torch.no_grad():
        out["mask_none"] = sh.cam2mask(t(images), t(boxes), vc, t(labels), 0.7, 0.25, downscale=2).numpy()
        out["mask_none_ds0"] = sh.cam2mask(t(images), t(boxes), vc, t(labels), 0.7, 0.25, downscale=0).numpy()
        par = par_mod.PAR(num_iter=10, dilations=DIL)
        out["mask_par"] = sh.cam2mask(t(images), t(boxes), vc, t(labels), 0.7, 0.25, refine_model=par, downscale=2).numpy()
        out["mask_par_coco_thr"] = sh.cam2mask(t(images), t(boxes), vc, t(labels), 0.65, 0.25, refine_model=par,
                                               downscale=2).numpy()
    np.savez_compressed(os.path.join(OUT, "cam2mask.npz"), **out)


class _StubModel:
    """deterministic stand-in network for multi_scale_camseg: returns smooth functions of the input so that the
    flip / scale / 'last-scale-only' bookkeeping is what the golden pins."""

    def __init__(self, C):
        self.C = C

    def __call__(self, x, cam_only=False):
        B, _, H, W = x.shape
        f = F.avg_pool2d(x, 8)                                   # [B,3,H/8,W/8]
        w_cam = torch.linspace(-1.0, 1.5, self.C * 3).reshape(self.C, 3, 1, 1)
        w_aux = torch.linspace(1.2, -0.8, self.C * 3).reshape(self.C, 3, 1, 1)
        w_seg = torch.linspace(-0.5, 0.9, (self.C + 1) * 3).reshape(self.C + 1, 3, 1, 1)
        cam = F.conv2d(f, w_cam) + 0.1 * f[:, :1].roll(1, -1)
        cam_aux = F.conv2d(f, w_aux) - 0.05 * f[:, 1:2].roll(1, -2)
        seg = F.conv2d(f, w_seg)
        # (classification logits only matter to multi_scale_camsegv3(getcls=True): any deterministic function will do)
        return cam.mean((2, 3)), cam_aux.amax((2, 3)), None, seg, cam, cam_aux


def stub_outputs(x, C):
    return _StubModel(C)(x)


def gen_camseg(rng):
    sh = ref_loader.seg_helper()
    b, S, C = 2, 64, 5
    imgs = torch.from_numpy(smooth_field(rng, b * 3, S, S, -2, 2).reshape(b, 3, S, S))
    cam, cam_aux, seg = sh.multi_scale_camseg(_StubModel(C), imgs, [1.0, 0.5, 1.5])
    np.savez_compressed(os.path.join(OUT, "camseg_tail.npz"), imgs=imgs.numpy(), cam=cam.numpy(), cam_aux=cam_aux.numpy(),
                        seg=seg.numpy(), C=C)


def gen_bilateral(rng):
    sh = ref_loader.seg_helper()
    out = {}
    for tag, (N, K, H, W, kind) in {"smooth": (2, 3, 32, 32, "s"), "noise": (2, 3, 32, 32, "n"), "odd": (1, 2, 30, 37, "s")}.items():
        img = synth_image255(rng, N, H, W) if kind == "s" else rng.uniform(0, 255, (N, 3, H, W)).astype(np.float32)
        seg = torch.from_numpy(smooth_field(rng, N * K, H, W).reshape(N, K, H, W)).softmax(1).numpy()
        ref = np.zeros_like(seg).reshape(-1)
        c_oracle.ref_bilateralfilter_batch(img, seg, ref, N, K, H, W, 15.0, 50.0)
        out[f"{tag}_img"] = img
        out[f"{tag}_seg"] = seg
        out[f"{tag}_out"] = ref.reshape(N, K, H, W)
    # DenseEnergyLoss through the reference's own autograd Function (forward value + gradient)
    N, K, S = 2, 4, 64
    img = synth_image255(rng, N, S, S)
    logits = torch.from_numpy(smooth_field(rng, N * K, S, S, -2, 2).reshape(N, K, S, S)).requires_grad_(True)
    probs = logits.softmax(1)
    roi = torch.zeros(N, S, S)
    roi[0] = 1
    roi[1, 4:60, 8:50] = 1
    label = torch.from_numpy(rng.choice([0, 1, 2, 255], size=(N, 1, S, S), p=[0.4, 0.2, 0.2, 0.2]).astype(np.uint8))
    layer = sh.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    loss = layer(torch.from_numpy(img), probs, roi.clone(), label)
    loss.backward()
    out.update(del_img=img, del_logits=logits.detach().numpy(), del_roi=roi.numpy(), del_label=label.numpy(),
               del_loss=loss.detach().numpy(), del_grad=logits.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "bilateral.npz"), **out)


def gen_misc(rng):
    sh = ref_loader.seg_helper()
    th = ref_loader.torch_helper_fns()
    out = {}
    x = torch.from_numpy(rng.normal(0, 1.2, (2, 3, 16, 16)).astype(np.float32))
    out["denorm_in"] = x.numpy()
    out["denorm_out"] = th.denormalize_img(x).numpy()
    # LR schedule of PolyWarmupAdamW
    p = torch.nn.Parameter(torch.zeros(1))
    opt = th.PolyWarmupAdamW([{"params": [p], "lr": 6e-5}], lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=1500,
                             max_iter=32000, warmup_ratio=1e-6, power=0.9, min_mult=0.0)
    steps = [0, 1, 2, 750, 1499, 1500, 1501, 16000, 31999]
    lrs = []
    for st in steps:
        opt.global_step = st
        p.grad = torch.zeros(1)
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
    out["lr_steps"] = np.array(steps)
    out["lr_values"] = np.array(lrs, np.float64)
    # losses
    B, C, S = 2, 5, 32
    seg_pred = torch.from_numpy(rng.normal(0, 1, (B, C + 1, S, S)).astype(np.float32))
    mask = torch.from_numpy(rng.choice([0, 1, 3, 5, 255], size=(B, S, S)).astype(np.float32))
    out["segloss_pred"] = seg_pred.numpy()
    out["segloss_mask"] = mask.numpy()
    out["segloss_out"] = sh.seg_loss(seg_pred, mask, fg_alpha=0.5).numpy()
    labels = torch.tensor([[1, 0, 0, 1, 0], [0, 1, 1, 0, 0]], dtype=torch.float32)
    seg = torch.from_numpy(rng.normal(0, 1, (B, C + 1, S, S)).astype(np.float32))
    ref = sh.seg_refine_by_label(seg, labels, softmaxtemp=0.01)
    out["refine_seg"] = seg.numpy()
    out["refine_labels"] = labels.numpy()
    out["refine_out"] = ref.numpy()
    cam = torch.from_numpy(rng.normal(0, 1, (B, C, 4, 4)).astype(np.float32))
    out["camloss_cam"] = cam.numpy()
    out["camloss_out"] = sh.cam_loss(cam, ref).numpy()
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)


def gen_vit(rng):
    """tiny ViT (embed 128, 2 heads x 64, depth 3) from the reference's own VisionTransformer + LargeFOV classes,
    composed the way models/__init__.py:163-206 composes them."""
    vit = ref_loader.vit_module()
    head = ref_loader.conv_head_module()
    from functools import partial
    torch.manual_seed(3)
    C1, E = 7, 128
    enc = vit.VisionTransformer(patch_size=16, embed_dim=E, depth=3, num_heads=2, mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), aux_layer=-2, num_classes=10)
    dec = head.LargeFOV(in_planes=E, out_planes=C1)
    cls_w = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    aux_w = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    with torch.no_grad():   # non-trivial biases / norms so every term is exercised
        for n_, p_ in enc.named_parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
        # LargeFOV is hard-wired to 512 channels (conv_head.py:14): keep the fixture small by giving the decoder
        # sparse int8-valued weights (stored as int8, value = q / 256)
        qstore = {}
        for n_, p_ in dec.named_parameters():
            q = rng.integers(-24, 25, size=tuple(p_.shape)).astype(np.int8)
            if p_.numel() > 1_000_000:
                q[rng.uniform(size=q.shape) > 0.06] = 0
            qstore["decoder." + n_] = q
            p_.copy_(torch.from_numpy(q.astype(np.float32) / 256.0))
    enc.eval(); dec.eval()
    x = torch.from_numpy(rng.normal(0, 1, (2, 3, 96, 64)).astype(np.float32))
    with torch.no_grad():
        cls_tok, tok, tok_aux = enc.forward_features(x)
        h, w = x.shape[-2] // 16, x.shape[-1] // 16
        to2d = lambda t: t.transpose(1, 2).reshape(t.shape[0], E, h, w)
        x4, xa = to2d(tok), to2d(tok_aux)
        seg = dec(x4)
        cam = F.conv2d(x4, cls_w.weight)
        cam_aux = F.conv2d(xa, aux_w.weight)
        cls = cls_w(F.adaptive_max_pool2d(x4, (1, 1))).view(-1, C1 - 1)
        cls_aux = aux_w(F.adaptive_max_pool2d(xa, (1, 1))).view(-1, C1 - 1)
    out = {"x": x.numpy(), "cls": cls.numpy(), "cls_aux": cls_aux.numpy(), "x4": x4.numpy(), "seg": seg.numpy(),
           "cam": cam.numpy(), "cam_aux": cam_aux.numpy()}
    for k, v in enc.state_dict().items():
        out["sd/encoder." + k] = v.numpy()
    for k, q in qstore.items():
        out["sdq/" + k] = q
    out["sd/classifier.weight"] = cls_w.weight.detach().numpy()
    out["sd/aux_classifier.weight"] = aux_w.weight.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "vit_tiny.npz"), **out)


def recipe_state(shapes, seed0=5000):
    """weights by a recipe a test can replay without the reference: per SORTED key, `torch.randn(shape, generator=manual_seed(seed0 + index)) *
    s(key)` (+ 1 for LayerNorm weights), s = 0.1 for one-dimensional parameters (biases, norms: so that every term is exercised), 0.05 for the two
    CAM classifiers, 0.02 otherwise (the reference's trunc_normal std).  shapes: {key: shape}.  Returns ({key: tensor}, sha256 hex of the bytes)."""
    import hashlib
    sd, h = {}, hashlib.sha256()
    for i, k in enumerate(sorted(shapes)):
        g = torch.Generator().manual_seed(seed0 + i)
        shp = tuple(shapes[k])
        s = 0.1 if len(shp) == 1 else (0.05 if k.endswith("classifier.weight") else 0.02)
        t = torch.randn(shp, generator=g) * s
        if len(shp) == 1 and ".norm" in "." + k and k.endswith(".weight"):
            t = t + 1.0
        sd[k] = t
        h.update(k.encode())
        h.update(t.numpy().tobytes())
    return sd, h.hexdigest()


VIT_BASE_CFG = dict(embed_dim=768, depth=2, num_heads=12, aux_layer=-2, num_classes=21, img=(96, 64), batch=2)


def gen_vit_base(rng):
    """a reference-produced vector INSIDE the HIP kernels' envelope -- the reference's own VisionTransformer at ViT-B WIDTH
    (embed 768, 12 heads x 64, depth 2: vit.py:219-330) + LargeFOV (conv_head.py:11-41) + the two CAM classifiers, composed the way
    models/__init__.py:163-206 composes them, on a 2 x 3 x 96 x 64 batch (6 x 4 patches: the bicubic pos-embed resize 14^2 -> 6 x 4 is live).
    Weights come from `recipe_state` (replayed by the test); stored: input, the six outputs of VITNetwork.forward, sha256 of the weights."""
    vit = ref_loader.vit_module()
    head = ref_loader.conv_head_module()
    from functools import partial
    cfg = VIT_BASE_CFG
    E, C1 = cfg["embed_dim"], cfg["num_classes"]
    enc = vit.VisionTransformer(patch_size=16, embed_dim=E, depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), aux_layer=cfg["aux_layer"], num_classes=1000)
    dec = head.LargeFOV(in_planes=E, out_planes=C1)
    cls_w = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    aux_w = torch.nn.Conv2d(E, C1 - 1, 1, bias=False)
    shapes = {"encoder." + k: v.shape for k, v in enc.state_dict().items()}
    shapes.update({"decoder." + k: v.shape for k, v in dec.state_dict().items()})
    shapes["classifier.weight"], shapes["aux_classifier.weight"] = cls_w.weight.shape, aux_w.weight.shape
    sd, digest = recipe_state(shapes)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=True)
    with torch.no_grad():
        cls_w.weight.copy_(sd["classifier.weight"])
        aux_w.weight.copy_(sd["aux_classifier.weight"])
    enc.eval(); dec.eval()
    H, W = cfg["img"]
    x = torch.from_numpy(rng.normal(0, 1, (cfg["batch"], 3, H, W)).astype(np.float32))
    with torch.no_grad():
        cls_tok, tok, tok_aux = enc.forward_features(x)
        h, w = H // 16, W // 16
        to2d = lambda t: t.transpose(1, 2).reshape(t.shape[0], E, h, w)
        x4, xa = to2d(tok), to2d(tok_aux)
        seg = dec(x4)
        cam = F.conv2d(x4, cls_w.weight)
        cam_aux = F.conv2d(xa, aux_w.weight)
        cls = cls_w(F.adaptive_max_pool2d(x4, (1, 1))).view(-1, C1 - 1)
        cls_aux = aux_w(F.adaptive_max_pool2d(xa, (1, 1))).view(-1, C1 - 1)
    out = {"x": x.numpy(), "cls": cls.numpy(), "cls_aux": cls_aux.numpy(), "x4": x4.numpy(), "seg": seg.numpy(),
           "cam": cam.numpy(), "cam_aux": cam_aux.numpy(), "weights_sha256": np.array(digest),
           "shape_keys": np.array(sorted(shapes)), "shape_dims": np.array([",".join(str(d) for d in shapes[k]) for k in sorted(shapes)])}
    np.savez_compressed(os.path.join(OUT, "vit_base_d2.npz"), **out)
    print("vit_base_d2: weights sha256", digest[:16], "cam range", float(cam.abs().max()), "seg range", float(seg.abs().max()))


EVAL_SIZES = [(333, 500), (375, 500), (500, 281)]


def eval_inputs(rng, C=4, S=448):
    """inputs of the evaluation-path goldens, rebuilt identically by the tests from the same seed: smooth low-res fields
    taken to (S,S) by an exact 16x bilinear resize (scale 1/16 is a power of two: no rounding in the source index)."""
    cam_lr = torch.from_numpy(smooth_field(rng, C, S // 16, S // 16, 0.0, 1.0).reshape(1, C, S // 16, S // 16))
    seg_lr = torch.from_numpy(smooth_field(rng, C + 1, S // 16, S // 16, -3.0, 3.0).reshape(1, C + 1, S // 16, S // 16))
    cam = F.interpolate(cam_lr, size=(S, S), mode="bilinear", align_corners=False)
    seg = F.interpolate(seg_lr, size=(S, S), mode="bilinear", align_corners=False)
    cls_label = torch.tensor([[1.0, 0.0, 1.0, 1.0]])[:, :C]
    return cam, seg, cls_label


def gen_eval(rng):
    """evaluation path (SURVEY f-1): evaluation_engine.py:74-126,198-207 + utils/evaluation.py:17-70 + torch_helper.compute_mAP"""
    sh = ref_loader.seg_helper()
    ev = ref_loader.evaluation_module()
    th = ref_loader.torch_helper_fns()
    out = {}
    # (1) multi_scale_camsegv3 with the stub network: 5 scales x 2 flips, cam_aux from the LAST scale only, summed cls logits
    b, S, C = 2, 64, 5
    imgs = torch.from_numpy(smooth_field(rng, b * 3, S, S, -2, 2).reshape(b, 3, S, S))
    cam, cam_aux, seg, cls_f, cls_a = sh.multi_scale_camsegv3(_StubModel(C), imgs, [1.0, 0.5, 1.5, 0.75, 1.25], getcls=True)
    out.update(v3_imgs=imgs.numpy(), v3_cam=cam.numpy(), v3_cam_aux=cam_aux.numpy(), v3_seg=seg.numpy(), v3_cls_f=cls_f.numpy(),
               v3_cls_a=cls_a.numpy(), v3_C=C)
    # (2) CAM / seg -> label maps at the ground truth's resolution (evaluation_engine.py:96-126,198-200)
    cam, seg, cls_label = eval_inputs(np.random.default_rng(171))
    for i, (H, W) in enumerate(EVAL_SIZES):
        rc = F.interpolate(cam, size=(H, W), mode="bilinear", align_corners=False)
        lab = sh.cam_to_label(rc.clone(), cls_label, bkg_thre=0.5, high_thre=0.7, low_thre=0.25, ignore_index=255)
        rs = F.interpolate(seg, size=(H, W), mode="bilinear", align_corners=False)
        vd = sh.seg_validation(rs, cls_label)
        out[f"lab_cam_{i}"] = lab.numpy().astype(np.uint8)
        out[f"lab_ps_{i}"] = torch.argmax(rs, dim=1).numpy().astype(np.uint8)
        out[f"lab_vd_{i}"] = torch.argmax(vd, dim=1).numpy().astype(np.uint8)
    # (3) cam_to_label with boxes + ignore_mid (returns (valid_cam, pseudo_label))
    c2 = torch.from_numpy(smooth_field(rng, 2 * 4, 40, 56, 0.0, 1.0).reshape(2, 4, 40, 56))
    l2 = torch.tensor([[1.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 1.0]])
    boxes = torch.tensor([[0, 40, 0, 56], [3, 33, 5, 50]])
    valid_cam, pl = sh.cam_to_label(c2.clone(), l2, img_box=boxes, bkg_thre=0.5, high_thre=0.7, low_thre=0.25, ignore_mid=True,
                                    ignore_index=255)
    out.update(box_cam=c2.numpy(), box_cls=l2.numpy(), box_boxes=boxes.numpy(), box_valid_cam=valid_cam.numpy(),
               box_label=pl.numpy().astype(np.int64))
    # (4) confusion-matrix scores
    nc = 6
    gts = [rng.integers(0, nc, size=(30 + 7 * i, 41)).astype(np.uint8) for i in range(3)]
    for g in gts:
        g[rng.random(g.shape) < 0.1] = 255
    gts[2][gts[2] == 4] = 1                      # a class absent from one image / rare overall
    preds = [np.where(rng.random(g.shape) < 0.7, np.minimum(g, nc - 1), rng.integers(0, nc, size=g.shape)).astype(np.uint8) for g in gts]
    ppreds = [p.copy() for p in preds]
    for p in ppreds:
        p[rng.random(p.shape) < 0.15] = 255
    sc = ev.scores(gts, preds, nc)
    ps = ev.pseudo_scores([g.copy() for g in gts], [p.copy() for p in ppreds], nc)
    for k, d in (("sc", sc), ("ps", ps)):
        out[f"{k}_pAcc"], out[f"{k}_mAcc"], out[f"{k}_miou"] = d["pAcc"], d["mAcc"], d["miou"]
        out[f"{k}_iou"] = np.array([d["iou"][i] for i in range(nc)])
    for i in range(3):
        out[f"sc_gt_{i}"], out[f"sc_pred_{i}"], out[f"sc_ppred_{i}"] = gts[i], preds[i], ppreds[i]
    out["sc_nc"] = nc
    # (5) per-sample average precision (torch_helper.compute_mAP -> sklearn average_precision_score), incl. tied scores
    y = (rng.random((6, 8)) < 0.4).astype(np.float32)
    y[3] = 0                                       # a sample without positives is skipped
    p = rng.random((6, 8)).astype(np.float32)
    p[1, 2] = p[1, 5]
    p[4, :4] = 0.5
    out.update(ap_labels=y, ap_scores=p, ap=np.array(th.compute_mAP(torch.from_numpy(y), torch.from_numpy(p))))
    np.savez_compressed(os.path.join(OUT, "eval.npz"), **out)


def gmm_queue(rng, rows, dim, random_rows):
    """a DynamicQueue as the training loop leaves it: rows of float32 CAM maxima (background near 0, object cells spread up to
    1, stored as float64) plus `random_rows` rows still holding the uniform float64 noise the queue is created with."""
    bgmask = rng.random((rows, dim)) < 0.55
    vals = np.where(bgmask, np.abs(rng.normal(0.0, 0.04, (rows, dim))),
                    np.clip(rng.normal(0.62, 0.2, (rows, dim)), 0, 1) * (rng.random((rows, dim)) < 0.8)
                    + np.clip(rng.normal(0.25, 0.06, (rows, dim)), 0, 1) * 0.0)
    mid = rng.random((rows, dim)) < 0.2
    vals = np.where(mid & ~bgmask, np.clip(rng.normal(0.28, 0.07, (rows, dim)), 0, 1), vals)
    q = vals.astype(np.float32).astype(np.float64)
    if random_rows:
        q[:random_rows] = rng.random((random_rows, dim))
    return q


def gen_gmm(rng):
    """adaptive thresholds: the reference's rungmm (utils/seg_helper.py:924-943) on synthetic queues; the iteration count of
    the same scikit-learn estimator is stored next to it as an extra pin for the restatement."""
    import sklearn
    import sklearn.mixture as skm
    sh = ref_loader.seg_helper()
    out = {"sklearn_version": np.array(sklearn.__version__)}
    cases = [("a", 48, 196, 0, 3, 0.05), ("b", 40, 196, 12, 3, 0.05), ("c", 32, 196, 0, 2, 0.05), ("d", 64, 49, 5, 3, 0.1),
             ("e", 24, 49, 24, 3, 0.05), ("f", 200, 196, 20, 3, 0.05)]      # e: the queue as created (all noise)
    for name, rows, dim, rnd, modal, thr in cases:
        q = gmm_queue(rng, rows, dim, rnd)
        res = sh.rungmm(q.copy(), modal=modal, filter_thre=thr)
        x = q.flatten()
        x = x[x > thr].reshape(-1, 1)
        init = [[x.min()], [np.median(x)], [x.max()]] if modal == 3 else [[x.min()], [x.max()]]
        gm = skm.GaussianMixture(modal, weights_init=[1 / modal] * modal, means_init=init, precisions_init=[[[1.0]]] * modal).fit(x)
        out[f"{name}_queue32"] = q[rnd:].astype(np.float32)          # exact: these rows hold float32 values
        out[f"{name}_queue_rand"] = q[:rnd]
        out[f"{name}_modal"], out[f"{name}_filter"] = np.array(modal), np.array(thr)
        out[f"{name}_thresholds"] = np.atleast_1d(np.array(res, np.float64))
        out[f"{name}_n_iter"] = np.array(gm.n_iter_)
        out[f"{name}_means"] = gm.means_.ravel()
        print("gmm", name, x.size, "samples", res, "iters", gm.n_iter_)
    np.savez_compressed(os.path.join(OUT, "gmm.npz"), **out)


def gen_augment(rng):
    """training input pipeline: the reference's own transforms.py / randaug.py functions, called in the order of
    dataloaders/voc.py:262-275, under fixed seeds of Python's `random` and numpy's global generator.  Stored: the source image,
    the seeds, and the three uint8 stages (crop, weak = after blur, strong) + img_box.  Seeds are scanned until all nine strong
    ops, blur on/off, flips and up/down-scaling are covered."""
    import random
    from PIL import Image
    tr, ra = ref_loader.dataloader_modules()
    crop = 64
    blur = tr.GaussianBlur(p=0.5)
    strong = ra.OneOf(transforms=[ra.Identity(), ra.AutoContrast(), ra.RandEqualize(), ra.RandSolarize(), ra.RandColor(),
                                  ra.RandContrast(), ra.RandBrightness(), ra.RandSharpness(), ra.RandPosterize()])
    names = ["Identity", "AutoContrast", "RandEqualize", "RandSolarize", "RandColor", "RandContrast", "RandBrightness",
             "RandSharpness", "RandPosterize"]
    out = {"crop_size": np.array(crop)}
    seen, n = set(), 0
    for seed in range(400):
        h, w = int(rng.integers(40, 110)), int(rng.integers(40, 110))
        base = synth_image255(rng, 1, h, w, noise=3.0)[0].transpose(1, 2, 0)
        image = np.clip(base, 0, 255).astype(np.uint8)
        random.seed(seed)
        np.random.seed(seed)
        img = tr.random_scaling(image, scale_range=[0.5, 2.0])
        img = tr.random_fliplr(img)
        cimg, box = tr.random_crop(img, crop_size=crop, mean_rgb=[0, 0, 0], ignore_index=255)
        pil = blur(Image.fromarray(cimg))
        # which op was chosen: replay the choice on a saved copy of the generator state
        state = np.random.get_state()
        chosen = np.random.choice(strong.transforms)
        np.random.set_state(state)
        spil = strong(pil)
        op = names.index(type(chosen).__name__)
        key = (op, np.asarray(pil).tobytes() != cimg.tobytes())
        if key in seen:
            continue
        seen.add(key)
        out[f"{n}_image"], out[f"{n}_seed"] = image, np.array(seed)
        out[f"{n}_crop"], out[f"{n}_weak"], out[f"{n}_strong"], out[f"{n}_box"] = cimg, np.asarray(pil), np.asarray(spil), box
        out[f"{n}_solarize_restated"] = np.array(op == 3)          # this case went through the restatement of mmcv.solarize
        out[f"{n}_op"] = np.array(op)
        n += 1
        if len(seen) == 18:
            break
    out["n"] = np.array(n)
    print("augment:", n, "cases", sorted(seen))
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)


def gen_signatures():
    """The reference's call surface for the path as data: parameter names and defaults of the functions / constructors
    the launcher touches, read with inspect.signature from the reference's own files.  tests/test_boundary.py holds cosa_amd's
    equivalents to it."""
    import inspect
    import json
    seg, par, th, ev = ref_loader.seg_helper(), ref_loader.par_module(), ref_loader.torch_helper_fns(), ref_loader.evaluation_module()

    def sig(f):
        out = []
        for name, p_ in inspect.signature(f).parameters.items():
            if name == "self":
                continue
            d = p_.default
            out.append([name, None if d is inspect.Parameter.empty else repr(d), p_.kind.name])
        return out
    table = {}
    for name in ("multi_scale_camseg", "multi_scale_camsegv3", "cam_validation", "cam2mask", "seg_loss", "get_energy_loss", "seg_refine_by_label",
                 "cam_loss", "cam_to_label", "seg_validation", "rungmm"):
        table["utils.seg_helper." + name] = sig(getattr(seg, name))
    table["utils.seg_helper.DenseEnergyLoss.__init__"] = sig(seg.DenseEnergyLoss.__init__)
    table["utils.seg_helper.DenseEnergyLoss.forward"] = sig(seg.DenseEnergyLoss.forward)
    table["utils.seg_helper.DynamicQueue.__init__"] = sig(seg.DynamicQueue.__init__)
    table["models.PAR.PAR.__init__"] = sig(par.PAR.__init__)
    table["models.PAR.PAR.forward"] = sig(par.PAR.forward)
    table["utils.torch_helper.PolyWarmupAdamW.__init__"] = sig(th.PolyWarmupAdamW.__init__)
    for name in ("denormalize_img", "setup_seed", "save_best", "compute_mAP"):
        table["utils.torch_helper." + name] = sig(getattr(th, name))
    table["utils.torch_helper.EMAtracker.__init__"] = sig(th.EMAtracker.__init__)
    for name in ("scores", "pseudo_scores"):
        table["utils.evaluation." + name] = sig(getattr(ev, name))
    with open(os.path.join(OUT, "ref_signatures.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print("signatures:", len(table), "entries")


def main():
    import sys
    assert ref_loader.available(), "reference tree not present"
    os.makedirs(OUT, exist_ok=True)
    c_oracle.build()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "--only-vit-base" in sys.argv:          # (round 6: one new fixture; the others stay as committed)
        gen_vit_base(np.random.default_rng(20))
        return
    gen_par(np.random.default_rng(11))
    gen_cam2mask(np.random.default_rng(12))
    gen_camseg(np.random.default_rng(13))
    gen_bilateral(np.random.default_rng(14))
    gen_misc(np.random.default_rng(15))
    gen_vit(np.random.default_rng(16))
    gen_eval(np.random.default_rng(17))
    gen_gmm(np.random.default_rng(18))
    gen_augment(np.random.default_rng(19))
    gen_vit_base(np.random.default_rng(20))
    gen_signatures()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
