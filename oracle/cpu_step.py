"""CPU restatement of ONE training iteration (main.py:106-252) from oracle pieces.  TEST INFRASTRUCTURE ONLY.

Used by (1) bench.py's `cpu_baseline` leg (timed on the GPU box's host cores) and (2) the
whole-step parity test.  The reference has no CPU path as shipped (hard-coded .cuda(),
nccl-only init), so this file composes the oracle's restatements in the reference's order:
teacher multi-scale pass -> student pass -> cls losses -> cam2mask (x2) -> seg loss -> dense-energy
loss -> refined-seg -> cam loss -> weighted sum -> backward -> AdamW (poly/warm-up LR) -> EMA.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle, torch_oracle as to


class CpuStep:
    def __init__(self, state_dict, num_classes=21, aux_layer=-4, vit_kwargs=None, args=None):
        kw = dict(num_classes=num_classes, aux_layer=aux_layer)
        kw.update(vit_kwargs or {})
        self.student = to.OracleViT(**kw)
        self.teacher = to.OracleViT(**kw)
        self.student.load_named(state_dict)
        self.teacher.load_named(state_dict)
        for p in self.teacher.parameters():
            p.requires_grad = False
        self.student.p("encoder.pos_embed").requires_grad = False
        for n in ("encoder.head.weight", "encoder.head.bias"):
            self.student.p(n).requires_grad = False
        self.a = dict(lr=6e-5, wt_dec=1e-2, lrscale=10.0, max_iters=32000, warmup_iters=6000, momentum=0.9994, seg_weight=0.1,
                      cam_weight=0.05, reg_weight=0.05, seg_softmaxtemp=0.01, high_thre=0.7, low_thre=0.25, high_thre_aux=0.7,
                      low_thre_aux=0.25, scales=[1.0, 0.5, 1.5], par=None)
        self.a.update(args or {})
        names = self.student.named_state().keys()
        g = [[], [], [], []]
        for n in names:
            p = self.student.p(n)
            if not p.requires_grad:
                continue
            if n.startswith("encoder."):
                g[1 if "norm" in n else 0].append(p)
            elif n.startswith("decoder."):
                g[3].append(p)
            else:
                g[2].append(p)
        lr = self.a["lr"]
        self.base_lr = [lr, lr, lr * self.a["lrscale"], lr * self.a["lrscale"]]
        self.opt = torch.optim.AdamW([{"params": g[i], "lr": self.base_lr[i], "weight_decay": self.a["wt_dec"]} for i in range(4)],
                                     lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=self.a["wt_dec"])
        self.global_step = 0

    def losses(self, wimg, simg, cls_label, img_box, n_iter, timers=None):
        a = self.a
        t = time.perf_counter

        def tick(name, t0):
            if timers is not None:
                timers[name] = timers.get(name, 0.0) + (t() - t0)

        t0 = t()
        cam_ps, cam_aux_ps, seg_ps = to.multi_scale_camseg(self.teacher, wimg, a["scales"])
        tick("teacher_fwd", t0)
        t0 = t()
        cls, cls_aux, _x4, seg_pred, cam_pred, cam_aux_pred = self.student(simg)
        tick("student_fwd", t0)
        cls_loss = F.multilabel_soft_margin_loss(cls, cls_label)
        cls_loss_aux = F.multilabel_soft_margin_loss(cls_aux, cls_label)
        t0 = t()
        denorm = c_oracle.denormalize_img(simg.numpy()) if a["par"] else None
        boxes = np.asarray(img_box, np.int32)
        mask = torch.from_numpy(c_oracle.cam2mask(denorm, boxes, cam_ps.numpy(), cls_label.numpy(), a["high_thre"], a["low_thre"],
                                                  2, par=a["par"]))
        mask_aux = torch.from_numpy(c_oracle.cam2mask(denorm, boxes, cam_aux_ps.numpy(), cls_label.numpy(), a["high_thre_aux"],
                                                      a["low_thre_aux"], 2, par=a["par"]))
        tick("cam2mask", t0)
        t0 = t()
        seg_up = F.interpolate(seg_pred, size=mask.shape[1:], mode="bilinear", align_corners=False)
        seg_loss = 0.5 * to.seg_loss(seg_up, mask) + 0.5 * to.seg_loss(seg_up, mask_aux)
        tick("seg_loss", t0)
        t0 = t()
        prob = F.softmax(seg_up, 1)
        b, K, h, w = prob.shape
        roi = torch.zeros(b, h, w)
        for i, bx in enumerate(boxes):
            roi[i, bx[0]:bx[1], bx[2]:bx[3]] = 1
        img255 = simg * torch.tensor((58.395, 57.12, 57.375))[None, :, None, None] + \
            torch.tensor((123.675, 116.28, 103.53))[None, :, None, None]
        reg_loss = _DenseEnergy.apply(img255, prob, roi, mask.to(torch.uint8).unsqueeze(1))
        tick("bilateral", t0)
        valid_seg = to.seg_refine_by_label(seg_ps, cls_label, a["seg_softmaxtemp"])
        cam_loss = to.cam_loss(cam_pred, valid_seg)
        if n_iter <= a["warmup_iters"]:
            loss = cls_loss + cls_loss_aux + 0.0 * seg_loss + 0.0 * cam_loss + 0.0 * reg_loss
        else:
            loss = cls_loss + cls_loss_aux + a["seg_weight"] * seg_loss + a["cam_weight"] * cam_loss + a["reg_weight"] * reg_loss
        return loss, dict(overall_loss=loss.detach(), cls_loss=cls_loss.detach(), cls_aux_loss=cls_loss_aux.detach(),
                          seg_loss=seg_loss.detach(), cam_loss=cam_loss.detach(), reg_loss=reg_loss.detach(), mask=mask,
                          mask_aux=mask_aux, cam_ps=cam_ps, cam_aux_ps=cam_aux_ps, seg_ps=seg_ps)

    def step(self, wimg, simg, cls_label, img_box, n_iter, timers=None):
        loss, logs = self.losses(wimg, simg, cls_label, img_box, n_iter, timers)
        t0 = time.perf_counter()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if timers is not None:
            timers["student_bwd"] = timers.get("student_bwd", 0.0) + (time.perf_counter() - t0)
        t0 = time.perf_counter()
        lr_now = [to.poly_warmup_lr(self.global_step, b, 1500, self.a["max_iters"]) for b in self.base_lr]
        for g, lr in zip(self.opt.param_groups, lr_now):
            if lr is not None:
                g["lr"] = lr
        self.opt.step()
        self.global_step += 1
        m = self.a["momentum"]
        with torch.no_grad():
            for k in self.student.P.keys():
                self.teacher.P[k].mul_(m).add_((1 - m) * self.student.P[k])
        if timers is not None:
            timers["optim_ema"] = timers.get("optim_ema", 0.0) + (time.perf_counter() - t0)
        return logs


class _DenseEnergy(torch.autograd.Function):
    """DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5) with the C-oracle filter."""

    @staticmethod
    def forward(ctx, img255, prob, roi, label_u8):
        sf, w = 0.5, 1e-7
        s_img = F.interpolate(img255, scale_factor=sf, recompute_scale_factor=True)
        s_seg = F.interpolate(prob, scale_factor=sf, mode="bilinear", align_corners=False, recompute_scale_factor=True)
        s_roi = F.interpolate(roi.unsqueeze(1), scale_factor=sf, recompute_scale_factor=True).squeeze(1)
        s_lab = F.interpolate(label_u8.float(), scale_factor=sf, mode="nearest", recompute_scale_factor=True)
        unl = (s_lab.long() == 255).squeeze(1)
        loss, AS = c_oracle.dense_energy_forward(s_img.numpy(), s_seg.detach().numpy(), s_roi.numpy(), unl.numpy().astype(np.uint8),
                                                 15.0, 100.0 * sf)
        ctx.save_for_backward(torch.from_numpy(AS), s_roi)
        ctx.shape = prob.shape
        return torch.tensor([loss * w], dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        AS, s_roi = ctx.saved_tensors
        N = AS.shape[0]
        g_seg = -2.0 * 1e-7 * g * AS / N * s_roi.unsqueeze(1)
        # adjoint of the exact x0.5 bilinear (2x2 mean): spread a quarter to each source pixel
        g_full = F.interpolate(g_seg, scale_factor=2, mode="nearest") * 0.25
        return None, g_full, None, None
