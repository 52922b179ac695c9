"""PAR -- pixel-adaptive refinement, HIP implementation behind the reference's module interface.

Reference: models/PAR.py:26-91 (`PAR(dilations, num_iter)`, `forward(imgs, masks)`).
The whole batch runs in `num_iter + 1` kernel launches (affinity once, then one launch per
propagation step) instead of the reference's ~6 dilated convolutions per step per image.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _C


class PAR(nn.Module):
    def __init__(self, dilations, num_iter):
        super().__init__()
        self.dilations = [int(d) for d in dilations]
        self.num_iter = int(num_iter)
        self.w1 = 0.3
        self.w2 = 0.01

    def forward(self, imgs, masks):
        """imgs [B,3,h,w] in [0,1]; masks [B,K,H,W] -> refined [B,K,h,w] (float32)."""
        _C.require_cuda(imgs, masks)
        if masks.shape[-2:] != imgs.shape[-2:]:
            # models/PAR.py:66 -- plain resize, done by torch (plumbing); identity when sizes agree
            masks = F.interpolate(masks, size=imgs.shape[-2:], mode="bilinear", align_corners=True)
        imgs = imgs.contiguous().float()
        masks = masks.contiguous().float()
        B, K, h, w = masks.shape
        if imgs.shape[0] != B or imgs.shape[1] != 3:
            raise ValueError("PAR: imgs must be [B,3,h,w] with the same batch as masks")
        out = torch.empty_like(masks)
        L = _C.lib()
        nd = len(self.dilations)
        ws = _C.workspace(L.cosa_par_workspace_bytes(B, K, h, w, nd), imgs.device, "par")
        _C.check(L.cosa_par_forward(_C.ptr(imgs), _C.ptr(masks), _C.ptr(out), B, K, h, w, _C.int_array(self.dilations), nd,
                                    self.num_iter, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_par_forward")
        return out
