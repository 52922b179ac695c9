"""models -- `build_model(args)` / `VITNetwork` with the reference's call surface.

Reference: models/__init__.py:13-79 (build_model), :82-206 (VITNetwork), models/decoder/conv_head.py:11-41
(LargeFOV).  state_dict keys are identical to the reference: encoder.* (timm ViT names),
decoder.conv6/7/8.weight, classifier.weight, aux_classifier.weight.
"""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _C, nn_ops
from . import vit as vitencoder
from .PAR import PAR  # noqa: F401


class LargeFOV(nn.Module):
    """models/decoder/conv_head.py:11-41: 3x3 d5 (768->512) ReLU, 3x3 d5 (512->512) ReLU, 1x1 -> classes; no bias."""

    def __init__(self, in_planes, out_planes, dilation=5):
        super().__init__()
        self.embed_dim = 512
        self.dilation = dilation
        self.conv6 = nn.Conv2d(in_planes, self.embed_dim, 3, padding=dilation, dilation=dilation, bias=False)
        self.conv7 = nn.Conv2d(self.embed_dim, self.embed_dim, 3, padding=dilation, dilation=dilation, bias=False)
        self.conv8 = nn.Conv2d(self.embed_dim, out_planes, 1, bias=False)

    batch_invariant = False        # set by VITNetwork together with its own flag

    def forward_tokens(self, tok, B, h, w):
        """no-grad 16-bit path (bf16 or fp16 tokens): both dilated convs as implicit-GEMM MFMA kernels on the NHWC tokens, conv8 as a bare GEMM.
        tok [B, h*w, 768] bf16 (may be the strided `tokens[:, 1:]` view) -> seg [B, classes, h, w] fp32"""
        c, dt = nn_ops.cast_param, tok.dtype
        y = nn_ops.conv3x3_dilated_tokens(tok, c(self.conv6.weight, dt), B, h, w, self.dilation, relu=True)
        y = nn_ops.conv3x3_dilated_tokens(y.view(B, h * w, -1), c(self.conv7.weight, dt), B, h, w, self.dilation, relu=True)
        w8 = c(self.conv8.weight, dt).reshape(self.conv8.weight.shape[0], -1)
        seg = nn_ops.head_linear(y.view(B, h * w, -1), w8.contiguous(), round_bf16=True)
        if seg is None:
            seg = nn_ops.reference_op("linear", f"LargeFOV.conv8 ({tuple(w8.shape)}, {dt})", y, w8, None).float()
        return seg.view(B, h, w, -1).permute(0, 3, 1, 2).contiguous()

    def forward_tokens_train(self, tok, B, h, w):
        """bf16 training path (autograd): conv6 / conv7 forward, input- and weight-gradient on our MFMA kernels
        (nn_ops.DilatedConvReluFn), conv8 (1x1) as a GEMM over the tokens.  tok [B, h*w, 768] bf16 -> seg [B, classes, h, w] fp32"""
        if tok.stride(2) != 1 or (B > 1 and tok.stride(0) % tok.stride(1)):
            tok = tok.contiguous()
        y = nn_ops.DilatedConvReluFn.apply(tok, self.conv6.weight, B, h, w, self.dilation)
        y = nn_ops.DilatedConvReluFn.apply(y.view(B, h * w, -1), self.conv7.weight, B, h, w, self.dilation)
        seg = nn_ops.narrow_linear(y, self.conv8.weight)
        if seg is None:
            w8 = nn_ops.cast_param(self.conv8.weight, torch.bfloat16).reshape(self.conv8.weight.shape[0], -1)
            seg = nn_ops.reference_op("linear", f"LargeFOV.conv8 with autograd ({tuple(self.conv8.weight.shape)})", y, w8, None).float()
        return seg.view(B, h, w, -1).permute(0, 3, 1, 2).contiguous()

    def forward_nhwc(self, x, dt):
        """x: [B,C,h,w] view with channels-last strides (tokens are NHWC already): channel counts / dtypes the implicit-GEMM conv kernels do
        not cover -- outside the envelope (conv_head.py:32-41 as a reference operator of the test-suite)"""
        c = nn_ops.cast_param
        return nn_ops.reference_op("largefov", f"LargeFOV decoder ({dt}, {self.conv6.weight.shape[1]} channels, grad {torch.is_grad_enabled()})",
                                   x, c(self.conv6.weight, dt), c(self.conv7.weight, dt), c(self.conv8.weight, dt), self.dilation)


class VITNetwork(nn.Module):
    """models/__init__.py:82-206"""

    # narrow heads (CAM / aux CAM / conv8 / classification) of the no-grad paths on the batch-invariant kernel instead of library
    # GEMMs: evaluate() turns it on so that grouped loader items reproduce the one-at-a-time scores exactly; the training step's
    # teacher keeps the (tuned) library GEMMs, which are faster there (skinny VALU kernel: 1.57 ms per step against ~0.4 ms)
    batch_invariant_heads = False

    def __init__(self, backbone, num_classes, pretrained=True, aux_layer=-3, isgap=False, decoder='LargeFOV',
                 compute_dtype=torch.bfloat16, pretrained_path=None):
        super().__init__()
        assert decoder in ['LargeFOV'], "cosa_amd builds the LargeFOV decoder (the run scripts' default)"
        self.num_classes = num_classes
        self.encoder = getattr(vitencoder, backbone)(pretrained=pretrained, pretrained_path=pretrained_path, aux_layer=aux_layer,
                                                     compute_dtype=compute_dtype)
        self.in_channels = [self.encoder.embed_dim] * 4
        self.isgap = isgap
        self.decoder = LargeFOV(in_planes=self.in_channels[-1], out_planes=self.num_classes)
        self.isdecoder_trans = False
        self.classifier = nn.Conv2d(self.in_channels[-1], self.num_classes - 1, kernel_size=1, bias=False)
        self.aux_classifier = nn.Conv2d(self.in_channels[-1], self.num_classes - 1, kernel_size=1, bias=False)
        self.compute_dtype = compute_dtype

    def set_compute_dtype(self, dt):
        self.compute_dtype = dt
        self.encoder.compute_dtype = dt
        return self

    def set_nograd_precision(self, mode):
        """operand precision of the no-grad passes (teacher pseudo-labels, evaluation): "bf16" (8 significant bits), "fp16" (11; the
        same kernels built for fp16 operands), "bf16x3" (16; hi + lo bf16 halves, three MFMA terms), "fp16x3" (22: hi + lo fp16 halves, the same three
        terms at the same cost), "fp16c8" (fp16 + 8-bit correction
        terms on the block-scaled MFMA, ~14 bits at twice the 16-bit work; attention operands plain fp16) or "fp16c4" (the same with FP4 MX-block
        correction terms at 4x the fp16 rate: ~1.6x the 16-bit work); "-n" suffix: blocks from index n on plain fp16, "-nmk": their MLP halves (fc1, fc2) from
        block k <= n on: DESIGN.md section 3"""
        base, sep, tail = mode.partition("-")            # "fp16c8-9": fp16c8 with the blocks from index 9 on plain fp16 operands;
        m = re.fullmatch(r"(\d+)(?:m(\d+))?(q?)", tail) if tail else None    # "fp16c4-9m7": ... and the MLP halves already from block 7 on;
        #                                                                         trailing "q": the qkv projections of the corrected blocks on plain fp16 too
        mx = re.fullmatch(r"(?:x(\d+)(?:m(\d+))?)?(?:c(\d+))?", tail) if tail and not m else None     # "fp16c8-x6": the blocks BELOW index 6 on bf16x3
        #                       operands (round 5); "fp16c8-x6m4": their attention halves below 6, their MLP halves below 4; "fp16c8-x2c6" / "fp16c8-c6":
        #                       the blocks from index 6 on take qkv / fc1 / fc2 on fp16c4 operands (fp16c8 base only)
        assert base in ("bf16", "fp16", "bf16x3", "fp16x3", "fp16c8", "fp16c4") and (bool(tail) == bool(sep)) and \
            (not tail or (base in ("fp16c8", "fp16c4") and (m or mx))), mode
        self.set_compute_dtype(torch.float16 if base in ("fp16", "fp16x3", "fp16c8", "fp16c4") else torch.bfloat16)
        # "fp16x3" (round 6): the three-term path with fp16 halves (hi + lo: 11 + 11 significant bits; bf16x3: 8 + 8) -- same kernels, same cost
        self.encoder.precision = "bf16x3" if base == "fp16x3" else (base if base in ("bf16x3", "fp16c8", "fp16c4") else None)
        self.encoder.x3_dtype = torch.float16 if base == "fp16x3" else torch.bfloat16
        self.encoder.c8_plain_from = int(m.group(1)) if m else None
        self.encoder.c8_plain_mlp_from = int(m.group(2)) if m and m.group(2) else None
        self.encoder.c8_plain_qkv = bool(m and m.group(3))
        assert not mx or not mx.group(3) or base == "fp16c8", mode
        self.encoder.x3_until = int(mx.group(1)) if mx and mx.group(1) else None
        self.encoder.x3_mlp_until = int(mx.group(2)) if mx and mx.group(2) else None
        self.encoder.c4_from = int(mx.group(3)) if mx and mx.group(3) else None
        return self

    def get_param_groups(self):
        """models/__init__.py:126-144: backbone; backbone norms; cls heads; decoder"""
        groups = [[], [], [], []]
        for name, param in self.encoder.named_parameters():
            groups[1 if "norm" in name else 0].append(param)
        groups[2].append(self.classifier.weight)
        groups[2].append(self.aux_classifier.weight)
        groups[3].extend(self.decoder.parameters())
        return groups

    def _pool(self, tok):
        return tok.mean(dim=1) if self.isgap else tok.amax(dim=1)

    def _cam(self, tok, weight, B, h, w, detach_feat, detach_w):
        """1x1 conv over tokens == tokens @ W^T; returned as fp32 NCHW [B,C,h,w] (models/__init__.py:190-192).  Own kernels: the
        exact-fp32 MFMA narrow-head kernel (no-grad paths; fp32 tokens of the fused teacher keep the head in fp32) and, with autograd,
        nn_ops.NarrowLinearFn; other shapes are outside the envelope (nn_ops.reference_op)."""
        dt = tok.dtype
        if detach_feat:
            tok = tok.detach()
        cam = None
        if tok.is_cuda and not torch.is_grad_enabled():
            wgt = nn_ops.cast_param(weight, dt).reshape(weight.shape[0], -1)
            cam = nn_ops.head_linear(tok, wgt.contiguous(), round_bf16=tok.dtype != torch.float32)
        elif tok.is_cuda and dt == torch.bfloat16:
            cam = nn_ops.narrow_linear(tok.reshape(-1, tok.shape[-1]).contiguous(), weight.detach() if detach_w else weight)
        if cam is None:
            wgt = nn_ops.cast_param(weight, dt).reshape(weight.shape[0], -1)
            cam = nn_ops.reference_op("linear", f"CAM head ({tuple(weight.shape)}, tokens {dt})", tok, wgt.detach() if detach_w else wgt, None).float()
        return cam.reshape(B, h, w, -1).permute(0, 3, 1, 2).contiguous()

    def _cls_head(self, pooled, weight, dt):
        """classification logits from the pooled tokens (models/__init__.py:196-204), on the same narrow-head kernels"""
        y = None
        if pooled.is_cuda and not torch.is_grad_enabled():
            y = nn_ops.head_linear(pooled.to(dt).unsqueeze(1).contiguous(), nn_ops.cast_param(weight, dt).reshape(weight.shape[0], -1).contiguous(),
                                   round_bf16=dt != torch.float32)
        elif pooled.is_cuda and dt == torch.bfloat16:
            y = nn_ops.narrow_linear(pooled.to(dt).contiguous(), weight)
        if y is None:
            y = nn_ops.reference_op("linear", f"classification head ({tuple(weight.shape)}, {dt})", pooled.to(dt),
                                    nn_ops.cast_param(weight, dt).reshape(weight.shape[0], -1), None).float()
        return y

    def refresh_shadows(self):
        """16-bit shadows of every parameter, current as of this call (nn_ops.ensure_shadows); a no-op in fp32 mode / on the host"""
        if self.compute_dtype != torch.float32 and self.classifier.weight.is_cuda:
            nn_ops.ensure_shadows(self, self.compute_dtype)

    def forward_multi(self, xs, flip_pairs=False, need_cls=True):
        """forward() for several image batches of different sizes at once (no-grad bf16 only): the encoder runs all of
        them through shared GEMM / LayerNorm launches (VisionTransformer._forward_features_fused_multi).  flip_pairs: every batch x
        stands for cat(x, x.flip(-1)) (the multi-scale passes of seg_helper.py:241-246); the mirror images then exist only as im2col rows.
        need_cls=False leaves out global pooling and the two classification heads (their slots in the result tuples are None)."""
        for x in xs:
            _C.require_cuda(x)
        self.refresh_shadows()
        feats = self.encoder._forward_features_fused_multi(xs, flip_pairs=flip_pairs)
        if flip_pairs:          # _heads reads only the shape of its image argument
            xs = [torch.empty((2 * x.shape[0],) + tuple(x.shape[1:]), device="meta") for x in xs]
        return [self._heads(x, f, False, False, 'none', need_cls=need_cls) for x, f in zip(xs, feats)]

    def can_forward_multi(self, x):
        return self.encoder.use_fused(x)

    def forward(self, x, cam_only=False, seg_only=False, detach='none'):
        """models/__init__.py:163-206 -> (cls, cls_aux, feat[B,768,h,w], seg, cam, cam_aux)"""
        assert detach in ['all', 'feat', 'none', 'cls']
        _C.require_cuda(x)                                            # MI355X only: there is no CPU path
        if not torch.is_grad_enabled():
            self.refresh_shadows()
        return self._heads(x, self.encoder.features_ex(x), cam_only, seg_only, detach)

    def _heads_train_f32(self, x, feats, cam_only, seg_only, detach):
        """the heads of the training path on the fp32 residual stream (models/__init__.py:163-206): every consumer of the final tokens
        (decoder, CAM head, pooled classification head) and of the auxiliary tokens (aux CAM head, aux classification head) takes its own
        bf16 view of ONE cast (nn_ops.patch_fanout_bf16), whose backward adds the consumers' gradients in fp32"""
        B = x.shape[0]
        p = self.encoder.patch_size
        h, w = x.shape[-2] // p, x.shape[-1] // p
        # (patch tokens only: the class token feeds no head.  The returned feature map x4 gets a view of its own: as a view of the decoder's
        # input its gradient would meet the decoder's in bf16 before the fp32 junction -- ADVICE r4; no loss of the reference differentiates
        # through it, and the junction kernel adds at most three live gradients: a fourth raises instead of rounding silently)
        t_dec, t_cam, t_cls, t_x4 = nn_ops.patch_fanout_bf16(feats.final, 4)
        a_cam, a_cls = nn_ops.patch_fanout_bf16(feats.aux, 2)
        x4 = t_x4.reshape(B, h, w, -1).permute(0, 3, 1, 2)
        seg = self.decoder.forward_tokens_train(t_dec, B, h, w)
        if seg_only:
            return seg
        cam = self._cam(t_cam, self.classifier.weight, B, h, w, detach == 'feat', detach == 'cls')
        cam_aux = self._cam(a_cam, self.aux_classifier.weight, B, h, w, detach == 'feat', detach == 'cls')
        if detach == 'all':
            cam, cam_aux = cam.detach(), cam_aux.detach()
        if cam_only:
            return cam, cam_aux
        dt = self.compute_dtype
        cls_x4 = self._cls_head(self._pool(t_cls), self.classifier.weight, dt)
        cls_aux = self._cls_head(self._pool(a_cls), self.aux_classifier.weight, dt)
        return cls_x4, cls_aux, x4, seg, cam, cam_aux

    def _heads(self, x, feats, cam_only, seg_only, detach, need_cls=True):
        if isinstance(feats, vitencoder.FP32Tokens):
            return self._heads_train_f32(x, feats, cam_only, seg_only, detach)
        dt = self.compute_dtype
        B = x.shape[0]
        _, tok, tok_aux, tok32 = feats
        p = self.encoder.patch_size
        h, w = x.shape[-2] // p, x.shape[-1] // p
        # NCHW view, channels-last strides (no copy).  The fused no-grad path returns the fp32 tokens as the feature map, like the reference (round 6:
        # it returned its 16-bit copy -- 3e-3 of the range in the bf16 builds against the reference at ViT-B width; no loss reads it)
        x4 = (tok if tok32 is None else tok32).reshape(B, h, w, -1).permute(0, 3, 1, 2)
        if tok32 is not None and tok.dtype in vitencoder._OP16 and self.decoder.conv6.weight.shape[1] % 64 == 0:
            seg = self.decoder.forward_tokens(tok, B, h, w)            # fused no-grad path: own implicit-GEMM convs
        elif tok.dtype == torch.bfloat16 and tok.is_cuda and self.decoder.conv6.weight.shape[1] % 128 == 0 \
                and torch.is_grad_enabled():
            seg = self.decoder.forward_tokens_train(tok, B, h, w)      # training: forward + both gradients on own kernels
        else:
            seg = self.decoder.forward_nhwc(x4, dt).float().contiguous()
        if seg_only:
            return seg
        cam = self._cam(tok if tok32 is None else tok32, self.classifier.weight, B, h, w, detach == 'feat', detach == 'cls')
        cam_aux = self._cam(tok_aux, self.aux_classifier.weight, B, h, w, detach == 'feat', detach == 'cls')
        if detach == 'all':
            cam, cam_aux = cam.detach(), cam_aux.detach()
        if cam_only:
            return cam, cam_aux
        if not need_cls:        # the training loop's teacher passes never read the classification logits (seg_helper.py:247-249 drops them)
            return None, None, x4, seg, cam, cam_aux
        # the classification logits read the SAME tokens as the CAM heads (round 6: the fused no-grad path pooled its 16-bit copy of the final
        # tokens while the CAMs came from the fp32 ones -- 2e-3 of the logits' range against the reference at ViT-B width,
        # tests/test_network_gpu.py::test_hip_network_at_vit_b_width_vs_reference_golden; evaluation's mAP reads them)
        p_fin = tok if tok32 is None else tok32
        cls_x4 = self._cls_head(self._pool(p_fin), self.classifier.weight, dt if p_fin.dtype != torch.float32 else torch.float32)
        cls_aux = self._cls_head(self._pool(tok_aux), self.aux_classifier.weight, dt if tok_aux.dtype != torch.float32 else torch.float32)
        return cls_x4, cls_aux, x4, seg, cam, cam_aux


def build_model(args):
    """models/__init__.py:13-24 (the `vit` branch; the other zoos are commented out in the reference too)."""
    model = getattr(args, "model", "vit")
    if model != 'vit':
        raise NotImplementedError("cosa_amd builds args.model == 'vit' (the only live branch of the reference)")
    dt = getattr(args, "compute_dtype", torch.bfloat16)
    return VITNetwork(backbone=args.backbone, num_classes=args.num_classes, pretrained=getattr(args, "pretrained", False),
                      aux_layer=args.aux_layer, isgap=getattr(args, "isgap", False),
                      decoder=getattr(args, "decoder", "LargeFOV"), compute_dtype=dt,
                      pretrained_path=getattr(args, "pretrained_path", None))
