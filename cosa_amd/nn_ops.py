"""Differentiable building blocks of the ViT hot path, backed by the HIP kernels in libcosa_hip.so.

Every op here is an explicit torch.autograd.Function around C-ABI calls (no tracing compiler, no
Triton, no library GEMM).  Shapes / dtypes outside the HIP kernels' envelope RAISE; torch's own
operators exist only in the test-suite (tests/torch_reference.py installs them for the host-logic tests: reference_op below).
"""
import ctypes
import math

import os

import time

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _C


# --------------------------------------------------------------------------------------------
# attention  (models/vit/vit.py:119-137)
# --------------------------------------------------------------------------------------------
_flops = {}   # algorithmic FLOPs issued per profiled kernel (read by bench.py)


# --------------------------------------------------------------------------------------------
# outside the HIP path's envelope: ONE operator implementation in the product
# --------------------------------------------------------------------------------------------
# The product path is ViT-B (embed 768) on 16-bit MFMA operands; every operator of it is a kernel of this repository.  Shapes / dtypes
# outside that envelope (an fp32 "parity mode", the 128-wide toy encoders of the golden vectors) have NO implementation here: every such
# site calls reference_op(), which raises CosaError.  The host-logic tests (module wiring, state-dict names, the loss algebra on toy
# encoders) install torch (ATen) implementations of the named operators from tests/torch_reference.py -- round 4 kept those ATen branches
# in these modules behind a switch; they live with the tests now.
_reference_ops = None          # None in the product; tests/torch_reference.py: {"linear": f, "attention": f, "layer_norm": f, "largefov": f}


def reference_op(name, what, *args, **kw):
    """called at every site that would leave the HIP path: raises unless a test has installed reference operators"""
    if _reference_ops is None:
        raise _C.CosaError(f"{what}: no HIP kernel covers this shape / dtype and cosa_amd has no second backend "
                           "(torch reference operators exist only in the test-suite: tests/torch_reference.py)")
    return _reference_ops[name](*args, **kw)


class KernelStamps:
    """Device-side launch spans for the dominant kernel (bench.py's roofline leg).

    HIP events cannot be recorded inside a captured hipGraph on ROCm ("External events are disallowed"), so the attention
    forward kernel can stamp the 100 MHz device wall clock itself: min start / max end over its workgroups, into a slot
    fixed at launch-issue (= capture) time.  `reset()` is a device op, so it is captured too and re-arms every replay.
    A slot is 64 (start, end) shards (workgroup id & 63): thousands of atomics on ONE address serialise at ~20 ns each."""
    SHARDS = 64

    def __init__(self, device, max_launches=4096):
        self.buf = torch.zeros((max_launches, self.SHARDS, 2), dtype=torch.int64, device=device)
        self.half = max_launches // 2
        self.n = 0                  # slots [0, half): launches of the teacher section (captured in its hipGraph, re-armed by the captured reset)
        self.flops = []             # per slot: (algorithmic FLOPs, issued MFMA FLOPs -- more for operand formats with correction terms)
        self.eager = False          # slots [half, ..): the step's eager launches (student), re-armed on their own stream by begin_eager()
        self.n_eager = 0
        self.flops_eager = []

    def reset(self):
        """re-arm the teacher section's slots (a device op: captured with the section and replayed with it)"""
        self.buf[: self.half, :, 0] = torch.iinfo(torch.int64).max
        self.buf[: self.half, :, 1] = 0

    def begin_eager(self):
        """the teacher section has been issued: the launches that follow are eager ones on the current stream.  Their slots are re-armed
        here, on that stream -- the captured reset runs on the teacher's side stream, concurrently with them (a reset landing between a
        kernel's two stamps used to leave (start = max, end = t): negative spans in the round-1 / early round-2 bench lines)."""
        self.eager = True
        self.n_eager = 0
        self.flops_eager = []
        self.buf[self.half:, :, 0] = torch.iinfo(torch.int64).max
        self.buf[self.half:, :, 1] = 0

    def begin_section(self):
        self.eager = False
        self.n = 0
        self.flops = []

    def next_slot(self, flops, issued=None):
        ent = (flops, flops if issued is None else issued)
        if self.eager:
            i = self.half + self.n_eager
            self.n_eager += 1
            self.flops_eager.append(ent)
        else:
            i = self.n
            self.n += 1
            self.flops.append(ent)
        assert i < self.buf.shape[0] and (self.eager or i < self.half), "KernelStamps: out of slots"
        return ctypes.c_void_p(self.buf.data_ptr() + 16 * self.SHARDS * i)

    def read(self):
        """-> (n_launches, total_seconds, total_flops) of the launches stamped since the last re-arm (after a sync)"""
        b = torch.cat([self.buf[: self.n], self.buf[self.half: self.half + self.n_eager]])
        b = torch.stack([b[:, :, 0].amin(1), b[:, :, 1].amax(1)], 1).cpu()          # min start / max end over the shards
        fl_all = list(self.flops) + list(self.flops_eager)
        ok = (b[:, 1] > 0) & (b[:, 0] < b[:, 1])
        ticks = (b[:, 1] - b[:, 0])[ok]
        fl = sum(f[0] for f, k in zip(fl_all, ok.tolist()) if k)
        self.last_issued = sum(f[1] for f, k in zip(fl_all, ok.tolist()) if k)
        return int(ok.sum()), float(ticks.sum()) * 1e-8, fl


stamps = None     # set by bench.py to a KernelStamps to switch the stamping on (attention forward)
gemm_stamps = None   # likewise for the projection GEMMs (persistent 256x256 kernel)


def _attn_fwd(qkv, B, N, H, out=None, nograd=False):
    """fused attention forward on packed qkv [B,N,3,H,64] (bf16 or fp16 operands: the kernel build is picked by qkv's dtype).
    nograd: a pass without backward (teacher, evaluation) -- flag bit 10 lets the kernel hold Q pre-scaled in the operand type and take the
    softmax's running maximum through the score MFMAs (csrc/attn_kernels.hip: AUGM); the passes that keep LSE for a backward do not."""
    dt = qkv.dtype
    if out is None:
        out = torch.empty((B, N, H * 64), device=qkv.device, dtype=dt)
    lse = torch.empty((B, H, N), device=qkv.device, dtype=torch.float32)
    ws = _C.workspace(_C.fn16("cosa_attn_workspace_bytes", dt)(B, N, H), qkv.device, "attn")
    fl = 4.0 * B * H * N * N * 64
    st = stamps.next_slot(fl) if stamps is not None else None
    with _C.profiled("attn_fwd"):
        _C.check(_C.fn16("cosa_attn_fwd", dt)(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, 0x400 if nograd else 0, st,
                                              _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_fwd")
    _flops["attn_fwd"] = _flops.get("attn_fwd", 0) + fl
    return out, lse


class FusedAttention(Function):
    """softmax(q k^T / sqrt(64)) v over packed qkv [B,N,3*H*64] (bf16) -> [B,N,H*64].

    Forward is the LDS-tiled MFMA kernel (cosa_attn_fwd); backward is the pair of MFMA kernels behind
    cosa_attn_bwd (P recomputed from the saved log-sum-exp, nothing of size N^2 in HBM, no atomics)."""

    @staticmethod
    def forward(ctx, qkv, H):
        B, N, D3 = qkv.shape
        assert D3 == 3 * H * 64 and qkv.dtype == torch.bfloat16
        qkv = qkv.contiguous()
        out, lse = _attn_fwd(qkv, B, N, H)
        ctx.save_for_backward(qkv, out, lse)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        H = ctx.H
        B, N, _ = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        L = _C.lib()
        ws = _C.workspace(L.cosa_attn_bwd_workspace_bytes(B, N, H), qkv.device, "attn_bwd")
        with _C.profiled("attn_bwd"):
            _C.check(L.cosa_attn_bwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(dout), _C.ptr(lse), _C.ptr(dqkv), B, N, H, 64, 0.125,
                                     _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_bwd")
        _flops["attn_bwd"] = _flops.get("attn_bwd", 0) + 10.0 * B * H * N * N * 64
        return dqkv, None


def attention(qkv, H):
    """qkv [B,N,3*H*64] bf16 -> the fused HIP kernels (forward + backward); anything else is outside the envelope"""
    if qkv.dtype == torch.bfloat16:
        return FusedAttention.apply(qkv, H)
    return reference_op("attention", f"attention on {qkv.dtype} operands", qkv, H)


# --------------------------------------------------------------------------------------------
# fused projections / LayerNorm (inference-side: the teacher's six passes per step)
# --------------------------------------------------------------------------------------------
EPI_BIAS, EPI_GELU, EPI_RESIDUAL = 0, 1, 2


def gemm_bf16(x, w, b, epilogue=EPI_BIAS, residual=None, out=None):
    """x [M,K], w [N,K], b [N] (all bf16, or all fp16) -> [M,N] (same 16-bit type, or fp32 for the residual epilogue; may be in place)."""
    M, K = x.shape
    N = w.shape[0]
    dt = x.dtype
    assert w.dtype == dt and b.dtype == dt, "gemm: operands must share one 16-bit dtype"
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if epilogue == EPI_RESIDUAL else dt)
    if gemm_stamps is not None and M >= 4096:
        # the stamped span is the persistent kernel's; when the job count leaves a small remainder its tail runs as a second, tiny launch
        # (csrc/gemm_kernels.hip: launch_v6) whose share of the flops is not inside the span
        nt = ((M + 255) // 256) * (N // 256) if N % 256 == 0 else 0
        rem, rounds_up = nt % 256, (nt + 255) // 256
        frac = (nt - rem) / nt if (nt > 256 and 0 < rem <= 48 and (256 - rem) * 10 > 256 * rounds_up) else 1.0
        _C.fn16("cosa_gemm_set_stamp_slot", dt)(gemm_stamps.next_slot(2.0 * M * N * K * frac))
    with _C.profiled("gemm_bf16"):
        _C.check(_C.fn16("cosa_gemm_bf16", dt)(_C.ptr(x), _C.ptr(w), _C.ptr(b), _C.ptr(residual), _C.ptr(out), M, N, K, epilogue,
                                               _C.stream_ptr()), "cosa_gemm_bf16")
    _flops["gemm_bf16"] = _flops.get("gemm_bf16", 0) + 2.0 * M * N * K
    return out


def conv3x3_dilated_tokens(tok, w16, B, h, w, dilation, relu=True):
    """LargeFOV conv on NHWC tokens through the implicit-GEMM MFMA kernel (no MIOpen).  tok: [B, h*w, Cin] bf16, possibly a
    row-strided view (stride(1) == Cin-dim leading size, e.g. the encoder tokens without their cls row);
    w16: [Cout, Cin, 3, 3] bf16.  Returns [B*h*w, Cout] bf16."""
    Cout, Cin = w16.shape[0], w16.shape[1]
    assert tok.dtype in (torch.bfloat16, torch.float16) and tok.dtype == w16.dtype
    assert tok.stride(2) == 1 and tok.shape[1] == h * w and tok.shape[2] == Cin
    ldx = tok.stride(1)
    img_rows = tok.stride(0) // ldx if B > 1 else h * w
    assert B == 1 or tok.stride(0) == img_rows * ldx
    wt = w16.permute(2, 3, 0, 1).reshape(9, Cout, Cin).contiguous()
    y = torch.empty((B * h * w, Cout), device=tok.device, dtype=tok.dtype)
    with _C.profiled("conv3x3"):
        _C.check(_C.fn16("cosa_conv3x3_dilated_nhwc", tok.dtype)(_C.ptr(tok), _C.ptr(wt), _C.ptr(y), B, h, w, Cin, Cout, int(dilation), img_rows, 0,
                                                    ldx, int(relu), _C.stream_ptr()), "cosa_conv3x3_dilated_nhwc")
    _flops["conv3x3"] = _flops.get("conv3x3", 0) + 2.0 * B * h * w * Cout * Cin * 9
    return y


def _token_view_geometry(tok, B, h, w):
    ldx = tok.stride(1)
    img_rows = tok.stride(0) // ldx if B > 1 else h * w
    assert tok.dtype == torch.bfloat16 and tok.stride(2) == 1 and tok.shape[1] == h * w
    assert B == 1 or tok.stride(0) == img_rows * ldx
    return ldx, img_rows


def conv3x3_dilated_wgrad(dy, tok, B, h, w, dilation):
    """weight gradient of conv3x3_dilated_tokens: dy [B*h*w, Cout] bf16 (contiguous), tok as in the forward call ->
    dW [Cout, Cin, 3, 3] fp32 (a permuted view of the kernel's tap-major [Cout, 9*Cin] result).  Implicit im2col inside the TN MFMA
    kernel: no MIOpen, no materialised patches."""
    Cout, Cin = dy.shape[1], tok.shape[2]
    ldx, img_rows = _token_view_geometry(tok, B, h, w)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.shape[0] == B * h * w
    dw = _wgrad_alloc(Cout * 9 * Cin, dy.device)
    if dw is None:
        dw = torch.empty(Cout * 9 * Cin, device=dy.device, dtype=torch.float32)
    ws = _C.workspace(_C.lib().cosa_gemm_wgrad_workspace_bytes(B * h * w, Cout, 9 * Cin), dy.device, "wgrad")
    with _C.profiled("conv3x3_wgrad"):
        _C.check(_C.lib().cosa_conv3x3_dilated_wgrad(_C.ptr(dy), _C.ptr(tok), _C.ptr(dw), B, h, w, Cin, Cout, int(dilation), img_rows, 0,
                                                     ldx, 1, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_conv3x3_dilated_wgrad")
    _flops["conv3x3_wgrad"] = _flops.get("conv3x3_wgrad", 0) + 2.0 * B * h * w * Cout * Cin * 9
    return dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)


class DilatedConvReluFn(torch.autograd.Function):
    """relu(conv3x3(x, W, dilation=d, padding=d, bias=None)) on NHWC tokens, forward and backward on our MFMA kernels:
    forward = implicit-GEMM conv (+ReLU in the epilogue); dX = the same kernel on the gated output gradient with the taps
    flipped and (in, out) swapped; dW = the TN weight-gradient kernel over an implicit im2col (conv3x3_dilated_wgrad)."""

    @staticmethod
    def forward(ctx, tok, weight, B, h, w, dilation):
        w16 = cast_param(weight, torch.bfloat16)
        y = conv3x3_dilated_tokens(tok, w16, B, h, w, dilation, relu=True)
        ctx.save_for_backward(tok, w16, y)
        ctx.geom = (B, h, w, dilation)
        return y

    @staticmethod
    def backward(ctx, dy):
        tok, w16, y = ctx.saved_tensors
        B, h, w, dilation = ctx.geom
        dz = torch.where(y > 0, dy.to(torch.bfloat16), torch.zeros((), device=dy.device, dtype=torch.bfloat16)).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wd = w16.flip(2, 3).permute(1, 0, 2, 3)                         # [Cin, Cout, 3, 3]: taps mirrored, roles swapped
            dx = conv3x3_dilated_tokens(dz.view(B, h * w, -1), wd, B, h, w, dilation, relu=False).view(B, h * w, -1)
        if ctx.needs_input_grad[1]:
            dw = conv3x3_dilated_wgrad(dz, tok, B, h, w, dilation)
        return dx, dw, None, None, None, None


def head_linear(tok, weight, round_bf16=False):
    """tok [B, n, K] (fp32, bf16 or fp16; possibly the strided view without the cls row) x weight [N, K] (same dtype) -> [B*n, N] fp32 on the
    narrow-head kernel (exact-fp32 MFMA, fixed reduction order per row).  None when the shape is outside the kernel's envelope."""
    B, n, K = tok.shape
    N = weight.shape[0]
    al = 4 if tok.dtype == torch.float32 else 8
    if tok.dtype != weight.dtype or tok.dtype not in (torch.float32, torch.bfloat16, torch.float16) or K % 64 or tok.stride(2) != 1 \
            or not weight.is_contiguous() or tok.stride(1) % al or (B > 1 and tok.stride(0) % al) \
            or tok.data_ptr() % 16:
        return None
    y = torch.empty((B * n, N), device=tok.device, dtype=torch.float32)
    dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[tok.dtype]
    with _C.profiled("head_gemm"):
        _C.check(_C.lib().cosa_head_gemm(_C.ptr(tok), _C.ptr(weight), _C.ptr(y), B * n, N, K, n, tok.stride(0) if B > 1 else n * tok.stride(1),
                                         tok.stride(1), dt, int(round_bf16), N, 0, _C.stream_ptr()), "cosa_head_gemm")
    return y


class NarrowLinearFn(Function):
    """y[M,N] (fp32) = x[M,K] W[N,K]^T for the narrow heads of the TRAINING path (CAM / aux-CAM / classification heads: 1x1 convs with
    20 | 80 outputs, models/__init__.py:190-204; LargeFOV conv8, conv_head.py:38), forward and backward on the exact-fp32 MFMA narrow-head
    kernels (csrc/vit_kernels.hip): dX = dY W and dW = dY^T X with dY in fp32, slices of <= 32 weight rows for the wider (COCO) heads."""

    @staticmethod
    def forward(ctx, x2, weight, w16):
        y = head_linear(x2.unsqueeze(0), w16)
        ctx.save_for_backward(x2, w16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w16 = ctx.saved_tensors
        M, N = dy.shape
        K = x2.shape[1]
        dy = dy.float().contiguous()
        st = _C.stream_ptr()
        dx = dw = None
        with _C.profiled("head_gemm_bwd"):
            if ctx.needs_input_grad[0]:
                dx = torch.empty((M, K), device=dy.device, dtype=torch.bfloat16)
                _C.check(_C.lib().cosa_head_gemm_dgrad(_C.ptr(dy), _C.ptr(w16), _C.ptr(dx), M, N, K, st), "cosa_head_gemm_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty((N, K), device=dy.device, dtype=torch.float32)
                ws = torch.empty(_C.lib().cosa_head_gemm_wgrad_workspace(M, K), device=dy.device, dtype=torch.uint8)
                for c0 in range(0, N, 32):
                    nn_ = min(32, N - c0)
                    dys = dy if nn_ == N else dy[:, c0:c0 + nn_].contiguous()
                    _C.check(_C.lib().cosa_head_gemm_wgrad(_C.ptr(dys), _C.ptr(x2), _C.ptr(dw[c0:c0 + nn_]), _C.ptr(ws), M, nn_, K, st),
                             "cosa_head_gemm_wgrad")
        return dx, dw, None


def narrow_linear(x2, weight):
    """x2 [M,K] bf16 (contiguous) x fp32 master weight [N, K(,1,1)] -> [M,N] fp32 with autograd, every kernel an own one; None when the
    shape is outside the kernels' envelope (K % 128, N <= 128)."""
    w = weight.reshape(weight.shape[0], -1)
    N, K = w.shape
    if not x2.is_cuda or x2.dtype != torch.bfloat16 or K % 128 or not x2.is_contiguous():
        return None
    w16 = cast_param(weight, torch.bfloat16).reshape(N, K)
    if torch.is_grad_enabled() and (weight.requires_grad or x2.requires_grad):
        return NarrowLinearFn.apply(x2, w, w16.detach().contiguous())
    return head_linear(x2.unsqueeze(0), w16.contiguous())


def layernorm_f32(x, g, b, eps, want_bf16=True, want_f32=False):
    """x [rows,768] fp32, gamma / beta 16-bit (bf16 or fp16: that is also the 16-bit output type) -> (16-bit | None, fp32 | None)"""
    rows, D = x.shape
    y16 = torch.empty((rows, D), device=x.device, dtype=g.dtype) if want_bf16 else None
    y32 = torch.empty((rows, D), device=x.device, dtype=torch.float32) if want_f32 else None
    _C.check(_C.fn16("cosa_layernorm", g.dtype)(_C.ptr(x), _C.ptr(g), _C.ptr(b), _C.ptr(y16), _C.ptr(y32), rows, D, float(eps),
                                                _C.stream_ptr()), "cosa_layernorm")
    return y16, y32


# --------------------------------------------------------------------------------------------
# bf16x3 ("split") operands: the parity-grade precision of the no-grad passes (include/cosa_hip.h; csrc/split_kernels.hip)
#   a split row of logical width K: [hi (K) | lo (K) | aug (64)] bf16, row stride 2K + 64
# --------------------------------------------------------------------------------------------
_zero_bias = {}


def split_ld(K):
    return 2 * K + 64


def _x3_fn(name, dtype):
    """entry point of the three-term ("x3") path for halves of `dtype`: bf16 -> bf16x3, fp16 -> fp16x3 (round 6)"""
    if dtype == torch.bfloat16:
        return getattr(_C.lib(), name)
    if dtype == torch.float16:
        return getattr(_C.lib(), {"cosa_gemm_bf16x3": "cosa_gemm_f16x3", "cosa_attn_fwd_bf16x3": "cosa_attn_fwd_f16x3"}.get(name, name + "_f16"))
    raise _C.CosaError(f"{name}: split halves must be bfloat16 or float16, got {dtype}")


def split_rows(src, bias=None, ones=False, out=None, dtype=torch.bfloat16):
    """fp32 [R, K] (unit column stride; any row stride) -> split rows [R, 2K + 64] of `dtype` halves (out.dtype when `out` is given); aug
    block = (bias_hi, bias_lo, 0..) per row, (1, 1, 0..) with ones=True, zeros otherwise"""
    R, K = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1
    if out is None:
        out = torch.empty((R, split_ld(K)), device=src.device, dtype=dtype)
    _C.check(_x3_fn("cosa_split_rows", out.dtype)(_C.ptr(src), _C.ptr(bias), _C.ptr(out), R, K, src.stride(0), int(ones), _C.stream_ptr()),
             "cosa_split_rows")
    return out


def layernorm_split(x, g, b, eps, out=None, want_f32=False, dtype=torch.bfloat16):
    """LayerNorm(768) over the fp32 stream with fp32 gamma / beta -> (split rows [rows, 1600] | None, fp32 | None)"""
    rows, D = x.shape
    y32 = torch.empty((rows, D), device=x.device, dtype=torch.float32) if want_f32 else None
    _C.check(_x3_fn("cosa_layernorm_split", out.dtype if out is not None else dtype)(_C.ptr(x), _C.ptr(g), _C.ptr(b), _C.ptr(out), _C.ptr(y32), rows, D,
                                                                                     float(eps), _C.stream_ptr()), "cosa_layernorm_split")
    return out, y32


def gemm_x3(xs, ws, M, N, K, epilogue=EPI_BIAS, residual=None, out=None, ldy=None):
    """xs [M, 2K+64], ws [N, 2K+64] split rows of the same 16-bit type (bias inside ws) -> epilogue 0/1: that type [M, ldy >= 2N] = [hi | lo | ...];
    epilogue 2: fp32 [M, N] = residual + . (in place allowed)"""
    dev, dt = xs.device, xs.dtype
    assert ws.dtype == dt
    z = _zero_bias.get((dev, dt))
    if z is None:
        z = _zero_bias[(dev, dt)] = torch.zeros(8192, device=dev, dtype=dt)
    if epilogue == EPI_RESIDUAL:
        ldy = N
        if out is None:
            out = torch.empty((M, N), device=dev, dtype=torch.float32)
    else:
        ldy = ldy or 2 * N
        if out is None:
            out = torch.empty((M, ldy), device=dev, dtype=dt)
    if gemm_stamps is not None and M >= 4096:
        _C.fn16("cosa_gemm_set_stamp_slot", dt)(gemm_stamps.next_slot(2.0 * M * N * K, 2.0 * M * N * K * (3 * (K // 64) + 1) / (K // 64)))
    with _C.profiled("gemm_x3"):
        _C.check(_x3_fn("cosa_gemm_bf16x3", dt)(_C.ptr(xs), _C.ptr(ws), _C.ptr(z), _C.ptr(residual), _C.ptr(out), M, N, K, epilogue, ldy,
                                                _C.stream_ptr()), "cosa_gemm_x3")
    _flops["gemm_x3"] = _flops.get("gemm_x3", 0) + 6.0 * M * N * K
    return out


def attn_fwd_x3(qkv_s, B, N, H, out_s, lse=None):
    """attention on split qkv rows [B*N, ldq] -> split rows out_s [B*N, ldo] (hi | lo | aug) for the output projection"""
    assert out_s.dtype == qkv_s.dtype
    fl = 4.0 * B * H * N * N * 64          # algorithmic; issued: three MFMA terms per product
    st = stamps.next_slot(fl, 3.0 * fl) if stamps is not None else None
    _C.check(_x3_fn("cosa_attn_fwd_bf16x3", qkv_s.dtype)(_C.ptr(qkv_s), _C.ptr(out_s), _C.ptr(lse), B, N, H, 64, 0.125, qkv_s.stride(0),
                                                         out_s.stride(0), st, _C.stream_ptr()), "cosa_attn_fwd_x3")
    _flops["attn_x3"] = _flops.get("attn_x3", 0) + 12.0 * B * H * N * N * 64
    return out_s


# --------------------------------------------------------------------------------------------
# fp16c8 operands: parity-grade no-grad passes at 2x the 16-bit MFMA work (include/cosa_hip.h; csrc/c8.hpp)
#   a c8 row of logical width K, in bytes: [hi fp16 (2K) | lo8 e5m2 (K) | hi8 e5m2 (K) | aug fp16 (128)]; held as fp16 tensors of
#   2K + 64 columns (the same stride as a bf16x3 row)
# --------------------------------------------------------------------------------------------
_c8_zero_bias = {}


def c8_rows(src, bias=None, ones=False, out=None):
    """fp32 [R, K] (unit column stride; any row stride; K % 128 == 0) -> c8 rows [R, 2K + 64] (fp16 units); aug block = (bias_hi,
    bias_lo, 0..) per row, (1, 1, 0..) with ones=True, zeros otherwise"""
    R, K = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1
    if out is None:
        out = torch.empty((R, split_ld(K)), device=src.device, dtype=torch.float16)
    _C.check(_C.lib().cosa_c8_rows(_C.ptr(src), _C.ptr(bias), _C.ptr(out), R, K, src.stride(0), int(ones), _C.stream_ptr()), "cosa_c8_rows")
    return out


def layernorm_c8(x, g, b, eps, out=None, want_f32=False):
    """LayerNorm(768) over the fp32 stream with fp32 gamma / beta -> (c8 rows [rows, 1600 fp16 units] | None, fp32 | None)"""
    rows, D = x.shape
    y32 = torch.empty((rows, D), device=x.device, dtype=torch.float32) if want_f32 else None
    _C.check(_C.lib().cosa_layernorm_c8(_C.ptr(x), _C.ptr(g), _C.ptr(b), _C.ptr(out), _C.ptr(y32), rows, D, float(eps), _C.stream_ptr()),
             "cosa_layernorm_c8")
    return out, y32


def gemm_c8(xs, ws, M, N, K, epilogue=EPI_BIAS, residual=None, out=None, ldy=None):
    """xs [M, 2K+64], ws [N, 2K+64] c8 rows (bias inside ws).  epilogue 0: plain fp16 [M, ldy >= N]; 1 (GELU): c8 rows [M, 2N + 64]
    (hi | lo8 | hi8 written; the caller owns the aug block); 2: fp32 [M, N] = residual + . (in place allowed)"""
    dev = xs.device
    z = _c8_zero_bias.get(dev)
    if z is None:
        z = _c8_zero_bias[dev] = torch.zeros(8192, device=dev, dtype=torch.float16)
    if epilogue == EPI_RESIDUAL:
        ldy = N
        if out is None:
            out = torch.empty((M, N), device=dev, dtype=torch.float32)
    else:
        ldy = ldy or (split_ld(N) if epilogue == EPI_GELU else N)
        if out is None:
            out = torch.empty((M, ldy), device=dev, dtype=torch.float16)
    if gemm_stamps is not None and M >= 4096:
        _C.lib().cosa_gemm_set_stamp_slot_f16(gemm_stamps.next_slot(2.0 * M * N * K, 2.0 * M * N * K * (2 * (K // 64) + 1) / (K // 64)))
    with _C.profiled("gemm_c8"):
        _C.check(_C.lib().cosa_gemm_f16c8(_C.ptr(xs), _C.ptr(ws), _C.ptr(z), _C.ptr(residual), _C.ptr(out), M, N, K, epilogue, ldy,
                                          _C.stream_ptr()), "cosa_gemm_f16c8")
    _flops["gemm_c8"] = _flops.get("gemm_c8", 0) + 2.0 * M * N * K * (2 * (K // 64) + 1) / (K // 64)
    return out


# --------------------------------------------------------------------------------------------
# fp16c4 operands (round 4; csrc/c4.hpp): fp16 + FP4 (e2m1) correction terms in MX blocks -- ~1.58x the 16-bit MFMA work instead of fp16c8's ~2.08x
#   an operand = (rows [R, 2K + 64] fp16 units: hi | 16-byte blocks | unused | aug,  scales uint8 [c4_scale_bytes(R, K)])
# --------------------------------------------------------------------------------------------
def c4_scales(R, K, device):
    """zero-initialised scale tensor of a c4 operand with R rows of logical width K"""
    return torch.zeros(int(_C.lib().cosa_c4_scale_bytes(R, K)), device=device, dtype=torch.uint8)


def c4_rows(src, bias=None, ones=False, weight=False, out=None, scales=None):
    """fp32 [R, K] (K % 256 == 0) -> (c4 rows [R, 2K + 64] fp16 units, scales); weight: weight block order ([hi | lo']) and the 2^-11 of the
    correction terms in the scale bytes; aug block = (bias_hi, bias_lo, 0..) per row, (1, 1, 0..) with ones=True, zeros otherwise"""
    R, K = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1 and K % 256 == 0
    if out is None:
        out = torch.zeros((R, split_ld(K)), device=src.device, dtype=torch.float16)
    if scales is None:
        scales = c4_scales(R, K, src.device)
    _C.check(_C.lib().cosa_c4_rows(_C.ptr(src), _C.ptr(bias), _C.ptr(out), _C.ptr(scales), R, K, src.stride(0), int(ones), int(weight),
                                   _C.stream_ptr()), "cosa_c4_rows")
    return out, scales


def layernorm_c4(x, g, b, eps, out=None, scales=None, want_f32=False):
    """LayerNorm(768) over the fp32 stream with fp32 gamma / beta -> (c4 rows | None, fp32 | None); scales written beside the rows"""
    rows, D = x.shape
    y32 = torch.empty((rows, D), device=x.device, dtype=torch.float32) if want_f32 else None
    _C.check(_C.lib().cosa_layernorm_c4(_C.ptr(x), _C.ptr(g), _C.ptr(b), _C.ptr(out), _C.ptr(scales), _C.ptr(y32), rows, D, float(eps),
                                        _C.stream_ptr()), "cosa_layernorm_c4")
    return out, y32


def gemm_c4(xs, xsc, ws, wsc, M, N, K, epilogue=EPI_BIAS, residual=None, out=None, out_scales=None, ldy=None):
    """xs [M, 2K+64] / ws [N, 2K+64] c4 rows with their scale tensors (bias inside ws).  epilogue 0: plain fp16 [M, ldy >= N]; 1 (GELU):
    c4 rows [M, 2N + 64] + out_scales (c4_scales(M, N)); 2: fp32 [M, N] = residual + . (in place allowed)"""
    dev = xs.device
    z = _c8_zero_bias.get(dev)
    if z is None:
        z = _c8_zero_bias[dev] = torch.zeros(8192, device=dev, dtype=torch.float16)
    if epilogue == EPI_RESIDUAL:
        ldy = N
        if out is None:
            out = torch.empty((M, N), device=dev, dtype=torch.float32)
    else:
        ldy = ldy or (split_ld(N) if epilogue == EPI_GELU else N)
        if out is None:
            out = torch.zeros((M, ldy), device=dev, dtype=torch.float16)
        if epilogue == EPI_GELU and out_scales is None:
            out_scales = c4_scales(M, N, dev)
    if gemm_stamps is not None and M >= 4096:
        _C.lib().cosa_gemm_set_stamp_slot_f16(gemm_stamps.next_slot(2.0 * M * N * K, 2.0 * M * N * K * (K // 64 + 1 + K // 128) / (K // 64)))
    with _C.profiled("gemm_c4"):
        _C.check(_C.lib().cosa_gemm_f16c4(_C.ptr(xs), _C.ptr(xsc), _C.ptr(ws), _C.ptr(wsc), _C.ptr(z), _C.ptr(residual), _C.ptr(out),
                                          _C.ptr(out_scales), M, N, K, epilogue, ldy, _C.stream_ptr()), "cosa_gemm_f16c4")
    _flops["gemm_c4"] = _flops.get("gemm_c4", 0) + 2.0 * M * N * K * (K // 64 + 1 + K // 128) / (K // 64)
    return (out, out_scales) if epilogue == EPI_GELU else out


def attn_fwd_c4(qkv, B, N, H, out_c4, out_scales, row0, lse=None, q_prescaled=False):
    """attention on plain fp16 qkv [B, N, 3 H 64] -> c4 rows out_c4 [B*N, 2 H 64 + 64 fp16 units] (a row slice of an operand) + the scale bytes of
    those rows in out_scales, the scale tensor of the whole operand; row0 = index of out_c4's first row in that operand"""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and out_c4.stride(0) == split_ld(H * 64)
    fl = 4.0 * B * H * N * N * 64
    st = stamps.next_slot(fl) if stamps is not None else None
    with _C.profiled("attn_fwd"):
        _C.check(_C.lib().cosa_attn_fwd_f16c4(_C.ptr(qkv), _C.ptr(out_c4), _C.ptr(out_scales), int(row0), _C.ptr(lse), B, N, H, 64, LN2 if q_prescaled else 0.125, st,
                                              _C.stream_ptr()), "cosa_attn_fwd_f16c4")
    _flops["attn_fwd"] = _flops.get("attn_fwd", 0) + fl
    return out_c4


LN2 = 0.6931471805599453          # attention scale to pass when q arrives pre-multiplied by 64^-0.5 log2(e) (weights folded: models/vit.py)


def attn_fwd_c8(qkv, B, N, H, out_c8, lse=None, q_prescaled=False):
    """attention on plain fp16 qkv [B, N, 3 H 64] -> c8 rows out_c8 [B*N, 2 H 64 + 64 fp16 units] (hi | lo8 | hi8 | aug)"""
    assert qkv.dtype == torch.float16 and qkv.is_contiguous() and out_c8.stride(0) == split_ld(H * 64)
    fl = 4.0 * B * H * N * N * 64
    st = stamps.next_slot(fl) if stamps is not None else None
    with _C.profiled("attn_fwd"):
        _C.check(_C.lib().cosa_attn_fwd_f16c8(_C.ptr(qkv), _C.ptr(out_c8), _C.ptr(lse), B, N, H, 64, LN2 if q_prescaled else 0.125, st,
                                              _C.stream_ptr()), "cosa_attn_fwd_f16c8")
    _flops["attn_fwd"] = _flops.get("attn_fwd", 0) + fl
    return out_c8


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
_shadows = {}          # id(param) -> (param, persistent 16-bit shadow at a fixed address; refreshed explicitly, hipGraph-safe)


class ShadowSet:
    """Persistent 16-bit copies of a module's fp32 parameters, refreshed with ONE multi-tensor copy.

    The teacher is read 6x per step and written once (EMA).  Keeping fixed-address shadows (a) casts every weight
    exactly once per step and (b) makes the whole teacher pass capturable in a hipGraph: the refresh is the first node
    of the graph, the kernels after it read fixed addresses."""

    def __init__(self, module, dtype=torch.bfloat16):
        self.params = [p for p in module.parameters()]
        self.dtype = dtype
        self.shadows = [torch.empty_like(p, dtype=dtype) for p in self.params]
        for p, s in zip(self.params, self.shadows):
            _shadows[id(p)] = (p, s)
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        torch._foreach_copy_(self.shadows, self.params)


def ensure_shadows(module, dtype=torch.bfloat16):
    """Shadows of ALL parameters of `module` (a VITNetwork, possibly DDP-wrapped), created on first use, and -- unless an optimizer
    kernel keeps them current (`module._cosa_shadow_auto = False`, set by CoSATrainer with the fused AdamW+EMA step) -- refreshed
    here with one multi-tensor copy.  Called at the start of every no-grad entry point (multi_scale_camseg*, evaluate,
    VITNetwork.forward under no_grad), so weights written by ANY route (optimizer.step, load_state_dict, the reference loop's
    `param.data.mul_().add_()` EMA, which does not bump `_version`) are seen by the next forward: nothing is cached by version."""
    module = getattr(module, "module", module)
    ss = module.__dict__.get("_cosa_shadowset")
    if ss is not None and shadows_fresh.depth > 0:
        return ss
    if ss is None or ss.dtype != dtype or any(a is not b for a, b in zip(ss.params, module.parameters())) \
            or (ss.params and ss.params[0].device != ss.shadows[0].device):
        if ss is not None and not module.__dict__.get("_cosa_shadow_auto", True):
            # a trainer has baked the addresses / dtype of the existing shadows into its fused optimizer records and its captured teacher
            # graph (CoSATrainer): replacing the set would leave both writing to and reading from freed memory, and the new shadows would
            # never be refreshed
            raise RuntimeError("ensure_shadows: this module's 16-bit shadows are owned by a CoSATrainer (fused AdamW + EMA step, captured "
                               "teacher graph); changing its compute dtype / precision or its parameters after the trainer was built is "
                               "not supported -- build a new trainer, or evaluate a copy of the network")
        ss = ShadowSet(module, dtype)
        module.__dict__["_cosa_shadowset"] = ss
        return ss
    if module.__dict__.get("_cosa_shadow_auto", True):
        ss.refresh()
    return ss


class shadows_fresh:
    """`with shadows_fresh():` -- nested no-grad entry points skip their own refresh (one refresh per multi-scale pass, not per scale)"""
    depth = 0

    def __enter__(self):
        shadows_fresh.depth += 1

    def __exit__(self, *exc):
        shadows_fresh.depth -= 1


def shadow_of(p):
    ent = _shadows.get(id(p))
    return ent[1] if ent is not None and ent[0] is p else None


def cast_param(p, dtype):
    """16-bit view of an fp32 master parameter.  Registered shadows are returned as they are (no-grad paths); with grad the
    cast is differentiable (student); a parameter without a shadow is cast on every call -- no version-keyed cache (in-place
    writes through `.data` or raw pointers do not bump `_version`)."""
    if p.dtype == dtype:
        return p
    if torch.is_grad_enabled() and p.requires_grad:
        return p.to(dtype)
    ent = _shadows.get(id(p))
    if ent is not None and ent[0] is p and ent[1].dtype == dtype and ent[1].device == p.device:
        return ent[1]
    return p.detach().to(dtype)


_MM_OUT_DTYPE = None      # does torch.mm(bf16, bf16, out_dtype=fp32) work on this build?  probed on first use


def _mm_f32(a, b):
    """a @ b for bf16 operands with an fp32 result (weight gradients land in the fp32 master .grad without a cast pass)."""
    global _MM_OUT_DTYPE
    if _MM_OUT_DTYPE is None:
        try:
            torch.mm(a[:8, :8].contiguous(), b[:8, :8].contiguous(), out_dtype=torch.float32)
            _MM_OUT_DTYPE = True
        except Exception:
            _MM_OUT_DTYPE = False
    if _MM_OUT_DTYPE:
        return torch.mm(a, b, out_dtype=torch.float32)
    return torch.mm(a, b).float()


class _WgradArena:
    """Home of one step's weight gradients: the gradients of a step are carved out of ONE buffer in call order (stable addresses from
    step to step, no allocator traffic in the backward pass).  Since round 3 the weight-gradient kernels OVERWRITE their output (partial
    slabs + a fixed-order reduction instead of fp32 atomics), so the arena is no longer cleared per step.
    Off unless a trainer turns it on: with it on, gradients of step t are invalid once step t+1 begins."""
    buf, off, high, enabled = None, 0, 0, False


def wgrad_arena_begin(device):
    a = _WgradArena
    a.enabled = True
    a.off = 0


def _wgrad_alloc(n, device):
    a = _WgradArena
    n_al = (n + 63) // 64 * 64
    if not a.enabled:
        return None
    if a.buf is None or a.buf.device != device or a.off + n_al > a.buf.numel():
        if a.buf is None or a.buf.device != device:
            a.buf, a.off, a.high = torch.zeros(128 << 20, device=device, dtype=torch.float32), 0, 0       # 512 MB: 92.5 M params + slack
        if a.off + n_al > a.buf.numel():
            return None                                      # arena exhausted: this tensor falls back to its own memset
    out = a.buf[a.off: a.off + n]
    a.off += n_al
    a.high = max(a.high, a.off)
    return out


def gemm_wgrad(dy2, x2, want_bias=False):
    """dW[N,K] fp32 = dy2[M,N]^T x2[M,K] (bf16) and, optionally, db[N] = column sums of dy2: the TN MFMA kernel with
    transposing LDS reads; token range split over workgroups, partial slabs reduced in a fixed order (deterministic, no atomics)."""
    M, N = dy2.shape
    K = x2.shape[1]
    dw = _wgrad_alloc(N * K, dy2.device)
    db = _wgrad_alloc(N, dy2.device) if (want_bias and dw is not None) else None
    if dw is None or (want_bias and db is None):
        dw = torch.empty((N, K), device=dy2.device, dtype=torch.float32)
        db = torch.empty((N,), device=dy2.device, dtype=torch.float32) if want_bias else None
    else:
        dw = dw.view(N, K)
    ws = _C.workspace(_C.lib().cosa_gemm_wgrad_workspace_bytes(M, N, K), dy2.device, "wgrad")
    with _C.profiled("gemm_wgrad"):
        _C.check(_C.lib().cosa_gemm_wgrad_bf16(_C.ptr(dy2), _C.ptr(x2), _C.ptr(dw), _C.ptr(db), M, N, K, 1, _C.ptr(ws), ws.numel(),
                                               _C.stream_ptr()), "cosa_gemm_wgrad_bf16")
    return (dw, db) if want_bias else dw


class _WgradItem(ctypes.Structure):
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("N", ctypes.c_int),
                ("K", ctypes.c_int)]


def gemm_wgrad_batched(pairs):
    """[(dy2 [M,N] bf16, x2 [M,K] bf16, want_bias)] -> [(dW [N,K] fp32, db [N] fp32 or None)]: every weight gradient in ONE persistent launch
    (cosa_gemm_wgrad_batched): with hundreds of tiles no item needs split-K, every tile runs the whole token loop and writes dW once."""
    outs, by_m = [None] * len(pairs), {}
    for i, (dy2, x2, want_bias) in enumerate(pairs):
        assert dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and dy2.is_contiguous() and x2.is_contiguous()
        M, N = dy2.shape
        K = x2.shape[1]
        dw = _wgrad_alloc(N * K, dy2.device)
        dw = dw.view(N, K) if dw is not None else torch.empty((N, K), device=dy2.device, dtype=torch.float32)
        db = None
        if want_bias:
            db = _wgrad_alloc(N, dy2.device)
            db = db if db is not None else torch.empty((N,), device=dy2.device, dtype=torch.float32)
        outs[i] = (dw, db)
        by_m.setdefault(M, []).append(i)
    for M, idx in by_m.items():
        arr = (_WgradItem * len(idx))()
        for j, i in enumerate(idx):
            dy2, x2, _ = pairs[i]
            dw, db = outs[i]
            arr[j] = _WgradItem(dy2.data_ptr(), x2.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, dy2.shape[1], x2.shape[1])
        with _C.profiled("gemm_wgrad"):
            _C.check(_C.lib().cosa_gemm_wgrad_batched(ctypes.cast(arr, ctypes.c_void_p), len(idx), M, _C.stream_ptr()), "cosa_gemm_wgrad_batched")
    return outs


class WgradCollector:
    """The (dY, X) pairs of the linears whose weight gradients are computed together (DeferredWgrad): LinearShadowFn.backward adds its pair
    instead of launching its own weight-gradient GEMM when its master weight is one of `params`."""

    def __init__(self, params):
        self.keys = {id(p) for p in params}
        self.pending = {}
        self.used = set()            # weights whose linear ran in the forward pass (each owes a pair)

    def mark_used(self, w):
        self.used.add(id(w))

    def add(self, w, dy2, x2):
        if id(w) in self.pending:
            raise RuntimeError("WgradCollector: a weight was used twice inside one deferred region (not supported)")
        self.pending[id(w)] = (dy2, x2)


_wgrad_collector = None


class collecting:
    """with collecting(c): linears run inside add their weight-gradient work to c"""

    def __init__(self, c):
        self.c = c

    def __enter__(self):
        global _wgrad_collector
        self.prev, _wgrad_collector = _wgrad_collector, self.c
        return self.c

    def __exit__(self, *exc):
        global _wgrad_collector
        _wgrad_collector = self.prev
        return False


event_log = None          # tests / tools/ddp_check.py: a list that receives ("defer", n_linears, t) when a DeferredWgrad node runs its backward


class DeferredWgrad(Function):
    """Identity on x, placed at the INPUT of a run of layers: its backward executes after theirs (autograd reaches the input last), launches
    the collected weight gradients in one batch and hands them to the masters (`params` = w0, b0, w1, b1, ...: AccumulateGrad -- and with it
    DistributedDataParallel's bucket hooks -- see them exactly as if each linear had returned its own)."""

    @staticmethod
    def forward(ctx, x, collector, *params):
        ctx.collector = collector
        ctx.ids = [id(p) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, gx):
        if event_log is not None:
            event_log.append(("defer", len(ctx.ids) // 2, time.perf_counter()))
        c = ctx.collector
        jobs, slots = [], []
        for j in range(0, len(ctx.ids), 2):                  # (weight, bias) pairs
            ent = c.pending.pop(ctx.ids[j], None)
            if ent is not None:
                jobs.append((ent[0], ent[1], True))
                slots.append(j)
            elif ctx.needs_input_grad[2 + j] and ctx.ids[j] in c.used:
                # the linear ran in the forward pass but its backward has not happened yet: this node sits on a tensor whose gradient is
                # complete BEFORE the group's last backward -- its weight gradient would be lost without a word
                raise RuntimeError("DeferredWgrad: reached before the backward of a linear it collects for (place it on the tensor whose "
                                   "gradient the group's backward produces last)")
        grads = [None] * len(ctx.ids)
        if jobs:
            for j, (dw, db) in zip(slots, gemm_wgrad_batched(jobs)):
                grads[j], grads[j + 1] = dw, db
        if c.pending:
            raise RuntimeError("DeferredWgrad: collected a weight that is not among its parameters")
        return (gx, None) + tuple(grads)


def defer_wgrads(x, linears):
    """x -> (x', collector): route the weight / bias gradients of the nn.Linear modules `linears` (those that run on LinearShadowFn) through
    one batched launch that executes when the backward pass reaches x.  Use: x, c = defer_wgrads(x, mods); with collecting(c): ..."""
    params = []
    for m in linears:
        params += [m.weight, m.bias]
    c = WgradCollector(params[0::2])
    return DeferredWgrad.apply(x, c, *params), c


class TransposedShadows:
    """bf16 W^T copies of the student's projection weights, rebuilt from the fp32 masters by ONE batched launch per step (after the
    optimizer).  With them the input-gradient GEMM dX = dY W of an nn.Linear is the forward GEMM kernel applied to W^T: no NN kernel
    family, no library call."""

    def __init__(self, weights):
        import numpy as np
        self.weights = list(weights)
        dev = self.weights[0].device
        self.t16 = [torch.empty((w.numel() // w.shape[0], w.shape[0]), device=dev, dtype=torch.bfloat16) for w in self.weights]
        rec_dt = np.dtype([("src", "u8"), ("dst", "u8"), ("rows", "i4"), ("cols", "i4"), ("tile0", "i4"), ("tiles_c", "i4")])
        assert rec_dt.itemsize == _C.lib().cosa_transpose_record_bytes()
        rec = np.zeros(len(self.weights), rec_dt)
        tiles = 0
        for i, (w, t) in enumerate(zip(self.weights, self.t16)):
            assert w.dtype == torch.float32 and w.is_contiguous()
            rows, cols = w.shape[0], w.numel() // w.shape[0]                 # (a conv weight counts as its [out, -1] view)
            tr, tc = (rows + 63) // 64, (cols + 63) // 64
            rec[i] = (w.data_ptr(), t.data_ptr(), rows, cols, tiles, tc)
            tiles += tr * tc
            _transposed[id(w)] = (w, t)
        self.total_tiles = tiles
        self.d_rec = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self.refresh()

    def refresh(self):
        _C.check(_C.lib().cosa_transpose_cast_batched(_C.ptr(self.d_rec), len(self.weights), self.total_tiles, _C.stream_ptr()),
                 "cosa_transpose_cast_batched")


_transposed = {}       # id(weight) -> (weight, bf16 W^T)
_zeros16 = {}


def _zero_bias16(n, dev):
    z = _zeros16.get(dev)
    if z is None or z.numel() < n:
        z = _zeros16[dev] = torch.zeros(max(n, 8192), device=dev, dtype=torch.bfloat16)
    return z


def _own_gemm_ok(M, N, K):
    return N % 128 == 0 and K % 64 == 0


class LinearShadowFn(Function):
    """y = x W^T + b (act = 1: gelu(.) of it, mlp.fc1) with the bf16 SHADOW of the fp32 master weight, forward and backward on the
    MFMA GEMM kernels of this repository: forward = cosa_gemm_bf16 (fc1: the dual epilogue that also keeps the pre-activation),
    dX = the same kernel on the transposed shadow W^T (TransposedShadows), dW / db = the TN weight-gradient kernel in fp32.
    Gradients are routed to the masters (w, b)."""

    @staticmethod
    def forward(ctx, x, w, b, w16, b16, wT16, act):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, N = x2.shape[0], w16.shape[0]
        h = None
        if act:
            h = torch.empty((M, N), device=x.device, dtype=torch.bfloat16)
            y = torch.empty_like(h)
            with _C.profiled("gemm_bf16"):
                _C.check(_C.lib().cosa_gemm_bf16_dual_gelu(_C.ptr(x2), _C.ptr(w16), _C.ptr(b16), _C.ptr(h), _C.ptr(y), M, N, K, _C.stream_ptr()),
                         "cosa_gemm_bf16_dual_gelu")
            _flops["gemm_bf16"] = _flops.get("gemm_bf16", 0) + 2.0 * M * N * K
        else:
            y = gemm_bf16(x2, w16, b16, EPI_BIAS)
        ctx.save_for_backward(x2, wT16, h)
        ctx.xshape = x.shape
        c = _wgrad_collector
        ctx.collect = (c, w) if (c is not None and id(w) in c.keys) else None
        if ctx.collect is not None:
            c.mark_used(w)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wT16, h = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if h is not None:                                   # through the GELU: dH = dA * gelu'(H)
            dh = torch.empty_like(h)
            _C.check(_C.lib().cosa_gelu_backward(_C.ptr(dy2), _C.ptr(h), _C.ptr(dh), dh.numel(), _C.stream_ptr()), "cosa_gelu_backward")
            dy2 = dh
        M, N = dy2.shape
        K = x2.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_bf16(dy2, wT16, _zero_bias16(K, dy2.device)[:K], EPI_BIAS).view(ctx.xshape)      # [M,N] x (W^T [K,N])^T
        # (running the weight-gradient GEMM on a side stream next to the input-gradient GEMM, which often leaves CUs idle, was measured
        #  on one box in both issue orders: 324.4 / 322.3 img/s against 323.0 -- no gain, not kept)
        if ctx.needs_input_grad[1]:
            if ctx.collect is not None:
                ctx.collect[0].add(ctx.collect[1], dy2, x2)                 # computed with the others when the backward reaches DeferredWgrad
            else:
                dw, db = gemm_wgrad(dy2, x2, want_bias=True)
        return dx, dw, db, None, None, None, None


def linear_view2d(x, weight, bias, dtype):
    """nn.Linear with a weight parameter of more than two dimensions used as its [out, -1] view (the patch projection's conv weight):
    LinearShadowFn; the parameter needs registered shadows (bf16 copy and bf16 transposed copy of the 2-D view)"""
    ew, eb, et = _shadows.get(id(weight)), _shadows.get(id(bias)), _transposed.get(id(weight))
    w2 = weight.view(weight.shape[0], -1)
    if ew is not None and eb is not None and et is not None and ew[0] is weight and eb[0] is bias and et[0] is weight \
            and w2.shape[0] % 128 == 0 and w2.shape[1] % 128 == 0:
        return LinearShadowFn.apply(x, w2, bias, ew[1].view(w2.shape), eb[1], et[1], False)
    return reference_op("linear", f"linear_view2d (weight {tuple(w2.shape)}, {dtype}; shadows registered: {ew is not None and et is not None})",
                        x, cast_param(weight, dtype).view(w2.shape), cast_param(bias, dtype))


def linear(x, weight, bias, dtype, act=False):
    """nn.Linear (+ GELU with act=True) on `dtype` operands from fp32 masters.  Training on the GPU with registered shadows (bf16 W, b and
    W^T): LinearShadowFn, every GEMM an own kernel; anything else (fp32 parity mode, odd shapes) is outside the envelope."""
    if torch.is_grad_enabled() and weight.requires_grad and dtype == torch.bfloat16 and x.is_cuda:
        ew, eb, et = _shadows.get(id(weight)), _shadows.get(id(bias)), _transposed.get(id(weight))
        if ew is not None and eb is not None and et is not None and ew[0] is weight and eb[0] is bias and et[0] is weight \
                and _own_gemm_ok(x.numel() // x.shape[-1], weight.shape[0], weight.shape[1]) and weight.shape[0] % 64 == 0 \
                and weight.shape[0] % 128 == 0 and weight.shape[1] % 128 == 0:
            return LinearShadowFn.apply(x, weight, bias, ew[1], eb[1], et[1], bool(act))
    return reference_op("linear", f"linear (weight {tuple(weight.shape)}, {dtype}, grad {torch.is_grad_enabled()})",
                        x, cast_param(weight, dtype), cast_param(bias, dtype), act=act)


# --------------------------------------------------------------------------------------------
# student blocks: residual add + LayerNorm in one kernel, with its backward  (models/vit/vit.py:154-158)
# --------------------------------------------------------------------------------------------
def _ln_backward(dy, x_new, mean, rstd, w16, dskip):
    dy = dy.contiguous()
    rows = dy.numel() // 768
    dx = torch.empty_like(dy)
    dgamma = torch.empty(768, device=dy.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    L = _C.lib()
    ws = _C.workspace(L.cosa_layernorm_bwd_workspace_bytes(rows, 768), dy.device, "ln_bwd")
    _C.check(L.cosa_layernorm_bwd(_C.ptr(dy), _C.ptr(x_new), _C.ptr(mean), _C.ptr(rstd), _C.ptr(w16),
                                  _C.ptr(dskip.contiguous() if dskip is not None else None), _C.ptr(dx), _C.ptr(dgamma), _C.ptr(dbeta),
                                  0, rows, 768, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_layernorm_bwd")
    return dx, dgamma, dbeta


def _ln_forward(x, delta, w16, b16, eps):
    rows = x.numel() // 768
    x_new = torch.empty_like(x) if delta is not None else x
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    _C.check(_C.lib().cosa_add_layernorm_fwd(_C.ptr(x), _C.ptr(delta), _C.ptr(w16), _C.ptr(b16), _C.ptr(x_new if delta is not None else None),
                                             _C.ptr(y), _C.ptr(mean), _C.ptr(rstd), rows, 768, float(eps), _C.stream_ptr()),
             "cosa_add_layernorm_fwd")
    return x_new, y, mean, rstd


class AddLayerNormFn(Function):
    """(x, delta) -> (x_new = x + delta, y = LayerNorm(x_new)) on the bf16 residual stream; gradients of gamma / beta go to the fp32
    masters.  The backward folds the gradient arriving at x_new through the skip connection into the same pass."""

    @staticmethod
    def forward(ctx, x, delta, w, b, w16, b16, eps):
        x_new, y, mean, rstd = _ln_forward(x.contiguous(), delta.contiguous(), w16, b16, eps)
        ctx.save_for_backward(x_new, mean, rstd, w16)
        return x_new, y

    @staticmethod
    def backward(ctx, dx_new, dy):
        x_new, mean, rstd, w16 = ctx.saved_tensors
        if dy is None:                                  # y unused: only the skip gradient flows
            return dx_new, dx_new, None, None, None, None, None
        dx, dgamma, dbeta = _ln_backward(dy, x_new, mean, rstd, w16, dx_new)
        return dx, dx, dgamma, dbeta, None, None, None


class LayerNormFn(Function):
    """y = LayerNorm(x) (no residual add in front: the first block's norm1); backward through the same kernel, no skip term."""

    @staticmethod
    def forward(ctx, x, w, b, w16, b16, eps):
        x = x.contiguous()
        _, y, mean, rstd = _ln_forward(x, None, w16, b16, eps)
        ctx.save_for_backward(x, mean, rstd, w16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w16 = ctx.saved_tensors
        dx, dgamma, dbeta = _ln_backward(dy, x, mean, rstd, w16, None)
        return dx, dgamma, dbeta, None, None, None


def add_layernorm(x, delta, weight, bias, eps):
    """-> (x + delta, LayerNorm(x + delta)); delta may be None.  bf16 [.., 768] on the GPU only."""
    ew, eb = _shadows.get(id(weight)), _shadows.get(id(bias))
    if ew is not None and eb is not None and ew[0] is weight and eb[0] is bias:
        w16, b16 = ew[1], eb[1]
    else:
        w16, b16 = weight.detach().to(torch.bfloat16), bias.detach().to(torch.bfloat16)
    if delta is None:
        return x, LayerNormFn.apply(x, weight, bias, w16, b16, eps)
    return AddLayerNormFn.apply(x, delta, weight, bias, w16, b16, eps)


# --------------------------------------------------------------------------------------------
# student blocks on an fp32 residual stream (round 4; the reference trains in fp32, main.py:124-246):
#   x' = x + Linear(a)   -> the projection GEMM's fp32 residual epilogue (cosa_gemm_bf16, epilogue 2): the sum is formed in fp32 from the
#                           fp32 accumulators, never rounded to 16 bits
#   y  = LayerNorm(x')   -> cosa_layernorm (fp32 row in, bf16 row out: the next GEMM's operand)
# and ONE autograd node per (projection, following LayerNorm) pair, so that its backward sees both gradients of x' -- through the
# LayerNorm and through the skip connection -- and adds them in fp32 inside the LayerNorm-backward kernel (cosa_layernorm_bwd_f32), which
# also emits the bf16 copy that the projection's input- / weight-gradient GEMMs take as dY.
# --------------------------------------------------------------------------------------------
def _gamma16(weight, bias):
    ew, eb = _shadows.get(id(weight)), _shadows.get(id(bias))
    if ew is not None and eb is not None and ew[0] is weight and eb[0] is bias and ew[1].dtype == torch.bfloat16:
        return ew[1], eb[1]
    return weight.detach().to(torch.bfloat16), bias.detach().to(torch.bfloat16)


def _ln_backward_f32(dy, x, g16, eps, dskip, want16):
    """-> (dx fp32, dx16 bf16 | None, dgamma, dbeta) for y = LayerNorm(x) with x fp32 [rows, 768], dy bf16, dskip fp32 | None"""
    rows = x.numel() // 768
    dy = dy.contiguous()
    dx = torch.empty((rows, 768), device=x.device, dtype=torch.float32)
    dx16 = torch.empty((rows, 768), device=x.device, dtype=torch.bfloat16) if want16 else None
    dgamma = torch.empty(768, device=x.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    L = _C.lib()
    ws = _C.workspace(L.cosa_layernorm_bwd_workspace_bytes(rows, 768), x.device, "ln_bwd")
    assert dy.dtype in (torch.bfloat16, torch.float32)
    _C.check(L.cosa_layernorm_bwd_f32(_C.ptr(dy), int(dy.dtype == torch.float32), _C.ptr(x), _C.ptr(g16),
                                      _C.ptr(dskip.contiguous() if dskip is not None else None),
                                      _C.ptr(dx), _C.ptr(dx16), _C.ptr(dgamma), _C.ptr(dbeta), 0, rows, 768, float(eps), _C.ptr(ws),
                                      ws.numel(), _C.stream_ptr()), "cosa_layernorm_bwd_f32")
    return dx, dx16, dgamma, dbeta


class StreamLayerNormFn(Function):
    """x (fp32 stream) -> (x itself, y = LayerNorm(x) in bf16).  Returning the stream from the same node lets the backward fold the
    gradient that reaches x through the skip connection into the LayerNorm-backward pass (no separate fp32 add)."""

    @staticmethod
    def forward(ctx, x, w, b, w16, b16, eps):
        x = x.contiguous()
        y, _ = layernorm_f32(x.view(-1, 768), w16, b16, eps)
        ctx.save_for_backward(x, w16)
        ctx.eps = eps
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, dx_skip, dy):
        x, w16 = ctx.saved_tensors
        if dy is None:
            return dx_skip, None, None, None, None, None
        dx, _, dgamma, dbeta = _ln_backward_f32(dy, x, w16, ctx.eps, dx_skip, False)
        return dx.view(x.shape), dgamma, dbeta, None, None, None


def stream_layernorm(x, weight, bias, eps):
    """fp32 stream x [.., 768] -> (x, LayerNorm(x) bf16)"""
    w16, b16 = _gamma16(weight, bias)
    return StreamLayerNormFn.apply(x, weight, bias, w16, b16, eps)


class ResidualLinearLNFn(Function):
    """(a bf16 [.., K], x fp32 [.., 768]) -> (x' = x + a W^T + b  (fp32), y = LayerNorm(x'; gamma, beta) (bf16)).
    models/vit/vit.py:154-158: `x = x + attn(...)` / `x = x + mlp(...)` followed by the next norm.  Forward: the persistent GEMM with the
    fp32 residual epilogue + the fp32 -> bf16 LayerNorm kernel; backward: LayerNorm' + skip gradient in fp32 (one kernel, which also writes
    the bf16 dY), dA = dY W on the transposed shadow, dW / db on the TN kernel (or handed to the group's DeferredWgrad)."""

    @staticmethod
    def forward(ctx, a, x, w, b, w16, b16, wT16, gw, gb, gw16, gb16, eps, y_f32=False):
        K = a.shape[-1]
        a2 = a.reshape(-1, K)
        if not a2.is_contiguous():
            a2 = a2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xn = gemm_bf16(a2, w16, b16, EPI_RESIDUAL, residual=x2)                  # fp32 [M, N], a fresh buffer: x stays what earlier nodes saved
        # y_f32: the final norm -- its output feeds the decoder and the heads, whose gradients must meet in fp32 (fanout_bf16)
        y16, y32 = layernorm_f32(xn, gw16, gb16, eps, want_bf16=not y_f32, want_f32=y_f32)
        y = y32 if y_f32 else y16
        ctx.save_for_backward(a2, wT16, xn, gw16)
        ctx.eps = eps
        ctx.ashape, ctx.xshape = a.shape, x.shape
        c = _wgrad_collector
        ctx.collect = (c, w) if (c is not None and id(w) in c.keys) else None
        if ctx.collect is not None:
            c.mark_used(w)
        return xn.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, g_x, g_y):
        a2, wT16, xn, gw16 = ctx.saved_tensors
        dgamma = dbeta = None
        if g_y is not None:
            dxt, dxt16, dgamma, dbeta = _ln_backward_f32(g_y, xn, gw16, ctx.eps, g_x, True)
        else:                                   # the LayerNorm output was not used: only the skip gradient flows
            dxt = g_x.reshape(-1, g_x.shape[-1]).contiguous()
            dxt16 = dxt.to(torch.bfloat16)
        K = a2.shape[1]
        da = dw = db = None
        if ctx.needs_input_grad[0]:
            da = gemm_bf16(dxt16, wT16, _zero_bias16(K, dxt16.device)[:K], EPI_BIAS).view(ctx.ashape)
        if ctx.needs_input_grad[2]:
            if ctx.collect is not None:
                ctx.collect[0].add(ctx.collect[1], dxt16, a2)
            else:
                dw, db = gemm_wgrad(dxt16, a2, want_bias=True)
        return da, dxt.view(ctx.xshape), dw, db, None, None, None, dgamma, dbeta, None, None, None, None


def residual_linear_ln(a, x, lin, norm, y_f32=False):
    """x' = x + lin(a) on the fp32 stream and y = norm(x') (bf16; fp32 with y_f32) -> (x', y); `lin` an nn.Linear with registered shadows,
    `norm` an nn.LayerNorm"""
    weight, bias = lin.weight, lin.bias
    ew, eb, et = _shadows.get(id(weight)), _shadows.get(id(bias)), _transposed.get(id(weight))
    if not (ew is not None and eb is not None and et is not None and ew[0] is weight and eb[0] is bias and et[0] is weight
            and ew[1].dtype == torch.bfloat16 and weight.shape[0] % 128 == 0 and weight.shape[1] % 128 == 0):
        raise _C.CosaError("residual_linear_ln: the projection needs registered bf16 shadows (W, b, W^T) and 128-aligned shapes "
                           "(CoSATrainer / nn_ops.ensure_shadows + TransposedShadows register them)")
    gw16, gb16 = _gamma16(norm.weight, norm.bias)
    return ResidualLinearLNFn.apply(a, x, weight, bias, ew[1], eb[1], et[1], norm.weight, norm.bias, gw16, gb16, norm.eps, bool(y_f32))


class PatchFanoutFn(Function):
    """fp32 tokens x [B, N, 768] -> n views of ONE bf16 copy of the PATCH tokens [B, N - 1, 768] (the class token feeds none of the heads'
    consumers: models/__init__.py:163-206); backward: the consumers' bf16 gradients are summed in fp32 (in consumer order) into dx with a zero class-token row by one kernel -- autograd's version of the same sum is a zero-fill and a slice copy per
    consumer, a cast and the adds.  Why fp32: with a single bf16 tensor feeding several consumers autograd would form that sum in bf16 -- and
    the sum at the final norm's output (decoder + CAM head + classification head) goes straight into LayerNorm', which subtracts its row
    mean: the bf16 rounding of the sum survives that cancellation as a few per cent of the result (measured in round 3: weight-gradient
    cosine 0.995 in the LAST block already)."""

    @staticmethod
    def forward(ctx, x, n):
        assert x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 768 and x.is_cuda
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)                 # an unused view (the returned feature map) arrives as None, not as a zero tensor
        p16 = x[:, 1:].to(torch.bfloat16)                # one strided-read cast kernel
        return tuple(p16.view_as(p16) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        B, N, D = ctx.shape
        gs = [g.contiguous() if g is not None else None for g in gs]
        live = [g for g in gs if g is not None]
        if not (1 <= len(live) <= 3 and all(g.dtype == torch.bfloat16 and g.shape == (B, N - 1, D) for g in live)):      # (not an assert: python -O must not strip it)
            raise _C.CosaError(f"PatchFanoutFn.backward: {len(live)} live gradients (the junction kernel sums at most three bf16 [B, N-1, 768] maps)")
        live += [None] * (3 - len(live))
        dx = torch.empty((B, N, D), device=live[0].device, dtype=torch.float32)
        _C.check(_C.lib().cosa_token_junction_bwd(_C.ptr(live[0]), _C.ptr(live[1]), _C.ptr(live[2]), _C.ptr(dx), B, N, D, _C.stream_ptr()),
                 "cosa_token_junction_bwd")
        return dx, None


def patch_fanout_bf16(x, n):
    return PatchFanoutFn.apply(x, n)
