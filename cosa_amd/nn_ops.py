"""Differentiable building blocks of the ViT hot path, backed by the HIP kernels in libcosa_hip.so.

Every op here is an explicit torch.autograd.Function around C-ABI calls (no tracing compiler, no
Triton).  Plain library GEMMs (F.linear -> hipBLASLt) are used only where a bare GEMM is all
there is to do; everything fused or attention-shaped is hand-written HIP.
"""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _C


# --------------------------------------------------------------------------------------------
# attention  (models/vit/vit.py:119-137)
# --------------------------------------------------------------------------------------------
_flops = {}   # algorithmic FLOPs issued per profiled kernel (read by bench.py)


def _attn_fwd(qkv, B, N, H):
    out = torch.empty((B, N, H * 64), device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty((B, H, N), device=qkv.device, dtype=torch.float32)
    L = _C.lib()
    ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), qkv.device, "attn")
    _C.check(L.cosa_attn_prepare_vt(_C.ptr(qkv), B, N, H, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "cosa_attn_prepare_vt")
    with _C.profiled("attn_fwd"):
        _C.check(L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, 1, _C.ptr(ws), ws.numel(),
                                 _C.stream_ptr()), "cosa_attn_fwd")
    _flops["attn_fwd"] = _flops.get("attn_fwd", 0) + 4.0 * B * H * N * N * 64
    return out, lse


class FusedAttention(Function):
    """softmax(q k^T / sqrt(64)) v over packed qkv [B,N,3*H*64] (bf16) -> [B,N,H*64].

    Forward is the LDS-tiled MFMA kernel (cosa_attn_fwd).  Backward recomputes P from the saved
    log-sum-exp per head with batched GEMMs (rocBLAS/hipBLASLt) -- the student pass is 3 % of the
    step's attention FLOPs; a fused backward kernel is the next kernel on the list."""

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
        q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)          # [B,H,N,64] views
        do = dout.reshape(B, N, H, 64).permute(0, 2, 1, 3)
        o = out.view(B, N, H, 64).permute(0, 2, 1, 3)
        s = torch.matmul(q, k.transpose(-1, -2)).float().mul_(0.125)        # [B,H,N,N]
        p = torch.exp(s - lse.unsqueeze(-1))
        del s
        delta = (do.float() * o.float()).sum(-1, keepdim=True)
        pb = p.to(torch.bfloat16)
        dv = torch.matmul(pb.transpose(-1, -2), do)
        dp = torch.matmul(do, v.transpose(-1, -2)).float()
        ds = (p * (dp - delta)).mul_(0.125).to(torch.bfloat16)
        del p, dp
        dq = torch.matmul(ds, k)
        dk = torch.matmul(ds.transpose(-1, -2), q)
        dqkv = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * 64)
        return dqkv, None


def attention(qkv, H):
    """qkv [B,N,3*H*64].  bf16 -> HIP kernel; fp32 (parity mode) -> exact fp32 math in torch."""
    if qkv.dtype == torch.bfloat16:
        return FusedAttention.apply(qkv, H)
    B, N, _ = qkv.shape
    q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = torch.matmul(q, k.transpose(-1, -2)) * 0.125
    return torch.matmul(att.softmax(-1), v).transpose(1, 2).reshape(B, N, H * 64)


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
_cast_cache = {}


def cast_param(p, dtype):
    """bf16 view of an fp32 master parameter.  With grad: differentiable cast.  Without grad
    (teacher passes): cached per parameter version, so 6 teacher forwards cast each weight once."""
    if p.dtype == dtype:
        return p
    if torch.is_grad_enabled() and p.requires_grad:
        return p.to(dtype)
    key = id(p)
    ent = _cast_cache.get(key)
    if ent is None or ent[0] != p._version or ent[1].device != p.device or ent[2] is not p:
        ent = (p._version, p.detach().to(dtype), p)
        _cast_cache[key] = ent
    return ent[1]


def gelu(x):
    return F.gelu(x)
