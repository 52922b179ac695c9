"""torch (ATen) reference operators for the HOST-LOGIC tests -- test infrastructure, not product.

The product (cosa_amd/) has exactly one implementation of every operator: a kernel of libcosa_hip.so.  Shapes / dtypes outside the
kernels' envelope (the 128-wide toy encoders of tests/golden/vit_tiny.npz, an fp32 "parity mode") raise CosaError at
cosa_amd.nn_ops.reference_op.  The tests that check module wiring, state-dict names and the loss algebra on such shapes install the
operators below for their duration:

    with torch_reference_ops():
        out = net(x)

What they check is then the HOST code around the operators, not HIP kernels, and they say so."""
import torch
import torch.nn.functional as F

from cosa_amd import nn_ops


def _linear(x, w, b, act=False):                     # nn.Linear (+ nn.GELU): models/vit/vit.py:96-102,121,135
    y = F.linear(x, w, b)
    return F.gelu(y) if act else y


def _attention(qkv, H):                              # models/vit/vit.py:128-134, head dim 64, exact math in the tensor's dtype
    B, N, _ = qkv.shape
    q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = torch.matmul(q, k.transpose(-1, -2)) * 0.125
    return torch.matmul(att.softmax(-1), v).transpose(1, 2).reshape(B, N, H * 64)


def _layer_norm(x, w, b, eps):                       # nn.LayerNorm(dim, eps=1e-6)
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _largefov(x, w6, w7, w8, dilation):              # models/decoder/conv_head.py:32-41 on an NCHW view with channels-last strides
    cl = lambda w: w.contiguous(memory_format=torch.channels_last)
    x = F.relu(F.conv2d(x, cl(w6), padding=dilation, dilation=dilation))
    x = F.relu(F.conv2d(x, cl(w7), padding=dilation, dilation=dilation))
    return F.conv2d(x, w8)


OPS = {"linear": _linear, "attention": _attention, "layer_norm": _layer_norm, "largefov": _largefov}


class torch_reference_ops:
    """`with torch_reference_ops():` -- install the reference operators into cosa_amd.nn_ops for the duration of a host-logic test"""

    def __enter__(self):
        self._prev = nn_ops._reference_ops
        nn_ops._reference_ops = OPS
        return self

    def __exit__(self, *exc):
        nn_ops._reference_ops = self._prev
        return False
