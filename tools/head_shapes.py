"""Time the narrow-head kernel on the shapes of one training step (CAM heads: 768 -> 20, LargeFOV conv8: 512 -> 21).

    python tools/head_shapes.py [reps]
"""
import sys
import torch

sys.path.insert(0, '.')
from cosa_amd import nn_ops  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dt in (torch.float32, torch.bfloat16):
    for B, n, K, N in ((16, 784, 768, 20), (32, 1764, 768, 20), (16, 784, 512, 21), (16, 1, 768, 20)):
        tok = torch.randn(B, n + 1, K, device='cuda').to(dt)[:, 1:]
        w = (torch.randn(N, K, device='cuda') * 0.05).to(dt)
        for _ in range(3):
            nn_ops.head_linear(tok, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            nn_ops.head_linear(tok, w)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REPS
        byts = B * n * K * tok.element_size()
        print("%-8s B=%3d n=%5d K=%4d N=%3d  %7.1f us  %6.2f TB/s" % (str(dt)[6:], B, n, K, N, us, byts / us * 1e-6), flush=True)
