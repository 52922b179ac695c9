"""three-term attention forward (the fp16x3 teacher's; `bf16` as argument: bf16x3 halves) at the sequence lengths of the step: microseconds and
algorithmic TFLOP/s (3 MFMA terms are issued per product: the issued rate is three times the printed one)"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
H = 12
hdt = torch.bfloat16 if 'bf16' in sys.argv[1:] else torch.float16
for B, N in [(32, 785), (32, 197), (32, 1765)]:
    torch.manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, device='cuda') * 1.5
    qs = nn_ops.split_rows(qkv, dtype=hdt)[:, :6 * H * 64].contiguous()
    out = torch.zeros(B * N, 2 * H * 64 + 64, device='cuda', dtype=hdt)
    lse = torch.empty(B, H, N, device='cuda')
    for _ in range(3):
        nn_ops.attn_fwd_x3(qs, B, N, H, out, lse)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        nn_ops.attn_fwd_x3(qs, B, N, H, out, lse)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print("B=%d N=%4d  %8.1f us  %6.1f TFLOP/s (algorithmic)" % (B, N, us, 4.0 * B * H * N * N * 64 / us * 1e-6), flush=True)
