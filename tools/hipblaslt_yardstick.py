"""Vendor yardstick (NOT part of the product): tuned hipBLASLt through torch (TunableOp) against this repository's persistent
GEMM on the projection shapes of a training step -- the teacher's M = 87 904 rows and the student's M = 12 560 -- bf16 operands, bias epilogue.
Run on the GPU box:  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python tools/hipblaslt_yardstick.py > gpurun_out/r04_hipblaslt_yardstick.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from cosa_amd import nn_ops

dev = torch.device("cuda", 0)


def timed(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3          # us


print(f"# TunableOp enabled={os.environ.get('PYTORCH_TUNABLEOP_ENABLED')} tuning={os.environ.get('PYTORCH_TUNABLEOP_TUNING')}; torch {torch.__version__}")
print("# shape (M, N, K)            own kernel us (TFLOP/s)      torch F.linear us (TFLOP/s)      torch mm (no bias) us      own / library")
for M in (87904, 12560):
    for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        fl = 2.0 * M * N * K
        t_own = timed(lambda: nn_ops.gemm_bf16(x, w, b, nn_ops.EPI_BIAS))
        t_lin = timed(lambda: F.linear(x, w, b))
        t_mm = timed(lambda: torch.mm(x, w.t()))
        best = min(t_lin, t_mm)
        print(f"{name:5s} ({M:6d}, {N:4d}, {K:4d})   {t_own:8.1f} ({fl / t_own / 1e6:6.0f})        {t_lin:8.1f} ({fl / t_lin / 1e6:6.0f})        {t_mm:8.1f} ({fl / t_mm / 1e6:6.0f})        {t_own / best:.3f}", flush=True)
