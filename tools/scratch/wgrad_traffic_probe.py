"""Is the batched weight gradient (gemm_wgrad_batched_kernel, 2592 tiles of 256 x 128 over M = 12 560 tokens) limited by operand traffic?
Three launches of the SAME job list (48 linears of 12 blocks), differing only in which memory the operands alias:
  A  every linear its own dY / X tensors (the training step: 3.7 GB of distinct operand bytes, more than the 256-MB Infinity Cache)
  B  all 12 blocks alias block 0's tensors (0.31 GB distinct: re-reads are served on-die)
  C  B with dY / X of every linear aliasing ONE small pair per shape class rotated so that concurrent jobs read different addresses: same as B
     but the launch is repeated back to back (fully cache-warm)
If A, B take the same time the launch is not bound by HBM / fabric traffic, whatever FETCH_SIZE says.  usage (GPU box): python tools/scratch/wgrad_traffic_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cosa_amd import nn_ops

dev = torch.device("cuda", 0)
M = 12560
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]          # (N, K) of qkv, proj, fc1, fc2


def make(blocks_distinct):
    pairs = []
    base = [(torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16()) for N, K in shapes]
    for blk in range(12):
        for i, (N, K) in enumerate(shapes):
            if blocks_distinct and blk > 0:
                pairs.append((torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(), True))
            else:
                pairs.append((base[i][0], base[i][1], True))
    return pairs


def timed(pairs, n=6):
    for _ in range(2):
        nn_ops.gemm_wgrad_batched(pairs)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        nn_ops.gemm_wgrad_batched(pairs)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


flops = sum(2.0 * M * N * K for N, K in shapes) * 12
A = timed(make(True))
B = timed(make(False))
print(json.dumps({"A_distinct_ms": round(A, 3), "B_aliased_blocks_ms": round(B, 3), "TFLOPs_A": round(flops / A / 1e9, 1), "TFLOPs_B": round(flops / B / 1e9, 1)}))
