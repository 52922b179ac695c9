import sys, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops
M=87904
for N,K,epi in ((3072,768,0),(3072,768,1),(3072,768,0),(3072,768,1)):
    x = (torch.randn(M, K, device='cuda')).bfloat16(); w = (torch.randn(N, K, device='cuda') * 0.03).bfloat16(); b = torch.randn(N, device='cuda').bfloat16()
    for _ in range(3): nn_ops.gemm_bf16(x, w, b, epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): nn_ops.gemm_bf16(x, w, b, epi)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("N=%d K=%d epi=%d %.1f us %.0f TF" % (N, K, epi, us, 2.0*M*N*K/us*1e-6), flush=True)
