import os, sys, torch
sys.path.insert(0, ".")
from cosa_amd import nn_ops
M = 87904
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
out = []
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    timeit(lambda: nn_ops.gemm_bf16(x, w, b, epi, out=o), 10)
    out.append(f"{name} {timeit(lambda: nn_ops.gemm_bf16(x, w, b, epi, out=o)):.1f}")
    ref = torch.nn.functional.linear(x.float()[:4096], w.float(), b.float())
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    out.append(f"err {(o[:4096].float() - ref).abs().max().item():.3f}")
print("stagger", os.environ.get("COSA_GEMM_STAGGER"), " | ".join(out), flush=True)
