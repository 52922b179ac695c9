import torch, sys
sys.path.insert(0, '.')
from cosa_amd import _C, nn_ops
torch.manual_seed(0)
for (M, N, K) in [(130, 3072, 768), (4096, 256, 128), (4099, 768, 768), (12560, 3072, 768), (70000, 256, 128)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5 * 3).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    h = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    a = torch.full((M, N), 9.0, device="cuda", dtype=torch.bfloat16)
    _C.check(_C.lib().cosa_gemm_bf16_dual_gelu(_C.ptr(x), _C.ptr(w), _C.ptr(b), _C.ptr(h), _C.ptr(a), M, N, K, _C.stream_ptr()), "dual")
    ref = x.float() @ w.float().t() + b.float()
    eh = (h.float() - ref).abs()
    ea = (a.float() - torch.nn.functional.gelu(ref)).abs()
    print(M, N, K, "h err", eh.max().item(), "a err", ea.max().item(), "untouched h", (h == 7).float().mean().item(), "untouched a", (a == 9).float().mean().item())
    bad = (ea > 0.1).nonzero()
    if len(bad):
        print("  bad a rows", bad[:, 0].min().item(), bad[:, 0].max().item(), "cols", bad[:, 1].min().item(), bad[:, 1].max().item(), "count", len(bad))
        r, c = bad[0].tolist()
        print("  first bad", r, c, a[r, c].item(), torch.nn.functional.gelu(ref)[r, c].item(), "h there", h[r, c].item(), ref[r, c].item())
        # histogram of bad rows mod 256 / cols mod 256
        print("  rows%256", torch.unique(bad[:, 0] % 256)[:20].tolist(), "cols%256", torch.unique(bad[:, 1] % 256)[:40].tolist())
