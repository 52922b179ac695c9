import sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cosa_amd import nn_ops
from test_precision_gpu import _c8_fields
for (B, N, H) in [(1, 64, 2), (2, 100, 3)]:
    torch.manual_seed(N)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D, device="cuda") * 1.5).half()
    plain, lse_p = nn_ops._attn_fwd(qkv, B, N, H)
    out = torch.zeros(B * N, 2 * D + 64, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, N, device="cuda")
    nn_ops.attn_fwd_c8(qkv, B, N, H, out, lse)
    hi, lo8, hi8, aug = _c8_fields(out, D)
    p = plain.view(B * N, D).float()
    bad = (hi != p)
    print(B, N, H, "hi mismatches", int(bad.sum()), "of", bad.numel(), "lse equal", torch.equal(lse, lse_p), "max diff", (hi - p).abs().max().item())
    if bad.any():
        idx = bad.nonzero()
        print(" rows", idx[:, 0].unique()[:20].tolist(), " cols", idx[:, 1].unique()[:40].tolist())
        r, c = idx[0].tolist()
        print(" first", r, c, hi[r, c].item(), p[r, c].item())
