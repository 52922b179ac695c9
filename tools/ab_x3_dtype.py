"""fp16x3 against bf16x3 on the SAME three-term kernels, interleaved on one box: the teacher's projection launches (qkv, proj, fc1 + GELU, fc2 +
residual at M = all tokens of a step) and the attention forward at N = 1765.  The two instantiations differ only in the element type of the
hi / lo halves (the MFMA instruction, the split conversions); prints microseconds per launch, best and median of `reps` rounds."""
import sys, statistics, torch
sys.path.insert(0, '.')
from cosa_amd import nn_ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = 87904
H = 12


def timed(fn, n=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def cases(hdt):
    out = {}
    for name, N, K, epi in (("qkv", 2304, 768, nn_ops.EPI_BIAS), ("proj", 768, 768, nn_ops.EPI_RESIDUAL), ("fc1", 3072, 768, nn_ops.EPI_GELU),
                            ("fc2", 768, 3072, nn_ops.EPI_RESIDUAL)):
        torch.manual_seed(K + N)
        x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.03; b = torch.randn(N, device='cuda')
        xs = nn_ops.split_rows(x, ones=True, dtype=hdt)
        ws = nn_ops.split_rows(w, bias=b, dtype=hdt)
        del x, w
        if epi == nn_ops.EPI_RESIDUAL:
            res = torch.randn(M, N, device='cuda')
            out[name] = (lambda xs=xs, ws=ws, N=N, K=K, res=res: nn_ops.gemm_x3(xs, ws, M, N, K, nn_ops.EPI_RESIDUAL, residual=res, out=res))
        else:
            y = torch.zeros((M, nn_ops.split_ld(N)), device='cuda', dtype=hdt)
            out[name] = (lambda xs=xs, ws=ws, N=N, K=K, y=y, epi=epi: nn_ops.gemm_x3(xs, ws, M, N, K, epi, out=y, ldy=nn_ops.split_ld(N)))
    B, N = 32, 1765
    torch.manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, device='cuda') * 1.5
    qs = nn_ops.split_rows(qkv, dtype=hdt)[:, :6 * H * 64].contiguous()
    o = torch.zeros(B * N, 2 * H * 64 + 64, device='cuda', dtype=hdt)
    lse = torch.empty(B, H, N, device='cuda')
    out["attn1765"] = (lambda: nn_ops.attn_fwd_x3(qs, B, N, H, o, lse))
    return out


fns = {"fp16": cases(torch.float16), "bf16": cases(torch.bfloat16)}
t = {k: {n: [] for n in v} for k, v in fns.items()}
for k in fns:
    for f in fns[k].values():
        f()
torch.cuda.synchronize()
for _ in range(reps):
    for name in fns["fp16"]:
        for k in ("fp16", "bf16"):
            t[k][name].append(timed(fns[k][name]))
for name in fns["fp16"]:
    a, b = t["fp16"][name], t["bf16"][name]
    print("%-9s fp16x3 best %7.1f median %7.1f us | bf16x3 best %7.1f median %7.1f us | fp16/bf16 (median) %.3f" % (
        name, min(a), statistics.median(a), min(b), statistics.median(b), statistics.median(a) / statistics.median(b)), flush=True)
