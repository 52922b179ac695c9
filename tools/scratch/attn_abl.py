import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cosa_amd import _C
B, N, H = 32, 1765, 12
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, device='cuda').bfloat16()
L = _C.lib(); ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), 'cuda', 'attn')
out = torch.empty(B, N, H * 64, device='cuda', dtype=torch.bfloat16); lse = torch.empty(B, H, N, device='cuda')
def run(abl, n=10):
    f = lambda: L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, abl << 8, None, _C.ptr(ws), ws.numel(), _C.stream_ptr())
    for _ in range(3): f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
fl = 4.0 * B * H * N * N * 64
names = {0: "full", 1: "no S mfma", 2: "no exp", 4: "no PV mfma", 8: "no wait+barrier", 16: "no K/V dma", 24: "no dma, no barrier", 5: "no mfma at all", 7: "no mfma, no exp", 2 | 8: "no exp, no barrier", 31: "everything off"}
for abl, nm in names.items():
    t = run(abl)
    print(f"abl {abl:2d} {nm:22s}: {t:7.1f} us  ({fl / t / 1e6:6.0f} TF/s equiv)", flush=True)
