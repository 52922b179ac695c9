"""Timing ablations of attn_fwd2_kernel.  NOT runnable against the shipped library: the numbers were taken with a temporary
build in which the kernel took an `abl` argument (flags >> 8) that switched off, one at a time, the S MFMAs (bit 0), the exponentials (bit 1:
the fma kept), the PV MFMAs (bit 2), the vmcnt(0) + barrier at the end of a tile (bit 3) and the K/V LDS-DMA (bit 4); the runtime branches
themselves slowed that build down (539 us against 380 us for the shipped kernel at B = 32, N = 1765), so only the differences mean anything.
In the shipped library flags bits 8 / 9 select 4 / 2 waves per workgroup instead."""
import sys
if "--i-have-the-ablation-build" not in sys.argv:
    sys.exit(__doc__)
import os, torch
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
