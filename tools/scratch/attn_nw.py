import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cosa_amd import _C
L = _C.lib()
torch.manual_seed(0)
for (B, N, H) in [(32, 1765, 12), (32, 785, 12), (16, 785, 12), (32, 197, 12), (8, 3601, 12), (32, 1601, 12)]:
    qkv = torch.randn(B, N, 3 * H * 64, device='cuda').bfloat16()
    ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), 'cuda', 'attn')
    res = {}
    for rep in range(4):
      for name, fl in (("NW=2", 0x200), ("NW=4", 0x100)):
        out = torch.empty(B, N, H * 64, device='cuda', dtype=torch.bfloat16); lse = torch.empty(B, H, N, device='cuda')
        f = lambda: L.cosa_attn_fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, fl, None, _C.ptr(ws), ws.numel(), _C.stream_ptr())
        for _ in range(3): f()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / 10 * 1e3
        if name not in res or t < res[name][0]:
            res[name] = (t, out.clone(), lse.clone())
    flops = 4.0 * B * H * N * N * 64
    same = torch.equal(res["NW=2"][1], res["NW=4"][1]) and torch.equal(res["NW=2"][2], res["NW=4"][2])
    print(f"B={B} N={N}: " + "  ".join(f"{k} {v[0]:7.1f} us ({flops / v[0] / 1e6:5.0f} TF/s)" for k, v in res.items()) + f"  bit-identical {same}", flush=True)
