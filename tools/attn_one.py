"""one launch shape of the attention forward for rocprofv3 --pmc passes: the teacher's scale-1.5 launch (B = 32, N = 1765, 12 heads), fp16 operands,
no-grad variant (flag bit 10: the kernel the teacher's passes run); `train` as argument: the training kernel on bf16 operands"""
import sys, torch
sys.path.insert(0, '.')
from cosa_amd import _C
train = len(sys.argv) > 1 and sys.argv[1] == "train"
B, N, H = 32, 1765, 12
dt = torch.bfloat16 if train else torch.float16
qkv = torch.randn(B, N, 3 * H * 64, device='cuda').to(dt)
L = _C.lib()
fwd = L.cosa_attn_fwd if train else L.cosa_attn_fwd_f16
ws = _C.workspace(L.cosa_attn_workspace_bytes(B, N, H), 'cuda', 'attn')
out = torch.empty(B, N, H * 64, device='cuda', dtype=dt)
lse = torch.empty(B, H, N, device='cuda')
for _ in range(3):
    fwd(_C.ptr(qkv), _C.ptr(out), _C.ptr(lse), B, N, H, 64, 0.125, 0 if train else 0x400, None, _C.ptr(ws), ws.numel(), _C.stream_ptr())
torch.cuda.synchronize()
