"""which hipBLASLt kernels TunableOp picks for the teacher's qkv / fc1 shapes (run under rocprofv3 --kernel-trace --stats)"""
import torch
dev = torch.device("cuda", 0)
for N, K in ((2304, 768), (3072, 768)):
    x = torch.randn(87904, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    for _ in range(12):
        torch.mm(x, w.t())
torch.cuda.synchronize()
