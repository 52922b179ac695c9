import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from cosa_amd import nn_ops
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (B,h,Cin,Cout) in [(32,28,768,512),(32,28,512,512),(32,42,768,512),(32,14,768,512)]:
    tok=torch.randn(B,h*h,Cin,device='cuda').bfloat16(); w=(torch.randn(Cout,Cin,3,3,device='cuda')*0.02).bfloat16()
    x=tok.view(B,h,h,Cin).permute(0,3,1,2); wcl=w.contiguous(memory_format=torch.channels_last)
    t1=timeit(lambda: nn_ops.conv3x3_dilated_tokens(tok,w,B,h,h,5)); t2=timeit(lambda: F.relu(F.conv2d(x,wcl,padding=5,dilation=5)))
    fl=2.0*B*h*h*Cout*Cin*9/1e12
    print(f"conv B={B} {h}x{h} {Cin}->{Cout}: mine {t1*1e3:.0f}us {fl/t1*1e3:.0f} TF | MIOpen {t2*1e3:.0f}us {fl/t2*1e3:.0f} TF")
