import torch, inspect
print(torch.__version__)
try:
    a=torch.cuda.Event(enable_timing=True, external=True); b=torch.cuda.Event(enable_timing=True, external=True)
except Exception as e:
    print("external events unsupported:", e); raise SystemExit
x=torch.randn(4096,4096,device='cuda'); y=torch.empty_like(x)
for _ in range(2): y=x@x
torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a.record()
    y=x@x
    b.record()
    z=y+1
for i in range(3):
    g.replay()
torch.cuda.synchronize()
print("elapsed inside graph (ms):", a.elapsed_time(b))
