import numpy as np, torch, sys
sys.path.insert(0, '.')
from cosa_amd import _C
from oracle import c_oracle as co
from oracle.gen_golden import synth_image255, smooth_field
L = _C.lib()
rng = np.random.default_rng(9)
def run(N,K,H,W,kind):
    img = synth_image255(rng, N, H, W) if kind=='s' else rng.uniform(0,255,(N,3,H,W)).astype(np.float32)
    seg = rng.uniform(0,1,(N,K,H,W)).astype(np.float32)
    ref, Mref = co.bilateralfilter_batch(img, seg, N,K,H,W,15.0,50.0)
    d = lambda a: torch.from_numpy(a).cuda()
    out = torch.empty(N,K,H,W,device='cuda'); Ms = torch.zeros(N,dtype=torch.int32,device='cuda')
    ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N,K,H,W),'cuda','dbg')
    _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(d(img)),_C.ptr(d(seg)),_C.ptr(out),N,K,H,W,15.0,50.0,_C.ptr(Ms),_C.ptr(ws),ws.numel(),_C.stream_ptr()))
    o = out.cpu().numpy()
    err = np.abs(o-ref)/np.maximum(np.abs(ref),1e-3)
    print(N,K,H,W,kind,'M',Ms.cpu().numpy(),Mref,'maxrel',err.max(),'mean out',o.mean(),'mean ref',ref.mean(), 'err flag', int(ws[:4].view(torch.int32).item()))
for cfg in [(1,3,64,64,'s'),(1,3,128,128,'s'),(1,3,224,224,'s'),(1,21,224,224,'s'),(2,21,224,224,'s'),(1,3,224,224,'n'),(2,21,224,224,'n')]:
    run(*cfg)
