"""PAR refine timing alone (the second half of the headline metric): cam2mask_multi on main+aux CAM sets, b=16 448^2.
usage: python tools/bench_par.py [chunk sizes ...]     (chunk sizes: also time the batch refined in slices of that many images -- does the
48-weight affinity tensor of a slice stay in the Infinity Cache across the ten steps?)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosa_amd.models.PAR import PAR
from cosa_amd.utils import seg_helper

dev = torch.device("cuda:0")
b, C, S = 16, 20, 448
g = torch.Generator(device="cpu").manual_seed(7)
up = lambda t: torch.nn.functional.interpolate(t.to(dev), size=(S, S), mode="bilinear")
cams, cams_aux = up(torch.rand(b, C, S // 8, S // 8, generator=g)), up(torch.rand(b, C, S // 8, S // 8, generator=g))
den = up(torch.rand(b, 3, S // 4, S // 4, generator=g))
lab = torch.zeros(b, C)
for i in range(b):
    lab[i, torch.randperm(C, generator=g)[: 1 + i % 3]] = 1
lab = lab.to(dev)
box = torch.tensor([[0, S, 0, S]] * b)
par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24])

def timed(f, n=10):
    f(); f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n

multi = lambda r: (lambda: seg_helper.cam2mask_multi(den, box, [cams, cams_aux], lab, [0.7, 0.7], [0.25, 0.25], refine_model=r, _fold_validation=True))
sep = lambda r: (lambda: (seg_helper.cam2mask(den, box, cams, lab, 0.7, 0.25, refine_model=r, _fold_validation=True),
                          seg_helper.cam2mask(den, box, cams_aux, lab, 0.7, 0.25, refine_model=r, _fold_validation=True)))
out = {
       "multi_ms": round(timed(multi(par)) - timed(multi(None)), 4), "separate_ms": round(timed(sep(par)) - timed(sep(None)), 4)}
for ch in [int(a) for a in sys.argv[1:]]:
    def chunked(r, ch=ch):
        def f():
            for i in range(0, b, ch):
                seg_helper.cam2mask_multi(den[i:i + ch], box[i:i + ch], [cams[i:i + ch], cams_aux[i:i + ch]], lab[i:i + ch], [0.7, 0.7], [0.25, 0.25],
                                          refine_model=r, _fold_validation=True)
        return f
    out[f"multi_chunks_of_{ch}_ms"] = round(timed(chunked(par)) - timed(chunked(None)), 4)
print(json.dumps(out))
