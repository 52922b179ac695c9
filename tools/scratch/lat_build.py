import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cosa_amd.train_step import synthetic_batch
from cosa_amd.utils import seg_helper
dev = torch.device("cuda:0")
b, C, S = 16, 20, 448
wimg, simg, lab, box = synthetic_batch(b, S, C, dev, seed=1234)
pl = seg_helper.PreparedLattice(15, 50.0)
def f():
    pl.start(simg, C + 1); pl.event.synchronize()
for _ in range(3): f()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
print("abl", os.environ.get("COSA_LAT_ABL"), "prepare ms", round(a.elapsed_time(e) / 10, 4), flush=True)
