"""evaluation-path throughput (SURVEY f-1): evaluate() over synthetic VOC-val-shaped images (batch 1, ~375x500, five scales x two flips)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from cosa_amd import evaluation_engine as ee
from cosa_amd.models import build_model
from cosa_amd.train_step import default_args
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
args = default_args("VOC12", crop_size=448, batch_size=1)
torch.manual_seed(0)
model = build_model(args).cuda().eval()
rng = np.random.default_rng(0)
sizes = [(375, 500), (333, 500), (500, 375), (366, 500), (500, 281)]
loader = []
for i in range(n):
    H, W = sizes[i % len(sizes)]
    cls = torch.zeros(1, 20); cls[0, rng.choice(20, 2, replace=False)] = 1
    loader.append(("x", torch.randn(1, 3, H, W), torch.from_numpy(rng.integers(0, 21, (1, H, W))), cls))
ee.evaluate(model, loader[:12], args, epoch=0, eval_group=int(sys.argv[2]) if len(sys.argv) > 2 else 4)                      # warm-up (kernel selection, workspaces)
torch.cuda.synchronize(); t0 = time.perf_counter()
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tab, miou, df, aps = ee.evaluate(model, loader, args, epoch=1, eval_group=grp)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"evaluate: {n} images in {dt:.2f} s = {n/dt:.1f} img/s ({dt/n*1e3:.1f} ms/img; 10 encoder passes each); VOC val (1449 images) ~ {1449*dt/n:.0f} s on one GPU")
