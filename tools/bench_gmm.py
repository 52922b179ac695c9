"""Adaptive-threshold fit at the training loop's size (batch 16 x ratio 100 rows x 28^2 cells): device launch vs the CPU path
(the numpy oracle = port of the reference's scikit-learn call; scikit-learn itself when importable).
usage: python tools/bench_gmm.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cosa_amd.utils import seg_helper
from oracle import gmm_oracle
from oracle.gen_golden import gmm_queue

q = gmm_queue(np.random.default_rng(5), 1600, 784, 300)
qd = torch.from_numpy(q).cuda()
for _ in range(3):
    out = seg_helper.rungmm_device(qd, 3, 0.05)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    out = seg_helper.rungmm_device(qd, 3, 0.05)
e.record()
torch.cuda.synchronize()
dev_ms = a.elapsed_time(e) / 20
t = time.perf_counter()
ref = gmm_oracle.rungmm(q, 3, 0.05)
cpu_s = time.perf_counter() - t
res = {"samples": int((q > 0.05).sum()), "device_ms_per_fit": round(dev_ms, 3), "iterations": int(out[2].item()),
       "cpu_port_s_per_fit": round(cpu_s, 3), "thresholds_equal": [float(out[0]), float(out[1])] == list(ref)}
try:
    import sklearn.mixture as skm
    x = q.flatten()
    x = x[x > 0.05].reshape(-1, 1)
    t = time.perf_counter()
    gm = skm.GaussianMixture(3, weights_init=[1 / 3] * 3, means_init=[[x.min()], [np.median(x)], [x.max()]],
                             precisions_init=[[[1.0]]] * 3)
    gm.fit_predict(x)
    res["sklearn_s_per_fit"] = round(time.perf_counter() - t, 3)
except ImportError:
    pass
print(json.dumps(res))
