import sys, numpy as np, torch
sys.path.insert(0, ".")
from cosa_amd.utils.seg_helper import DenseCRF
from oracle import c_oracle
H, W = 23, 31
for name, v in (("ones", np.ones((1, H, W), np.float32)), ("imp", None)):
    if v is None:
        v = np.zeros((1, H, W), np.float32); v[0, 11, 15] = 1
    ref, M = c_oracle.gaussian_filter_d2(v, H, W, 1.0)
    got = DenseCRF._filter_gauss(torch.from_numpy(v).cuda(), 1.0).cpu().numpy()
    print(name, "oracle M", M, "ref min/max", ref.min(), ref.max(), "got min/max", got.min(), got.max(), "maxdiff", np.abs(got - ref).max())
    if name == "imp":
        print("ref around", ref[0, 9:14, 13:18]); print("got around", got[0, 9:14, 13:18])
# the 5-D build for comparison
img = np.random.default_rng(0).random((1, 3, H, W)).astype(np.float32) * 255
v = np.random.default_rng(1).random((1, 2, H, W)).astype(np.float32)
ref5, _ = c_oracle.bilateralfilter_batch(img, v, 1, 2, H, W, 5.0, 20.0)
got5 = DenseCRF._filter_bilateral(torch.from_numpy(img[0]).cuda(), torch.from_numpy(v[0]).cuda(), 5.0, 20.0).cpu().numpy()
print("d5 maxdiff", np.abs(got5 - ref5[0]).max())
