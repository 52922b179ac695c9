"""Dense-CRF post-processing of the final evaluation (utils/seg_helper.py:961-996: `DenseCRF`, `crf_inference_infv2`; called from
evaluation_engine.py:204-211) -- CPU restatement.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic lives in pydensecrf (un-pinned git HEAD in the reference's README, a wrapper of Kraehenbuehl & Koltun's
densecrf), which is neither under /root/reference nor in this image, and the reference holds no fixture of its output.  This file restates
the published algorithm (Kraehenbuehl & Koltun, NIPS 2011; densecrf's DenseCRF::inference / DenseKernel / PottsCompatibility):

    unary    U = -log(clip(p, 1e-5, 1))                                            pydensecrf.utils.unary_from_softmax
    kernels  k_g: features (x / sxy, y / sxy), Potts weight pos_w                    addPairwiseGaussian (DIAG_KERNEL, NORMALIZE_SYMMETRIC)
             k_b: features (x / sxy, y / sxy, r / srgb, g / srgb, b / srgb), bi_w    addPairwiseBilateral
             K(v) = n * F(n * v),  n = 1 / sqrt(F(1) + 1e-20),  F = the permutohedral-lattice filter (splat, d + 1 blurs, slice)
    mean field   Q = softmax(-U);   repeat iter_max times:   Q = softmax(-U + pos_w K_g(Q) + bi_w K_b(Q))

The lattice filter F is NOT restated here: it is oracle/cosa_oracle.c, the code that is pinned bit for bit by the reference's own C++
bilateral filter at d = 5 (tests/golden/bilateral.npz); the 2-D kernel is the same source built with -DORC_PD=2.  (densecrf's scalar
lattice code rounds to the nearest lattice point by comparing distances where that code uses rint(): the two differ at exact ties only.)
"""
import numpy as np

from . import c_oracle


def unary_from_softmax(sm, clip=1e-5):
    return (-np.log(np.clip(np.asarray(sm, np.float32), clip, 1.0))).astype(np.float32)


def _softmax0(x):
    x = x - x.max(0, keepdims=True)
    e = np.exp(x, dtype=np.float32)
    return (e / e.sum(0, keepdims=True)).astype(np.float32)


def dense_crf(image_hwc, probmap, iter_max=1, pos_w=1.0, pos_xy_std=1.0, bi_w=4.0, bi_xy_std=121.0, bi_rgb_std=5.0):
    """image_hwc [H, W, 3] (0..255), probmap [C, H, W] probabilities -> Q [C, H, W] float32"""
    C, H, W = probmap.shape
    U = unary_from_softmax(probmap)
    img = np.ascontiguousarray(np.asarray(image_hwc, np.float32).transpose(2, 0, 1))

    def f_gauss(v):
        return c_oracle.gaussian_filter_d2(np.ascontiguousarray(v, np.float32), H, W, pos_xy_std)[0]

    def f_bilat(v):
        return c_oracle.bilateralfilter_batch(img[None], np.ascontiguousarray(v, np.float32)[None], 1, v.shape[0], H, W, bi_rgb_std, bi_xy_std)[0][0]

    one = np.ones((1, H, W), np.float32)
    n_g = (1.0 / np.sqrt(f_gauss(one) + np.float32(1e-20))).astype(np.float32)
    n_b = (1.0 / np.sqrt(f_bilat(one) + np.float32(1e-20))).astype(np.float32)
    Q = _softmax0(-U)
    for _ in range(int(iter_max)):
        t = -U + np.float32(pos_w) * (n_g * f_gauss(Q * n_g)) + np.float32(bi_w) * (n_b * f_bilat(Q * n_b))
        Q = _softmax0(t.astype(np.float32))
    return Q


def crf_inference_infv2(image_hwc, probmap):
    """the reference's only configuration (utils/seg_helper.py:989-996)"""
    return dense_crf(image_hwc, probmap, iter_max=1, pos_w=1, pos_xy_std=1, bi_w=4, bi_xy_std=121, bi_rgb_std=5)
