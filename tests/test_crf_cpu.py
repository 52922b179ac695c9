"""CPU checks of the dense-CRF oracle (oracle/crf_oracle.py; parity with pydensecrf itself is UNPINNED -- see that file's header): the
lattice filter it stands on is the pinned C oracle, built for a 2-D lattice from the same source."""
import numpy as np


def test_d2_lattice_filter_is_linear_symmetric_and_local(oracle_c):
    rng = np.random.default_rng(0)
    H, W = 23, 31
    a, b = rng.random((2, 1, H, W)).astype(np.float32)
    fa, M = oracle_c.gaussian_filter_d2(a, H, W, 1.0)
    fb, _ = oracle_c.gaussian_filter_d2(b, H, W, 1.0)
    fab, _ = oracle_c.gaussian_filter_d2(a + 2 * b, H, W, 1.0)
    np.testing.assert_allclose(fab, fa + 2 * fb, rtol=2e-5, atol=1e-6)                    # linear
    np.testing.assert_allclose((fa * b).sum(), (a * fb).sum(), rtol=1e-3)                 # <F a, b> = <a, F b> up to the fixed blur-axis order
    assert 0 < M <= 3 * H * W
    imp = np.zeros((1, H, W), np.float32)
    imp[0, 11, 15] = 1
    r, _ = oracle_c.gaussian_filter_d2(imp, H, W, 1.0)
    assert r[0, 11, 15] == r.max() and r[0, 0, 0] == 0 and (r >= 0).all()                 # a bump around the impulse, compact support
    wide, _ = oracle_c.gaussian_filter_d2(imp, H, W, 3.0)
    assert (wide > 0).sum() > (r > 0).sum()                                               # larger sigma, wider support
    # channels are filtered independently
    two, _ = oracle_c.gaussian_filter_d2(np.concatenate([a, b]), H, W, 1.0)
    assert np.array_equal(two[0], fa[0]) and np.array_equal(two[1], fb[0])


def test_crf_oracle_mean_field_properties(oracle_c):
    from oracle import crf_oracle
    rng = np.random.default_rng(1)
    H, W, C = 24, 36, 4
    img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    p = rng.random((C, H, W)).astype(np.float32) + 0.05
    p /= p.sum(0, keepdims=True)
    q0 = crf_oracle.dense_crf(img, p, iter_max=0)
    np.testing.assert_allclose(q0, p, rtol=1e-5, atol=1e-6)                               # no step: softmax(log p) = p
    q1 = crf_oracle.crf_inference_infv2(img, p)
    np.testing.assert_allclose(q1.sum(0), 1.0, rtol=1e-5)
    # zero pairwise weights: the update is the unary softmax again
    np.testing.assert_allclose(crf_oracle.dense_crf(img, p, iter_max=1, pos_w=0, bi_w=0), p, rtol=1e-5, atol=1e-6)
    # a constant image: the bilateral kernel sees positions only; the step smooths, so a one-pixel label flip in a constant field is removed
    flat = np.full((H, W, 3), 128, np.uint8)
    pf = np.full((C, H, W), 0.1 / (C - 1), np.float32)
    pf[0] = 0.9
    pf[:, 12, 18] = 0.02
    pf[0, 12, 18], pf[1, 12, 18] = 0.42, 0.54
    pf /= pf.sum(0, keepdims=True)
    assert pf[:, 12, 18].argmax() == 1
    assert crf_oracle.crf_inference_infv2(flat, pf)[:, 12, 18].argmax() == 0
