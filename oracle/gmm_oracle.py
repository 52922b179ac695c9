"""CPU restatement of the adaptive-threshold fit (TEST INFRASTRUCTURE: only tests/, smoke() and bench.py may import it).

Reference: utils/seg_helper.py:924-943 `rungmm` -- a 1-D Gaussian mixture (2 or 3 components) fitted by EM to the queue of
per-cell CAM maxima above `filter_thre`; the thresholds are the largest value assigned to component 0 and the smallest
value assigned to component 2.  The EM itself lives in a third-party dependency that is not part of the reference tree:
scikit-learn (requirements.txt pins 1.2.2; this image has 1.7.2) `sklearn.mixture.GaussianMixture(covariance_type='full',
tol=1e-3, reg_covar=1e-6, max_iter=100, n_init=1)` with weights / means / precisions given, so no k-means and no RNG.
Restated here for one feature (all matrices are 1x1):

    E:  lp_k = -0.5 (log 2pi + ((x - mu_k) pc_k)^2) + log pc_k + log w_k;  lpn = logsumexp_k lp_k;  log r_k = lp_k - lpn
    M:  n_k = sum r_k + 10 eps;  mu_k = sum r_k x / n_k;  var_k = sum r_k (x - mu_k)^2 / n_k + 1e-6;  pc_k = 1/sqrt(var_k)
        w_k = n_k / sum_j n_j                      (1.2.2 divides by the sample count: differs by 30 eps / N)
    stop when |mean(lpn) - previous mean(lpn)| < 1e-3 (the M step of that iteration is kept), at most 100 iterations;
    labels = argmax_k log r_k of one more E step.

Pinned by tests/golden/gmm.npz: thresholds returned by the reference's rungmm (run on this image's scikit-learn) and the
iteration counts of the same estimator.
"""
import numpy as np

LOG_2PI = float(np.log(2 * np.pi))


def _e_step(x, w, mu, pc):
    y = x[:, None] * pc[None, :] - (mu * pc)[None, :]
    lp = -0.5 * (LOG_2PI + y * y) + np.log(pc)[None, :] + np.log(w)[None, :]
    m = lp.max(axis=1)
    lpn = np.log(np.exp(lp - m[:, None]).sum(axis=1)) + m
    return lpn, lp - lpn[:, None]


def fit(x, modal, tol=1e-3, reg_covar=1e-6, max_iter=100):
    """x: 1-D float64 (already filtered).  Returns (labels, n_iter, (w, mu, pc))."""
    x = np.asarray(x, np.float64)
    if modal == 3:
        mu = np.array([x.min(), np.median(x), x.max()])
    elif modal == 2:
        mu = np.array([x.min(), x.max()])
    else:
        raise AssertionError("modal in [2,3]")
    w = np.full(modal, 1.0 / modal)
    pc = np.ones(modal)
    lb = -np.inf
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        prev = lb
        lpn, log_r = _e_step(x, w, mu, pc)
        r = np.exp(log_r)
        nk = r.sum(axis=0) + 10 * np.finfo(np.float64).eps
        mu = (r * x[:, None]).sum(axis=0) / nk
        d = x[:, None] - mu[None, :]
        var = (r * d * d).sum(axis=0) / nk + reg_covar
        w = nk / nk.sum()
        pc = 1.0 / np.sqrt(var)
        lb = lpn.mean()
        if abs(lb - prev) < tol:
            break
    _, log_r = _e_step(x, w, mu, pc)
    return log_r.argmax(axis=1), n_iter, (w, mu, pc)


def rungmm(queue, modal, filter_thre=0.05):
    """utils/seg_helper.py:924-943.  Returns max(component 0) for modal=2, (max(component 0), min(component 2)) for modal=3;
    an empty component raises ValueError as the reference's max()/min() of an empty sequence does."""
    q = np.asarray(queue, np.float64).flatten()
    q = q[q > filter_thre]
    labels, _, _ = fit(q, modal)
    lo = q[labels == 0]
    if lo.size == 0:
        raise ValueError("rungmm: no sample assigned to component 0")
    if modal == 2:
        return float(lo.max())
    hi = q[labels == 2]
    if hi.size == 0:
        raise ValueError("rungmm: no sample assigned to component 2")
    return float(lo.max()), float(hi.min())
