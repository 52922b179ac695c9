"""GPU parity: the adaptive-threshold fit (gmm_kernels.hip, SURVEY f-4) vs the reference's rungmm goldens and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def gmm_case(g, n):
    q = np.concatenate([g[f"{n}_queue_rand"], g[f"{n}_queue32"].astype(np.float64)], 0)
    return q, int(g[f"{n}_modal"]), float(g[f"{n}_filter"])


@pytest.mark.parametrize("case", list("abcdef"))
def test_rungmm_vs_reference_golden(golden, case):
    """thresholds are samples of the queue: they must be the reference's bit for bit; iteration count and means as scikit-learn's"""
    from cosa_amd.utils import seg_helper
    g = golden("gmm")
    q, modal, thr = gmm_case(g, case)
    res = np.atleast_1d(np.array(seg_helper.rungmm(torch.from_numpy(q).cuda(), modal, thr)))
    assert np.array_equal(res, g[f"{case}_thresholds"])
    out = seg_helper.rungmm_device(torch.from_numpy(q).cuda(), modal, thr).cpu().numpy()
    assert int(out[2]) == int(g[f"{case}_n_iter"]) and int(out[3]) == 0
    np.testing.assert_allclose(out[4:4 + modal], g[f"{case}_means"], rtol=1e-11)
    assert np.array_equal(np.atleast_1d(seg_helper.rungmm(q, modal, thr)), res)          # numpy queue, as the reference passes it


def test_rungmm_training_size_vs_oracle_and_deterministic():
    """the training loop's queue (batch 16 x ratio 100 rows x 28^2 cells = 1.25 M samples): thresholds equal the CPU oracle's,
    parameters agree to rounding, and two runs are bit-identical (fixed-order reductions, no float atomics)."""
    from oracle import gmm_oracle
    from oracle.gen_golden import gmm_queue
    from cosa_amd.utils import seg_helper
    q = gmm_queue(np.random.default_rng(5), 1600, 784, 300)
    qd = torch.from_numpy(q).cuda()
    a = seg_helper.rungmm_device(qd, 3, 0.05)
    b = seg_helper.rungmm_device(qd, 3, 0.05)
    assert torch.equal(a, b)
    x = q.flatten()
    x = x[x > 0.05]
    labels, n_iter, (w, mu, pc) = gmm_oracle.fit(x, 3)
    out = a.cpu().numpy()
    assert int(out[3]) == 0 and int(out[2]) == n_iter
    assert out[0] == x[labels == 0].max() and out[1] == x[labels == 2].min()
    np.testing.assert_allclose(out[4:7], mu, rtol=1e-10)
    np.testing.assert_allclose(out[7:10], w, rtol=1e-10)
    np.testing.assert_allclose(out[10:13], pc, rtol=1e-10)


def test_rungmm_slow_convergence_and_iteration_cap():
    """a tighter tolerance makes EM run long: same iteration count as the oracle, and max_iter caps it"""
    from oracle import gmm_oracle
    from oracle.gen_golden import gmm_queue
    from cosa_amd.utils import seg_helper
    q = gmm_queue(np.random.default_rng(6), 64, 196, 8)
    x = q.flatten()
    x = x[x > 0.05]
    labels, n_iter, (w, mu, pc) = gmm_oracle.fit(x, 3, tol=1e-7, max_iter=500)
    assert 10 < n_iter < 500
    out = seg_helper.rungmm_device(torch.from_numpy(q).cuda(), 3, 0.05, tol=1e-7, max_iter=500).cpu().numpy()
    assert int(out[2]) == n_iter and int(out[3]) == 0
    assert out[0] == x[labels == 0].max() and out[1] == x[labels == 2].min()
    np.testing.assert_allclose(out[4:7], mu, rtol=1e-9)
    capped = seg_helper.rungmm_device(torch.from_numpy(q).cuda(), 3, 0.05, tol=1e-7, max_iter=7).cpu().numpy()
    assert int(capped[2]) == 7
    l7, n7, _ = gmm_oracle.fit(x, 3, tol=1e-7, max_iter=7)
    assert n7 == 7 and capped[0] == x[l7 == 0].max() and capped[1] == x[l7 == 2].min()


def test_rungmm_error_behaviour():
    from cosa_amd.utils import seg_helper
    with pytest.raises(ValueError):                                 # everything in one component: the reference's min([]) raises
        seg_helper.rungmm(torch.full((4, 8), 0.5, device="cuda", dtype=torch.float64), 3)
    with pytest.raises(ValueError):                                 # nothing above the filter
        seg_helper.rungmm(torch.zeros(4, 8, device="cuda", dtype=torch.float64), 3)
    with pytest.raises(AssertionError):
        seg_helper.rungmm(torch.rand(4, 8, device="cuda", dtype=torch.float64), 4)
    out = seg_helper.rungmm_device(torch.full((4, 8), 0.5, device="cuda", dtype=torch.float64), 3)
    assert int(out[3].item()) & 2 and torch.isnan(out[1])


def test_cell_bilinear_is_atens_downsample():
    from cosa_amd.utils import seg_helper
    x = torch.rand(3, 5, 448, 448, device="cuda") * 3 - 0.5
    for g in (28, 14, 56, 30):                                       # 30: not a divisor -> the ATen fallback itself
        ref = torch.nn.functional.interpolate(x, size=(g, g), mode="bilinear", align_corners=False)
        assert torch.equal(seg_helper.cell_bilinear(x, g), ref)
    y = torch.rand(2, 3, 224, 224, device="cuda")
    assert torch.equal(seg_helper.cell_bilinear(y, 14), torch.nn.functional.interpolate(y, size=(14, 14), mode="bilinear"))


def test_tracker_and_device_thresholds_in_cam2mask():
    """EMAtracker on device scalars == the reference's float arithmetic; cam2mask reading thresholds from the device gives
    the label map of the same thresholds passed as host floats; a non-finite fit leaves the tracker unchanged."""
    from cosa_amd.utils import seg_helper, torch_helper
    t_host, t_dev = torch_helper.EMAtracker(0.7, decay=0.99), torch_helper.EMAtracker(0.7, decay=0.99)
    for v in (0.71234, 0.69, 0.7031):
        t_host.update(v)
        t_dev.update(torch.tensor(v, device="cuda", dtype=torch.float64))
    assert t_dev.get().item() == t_host.get()
    t_dev.update(torch.tensor(float("nan"), device="cuda", dtype=torch.float64))
    assert t_dev.get().item() == t_host.get()
    B, C, S = 4, 20, 224
    g = torch.Generator().manual_seed(3)
    cams = torch.nn.functional.interpolate(torch.rand(B, C, S // 8, S // 8, generator=g), size=(S, S), mode="bilinear").cuda()
    labels = torch.zeros(B, C)
    for b in range(B):
        labels[b, torch.randperm(C, generator=g)[: 1 + b]] = 1
    labels = labels.cuda()
    boxes = torch.tensor([[0, S, 0, S]] * B)
    img = torch.zeros(B, 3, S, S, device="cuda")
    hi, lo = t_host.get(), 0.2713
    m_host = seg_helper.cam2mask(img, boxes, cams, labels, hi, lo, _fold_validation=True)
    m_dev = seg_helper.cam2mask(img, boxes, cams, labels, t_dev.get(), torch.tensor(lo, device="cuda", dtype=torch.float64),
                                _fold_validation=True)
    assert torch.equal(m_host, m_dev)
    assert (m_host == 255).any() and (m_host == 0).any()


def test_training_step_with_adaptive_thresholds():
    """a few steps with usegmm: the trackers follow the oracle's fit of the SAME queue contents (thresholds are queue samples:
    exact), and the step still produces finite losses"""
    from oracle import gmm_oracle
    from cosa_amd.train_step import CoSATrainer, default_args
    torch.manual_seed(0)
    np.random.seed(0)
    args = default_args("VOC12", batch_size=2, crop_size=224, usegmm=True, queue_update_ratio=4, teacher_async=False)
    tr = CoSATrainer(args, torch.device("cuda:0"))
    g = torch.Generator().manual_seed(1)
    lab = torch.zeros(2, 20)
    lab[0, 3] = 1
    lab[1, [5, 11]] = 1
    lo, hi = args.low_thre, args.high_thre
    for it in range(3):
        wimg = torch.randn(2, 3, 224, 224, generator=g).cuda()
        simg = torch.randn(2, 3, 224, 224, generator=g).cuda()
        out = tr.step(wimg, simg, lab.cuda(), torch.tensor([[0, 224, 0, 224]] * 2), n_iter=args.warmup_iters + 1 + it)
        q = tr.cam_queue.getqueue().cpu().numpy()
        tl, th = gmm_oracle.rungmm(q, 3, args.gmmfilter_thre)
        lo = lo * args.gmmemadecay + tl * (1 - args.gmmemadecay)
        hi = hi * args.gmmemadecay + th * (1 - args.gmmemadecay)
        assert tr.ema_lowthre.get().item() == lo and tr.ema_highthre.get().item() == hi
        assert all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in out.values() if torch.is_tensor(v))
