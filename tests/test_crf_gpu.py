"""GPU checks of the dense-CRF post-processing (SURVEY f-4; utils/seg_helper.py:961-996) against oracle/crf_oracle.py.

PARITY UNPINNED with respect to the reference: the reference delegates to pydensecrf, which is neither under the reference tree nor in this
image, and holds no fixture of its output.  What IS pinned: the lattice filter both sides stand on (bit-exact against the reference's own
C++ at d = 5, tests/test_label_gpu.py / tests/golden/bilateral.npz); these tests tie the device implementation to the CPU restatement of the
published mean-field algorithm."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,K,sxy", [(23, 31, 1, 1.0), (60, 84, 6, 1.0), (64, 64, 21, 3.0), (375, 500, 3, 1.0)])
def test_d2_lattice_filter_vs_oracle(oracle_c, H, W, K, sxy):
    """the 2-D lattice build of csrc/permuto_kernels.hip against the 2-D build of the C oracle (the same pair that is bit-identical at d = 5)"""
    from cosa_amd.utils.seg_helper import DenseCRF
    rng = np.random.default_rng(H)
    v = rng.random((K, H, W)).astype(np.float32)
    ref, _ = oracle_c.gaussian_filter_d2(v, H, W, sxy)
    got = DenseCRF._filter_gauss(torch.from_numpy(v).cuda(), sxy).cpu().numpy()
    assert np.array_equal(got, ref), np.abs(got - ref).max()


@pytest.mark.parametrize("H,W,C", [(60, 84, 6), (125, 167, 21)])
def test_dense_crf_vs_oracle(H, W, C):
    from cosa_amd.utils import seg_helper
    from oracle import crf_oracle
    rng = np.random.default_rng(C)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([128 + 100 * np.sin(xx / 17.0), 128 + 90 * np.cos(yy / 11.0), 40 + 0.9 * xx], -1) + rng.normal(0, 3, (H, W, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    p = rng.random((C, H, W)).astype(np.float32) ** 3 + 0.01
    p /= p.sum(0, keepdims=True)
    ref = crf_oracle.crf_inference_infv2(img, p)
    got = seg_helper.crf_inference_infv2(img, p)                                          # numpy in -> numpy out, as the reference's call
    assert isinstance(got, np.ndarray) and got.shape == (C, H, W) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6)
    assert (got.argmax(0) == ref.argmax(0)).mean() >= 0.9995
    assert (ref.argmax(0) != p.argmax(0)).mean() > 0.01                                   # (the step does change labels on this input)
    on_dev = seg_helper.crf_inference_infv2(torch.from_numpy(img).cuda(), torch.from_numpy(p).cuda())
    assert torch.is_tensor(on_dev) and on_dev.is_cuda and np.array_equal(on_dev.cpu().numpy(), got)
    three = seg_helper.DenseCRF(iter_max=3, pos_w=3, pos_xy_std=3, bi_w=5, bi_xy_std=40, bi_rgb_std=7)(img, p)
    np.testing.assert_allclose(three, crf_oracle.dense_crf(img, p, 3, 3, 3, 5, 40, 7), rtol=1e-3, atol=1e-5)


def test_evaluate_getcrf_row_vs_oracle_composition(oracle_c):
    """evaluate(getcrf=True) (what finaleval runs, main.py:414-425): the extra row `Seg_crf` equals the score of the oracle's CRF applied to
    the engine's own network outputs; Seg_vd is returned as the reference does (the row before the last)"""
    from cosa_amd import evaluation_engine as ee
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    from cosa_amd.utils import seg_helper, torch_helper
    from oracle import crf_oracle
    torch.manual_seed(2)
    C, S = 4, 64
    args = default_args("VOC12", crop_size=S, batch_size=1)
    args.num_classes, args.bkg_thre = C + 1, 0.5
    model = build_model(args).cuda().eval()
    rng = np.random.default_rng(4)
    loader = []
    for (H, W) in [(50, 70), (64, 64), (81, 47)]:
        img = torch.from_numpy(rng.standard_normal((1, 3, H, W)).astype(np.float32))
        lab = torch.from_numpy(rng.integers(0, C + 1, (1, H, W)).astype(np.int64))
        cls = torch.zeros(1, C)
        cls[0, rng.choice(C, 2, replace=False)] = 1
        loader.append(("img", img, lab, cls))
    tab, seg_miou, df, _ = ee.evaluate(model, loader, args, epoch='best1', isfinal=True, getcrf=True)
    assert df["Metrics"] == ["Seg_vd", "Seg_crf"] and seg_miou == df["mIoU"][0] and "Seg_crf" in tab
    model.batch_invariant_heads = model.decoder.batch_invariant = True
    hist = np.zeros((C + 1, C + 1), np.int64)
    agree = []
    with torch.no_grad():
        for _, img, lab, cls in loader:
            x = torch.nn.functional.interpolate(img.cuda(), size=[S, S], mode="bilinear", align_corners=False)
            _, _, seg, _, _ = seg_helper.multi_scale_camsegv3(model, x, ee.EVAL_SCALES, getcls=True)
            H, W = lab.shape[1:]
            rs = torch.nn.functional.interpolate(seg, size=(H, W), mode="bilinear", align_corners=False)
            vd = seg_helper.seg_validation(rs, cls.cuda()).softmax(dim=1)[0].cpu().numpy()
            ori = torch_helper.denormalize_img_(img.cuda())[0].permute(1, 2, 0).cpu().numpy()
            q = crf_oracle.crf_inference_infv2(ori, vd)
            hist += oracle_c.confusion([lab[0].numpy().astype(np.uint8)], [q.argmax(0).astype(np.uint8)], C + 1)
    model.batch_invariant_heads = model.decoder.batch_invariant = False
    ref = np.round(np.array(list(oracle_c.scores_from_hist(hist)["iou"].values())) * 100, 2).mean()
    assert abs(df["mIoU"][1] - ref) < 0.05, (df["mIoU"][1], ref)          # (a handful of argmax ties between the two float paths at most)
