"""CPU suite, evaluation path (SURVEY f-1): the oracle's restatement against vectors produced by the reference itself
(tests/golden/eval.npz, oracle/gen_golden.py:gen_eval), and the host-side score arithmetic of the product."""
import numpy as np
import torch

SCALES = [1.0, 0.5, 1.5, 0.75, 1.25]


def _eval_inputs():
    from oracle.gen_golden import eval_inputs
    return eval_inputs(np.random.default_rng(171))


def test_camsegv3_oracle_vs_reference(golden):
    from oracle import torch_oracle as to
    from oracle.gen_golden import _StubModel
    g = golden("eval")
    cam, aux, seg, cf, ca = to.multi_scale_camsegv3(_StubModel(int(g["v3_C"])), torch.from_numpy(g["v3_imgs"]), SCALES, getcls=True)
    np.testing.assert_allclose(cam.numpy(), g["v3_cam"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(aux.numpy(), g["v3_cam_aux"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(seg.numpy(), g["v3_seg"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cf.numpy(), g["v3_cls_f"], rtol=1e-6)
    np.testing.assert_allclose(ca.numpy(), g["v3_cls_a"], rtol=1e-6)


def test_eval_label_maps_oracle_bit_exact_vs_reference(oracle_c, golden):
    """F.interpolate to the ground truth's size + cam_to_label / seg_validation / argmax: every label identical"""
    from oracle.gen_golden import EVAL_SIZES
    g = golden("eval")
    cam, seg, cls = _eval_inputs()
    for i, (H, W) in enumerate(EVAL_SIZES):
        a, b, c = oracle_c.eval_labels(cam[0].numpy(), seg[0].numpy(), cls[0].numpy(), H, W, 0.5)
        assert np.array_equal(a, g[f"lab_cam_{i}"][0]) and np.array_equal(b, g[f"lab_ps_{i}"][0]) and np.array_equal(c, g[f"lab_vd_{i}"][0])
        assert len(np.unique(a)) >= 3 and len(np.unique(c)) >= 2            # non-trivial maps
        # the resize itself is bit-identical to ATen's at these sizes (spec R)
        ref = torch.nn.functional.interpolate(cam, size=(H, W), mode="bilinear", align_corners=False)[0].numpy()
        assert np.array_equal(oracle_c.resize_bilinear(cam[0].numpy(), H, W), ref)


def test_cam_to_label_oracle_vs_reference(oracle_c, golden):
    g = golden("eval")
    vc, lab = oracle_c.cam_to_label(g["box_cam"], g["box_cls"], g["box_boxes"], 0.5, 0.7, 0.25, True, 255)
    assert np.array_equal(lab, g["box_label"]) and np.array_equal(vc, g["box_valid_cam"])
    assert set(np.unique(lab)) <= {0, 1, 2, 4, 255} and (lab == 255).any() and (lab == 0).any()


def test_scores_oracle_and_product_arithmetic_vs_reference(oracle_c, golden):
    from cosa_amd.utils import evaluation as ev
    g = golden("eval")
    nc = int(g["sc_nc"])
    gts, pr, pp = ([g[f"sc_{k}_{i}"] for i in range(3)] for k in ("gt", "pred", "ppred"))
    for tag, preds, pseudo in (("sc", pr, False), ("ps", pp, True)):
        hist = oracle_c.confusion(gts, preds, nc, pseudo)
        for fn in (oracle_c.scores_from_hist, ev.scores_from_hist):          # oracle and the product's host arithmetic
            s = fn(hist)
            assert s["pAcc"] == g[f"{tag}_pAcc"] and s["mAcc"] == g[f"{tag}_mAcc"] and s["miou"] == g[f"{tag}_miou"]
            assert np.array_equal(np.array([s["iou"][i] for i in range(nc)]), g[f"{tag}_iou"])


def test_average_precision_vs_reference(oracle_c, golden):
    from cosa_amd.utils import torch_helper as th
    g = golden("eval")
    y, p = g["ap_labels"], g["ap_scores"]
    ora = [oracle_c.average_precision(a, b) for a, b in zip(y, p) if a.sum() > 0]
    np.testing.assert_allclose(ora, g["ap"], rtol=1e-12)
    np.testing.assert_allclose(th.compute_mAP(torch.from_numpy(y), torch.from_numpy(p)), g["ap"], rtol=1e-12)
    assert len(g["ap"]) == 5                                                # the all-negative sample is skipped


def test_format_tabs_means():
    from cosa_amd.utils import torch_helper as th
    sc = [{"iou": {0: 0.5, 1: 0.25}}, {"iou": {0: 1.0, 1: 0.0}}]
    txt, last, means = th.format_tabs(sc, ["A", "B"], ["bg", "cat"])
    assert means == [37.5, 50.0] and last == 50.0 and "cat" in txt and "mIoU" in txt
