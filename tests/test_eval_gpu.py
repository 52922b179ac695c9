"""GPU parity of the evaluation path (SURVEY f-1): HIP kernels behind cosa_eval_labels / cosa_cam_to_label / cosa_confusion_hist and
the device-resident evaluation engine, against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SCALES = [1.0, 0.5, 1.5, 0.75, 1.25]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_eval_label_maps_bit_exact_vs_reference_golden(golden):
    from cosa_amd.utils import seg_helper
    from oracle.gen_golden import EVAL_SIZES, eval_inputs
    g = golden("eval")
    cam, seg, cls = eval_inputs(np.random.default_rng(171))
    for i, (H, W) in enumerate(EVAL_SIZES):
        a, b, c = seg_helper.eval_label_maps(cam.cuda(), seg.cuda(), cls.cuda(), (H, W), 0.5)
        assert a.dtype == torch.uint8 and a.shape == (1, H, W)
        assert np.array_equal(a.cpu().numpy(), g[f"lab_cam_{i}"]) and np.array_equal(b.cpu().numpy(), g[f"lab_ps_{i}"])
        assert np.array_equal(c.cpu().numpy(), g[f"lab_vd_{i}"])
        only_cam, none1, none2 = seg_helper.eval_label_maps(cam.cuda(), None, cls.cuda(), (H, W), 0.5)
        assert torch.equal(only_cam, a) and none1 is None and none2 is None


@pytest.mark.parametrize("B,C,S,H,W", [(2, 20, 448, 375, 500), (1, 80, 448, 480, 640), (3, 4, 64, 64, 64), (1, 2, 32, 7, 301)])
def test_eval_label_maps_vs_oracle_random(oracle_c, B, C, S, H, W):
    """noise instead of smooth fields (argmax flips between neighbouring pixels), several classes absent, batch > 1"""
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(B * 1000 + C)
    cam = rng.random((B, C, S, S), dtype=np.float32)
    seg = rng.standard_normal((B, C + 1, S, S)).astype(np.float32)
    cls = (rng.random((B, C)) < 0.3).astype(np.float32)
    cls[:, 0] = 1
    a, b, c = seg_helper.eval_label_maps(dev(cam), dev(seg), dev(cls), (H, W), 0.5)
    for i in range(B):
        ra, rb, rc = oracle_c.eval_labels(cam[i], seg[i], cls[i], H, W, 0.5)
        assert np.array_equal(a[i].cpu().numpy(), ra) and np.array_equal(b[i].cpu().numpy(), rb) and np.array_equal(c[i].cpu().numpy(), rc)
    present = np.concatenate([np.ones((B, 1)), cls], 1)
    vd = c.cpu().numpy()
    for i in range(B):
        assert present[i][np.unique(vd[i])].all()                # validated predictions only name present classes


def test_cam_to_label_vs_reference_golden_and_oracle(oracle_c, golden):
    from cosa_amd.utils import seg_helper
    g = golden("eval")
    vc, lab = seg_helper.cam_to_label(dev(g["box_cam"]), dev(g["box_cls"]), img_box=torch.from_numpy(g["box_boxes"]), bkg_thre=0.5,
                                      high_thre=0.7, low_thre=0.25, ignore_mid=True, ignore_index=255)
    assert lab.dtype == torch.int64
    assert np.array_equal(lab.cpu().numpy(), g["box_label"]) and np.array_equal(vc.cpu().numpy(), g["box_valid_cam"])
    # img_box None -> label map only; cls_label None -> raw CAM
    rng = np.random.default_rng(5)
    cam = rng.random((2, 7, 33, 47), dtype=np.float32)
    cls = (rng.random((2, 7)) < 0.5).astype(np.float32)
    for cl in (cls, None):
        out = seg_helper.cam_to_label(dev(cam), dev(cl) if cl is not None else None, bkg_thre=0.6)
        assert np.array_equal(out.cpu().numpy(), oracle_c.cam_to_label(cam, cl, None, 0.6))
    with pytest.raises(TypeError):
        seg_helper.cam_to_label(dev(cam), dev(cls))                          # bkg_thre is mandatory, as in the reference


def test_seg_validation_matches_reference_semantics():
    from cosa_amd.utils import seg_helper
    seg = torch.randn(2, 4, 5, 6, device="cuda")
    cls = torch.tensor([[1.0, 0.0, 1.0], [0.0, 0.0, 0.0]], device="cuda")
    out = seg_helper.seg_validation(seg, cls)
    assert torch.equal(out[0, 0], seg[0, 0]) and torch.equal(out[0, 1], seg[0, 1]) and torch.all(out[0, 2] == -1e5)
    assert torch.all(out[1, 1:] == -1e5) and torch.equal(out[1, 0], seg[1, 0])
    assert seg_helper.seg_validation(seg, None) is seg


def test_scores_vs_reference_golden(golden):
    from cosa_amd.utils import evaluation as ev
    g = golden("eval")
    nc = int(g["sc_nc"])
    gts, pr, pp = ([g[f"sc_{k}_{i}"] for i in range(3)] for k in ("gt", "pred", "ppred"))
    pp_before = [p.copy() for p in pp]
    for tag, s in (("sc", ev.scores(gts, pr, nc)), ("ps", ev.pseudo_scores(gts, pp, nc))):
        assert s["pAcc"] == g[f"{tag}_pAcc"] and s["mAcc"] == g[f"{tag}_mAcc"] and s["miou"] == g[f"{tag}_miou"]
        assert np.array_equal(np.array([s["iou"][i] for i in range(nc)]), g[f"{tag}_iou"])
    assert all(np.array_equal(a, b) for a, b in zip(pp, pp_before))          # inputs untouched


@pytest.mark.parametrize("nc,n", [(21, 1_000_003), (81, 3_500_000), (2, 17), (90, 4096)])
def test_confusion_hist_vs_oracle(oracle_c, nc, n):
    from cosa_amd.utils import evaluation as ev
    rng = np.random.default_rng(nc)
    gt = rng.integers(0, nc, n).astype(np.uint8)
    gt[rng.random(n) < 0.07] = 255
    pr = rng.integers(0, nc, n).astype(np.uint8)
    m = ev.ConfusionMeter(nc)
    m.update(dev(gt)[: n // 2], dev(pr)[: n // 2])                           # accumulates over calls; odd split -> unaligned second half
    m.update(dev(gt)[n // 2:], dev(pr)[n // 2:])
    assert np.array_equal(m.hist.cpu().numpy(), oracle_c.confusion([gt], [pr], nc))
    pr2 = pr.copy()
    pr2[rng.random(n) < 0.2] = 255
    m2 = ev.ConfusionMeter(nc, pseudo=True)
    m2.update(gt, pr2)                                                        # numpy inputs are uploaded
    assert np.array_equal(m2.hist.cpu().numpy(), oracle_c.confusion([gt], [pr2], nc, True))
    assert int(m2.hist.sum()) == int(((gt < nc) & (pr2 != 255)).sum())


def test_camsegv3_vs_reference_golden(golden):
    from cosa_amd.utils import seg_helper
    from oracle.gen_golden import _StubModel
    g = golden("eval")
    stub = _StubModel(int(g["v3_C"]))

    def model(x, cam_only=False):
        return tuple(o.cuda() if o is not None else None for o in stub(x.cpu()))

    cam, aux, seg, cf, ca = seg_helper.multi_scale_camsegv3(model, dev(g["v3_imgs"]), SCALES, getcls=True)
    np.testing.assert_allclose(cam.cpu().numpy(), g["v3_cam"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(aux.cpu().numpy(), g["v3_cam_aux"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(seg.cpu().numpy(), g["v3_seg"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cf.cpu().numpy(), g["v3_cls_f"], rtol=1e-5)
    np.testing.assert_allclose(ca.cpu().numpy(), g["v3_cls_a"], rtol=1e-5)
    three = seg_helper.multi_scale_camsegv3(model, dev(g["v3_imgs"]), SCALES)
    assert len(three) == 3 and torch.equal(three[0], cam)


def test_average_precision_on_device_vs_reference_golden(golden):
    from cosa_amd.utils import torch_helper as th
    g = golden("eval")
    np.testing.assert_allclose(th.compute_mAP(dev(g["ap_labels"]), dev(g["ap_scores"])), g["ap"], rtol=1e-12)


def test_evaluate_end_to_end_vs_oracle_composition(oracle_c):
    """the engine (network passes, label kernels, device confusion matrices, AP accumulation, table) against the same quantities
    rebuilt from the engine's own network outputs with the CPU oracle's label / histogram / score functions"""
    from cosa_amd import evaluation_engine as ee
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    from cosa_amd.utils import seg_helper
    torch.manual_seed(0)
    C, S = 4, 64
    args = default_args("VOC12", crop_size=S, batch_size=1)
    args.num_classes, args.bkg_thre = C + 1, 0.5
    model = build_model(args).cuda().eval()
    rng = np.random.default_rng(3)
    loader = []
    for (H, W) in [(50, 70), (64, 64), (81, 47), (33, 90), (64, 64), (70, 50)]:      # > 2 images: the captured-graph path replays
        img = torch.from_numpy(rng.standard_normal((1, 3, H, W)).astype(np.float32))
        lab = torch.from_numpy(rng.integers(0, C + 1, (1, H, W)).astype(np.int64))
        lab[0, :3] = 255
        cls = torch.zeros(1, C)
        cls[0, rng.choice(C, 2, replace=False)] = 1
        loader.append(("img", img, lab, cls))
    tab, seg_miou, cam_miou, df, cls_aps = ee.evaluate(model, loader, args, epoch=7, s_or_t='s', get_camiou=True)
    # oracle composition from the same network outputs (evaluate() runs the narrow heads on the batch-invariant kernel: do the same here)
    model.batch_invariant_heads = model.decoder.batch_invariant = True
    hist = {k: np.zeros((C + 1, C + 1), np.int64) for k in ("cam", "aux", "vd")}
    aps = [[], []]
    with torch.no_grad():
        for _, img, lab, cls in loader:
            x = torch.nn.functional.interpolate(img.cuda(), size=[S, S], mode="bilinear", align_corners=False)
            cam, aux, seg, cf, ca = seg_helper.multi_scale_camsegv3(model, x, ee.EVAL_SCALES, getcls=True)
            H, W = lab.shape[1:]
            a, _, c = oracle_c.eval_labels(cam[0].cpu().numpy(), seg[0].cpu().numpy(), cls[0].numpy(), H, W, 0.5)
            a2, _, _ = oracle_c.eval_labels(aux[0].cpu().numpy(), seg[0].cpu().numpy(), cls[0].numpy(), H, W, 0.5)
            gt = lab[0].numpy().astype(np.uint8)
            for k, p in (("cam", a), ("aux", a2), ("vd", c)):
                hist[k] += oracle_c.confusion([gt], [p], C + 1)
            for j, lg in enumerate((cf, ca)):
                aps[j].append(oracle_c.average_precision(cls[0].numpy(), torch.sigmoid(lg[0].float()).cpu().numpy()))
    model.batch_invariant_heads = model.decoder.batch_invariant = False
    ref = [oracle_c.scores_from_hist(hist[k]) for k in ("cam", "aux", "vd")]
    ref_miou = [np.round(np.array(list(r["iou"].values())) * 100, 2).mean() for r in ref]
    assert abs(cam_miou - ref_miou[0]) < 1e-9 and abs(seg_miou - ref_miou[2]) < 1e-9
    assert df["Metrics"] == ["CAM", "aux_CAM", "Seg_vd"] and df["Iterations"] == [7, 7, 7] and df["ST"] == ["s"] * 3
    np.testing.assert_allclose(df["mIoU"], ref_miou, atol=1e-9)
    np.testing.assert_allclose(cls_aps, [np.mean(aps[0]), np.mean(aps[1])], rtol=1e-6)
    assert "mIoU" in tab and model.training is False
    tab2, seg_miou2, cam_miou2, _, aps2 = ee.evaluate(model, loader, args, epoch=7, s_or_t='s', get_camiou=True, use_graph=False)
    assert abs(seg_miou2 - seg_miou) < 1e-9 and abs(cam_miou2 - cam_miou) < 1e-9 and np.allclose(aps2, cls_aps, rtol=1e-9)
    # grouping loader items into one multi-scale pass: every kernel on the CAM / seg path is batch-invariant per element (the narrow
    # heads and the classification logits included), so the score table and the AP are identical
    tab3, seg_miou3, cam_miou3, df3, aps3 = ee.evaluate(model, loader, args, epoch=7, s_or_t='s', get_camiou=True, eval_group=3)
    assert tab3 == tab and seg_miou3 == seg_miou and cam_miou3 == cam_miou and df3["mIoU"] == df["mIoU"]
    assert np.allclose(aps3, cls_aps, rtol=1e-12)
    with pytest.raises(NotImplementedError):
        ee.evaluate(model, loader, args, epoch=1, save_result=True)


def test_evaluate_threshold_filters_sweep_vs_oracle(oracle_c):
    """evaluation_engine.py:43-50,132-152,252-262: per threshold t the pseudo-label maps cam2mask(valid CAM, 1 - t, t) of both CAMs, scored
    with pseudo_scores, as rows cam_<t> / camaux_<t> after the first three rows.  Square ground truths go through the fused HIP cam2mask
    and are compared with the C oracle's cam2mask + pseudo confusion on the engine's own network outputs; a non-square item exercises the
    per-image path."""
    from cosa_amd import evaluation_engine as ee
    from cosa_amd.models import build_model
    from cosa_amd.train_step import default_args
    from cosa_amd.utils import seg_helper
    torch.manual_seed(1)
    C, S = 4, 64
    args = default_args("VOC12", crop_size=S, batch_size=1)
    args.num_classes, args.bkg_thre = C + 1, 0.5
    model = build_model(args).cuda().eval()
    rng = np.random.default_rng(9)
    loader = []
    for _ in range(4):
        img = torch.from_numpy(rng.standard_normal((1, 3, S, S)).astype(np.float32))
        lab = torch.from_numpy(rng.integers(0, C + 1, (1, S, S)).astype(np.int64))
        cls = torch.zeros(1, C)
        cls[0, rng.choice(C, 2, replace=False)] = 1
        loader.append(("img", img, lab, cls))
    thr = [0.2, 0.35]
    tab, seg_miou, cam_miou, df, _ = ee.evaluate(model, loader, args, epoch=3, s_or_t='t', get_camiou=True, threshold_filters=thr)
    assert df["Metrics"] == ["CAM", "aux_CAM", "Seg_vd", "cam_0.2", "cam_0.35", "camaux_0.2", "camaux_0.35"]
    assert np.array_equal([seg_miou], [df["mIoU"][-1]], equal_nan=True)        # the reference returns the LAST row here (:289)
    model.batch_invariant_heads = model.decoder.batch_invariant = True
    hist = {k: np.zeros((C + 1, C + 1), np.int64) for k in df["Metrics"][3:]}
    box = np.array([[0, S - 1, 0, S - 1]], np.int32)                           # [0, -1, 0, -1] as slice bounds
    with torch.no_grad():
        for _, img, lab, cls in loader:
            x = torch.nn.functional.interpolate(img.cuda(), size=[S, S], mode="bilinear", align_corners=False)
            cam, aux, _, _, _ = seg_helper.multi_scale_camsegv3(model, x, ee.EVAL_SCALES, getcls=True)
            gt = lab[0].numpy().astype(np.uint8)
            for t in thr:
                for key, c in ((f"cam_{t}", cam), (f"camaux_{t}", aux)):
                    m = oracle_c.cam2mask(None, box, c.cpu().numpy(), cls.numpy(), 1 - t, t, 2, par=None)
                    hist[key] += oracle_c.confusion([gt], [m[0].astype(np.uint8)], C + 1, True)
    model.batch_invariant_heads = model.decoder.batch_invariant = False
    ref_miou = [np.round(np.array(list(oracle_c.scores_from_hist(hist[k])["iou"].values())) * 100, 2).mean() for k in df["Metrics"][3:]]
    np.testing.assert_allclose(df["mIoU"][3:], ref_miou, atol=1e-9)
    # a non-square ground truth: the per-image path (the reference's own loop on the device)
    img = torch.from_numpy(rng.standard_normal((1, 3, 50, 70)).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, C + 1, (1, 50, 70)).astype(np.int64))
    tab2, _, _, df2, _ = ee.evaluate(model, [("img", img, lab, loader[0][3])], args, epoch=3, get_camiou=True, threshold_filters=[0.3], isfinal=True)
    assert df2["Metrics"] == ["Seg_vd", "cam_0.3", "camaux_0.3"] and all(np.isfinite(df2["mIoU"]))
