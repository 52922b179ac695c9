"""CPU suite: the oracle (oracle/) against the golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

DIL = [1, 2, 4, 8, 12, 24]


def test_expf_accuracy(oracle_c):
    xs = np.linspace(-30, 10, 4001).astype(np.float32)
    e = np.array([oracle_c.expf(x) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 2e-7
    assert oracle_c.expf(-100.0) == 0.0 and oracle_c.expf(0.0) == 1.0


def test_par_vs_reference(oracle_c, golden):
    g = golden("par")
    for tag in ("k2", "k4"):
        out = oracle_c.par_forward(g[f"{tag}_img"][0], g[f"{tag}_masks"][0], DIL, 10)
        ref = g[f"{tag}_out"][0]
        assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)) < 2e-5
    for i in range(2):
        out = oracle_c.par_forward(g["b2_img"][i], g["b2_masks"][i], g["b2_dil"], int(g["b2_iter"]))
        np.testing.assert_allclose(out, g["b2_out"][i], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("key,ds,par,thr_hi", [("mask_none", 2, None, 0.7), ("mask_none_ds0", 0, None, 0.7),
                                              ("mask_par", 2, (DIL, 10), 0.7), ("mask_par_coco_thr", 2, (DIL, 10), 0.65)])
def test_cam2mask_bit_exact_vs_reference(oracle_c, golden, key, ds, par, thr_hi):
    g = golden("cam2mask")
    m = oracle_c.cam2mask(g["images"], g["boxes"], g["cams"], g["labels"], thr_hi, 0.25, ds, par=par)
    ref = g[key]
    assert m.shape == ref.shape
    assert set(np.unique(ref)) <= set([0, 1, 2, 3, 4, 5, 6, 255])
    assert np.array_equal(m, ref), f"{(m != ref).sum()} of {m.size} labels differ"


def test_cam2mask_golden_is_nontrivial(golden):
    g = golden("cam2mask")
    for key in ("mask_none", "mask_par"):
        vals, cnt = np.unique(g[key], return_counts=True)
        assert len(vals) >= 5 and (cnt > 50).sum() >= 4      # several classes, bg and ignore all present


def test_bilateral_bit_exact_vs_reference_cpp(oracle_c, golden):
    g = golden("bilateral")
    for tag in ("smooth", "noise", "odd"):
        img, seg, ref = g[f"{tag}_img"], g[f"{tag}_seg"], g[f"{tag}_out"]
        N, K, H, W = seg.shape
        out, M = oracle_c.bilateralfilter_batch(img, seg, N, K, H, W, 15.0, 50.0)
        assert np.array_equal(out, ref)
        assert (M > 0).all()


def test_bilateral_against_compiled_reference_live(oracle_c):
    """oracle/_ref (the reference's own C++) on a fresh input, when it has been built."""
    if oracle_c.ref_lib() is None:
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(5)
    N, K, H, W = 1, 2, 24, 40
    img = rng.uniform(0, 255, (N, 3, H, W)).astype(np.float32)
    seg = rng.uniform(0, 1, (N, K, H, W)).astype(np.float32)
    ref = np.zeros(seg.size, np.float32)
    oracle_c.ref_bilateralfilter_batch(img, seg, ref, N, K, H, W, 15.0, 50.0)
    out, _ = oracle_c.bilateralfilter_batch(img, seg, N, K, H, W, 15.0, 50.0)
    assert np.array_equal(out.reshape(-1), ref)


def test_dense_energy_loss_and_grad_vs_reference(golden):
    from oracle import torch_oracle as to
    g = golden("bilateral")
    logits = torch.from_numpy(g["del_logits"]).requires_grad_(True)
    prob = logits.softmax(1)
    loss, grad = to.dense_energy(torch.from_numpy(g["del_img"]), prob, torch.from_numpy(g["del_roi"]),
                                 torch.from_numpy(g["del_label"]), 1e-7, 15.0, 100.0, 0.5, leaf=logits)
    np.testing.assert_allclose(loss.numpy(), g["del_loss"], rtol=1e-5)
    np.testing.assert_allclose(grad.numpy(), g["del_grad"], rtol=1e-4, atol=1e-12)


def test_camseg_tail_vs_reference(golden):
    from oracle import torch_oracle as to
    from oracle.gen_golden import _StubModel
    g = golden("camseg_tail")
    cam, aux, seg = to.multi_scale_camseg(_StubModel(int(g["C"])), torch.from_numpy(g["imgs"]), [1.0, 0.5, 1.5])
    np.testing.assert_allclose(cam.numpy(), g["cam"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(aux.numpy(), g["cam_aux"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(seg.numpy(), g["seg"], rtol=1e-6, atol=1e-6)


def test_cam_minmax_norm_c_vs_torch(oracle_c):
    rng = np.random.default_rng(3)
    x = np.maximum(rng.normal(0.3, 1, (2, 3, 17, 19)), 0).astype(np.float32)
    t = torch.from_numpy(x)
    t = t - t.amin(dim=(2, 3), keepdim=True)
    t = t / (t.amax(dim=(2, 3), keepdim=True) + 1e-5)
    assert np.array_equal(oracle_c.cam_minmax_norm(x), t.numpy())


def test_misc_vs_reference(oracle_c, golden):
    from oracle import torch_oracle as to
    g = golden("misc")
    assert np.array_equal(oracle_c.denormalize_img(g["denorm_in"]), g["denorm_out"])
    for st, lr in zip(g["lr_steps"], g["lr_values"]):
        assert to.poly_warmup_lr(int(st), 6e-5) == pytest.approx(lr, rel=1e-12)
    t = torch.from_numpy
    np.testing.assert_allclose(to.seg_loss(t(g["segloss_pred"]), t(g["segloss_mask"])).numpy(), g["segloss_out"], rtol=1e-6)
    ref = to.seg_refine_by_label(t(g["refine_seg"]), t(g["refine_labels"]), 0.01)
    np.testing.assert_allclose(ref.numpy(), g["refine_out"], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(to.cam_loss(t(g["camloss_cam"]), ref).numpy(), g["camloss_out"], rtol=1e-6)


def gmm_case(g, n):
    q = np.concatenate([g[f"{n}_queue_rand"], g[f"{n}_queue32"].astype(np.float64)], 0)
    return q, int(g[f"{n}_modal"]), float(g[f"{n}_filter"])


@pytest.mark.parametrize("case", list("abcdef"))
def test_gmm_thresholds_vs_reference(golden, case):
    """utils/seg_helper.py:924-943 rungmm: the restated EM gives the reference's thresholds bit for bit, in as many iterations
    as scikit-learn's estimator took, with the same means to rounding."""
    from oracle import gmm_oracle
    g = golden("gmm")
    q, modal, thr = gmm_case(g, case)
    res = np.atleast_1d(np.array(gmm_oracle.rungmm(q, modal, thr)))
    assert np.array_equal(res, g[f"{case}_thresholds"])
    x = q.flatten()
    _, n_iter, (w, mu, pc) = gmm_oracle.fit(x[x > thr], modal)
    assert n_iter == int(g[f"{case}_n_iter"])
    np.testing.assert_allclose(mu, g[f"{case}_means"], rtol=1e-12)


def test_gmm_empty_component_raises():
    from oracle import gmm_oracle
    with pytest.raises(ValueError):
        gmm_oracle.rungmm(np.full((4, 8), 0.5), 3)                 # identical samples: everything lands in one component


def test_product_poly_warmup_adamw_lr_vs_reference(golden):
    """the PRODUCT's PolyWarmupAdamW (cosa_amd/utils/torch_helper.py; utils/torch_helper.py:261-293 of the reference) steps to the learning
    rates the reference's optimizer produced at the same global steps (tests/golden/misc.npz:lr_values), to the last digit"""
    from cosa_amd.utils import torch_helper as th
    g = golden("misc")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = th.PolyWarmupAdamW([{"params": [p], "lr": 6e-5}], lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=1500,
                             max_iter=32000, warmup_ratio=1e-6, power=0.9, min_mult=0.0)
    for st, lr in zip(g["lr_steps"], g["lr_values"]):
        opt.global_step = int(st)
        p.grad = torch.zeros(1)
        opt.step()
        assert opt.param_groups[0]["lr"] == lr, (st, opt.param_groups[0]["lr"], lr)
        assert th.poly_warmup_lr_mult(int(st), 1500, 32000, 1e-6, 0.9, 0.0) * 6e-5 == lr


def test_oracle_vit_vs_reference_golden(golden):
    """PIN of the yardstick behind every teacher-precision number (profiles/r0*_accuracy_teacher.txt) and behind oracle/cpu_step.py:
    oracle/torch_oracle.py:OracleViT (fp32 torch-CPU restatement of models/vit/vit.py:86-181,219-330, models/__init__.py:82-206,
    models/decoder/conv_head.py:11-41) against the six outputs the REFERENCE's own VITNetwork produced for the 128-wide toy encoder of
    tests/golden/vit_tiny.npz (written by oracle/gen_golden.py from the reference loaded by path).  Measured difference: 0.0."""
    from oracle.torch_oracle import OracleViT, load_golden_state
    g = golden("vit_tiny")
    m = OracleViT(num_classes=7, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4.0, aux_layer=-2)
    sd = {k: v for k, v in load_golden_state(g).items() if not k.startswith("encoder.head.")}   # (the dead ImageNet head has 10 rows there)
    assert set(sd) == {k for k in m.named_state() if not k.startswith("encoder.head.")}         # every parameter of the oracle is set
    m.load_named(sd)
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]))
    for name, o in zip(["cls", "cls_aux", "x4", "seg", "cam", "cam_aux"], out):
        ref = g[name]
        assert o.shape == ref.shape, name
        assert np.abs(o.numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (name, np.abs(o.numpy() - ref).max())


def _vit_base_state(g):
    """the recipe weights of tests/golden/vit_base_d2.npz, replayed (oracle/gen_golden.py:recipe_state) and checked against the stored sha256"""
    from oracle.gen_golden import recipe_state
    shapes = {str(k): tuple(int(d) for d in str(s).split(",")) for k, s in zip(g["shape_keys"], g["shape_dims"])}
    sd, digest = recipe_state(shapes)
    assert digest == str(g["weights_sha256"]), "the replayed weights are not the ones the reference's forward was run with"
    return sd


def test_oracle_vit_at_vit_b_width_vs_reference_golden(golden):
    """the same pin at the HIP kernels' own width: OracleViT(embed 768, 12 heads, depth 2) against the six outputs the
    REFERENCE's VisionTransformer + LargeFOV + classifiers produced on tests/golden/vit_base_d2.npz (2 x 3 x 96 x 64; weights by recipe)"""
    from oracle.torch_oracle import OracleViT
    from oracle.gen_golden import VIT_BASE_CFG as cfg
    g = golden("vit_base_d2")
    sd = _vit_base_state(g)
    m = OracleViT(num_classes=cfg["num_classes"], embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], aux_layer=cfg["aux_layer"])
    assert set(sd) == set(m.named_state())
    m.load_named(sd)
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]))
    for name, o in zip(["cls", "cls_aux", "x4", "seg", "cam", "cam_aux"], out):
        ref = g[name]
        assert o.shape == ref.shape, name
        assert np.abs(o.numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (name, np.abs(o.numpy() - ref).max())
