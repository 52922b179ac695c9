"""GPU parity: HIP label / PAR / CAM-tail / bilateral kernels (through the C ABI) vs the CPU oracle and the goldens."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DIL = [1, 2, 4, 8, 12, 24]


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to("cuda", dtype=dtype)


def smooth(rng, n, h, w):
    from oracle.gen_golden import smooth_field
    return smooth_field(rng, n, h, w)


def test_library_is_loaded_native():
    from cosa_amd import _C
    assert _C.lib().cosa_abi_version() >= 1


def test_denorm_bit_exact(oracle_c, golden):
    from cosa_amd.utils import torch_helper
    g = golden("misc")
    out = torch_helper.denormalize_img(dev(g["denorm_in"])).cpu().numpy()
    assert np.array_equal(out, g["denorm_out"])
    x = np.random.default_rng(0).normal(0, 1.5, (3, 3, 50, 70)).astype(np.float32)
    assert np.array_equal(torch_helper.denormalize_img(dev(x)).cpu().numpy(), oracle_c.denormalize_img(x))


def test_minmax_norm_bit_exact(oracle_c):
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(1)
    for shape in [(2, 3, 17, 19), (2, 20, 448, 448)]:
        x = np.maximum(rng.normal(0.2, 1, shape), 0).astype(np.float32)
        out = seg_helper.cam_minmax_norm_(dev(x).clone()).cpu().numpy()
        assert np.array_equal(out, oracle_c.cam_minmax_norm(x))


def test_par_vs_golden_and_oracle_bit_exact(oracle_c, golden):
    from cosa_amd.models.PAR import PAR
    g = golden("par")
    par = PAR(num_iter=10, dilations=DIL)
    for tag in ("k2", "k4"):
        out = par(dev(g[f"{tag}_img"]), dev(g[f"{tag}_masks"])).cpu().numpy()
        ref = g[f"{tag}_out"]
        assert np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)) < 2e-5          # vs the reference
        orc = oracle_c.par_forward(g[f"{tag}_img"][0], g[f"{tag}_masks"][0], DIL, 10)
        assert np.array_equal(out[0], orc)                                               # vs the oracle: every bit
    par3 = PAR(num_iter=int(g["b2_iter"]), dilations=list(g["b2_dil"]))
    out = par3(dev(g["b2_img"]), dev(g["b2_masks"])).cpu().numpy()
    np.testing.assert_allclose(out, g["b2_out"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("K,h,w", [(21, 40, 52), (17, 33, 47), (5, 30, 31)])
def test_par_many_planes_and_odd_sizes_bit_exact(oracle_c, K, h, w):
    """more planes than one thread holds (two plane groups), plane counts that fall into every body size, odd / non-square sizes
    smaller than the largest dilation: equal to the oracle in every bit"""
    from cosa_amd.models.PAR import PAR
    rng = np.random.default_rng(K)
    img = rng.uniform(0, 1, (1, 3, h, w)).astype(np.float32)
    masks = rng.uniform(0, 1, (1, K, h, w)).astype(np.float32)
    out = PAR(num_iter=10, dilations=DIL)(dev(img), dev(masks)).cpu().numpy()
    assert np.array_equal(out[0], oracle_c.par_forward(img[0], masks[0], DIL, 10))


def test_par_affinity_rows_sum(oracle_c):
    """size-independent property: every affinity row sums to 1 + w2 => a constant mask stays constant * 1.01^T."""
    from cosa_amd.models.PAR import PAR
    rng = np.random.default_rng(2)
    img = dev(rng.uniform(0, 1, (2, 3, 224, 224)).astype(np.float32))
    masks = torch.full((2, 3, 224, 224), 0.5, device="cuda")
    out = PAR(num_iter=10, dilations=DIL)(img, masks)
    assert torch.allclose(out, torch.full_like(out, 0.5 * 1.01 ** 10), rtol=1e-5)


@pytest.mark.parametrize("key,ds,use_par,thr_hi", [("mask_none", 2, False, 0.7), ("mask_none_ds0", 0, False, 0.7),
                                                  ("mask_par", 2, True, 0.7), ("mask_par_coco_thr", 2, True, 0.65)])
def test_cam2mask_golden_bit_exact(golden, key, ds, use_par, thr_hi):
    from cosa_amd.models.PAR import PAR
    from cosa_amd.utils import seg_helper
    g = golden("cam2mask")
    par = PAR(num_iter=10, dilations=DIL) if use_par else None
    vc = seg_helper.cam_validation(dev(g["cams"]), dev(g["labels"]))
    m = seg_helper.cam2mask(dev(g["images"]), torch.from_numpy(g["boxes"]), vc, dev(g["labels"]), thr_hi, 0.25,
                            refine_model=par, downscale=ds)
    assert np.array_equal(m.cpu().numpy(), g[key])
    # raw cams + folded validation give the same labels
    m2 = seg_helper.cam2mask(dev(g["images"]), torch.from_numpy(g["boxes"]), dev(g["cams"]), dev(g["labels"]), thr_hi, 0.25,
                             refine_model=par, downscale=ds, _fold_validation=True)
    assert torch.equal(m, m2)


@pytest.mark.parametrize("use_par", [False, True])
def test_cam2mask_full_size_vs_oracle(oracle_c, use_par):
    """BASELINE config shape (448x448, 20 classes), a few images; ragged label counts incl. an image with no fg."""
    from cosa_amd.models.PAR import PAR
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(7)
    B, C, S = 3, 20, 448
    cams = np.maximum(smooth(rng, B * C, S, S).reshape(B, C, S, S) * 1.3 - 0.15, 0).astype(np.float32)
    labels = np.zeros((B, C), np.float32)
    labels[0, [3]] = 1
    labels[1, [0, 7, 19]] = 1          # image 2: no foreground class at all
    boxes = np.array([[0, S, 0, S], [10, 400, 33, 448], [0, 448, 0, 100]], np.int32)
    images = smooth(rng, B * 3, S, S).reshape(B, 3, S, S)
    par = PAR(num_iter=10, dilations=DIL) if use_par else None
    m = seg_helper.cam2mask(dev(images), torch.from_numpy(boxes), dev(cams), dev(labels), 0.7, 0.25, refine_model=par,
                            _fold_validation=True).cpu().numpy()
    ref = oracle_c.cam2mask(images, boxes, cams, labels, 0.7, 0.25, 2, par=(DIL, 10) if use_par else None)
    assert np.array_equal(m, ref), f"{(m != ref).sum()} labels differ"
    assert np.all(m[2][:, 100:] == 255) and set(np.unique(m[2][:, :100])) == {0.0}


@pytest.mark.parametrize("use_par", [False, True])
def test_cam2mask_multi_vs_oracle(oracle_c, use_par):
    """two CAM sets of the same images with their own thresholds through ONE pass == the oracle's two separate calls,
    bit for bit; label counts 1 / 3 / 6 / 0 classes so that every plane-count bracket of the PAR step runs (2..28 live planes)."""
    from cosa_amd.models.PAR import PAR
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(17)
    B, C, S = 4, 20, 224
    cams = [np.maximum(smooth(rng, B * C, S, S).reshape(B, C, S, S) * 1.3 - 0.15, 0).astype(np.float32) for _ in range(2)]
    labels = np.zeros((B, C), np.float32)
    labels[0, [3]] = 1
    labels[1, [0, 7, 19]] = 1
    labels[2, [1, 2, 5, 8, 13, 18]] = 1
    boxes = np.array([[0, S, 0, S], [10, 200, 33, S], [0, S, 0, 100], [5, 220, 0, S]], np.int32)
    images = smooth(rng, B * 3, S, S).reshape(B, 3, S, S)
    par = PAR(num_iter=10, dilations=DIL) if use_par else None
    thr_hi, thr_lo = [0.7, 0.55], [0.25, 0.35]
    ms = seg_helper.cam2mask_multi(dev(images), torch.from_numpy(boxes), [dev(c) for c in cams], dev(labels), thr_hi, thr_lo,
                                   refine_model=par, _fold_validation=True)
    for g in range(2):
        ref = oracle_c.cam2mask(images, boxes, cams[g], labels, thr_hi[g], thr_lo[g], 2, par=(DIL, 10) if use_par else None)
        m = ms[g].cpu().numpy()
        assert np.array_equal(m, ref), f"set {g}: {(m != ref).sum()} labels differ"
        one = seg_helper.cam2mask(dev(images), torch.from_numpy(boxes), dev(cams[g]), dev(labels), thr_hi[g], thr_lo[g],
                                  refine_model=par, _fold_validation=True)
        assert torch.equal(one, ms[g])


def test_cam2mask_properties_at_bench_size():
    """size-independent checks at b=16: outside-box = 255; values in {0, active classes, 255}; hi/lo merge rule."""
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(8)
    B, C, S = 16, 20, 448
    cams = torch.rand(B, C, S // 8, S // 8, device="cuda")
    cams = torch.nn.functional.interpolate(cams, size=(S, S), mode="bilinear")
    labels = torch.zeros(B, C, device="cuda")
    for b in range(B):
        labels[b, rng.choice(C, size=rng.integers(1, 4), replace=False)] = 1
    boxes = torch.tensor([[0, S, 0, S]] * 8 + [[16, 400, 32, 432]] * 8, dtype=torch.int16)
    m = seg_helper.cam2mask(torch.zeros(B, 3, S, S, device="cuda"), boxes, cams, labels, 0.7, 0.25, _fold_validation=True)
    assert torch.all(m[8:, :16] == 255) and torch.all(m[8:, :, 432:] == 255)
    for b in range(B):
        allowed = {0.0, 255.0} | {float(c + 1) for c in torch.nonzero(labels[b])[:, 0].tolist()}
        assert set(torch.unique(m[b]).tolist()) <= allowed
    # thresholds: raising the high threshold can only turn fg into ignore, never into another class
    m2 = seg_helper.cam2mask(torch.zeros(B, 3, S, S, device="cuda"), boxes, cams, labels, 0.9, 0.25, _fold_validation=True)
    changed = m != m2
    assert torch.all((m2[changed] == 255) | (m2[changed] == 0))


def test_camseg_tail_vs_golden(golden):
    from cosa_amd.utils import seg_helper
    from oracle.gen_golden import _StubModel
    g = golden("camseg_tail")
    C = int(g["C"])
    stub = _StubModel(C)

    def model(x, cam_only=False):
        outs = stub(x.cpu())
        return tuple(o.cuda() if o is not None else None for o in outs)

    cam, aux, seg = seg_helper.multi_scale_camseg(model, dev(g["imgs"]), [1.0, 0.5, 1.5])
    np.testing.assert_allclose(cam.cpu().numpy(), g["cam"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(aux.cpu().numpy(), g["cam_aux"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(seg.cpu().numpy(), g["seg"], rtol=1e-6, atol=1e-6)


def test_bilateral_vs_golden(oracle_c, golden):
    from cosa_amd import _C
    g = golden("bilateral")
    L = _C.lib()
    for tag in ("smooth", "noise", "odd"):
        img, seg, ref = dev(g[f"{tag}_img"]), dev(g[f"{tag}_seg"]), g[f"{tag}_out"]
        N, K, H, W = seg.shape
        out = torch.empty_like(seg)
        Ms = torch.zeros(N, dtype=torch.int32, device="cuda")
        ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), "cuda", "t")
        _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(img), _C.ptr(seg), _C.ptr(out), N, K, H, W, 15.0, 50.0, _C.ptr(Ms),
                                                  _C.ptr(ws), ws.numel(), _C.stream_ptr()))
        _, M_ref = oracle_c.bilateralfilter_batch(g[f"{tag}_img"], g[f"{tag}_seg"], N, K, H, W, 15.0, 50.0)
        assert np.array_equal(Ms.cpu().numpy(), M_ref)               # identical lattice (integer work): exact
        # sorted splat (round 2): every vertex adds its pixels up in the reference's order -> the reference's bits, no tolerance
        assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("N", [1, 2, 3, 5, 6, 7, 8, 9, 16, 17])
def test_bilateral_every_batch_size_places_its_images_on_xcds(oracle_c, N):
    """round 5: the lattice kernels map a 1-D grid to (image, workgroup) so that an image's workgroups sit on one XCD (N >= 8) or on 8 / N of them
    (permuto_kernels.hip: image_wg, image_parts -- 8, 4, 8, 4, 4, 1, 1, ... XCDs per image for these N; idle XCDs and a second round of images
    for N = 9, 17): the same bits as the serial reference for every split, lattice sizes included"""
    from cosa_amd import _C
    from oracle.gen_golden import synth_image255
    rng = np.random.default_rng(100 + N)
    K, H, W = 5, 36, 44
    img = synth_image255(rng, N, H, W)
    if N > 1:
        img[1] = rng.uniform(0, 255, img[1].shape).astype(np.float32)          # one noise image: a lattice several times the others' size
    seg = rng.standard_normal((N, K, H, W)).astype(np.float32)
    L = _C.lib()
    out = torch.empty((N, K, H, W), device="cuda")
    Ms = torch.zeros(N, dtype=torch.int32, device="cuda")
    ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), "cuda", "t")
    d_img, d_seg = dev(img), dev(seg)
    _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(d_img), _C.ptr(d_seg), _C.ptr(out), N, K, H, W, 15.0, 50.0, _C.ptr(Ms), _C.ptr(ws), ws.numel(),
                                              _C.stream_ptr()))
    ref, M_ref = oracle_c.bilateralfilter_batch(img, seg, N, K, H, W, 15.0, 50.0)
    assert np.array_equal(Ms.cpu().numpy(), M_ref)
    assert np.array_equal(out.cpu().numpy(), ref.reshape(N, K, H, W))


def test_bilateralfilter_module_numpy_signature(golden):
    from cosa_amd import bilateralfilter as bf
    g = golden("bilateral")
    img, seg, ref = g["noise_img"], g["noise_seg"], g["noise_out"]
    N, K, H, W = seg.shape
    out = np.zeros(seg.size, np.float32)
    bf.bilateralfilter_batch(img.flatten(), seg.flatten(), out, N, K, H, W, 15.0, 50.0)
    assert np.array_equal(out.reshape(ref.shape), ref)
    out1 = np.zeros(K * H * W, np.float32)
    bf.bilateralfilter(img[0].flatten(), seg[0].flatten(), out1, H, W, 15.0, 50.0)
    assert np.array_equal(out1.reshape(ref[0].shape), ref[0])
    with pytest.raises(TypeError):
        bf.bilateralfilter_batch(img, seg, np.zeros(3, np.float64), N, K, H, W, 15.0, 50.0)


def test_dense_energy_loss_and_grad_vs_golden(golden):
    from cosa_amd.utils import seg_helper, rrm_utils
    assert rrm_utils.DenseEnergyLoss is seg_helper.DenseEnergyLoss
    g = golden("bilateral")
    logits = dev(g["del_logits"]).requires_grad_(True)
    layer = seg_helper.DenseEnergyLoss(weight=1e-7, sigma_rgb=15, sigma_xy=100, scale_factor=0.5)
    loss = layer(dev(g["del_img"]), logits.softmax(1), dev(g["del_roi"]), dev(g["del_label"], torch.uint8))
    loss.backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["del_loss"], rtol=1e-4)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g["del_grad"], rtol=1e-3, atol=1e-11)


def test_dense_energy_full_size_vs_oracle(oracle_c):
    """224x224 (the size the training step filters at), K=21, smooth image + noise image."""
    from cosa_amd import _C
    from oracle.gen_golden import synth_image255
    rng = np.random.default_rng(9)
    N, K, H, W = 2, 21, 224, 224
    img = synth_image255(rng, N, H, W)
    img[1] = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    seg = torch.from_numpy(smooth(rng, N * K, H, W).reshape(N, K, H, W) * 4).softmax(1).numpy()
    roi = np.ones((N, H, W), np.float32)
    roi[1, :20] = 0
    unl = (rng.uniform(size=(N, H, W)) < 0.3).astype(np.uint8)
    loss_ref, AS_ref = oracle_c.dense_energy_forward(img, seg, roi, unl, 15.0, 50.0)
    L = _C.lib()
    AS = torch.empty(N, K, H, W, device="cuda")
    loss = torch.empty(1, device="cuda")
    ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), "cuda", "t")
    d_img, d_seg, d_roi, d_unl = dev(img), dev(seg), dev(roi), dev(unl, torch.uint8)     # keep alive across the launch
    _C.check(L.cosa_dense_energy_forward(_C.ptr(d_img), _C.ptr(d_seg), _C.ptr(d_roi), _C.ptr(d_unl),
                                         _C.ptr(AS), _C.ptr(loss), N, K, H, W, 15.0, 50.0, _C.ptr(ws), ws.numel(),
                                         _C.stream_ptr()))
    np.testing.assert_allclose(AS.cpu().numpy(), AS_ref, rtol=1e-4, atol=1e-5)
    assert loss.item() == pytest.approx(loss_ref, rel=1e-4)
    n_exact = int((AS.cpu().numpy() == AS_ref.reshape(N, K, H, W)).sum())
    print("dense energy AS: %d of %d elements bit-identical to the oracle" % (n_exact, AS.numel()))
    # no float atomics anywhere on the path (sorted splat, fixed-order loss reduction): a second run gives the same bits
    AS2 = torch.empty_like(AS)
    loss2 = torch.empty(1, device="cuda")
    _C.check(L.cosa_dense_energy_forward(_C.ptr(d_img), _C.ptr(d_seg), _C.ptr(d_roi), _C.ptr(d_unl),
                                         _C.ptr(AS2), _C.ptr(loss2), N, K, H, W, 15.0, 50.0, _C.ptr(ws), ws.numel(),
                                         _C.stream_ptr()))
    assert torch.equal(AS, AS2) and torch.equal(loss, loss2)


def test_errors_are_loud():
    from cosa_amd import _C
    from cosa_amd.utils import seg_helper
    with pytest.raises(TypeError):
        seg_helper.cam2mask(torch.zeros(1, 3, 8, 8, device="cuda"), torch.tensor([[0, 8, 0, 8]]), torch.zeros(1, 2, 8, 8, device="cuda"),
                            torch.ones(1, 2, device="cuda"), 0.7, 0.25, refine_model="not callable")
    with pytest.raises(_C.CosaError):
        _C.check(_C.lib().cosa_cam_minmax_norm(None, 0, 0, None, None), "bad call")


def test_cam2mask_generic_hook_matches_the_fused_path():
    """utils/seg_helper.py:787-792 accepts ANY callable as refine_model (and non-square crops): those go through the reference's
    per-image loop on the GPU.  With an identity refine model it must agree with the fused kernels (same arithmetic spec up to the
    contraction of torch's GPU bilinear kernel: a handful of tie pixels at most)."""
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(5)
    B, C, S = 2, 20, 64
    cams = np.maximum(smooth(rng, B * C, S, S).reshape(B, C, S, S) * 1.3 - 0.15, 0).astype(np.float32)
    labels = np.zeros((B, C), np.float32)
    labels[0, [1, 7]] = 1
    labels[1, [12]] = 1
    boxes = torch.tensor([[0, S, 0, S], [4, 60, 8, 50]], dtype=torch.int32)
    img = dev(smooth(rng, B * 3, S, S).reshape(B, 3, S, S))
    calls = []

    def identity(images, c):
        calls.append((tuple(images.shape), tuple(c.shape)))
        return c
    fused = seg_helper.cam2mask(img, boxes, dev(cams), dev(labels), 0.7, 0.25, _fold_validation=True)
    gen = seg_helper.cam2mask(img, boxes, dev(cams), dev(labels), 0.7, 0.25, refine_model=identity, _fold_validation=True)
    assert len(calls) == 2 * B and calls[0] == ((1, 3, S // 2, S // 2), (1, 3, S // 2, S // 2))
    assert gen.shape == fused.shape and (gen == fused).float().mean().item() > 0.995
    assert set(np.unique(gen.cpu().numpy())) <= set([0.0, 255.0] + [float(k + 1) for k in range(C)])
    # non-square crops take the same path
    m = seg_helper.cam2mask(img[:, :, :48], torch.tensor([[0, 48, 0, S], [4, 40, 8, 50]]), dev(cams[:, :, :48]), dev(labels), 0.7, 0.25,
                            _fold_validation=True)
    assert m.shape == (B, 48, S)


def test_bilateral_noise_and_smooth_images_in_one_batch(oracle_c):
    """uniform-noise image (almost every pixel owns its lattice vertices: the content-dependent worst case) next to a smooth one in the
    same batch: lattice sizes equal to the CPU oracle's, filtered values to 2e-5"""
    from cosa_amd import _C
    rng = np.random.default_rng(21)
    N, K, H, W = 2, 5, 48, 64
    img = np.stack([rng.uniform(0, 255, (3, H, W)), smooth(rng, 3, H, W) * 255]).astype(np.float32)
    x = rng.uniform(0, 1, (N, K, H, W)).astype(np.float32)
    L = _C.lib()
    di, dx = dev(img), dev(x)
    out = torch.empty_like(dx)
    ws = _C.workspace(L.cosa_bilateral_workspace_bytes(N, K, H, W), di.device, "bilateral")
    sizes = torch.zeros(N, dtype=torch.int32, device="cuda")
    _C.check(L.cosa_bilateralfilter_batch_dev(_C.ptr(di), _C.ptr(dx), _C.ptr(out), N, K, H, W, 15.0, 50.0, _C.ptr(sizes), _C.ptr(ws),
                                              ws.numel(), _C.stream_ptr()), "bilateral")
    ref, M = oracle_c.bilateralfilter_batch(img, x, N, K, H, W, 15.0, 50.0)
    np.testing.assert_allclose(out.cpu().numpy().reshape(-1), np.asarray(ref).reshape(-1), rtol=2e-5, atol=2e-6)
    assert np.array_equal(sizes.cpu().numpy(), np.asarray(M)) and int(sizes[0]) > 2 * int(sizes[1])


def test_cam2mask_multi_with_par_at_bench_size():
    """size-independent property at BASELINE's configuration (b=16, 448^2, PAR T=10, 6 dilations): the shared pass over main + aux CAM
    sets equals the two separate calls bit for bit (different plane groupings, XCD pinning and launch shapes), and the invariants hold"""
    from cosa_amd.models.PAR import PAR
    from cosa_amd.utils import seg_helper
    rng = np.random.default_rng(9)
    B, C, S = 16, 20, 448
    up = lambda t: torch.nn.functional.interpolate(t.cuda(), size=(S, S), mode="bilinear")
    g = torch.Generator().manual_seed(9)
    cams = [up(torch.rand(B, C, S // 8, S // 8, generator=g)) for _ in range(2)]
    img = up(torch.rand(B, 3, S // 4, S // 4, generator=g))
    labels = torch.zeros(B, C, device="cuda")
    for b in range(B):
        labels[b, rng.choice(C, size=rng.integers(1, 5), replace=False)] = 1
    boxes = torch.tensor([[0, S, 0, S]] * 8 + [[16, 400, 32, 432]] * 8, dtype=torch.int16)
    par = PAR(num_iter=10, dilations=DIL)
    ms = seg_helper.cam2mask_multi(img, boxes, cams, labels, [0.7, 0.6], [0.25, 0.3], refine_model=par, _fold_validation=True)
    for i, (hi, lo) in enumerate(((0.7, 0.25), (0.6, 0.3))):
        one = seg_helper.cam2mask(img, boxes, cams[i], labels, hi, lo, refine_model=par, _fold_validation=True)
        assert torch.equal(one, ms[i])
        assert torch.all(ms[i][8:, :16] == 255) and torch.all(ms[i][8:, :, 432:] == 255)
        for b in range(B):
            allowed = {0.0, 255.0} | {float(c + 1) for c in torch.nonzero(labels[b])[:, 0].tolist()}
            assert set(torch.unique(ms[i][b]).tolist()) <= allowed


def test_multi_scale_camseg_persistent_buffers_track_the_active_planes(golden):
    """the training loop's CAM buffers live from step to step and only the planes that were live LAST time are cleared: three calls with
    different label sets must each equal the stateless path followed by cam_validation (seg_helper.py:547-551), bit for bit"""
    from cosa_amd.utils import seg_helper
    from oracle.gen_golden import _StubModel
    g = golden("camseg_tail")
    C = int(g["C"])
    stub = _StubModel(C)

    def model(x, cam_only=False):
        outs = stub(x.cpu())
        return tuple(o.cuda() if o is not None else None for o in outs)

    imgs = dev(g["imgs"])
    b = imgs.shape[0]
    rng = np.random.default_rng(3)
    for trial in range(3):
        lab = (rng.uniform(size=(b, C)) < (0.5, 0.15, 0.8)[trial]).astype(np.float32)
        lab[:, trial % C] = 1.0
        labels = dev(lab)
        cam, aux, _ = seg_helper.multi_scale_camseg(model, imgs, [1.0, 0.5, 1.5], _active_labels=labels)
        ref_cam, ref_aux, _ = seg_helper.multi_scale_camseg(model, imgs, [1.0, 0.5, 1.5])
        # stateless reference: normalise all planes, then zero the absent ones -- the normalisation is per plane, so the live planes agree
        m = labels[:, :, None, None]
        assert torch.equal(cam, ref_cam * m) and torch.equal(aux, ref_aux * m)


@pytest.mark.parametrize("n,bits", [(1, 8), (2047, 5), (2048, 8), (2049, 16), (100_003, 23), (4_816_896, 23), (300_000, 32)])
def test_radix_sort_is_a_stable_sort(n, bits):
    """the lattice's own LSD radix sort (csrc/radix_sort.hpp; round 4: replaces rocPRIM) against numpy's stable argsort on keys with many
    duplicates -- equal keys must keep their input order (the bit-exactness of the bilateral filter rests on it)"""
    import torch
    from cosa_amd import _C
    rng = np.random.default_rng(n)
    hi = (1 << bits) - 1
    keys = (rng.integers(0, min(hi, 5000) + 1, n, dtype=np.uint64) * max(1, hi // 5000)).astype(np.uint32) & np.uint32(hi)
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = torch.from_numpy(keys.view(np.int32)).cuda(), torch.from_numpy(vals.view(np.int32)).cuda()
    ok, ov = torch.empty_like(dk), torch.empty_like(dv)
    L = _C.lib()
    ws = torch.empty(int(L.cosa_radix_sort_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    _C.check(L.cosa_radix_sort_pairs(_C.ptr(dk), _C.ptr(dv), _C.ptr(ok), _C.ptr(ov), n, bits, _C.ptr(ws), ws.numel(), _C.stream_ptr()), "sort")
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ok.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(ov.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("size", [(224, 224), (672, 672), (37, 91), (448, 448), (960, 960)])
def test_resize_bilinear_is_atens_formula(size):
    """the teacher's input rescale (utils/seg_helper.py:247-250: F.interpolate(..., mode='bilinear', align_corners=False)) on the own kernel:
    ATen's formula with its fused source-index multiply-add -- against the fp32 CPU operator (what the oracle runs) within 2 ulp of the image range
    (an unfused index is 3e-5 off: the 1.5x pass of the teacher then starts from other pixels than the oracle's), and the same
    sampling geometry (a constant image stays constant to the last bit or two)"""
    from cosa_amd.utils import seg_helper
    torch.manual_seed(size[0])
    for b, c, h, w in ((2, 3, 448, 448), (1, 3, 50, 70)):
        x = torch.randn(b, c, h, w) * 2
        ref = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
        got = seg_helper.resize_bilinear(x.cuda(), size).cpu()
        assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2 * 2.0 ** -23 * x.abs().max().item()
        const = torch.full((1, 1, h, w), 0.37)
        assert (seg_helper.resize_bilinear(const.cuda(), size).cpu() - 0.37).abs().max().item() <= 1e-7
