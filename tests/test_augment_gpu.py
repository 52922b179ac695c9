"""GPU parity: the device input pipeline (aug_kernels.hip, SURVEY f-2) vs the reference's golden vectors and the CPU oracle."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_vs_reference_golden(golden):
    """all 18 reference cases (nine strong ops x blur on/off, flips, up- and down-scaling, padding) as ONE device batch: same
    seeds -> same draws, and crop / weak / strong uint8 stages + img_box bit for bit; float outputs = ToTensor + Normalize"""
    from oracle import aug_oracle
    from cosa_amd.dataloaders import DeviceAugmenter, draw_params
    g = golden("augment")
    S, n = int(g["crop_size"]), int(g["n"])
    images, params = [], []
    for i in range(n):
        img, seed = g[f"{i}_image"], int(g[f"{i}_seed"])
        random.seed(seed)
        np.random.seed(seed)
        images.append(img)
        params.append(draw_params(img.shape[0], img.shape[1], crop_size=S))
        assert params[-1]["op"] == int(g[f"{i}_op"])
    wimg, simg, box, (crop, weak, strong) = DeviceAugmenter(S)(images, params, debug=True)
    crop, weak, strong = crop.cpu().numpy(), weak.cpu().numpy(), strong.cpu().numpy()
    for i in range(n):
        assert np.array_equal(box[i].numpy(), g[f"{i}_box"]), i
        assert np.array_equal(crop[i], g[f"{i}_crop"]), (i, "crop")
        assert np.array_equal(weak[i], g[f"{i}_weak"]), (i, "weak")
        assert np.array_equal(strong[i], g[f"{i}_strong"]), (i, "strong", params[i]["op"])
        assert np.array_equal(wimg[i].cpu().numpy(), aug_oracle.normalize(g[f"{i}_weak"]))
        assert np.array_equal(simg[i].cpu().numpy(), aug_oracle.normalize(g[f"{i}_strong"]))
    assert box.dtype == torch.int16 and wimg.shape == (n, 3, S, S)


def test_pipeline_full_size_vs_oracle():
    """BASELINE's configuration: 448^2 crops from VOC-sized images; every stage equal to the CPU oracle bit for bit, every op seen"""
    from oracle import aug_oracle
    from cosa_amd.dataloaders import DeviceAugmenter, draw_params
    rng = np.random.default_rng(4)
    S = 448
    images, params, seen = [], [], set()
    random.seed(11)
    np.random.seed(11)
    sizes = [(375, 500), (500, 333), (281, 500), (500, 500), (112, 150), (400, 300)]
    while len(seen) < 9 or len(images) < 12:
        h, w = sizes[len(images) % len(sizes)]
        small = rng.integers(0, 256, (h // 6 + 2, w // 6 + 2, 3), dtype=np.uint8)
        img = aug_oracle.resize_bilinear(small, w, h)                       # smooth content
        img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
        p = draw_params(h, w, crop_size=S)
        images.append(img)
        params.append(p)
        seen.add(p["op"])
        assert len(images) < 64
    wimg, simg, box, (crop, weak, strong) = DeviceAugmenter(S)(images, params, debug=True)
    crop, weak, strong = crop.cpu().numpy(), weak.cpu().numpy(), strong.cpu().numpy()
    for i, (img, p) in enumerate(zip(images, params)):
        q = dict(p)
        c, wk, st = aug_oracle.apply(img, q, S)
        assert np.array_equal(crop[i], c), (i, "crop", p)
        assert np.array_equal(weak[i], wk), (i, "weak", p)
        assert np.array_equal(strong[i], st), (i, "strong", p)
        assert np.array_equal(wimg[i].cpu().numpy(), aug_oracle.normalize(wk))
        assert np.array_equal(simg[i].cpu().numpy(), aug_oracle.normalize(st))
        b0, b1, b2, b3 = (int(v) for v in box[i])
        assert not crop[i][:b0].any() and not crop[i][b1:].any() and not crop[i][:, :b2].any() and not crop[i][:, b3:].any()


def test_draws_follow_the_reference_order():
    """size-independent property: the draw sequence consumes Python's and numpy's generators exactly as the oracle's restatement
    of the reference does (same parameters for the same seeds, and the same generator state afterwards)"""
    from oracle import aug_oracle
    from cosa_amd.dataloaders import draw_params
    for seed in range(20):
        random.seed(seed)
        np.random.seed(seed)
        a = draw_params(375, 500)
        sa = (random.getstate(), np.random.get_state()[1].copy(), np.random.get_state()[2])
        random.seed(seed)
        np.random.seed(seed)
        b = aug_oracle.draw_params(375, 500)
        sb = (random.getstate(), np.random.get_state()[1].copy(), np.random.get_state()[2])
        for k in ("new_w", "new_h", "flip", "H_pad", "W_pad", "H_start", "W_start", "blur", "radius", "op", "magnitude"):
            assert a[k] == b[k], (seed, k)
        assert np.array_equal(a["img_box"], b["img_box"])
        assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]


def test_augmenter_rejects_bad_input():
    from cosa_amd.dataloaders import DeviceAugmenter, draw_params
    aug = DeviceAugmenter(64)
    p = draw_params(40, 50, crop_size=64)
    with pytest.raises(ValueError):
        aug([np.zeros((40, 51, 3), np.uint8)], [p])
    with pytest.raises(ValueError):
        aug([np.zeros((40, 50, 3), np.float32)], [p])
    with pytest.raises(ValueError):
        aug([], [])


def test_train_loader_end_to_end(tmp_path, make_voc_tree):
    """JPEG files -> DeviceTrainLoader -> the reference's batch tuple; pixels equal the CPU oracle applied to the same decode + draws"""
    from oracle import aug_oracle
    from cosa_amd.dataloaders import DeviceTrainLoader, VOC12ClsDatasetNew
    root, lists, names, labels = make_voc_tree(tmp_path, n=6)
    ds = VOC12ClsDatasetNew(root_dir=root, name_list_dir=lists, crop_size=64)
    loader = DeviceTrainLoader(ds, batch_size=4, num_workers=0)
    assert len(loader) == 1
    random.seed(9)
    np.random.seed(9)
    (img_name, wimg, simg, cls_label, img_box), = list(loader)
    assert img_name == names[:4] and wimg.shape == simg.shape == (4, 3, 64, 64) and wimg.is_cuda
    assert cls_label.shape == (4, 20) and img_box.shape == (4, 4) and img_box.dtype == torch.int16
    random.seed(9)
    np.random.seed(9)
    for i in range(4):
        _, image, params, label = ds[i]
        _, wk, st = aug_oracle.apply(image, params, 64)
        assert np.array_equal(wimg[i].cpu().numpy(), aug_oracle.normalize(wk))
        assert np.array_equal(simg[i].cpu().numpy(), aug_oracle.normalize(st))
        assert np.array_equal(img_box[i].numpy(), params["img_box"]) and np.array_equal(cls_label[i].numpy(), label)
