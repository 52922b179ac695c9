"""CPU suite: the input-pipeline oracle (oracle/aug_oracle.py, SURVEY f-2) against the reference's own transforms (golden
vectors) and against Pillow, the library whose arithmetic it restates."""
import random

import numpy as np
import pytest


def test_pipeline_vs_reference_golden(golden):
    """same seeds -> same draws (scale, flip, pads, crop window, blur, op, magnitude), and the three uint8 stages + img_box of the
    reference's transforms.py / randaug.py functions bit for bit; all nine strong ops x blur on/off are in the vectors"""
    from oracle import aug_oracle
    g = golden("augment")
    crop = int(g["crop_size"])
    seen = set()
    for i in range(int(g["n"])):
        img, seed = g[f"{i}_image"], int(g[f"{i}_seed"])
        random.seed(seed)
        np.random.seed(seed)
        p = aug_oracle.draw_params(img.shape[0], img.shape[1], crop_size=crop)
        c, wk, st = aug_oracle.apply(img, p, crop)
        assert p["op"] == int(g[f"{i}_op"]) and np.array_equal(p["img_box"], g[f"{i}_box"])
        assert np.array_equal(c, g[f"{i}_crop"]) and np.array_equal(wk, g[f"{i}_weak"]) and np.array_equal(st, g[f"{i}_strong"])
        seen.add((p["op"], p["blur"]))
    assert {o for o, _ in seen} == set(range(9)) and {b for _, b in seen} == {True, False}


def test_resize_and_blur_vs_pillow():
    from PIL import Image, ImageFilter
    from oracle import aug_oracle
    rng = np.random.default_rng(0)
    for _ in range(6):
        h, w = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        s = rng.uniform(0.5, 2.0)
        W, H = int(s * w), int(s * h)
        ref = np.asarray(Image.fromarray(img).resize((W, H), resample=Image.BILINEAR))
        assert np.array_equal(aug_oracle.resize_bilinear(img, W, H), ref)
        r = float(rng.uniform(0.1, 2.0))
        ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=r)))
        assert np.array_equal(aug_oracle.gaussian_blur(img, r), ref)
    img = rng.integers(0, 256, (500, 375, 3), dtype=np.uint8)          # VOC-sized, both directions of scaling
    for W, H in ((187, 250), (750, 1000), (375, 333)):
        assert np.array_equal(aug_oracle.resize_bilinear(img, W, H), np.asarray(Image.fromarray(img).resize((W, H), resample=Image.BILINEAR)))


@pytest.mark.parametrize("magnitude", [1, 5, 9])
def test_strong_ops_vs_pillow(magnitude):
    from PIL import Image, ImageEnhance, ImageOps
    from oracle import aug_oracle
    rng = np.random.default_rng(magnitude)
    small = rng.integers(0, 256, (14, 17, 3), dtype=np.uint8)
    for img in (rng.integers(30, 220, (61, 47, 3), dtype=np.uint8), np.asarray(Image.fromarray(small).resize((68, 56), Image.BICUBIC)),
                np.full((9, 9, 3), 77, np.uint8)):
        pil = Image.fromarray(img)
        v = aug_oracle.enhance_factor(magnitude)
        refs = {1: ImageOps.autocontrast(pil), 2: ImageOps.equalize(pil), 4: ImageEnhance.Color(pil).enhance(v),
                5: ImageEnhance.Contrast(pil).enhance(v), 6: ImageEnhance.Brightness(pil).enhance(v),
                7: ImageEnhance.Sharpness(pil).enhance(v), 8: ImageOps.posterize(pil, 4 - int(magnitude * 4 / 10))}
        for op, ref in refs.items():
            assert np.array_equal(aug_oracle.strong_op(img, op, magnitude), np.asarray(ref)), aug_oracle.OPS[op]
        assert np.array_equal(aug_oracle.strong_op(img, 0, magnitude), img)
        thr = min(int(magnitude * 256 / 10), 255)
        assert np.array_equal(aug_oracle.strong_op(img, 3, magnitude), np.asarray(ImageOps.solarize(pil, thr)))   # same rule as mmcv's


def test_normalize_is_totensor_normalize():
    import torch
    from oracle import aug_oracle
    u8 = np.random.default_rng(3).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    t = torch.from_numpy(u8).permute(2, 0, 1).to(torch.float32).div(255)                    # torchvision ToTensor
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    ref = t.sub(mean).div(std)                                                              # torchvision Normalize
    assert np.array_equal(aug_oracle.normalize(u8), ref.numpy())


def test_train_dataset_contract(tmp_path, make_voc_tree):
    """dataloaders/voc.py:219-305 minus the pixels: name, decoded image, the draws (seed-reproducible, reference order), label"""
    from PIL import Image
    from oracle import aug_oracle
    from cosa_amd.dataloaders import VOC12ClsDatasetNew
    root, lists, names, labels = make_voc_tree(tmp_path)
    ds = VOC12ClsDatasetNew(root_dir=root, name_list_dir=lists, crop_size=64, rescale_range=[0.5, 2.0])
    assert len(ds) == len(names)
    random.seed(5)
    np.random.seed(5)
    name, image, params, label = ds[2]
    assert name == names[2] and np.array_equal(label, labels[names[2]])
    assert np.array_equal(image, np.asarray(Image.open(f"{root}/JPEGImages/{name}.jpg").convert("RGB")))
    random.seed(5)
    np.random.seed(5)
    ref = aug_oracle.draw_params(image.shape[0], image.shape[1], crop_size=64)
    assert all(params[k] == ref[k] for k in ("new_w", "new_h", "flip", "H_pad", "W_pad", "H_start", "W_start", "blur", "radius", "op",
                                             "magnitude")) and np.array_equal(params["img_box"], ref["img_box"])


def test_val_dataset_contract(tmp_path, make_voc_tree):
    """dataloaders/voc.py:306-368 (aug=False): (name, normalised CHW float32 image, label map, cls_label); normalize_img is the
    reference's float64 expression stored as float32"""
    from types import SimpleNamespace
    from PIL import Image
    from cosa_amd.dataloaders import VOC12SegDataset, build_val_loader, normalize_img
    root, lists, names, labels = make_voc_tree(tmp_path)
    import os
    import shutil
    os.makedirs(f"{root}/SegmentationClassAug")
    for n in names:
        im = np.asarray(Image.open(f"{root}/JPEGImages/{n}.jpg"))
        Image.fromarray((im[..., 0] // 13).astype(np.uint8)).save(f"{root}/SegmentationClassAug/{n}.png")
    shutil.copy(f"{lists}/train_aug.txt", f"{lists}/val.txt")
    ds = VOC12SegDataset(root_dir=root, name_list_dir=lists, split="val", stage="val", aug=False)
    name, image, label, cls = ds[1]
    raw = np.asarray(Image.open(f"{root}/JPEGImages/{name}.jpg").convert("RGB"))
    ref = np.empty_like(raw, np.float32)
    for c, (m, sd) in enumerate(zip([123.675, 116.28, 103.53], [58.395, 57.12, 57.375])):
        ref[..., c] = (raw[..., c] - m) / sd
    assert image.dtype == np.float32 and np.array_equal(image, ref.transpose(2, 0, 1)) and label.shape == raw.shape[:2]
    assert np.array_equal(cls, labels[name]) and np.array_equal(normalize_img(raw), ref)
    args = SimpleNamespace(dataset="VOC12", voc12_root=root, name_list_dir=lists, ignore_index=255, num_classes=21)
    batch = next(iter(build_val_loader(args, num_workers=0)))
    assert list(batch[0]) == [names[0]] and batch[1].shape[:2] == (1, 3) and batch[3].shape == (1, 20)


def test_train_dataset_through_worker_processes(tmp_path, make_voc_tree):
    """decode + draws run inside DataLoader worker processes (as the reference's transforms do): items survive pickling, batches keep the
    (names, images, draws, labels) layout, every image matches its draws"""
    from torch.utils.data import DataLoader
    from cosa_amd.dataloaders import VOC12ClsDatasetNew
    from cosa_amd.dataloaders.train_loader import _collate
    root, lists, names, labels = make_voc_tree(tmp_path, n=6)
    ds = VOC12ClsDatasetNew(root_dir=root, name_list_dir=lists, crop_size=64)
    got = []
    for nm, images, params, lab in DataLoader(ds, batch_size=3, num_workers=2, collate_fn=_collate, drop_last=True):
        assert len(nm) == len(images) == len(params) == 3 and lab.shape == (3, 20)
        for im, p in zip(images, params):
            assert im.dtype == np.uint8 and im.shape == (p["h"], p["w"], 3) and p["img_box"].dtype == np.int16
            assert 0 <= p["op"] < 9 and 1 <= p["magnitude"] <= 9 and p["new_h"] == int(p["new_h"]) > 0
        got += nm
    assert got == names
