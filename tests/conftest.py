import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never run without a device; skip them early with a clear reason
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


@pytest.fixture(scope="session")
def oracle_c():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def make_voc_tree():
    def _make(tmp_path, n=5):
        """a tiny VOC-shaped tree: JPEGImages/*.jpg, <lists>/train_aug.txt, <lists>/cls_labels_onehot.npy"""
        from PIL import Image
        rng = np.random.default_rng(0)
        root, lists = tmp_path / "VOC2012", tmp_path / "lists"
        (root / "JPEGImages").mkdir(parents=True)
        lists.mkdir()
        names, labels = [], {}
        for i in range(n):
            name = f"2007_{i:06d}"
            h, w = int(rng.integers(60, 120)), int(rng.integers(60, 120))
            small = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
            Image.fromarray(small).resize((w, h), Image.BICUBIC).save(root / "JPEGImages" / (name + ".jpg"), quality=92)
            names.append(name)
            lab = np.zeros(20, np.uint8)
            lab[rng.choice(20, size=int(rng.integers(1, 3)), replace=False)] = 1
            labels[name] = lab
        np.savetxt(lists / "train_aug.txt", np.array(names), fmt="%s")
        np.save(lists / "cls_labels_onehot.npy", labels, allow_pickle=True)
        return str(root), str(lists), names, labels
    return _make
