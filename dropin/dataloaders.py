"""alias: the reference's `dataloaders` package -> cosa_amd.dataloaders (+ build_dataloader with the reference's return convention)"""
from cosa_amd.dataloaders import *  # noqa: F401,F403
from cosa_amd.dataloaders import build_test_loader, build_train_loader, build_val_loader


def build_dataloader(args, is_train=True):
    """dataloaders/__init__.py:93-124: (train_loader, val_loader) for training, the TEST loader alone otherwise (build_test_dataset:
    VOC12 `val`, COCO the full `val` split)"""
    if not is_train:
        return build_test_loader(args)
    return build_train_loader(args, device=getattr(args, "device", "cuda"), num_workers=getattr(args, "num_workers", 4)), build_val_loader(args)
