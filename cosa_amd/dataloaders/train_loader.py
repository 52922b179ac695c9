"""Training loader with the reference's per-batch contract, pixels on the device.

Reference: dataloaders/__init__.py:62-103 (build_train_datasetv2 / build_dataloader), dataloaders/voc.py:219-305
(VOC12ClsDatasetNew), dataloaders/coco.py:70-140 (COCOClsDatasetNew).  The reference's worker decodes the JPEG and runs the whole
Pillow pipeline per image; here the workers decode and DRAW (same generators, same order, same place: inside __getitem__ in
the worker process), and the batch is augmented on the GPU by DeviceAugmenter.  Iterating yields what main.py:114 unpacks:

    img_name (list of str), wimg [b,3,S,S], simg [b,3,S,S] (device float32), cls_label [b,C] (uint8 -> as stored), img_box [b,4]
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from .augment import DeviceAugmenter, draw_params


def load_img_name_list(img_name_list_path):
    return np.loadtxt(img_name_list_path, dtype=str)


def load_cls_label_list(name_list_dir):
    return np.load(os.path.join(name_list_dir, 'cls_labels_onehot.npy'), allow_pickle=True).item()


class _ClsDatasetNew(Dataset):
    """image list + one-hot labels; __getitem__ -> (img_name, decoded uint8 image, draws, cls_label)"""

    def __init__(self, img_dir, name_list_dir, split, rescale_range, crop_size, num_classes):
        super().__init__()
        self.img_dir = img_dir
        self.name_list = load_img_name_list(os.path.join(name_list_dir, split + '.txt'))
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)
        self.rescale_range = rescale_range
        self.crop_size = crop_size
        self.num_classes = num_classes

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, idx):
        img_name = str(self.name_list[idx])
        image = np.asarray(Image.open(os.path.join(self.img_dir, img_name + '.jpg')).convert('RGB'))
        params = draw_params(image.shape[0], image.shape[1], crop_size=self.crop_size, scale_range=self.rescale_range)
        return img_name, image, params, self.label_list[img_name]


class VOC12ClsDatasetNew(_ClsDatasetNew):
    """dataloaders/voc.py:219-305"""

    def __init__(self, root_dir, name_list_dir=None, split='train_aug', stage='train', rescale_range=[0.5, 2.0], crop_size=448,
                 img_fliplr=True, ignore_index=255, num_classes=21, aug=True, **kwargs):
        assert aug
        super().__init__(os.path.join(root_dir, 'JPEGImages_test' if split == 'test' else 'JPEGImages'), name_list_dir, split,
                         rescale_range, crop_size, num_classes)


class COCOClsDatasetNew(_ClsDatasetNew):
    """dataloaders/coco.py:70-140"""

    def __init__(self, root_dir, name_list_dir=None, split='train', stage='train', rescale_range=[0.5, 2.0], crop_size=448,
                 img_fliplr=True, ignore_index=255, num_classes=81, aug=True, **kwargs):
        assert aug
        splitclean = 'val' if split[:3] == 'val' else split
        super().__init__(os.path.join(root_dir, splitclean + '2014'), name_list_dir, split, rescale_range, crop_size, num_classes)


def _collate(samples):
    names, images, params, labels = zip(*samples)
    return list(names), list(images), list(params), torch.from_numpy(np.stack([np.asarray(l) for l in labels]))


class DeviceTrainLoader:
    """DataLoader of (decode + draws) -> DeviceAugmenter.  Same knobs as the reference's DataLoader (batch_size, sampler,
    drop_last=True, pin_memory irrelevant: the raw bytes are staged pinned by the augmenter); num_workers is free to be > 1
    because a worker now costs one JPEG decode per image instead of the Pillow pipeline."""

    def __init__(self, dataset, batch_size, device="cuda", sampler=None, num_workers=1, shuffle=False):
        self.loader = DataLoader(dataset=dataset, batch_size=batch_size, num_workers=num_workers, drop_last=True, sampler=sampler,
                                 shuffle=shuffle and sampler is None, collate_fn=_collate)
        self.sampler = sampler
        self.dataset = dataset
        self.augment = DeviceAugmenter(dataset.crop_size, device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for names, images, params, labels in self.loader:
            wimg, simg, img_box = self.augment(images, params)
            yield names, wimg, simg, labels, img_box


def build_train_datasetv2(args):
    """dataloaders/__init__.py:62-91"""
    name_dir = getattr(args, "name_list_dir", None)
    if args.dataset == 'VOC12':
        return VOC12ClsDatasetNew(root_dir=args.voc12_root, name_list_dir=name_dir or './dataloaders/voc/', split='train_aug',
                                  stage='train', aug=True, rescale_range=args.scales, crop_size=args.crop_size, img_fliplr=True,
                                  ignore_index=args.ignore_index, num_classes=args.num_classes)
    if args.dataset == 'COCO':
        return COCOClsDatasetNew(root_dir=args.coco_root, name_list_dir=name_dir or './dataloaders/coco/', split='train', stage='train',
                                 aug=True, rescale_range=args.scales, crop_size=args.crop_size, img_fliplr=True,
                                 ignore_index=args.ignore_index, num_classes=args.num_classes)
    raise NotImplementedError


def build_train_loader(args, device="cuda", num_workers=4):
    """the training half of dataloaders/__init__.py:95-103 (DistributedSampler(shuffle=True), batch_size, drop_last)"""
    dataset = build_train_datasetv2(args)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True) if torch.distributed.is_initialized() else None
    return DeviceTrainLoader(dataset, args.batch_size, device=device, sampler=sampler, num_workers=num_workers, shuffle=True)


# ---- validation / test datasets (dataloaders/voc.py:306-368, coco.py:142-200; the aug=False path the reference uses) -------------
def normalize_img(img, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
    """dataloaders/transforms.py:42-50: (uint8 - mean) / std evaluated in float64, stored as float32, HWC"""
    return ((np.asarray(img).astype(np.float64) - np.asarray(mean)) / np.asarray(std)).astype(np.float32)


class _SegDataset(Dataset):
    def __init__(self, img_dir, label_dir, name_list_dir, split, stage):
        super().__init__()
        self.img_dir, self.label_dir, self.stage, self.split = img_dir, label_dir, stage, split
        self.name_list = load_img_name_list(os.path.join(name_list_dir, split + '.txt'))
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, idx):
        img_name = str(self.name_list[idx])
        image = np.asarray(Image.open(os.path.join(self.img_dir, img_name + '.jpg')).convert('RGB'))
        if self.stage == "test":
            label, cls_label = image[:, :, 0], 0
        else:
            label = np.asarray(Image.open(os.path.join(self.label_dir, img_name + '.png')))      # class indices (L or P mode)
            cls_label = self.label_list[img_name]
        return img_name, np.transpose(normalize_img(image), (2, 0, 1)), label, cls_label


class VOC12SegDataset(_SegDataset):
    """dataloaders/voc.py:306-368 with aug=False (how build_val_dataset / build_test_dataset construct it)"""

    def __init__(self, root_dir=None, name_list_dir=None, split='train', stage='train', aug=False, ignore_index=255, **kwargs):
        if aug:
            raise NotImplementedError("VOC12SegDataset: the augmented variant is not part of the reference's training or evaluation runs")
        super().__init__(os.path.join(root_dir, 'JPEGImages_test' if split == 'test' else 'JPEGImages'),
                         os.path.join(root_dir, 'SegmentationClassAug'), name_list_dir, split, stage)


class COCOSegDataset(_SegDataset):
    """dataloaders/coco.py (COCOSegDataset) with aug=False"""

    def __init__(self, root_dir=None, name_list_dir=None, split='train', stage='train', aug=False, ignore_index=255, **kwargs):
        if aug:
            raise NotImplementedError("COCOSegDataset: the augmented variant is not part of the reference's training or evaluation runs")
        splitclean = 'val' if split[:3] == 'val' else split
        super().__init__(os.path.join(root_dir, splitclean + '2014'), os.path.join(root_dir, f'SegmentationClass/{splitclean}2014'),
                         name_list_dir, split, stage)


def build_val_dataset(args):
    """dataloaders/__init__.py:10-33"""
    name_dir = getattr(args, "name_list_dir", None)
    if args.dataset == 'VOC12':
        return VOC12SegDataset(root_dir=args.voc12_root, name_list_dir=name_dir or './dataloaders/voc/', split='val', stage='val', aug=False,
                               ignore_index=args.ignore_index, num_classes=args.num_classes)
    if args.dataset == 'COCO':
        return COCOSegDataset(root_dir=args.coco_root, name_list_dir=name_dir or './dataloaders/coco/',
                              split='val_part' if not getattr(args, "valfull", False) else 'val', stage='val', aug=False,
                              ignore_index=args.ignore_index, num_classes=args.num_classes)
    raise NotImplementedError


def build_test_dataset(args):
    """dataloaders/__init__.py:34-58: what `finaleval` evaluates on -- VOC12 `val`, COCO the FULL `val` split (build_val_dataset takes
    `val_part` unless --valfull)"""
    name_dir = getattr(args, "name_list_dir", None)
    if args.dataset == 'VOC12':
        return VOC12SegDataset(root_dir=args.voc12_root, name_list_dir=name_dir or './dataloaders/voc/', split='val', stage='val', aug=False,
                               ignore_index=args.ignore_index, num_classes=args.num_classes)
    if args.dataset == 'COCO':
        return COCOSegDataset(root_dir=args.coco_root, name_list_dir=name_dir or './dataloaders/coco/', split='val', stage='val', aug=False,
                              ignore_index=args.ignore_index, num_classes=args.num_classes)
    raise NotImplementedError


def _eval_loader(dataset, num_workers):
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=False, drop_last=True) \
        if torch.distributed.is_initialized() else None
    return DataLoader(dataset=dataset, batch_size=1, shuffle=False, num_workers=num_workers, pin_memory=False, sampler=sampler,
                      drop_last=False)


def build_test_loader(args, num_workers=1):
    """the `is_train=False` branch of dataloaders/__init__.py:114-124"""
    return _eval_loader(build_test_dataset(args), num_workers)


def build_val_loader(args, num_workers=1):
    """the validation half of dataloaders/__init__.py:104-113: batch 1, DistributedSampler(shuffle=False, drop_last=True)"""
    dataset = build_val_dataset(args)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=False, drop_last=True) \
        if torch.distributed.is_initialized() else None
    return DataLoader(dataset=dataset, batch_size=1, shuffle=False, num_workers=num_workers, pin_memory=False, sampler=sampler,
                      drop_last=False)
