"""dataloaders -- the training input pipeline (SURVEY f-2): draws on the host in the reference's order, pixels on the device."""
from .augment import OPS, DeviceAugmenter, draw_params  # noqa: F401
from .train_loader import (COCOClsDatasetNew, COCOSegDataset, DeviceTrainLoader, VOC12ClsDatasetNew, VOC12SegDataset,  # noqa: F401
                           build_test_dataset, build_test_loader, build_train_datasetv2, build_train_loader, build_val_dataset,
                           build_val_loader, normalize_img)
