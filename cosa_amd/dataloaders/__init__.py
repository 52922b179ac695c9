"""dataloaders -- the training input pipeline (SURVEY f-2): draws on the host in the reference's order, pixels on the device."""
from .augment import OPS, DeviceAugmenter, draw_params  # noqa: F401
from .train_loader import (COCOClsDatasetNew, DeviceTrainLoader, VOC12ClsDatasetNew, build_train_datasetv2,  # noqa: F401
                           build_train_loader)
