"""dataloaders -- the training input pipeline (SURVEY f-2): draws on the host in the reference's order, pixels on the device."""
from .augment import OPS, DeviceAugmenter, draw_params  # noqa: F401
