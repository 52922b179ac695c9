"""alias: the reference's `utils` package -> cosa_amd.utils"""
from . import evaluation, misc, rrm_utils, seg_helper, torch_helper  # noqa: F401
