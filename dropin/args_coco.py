"""alias: args_coco.py -> cosa_amd.args (the COCO defaults are chosen by --dataset COCO inside handle_defaults)"""
from cosa_amd.args import get_parser, handle_defaults, str2bool  # noqa: F401
