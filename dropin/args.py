"""alias: args.py -> cosa_amd.args (VOC12 defaults)"""
from cosa_amd.args import get_parser, handle_defaults, str2bool  # noqa: F401
