"""alias: utils/seg_helper.py -> cosa_amd.utils.seg_helper"""
from cosa_amd.utils.seg_helper import *  # noqa: F401,F403
import cosa_amd.utils.seg_helper as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
