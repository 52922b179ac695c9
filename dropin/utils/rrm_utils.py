"""alias: utils/rrm_utils.py -> cosa_amd.utils.rrm_utils"""
from cosa_amd.utils.rrm_utils import *  # noqa: F401,F403
import cosa_amd.utils.rrm_utils as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
