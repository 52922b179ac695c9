"""alias: utils/evaluation.py -> cosa_amd.utils.evaluation"""
from cosa_amd.utils.evaluation import *  # noqa: F401,F403
import cosa_amd.utils.evaluation as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
