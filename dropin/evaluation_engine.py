"""alias: evaluation_engine.py -> cosa_amd.evaluation_engine"""
from cosa_amd.evaluation_engine import *  # noqa: F401,F403
from cosa_amd.evaluation_engine import evaluate  # noqa: F401
