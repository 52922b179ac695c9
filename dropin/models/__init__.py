"""alias: the reference's `models` package -> cosa_amd.models"""
import importlib

from cosa_amd.models import *  # noqa: F401,F403
from cosa_amd.models import LargeFOV, VITNetwork, build_model  # noqa: F401

PAR = importlib.import_module(__name__ + ".PAR")        # `models.PAR.PAR` as in the reference (the star import bound the class to this name)
