"""alias: models/PAR.py -> cosa_amd.models.PAR"""
from cosa_amd.models.PAR import *  # noqa: F401,F403
from cosa_amd.models.PAR import PAR  # noqa: F401
