"""cosa_amd -- MI355X-native implementation of CoSA's per-iteration training hot path.

Host code mirrors the reference's call surface (models.PAR, models.build_model, utils.seg_helper,
utils.rrm_utils, utils.torch_helper, the `bilateralfilter` module); the compute is hand-written
HIP for gfx950 behind the C ABI in include/cosa_hip.h (libcosa_hip.so).  There is no CPU path.
"""
__version__ = "0.1.0"
