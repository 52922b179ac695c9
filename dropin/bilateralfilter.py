"""alias: the SWIG module `bilateralfilter` (utils/bilateralfilter/bilateralfilter.i) -> the C ABI of libcosa_hip.so"""
from cosa_amd.bilateralfilter import bilateralfilter, bilateralfilter_batch  # noqa: F401
