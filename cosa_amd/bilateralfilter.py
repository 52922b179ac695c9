"""Drop-in for the reference's SWIG module `bilateralfilter` (utils/bilateralfilter/bilateralfilter.i).

Same call signatures as the SWIG wrappers (each (ptr,len) pair is one NumPy array; `out(s)` is a
contiguous float32 1-D array written in place):

    bilateralfilter(image, in_, out, H, W, sigmargb, sigmaxy)
    bilateralfilter_batch(images, ins, outs, N, K, H, W, sigmargb, sigmaxy)     # utils/seg_helper.py:887

The arrays are host memory; the filter itself runs on the GPU (host-pointer entry points of
libcosa_hip.so).  The training path does not come through here -- it uses the device-resident
cosa_dense_energy_forward -- this module exists so that reference-side callers keep working.
"""
import ctypes

import numpy as np

from . import _C


def _in(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _inplace(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 1 and a.flags.c_contiguous):
        raise TypeError("out must be a contiguous 1-D float32 NumPy array (INPLACE_ARRAY1)")
    return a


def bilateralfilter(image, in_, out, H, W, sigmargb, sigmaxy):
    image, in_, out = _in(image), _in(in_), _inplace(out)
    _C.lib().bilateralfilter(image.ctypes.data_as(ctypes.c_void_p), image.size, in_.ctypes.data_as(ctypes.c_void_p),
                             in_.size, out.ctypes.data_as(ctypes.c_void_p), out.size, int(H), int(W), float(sigmargb),
                             float(sigmaxy))


def bilateralfilter_batch(images, ins, outs, N, K, H, W, sigmargb, sigmaxy):
    images, ins, outs = _in(images), _in(ins), _inplace(outs)
    _C.lib().bilateralfilter_batch(images.ctypes.data_as(ctypes.c_void_p), images.size,
                                   ins.ctypes.data_as(ctypes.c_void_p), ins.size,
                                   outs.ctypes.data_as(ctypes.c_void_p), outs.size, int(N), int(K), int(H), int(W),
                                   float(sigmargb), float(sigmaxy))
