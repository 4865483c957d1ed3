"""Development scripts that flip kernel-variant switches need the -DTGP_DEV build of the library (the product library has no
such switches).  use_dev_lib() builds libtgpose_hip_dev.so if it is missing, points tgpose_amd at it for this process and
returns the ctypes handle that carries the tgp_debug_* setters."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def use_dev_lib():
    from tgpose_amd import _lib, build
    if _lib._lib is not None:
        raise RuntimeError("use_dev_lib() must run before the first library call")
    if not os.path.exists(build.DEV_LIB) or os.path.getmtime(build.DEV_LIB) < os.path.getmtime(build.LIB):
        build.build(dev=True, verbose=False)
    _lib.LIB_PATH = build.DEV_LIB
    return ctypes.CDLL(build.DEV_LIB)
