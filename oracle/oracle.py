"""TEST INFRASTRUCTURE — loader of the CPU oracle (oracle/libbs_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from beam_slam_amd import capi  # noqa: E402

_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libbs_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("bs_oracle.cpp", "functors.h", "jet.h")] + \
           [os.path.join(os.path.dirname(_HERE), "include", "bsgpu.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.bso_quat_plus.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3
        _LIB.bso_plus_jacobian.argtypes = [ctypes.POINTER(ctypes.c_double)] * 2
    return _LIB


class Oracle(capi.Solver):
    def __init__(self, threads=None):
        super().__init__(lib(), "bso_")
        if threads:
            lib().bso_set_num_threads(self._ctx, int(threads))

    def set_reproj_mode(self, mode):
        """0 closed-form Jacobian, 1 the reference's forward-difference quaternion Jacobian
        (euclidean_reprojection_function.h:124-143), 2 autodiff twin."""
        lib().bso_set_reproj_mode(ctypes.c_void_p(self._ctx), int(mode))

    @staticmethod
    def max_threads():
        return lib().bso_max_threads()
