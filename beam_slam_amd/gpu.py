"""Loader of the product library libbsgpu.so (HIP kernels + C-ABI of include/bsgpu.h).

There is deliberately no fallback: if the library has not been built (``__graft_entry__.build()`` or
``make -C beam_slam_amd/csrc``) or no HIP device is usable, this module raises.
"""
import ctypes
import os

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbsgpu.so")
_LIB = None


class GpuLibraryMissing(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GpuLibraryMissing(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  beam_slam_amd has no CPU fallback.")
        _LIB = ctypes.CDLL(LIB_PATH)
        _LIB.bsgpu_time_reproj_jacobian_ms.restype = ctypes.c_double
        _LIB.bsgpu_time_reproj_jacobian_ms.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        _LIB.bsgpu_reproj_jacobian_bytes.restype = ctypes.c_int64
        _LIB.bsgpu_reproj_jacobian_bytes.argtypes = [ctypes.c_void_p]
        _LIB.bsgpu_dense_solve.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    return _LIB


def dense_solve(A, b, device=0, max_chains=4):
    """A x = b (SPD) through the reduced-camera-system kernels; returns (x, milliseconds)."""
    import numpy as np
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.empty_like(b)
    ms = ctypes.c_double(0.0)
    rc = lib().bsgpu_dense_solve(device, A.shape[0], A.ctypes.data, b.ctypes.data, x.ctypes.data, int(max_chains),
                                 ctypes.byref(ms))
    if rc != 0:
        raise capi.SolverError(rc, "bsgpu_dense_solve failed")
    return x, ms.value


class GpuSolver(capi.Solver):
    """One bsgpu context on HIP device `device`."""

    def __init__(self, device=0):
        super().__init__(lib(), "bsgpu_", device)

    def time_reproj_jacobian_ms(self, reps=20):
        ms = lib().bsgpu_time_reproj_jacobian_ms(self._ctx, int(reps))
        if ms < 0:
            raise capi.SolverError(capi.ERR_DEVICE, "bsgpu_time_reproj_jacobian_ms failed")
        return ms

    def plan_info(self):
        """(independent sub-chains, schedule steps, 64-wide tiles) of the reduced system's Cholesky plan."""
        a, b, c = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        self.finalize()
        lib().bsgpu_plan_info(ctypes.c_void_p(self._ctx), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def reproj_jacobian_bytes(self):
        return lib().bsgpu_reproj_jacobian_bytes(self._ctx)
