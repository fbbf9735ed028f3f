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
        _LIB.bsgpu_time_eval_ms.restype = ctypes.c_double
        _LIB.bsgpu_time_eval_ms.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        _LIB.bsgpu_eval_bytes.restype = ctypes.c_int64
        _LIB.bsgpu_eval_bytes.argtypes = [ctypes.c_void_p]
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


def preintegrate(sample_start, t, w, a, t_end, bg, ba, cov_w, cov_a, cov_bg, cov_ba, info_weight=1.0, device=0):
    """bs_common::PreIntegrator::Integrate for a batch of intervals on the device -> (n, 287) BSGPU_F_IMU_DELTA consts."""
    import numpy as np
    f64 = lambda x: np.ascontiguousarray(x, np.float64)
    ss = np.ascontiguousarray(sample_start, np.int32)
    n = ss.size - 1
    t, w, a, t_end, bg, ba = f64(t), f64(w), f64(a), f64(t_end), f64(bg), f64(ba)
    covs = [f64(np.asarray(c, float) * (np.eye(3) if np.ndim(c) == 0 else 1.0)) for c in (cov_w, cov_a, cov_bg, cov_ba)]
    out = np.empty((n, 287))
    fn = lib().bsgpu_preintegrate
    fn.argtypes = [ctypes.c_int, ctypes.c_int32] + [ctypes.c_void_p] * 11 + [ctypes.c_double, ctypes.c_void_p]
    rc = fn(device, n, ss.ctypes.data, t.ctypes.data, w.ctypes.data, a.ctypes.data, t_end.ctypes.data, bg.ctypes.data, ba.ctypes.data,
            covs[0].ctypes.data, covs[1].ctypes.data, covs[2].ctypes.data, covs[3].ctypes.data, float(info_weight), out.ctypes.data)
    if rc != 0:
        raise capi.SolverError(rc, "bsgpu_preintegrate failed")
    return out


class GpuSolver(capi.Solver):
    """One bsgpu context on HIP device `device`."""

    def __init__(self, device=0):
        super().__init__(lib(), "bsgpu_", device)

    def reprojection_errors(self, n):
        """Pixel error of the n reprojection factors (REPROJ then REPROJ_ONLINE_CALIB, insertion order) at the current values."""
        import numpy as np
        out = np.zeros(n)
        fn = lib().bsgpu_reprojection_errors
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(fn(self._ctx, out.ctypes.data))
        return out

    def triangulate(self, track_start, q_block, p_block, pixels, camera=0, truncate_pixels=True, max_dist=-1.0, max_reproj=-1.0):
        """bsgpu_triangulate: DLT triangulation of a batch of feature tracks at the current values -> (points (n,3), status (n,))."""
        import numpy as np
        ts = np.ascontiguousarray(track_start, np.int32)
        qb, pb = np.ascontiguousarray(q_block, np.int32), np.ascontiguousarray(p_block, np.int32)
        px = np.ascontiguousarray(pixels, np.float64)
        n = ts.size - 1
        pts, st = np.zeros((n, 3)), np.zeros(n, np.int32)
        fn = lib().bsgpu_triangulate
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                                                                  ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(fn(self._ctx, n, ts.ctypes.data, qb.ctypes.data, pb.ctypes.data, px.ctypes.data, int(camera), int(bool(truncate_pixels)),
                     float(max_dist), float(max_reproj), pts.ctypes.data, st.ctypes.data))
        return pts, st

    @staticmethod
    def batch_stats():
        """(windows solved by the batched launches of bsgpu_solve_batch so far in this process, rounds = sets of launches they took)."""
        w, r = ctypes.c_int64(0), ctypes.c_int64(0)
        lib().bsgpu_batch_stats(ctypes.byref(w), ctypes.byref(r))
        return w.value, r.value

    def time_reproj_jacobian_ms(self, reps=20):
        ms = lib().bsgpu_time_reproj_jacobian_ms(self._ctx, int(reps))
        if ms < 0:
            raise capi.SolverError(capi.ERR_DEVICE, "bsgpu_time_reproj_jacobian_ms failed")
        return ms

    def time_eval_ms(self, reps=20):
        """Mean milliseconds of one evaluation of residuals + Jacobians of every factor type (HIP events on the solver's stream)."""
        ms = lib().bsgpu_time_eval_ms(self._ctx, int(reps))
        if ms < 0:
            raise capi.SolverError(capi.ERR_DEVICE, "bsgpu_time_eval_ms failed")
        return ms

    def eval_bytes(self):
        return lib().bsgpu_eval_bytes(self._ctx)

    def bsr_info(self):
        """(block rows, non-zero 3x3 blocks) of the block-sparse PCG path; (0, 0) on the dense Schur path."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self._chk(lib().bsgpu_bsr_info(ctypes.c_void_p(self._ctx), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    PHASES = ("eval_reproj", "eval_other", "landmark", "pairs", "assemble_other", "factor", "backsolve", "backsub", "candidate")

    def profile_step(self, options=None, reps=10):
        """bsgpu_profile_step: {phase: (mean ms, algorithmic work)} of a full LM step timed in situ with HIP events (bytes; flops for
        'factor')."""
        import numpy as np
        opt = options if options is not None else self.options_default()
        ms, work = np.zeros(len(self.PHASES)), np.zeros(len(self.PHASES))
        fn = lib().bsgpu_profile_step
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(fn(self._ctx, ctypes.byref(opt), int(reps), ms.ctypes.data, work.ctypes.data))
        return {name: (float(ms[i]), float(work[i])) for i, name in enumerate(self.PHASES)}

    def set_plan_preference(self, throughput):
        """bsgpu_set_plan_preference: BSGPU_PLAN_THROUGHPUT for a window that is one of many in bsgpu_solve_batch, BSGPU_PLAN_LATENCY (default) otherwise."""
        fn = lib().bsgpu_set_plan_preference
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        self._chk(fn(self._ctx, 1 if throughput else 0))

    def plan_info(self):
        """(independent sub-chains, schedule steps, 64-wide tiles) of the reduced system's Cholesky plan."""
        a, b, c = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        self.finalize()
        lib().bsgpu_plan_info(ctypes.c_void_p(self._ctx), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def reproj_jacobian_bytes(self):
        return lib().bsgpu_reproj_jacobian_bytes(self._ctx)
