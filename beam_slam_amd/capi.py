"""ctypes view of include/bsgpu.h.

The same declarations serve the product library (prefix ``bsgpu_``, loaded by
:mod:`beam_slam_amd.gpu`) and — in tests only — the CPU oracle (prefix ``bso_``), because the
oracle deliberately exports the same entry points.  Nothing in this module computes anything.
"""
import ctypes as C

import numpy as np

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NUMERIC = -1, -2, -3, -4

MANIFOLD_EUCLIDEAN, MANIFOLD_QUAT_RIGHT = 0, 1
LOSS_TRIVIAL, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2

F_REPROJ = 0
F_REPROJ_ONLINE_CALIB = 1
F_IMU_DELTA = 2
F_IMU_PRIOR = 3
F_RELPOSE_EXT = 4
F_RELPOSE = 5
F_ABSPOSE = 6
F_ABS_VEC3 = 7
F_REL_VEC3 = 8
F_GRAVITY = 9
F_IDP_REPROJ = 10
F_IDP_REPROJ_UNARY = 11
F_NUM_TYPES = 12

LINEAR_AUTO, LINEAR_SCHUR_CHOLESKY, LINEAR_PCG, LINEAR_SCHUR_PCG = 0, 1, 2, 3
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("linear_solver_type", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("max_solver_time_in_seconds", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("pcg_max_iterations", C.c_int32),
        ("reserved0", C.c_int32),
        ("pcg_tolerance", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32),
        ("is_solution_usable", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("num_parameters_tangent", C.c_int32),
        ("num_residuals", C.c_int32),
        ("linear_solver_used", C.c_int32),
        ("num_linear_solves", C.c_int32),
        ("num_inner_iterations", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("fixed_cost", C.c_double),
        ("total_time_in_seconds", C.c_double),
        ("device_time_in_seconds", C.c_double),
        ("time_eval_seconds", C.c_double),
        ("time_assemble_seconds", C.c_double),
        ("time_linear_solve_seconds", C.c_double),
        ("message", C.c_char * 160),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("reserved0", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("gradient_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("model_cost_change", C.c_double),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("R_cam_baselink", C.c_double * 9),
        ("t_cam_baselink", C.c_double * 3),
    ]


#: every symbol include/bsgpu.h declares (without prefix); tests check the library exports them all
SYMBOLS = [
    "nidx", "nconst", "nres", "options_default", "options_vio", "create", "create_error", "destroy",
    "last_error", "abi_version", "clear", "set_blocks", "set_values", "set_cameras", "add_factors", "add_factors_indirect", "sync_factors_indirect",
    "add_marginal",
    "finalize", "solve", "get_blocks", "reset_values", "num_iterations_recorded", "get_iteration",
    "evaluate", "num_residuals", "num_parameters_tangent", "tangent_offset", "covariance", "marginalize", "get_marginal",
    "reprojection_errors", "preintegrate", "triangulate", "time_reproj_jacobian_ms", "reproj_jacobian_bytes", "dense_solve", "plan_info",
    "profile_step", "time_eval_ms", "eval_bytes", "bsr_info", "covariance_joint", "update_marginal", "solve_batch", "batch_stats", "set_plan_preference",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class SolverError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class Solver:
    """Thin handle over one context of a library exporting the bsgpu.h entry points."""

    def __init__(self, lib, prefix, device=0):
        self._lib, self._p = lib, prefix
        f = self._f
        f("create").restype = C.c_void_p
        f("create").argtypes = [C.c_int]
        f("destroy").argtypes = [C.c_void_p]
        f("last_error").restype = C.c_char_p
        f("last_error").argtypes = [C.c_void_p]
        f("set_blocks").argtypes = [C.c_void_p, C.c_int32, _dp, _ip, _bp, _bp, _bp]
        f("set_values").argtypes = [C.c_void_p, _dp, C.c_int64]
        f("set_cameras").argtypes = [C.c_void_p, C.c_int32, C.POINTER(Camera)]
        f("add_factors").argtypes = [C.c_void_p, C.c_int32, C.c_int32, _ip, _dp, _ip, _dp]
        f("add_marginal").argtypes = [C.c_void_p, C.c_int32, _ip, C.c_int32, _dp, _dp, _dp]
        f("marginalize").argtypes = [C.c_void_p, C.c_int32, _ip, _ip, _ip, _ip]
        f("get_marginal").argtypes = [C.c_void_p, _ip, _dp, _dp, _dp]
        f("finalize").argtypes = [C.c_void_p]
        f("clear").argtypes = [C.c_void_p]
        f("solve").argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Summary)]
        f("get_blocks").argtypes = [C.c_void_p, _dp, C.c_int64]
        f("reset_values").argtypes = [C.c_void_p]
        f("num_iterations_recorded").argtypes = [C.c_void_p]
        f("get_iteration").argtypes = [C.c_void_p, C.c_int32, C.POINTER(Iteration)]
        f("evaluate").argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        f("num_residuals").argtypes = [C.c_void_p]
        f("num_parameters_tangent").argtypes = [C.c_void_p]
        f("tangent_offset").argtypes = [C.c_void_p, C.c_int32]
        f("covariance").argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp]
        f("options_default").argtypes = [C.POINTER(Options)]
        f("options_vio").argtypes = [C.POINTER(Options)]
        self._ctx = f("create")(device)
        if not self._ctx:
            why = ""
            try:
                ce = f("create_error")
                ce.restype = C.c_char_p
                why = ce().decode()
            except AttributeError:
                pass
            raise SolverError(ERR_DEVICE, f"{prefix}create({device}) failed: {why}")
        self._nvalues = 0

    def _f(self, name):
        return getattr(self._lib, self._p + name)

    def has(self, name):
        """Whether the library behind this handle exports the entry point (the test oracle has only what its tests need)."""
        return hasattr(self._lib, self._p + name)

    def _chk(self, rc):
        if rc != OK:
            raise SolverError(rc, self._f("last_error")(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._f("destroy")(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- options ---------------------------------------------------------------------------
    def options_default(self):
        o = Options()
        self._f("options_default")(C.byref(o))
        return o

    def options_vio(self):
        o = Options()
        self._f("options_vio")(C.byref(o))
        return o

    # -- problem ---------------------------------------------------------------------------
    def clear(self):
        self._chk(self._f("clear")(self._ctx))

    def set_blocks(self, values, offset, size, manifold, is_const):
        values = np.ascontiguousarray(values, np.float64)
        offset = np.ascontiguousarray(offset, np.int32)
        size = np.ascontiguousarray(size, np.uint8)
        manifold = np.ascontiguousarray(manifold, np.uint8)
        is_const = np.ascontiguousarray(is_const, np.uint8)
        self._nvalues = values.size
        self._chk(self._f("set_blocks")(self._ctx, offset.size, _ptr(values, _dp), _ptr(offset, _ip),
                                        _ptr(size, _bp), _ptr(manifold, _bp), _ptr(is_const, _bp)))

    def set_values(self, values):
        values = np.ascontiguousarray(values, np.float64)
        self._chk(self._f("set_values")(self._ctx, _ptr(values, _dp), values.size))

    def set_cameras(self, cams):
        arr = (Camera * len(cams))(*cams)
        self._chk(self._f("set_cameras")(self._ctx, len(cams), arr))

    def add_factors(self, ftype, block_idx, consts, loss_kind=None, loss_a=None):
        block_idx = np.ascontiguousarray(block_idx, np.int32)
        consts = np.ascontiguousarray(consts, np.float64)
        n = block_idx.shape[0] if block_idx.ndim == 2 else 0
        if loss_kind is not None:
            loss_kind = np.ascontiguousarray(np.broadcast_to(loss_kind, (n,)), np.int32)
        if loss_a is not None:
            loss_a = np.ascontiguousarray(np.broadcast_to(loss_a, (n,)), np.float64)
        self._chk(self._f("add_factors")(self._ctx, ftype, n, _ptr(block_idx, _ip), _ptr(consts, _dp),
                                         _ptr(loss_kind, _ip), _ptr(loss_a, _dp)))

    def add_factors_indirect(self, ftype, slot_idx, slot_to_block, consts, loss_kind=None, loss_a=None):
        """add_factors for a caller that keeps its tables across solves: the block columns of slot_idx hold caller-side variable
        slots, translated through slot_to_block on the way in (camera columns are not translated)."""
        slot_idx = np.ascontiguousarray(slot_idx, np.int32)
        slot_to_block = np.ascontiguousarray(slot_to_block, np.int32)
        consts = np.ascontiguousarray(consts, np.float64)
        n = slot_idx.shape[0] if slot_idx.ndim == 2 else 0
        if loss_kind is not None:
            loss_kind = np.ascontiguousarray(np.broadcast_to(loss_kind, (n,)), np.int32)
        if loss_a is not None:
            loss_a = np.ascontiguousarray(np.broadcast_to(loss_a, (n,)), np.float64)
        fn = self._f("add_factors_indirect")
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _ip, C.c_int32, _ip, _dp, _ip, _dp]
        fn.restype = C.c_int
        self._chk(fn(self._ctx, ftype, n, _ptr(slot_idx, _ip), slot_to_block.size, _ptr(slot_to_block, _ip), _ptr(consts, _dp),
                     _ptr(loss_kind, _ip), _ptr(loss_a, _dp)))

    def sync_factors_indirect(self, ftype, slot_idx, slot_to_block, consts, loss_kind=None, loss_a=None, changed_rows=None):
        """add_factors_indirect for a type's WHOLE table plus the rows that differ from the table of the previous call for the type
        (None: no promise, the table is read whole).  The back-end keeps slot-named host / device copies of the reprojection table
        across clear() and patches them (include/bsgpu.h)."""
        slot_idx = np.ascontiguousarray(slot_idx, np.int32)
        slot_to_block = np.ascontiguousarray(slot_to_block, np.int32)
        consts = np.ascontiguousarray(consts, np.float64)
        n = slot_idx.shape[0] if slot_idx.ndim == 2 else 0
        if loss_kind is not None:
            loss_kind = np.ascontiguousarray(np.broadcast_to(loss_kind, (n,)), np.int32)
        if loss_a is not None:
            loss_a = np.ascontiguousarray(np.broadcast_to(loss_a, (n,)), np.float64)
        if changed_rows is not None:
            changed_rows = np.ascontiguousarray(changed_rows, np.int32)
        fn = self._f("sync_factors_indirect")
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _ip, C.c_int32, _ip, _dp, _ip, _dp, C.c_int32, _ip]
        fn.restype = C.c_int
        self._chk(fn(self._ctx, ftype, n, _ptr(slot_idx, _ip), slot_to_block.size, _ptr(slot_to_block, _ip), _ptr(consts, _dp),
                     _ptr(loss_kind, _ip), _ptr(loss_a, _dp), -1 if changed_rows is None else changed_rows.size,
                     _ptr(changed_rows, _ip) if changed_rows is not None and changed_rows.size else None))

    def add_marginal(self, blocks, A, b, xbar):
        """fuse_constraints::MarginalConstraint: r = b + sum_i A_i (x_i [-] xbar_i)."""
        blocks = np.ascontiguousarray(blocks, np.int32)
        A = np.ascontiguousarray(np.atleast_2d(A), np.float64)
        b = np.ascontiguousarray(b, np.float64)
        xbar = np.ascontiguousarray(xbar, np.float64)
        assert A.shape[0] == b.size
        self._chk(self._f("add_marginal")(self._ctx, blocks.size, _ptr(blocks, _ip), A.shape[0], _ptr(A, _dp), _ptr(b, _dp),
                                          _ptr(xbar, _dp)))

    def update_marginal(self, index, A, b, xbar):
        """Replaces the payload of the index-th dense prior in place (bsgpu_update_marginal): no re-finalize."""
        A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64); xbar = np.ascontiguousarray(xbar, np.float64)
        fn = self._f("update_marginal")
        assert A.ndim == 2 and b.size == A.shape[0]
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]
        self._chk(fn(self._ctx, int(index), A.shape[0], A.shape[1], xbar.size, A.ctypes.data_as(_dp), b.ctypes.data_as(_dp), xbar.ctypes.data_as(_dp)))

    def covariance_joint(self, blocks, tangent_sizes):
        """Joint marginal covariance (D x D, D = sum of tangent sizes <= 64) of several pose-side blocks (bsgpu_covariance_joint)."""
        bl = np.ascontiguousarray(blocks, np.int32)
        D = int(np.sum(tangent_sizes))
        out = np.zeros((D, D))
        fn = self._f("covariance_joint")
        fn.argtypes = [C.c_void_p, C.c_int32, _ip, _dp]
        self._chk(fn(self._ctx, int(bl.size), bl.ctypes.data_as(_ip), out.ctypes.data_as(_dp)))
        return out

    def marginalize(self, blocks, sizes):
        """fuse_constraints::marginalizeVariables at the current values: returns (kept_blocks, A, b, xbar), the payload
        of the MarginalConstraint that replaces every factor touching `blocks`.  `sizes` = ambient block sizes."""
        blocks = np.ascontiguousarray(blocks, np.int32)
        nk, nr, nc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._chk(self._f("marginalize")(self._ctx, blocks.size, _ptr(blocks, _ip), C.byref(nk), C.byref(nr), C.byref(nc)))
        kept = np.zeros(nk.value, np.int32)
        self._chk(self._f("get_marginal")(self._ctx, _ptr(kept, _ip), None, None, None))
        A = np.zeros((nr.value, nc.value)); b = np.zeros(nr.value)
        xbar = np.zeros(int(sum(sizes[k] for k in kept)))
        self._chk(self._f("get_marginal")(self._ctx, _ptr(kept, _ip), _ptr(A, _dp), _ptr(b, _dp), _ptr(xbar, _dp)))
        return kept, A, b, xbar

    def finalize(self):
        self._chk(self._f("finalize")(self._ctx))

    # -- solve -----------------------------------------------------------------------------
    def solve(self, options=None):
        o = options if options is not None else self.options_default()
        s = Summary()
        self._chk(self._f("solve")(self._ctx, C.byref(o), C.byref(s)))
        return s

    @staticmethod
    def solve_batch(solvers, options=None):
        """bsgpu_solve_batch: the solvers' windows at once (landmark windows advance together, one set of launches per LM iteration;
        any other window on a library thread of its own); returns their summaries."""
        n = len(solvers)
        first = solvers[0]
        opts = options if isinstance(options, (list, tuple)) else [options if options is not None else first.options_default()]
        oa = (Options * len(opts))(*opts)
        sa = (Summary * n)()
        ca = (C.c_void_p * n)(*[sv._ctx for sv in solvers])
        fn = first._f("solve_batch")
        fn.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Options), C.c_int32, C.POINTER(Summary)]
        rc = fn(ca, n, oa, 1 if len(opts) > 1 else 0, sa)
        if rc != OK:
            for sv in solvers:
                msg = sv._f("last_error")(sv._ctx).decode()
                if msg:
                    raise SolverError(rc, msg)
            raise SolverError(rc, "solve_batch failed")
        return list(sa)

    def get_blocks(self):
        out = np.empty(self._nvalues, np.float64)
        self._chk(self._f("get_blocks")(self._ctx, _ptr(out, _dp), out.size))
        return out

    def reset_values(self):
        self._chk(self._f("reset_values")(self._ctx))

    def iterations(self):
        n = self._f("num_iterations_recorded")(self._ctx)
        out = []
        for i in range(n):
            it = Iteration()
            self._chk(self._f("get_iteration")(self._ctx, i, C.byref(it)))
            out.append(it)
        return out

    # -- evaluate --------------------------------------------------------------------------
    def num_residuals(self):
        return self._f("num_residuals")(self._ctx)

    def num_parameters_tangent(self):
        return self._f("num_parameters_tangent")(self._ctx)

    def tangent_offset(self, block):
        return self._f("tangent_offset")(self._ctx, block)

    def evaluate(self, residuals=True, gradient=True, jacobian=False):
        self.finalize()
        m, n = self.num_residuals(), self.num_parameters_tangent()
        cost = C.c_double(0.0)
        r = np.zeros(m) if residuals else None
        g = np.zeros(n) if gradient else None
        J = np.zeros((m, n)) if jacobian else None
        self._chk(self._f("evaluate")(self._ctx, C.byref(cost), _ptr(r, _dp), _ptr(g, _dp), _ptr(J, _dp)))
        return cost.value, r, g, J

    def covariance(self, block_a, block_b, ta=3, tb=3):
        out = np.zeros((ta, tb))
        self._chk(self._f("covariance")(self._ctx, block_a, block_b, _ptr(out, _dp)))
        return out
