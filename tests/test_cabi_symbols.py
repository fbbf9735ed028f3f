"""CPU-side checks of the product library: it loads, exports every symbol include/bsgpu.h declares,
and fails loudly without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from beam_slam_amd import capi, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bsgpu.h")).read()
    return sorted(set(re.findall(r"\b(bsgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = gpu.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bsgpu.h but not exported"
    assert set("bsgpu_" + s for s in capi.SYMBOLS) == set(declared)
    lib.bsgpu_abi_version.restype = ctypes.c_int
    assert lib.bsgpu_abi_version() == 1


def test_type_tables_match_python_ir():
    from beam_slam_amd import problem
    lib = gpu.lib()
    for t in range(capi.F_NUM_TYPES):
        assert lib.bsgpu_nidx(t) == problem.NIDX[t]
        assert lib.bsgpu_nconst(t) == problem.NCONST[t]
        assert lib.bsgpu_nres(t) == problem.NRES[t]
    assert lib.bsgpu_nidx(capi.F_NUM_TYPES) == -1


def test_struct_layouts_match_header():
    # sizes derived by hand from include/bsgpu.h (all members naturally aligned)
    assert ctypes.sizeof(capi.Options) == 4 * 4 + 10 * 8 + 2 * 4 + 8
    assert ctypes.sizeof(capi.Summary) == 10 * 4 + 8 * 8 + 160
    assert ctypes.sizeof(capi.Iteration) == 4 * 4 + 8 * 8
    assert ctypes.sizeof(capi.Camera) == 16 * 8


def test_no_cpu_fallback():
    """Without a HIP device bsgpu_create must fail with a message — never compute on the CPU."""
    has_gpu = os.path.exists("/dev/kfd")      # (torch.cuda.is_available() is unreliable once another HIP runtime is loaded)
    if not has_gpu:
        try:
            import torch
            has_gpu = torch.cuda.is_available()
        except Exception:
            pass
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SolverError) as e:
        gpu.GpuSolver(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_options_presets():
    lib = gpu.lib()
    o = capi.Options()
    lib.bsgpu_options_vio(ctypes.byref(o))
    # beam_slam_launch/config/vio.yaml:13-17
    assert o.max_num_iterations == 10 and o.max_solver_time_in_seconds == 0.05
    assert o.gradient_tolerance == o.parameter_tolerance == o.function_tolerance == 1.5e-7
    lib.bsgpu_options_default(ctypes.byref(o))
    assert o.max_num_iterations == 50 and o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3


def test_null_context_is_an_error_not_a_crash():
    """Every context-taking entry point refuses a NULL context with BSGPU_ERR_INVALID (no compute, so no GPU needed)."""
    lib = gpu.lib()
    null = ctypes.c_void_p(None)
    z = ctypes.c_void_p(None)
    for name, args in [("clear", ()), ("finalize", ()), ("reset_values", ()), ("set_values", (z, ctypes.c_int64(0))),
                       ("set_cameras", (ctypes.c_int32(0), z)), ("get_blocks", (z, ctypes.c_int64(0))),
                       ("covariance", (ctypes.c_int32(0), ctypes.c_int32(0), z)), ("reprojection_errors", (z,)),
                       ("add_factors", (ctypes.c_int32(0), ctypes.c_int32(0), z, z, z, z))]:
        fn = getattr(lib, "bsgpu_" + name)
        saved = fn.argtypes
        fn.argtypes, fn.restype = None, ctypes.c_int     # (capi installs typed prototypes; raw NULLs here)
        try:
            assert fn(null, *args) == capi.ERR_INVALID, name
        finally:
            fn.argtypes = saved
    lib.bsgpu_last_error.restype = ctypes.c_char_p
    assert lib.bsgpu_last_error(null) == b"null context"
    assert lib.bsgpu_num_residuals(null) == -1 and lib.bsgpu_tangent_offset(null, 0) == -1
    lib.bsgpu_destroy(null)     # no-op
