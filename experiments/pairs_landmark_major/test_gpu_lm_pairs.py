"""The landmark-major assembly of the camera blocks (csrc/lm_pairs_plan.h, pairs_lm_kernel) against the camera-pair segments
(pairs_kernel) it replaces on large windows: the same reduced system, so the same LM iterations — compared on windows small enough to
run in seconds, with the form forced either way (BSGPU_PAIRS_LM is read when a problem is finalized)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(pr, form, iters=6, batch=False):
    from beam_slam_amd.gpu import GpuSolver
    old = os.environ.get("BSGPU_PAIRS_LM")
    os.environ["BSGPU_PAIRS_LM"] = form
    try:
        g = GpuSolver(0)
        pr.load(g)
        info = g.assembly_info()
        o = g.options_vio(); o.max_solver_time_in_seconds = 0.0; o.max_num_iterations = iters
        s = g.solve(o)
        its = g.iterations()
        return info, s, its, g.get_blocks()
    finally:
        if old is None:
            del os.environ["BSGPU_PAIRS_LM"]
        else:
            os.environ["BSGPU_PAIRS_LM"] = old


def _same(a, b):
    (_, sa, ia, xa), (_, sb, ib, xb) = a, b
    assert sa.num_iterations == sb.num_iterations
    assert [i.step_is_successful for i in ia] == [i.step_is_successful for i in ib]
    assert abs(sa.final_cost - sb.final_cost) <= 1e-9 * abs(sb.final_cost)
    for u, v in zip(ia, ib):
        assert abs(u.cost - v.cost) <= 1e-9 * abs(v.cost)
        assert abs(u.gradient_max_norm - v.gradient_max_norm) <= 1e-6 * max(1.0, abs(v.gradient_max_norm))
    assert np.allclose(xa, xb, rtol=0, atol=1e-8)


@pytest.mark.parametrize("kw", [dict(n_kf=90, n_lm=6000, seed=77),                                 # tracks of 4..12 key frames
                                dict(n_kf=40, n_lm=2500, seed=5, track_min=2, track_max=16),        # ... up to the accumulator's 16 camera poses
                                dict(n_kf=24, n_lm=900, seed=9, track_min=1, track_max=3)])         # landmarks seen once: the diagonal entries alone
def test_landmark_major_equals_pair_segments(kw):
    from beam_slam_amd import synthetic
    pr = synthetic.vio_window(**kw)
    seg = _solve(pr, "0")
    lm = _solve(pr, "1")
    assert seg[0][1] == 0 and seg[0][0] > 0
    assert lm[0][1] > 0 and lm[0][2] >= lm[0][1]
    _same(lm, seg)


def test_constant_landmarks_and_constant_poses():
    """Factors of constant landmarks (a unit of their own each) and of a constant camera pose (its rows and columns are skipped)."""
    from beam_slam_amd import synthetic
    pr = synthetic.vio_window(n_kf=30, n_lm=1500, seed=21)
    for b in pr.meta["lm_blocks"][::7]:
        pr.is_const[int(b)] = 1
    for b in pr.meta["kf_blocks"][5][:2]:
        pr.is_const[int(b)] = 1
    seg = _solve(pr, "0")
    lm = _solve(pr, "1")
    assert lm[0][1] > 0
    _same(lm, seg)


def test_random_windows_both_forms():
    """The seeded random windows of test_gpu_random.py (ragged tracks, constant blocks, every factor type) with the form forced."""
    from test_gpu_random import _random_case as make_problem
    for seed in range(0, 40, 4):   # (kind 0: visual-inertial windows)
        pr = make_problem(seed)
        seg = _solve(pr, "0", iters=4)
        lm = _solve(pr, "1", iters=4)
        if lm[0][0] > 0:
            assert lm[0][1] > 0
        _same(lm, seg)


def test_window_with_a_long_track_keeps_the_pair_segments():
    from beam_slam_amd import synthetic
    pr = synthetic.vio_window(n_kf=30, n_lm=800, seed=3, track_min=17, track_max=20)   # more camera poses than the accumulator holds
    info, s, _, _ = _solve(pr, "1", iters=3)
    assert info[1] == 0 and info[0] > 0
    assert np.isfinite(s.final_cost)
