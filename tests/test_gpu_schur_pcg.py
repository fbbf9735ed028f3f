"""BSGPU_LINEAR_SCHUR_PCG: landmark Schur complement + block-Jacobi PCG on the reduced camera system (Ceres ITERATIVE_SCHUR with the
SCHUR_JACOBI preconditioner named by beam_slam_launch/config/optimization/ceres_config.json:11-12), against the exact Schur +
Cholesky step of the same library and against the oracle: with a tight inner tolerance the two linear solvers give the same LM
trajectory, so the final costs agree far inside the north-star 1e-6."""
import numpy as np
import pytest

from beam_slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _solve(pr, cls, linear, iters, tol=1e-12, max_inner=2000, vio=False):
    g = cls(0)
    pr.load(g)
    opt = g.options_vio() if vio else g.options_default()
    opt.max_solver_time_in_seconds = 0.0
    opt.max_num_iterations = iters
    opt.linear_solver_type = linear
    opt.pcg_tolerance = tol
    opt.pcg_max_iterations = max_inner
    s = g.solve(opt)
    return g, s


def test_c1_window_schur_pcg_matches_cholesky_and_oracle(oracle_cls, gpu_solver_cls):
    pr = synthetic.c1()
    gd, sd = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_CHOLESKY, 25)
    gp, sp = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_PCG, 25)
    assert sd.linear_solver_used == capi.LINEAR_SCHUR_CHOLESKY and sp.linear_solver_used == capi.LINEAR_SCHUR_PCG
    assert sp.num_inner_iterations > 0
    assert sp.num_iterations == sd.num_iterations
    for a, b in zip(gp.iterations(), gd.iterations()):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-8 * b.cost
    assert abs(sp.final_cost - sd.final_cost) <= 1e-9 * sd.final_cost
    assert np.abs(gp.get_blocks() - gd.get_blocks()).max() < 1e-6
    o = oracle_cls()
    pr.load(o)
    opt = gd.options_default()
    opt.max_num_iterations = 25
    so = o.solve(opt)
    assert abs(sp.final_cost - so.final_cost) <= 1e-6 * so.final_cost


def test_c2_window_schur_pcg_matches_cholesky(gpu_solver_cls):
    """The headline window with the iterative step: same decisions, final cost within 1e-6 of the exact path (north-star)."""
    pr = synthetic.c2()
    gd, sd = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_CHOLESKY, 10, vio=True)
    gp, sp = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_PCG, 10, tol=1e-10, vio=True)
    assert sp.linear_solver_used == capi.LINEAR_SCHUR_PCG and sp.num_inner_iterations > 0
    assert sp.num_iterations == sd.num_iterations
    assert abs(sp.final_cost - sd.final_cost) <= 1e-6 * sd.final_cost
    assert np.abs(gp.get_blocks() - gd.get_blocks()).max() < 1e-5


def test_inexact_steps_still_converge(gpu_solver_cls):
    """Ceres' default forcing tolerance for ITERATIVE_SCHUR is 0.1: truncated inner solves give a different trajectory but the same
    optimum on a well-posed window."""
    pr = synthetic.c1()
    gd, sd = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_CHOLESKY, 50)
    gp, sp = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_PCG, 50, tol=0.1, max_inner=500)
    assert sp.is_solution_usable
    assert abs(sp.final_cost - sd.final_cost) <= 1e-4 * sd.final_cost


def test_pose_only_window(gpu_solver_cls):
    """No landmarks (C3-shaped LIO window): the reduced system is the whole system."""
    pr = synthetic.lio_window(n_kf=30, n_rel=600, seed=7)
    gd, sd = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_CHOLESKY, 10, vio=True)
    gp, sp = _solve(pr, gpu_solver_cls, capi.LINEAR_SCHUR_PCG, 10, vio=True)
    assert sp.num_iterations == sd.num_iterations
    assert abs(sp.final_cost - sd.final_cost) <= 1e-8 * sd.final_cost
