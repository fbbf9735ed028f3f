"""Inverse-depth landmarks at scale (missing #1).  The scalar rho blocks stay in the reduced system,
but the tile planner (csrc/dense_plan.h) orders the tiles made of them FIRST — they are coupled to keyframes only, never to each
other — so that the tiled factorisation eliminates them as a landmark Schur complement instead of filling a landmark x landmark block:
a window with 20 000 inverse-depth landmarks (21 000 reduced dimensions) solves through the exact path."""
import time

import numpy as np
import pytest

from beam_slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _solve(pr, cls, iters=15):
    g = cls(0)
    pr.load(g)
    opt = g.options_default()
    opt.max_num_iterations = iters
    opt.linear_solver_type = capi.LINEAR_SCHUR_CHOLESKY
    s = g.solve(opt)
    return g, s


def test_medium_window_matches_oracle_and_the_natural_order(oracle_cls, gpu_solver_cls, monkeypatch):
    pr = synthetic.idp_window(n_kf=30, n_lm=2000, seed=21)
    g, s = _solve(pr, gpu_solver_cls)
    _, n_steps, n_tiles = g.plan_info()
    assert n_tiles >= 30
    o = oracle_cls()
    pr.load(o)
    opt = g.options_default()
    opt.max_num_iterations = 15
    so = o.solve(opt)
    assert s.num_iterations == so.num_iterations
    for a, b in zip(g.iterations(), o.iterations()):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-8 * abs(b.cost)
    assert abs(s.final_cost - so.final_cost) <= 1e-6 * so.final_cost
    assert np.abs(g.get_blocks() - o.get_blocks()).max() < 1e-6
    # the same window with the landmark tiles left where the block order puts them (behind the keyframes)
    monkeypatch.setenv("BSGPU_NO_LEAF_TILES", "1")
    g2, s2 = _solve(pr, gpu_solver_cls)
    assert abs(s2.final_cost - s.final_cost) <= 1e-9 * s.final_cost
    assert g2.plan_info()[1] != n_steps     # (a different elimination order)


def test_twenty_thousand_inverse_depth_landmarks_solve(gpu_solver_cls):
    pr = synthetic.idp_window(n_kf=60, n_lm=20000, seed=22)
    g = gpu_solver_cls(0)
    pr.load(g)
    opt = g.options_default()
    opt.max_num_iterations = 10
    t0 = time.perf_counter()
    s = g.solve(opt)
    dt = time.perf_counter() - t0
    assert s.linear_solver_used == capi.LINEAR_SCHUR_CHOLESKY
    assert s.is_solution_usable and s.num_successful_steps >= 3
    assert s.final_cost < 0.2 * s.initial_cost
    costs = [it.cost for it in g.iterations() if it.step_is_successful]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    x = g.get_blocks()
    rho = np.array([x[pr.offset[b]] for b in pr.meta["rho_blocks"]])
    rho0 = np.array([pr.values[pr.offset[b]] for b in pr.meta["rho_blocks"]])
    assert np.abs(rho - pr.meta["rho_true"]).mean() < 0.5 * np.abs(rho0 - pr.meta["rho_true"]).mean()
    print("20000 inverse-depth landmarks x 60 keyframes: %d iterations in %.1f ms (incl. finalize)" % (s.num_iterations, 1e3 * dt))


def test_covariance_marginal_and_iterative_step_with_leaf_tiles(oracle_cls, gpu_solver_cls):
    """The other users of the tiled factorisation on a window whose landmark tiles are ordered first: marginal covariance blocks
    (keyframe x keyframe and keyframe x landmark), and the iterative step on the same assembled system."""
    pr = synthetic.idp_window(n_kf=8, n_lm=300, seed=23)
    g, s = _solve(pr, gpu_solver_cls)
    o = oracle_cls()
    pr.load(o)
    o.set_values(g.get_blocks())
    kf, rho = pr.meta["kf_blocks"], pr.meta["rho_blocks"]
    for a, b in [(kf[2, 0], kf[2, 0]), (kf[1, 1], kf[6, 0]), (kf[3, 1], rho[100]), (rho[250], rho[250]), (rho[10], rho[200])]:
        cg, co = g.covariance(int(a), int(b)), o.covariance(int(a), int(b))
        ref = np.sqrt(np.abs(o.covariance(int(a), int(a))).max() * np.abs(o.covariance(int(b), int(b))).max())
        assert np.abs(cg - co).max() <= 1e-7 * max(np.abs(co).max(), ref)
    g2 = gpu_solver_cls(0)
    pr.load(g2)
    opt = g2.options_default()
    opt.max_num_iterations = 15
    opt.linear_solver_type = capi.LINEAR_SCHUR_PCG
    opt.pcg_tolerance = 1e-12
    opt.pcg_max_iterations = 3000
    s2 = g2.solve(opt)
    assert s2.linear_solver_used == capi.LINEAR_SCHUR_PCG
    assert abs(s2.final_cost - s.final_cost) <= 1e-8 * s.final_cost
