"""The block-sparse PCG solve as ONE resident launch (k_pcg.hip pcg_persistent_kernel: row ranges per workgroup, dot products as slot
stores, z gathered in flight) against the launch-per-iteration path it replaces and against the exact factorisation: same steps, same
optimum; lists of named columns that are not pose pairs; two solver threads at once (the second one takes the launches)."""
import threading

import numpy as np
import pytest

from beam_slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _solve(gpu_solver_cls, pr, linear=capi.LINEAR_PCG, iters=8):
    g = gpu_solver_cls(0)
    pr.load(g)
    opt = g.options_default()
    opt.max_num_iterations = iters
    opt.linear_solver_type = linear
    opt.pcg_tolerance = 1e-11
    opt.pcg_max_iterations = 3000
    s = g.solve(opt)
    return s, g.get_blocks(), [i.step_is_successful for i in g.iterations()]


def test_one_launch_equals_launch_per_iteration(gpu_solver_cls, monkeypatch):
    """(with the block-Jacobi preconditioner alone, BSGPU_PCG_COARSE=0: the launch-per-iteration path has no coarse space)"""
    pr = synthetic.pose_graph(n_pose=600, n_loop=2500, seed=5)
    monkeypatch.setenv("BSGPU_PCG_COARSE", "0")
    s1, x1, acc1 = _solve(gpu_solver_cls, pr)
    monkeypatch.setenv("BSGPU_PCG_LAUNCHES", "1")
    s2, x2, acc2 = _solve(gpu_solver_cls, pr)
    monkeypatch.delenv("BSGPU_PCG_LAUNCHES")
    monkeypatch.delenv("BSGPU_PCG_COARSE")
    assert s1.linear_solver_used == s2.linear_solver_used == capi.LINEAR_PCG
    assert acc1 == acc2
    assert abs(s1.final_cost - s2.final_cost) <= 1e-9 * s2.final_cost
    assert abs(s1.num_inner_iterations - s2.num_inner_iterations) <= 0.05 * s2.num_inner_iterations + 5
    assert np.abs(x1 - x2).max() < 1e-7
    s3, x3, _ = _solve(gpu_solver_cls, pr, capi.LINEAR_SCHUR_CHOLESKY)      # the exact step
    assert abs(s1.final_cost - s3.final_cost) <= 1e-7 * s3.final_cost


def test_two_level_preconditioner(gpu_solver_cls, monkeypatch):
    """M^-1 = block-Jacobi + W E^-1 W^T with W the six rigid motions of the free poses: the same LM trajectory as block-Jacobi alone and
    as the exact step, in at most 60 % of the inner iterations (the six gauge-like modes the anchor alone holds are what costs them)"""
    pr = synthetic.pose_graph(n_pose=600, n_loop=2500, seed=5)
    s1, x1, acc1 = _solve(gpu_solver_cls, pr)
    monkeypatch.setenv("BSGPU_PCG_COARSE", "0")
    s2, x2, acc2 = _solve(gpu_solver_cls, pr)
    monkeypatch.delenv("BSGPU_PCG_COARSE")
    s3, x3, acc3 = _solve(gpu_solver_cls, pr, capi.LINEAR_SCHUR_CHOLESKY)
    assert s1.linear_solver_used == capi.LINEAR_PCG
    assert acc1 == acc2 == acc3
    assert abs(s1.final_cost - s2.final_cost) <= 1e-9 * s2.final_cost
    assert abs(s1.final_cost - s3.final_cost) <= 1e-7 * s3.final_cost
    assert np.abs(x1 - x2).max() < 1e-7
    assert 0 < s1.num_inner_iterations <= 0.6 * s2.num_inner_iterations
    # no anchor at all (the prior left out): the gauge is held by the LM diagonal only, every pose is free
    pr2 = synthetic.pose_graph(n_pose=400, n_loop=1500, seed=6)
    del pr2.factors[capi.F_ABSPOSE]
    s4, x4, acc4 = _solve(gpu_solver_cls, pr2)
    s5, x5, acc5 = _solve(gpu_solver_cls, pr2, capi.LINEAR_SCHUR_CHOLESKY)
    assert acc4 == acc5 and abs(s4.final_cost - s5.final_cost) <= 1e-7 * s5.final_cost


def test_named_columns_that_are_not_pose_pairs(gpu_solver_cls):
    """every third orientation held constant: those poses contribute one block column, the lists are no longer made of pairs"""
    pr = synthetic.pose_graph(n_pose=500, n_loop=2000, seed=9)
    blocks = pr.meta["blocks"]
    for k in range(1, blocks.shape[0], 3):
        pr.is_const[int(blocks[k, 1])] = 1
    s1, x1, acc1 = _solve(gpu_solver_cls, pr)
    s3, x3, acc3 = _solve(gpu_solver_cls, pr, capi.LINEAR_SCHUR_CHOLESKY)
    assert s1.linear_solver_used == capi.LINEAR_PCG and s1.num_inner_iterations > 0
    assert acc1 == acc3
    assert abs(s1.final_cost - s3.final_cost) <= 1e-7 * s3.final_cost
    assert np.abs(x1 - x3).max() < 1e-5


def test_two_solver_threads(gpu_solver_cls):
    """the resident launch needs the device to itself: a second thread solving at the same time takes the launch-per-iteration path"""
    prs = [synthetic.pose_graph(n_pose=700, n_loop=3000, seed=30 + i) for i in range(2)]
    ref = [_solve(gpu_solver_cls, pr)[0].final_cost for pr in prs]
    out = [[None] * 3 for _ in prs]

    def work(i):
        for rep in range(3):
            out[i][rep] = _solve(gpu_solver_cls, prs[i])[0].final_cost

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th)
    for i in range(2):
        for c in out[i]:
            assert c is not None and abs(c - ref[i]) <= 1e-9 * ref[i]


def test_many_rows_per_workgroup(gpu_solver_cls, monkeypatch):
    """a long, sparsely closed trajectory: ~150 block rows per workgroup — more than the 64 row groups hold in registers, and more
    tail blocks than the LDS keeps (the rest comes from memory)"""
    pr = synthetic.pose_graph(n_pose=20000, n_loop=4000, seed=11)
    s1, x1, acc1 = _solve(gpu_solver_cls, pr, iters=4)
    monkeypatch.setenv("BSGPU_PCG_LAUNCHES", "1")
    s2, x2, acc2 = _solve(gpu_solver_cls, pr, iters=4)
    assert acc1 == acc2
    assert abs(s1.final_cost - s2.final_cost) <= 1e-8 * s2.final_cost


@pytest.mark.parametrize("nth", [1, 3])
def test_a_resident_launch_that_is_given_up_has_its_step_computed_again(gpu_solver_cls, monkeypatch, nth, capfd):
    """The resident launch's verdict is read with the step's other scalars (bsgpu_solve.cpp pcg_check); when it did not finish — a shared
    device — the step is asked for again (lm_state.h retry) and computed launch per iteration.  BSGPU_PCG_GIVE_UP declares the n-th
    verdict a failure: the first step's, and one in the middle of the solve.  (BSGPU_PCG_COARSE=0: the launch-per-iteration path has no
    coarse space, so the two paths are the same iteration.)"""
    monkeypatch.setenv("BSGPU_PCG_COARSE", "0")
    pr = synthetic.pose_graph(n_pose=2200, n_loop=3000, seed=3)
    def run():
        g = gpu_solver_cls(0)
        pr.load(g)
        o = g.options_default(); o.max_num_iterations = 6; o.pcg_tolerance = 1e-10
        s = g.solve(o)
        return s, [i.cost for i in g.iterations()], [i.step_is_successful for i in g.iterations()], g.get_blocks()
    s0, c0, a0, x0 = run()
    capfd.readouterr()
    monkeypatch.setenv("BSGPU_PCG_GIVE_UP", str(nth))
    s1, c1, a1, x1 = run()
    # the path was really taken (the hook counts the verdicts of the context, not of the process): the library says so once
    assert capfd.readouterr().err.count("the resident PCG launch was given up") == 1
    assert a0 == a1 and s0.num_iterations == s1.num_iterations and s0.termination_type == s1.termination_type
    assert np.allclose(c0, c1, rtol=1e-7) and np.abs(x0 - x1).max() < 1e-4
