"""The HIP path against the reference's ACTUAL reprojection Jacobian.

EuclideanReprojection::Evaluate (bs_constraints/include/bs_constraints/visual/euclidean_reprojection_function.h:124-143)
does not return the closed-form d r / d q: it forward-differences the transformed point on the four quaternion
coefficients (EPSILON 1e-8, with re-normalisation) and Ceres multiplies by the manifold's PlusJacobian.  The HIP kernels
evaluate the closed form in the tangent space.  The oracle restates both (oracle/bs_oracle.cpp reproj_analytic, mode 0 and
mode 1); these tests run the oracle's LM with the reference's variant (mode 1) and hold the HIP solve to it: per-iteration
decisions and costs, final cost and values within the north-star 1e-6.  The gap that remains is the
forward difference's own truncation/rounding error, ~1e-8 relative on the Jacobian."""
import numpy as np
import pytest

from beam_slam_amd import synthetic

pytestmark = pytest.mark.gpu


def _solve_pair(pr, oracle_cls, gpu_solver_cls, iters):
    g, o = gpu_solver_cls(0), oracle_cls()
    pr.load(g); pr.load(o)
    o.set_reproj_mode(1)
    # the oracle must really be in the reference's variant: its Jacobian differs from the closed form by the FD error
    opt = g.options_vio()
    opt.max_solver_time_in_seconds = 0.0
    opt.max_num_iterations = iters
    sg, so = g.solve(opt), o.solve(opt)
    return g, o, sg, so


def _check(g, o, sg, so):
    ig, io = g.iterations(), o.iterations()
    assert len(ig) == len(io)
    worst = 0.0
    for a, b in zip(ig, io):
        assert a.step_is_successful == b.step_is_successful
        worst = max(worst, abs(a.cost - b.cost) / b.cost)
    assert worst <= 1e-6, worst
    rel = abs(sg.final_cost - so.final_cost) / so.final_cost
    assert rel <= 1e-6, rel
    dv = np.abs(g.get_blocks() - o.get_blocks()).max()
    assert dv <= 1e-6, dv
    return worst, rel, dv


def test_c1_solve_vs_reference_fd_variant(oracle_cls, gpu_solver_cls):
    pr = synthetic.c1()
    g, o, sg, so = _solve_pair(pr, oracle_cls, gpu_solver_cls, 15)
    worst, rel, dv = _check(g, o, sg, so)
    print(f"C1 vs reference FD variant: per-iteration cost {worst:.2e}, final cost {rel:.2e}, values {dv:.2e}")
    # and the variant is not the closed form in disguise: the same oracle in mode 0 lands measurably elsewhere or bit-identical
    # trajectories would make this test vacuous
    o0 = oracle_cls(); pr.load(o0); o0.set_reproj_mode(0)
    o1 = oracle_cls(); pr.load(o1); o1.set_reproj_mode(1)
    J0 = o0.evaluate(jacobian=True)[3]; J1 = o1.evaluate(jacobian=True)[3]
    assert 1e-12 < np.abs(J0 - J1).max() / np.abs(J0).max() < 1e-5


def test_c2_first_iterations_vs_reference_fd_variant(oracle_cls, gpu_solver_cls):
    pr = synthetic.c2()
    g, o, sg, so = _solve_pair(pr, oracle_cls, gpu_solver_cls, 3)
    worst, rel, dv = _check(g, o, sg, so)
    print(f"C2 (3 iterations) vs reference FD variant: per-iteration cost {worst:.2e}, final cost {rel:.2e}, values {dv:.2e}")
