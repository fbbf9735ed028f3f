"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must match them through the C-ABI (GPU)."""
import glob
import os

import numpy as np
import pytest

from beam_slam_amd.problem import Problem

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _check(solver, d, tol_r, tol_cost, tol_x):
    pr = Problem.from_arrays(d)
    pr.load(solver)
    cost, r, g, _ = solver.evaluate()
    assert abs(cost - float(d["exp_cost"])) <= 1e-12 * abs(float(d["exp_cost"]))
    assert np.abs(r - d["exp_residuals"]).max() <= tol_r * max(1.0, np.abs(d["exp_residuals"]).max())
    assert np.abs(g - d["exp_gradient"]).max() <= tol_r * max(1.0, np.abs(d["exp_gradient"]).max())
    opt = solver.options_default()
    opt.max_num_iterations = int(d["exp_max_iterations"])
    s = solver.solve(opt)
    its = solver.iterations()
    assert s.termination_type == int(d["exp_termination"])
    assert [i.step_is_successful for i in its] == list(d["exp_iter_ok"])
    ic = np.array([i.cost for i in its])
    ok = d["exp_iter_ok"].astype(bool)
    assert np.abs(ic[ok] - d["exp_iter_cost"][ok]).max() <= 1e-6 * np.abs(d["exp_iter_cost"][ok]).max()
    assert abs(s.final_cost - float(d["exp_final_cost"])) <= tol_cost * float(d["exp_final_cost"])
    assert np.abs(solver.get_blocks() - d["exp_final_values"]).max() <= tol_x


def test_fixtures_exist():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle_cls, path):
    _check(oracle_cls(threads=1), np.load(path), 1e-13, 1e-12, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_matches_golden(gpu_solver_cls, path):
    _check(gpu_solver_cls(0), np.load(path), 1e-9, 1e-6, 1e-4)
