"""The alternative code paths an environment variable selects in libbsgpu (the launch-per-step factorisation the fused kernel falls back
to, the legacy / non-deep back-substitutions, other tile orders and chain lengths, packed flag words, own-strip solves, a full table
hand-over ...): every one of them solves the same window to the same optimum as the default path.  One subprocess per setting — some of
the switches are read once per process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
kind = sys.argv[1] if len(sys.argv) > 1 else "vio"
pr = (synthetic.idp_window(n_kf=40, n_lm=3000, seed=78) if kind == "idp" else
      synthetic.lio_window(n_kf=60, n_rel=6000, seed=79) if kind == "lio" else synthetic.vio_window(n_kf=90, n_lm=6000, seed=77))
g = GpuSolver(0)
pr.load(g)
if kind in ("prior", "lio_prior"):   # the window after a slide with true marginalisation: its first key frame (and the landmarks only it sees) as a dense prior
    from beam_slam_amd import capi
    if kind == "prior":
        pr = synthetic.vio_window(n_kf=24, n_lm=700, seed=81)
        kf = pr.meta["kf_blocks"]
        idx = np.concatenate([c[0] for c in pr.factors[capi.F_REPROJ]])
        seen0 = set(int(v) for v in idx[idx[:, 0] == int(kf[0, 0]), 2])
        first_only = [l for l in seen0 if set(idx[idx[:, 2] == l][:, 0]) == {int(kf[0, 0])}]
    else:   # a lidar-inertial window: no landmarks, the prior couples the key frames the first one's scan registrations reached
        pr = synthetic.lio_window(n_kf=30, n_rel=1500, seed=82)
        kf = pr.meta["kf_blocks"]
        first_only = []
    marg = [int(b) for b in kf[0]] + first_only
    pr.load(g); g.solve()
    kept, A, b, xbar = g.marginalize(marg, pr.size)
    pr = pr.marginalized(marg, kept, A, b, xbar)
    g = GpuSolver(0); pr.load(g)
o = g.options_vio(); o.max_solver_time_in_seconds = 0.0; o.max_num_iterations = 6
s = g.solve(o)
x = g.get_blocks()
print(json.dumps({"cost": s.final_cost, "it": s.num_iterations, "acc": [int(i.step_is_successful) for i in g.iterations()],
                  "x": [float(v) for v in x[:2000:7]], "xn": float(np.abs(x).sum())}))
""" % ROOT


def _run(env_extra, window="vio"):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT, window], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def default_run():
    return _run({})


SETTINGS = [
    {"BSGPU_CHOL_FUSED": "0"},                                   # launch-per-step factorisation (the fused kernel's fall-back)
    {"BSGPU_BACKSOLVE_LEGACY": "1"},
    {"BSGPU_BACKSOLVE_FUSED": "0"},
    {"BSGPU_BACKSOLVE_NO_W": "1"},
    {"BSGPU_BACKSOLVE_FUSED": "0", "BSGPU_BACKSOLVE_NO_W": "1"},
    {"BSGPU_BACKSOLVE_GLOBAL_Y": "1"},                           # solution vector in global memory (windows above 12 288 dimensions)
    {"BSGPU_CHOL_FUSED": "0", "BSGPU_BACKSOLVE_LEGACY": "1"},
    {"BSGPU_CHAINS": "1"},
    {"BSGPU_REDUCE_LAUNCH": "1"},                                # the end-of-step reduction as a launch of its own instead of the first workgroups of the evaluation launched ahead of the decision
    {"BSGPU_REDUCE_LAUNCH": "1", "BSGPU_POSE_DIAG_LAUNCH": "1"},
    {"BSGPU_POSE_DIAG_LAUNCH": "1"},                             # the LM diagonal and the gradient norms in a launch of their own instead of as tasks of the factorisation's launch
    {"BSGPU_POSE_DIAG_LAUNCH": "1", "BSGPU_CHOL_EXT": "0"},
    {"BSGPU_CHOL_EXT": "0"},                                     # a separator's appendix tile (<= 16 real columns) as a panel of its own instead of riding in the tasks of the panel before it
    {"BSGPU_CHOL_EXT": "0", "BSGPU_CHOL_FUSED": "0"},
    {"BSGPU_DIM_ORDER": "0"},                                    # the tile-level nested dissection (runs of natural tiles) instead of the per-dimension order
    {"BSGPU_DIM_ORDER": "0", "BSGPU_CHAINS": "4"},
    {"BSGPU_DIM_ORDER_DEPTH": "2"},                              # a shallower dissection: fewer, larger pieces
    {"BSGPU_DIM_ORDER_DEPTH": "0"},                              # one supernode: natural order, the whole system one sequence of chains
    {"BSGPU_DIM_ABSORB": "0"},                                   # no separator joins its parent separator (dim_order.h absorb_separators)
    {"BSGPU_SHARED": "0"},
    {"BSGPU_GRAPH": "1"},                                        # the LM step replayed as hipGraphs
    {"BSGPU_FLATTEN": "device"},
    {"BSGPU_FLATTEN": "host"},
    {"BSGPU_FLATTEN": "host", "BSGPU_PAIR_ENTRIES_SORT": "1"},
    {"BSGPU_SCALARS_EVENT": "1"},                                # the host waits for an event behind the end-of-step reduction, not for the mirror's stamp
    {"BSGPU_UPDATE_SEPARATE": "1"},                              # the candidate update as a launch of its own
    {"BSGPU_EVAL_MERGE": "0"},                                   # the IMU factors evaluated by a launch of their own
    {"BSGPU_EVAL_MERGE": "1"},                                   # ... in the passes with Jacobians only    # pair entries by a comparison sort (windows of > 2 896 camera poses)
]


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_alternative_path_reaches_the_same_optimum(default_run, setting):
    r = _run(setting)
    d = default_run
    assert r["it"] == d["it"] and r["acc"] == d["acc"]
    assert abs(r["cost"] - d["cost"]) <= 1e-9 * d["cost"]
    assert max(abs(a - b) for a, b in zip(r["x"], d["x"])) < 1e-7


# the inverse-depth window (landmark-side elimination, k_idp.hip) under the switches that change what runs around it
IDP_SETTINGS = [
    {"BSGPU_GRAPH": "1"},
    {"BSGPU_CHOL_FUSED": "0", "BSGPU_BACKSOLVE_LEGACY": "1"},
    {"BSGPU_BACKSOLVE_FUSED": "0"},
    {"BSGPU_IDP_GENERIC_ASSEMBLY": "1"},
    {"BSGPU_IDP_ELIM": "0"},
    {"BSGPU_IDP_ELIM": "0", "BSGPU_NO_LEAF_TILES": "1"},
    {"BSGPU_PAIR_ENTRIES_SORT": "1"},
    {"BSGPU_CLEAR_AT_START": "1"},
]


@pytest.fixture(scope="module")
def default_idp_run():
    return _run({}, "idp")


@pytest.mark.parametrize("setting", IDP_SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_inverse_depth_window_under_alternative_paths(default_idp_run, setting):
    r = _run(setting, "idp")
    d = default_idp_run
    assert r["it"] == d["it"] and r["acc"] == d["acc"]
    assert abs(r["cost"] - d["cost"]) <= 1e-9 * d["cost"]
    assert max(abs(a - b) for a, b in zip(r["x"], d["x"])) < 1e-7


# a lidar-inertial window (relative-pose constraints with extrinsics + IMU factors, no landmarks)
LIO_SETTINGS = [
    {"BSGPU_NO_GROUP_ASSEMBLY": "1"},                            # every pose-only factor by segments (no same-slot groups)
    {"BSGPU_CLEAR_AT_START": "1"},                               # the step's clearing as a launch at its start, not in the previous step's last launch
    {"BSGPU_UPDATE_SEPARATE": "1"},
    {"BSGPU_EVAL_MERGE": "0"},
    {"BSGPU_CHOL_FUSED": "0", "BSGPU_BACKSOLVE_LEGACY": "1"},
    {"BSGPU_GRAPH": "1"},
]


@pytest.fixture(scope="module")
def default_lio_run():
    return _run({}, "lio")


@pytest.mark.parametrize("setting", LIO_SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_lidar_inertial_window_under_alternative_paths(default_lio_run, setting):
    r = _run(setting, "lio")
    d = default_lio_run
    assert r["it"] == d["it"] and r["acc"] == d["acc"]
    assert abs(r["cost"] - d["cost"]) <= 1e-9 * d["cost"]
    assert max(abs(a - b) for a, b in zip(r["x"], d["x"])) < 1e-7


# a window that carries a dense marginal prior (fixed_lag_smoother.cpp:269-272)
PRIOR_SETTINGS = [
    {"BSGPU_MARG_RIDE": "0"},                                    # the prior's evaluation, assembly and model-cost terms as launches of their own, not inside the other pose-only factors' launches
    {"BSGPU_EVAL_SEPARATE": "1"},
    {"BSGPU_GRAPH": "1"},
    {"BSGPU_CHOL_FUSED": "0", "BSGPU_BACKSOLVE_LEGACY": "1"},
]


@pytest.fixture(scope="module")
def default_prior_run():
    return _run({}, "prior")


@pytest.mark.parametrize("setting", PRIOR_SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_window_with_dense_prior_under_alternative_paths(default_prior_run, setting):
    r = _run(setting, "prior")
    d = default_prior_run
    assert r["it"] == d["it"] and r["acc"] == d["acc"]
    assert abs(r["cost"] - d["cost"]) <= 1e-9 * d["cost"]
    assert max(abs(a - b) for a, b in zip(r["x"], d["x"])) < 1e-7


@pytest.fixture(scope="module")
def default_lio_prior_run():
    return _run({}, "lio_prior")


@pytest.mark.parametrize("setting", [{"BSGPU_MARG_RIDE": "0"}, {"BSGPU_EVAL_SEPARATE": "1"}], ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()))
def test_lidar_inertial_window_with_dense_prior_under_alternative_paths(default_lio_prior_run, setting):
    """no landmark launches to ride in: the prior's evaluation and assembly ride with the relative-pose factors', its model-cost terms go by themselves"""
    r = _run(setting, "lio_prior")
    d = default_lio_prior_run
    assert r["it"] == d["it"] and r["acc"] == d["acc"]
    assert abs(r["cost"] - d["cost"]) <= 1e-9 * d["cost"]
    assert max(abs(a - b) for a, b in zip(r["x"], d["x"])) < 1e-7
