"""The trust-region decision taken on the device for the assembly issued ahead of the host's (bsgpu_solve.cpp enqueue_step, LmDecide; bsgpu_device.h
lm_decide — LmState::advance's arithmetic, [EXT] ceres TrustRegionMinimizer / LevenbergMarquardtStrategy::StepAccepted).  The host still decides
and adopts the assembly only when it names the same radius BIT FOR BIT, so a window solved with BSGPU_LM_DEVICE=1 takes the iterations of
BSGPU_LM_DEVICE=0; what this file adds is that the device's answer is never refused where both say "accepted" (then the mode would only cost
time), on accepted and rejected steps, a tight first radius (factors other than 3), a budget that ends in a gradient-only step, a robust loss."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
out = []
for case in range(7):
    if case == 5:
        pr = synthetic.vio_window(n_kf=40, n_lm=900, seed=5205, track_min=2, track_max=20)   # band landmarks AND pair entries (tracks longer than the band): both launches look at the word
    elif case == 6:
        pr = synthetic.vio_window(n_kf=16, n_lm=500, seed=5206, with_imu=False)              # no IMU factors: no evaluation launch that carries a reduction — all eight units in the landmark launch
    else:
        pr = synthetic.vio_window(n_kf=12 + 4 * case, n_lm=150 + 100 * case, seed=5200 + case, cauchy_a=[None, 5.0][case %% 2])
    g = GpuSolver(0); pr.load(g)
    o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
    if case == 1:
        o.initial_trust_region_radius = 1e-2      # tight first radius: the radius moves by factors other than three
    if case == 2:
        rng = np.random.default_rng(7)
        g.set_values(pr.values + 0.05 * rng.standard_normal(pr.values.size))   # a bad start: rejected steps
    if case == 3:
        o.max_num_iterations = 3                  # the budget ends in a gradient-only step
    if case == 4:
        o.function_tolerance = 1e-3               # ends by the function tolerance: the device must say "not accepted" there too
    s = g.solve(o)
    sys.stderr.write("CASE %%d done\n" %% case)
    out.append(dict(final=s.final_cost, n=s.num_iterations, msg=s.message.decode() if isinstance(s.message, bytes) else str(s.message),
                    its=[(it.cost, it.step_is_successful, it.trust_region_radius, it.gradient_max_norm) for it in g.iterations()],
                    x=[float(v) for v in g.get_blocks()[:50]]))
print("RESULT " + json.dumps(out))
"""


def _run(dev, split=1):
    env = dict(os.environ, BSGPU_LM_DEVICE=str(dev), BSGPU_TIMING="1", BSGPU_LM_DEVICE_SPLIT=str(split), BSGPU_PAIRS_BAND="1")   # (the band form whatever the size: what the large windows the mode is for take)
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):]), p.stderr


@pytest.mark.parametrize("split", [1, 0])   # 0: all eight units of the reduction in the landmark launch, its last one decides (windows whose evaluation carries no reduction)
def test_device_decision_changes_no_iteration_and_is_adopted(split):
    (a, _), (b, err) = _run(0), _run(1, split)
    assert len(a) == len(b) == 7
    kinds = set()
    for ra, rb in zip(a, b):
        assert ra["n"] == rb["n"] and len(ra["its"]) == len(rb["its"]) and ra["msg"] == rb["msg"]
        assert abs(ra["final"] - rb["final"]) <= 1e-7 * abs(ra["final"])
        for ia, ib in zip(ra["its"], rb["its"]):
            assert ia[1] == ib[1]                                       # accepted / rejected
            assert abs(ia[0] - ib[0]) <= 1e-7 * abs(ia[0])              # cost
            assert abs(ia[2] - ib[2]) <= 1e-6 * abs(ia[2])              # radius
        assert np.abs(np.array(ra["x"]) - np.array(rb["x"])).max() <= 1e-6
        kinds.update(int(it[1]) for it in ra["its"][1:])
    assert kinds == {0, 1}, "the cases are meant to hold rejected steps as well as accepted ones"
    # every refusal the host logged: the device had said "accepted" in none of them where the host accepted (same arithmetic, same bits), and
    # the host's rejected steps found the device's "not accepted" (its assembly had returned at once)
    refusals = re.findall(r"device decision not adopted: host kind (\d+) radius (\S+), device go (\S+) radius (\S+)", err)
    assert refusals, "rejected steps must show up as refusals (the log is how this test sees the device's answers)"
    for kind, r_host, go, r_dev in refusals:
        assert not (int(kind) == 1 and float(go) == 1.0), ("accepted on both sides at different radii", r_host, r_dev)
        if int(kind) == 2: assert float(go) == 0.0, "the device accepted a step the host rejected"
