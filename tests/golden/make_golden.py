"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference itself cannot be built or run in this environment, so the vectors come
from the oracle AFTER it passed the reference's own known-answer tests (tests/test_oracle_reference_kats.py)
and the finite-difference / scipy cross-checks.  Each .npz holds the inputs (flat IR) and the expected
outputs: cost, residuals, gradient, per-iteration LM costs / radii / accept flags, final values.

    python tests/golden/make_golden.py [case ...]      (no arguments: every case)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from beam_slam_amd import synthetic  # noqa: E402
from helpers import mixed_problem  # noqa: E402
from oracle import Oracle  # noqa: E402

CASES = {
    "all_types_seed0": (lambda: mixed_problem(0, consistent=True), 25),
    "all_types_seed7_const": (lambda: mixed_problem(7, hold_first=True, consistent=True), 25),
    "vio_window_6kf_60lm": (lambda: synthetic.vio_window(n_kf=6, n_lm=60, seed=101, track_min=3, track_max=6), 25),
    "lio_window_12kf": (lambda: synthetic.lio_window(n_kf=12, n_rel=80, seed=102), 25),
    "pose_graph_40": (lambda: synthetic.pose_graph(n_pose=40, n_loop=80, seed=103), 25),
    "idp_window_8kf_60lm": (lambda: synthetic.idp_window(n_kf=8, n_lm=60, seed=104), 25),
}


def main():
    for name, (make, iters) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        pr = make()
        o = Oracle(threads=1)
        pr.load(o)
        cost, r, g, _ = o.evaluate()
        opt = o.options_default()
        opt.max_num_iterations = iters
        s = o.solve(opt)
        its = o.iterations()
        out = pr.to_arrays()
        out.update(exp_cost=cost, exp_residuals=r, exp_gradient=g, exp_final_cost=s.final_cost,
                   exp_termination=s.termination_type, exp_max_iterations=iters,
                   exp_iter_cost=np.array([i.cost for i in its]), exp_iter_ok=np.array([i.step_is_successful for i in its]),
                   exp_iter_radius=np.array([i.trust_region_radius for i in its]), exp_final_values=o.get_blocks())
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-32s n_res %5d n_tan %4d  cost %.6e -> %.6e  (%d its)  %6.1f KB" % (
            name, r.size, g.size, cost, s.final_cost, len(its) - 1, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
