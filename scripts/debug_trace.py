"""Debug helper (not a test): per-iteration LM traces of the HIP path and the oracle side by side."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import Oracle
from beam_slam_amd.gpu import GpuSolver
from beam_slam_amd import synthetic
from helpers import mixed_problem

def run(name, pr, iters=12):
    g = GpuSolver(0); o = Oracle(threads=4)
    pr.load(g); pr.load(o)
    opt = g.options_default(); opt.max_num_iterations = iters
    sg, so = g.solve(opt), o.solve(opt)
    print("==", name, "n_tan", g.num_parameters_tangent(), "gpu:", sg.message.decode(), "| oracle:", so.message.decode())
    for a, b in zip(g.iterations(), o.iterations()):
        print("  it %2d ok %d/%d cost %.10e %.10e rel %.1e | mcc %.6e %.6e | rho %.4f %.4f | step %.3e %.3e | gmax %.3e %.3e" % (
            a.iteration, a.step_is_successful, b.step_is_successful, a.cost, b.cost, abs(a.cost - b.cost) / abs(b.cost),
            a.model_cost_change, b.model_cost_change, a.relative_decrease, b.relative_decrease, a.step_norm, b.step_norm,
            a.gradient_max_norm, b.gradient_max_norm))
    print("  final x diff", np.abs(g.get_blocks() - o.get_blocks()).max(), "time gpu %.4f s (dev %.4f) oracle %.4f s" % (sg.total_time_in_seconds, sg.device_time_in_seconds, so.total_time_in_seconds))

which = sys.argv[1:] or ["mixed0", "hold", "pg", "c1"]
if "mixed0" in which: run("mixed0", mixed_problem(0, n_state=5, n_lm=30))
if "hold" in which: run("hold", mixed_problem(5, hold_first=True))
if "pg" in which: run("pg", synthetic.pose_graph(n_pose=300, n_loop=900, seed=12))
if "c1" in which: run("c1", synthetic.c1())
if "c2" in which: run("c2", synthetic.c2(), iters=10)
