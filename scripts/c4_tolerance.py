#!/usr/bin/env python
"""C4 (5 000 poses, 50 000 constraints, block-sparse PCG): what the inner tolerance buys and costs — LM trajectory against the 1e-12 one,
inner iterations, LM iterations / s.   python scripts/c4_tolerance.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.c4()
ref = None
for tol in (1e-12, 1e-10, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_default(); opt.max_num_iterations = 10; opt.pcg_tolerance = tol; opt.pcg_max_iterations = 3000
    s = g.solve(opt)
    costs = np.array([i.cost for i in g.iterations()]); acc = [i.step_is_successful for i in g.iterations()]
    x = g.get_blocks().copy()
    for _ in range(2): g.reset_values(); g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(4): g.reset_values(); n += g.solve(opt).num_linear_solves
    dt = time.perf_counter() - t0
    if ref is None: ref = (costs, acc, x, s.final_cost)
    same = acc == ref[1] and len(costs) == len(ref[0])
    dc = np.abs(costs - ref[0]).max() / ref[0].max() if same else float("nan")
    print("tol %.0e: %4d inner its (%5.1f / LM step), %6.0f LM it/s | decisions %s, max |dcost|/cost %.2e, final cost rel %.2e, max |dx| %.2e" % (
        tol, s.num_inner_iterations, s.num_inner_iterations / max(1, s.num_linear_solves), n / dt, "same" if same else "DIFFER", dc,
        abs(s.final_cost - ref[3]) / ref[3], np.abs(x - ref[2]).max()))
    g.close()
