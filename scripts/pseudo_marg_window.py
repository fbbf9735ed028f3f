#!/usr/bin/env python
"""A 20-keyframe window with its oldest keyframes held constant (pseudo_marginalization: true, the shipped configuration:
bs_optimizers/src/fixed_lag_smoother.cpp:244-262) against the same window with every keyframe free.   python scripts/pseudo_marg_window.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for hold in (0, 3):
    pr = synthetic.vio_window(n_kf=20, n_lm=500, seed=5)
    kf = pr.meta["kf_blocks"]
    for k in range(hold):
        for b in kf[k]: pr.is_const[int(b)] = 1
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
    for _ in range(3): g.reset_values(); g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(30): g.reset_values(); n += g.solve(opt).num_linear_solves
    dt = time.perf_counter() - t0
    print("held-constant keyframes %d: %.0f LM it/s, %.1f us per iteration" % (hold, n / dt, 1e6 * dt / n))
