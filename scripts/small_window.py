#!/usr/bin/env python
"""LM iterations/s on the small windows the reference's VIO actually runs (C1: 20 keyframes x 500 landmarks, and a few sizes up):
per-iteration latency, where launch overheads and the dependent kernel chain dominate.   python scripts/small_window.py"""
import sys, time
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver

for n_kf, n_lm in ((20, 500), (30, 2000), (50, 5000), (100, 20000)):
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620)
    g = GpuSolver(0)
    pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
    for _ in range(3):
        g.reset_values(); s = g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(20):
        g.reset_values(); s = g.solve(opt); n += s.num_linear_solves
    dt = time.perf_counter() - t0
    print("%4d KF x %6d landmarks (%7d factors): %7.0f LM it/s, %.3f ms per iteration, %.2f ms per solve (%d it)" % (
        n_kf, n_lm, pr.n_factors(0), n / dt, 1e3 * dt / n, 1e3 * dt / 20, s.num_iterations))
    g.close()
