#!/usr/bin/env python
"""A window of the reference's size whose start is bad enough for rejected steps (the guesses of the assembly ahead miss): LM it/s.
    BSGPU_LM_DEVICE=0 python scripts/rejected_steps.py ; python scripts/rejected_steps.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for n_kf, n_lm in ((20, 500), (50, 5000)):
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620)
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0; opt.max_num_iterations = 20
    rng = np.random.default_rng(3)
    x0 = pr.values + 0.05 * rng.standard_normal(pr.values.size)
    def run():
        g.set_values(x0); return g.solve(opt)
    for _ in range(3): s = run()
    t0 = time.perf_counter(); n = 0
    for _ in range(10):
        s = run(); n += s.num_linear_solves
    dt = time.perf_counter() - t0
    its = g.iterations()
    print("%3d KF x %5d: %7.0f LM it/s (%d iterations, %d rejected; includes set_values)" % (n_kf, n_lm, n / dt, s.num_iterations, sum(1 for it in its[1:] if not it.step_is_successful)))
    g.close()
