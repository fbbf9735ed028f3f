#!/usr/bin/env python
"""One small window solved repeatedly, for a kernel trace:  rocprofv3 --kernel-trace ... -- python scripts/small_window_trace.py 20 500"""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
n_kf, n_lm = int(sys.argv[1]), int(sys.argv[2])
pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620)
g = GpuSolver(0)
pr.load(g); g.finalize()
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for _ in range(30):
    g.reset_values(); s = g.solve(opt)
print(s.num_iterations)
