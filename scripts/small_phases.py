#!/usr/bin/env python
"""Phases of one LM step (bsgpu_profile_step, HIP events in situ) on the reference-sized windows.   python scripts/small_phases.py"""
import sys
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for n_kf, n_lm in ((20, 500), (30, 2000), (50, 5000)):
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620)
    g = GpuSolver(0); pr.load(g)
    o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
    ph = g.profile_step(o, 30)
    print(n_kf, n_lm, "tiles", g.plan_info(), {k: round(v[0] * 1000, 1) for k, v in ph.items()}, "sum", round(sum(v[0] for v in ph.values()) * 1000, 1))
