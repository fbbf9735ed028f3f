#!/usr/bin/env python
"""Phases of one LM step (bsgpu_profile_step, HIP events in situ) on C2 / C3.   python scripts/c2_phases.py [c2|c3]"""
import sys
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
pr = synthetic.c3() if which == "c3" else synthetic.c2()
g = GpuSolver(0); pr.load(g)
o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
ph = g.profile_step(o, 30)
print(which, {k: round(v[0] * 1000, 1) for k, v in ph.items()}, "sum", round(sum(v[0] for v in ph.values()) * 1000, 1))
