#!/usr/bin/env python
"""Stress of the host's wait for the end-of-step scalars (the mirror's stamp, bsgpu_solve.cpp: fetch_scalars): thousands of short solves of
small windows on several threads; every solve must take the same LM trajectory as the first one of its window.
   python scripts/stamp_stress.py [solves per thread] [threads]"""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = []

def work(i):
    pr = synthetic.vio_window(n_kf=12 + 3 * i, n_lm=300 + 100 * i, seed=100 + i)
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
    ref = None
    for k in range(n):
        g.reset_values(); s = g.solve(opt)
        sig = (s.num_iterations, s.termination_type, tuple(int(it.step_is_successful) for it in g.iterations()))
        if ref is None:
            ref = (sig, s.final_cost)
        elif sig != ref[0] or abs(s.final_cost - ref[1]) > 1e-9 * abs(ref[1]):
            bad.append((i, k, sig, s.final_cost, ref))

th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
for t in th: t.start()
for t in th: t.join()
print("stamp stress: %d threads x %d solves, %d deviations" % (nt, n, len(bad)))
for b in bad[:5]: print(b)
sys.exit(1 if bad else 0)
