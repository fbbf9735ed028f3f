#!/usr/bin/env python
"""C2 with the iterative step (BSGPU_LINEAR_SCHUR_PCG) beside the exact one: LM it/s, inner iterations, final costs.
    python scripts/schur_pcg_bench.py [inner_tolerance]"""
import sys, time
sys.path.insert(0, ".")
from beam_slam_amd import capi, synthetic
from beam_slam_amd.gpu import GpuSolver

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
pr = synthetic.c2()
g = GpuSolver(0)
pr.load(g)
g.finalize()
for name, lin in (("cholesky", capi.LINEAR_SCHUR_CHOLESKY), ("schur_pcg", capi.LINEAR_SCHUR_PCG)):
    opt = g.options_vio()
    opt.max_solver_time_in_seconds = 0.0
    opt.linear_solver_type = lin
    opt.pcg_tolerance = tol
    opt.pcg_max_iterations = 2000
    for _ in range(2):
        g.reset_values(); s = g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(5):
        g.reset_values(); s = g.solve(opt); n += s.num_linear_solves
    dt = time.perf_counter() - t0
    print("%-10s %8.1f LM it/s  %6.2f ms/solve  inner iterations/solve %5d  final cost %.12e" % (name, n / dt, 1e3 * dt / 5, s.num_inner_iterations, s.final_cost))
