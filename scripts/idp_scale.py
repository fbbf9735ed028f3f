#!/usr/bin/env python
"""A window with many inverse-depth landmarks through the exact path: LM it/s with the landmark tiles ordered first (default)
and, for comparison, left behind the keyframes (BSGPU_NO_LEAF_TILES=1).   python scripts/idp_scale.py [n_kf] [n_lm]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import capi, synthetic
from beam_slam_amd.gpu import GpuSolver

n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_lm = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pr = synthetic.idp_window(n_kf=n_kf, n_lm=n_lm, seed=22)
g = GpuSolver(0)
pr.load(g)
t0 = time.perf_counter(); g.finalize(); t1 = time.perf_counter()
cold = t1 - t0
g.clear(); pr.load(g)       # what a sliding window pays: the context's device buffers come from its pool
t0 = time.perf_counter(); g.finalize(); t1 = time.perf_counter()
opt = g.options_default(); opt.max_num_iterations = 10
for _ in range(2):
    g.reset_values(); s = g.solve(opt)
t2 = time.perf_counter(); n = 0
for _ in range(3):
    g.reset_values(); s = g.solve(opt); n += s.num_linear_solves
dt = time.perf_counter() - t2
print("%d keyframes x %d inverse-depth landmarks (%d factors), leaf tiles %s: finalize %.1f ms (first call of the context: %.1f), %.1f LM it/s, %.2f ms/solve (%d it), cost %.6e -> %.6e, plan %s" % (
    n_kf, n_lm, pr.n_factors(capi.F_IDP_REPROJ) + pr.n_factors(capi.F_IDP_REPROJ_UNARY), "off" if os.environ.get("BSGPU_NO_LEAF_TILES") else "first",
    1e3 * (t1 - t0), 1e3 * cold, n / dt, 1e3 * dt / 3, s.num_iterations, s.initial_cost, s.final_cost, g.plan_info()))
