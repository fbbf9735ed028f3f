#!/usr/bin/env python
"""16 pose graphs of 200 poses through bsgpu_solve_batch (the bench leg other_configs.submap_pose_graphs), for a kernel trace:
    python scripts/batch_pg.py [n_windows] [local]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
local = len(sys.argv) > 2
gs = []
for w in range(n):
    pr = synthetic.pose_graph_local(n_pose=200, n_loop=300, row_len=10, seed=20250900 + w) if local else synthetic.pose_graph(n_pose=200, n_loop=300, seed=20250900 + w)
    g = GpuSolver(0); pr.load(g); g.finalize(); gs.append(g)
opt = gs[0].options_default(); opt.max_num_iterations = 10
for _ in range(2):
    for g in gs: g.reset_values()
    GpuSolver.solve_batch(gs, opt)
w0, r0 = GpuSolver.batch_stats()
t0 = time.perf_counter(); it = 0
for _ in range(10):
    for g in gs: g.reset_values()
    it += sum(s.num_linear_solves for s in GpuSolver.solve_batch(gs, opt))
dt = time.perf_counter() - t0
w1, r1 = GpuSolver.batch_stats()
print("%d windows: %.0f LM it/s aggregate, %.1f us per round, %d rounds" % (n, it / dt, 1e6 * dt / max(1, r1 - r0), r1 - r0))
g = gs[0]
for _ in range(3): g.reset_values(); g.solve(opt)
t0 = time.perf_counter(); it1 = 0
for _ in range(20): g.reset_values(); it1 += g.solve(opt).num_linear_solves
dt1 = time.perf_counter() - t0
print("alone: %.0f LM it/s, %.1f us per iteration; ratio %.2f" % (it1 / dt1, 1e6 * dt1 / it1, (it / dt) / (it1 / dt1)))
