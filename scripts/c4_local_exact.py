"""The exact step (the reference's SPARSE_NORMAL_CHOLESKY, submap_pose_graph_optimization.cpp:144-146) on a pose graph of C4's size whose
loop closures are spatially local (synthetic.pose_graph_local) — next to the same call on C4 itself, whose 45 000 uniformly random
loop closures make the reduced system fill in completely (scripts/c4_exact.py: 0.94 s per LM iteration), and to the block-sparse PCG:
   python scripts/c4_local_exact.py [iterations] [row_len]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import capi, gpu, synthetic
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
row_len = int(sys.argv[2]) if len(sys.argv) > 2 else 25
pr = synthetic.pose_graph_local(row_len=row_len)
g = gpu.GpuSolver(0)
pr.load(g)
t0 = time.time(); g.finalize(); t1 = time.time()
print("finalize (exact path) %.2f s; plan (chains, steps, tiles) %s" % (t1 - t0, g.plan_info()), flush=True)
o = g.options_default(); o.max_num_iterations = it      # BSGPU_LINEAR_AUTO: finalize() has chosen the exact path for this graph
g.solve(o); g.reset_values()
t1 = time.time(); s = g.solve(o); t2 = time.time()
print("linear solver used %d (1 = exact tiled factorisation, 2 = PCG): %d LM iterations, %.2f ms per iteration (%.0f LM it/s); costs %.6f -> %.6f" %
      (s.linear_solver_used, s.num_iterations, 1e3 * (t2 - t1) / max(1, s.num_linear_solves), s.num_linear_solves / (t2 - t1), s.initial_cost, s.final_cost), flush=True)
g2 = gpu.GpuSolver(0)
pr.load(g2)
o2 = g2.options_default(); o2.max_num_iterations = it; o2.linear_solver_type = capi.LINEAR_PCG
g2.solve(o2); g2.reset_values()
t3 = time.time(); s2 = g2.solve(o2); t4 = time.time()
print("block-sparse PCG: %d LM iterations, %.2f ms per iteration (%.0f LM it/s), %d inner iterations; final cost %.6f; relative difference %.2e" %
      (s2.num_iterations, 1e3 * (t4 - t3) / max(1, s2.num_linear_solves), s2.num_linear_solves / (t4 - t3), s2.num_inner_iterations, s2.final_cost,
       abs(s.final_cost - s2.final_cost) / s2.final_cost))
