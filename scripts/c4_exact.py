"""C4 (5 000 poses, 30 000 tangent dimensions) through the exact tiled factorisation instead of the block-sparse PCG:
   BSGPU_EXACT_POSE_GRAPH=1 python scripts/c4_exact.py [iterations]      (what the exact option costs)"""
import os, sys, time
os.environ.setdefault("BSGPU_EXACT_POSE_GRAPH", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import capi, gpu, synthetic
it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pr = synthetic.c4()
g = gpu.GpuSolver(0)
pr.load(g)
t0 = time.time(); g.finalize(); t1 = time.time()
o = g.options_default(); o.max_num_iterations = it; o.linear_solver_type = capi.LINEAR_SCHUR_CHOLESKY
s = g.solve(o); t2 = time.time()
print("finalize %.2f s; %d LM iterations in %.2f s = %.3f s per iteration; linear solver used %d; costs %s" %
      (t1 - t0, s.num_iterations, t2 - t1, (t2 - t1) / max(1, s.num_iterations), s.linear_solver_used, [round(i.cost, 6) for i in g.iterations()]))
g2 = gpu.GpuSolver(0)
os.environ.pop("BSGPU_EXACT_POSE_GRAPH")
pr.load(g2)
o2 = g2.options_default(); o2.max_num_iterations = it
g2.solve(o2); g2.reset_values()          # (the first solve builds the block-sparse structure)
t3 = time.time(); s2 = g2.solve(o2); t4 = time.time()
print("block-sparse PCG: %.3f s per iteration; costs %s; relative difference of the final costs %.2e" %
      ((t4 - t3) / max(1, s2.num_iterations), [round(i.cost, 6) for i in g2.iterations()], abs(s.final_cost - s2.final_cost) / s2.final_cost))
