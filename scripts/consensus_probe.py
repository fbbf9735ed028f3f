import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from beam_slam_amd import synthetic, sharding
from beam_slam_amd.gpu import GpuSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nkf, nlm = int(sys.argv[2]), int(sys.argv[3])
t0 = time.perf_counter()
wins = [synthetic.chain_window(r, N, n_kf=nkf, n_lm=nlm, seed=20250620) for r in range(N)]
print("generated %.2f s" % (time.perf_counter() - t0))
mps = []
for r, w in enumerate(wins):
    g = GpuSolver(0)
    opt = g.options_default(); opt.max_num_iterations = 30
    opt.function_tolerance = 1e-12; opt.gradient_tolerance = 1e-12; opt.parameter_tolerance = 1e-12
    mps.append(sharding.MessagePassing(g, w, r, w.meta["shared"], opt))
orig = sharding.MessagePassing.solve_and_summarise
log = []
def timed(self):
    t0 = time.perf_counter(); out = orig(self); dt = time.perf_counter() - t0
    log.append((self.pid, dt, self.last_summary.total_time_in_seconds, self.last_summary.num_iterations))
    return out
sharding.MessagePassing.solve_and_summarise = timed
t0 = time.perf_counter()
hist = sharding.message_passing_rounds(mps, 10, tol=1e-8)
print("rounds %d in %.1f ms" % (len(hist), 1e3 * (time.perf_counter() - t0)))
for h in hist: print(h)
for l in log: print("win %d: %.2f ms total, solve %.2f ms (%d it), overhead %.2f ms" % (l[0], 1e3*l[1], 1e3*l[2], l[3], 1e3*(l[1]-l[2])))
mp, maps = synthetic.merge_chain(wins)
m = GpuSolver(0); mp.load(m)
o = m.options_default(); o.max_num_iterations = 60; o.function_tolerance = 1e-14; o.gradient_tolerance = 1e-14; o.parameter_tolerance = 1e-14
best = m.solve(o)
print("merged optimum %.9f (%d it); consensus own-cost sum %.9f rel %.2e" % (best.final_cost, best.num_iterations, hist[-1][2], abs(hist[-1][2]-best.final_cost)/best.final_cost))
