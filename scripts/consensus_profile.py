import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from beam_slam_amd import synthetic, sharding, capi
from beam_slam_amd.gpu import GpuSolver
N, nkf, nlm = 2, 100, 25000
wins = [synthetic.chain_window(r, N, n_kf=nkf, n_lm=nlm, seed=20250620) for r in range(N)]
mps = []
for r, w in enumerate(wins):
    g = GpuSolver(0)
    opt = g.options_default(); opt.max_num_iterations = 30
    opt.function_tolerance = 1e-12; opt.gradient_tolerance = 1e-12; opt.parameter_tolerance = 1e-12
    mps.append(sharding.MessagePassing(g, w, r, w.meta["shared"], opt))
acc = {}
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(cls, name, g)
for n in ("get_blocks", "covariance_joint", "update_marginal", "solve"):
    wrap(capi.Solver, n)
wrap(sharding.MessagePassing, "solve_and_summarise")
hist = sharding.message_passing_rounds(mps, 10, tol=1e-8)
acc.clear()
for m in mps: m.reset()
t0 = time.perf_counter()
hist = sharding.message_passing_rounds(mps, 10, tol=1e-8)
tot = time.perf_counter() - t0
nr = len(hist) * N
print("rounds %d; per window-round: total %.3f ms" % (len(hist), 1e3 * tot / nr))
for k, v in acc.items(): print("  %-22s %.3f ms" % (k, 1e3 * v / nr))
