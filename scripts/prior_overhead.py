#!/usr/bin/env python
"""What a dense marginal prior (true marginalisation: one MarginalConstraint per window after the first slide) costs per LM iteration:
the same window with and without the prior of its marginalised first keyframe.   python scripts/prior_overhead.py [n_kf] [n_lm]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic, capi
from beam_slam_amd.gpu import GpuSolver

n_kf, n_lm = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 500
pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=5)
kf = pr.meta["kf_blocks"]
idx = np.concatenate([c[0] for c in pr.factors[capi.F_REPROJ]])
seen0 = set(int(v) for v in idx[idx[:, 0] == int(kf[0, 0]), 2])
first_only = [l for l in seen0 if set(idx[idx[:, 2] == l][:, 0]) == {int(kf[0, 0])}]
marg = [int(b) for b in kf[0]] + first_only
g = GpuSolver(0); pr.load(g); g.solve()
kept, A, b, xbar = g.marginalize(marg, pr.size)
# (the window with the prior starts where the window without it starts: from the optimum — `values=g.get_blocks()` — it converges in ONE iteration,
# and what is then divided by one is a solve's fixed cost, not an iteration's: the + 73 % of rounds 3 and 4 was that)
pm = pr.marginalized(marg, kept, A, b, xbar, values=g.get_blocks() if os.environ.get("PRIOR_AT_OPTIMUM") else None)

def rate(p, label):
    s = GpuSolver(0); p.load(s); s.finalize()
    opt = s.options_vio(); opt.max_solver_time_in_seconds = 0.0
    for _ in range(3): s.reset_values(); s.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(30): s.reset_values(); n += s.solve(opt).num_linear_solves
    dt = time.perf_counter() - t0
    print("%-44s %7.0f LM it/s, %.1f us per iteration, %.1f iterations per solve" % (label, n / dt, 1e6 * dt / n, n / 30.0))
    if os.environ.get("PRIOR_PHASES"):
        ph = s.profile_step(opt, 30)
        print("   plan", s.plan_info(), {k: round(v[0] * 1000, 1) for k, v in ph.items()}, "sum", round(sum(v[0] for v in ph.values()) * 1000, 1))
    return 1e6 * dt / n

if not os.environ.get("PRIOR_ONLY"):
    a = rate(pr, "%d KF x %d landmarks" % (n_kf, n_lm))
b_ = rate(pm, "... first keyframe marginalised (prior %dx%d)" % A.shape)
