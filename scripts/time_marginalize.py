"""Times bsgpu_marginalize (true marginalisation of the oldest keyframe) on a mid-size window."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic, capi
from beam_slam_amd.gpu import GpuSolver

n_kf, n_lm = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 4000
pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=5)
kf, lmb = pr.meta["kf_blocks"], pr.meta["lm_blocks"]
idx = np.concatenate([c[0] for c in pr.factors[capi.F_REPROJ]])
seen0 = set(int(v) for v in idx[idx[:, 0] == int(kf[0, 0]), 2])
counts = np.bincount(idx[:, 2], minlength=pr.n_blocks)
first_only = [l for l in seen0 if set(idx[idx[:, 2] == l][:, 0]) == {int(kf[0, 0])}]
marg = [int(b) for b in kf[0]] + first_only
g = GpuSolver(0)
pr.load(g)
g.solve()
for rep in range(3):
    t0 = time.perf_counter()
    kept, A, b, xbar = g.marginalize(marg, pr.size)
    dt = time.perf_counter() - t0
    print(f"window {n_kf} KF x {n_lm} lm: keyframe 0 sees {len(seen0)} landmarks; marginalising {len(marg)} blocks -> prior on {kept.size} blocks, A {A.shape}, {1e3 * dt:.1f} ms", flush=True)
pm = pr.marginalized(marg, kept, A, b, xbar, values=g.get_blocks())
g2 = GpuSolver(0)
pm.load(g2)
t0 = time.perf_counter()
s = g2.solve()
print(f"marginalised window: {s.num_iterations} LM iterations in {1e3 * (time.perf_counter() - t0):.1f} ms (incl. finalize), cost {s.initial_cost:.4e} -> {s.final_cost:.4e}, usable {s.is_solution_usable}, tangent dims pose-side {g2.tangent_offset(int(pm.meta['lm_blocks'][-1]))}")
