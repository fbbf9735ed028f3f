"""Aggregate LM iterations/s of several independent windows solved concurrently on ONE GPU: one bsgpu context (own HIP
stream) per host thread — the shape of the reference's local + global mapper processes and of submap refinement
(bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115), on a single device.  Not the bench.py metric."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver

# usage: concurrent_windows.py [--size KF:LM] [counts ...]     (default 200:50000 and 1 2 4 8)
args = sys.argv[1:]
n_kf, n_lm, steps = 200, 50000, 20
if args and args[0] == "--size":
    n_kf, n_lm = (int(v) for v in args[1].split(":")); args = args[2:]
    steps = 50 if n_kf <= 60 else 20
counts = [int(a) for a in args] or [1, 2, 4, 8]
print("windows of %d key frames x %d landmarks" % (n_kf, n_lm), flush=True)
windows = [synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250630 + i) for i in range(max(counts))]
solvers = []
for pr in windows:
    g = GpuSolver(0); pr.load(g); g.finalize()
    solvers.append(g)
opt = solvers[0].options_vio(); opt.max_solver_time_in_seconds = 0.0
for n in counts:
    its = [0] * n
    barrier = threading.Barrier(n + 1)
    def work(i):
        g = solvers[i]
        for _ in range(3): g.reset_values(); g.solve(opt)
        barrier.wait()
        for _ in range(steps):
            g.reset_values(); its[i] += g.solve(opt).num_linear_solves
        barrier.wait()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th: t.start()
    barrier.wait(); t0 = time.perf_counter(); barrier.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    print("%d concurrent windows: %.0f LM it/s aggregate, %.2f ms per solve per window" % (n, sum(its) / dt, 1e3 * dt / steps), flush=True)

# the same through the batched entry point (bsgpu_solve_batch: the library's threads, one call per round of solves)
for n in counts:
    sv = solvers[:n]
    for _ in range(3):
        for g in sv: g.reset_values()
        GpuSolver.solve_batch(sv, opt)
    its = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        for g in sv: g.reset_values()
        its += sum(s.num_linear_solves for s in GpuSolver.solve_batch(sv, opt))
    dt = time.perf_counter() - t0
    print("%d windows per bsgpu_solve_batch call: %.0f LM it/s aggregate, %.2f ms per call" % (n, its / dt, 1e3 * dt / steps), flush=True)
