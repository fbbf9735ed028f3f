"""Aggregate LM iterations/s of N independent windows on ONE GPU through bsgpu_solve_batch — the batched launches (one set per LM
iteration for all windows, csrc/bsgpu_batch.cpp) against the thread-per-window form (BSGPU_BATCH_THREADS=1) and the lone solve.
    python scripts/batch_windows.py --size 20:500 1 8 32 64
The windows are the reference's own sizes (vio.yaml:3,56: tens of key frames) or C2-shaped (BASELINE config 5 on one device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver

args = sys.argv[1:]
n_kf, n_lm = 20, 500
if args and args[0] == "--size":
    n_kf, n_lm = (int(v) for v in args[1].split(":")); args = args[2:]
counts = [int(a) for a in args] or [1, 8, 32, 64]
steps = 30 if n_kf <= 60 else 10
windows = [synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250630 + i) for i in range(max(counts))]
solvers = []
for pr in windows:
    g = GpuSolver(0); pr.load(g)
    if max(counts) > 1 and not os.environ.get("BSGPU_BATCH_LATENCY_PLANS"): g.set_plan_preference(True)   # (BSGPU_PLAN_THROUGHPUT: one of many)
    g.finalize(); solvers.append(g)
opt = solvers[0].options_vio(); opt.max_solver_time_in_seconds = 0.0
mode = "thread per window" if os.environ.get("BSGPU_BATCH_THREADS") else "batched launches"
print("windows of %d key frames x %d landmarks, bsgpu_solve_batch: %s" % (n_kf, n_lm, mode), flush=True)
for n in counts:
    sv = solvers[:n]
    for _ in range(3):
        for g in sv: g.reset_values()
        (GpuSolver.solve_batch(sv, opt) if n > 1 else [sv[0].solve(opt)])
    its = 0
    w0, r0 = GpuSolver.batch_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g in sv: g.reset_values()
        ss = GpuSolver.solve_batch(sv, opt) if n > 1 else [sv[0].solve(opt)]
        its += sum(s.num_linear_solves for s in ss)
    dt = time.perf_counter() - t0
    w1, r1 = GpuSolver.batch_stats()
    print("%3d windows: %8.0f LM it/s aggregate, %7.3f ms per call, %5.1f LM iterations per window and call, %d of %d window-solves batched, %.1f us per round"
          % (n, its / dt, 1e3 * dt / steps, its / steps / n, w1 - w0, n * steps, 1e6 * dt / max(1, r1 - r0)), flush=True)
