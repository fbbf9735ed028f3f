import sys, time
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for n_kf, n_lm in ((60, 8000), (70, 10000), (80, 13000), (90, 16000)):
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620)
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
    for _ in range(3): g.reset_values(); s = g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(20): g.reset_values(); s = g.solve(opt); n += s.num_linear_solves
    dt = time.perf_counter() - t0
    print("%4d KF x %6d (%7d factors): %7.0f LM it/s (%d it)" % (n_kf, n_lm, pr.n_factors(0), n / dt, s.num_iterations))
    g.close()
