import sys, time
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for name, pr in (("lidar-inertial 20 KF", synthetic.lio_window(n_kf=20, n_rel=300, seed=20250620)), ("pose graph 200", synthetic.pose_graph(n_pose=200, n_loop=400, seed=7))):
    g = GpuSolver(0); pr.load(g); g.finalize()
    opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
    for _ in range(5): g.reset_values(); s = g.solve(opt)
    t0 = time.perf_counter(); n = 0
    for _ in range(30): g.reset_values(); s = g.solve(opt); n += s.num_linear_solves
    dt = time.perf_counter() - t0
    print("%s: %.0f LM it/s (%d it)" % (name, n / dt, s.num_iterations))
