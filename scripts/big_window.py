import sys, time, os
sys.path.insert(0, "/root/repo")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
n_kf, n_lm = int(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=77)
print("generated in %.1f s, n_obs %d" % (time.perf_counter() - t0, pr.meta["n_obs"]), flush=True)
g = GpuSolver(0)
pr.load(g)
t0 = time.perf_counter(); g.finalize(); print("finalize %.1f ms" % (1e3 * (time.perf_counter() - t0)), "plan", g.plan_info(), flush=True)
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for rep in range(3):
    g.reset_values()
    t0 = time.perf_counter(); s = g.solve(opt); dt = time.perf_counter() - t0
    print("solve %.1f ms, %d it, cost %.6e -> %.6e, term %d" % (1e3 * dt, s.num_linear_solves, s.initial_cost, s.final_cost, s.termination_type), flush=True)
for it in g.iterations(): print(it.iteration, it.cost, it.step_is_successful, it.trust_region_radius)
