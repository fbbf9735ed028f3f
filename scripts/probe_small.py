import sys
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620)
g = GpuSolver(0); pr.load(g); g.finalize()
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for _ in range(8):
    g.reset_values(); s = g.solve(opt)
prof = g.profile_step(opt, reps=20)
print({k: round(1e3*v[0],2) for k,v in prof.items()})
