import sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620)
g = GpuSolver(0); pr.load(g); g.finalize()
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for _ in range(8):
    g.reset_values(); g.solve(opt)
print(g.plan_info())
