import sys
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
for n_kf, n_lm in ((20, 500), (200, 50000)):
  for imu in (True, False):
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=n_lm, seed=20250620, with_imu=imu)
    g = GpuSolver(0); pr.load(g)
    o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
    ph = g.profile_step(o, 30)
    print(n_kf, n_lm, "imu", imu, {k: round(v[0] * 1000, 1) for k, v in ph.items()})
