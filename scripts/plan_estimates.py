import sys, os
sys.path.insert(0, ".")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
ws = {"c2": synthetic.c2, "c3": synthetic.c3, "20x500": lambda: synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620), "30x2000": lambda: synthetic.vio_window(n_kf=30, n_lm=2000, seed=20250620),
      "50x5000": lambda: synthetic.vio_window(n_kf=50, n_lm=5000, seed=20250620), "lio20": lambda: synthetic.lio_window(n_kf=20, n_rel=300, seed=20250620)}
for k, f in ws.items():
    sys.stderr.write("WINDOW %s\n" % k); sys.stderr.flush()
    g = GpuSolver(0); f().load(g); g.finalize(); g.close()
