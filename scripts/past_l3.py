"""The reprojection Jacobian evaluation on a window the 256 MiB Infinity Cache cannot hold (800 keyframes x 300 000 landmarks, ~2.4 M
observations, ~490 MB per launch): 20 launches.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE by scripts/profile_all.sh for the
counter evidence behind bench.py's roofline.past_l3."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
big = synthetic.vio_window(n_kf=800, n_lm=300000, seed=20250621)
g = GpuSolver(0)
big.load(g)
g.finalize()
ms = g.time_reproj_jacobian_ms(20)
nb = g.reproj_jacobian_bytes()
print("past_l3: %d observations, %d bytes per launch, %.5f ms per launch, %.1f GB/s" % (big.n_factors(0), nb, ms, nb / ms / 1e6))
