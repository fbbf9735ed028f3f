import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from beam_slam_amd import gpu
rng = np.random.default_rng(0)
for n in [16, 40, 64, 100, 130, 300, 1000, 3000]:
    M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n); b = 50.0 * rng.normal(size=n)   # |L^-1 b| > 1: exercises the rhs-row pivot
    xr = np.linalg.solve(A, b)
    for v1 in (True, False):
        try:
            x, ms = gpu.dense_solve(A, b, use_v1=v1)
            print("n=%5d v1=%d rel err %.2e  time %.3f ms" % (n, v1, np.abs(x - xr).max() / np.abs(xr).max(), ms))
        except Exception as e:
            print("n=%5d v1=%d FAILED %s" % (n, v1, e))
