import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from beam_slam_amd import gpu
rng = np.random.default_rng(0)
def banded(n, bw):
    M = rng.normal(size=(n, n)); A = M @ M.T
    i, j = np.indices((n, n)); A[np.abs(i - j) > bw] = 0.0
    return A + (np.abs(A).sum(1).max() + 1.0) * np.eye(n)
for n, bw in [(16, None), (40, None), (64, None), (100, None), (300, None), (1000, None), (1500, 120), (3000, 180), (3000, None), (6000, 180)]:
    A = banded(n, bw) if bw else (lambda M: M @ M.T + n * np.eye(n))(rng.normal(size=(n, n)))
    b = 50.0 * rng.normal(size=n)
    xr = np.linalg.solve(A, b)
    for ch in (1, 2, 4, 8):
        if bw is None and ch > 1: continue
        try:
            x, ms = gpu.dense_solve(A, b, max_chains=ch)
            print("n=%5d bw=%s chains<=%d rel err %.2e  time %.3f ms" % (n, bw, ch, np.abs(x - xr).max() / np.abs(xr).max(), ms))
        except Exception as e:
            print("n=%5d bw=%s chains<=%d FAILED %s" % (n, bw, ch, e))
