#!/usr/bin/env python
"""Quick check of the reduced-system solver alone: random banded SPD systems through bsgpu_dense_solve vs numpy."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import gpu
if os.environ.get('BSGPU_LIB_OVERRIDE'): gpu.LIB_PATH = os.environ['BSGPU_LIB_OVERRIDE']

rng = np.random.default_rng(0)
for n, bw in [(100, 100), (300, 120), (1000, 200), (3000, 180)]:
    A = np.zeros((n, n))
    for i in range(n):
        j0 = max(0, i - bw)
        A[i, j0:i + 1] = rng.standard_normal(i + 1 - j0)
    A = A @ A.T + n * np.eye(n) * 0.1
    b = rng.standard_normal(n)
    t0 = time.time()
    x, ms = gpu.dense_solve(A, b)
    ref = np.linalg.solve(A, b)
    print("n=%d bw=%d  rel err %.2e  device ms %.3f  wall %.2f s" % (n, bw, np.abs(x - ref).max() / np.abs(ref).max(), ms, time.time() - t0), flush=True)
