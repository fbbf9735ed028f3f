import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from beam_slam_amd import gpu
rng = np.random.default_rng(0)
n, bw = 3000, 180
M = rng.normal(size=(n, n)); A = M @ M.T
i, j = np.indices((n, n)); A[np.abs(i - j) > bw] = 0.0
A += (np.abs(A).sum(1).max() + 1.0) * np.eye(n)
b = rng.normal(size=n)
ch = int(sys.argv[1])
for _ in range(3):
    x, ms = gpu.dense_solve(A, b, max_chains=ch)
    print(ch, ms)
