"""beam_slam_amd — MI355X-native fixed-lag-smoother solve path (drop-in for the
``graph->optimize()`` call of beam_slam's ``bs_optimizers::FixedLagSmoother``).

Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI of include/bsgpu.h -> libbsgpu.so
  host/        C++ host side mirroring the reference's fuse::Graph / Constraint / Optimizer surface
  capi.py      ctypes declarations of include/bsgpu.h
  gpu.py       loader of libbsgpu.so (fails loudly when the HIP library is missing)
  problem.py   flat factor-graph IR container
  synthetic.py BASELINE.json workloads C1..C5
"""
from . import capi  # noqa: F401
from .problem import Problem  # noqa: F401
