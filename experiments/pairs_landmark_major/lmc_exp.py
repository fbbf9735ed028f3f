"""C2's LM step phases in situ (bsgpu_profile_step) with the camera blocks assembled landmark-major (BSGPU_PAIRS_LM=1) or by
camera-pair segments (=0); BSGPU_LMC_PROBE=1 prints pairs_lm_kernel's cycles per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.c2()
g = GpuSolver(0); pr.load(g)
print("assembly (segments, groups, batches)", g.assembly_info())
o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
ph = g.profile_step(o, 20)
print({k: round(v[0] * 1000, 1) for k, v in ph.items()})
s = g.solve(o)
print("final cost", s.final_cost, "iterations", s.num_iterations)
