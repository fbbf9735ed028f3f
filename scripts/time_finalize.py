"""Times the host-side flattening (set_blocks / add_factors / finalize) of a window next to its solve.
BSGPU_TIMING=1 prints the phases of finalize()."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
pr = getattr(synthetic, which)()
g = GpuSolver(0)
for rep in range(3):
    t0 = time.perf_counter()
    pr.load(g)
    t1 = time.perf_counter()
    g.finalize()
    t2 = time.perf_counter()
    s = g.solve(g.options_vio_unclipped() if hasattr(g, "options_vio_unclipped") else None)
    t3 = time.perf_counter()
    print(f"{which} rep {rep}: load {1e3 * (t1 - t0):.1f} ms  finalize {1e3 * (t2 - t1):.1f} ms  solve {1e3 * (t3 - t2):.1f} ms ({s.num_iterations} it)", flush=True)
