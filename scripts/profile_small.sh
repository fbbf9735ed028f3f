#!/bin/bash
# kernel-trace stats of the reference-sized configurations (profiles/r06_ref20x500_kernel_stats.csv, r06_batch32_kernel_stats.csv, r06_lio20_kernel_stats.csv):
#   a lone 20 KF x 500 landmark window, 32 of them through one bsgpu_solve_batch, a lone lidar-inertial window of 20 key frames
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_small"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ref_one.py <<PY
import sys
sys.path.insert(0, "$ROOT")
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.lio_window(n_kf=20, n_rel=300, seed=20250620) if len(sys.argv) > 1 and sys.argv[1] == "lio" else synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620)
g = GpuSolver(0); pr.load(g); g.finalize()
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for _ in range(40): g.reset_values(); g.solve(opt)
PY
rm -rf /tmp/ks_a /tmp/ks_b /tmp/ks_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_a -o p -- python /tmp/ref_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_c -o p -- python /tmp/ref_one.py lio > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_b -o p -- python "$ROOT/scripts/batch_windows.py" 32 > /dev/null 2>&1
cp "$(find /tmp/ks_a -name '*kernel_stats.csv' | head -1)" "$OUT/r06_ref20x500_kernel_stats.csv"
cp "$(find /tmp/ks_c -name '*kernel_stats.csv' | head -1)" "$OUT/r06_lio20_kernel_stats.csv"
cp "$(find /tmp/ks_b -name '*kernel_stats.csv' | head -1)" "$OUT/r06_batch32_kernel_stats.csv"
ls -la "$OUT"
