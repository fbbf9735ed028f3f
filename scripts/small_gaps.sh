# kernel trace of the reference-sized window's solves: busy / idle per LM iteration and per kernel (scripts/trace_gaps.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kts
cat > /tmp/sw.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620)
g = GpuSolver(0); pr.load(g); g.finalize()
opt = g.options_vio(); opt.max_solver_time_in_seconds = 0.0
for _ in range(40):
    g.reset_values(); g.solve(opt)
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o p -- python /tmp/sw.py $GRAFT_REPO_ROOT > /dev/null 2>&1
f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f | head -16
