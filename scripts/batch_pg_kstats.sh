#!/bin/bash
# per-kernel times of 16 pose graphs of 200 poses through bsgpu_solve_batch (rocprofv3 kernel trace): bash scripts/batch_pg_kstats.sh [n]
N=${1:-16}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_bpg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_bpg -o p -- python "$ROOT/scripts/batch_pg.py" $N > /tmp/kt_bpg.log 2>&1
tail -2 /tmp/kt_bpg.log
f=$(find /tmp/kt_bpg -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s avg %9.1f us  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
