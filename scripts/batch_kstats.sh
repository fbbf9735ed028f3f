#!/bin/bash
# per-kernel times of the batched launches (rocprofv3 kernel trace): bash scripts/batch_kstats.sh 20:500 32
SIZE=${1:-20:500}; N=${2:-32}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_batch
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_batch -o p -- python "$ROOT/scripts/batch_windows.py" --size $SIZE $N > /tmp/kt_batch.log 2>&1
tail -2 /tmp/kt_batch.log
f=$(find /tmp/kt_batch -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %6s avg %9.1f us  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
