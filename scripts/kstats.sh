#!/bin/bash
# per-kernel average durations of one bench workload:  bash scripts/kstats.sh c2 [n_rows]
W=${1:-c2}; N=${2:-12}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$W
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$W -o p -- python "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-past-l3 > /tmp/ks_$W.log 2>&1
f=$(find /tmp/ks_$W -name "*kernel_stats.csv" | head -1)
python - "$f" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print("%-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -o '"value": [0-9.]*' /tmp/ks_$W.log | head -1
