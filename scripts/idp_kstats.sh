#!/bin/bash
# per-kernel average durations of the inverse-depth window (scripts/idp_scale.py):  bash scripts/idp_kstats.sh [n_kf] [n_lm] [csv out]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
KF=${1:-60}; LM=${2:-20000}; OUT=${3:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_idp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_idp -o p -- python "$ROOT/scripts/idp_scale.py" $KF $LM > /tmp/ks_idp.log 2>&1
tail -1 /tmp/ks_idp.log
f=$(find /tmp/ks_idp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel stats"; tail -5 /tmp/ks_idp.log; exit 1; }
[ -n "$OUT" ] && cp "$f" "$OUT"
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-60s calls %5s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
