#!/bin/bash
# SQ / TCP counters of ONE kernel of the C2 bench run, each group of counters in its own rocprofv3 pass (--kernel-trace only):
#   bash scripts/pmc_kernel.sh pairs_kernel "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY ..." "SQ_INSTS_VALU ..."
# prints the per-launch average of every counter for kernels whose name contains the first argument.
K=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=${BSGPU_PMC_WORKLOAD:-c2}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pk_$i -o p -- python "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-past-l3 > /tmp/pk_$i.log 2>&1
  python - /tmp/pk_$i "$K" <<'PY'
import csv, sys, glob, collections
d, k = sys.argv[1], sys.argv[2]
fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for c in acc:
    print("%-34s %16.1f  (%d launches)" % (c, acc[c] / cnt[c], cnt[c]))
if not acc:
    print("no rows for", k, "in", d, fs)
PY
done
