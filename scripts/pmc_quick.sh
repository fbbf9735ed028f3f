#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one workload's kernels, top rows:  bash scripts/pmc_quick.sh c2 [extra env assignments]
W=${1:-c2}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq_$c
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq_$c -o p -- python "$ROOT/bench.py" --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-past-l3 --no-other-configs --sustained-seconds 0 > /dev/null 2>&1
done
python - <<'PY'
import csv, collections, glob
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = {c: collections.Counter() for c in acc}
for c in acc:
    fs = glob.glob("/tmp/pq_%s/**/*counter_collection.csv" % c, recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        acc[c][k] += float(r["Counter_Value"]); cnt[c][k] += 1
keys = sorted(acc["FETCH_SIZE"], key=lambda k: -(acc["FETCH_SIZE"][k] + acc["WRITE_SIZE"].get(k, 0)))
for k in keys[:10]:
    n = cnt["FETCH_SIZE"][k]
    f, w = acc["FETCH_SIZE"][k] / n, acc["WRITE_SIZE"].get(k, 0) / max(1, cnt["WRITE_SIZE"][k])
    print("%-50s launches %4d  fetch %9.1f KB (x2 = %9.1f)  write %9.1f KB  -> traffic %8.1f MB" % (k[:50], n, f, 2 * f, w, (2 * f + w) / 1024))
PY
