#!/bin/bash
# HBM traffic per kernel from two separate PMC passes (FETCH_SIZE, WRITE_SIZE), averaged per launch -> profiles/r01_c2_pmc_hbm.csv
# (run on the GPU box: bash scripts/pmc_hbm.sh; counters are collected in their own runs, with --kernel-trace only)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - "$ROOT" <<'PY'
import csv, sys, collections, glob
root = sys.argv[1]
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = {c: collections.Counter() for c in acc}
for c in acc:
    f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % c)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        acc[c][k] += float(r["Counter_Value"]); cnt[c][k] += 1
rows = sorted(acc["FETCH_SIZE"], key=lambda k: -(acc["FETCH_SIZE"][k] + acc["WRITE_SIZE"].get(k, 0)))
with open(root + "/gpurun_out/r01_c2_pmc_hbm.csv", "w") as out:
    out.write("Kernel,Launches,avg_FETCH_SIZE_raw_KB,avg_WRITE_SIZE_raw_KB,note: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section)\n")
    for k in rows:
        n = cnt["FETCH_SIZE"][k]
        if n == 0 or not k.strip(): continue
        out.write('"%s",%d,%.1f,%.1f\n' % (k, n, acc["FETCH_SIZE"][k] / n, acc["WRITE_SIZE"].get(k, 0.0) / max(1, cnt["WRITE_SIZE"][k])))
print(open(root + "/gpurun_out/r01_c2_pmc_hbm.csv").read()[:1800])
PY
