#!/bin/bash
# HBM traffic per kernel of the inverse-depth window (scripts/idp_scale.py) from two separate PMC passes (FETCH_SIZE, WRITE_SIZE; each with
# --kernel-trace only), averaged per launch -> gpurun_out/r03_idp_pmc_hbm.csv.   bash scripts/idp_pmc_hbm.sh [n_kf] [n_lm]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
KF=${1:-60}; LM=${2:-20000}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_idp_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_idp_$c -o p -- python "$ROOT/scripts/idp_scale.py" $KF $LM > /tmp/pmc_idp_$c.log 2>&1
done
mkdir -p "$ROOT/gpurun_out"
python - "$ROOT" <<'PY'
import csv, sys, collections, glob
root = sys.argv[1]
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = {c: collections.Counter() for c in acc}
for c in acc:
    fs = glob.glob("/tmp/pmc_idp_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        print("no counter file for", c); sys.exit(1)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        acc[c][k] += float(r["Counter_Value"]); cnt[c][k] += 1
rows = sorted(acc["FETCH_SIZE"], key=lambda k: -(acc["FETCH_SIZE"][k] + acc["WRITE_SIZE"].get(k, 0)))
path = root + "/gpurun_out/r03_idp_pmc_hbm.csv"
with open(path, "w") as out:
    out.write("Kernel,Launches,avg_FETCH_SIZE_raw_KB,avg_WRITE_SIZE_raw_KB,note: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section); separate PMC passes; scripts/idp_scale.py 60 20000\n")
    for k in rows:
        n = cnt["FETCH_SIZE"][k]
        if n == 0 or not k.strip(): continue
        out.write('"%s",%d,%.1f,%.1f\n' % (k, n, acc["FETCH_SIZE"][k] / n, acc["WRITE_SIZE"].get(k, 0.0) / max(1, cnt["WRITE_SIZE"][k])))
print(open(path).read()[:2500])
PY
