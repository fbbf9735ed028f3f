#!/bin/bash
# rocprofv3 evidence for one round, run on the GPU box:   bash scripts/profile_all.sh r02
#   kernel-trace + stats for C2 / C3 / C4 (same command as the bench line, without the CPU baseline leg)
#   PMC passes for C2, each in its OWN run with --kernel-trace only: HBM traffic (FETCH_SIZE, WRITE_SIZE) and MFMA utilisation
#   (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE where the counter exists)
# Summaries land in gpurun_out/prof_<round>/; the ones to be judged are copied to profiles/ by hand.
ROUND=${1:-r02}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$ROUND"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c4; do
  rm -rf /tmp/kt_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$w -o p -- python "$ROOT/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-past-l3 > "$OUT/bench_$w.log" 2>&1
  f=$(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${ROUND}_${w}_kernel_stats.csv"
done
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-past-l3 > "$OUT/pmc_$c.log" 2>&1
done
python - "$OUT" "$ROUND" <<'PY'
import csv, sys, collections, glob
out, rnd = sys.argv[1], sys.argv[2]
cs = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]
acc = {c: collections.defaultdict(float) for c in cs}
cnt = {c: collections.Counter() for c in cs}
for c in cs:
    fs = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0]
        acc[c][k] += float(r["Counter_Value"]); cnt[c][k] += 1
keys = sorted(set(k for c in cs for k in acc[c]), key=lambda k: -(acc["FETCH_SIZE"].get(k, 0) + acc["WRITE_SIZE"].get(k, 0)))
with open("%s/%s_c2_pmc_hbm.csv" % (out, rnd), "w") as f:
    f.write("Kernel,Launches,avg_FETCH_SIZE_raw_KB,avg_WRITE_SIZE_raw_KB,note: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section)\n")
    for k in keys:
        n = cnt["FETCH_SIZE"][k]
        if n and k.strip():
            f.write('"%s",%d,%.1f,%.1f\n' % (k, n, acc["FETCH_SIZE"][k] / n, acc["WRITE_SIZE"].get(k, 0.0) / max(1, cnt["WRITE_SIZE"][k])))
with open("%s/%s_c2_pmc_mfma.csv" % (out, rnd), "w") as f:
    f.write("Kernel,Launches,avg_SQ_INSTS_VALU_MFMA_MOPS_F64,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_SQ_BUSY_CYCLES,avg_SQ_INSTS_VALU,avg_GRBM_GUI_ACTIVE\n")
    for k in sorted(set(k for c in cs[2:] for k in acc[c]), key=lambda k: -acc["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, 0)):
        n = max(cnt[c][k] for c in cs[2:])
        if not n or not k.strip():
            continue
        f.write('"%s",%d,%s\n' % (k, n, ",".join("%.1f" % (acc[c][k] / max(1, cnt[c][k])) if cnt[c][k] else "" for c in cs[2:])))
print(open("%s/%s_c2_pmc_mfma.csv" % (out, rnd)).read()[:1500])
PY
ls -la "$OUT"
