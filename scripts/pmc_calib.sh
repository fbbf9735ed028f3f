#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of scripts/pmc_calib.hip's kernels (known byte counts in landmark_kernel's and pairs_band_kernel's access patterns),
# each counter in its own rocprofv3 pass with --kernel-trace only.  Run on the GPU box: bash scripts/pmc_calib.sh [out.csv]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${1:-$ROOT/gpurun_out/pmc_calib.csv}
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT";; esac
BIN=$ROOT/scripts/pmc_calib.bin
[ -x "$BIN" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$ROOT/scripts/pmc_calib.hip" -o "$BIN"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/cal_$c -o p -- "$BIN" > /tmp/cal_$c.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
known = {  # kernel -> (bytes read, bytes written) per launch
    "calib_wide_copy": (400690 * 64, 400690 * 64),
    "calib_rows48(": (400690 * 64, 0),
    "calib_rows48_twice<false>": (400690 * 64, 400690 * 64),
    "calib_rows48_twice<true>": (400690 * 64, 400690 * 64),
    "calib_zero_tiles": (0, 600 * 32768),
    "calib_rows176": (400690 * 176, 0),
    "calib_atomics": (0, 200 * 6400 * 8),
}
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = {c: collections.Counter() for c in acc}
for c in acc:
    for f in glob.glob("/tmp/cal_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            for k in known:
                if k in r["Kernel_Name"]:
                    acc[c][k] += float(r["Counter_Value"]); cnt[c][k] += 1
with open(sys.argv[1], "w") as out:
    out.write("kernel,launches,known_read_MB,known_write_MB,FETCH_SIZE_raw_MB,WRITE_SIZE_raw_MB,FETCH_raw_over_known_read,WRITE_raw_over_known_write  (counter values are KB of 1024 B; MB = 1e6 B)\n")
    for k, (rd, wr) in known.items():
        n = max(1, cnt["FETCH_SIZE"][k])
        fe = acc["FETCH_SIZE"][k] / n * 1024 / 1e6
        we = acc["WRITE_SIZE"][k] / max(1, cnt["WRITE_SIZE"][k]) * 1024 / 1e6
        out.write("%s,%d,%.2f,%.2f,%.2f,%.2f,%s,%s\n" % (k.rstrip("("), cnt["FETCH_SIZE"][k], rd / 1e6, wr / 1e6, fe, we,
                                                     "%.3f" % (fe * 1e6 / rd) if rd else "", "%.3f" % (we * 1e6 / wr) if wr else ""))
print(open(sys.argv[1]).read())
PY
