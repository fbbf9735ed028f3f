# kernel trace of C2's solves: busy / idle per LM iteration and per kernel (scripts/trace_gaps.py), and one iteration's timeline
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktc
W=${1:-c2}
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktc -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-past-l3 > /tmp/ktc.log 2>&1
f=$(find /tmp/ktc -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/trace_gaps.py $f | head -16
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bsg::", "")) for r in rows))
# the last 30 kernels of a mid-run solve
idx = [i for i, e in enumerate(ev) if e[2].startswith("landmark")]
i0 = idx[len(idx) // 2]
t0 = ev[i0][0]
for s, e, n in ev[i0 - 3:i0 + 45]:
    print("%9.1f %9.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n[:50]))
PY
