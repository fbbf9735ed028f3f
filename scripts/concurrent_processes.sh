#!/bin/bash
# N independent PROCESSES (one window each) on one GPU, against N host threads in one process (scripts/concurrent_windows.py):
# separates what the HIP runtime serialises inside a process from what the GPU can overlap.
cd "$(dirname "$0")/.."
N=${1:-4}
pids=()
for i in $(seq 1 $N); do
  timeout 200 python scripts/concurrent_windows.py 1 > /tmp/cp_$i.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
python - $N <<'PY'
import sys, re
n = int(sys.argv[1]); tot = 0.0
for i in range(1, n + 1):
    m = re.search(r"(\d+) LM it/s", open("/tmp/cp_%d.log" % i).read())
    tot += float(m.group(1)) if m else 0.0
print("%d processes: %.0f LM it/s aggregate (sum of the per-process rates; the timed parts overlap only approximately)" % (n, tot))
PY
