#!/bin/bash
# many dense factorisations side by side with / without row segments (BSGPU_CHOL_ROWS=0/1): 16 pose graphs of 200 poses, 8 C2 windows, 32 lidar-inertial windows
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
for i in 1 2; do
  for v in 0 1; do
    echo "== BSGPU_CHOL_ROWS=$v"
    BSGPU_CHOL_ROWS=$v timeout 300 python scripts/batch_pg.py 16 2>&1 | tail -2
    BSGPU_CHOL_ROWS=$v timeout 300 python scripts/batch_windows.py --size 200:50000 8 2>&1 | tail -1
    BSGPU_CHOL_ROWS=$v timeout 300 python scripts/batch_windows.py --size 20:500 32 2>&1 | tail -1
  done
done
