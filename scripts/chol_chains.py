#!/usr/bin/env python
"""Chain tasks of a fused-Cholesky probe file (BSGPU_CHOL_PROBE=<file>): when each chain's tiles were final, when it ended.
    python scripts/chol_chains.py <probe file>"""
import sys
rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    v = [int(x) for x in line.split()]
    rows.append(v)
t0 = min(r[7] for r in rows if r[7])
us = lambda t: (t - t0) / 100.0
ch = [r for r in rows if r[4] & 1]
ch.sort(key=lambda r: r[13])
print("chain first-tile tiles | ticket taken  tiles final  end   (duration)")
for r in ch:
    print("  k %3d  m %d | %7.1f %7.1f %7.1f  (%5.1f)" % (r[1], r[2], us(r[7]), us(r[9]), us(r[13]), us(r[13]) - us(r[9])))
side = [r for r in rows if r[4] & 32]
print("side tasks: %d; span %.1f us" % (len(side), us(max(r[13] for r in rows))))
