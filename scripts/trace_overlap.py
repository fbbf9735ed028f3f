#!/usr/bin/env python
"""How much do the kernels of several windows solved concurrently on one GPU actually overlap?  Reads a rocprofv3 --kernel-trace csv
(scripts/concurrent_windows.py under the profiler) and prints, for the measured part of the run: busy time, the time with >= 2 / 3 / 4
kernels in flight, the kernels per queue, and the per-kernel mean duration next to a single-window trace if one is given.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o p -- python scripts/concurrent_windows.py 2
    python scripts/trace_overlap.py /tmp/kt2/*/p_kernel_trace.csv [single_window_kernel_trace.csv]
"""
import csv
import sys
import collections


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Queue_Id", "?")))
    rows.sort()
    return rows


rows = load(sys.argv[1])
t_end = rows[-1][1]
t_beg = rows[0][0] + int(0.6 * (t_end - rows[0][0]))      # the last 40 %: past set-up and warm-up
rows = [r for r in rows if r[0] >= t_beg]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    depth += d; last = t
span = ev[-1][0] - ev[0][0]
print("span %.2f ms, kernels %d, queues %s" % (span / 1e6, len(rows), sorted(set(r[3] for r in rows))))
for k in sorted(hist):
    print("  %d kernels in flight: %5.1f %%" % (k, 100.0 * hist[k] / span))
dur = collections.defaultdict(list)
for s, e, n, _ in rows:
    dur[n].append(e - s)
ref = {}
if len(sys.argv) > 2:
    r1 = load(sys.argv[2])
    t1 = r1[0][0] + int(0.6 * (r1[-1][1] - r1[0][0]))
    d1 = collections.defaultdict(list)
    for s, e, n, _ in r1:
        if s >= t1:
            d1[n].append(e - s)
    ref = {n: sum(v) / len(v) for n, v in d1.items()}
print("  %-48s %7s %9s %9s" % ("kernel", "calls", "mean us", "alone us"))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-48s %7d %9.1f %9s" % (n, len(v), sum(v) / len(v) / 1e3, ("%.1f" % (ref[n] / 1e3)) if n in ref else ""))
