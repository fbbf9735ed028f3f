"""Reads a rocprofv3 kernel-trace CSV and prints, for the steady-state part of the run, the busy time, the idle gaps between
consecutive kernels (by the kernel that FOLLOWS the gap) and the per-kernel totals — what bounds an LM iteration."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bsg::", "")) for r in rows))
ev = ev[len(ev) // 2:]          # second half: warm-up and the CPU baseline are over
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.Counter(); gapn = collections.Counter(); dur = collections.Counter(); cnt = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = max(0, s1 - e0)
    gaps[n1] += g; gapn[n1] += 1
for s, e, n in ev: dur[n] += e - s; cnt[n] += 1
n_it = cnt.get(sys.argv[2] if len(sys.argv) > 2 else "landmark_kernel", 1)   # a kernel launched once per LM iteration
print("kernels %d, LM iterations %d, span %.1f us/it, busy %.1f us/it, idle %.1f us/it" % (len(ev), n_it, span / n_it / 1e3, busy / n_it / 1e3, (span - busy) / n_it / 1e3))
print("%-44s %8s %10s %10s" % ("kernel", "calls/it", "busy us/it", "gap-before us/it"))
for n, d in dur.most_common(30):
    print("%-44s %8.1f %10.1f %10.1f" % (n[:44], cnt[n] / n_it, d / n_it / 1e3, gaps[n] / n_it / 1e3))
