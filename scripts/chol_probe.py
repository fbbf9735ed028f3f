#!/usr/bin/env python
"""Reads the per-task stamps of the fused Cholesky (BSGPU_CHOL_PROBE=<file>, k_chol.hip) and prints the critical path:
starting from the task that publishes last, follow the dependency (L_kk, A tiles, turn on the C tile) that was met last.

    BSGPU_CHOL_PROBE=gpurun_out/chol_probe.txt python bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python scripts/chol_probe.py gpurun_out/chol_probe.txt
"""
import sys
import collections

rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    v = [int(x) for x in line.split()]
    rows.append(dict(i=v[0], k=v[1], ti=v[2], tj=v[3], flags=v[4], need=v[5], tot=v[6], deq=v[7], got=v[8], deps=v[9], loaded=v[10],
                     solved=v[11], updated=v[12], pub=v[13], wg=v[14]))
t0 = min(r["deq"] for r in rows if r["deq"])
us = lambda t: (t - t0) / 100.0
end = max(r["pub"] for r in rows)
print("tasks %d, workgroups %d, span %.1f us" % (len(rows), len(set(r["wg"] for r in rows)), us(end)))
# producers: potrf_done[t] <- potrf-only task t, or the update task with ti == tj == t and need + 1 == tot;
#            tile (a, b) at count c <- the update task of (a, b) with need == c - 1
potrf_by = {}
upd_by = {}
split_by = collections.defaultdict(list)   # (ti, tj) -> kFusedSplit chunks (flags & 256): they feed the chain that owns the tile
for r in rows:
    if r["flags"] & 256:
        split_by[(r["ti"], r["tj"])].append(r)
        continue
    if r["flags"] & 1:   # a chain task: k = first tile, ti = number of tiles (its tiles' flags are set as it goes; the stamp is its end)
        for t in range(r["k"], r["k"] + r["ti"]):
            potrf_by[t] = r
    else:
        if r["need"] >= 0 and not (r["flags"] & 256):
            upd_by[(r["ti"], r["tj"], r["need"] + 1)] = r
tot = collections.Counter()
for r in rows:
    if not (r["flags"] & (1 | 256)) and r["need"] >= 0:
        tot[(r["ti"], r["tj"])] = max(tot[(r["ti"], r["tj"])], r["tot"])


def deps_of(r):
    if r["flags"] & (64 | 128):   # LM-diagonal / rider tasks: first update of their tile, nothing before them (not stamped)
        return []
    if r["flags"] & 1:   # the last updates of the chain's own tiles
        d = []
        for a in range(r["k"], r["k"] + r["ti"]):
            for b in range(r["k"], a + 1):
                c = tot.get((a, b), 0)
                if c and (a, b, c) in upd_by:
                    d.append(("C(%d,%d)" % (a, b), upd_by[(a, b, c)]))
                for ch in split_by.get((a, b), []):
                    d.append(("P(%d,%d)k%d" % (a, b, ch["k"]), ch))
        return d
    if r["flags"] & 256:
        d = []
        if r["k"] in potrf_by:
            d.append(("L%d" % r["k"], potrf_by[r["k"]]))
        for a in (r["ti"], r["tj"]):
            c = tot.get((a, r["k"]), 0)
            if c and (a, r["k"], c) in upd_by:
                d.append(("A(%d,%d)" % (a, r["k"]), upd_by[(a, r["k"], c)]))
        return d
    d = []
    if r["k"] in potrf_by:
        d.append(("L%d" % r["k"], potrf_by[r["k"]]))
    for a in (r["ti"], r["tj"]):
        c = tot.get((a, r["k"]), 0)
        if c and (a, r["k"], c) in upd_by:
            d.append(("A(%d,%d)" % (a, r["k"]), upd_by[(a, r["k"], c)]))
    if r["need"] > 0 and (r["ti"], r["tj"], r["need"]) in upd_by:
        d.append(("C(%d,%d)#%d" % (r["ti"], r["tj"], r["need"]), upd_by[(r["ti"], r["tj"], r["need"])]))
    return d


cur = max(rows, key=lambda r: r["pub"])
path = []
seen = set()
while cur is not None and cur["i"] not in seen:
    seen.add(cur["i"])
    d = deps_of(cur)
    nxt = max(d, key=lambda x: x[1]["pub"]) if d else None
    path.append((cur, nxt[0] if nxt else "-", nxt[1]["pub"] if nxt else 0))
    cur = nxt[1] if nxt else None
path.reverse()
print("critical path (%d tasks):" % len(path))
print("  task      k  ti  tj  wg | deq   got  deps(wait) loaded solved product+turn  rmw/potrf/publish | last dep            dep-published  hop")
for r, name, dpub in path:
    hop = (r["deps"] - dpub) / 100.0 if dpub else 0.0
    print("  %4d %s %3d %3d %3d %3d | %6.1f %5.1f %6.1f %6.1f %6.1f %6.1f %7.1f | %-18s %8.1f %6.1f" % (
        r["i"], "C" if r["flags"] & 1 else ("S" if r["flags"] & 256 else ("x" if r["flags"] & 6 else " ")), r["k"], r["ti"], r["tj"], r["wg"],
        us(r["deq"]), (r["got"] - r["deq"]) / 100.0, (r["deps"] - r["got"]) / 100.0 if r["deps"] else 0, (r["loaded"] - r["deps"]) / 100.0 if r["deps"] else 0,
        (r["solved"] - r["loaded"]) / 100.0 if r["loaded"] else 0, (r["updated"] - r["solved"]) / 100.0, (r["pub"] - r["updated"]) / 100.0, name,
        us(dpub) if dpub else 0, hop))
# aggregate: where a task's time goes
agg = collections.defaultdict(float)
n = 0
for r in rows:
    if r["flags"] & (1 | 64 | 128 | 256):
        continue
    n += 1
    agg["dequeue"] += (r["got"] - r["deq"]) / 100.0
    agg["wait"] += (r["deps"] - r["got"]) / 100.0
    agg["load"] += (r["loaded"] - r["deps"]) / 100.0
    agg["solve"] += (r["solved"] - r["loaded"]) / 100.0
    agg["update"] += (r["updated"] - r["solved"]) / 100.0
    agg["publish(+potrf)"] += (r["pub"] - r["updated"]) / 100.0
print("mean per update task (us):", {k: round(v / n, 2) for k, v in agg.items()})
