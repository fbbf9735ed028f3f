#!/usr/bin/env python
"""The graph of tangent blocks of a synthetic window's reduced camera system, as scripts/dim_order_tool.cpp reads it (what
bsgpu_finalize.cpp collects in its BlockGraph): pose-side blocks in block order; two blocks are coupled if a pose-only factor names
both, or if their camera poses share an eliminated landmark.  Usage: dim_order_graph.py c1|c2|c3|vio:<kf>:<lm> > file"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beam_slam_amd import synthetic, capi
from beam_slam_amd.problem import NIDX

which = sys.argv[1]
if which == "c1": pr = synthetic.c1()
elif which == "c2": pr = synthetic.c2()
elif which == "c3": pr = synthetic.c3()
elif which.startswith("pg:"):
    _, npose, nloop = which.split(":"); pr = synthetic.pose_graph(n_pose=int(npose), n_loop=int(nloop), seed=20250900)
else:
    _, kf, lm = which.split(":"); pr = synthetic.vio_window(n_kf=int(kf), n_lm=int(lm), seed=1)
nb = len(pr.size)
is_const = np.array(pr.is_const, bool)
tables = {}
for t, lst in pr.factors.items():
    if not lst: continue
    tables[t] = np.concatenate([np.asarray(i, np.int64).reshape(-1, NIDX[t]) for (i, *_rest) in lst], 0)
nvar = {capi.F_REPROJ: 3, capi.F_IMU_DELTA: 10, capi.F_IMU_PRIOR: 5, capi.F_RELPOSE_EXT: 6, capi.F_RELPOSE: 4, capi.F_ABSPOSE: 2,
        capi.F_ABS_VEC3: 1, capi.F_REL_VEC3: 2, capi.F_GRAVITY: 1}
lm_use = np.zeros(nb, int); other_use = np.zeros(nb, int)
for t, tab in tables.items():
    for sl in range(nvar[t]):
        if t == capi.F_REPROJ and sl == 2: np.add.at(lm_use, tab[:, sl], 1)
        else: np.add.at(other_use, tab[:, sl], 1)
size = np.array(pr.size); man = np.array(pr.manifold)
is_lm = (lm_use > 0) & (other_use == 0) & (size == 3) & ~is_const
pose = ~is_const & ~is_lm & (lm_use + other_use > 0)
tsize = np.where(man == capi.MANIFOLD_QUAT_RIGHT, 3, size)
bid = -np.ones(nb, int); t0 = []; w = []; t = 0
for b in range(nb):
    if pose[b]: bid[b] = len(t0); t0.append(t); w.append(int(tsize[b])); t += int(tsize[b])
edges = set()
for tt, tab in tables.items():
    if tt == capi.F_REPROJ:
        tab = tab[is_lm[tab[:, 2]]]
        order = np.argsort(tab[:, 2], kind="stable"); tab = tab[order]
        lm = tab[:, 2]; starts = np.flatnonzero(np.r_[True, lm[1:] != lm[:-1], True])
        pairs = set()
        for a, b in zip(starts[:-1], starts[1:]):
            q = np.unique(tab[a:b, 0] * nb + tab[a:b, 1])
            for x in q:
                for y in q: pairs.add((int(x), int(y)))
        for x, y in pairs:
            for bx in (x // nb, x % nb):
                for by in (y // nb, y % nb):
                    if bid[bx] >= 0 and bid[by] >= 0 and bx != by: edges.add((bid[bx], bid[by]))
    else:
        cols = [tab[:, sl] for sl in range(nvar[tt])]
        for i in range(len(cols)):
            for j in range(i):
                m = (bid[cols[i]] >= 0) & (bid[cols[j]] >= 0)
                for a, b in set(zip(bid[cols[i]][m].tolist(), bid[cols[j]][m].tolist())): edges.add((a, b))
print(len(t0))
for a, b in zip(t0, w): print(a, b)
print(len(edges))
for a, b in edges: print(a, b)
