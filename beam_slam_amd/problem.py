"""Flat factor-graph IR (the payload of include/bsgpu.h) as numpy arrays.

This is the Python mirror of what ``fuse_graphs::HashGraph::createProblem`` hands to Ceres
(SURVEY.md Appendix B): a parameter-block table plus per-type factor tables.  It only stores and
forwards data; all arithmetic happens behind the C-ABI.
"""
import numpy as np

from . import capi

NIDX = {capi.F_REPROJ: 4, capi.F_REPROJ_ONLINE_CALIB: 6, capi.F_IMU_DELTA: 10, capi.F_IMU_PRIOR: 5,
        capi.F_RELPOSE_EXT: 6, capi.F_RELPOSE: 4, capi.F_ABSPOSE: 2, capi.F_ABS_VEC3: 1,
        capi.F_REL_VEC3: 2, capi.F_GRAVITY: 1, capi.F_IDP_REPROJ: 6, capi.F_IDP_REPROJ_UNARY: 4}
NCONST = {capi.F_REPROJ: 3, capi.F_REPROJ_ONLINE_CALIB: 3, capi.F_IMU_DELTA: 287, capi.F_IMU_PRIOR: 241,
          capi.F_RELPOSE_EXT: 43, capi.F_RELPOSE: 43, capi.F_ABSPOSE: 43, capi.F_ABS_VEC3: 12,
          capi.F_REL_VEC3: 12, capi.F_GRAVITY: 7, capi.F_IDP_REPROJ: 6, capi.F_IDP_REPROJ_UNARY: 6}
NRES = {capi.F_REPROJ: 2, capi.F_REPROJ_ONLINE_CALIB: 2, capi.F_IMU_DELTA: 15, capi.F_IMU_PRIOR: 15,
        capi.F_RELPOSE_EXT: 6, capi.F_RELPOSE: 6, capi.F_ABSPOSE: 6, capi.F_ABS_VEC3: 3,
        capi.F_REL_VEC3: 3, capi.F_GRAVITY: 2, capi.F_IDP_REPROJ: 2, capi.F_IDP_REPROJ_UNARY: 2}


class Problem:
    def __init__(self):
        self._values = []
        self.offset, self.size, self.manifold, self.is_const = [], [], [], []
        self._nvalues = 0
        self.cameras = []
        self.factors = {}  # type -> list of (idx, consts, loss_kind, loss_a)
        self.marginals = []  # (blocks, A, b, xbar): fuse_constraints::MarginalConstraint payloads
        self.meta = {}

    # -- blocks ----------------------------------------------------------------------------
    def add_block(self, values, manifold=capi.MANIFOLD_EUCLIDEAN, const=False):
        v = np.asarray(values, np.float64).ravel()
        self.offset.append(self._nvalues)
        self.size.append(v.size)
        self.manifold.append(manifold)
        self.is_const.append(1 if const else 0)
        self._values.append(v)
        self._nvalues += v.size
        return len(self.offset) - 1

    def add_blocks(self, values2d, manifold=capi.MANIFOLD_EUCLIDEAN, const=False):
        """Adds n blocks of equal size at once; returns their indices."""
        v = np.ascontiguousarray(values2d, np.float64)
        n, k = v.shape
        first = len(self.offset)
        self.offset.extend((self._nvalues + k * np.arange(n)).tolist())
        self.size.extend([k] * n)
        self.manifold.extend([manifold] * n)
        self.is_const.extend([1 if const else 0] * n)
        self._values.append(v.ravel())
        self._nvalues += n * k
        return np.arange(first, first + n, dtype=np.int32)

    def add_quat(self, q_wxyz, const=False):
        return self.add_block(q_wxyz, capi.MANIFOLD_QUAT_RIGHT, const)

    @property
    def values(self):
        if len(self._values) != 1:
            self._values = [np.concatenate(self._values) if self._values else np.zeros(0)]
        return self._values[0]

    @values.setter
    def values(self, v):
        v = np.ascontiguousarray(v, np.float64)
        assert v.size == self._nvalues
        self._values = [v.copy()]

    @property
    def n_blocks(self):
        return len(self.offset)

    def block(self, b, values=None):
        v = self.values if values is None else values
        return v[self.offset[b]:self.offset[b] + self.size[b]]

    # -- cameras / factors -----------------------------------------------------------------
    def add_camera(self, fx, fy, cx, cy, R_cam_baselink, t_cam_baselink):
        c = capi.Camera()
        c.fx, c.fy, c.cx, c.cy = fx, fy, cx, cy
        c.R_cam_baselink[:] = list(np.asarray(R_cam_baselink, float).ravel())
        c.t_cam_baselink[:] = list(np.asarray(t_cam_baselink, float).ravel())
        self.cameras.append(c)
        return len(self.cameras) - 1

    def add_factors(self, ftype, idx, consts, loss_kind=capi.LOSS_TRIVIAL, loss_a=1.0):
        idx = np.atleast_2d(np.asarray(idx, np.int32))
        consts = np.atleast_2d(np.asarray(consts, np.float64))
        assert idx.shape[1] == NIDX[ftype], (idx.shape, NIDX[ftype])
        assert consts.shape == (idx.shape[0], NCONST[ftype]), (consts.shape, NCONST[ftype])
        n = idx.shape[0]
        lk = np.ascontiguousarray(np.broadcast_to(np.asarray(loss_kind, np.int32), (n,)))
        la = np.ascontiguousarray(np.broadcast_to(np.asarray(loss_a, np.float64), (n,)))
        self.factors.setdefault(ftype, []).append((idx, consts, lk, la))

    def add_marginal(self, blocks, A, b, xbar):
        """Dense linear prior r = b + sum_i A_i (x_i [-] xbar_i) over `blocks` (include/bsgpu.h: bsgpu_add_marginal)."""
        blocks = np.asarray(blocks, np.int32).ravel()
        A = np.atleast_2d(np.asarray(A, np.float64))
        cols = sum(3 if self.manifold[b_] == capi.MANIFOLD_QUAT_RIGHT else self.size[b_] for b_ in blocks)
        assert A.shape[1] == cols and np.size(b) == A.shape[0] and np.size(xbar) == sum(self.size[b_] for b_ in blocks)
        self.marginals.append((blocks, A, np.asarray(b, np.float64).ravel(), np.asarray(xbar, np.float64).ravel()))

    def n_factors(self, ftype=None):
        if ftype is None:
            return sum(self.n_factors(t) for t in self.factors)
        return sum(f[0].shape[0] for f in self.factors.get(ftype, []))

    def n_residuals(self):
        return sum(self.n_factors(t) * NRES[t] for t in self.factors) + sum(m[1].shape[0] for m in self.marginals)

    # -- (de)serialisation: small fixtures under tests/golden/ ----------------------------------
    def to_arrays(self):
        d = dict(values=self.values, offset=np.asarray(self.offset, np.int32), size=np.asarray(self.size, np.uint8),
                 manifold=np.asarray(self.manifold, np.uint8), is_const=np.asarray(self.is_const, np.uint8))
        d["cameras"] = np.array([[c.fx, c.fy, c.cx, c.cy, *c.R_cam_baselink, *c.t_cam_baselink] for c in self.cameras]).reshape(-1, 16)
        for t, chunks in self.factors.items():
            d[f"f{t}_idx"] = np.concatenate([c[0] for c in chunks])
            d[f"f{t}_consts"] = np.concatenate([c[1] for c in chunks])
            d[f"f{t}_loss_kind"] = np.concatenate([c[2] for c in chunks])
            d[f"f{t}_loss_a"] = np.concatenate([c[3] for c in chunks])
        for i, (blocks, A, b, xbar) in enumerate(self.marginals):
            d[f"m{i}_blocks"], d[f"m{i}_A"], d[f"m{i}_b"], d[f"m{i}_xbar"] = blocks, A, b, xbar
        return d

    @classmethod
    def from_arrays(cls, d):
        pr = cls()
        pr._values = [np.asarray(d["values"], np.float64).copy()]
        pr._nvalues = pr._values[0].size
        pr.offset = [int(v) for v in d["offset"]]
        pr.size = [int(v) for v in d["size"]]
        pr.manifold = [int(v) for v in d["manifold"]]
        pr.is_const = [int(v) for v in d["is_const"]]
        for row in np.asarray(d["cameras"]).reshape(-1, 16):
            pr.add_camera(row[0], row[1], row[2], row[3], row[4:13], row[13:16])
        for t in range(capi.F_NUM_TYPES):
            if f"f{t}_idx" in d:
                pr.add_factors(t, d[f"f{t}_idx"], d[f"f{t}_consts"], d[f"f{t}_loss_kind"], d[f"f{t}_loss_a"])
        i = 0
        while f"m{i}_blocks" in d:
            pr.add_marginal(d[f"m{i}_blocks"], d[f"m{i}_A"], d[f"m{i}_b"], d[f"m{i}_xbar"])
            i += 1
        return pr

    # -- true marginalisation (the transaction fuse_constraints::marginalizeVariables returns) -------
    def connected_factors(self, blocks):
        """(type, chunk, row) of every fixed-size factor and index of every marginal factor touching `blocks`."""
        bs = set(int(b) for b in blocks)
        nvar = {t: NIDX[t] - (1 if t in (capi.F_REPROJ, capi.F_REPROJ_ONLINE_CALIB, capi.F_IDP_REPROJ, capi.F_IDP_REPROJ_UNARY) else 0)
                for t in NIDX}
        fixed = [(t, ci, r) for t, chunks in self.factors.items() for ci, ch in enumerate(chunks)
                 for r in range(ch[0].shape[0]) if bs & set(int(v) for v in ch[0][r, :nvar[t]])]
        marg = [i for i, m in enumerate(self.marginals) if bs & set(int(v) for v in m[0])]
        return fixed, marg

    def marginalized(self, blocks, kept, A, b, xbar, values=None):
        """New Problem: the factors touching `blocks` removed, those blocks frozen (they no longer take part), the
        dense prior (kept, A, b, xbar) added; block indices are unchanged."""
        fixed, marg = self.connected_factors(blocks)
        drop = {}
        for t, ci, r in fixed:
            drop.setdefault((t, ci), set()).add(r)
        out = Problem()
        out._values = [(self.values if values is None else np.asarray(values, np.float64)).copy()]
        out._nvalues = self._nvalues
        out.offset, out.size, out.manifold = list(self.offset), list(self.size), list(self.manifold)
        out.is_const = [1 if i in set(int(v) for v in blocks) else c for i, c in enumerate(self.is_const)]
        out.cameras = list(self.cameras)
        for t, chunks in self.factors.items():
            for ci, (idx, consts, lk, la) in enumerate(chunks):
                keep = np.array([r not in drop.get((t, ci), ()) for r in range(idx.shape[0])], bool)
                if keep.any():
                    out.factors.setdefault(t, []).append((idx[keep], consts[keep], lk[keep], la[keep]))
        out.marginals = [m for i, m in enumerate(self.marginals) if i not in marg]
        if A.shape[0]:
            out.add_marginal(kept, A, b, xbar)
        out.meta = dict(self.meta)
        return out

    # -- hand over -------------------------------------------------------------------------
    def load(self, solver):
        """Pushes the whole problem through the C-ABI into `solver` (a capi.Solver)."""
        solver.clear()
        solver.set_blocks(self.values, self.offset, self.size, self.manifold, self.is_const)
        if self.cameras:
            solver.set_cameras(self.cameras)
        for t in sorted(self.factors):
            for idx, consts, lk, la in self.factors[t]:
                solver.add_factors(t, idx, consts, lk, la)
        for blocks, A, b, xbar in self.marginals:
            solver.add_marginal(blocks, A, b, xbar)
        return solver
