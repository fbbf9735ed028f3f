"""Multi-GPU layer of the solve path: independent sliding windows / submap refinements
shard one-per-GPU (one process per GPU, launched by torch.distributed.run); there is NO data-path
collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests) carries only
  * the timing barrier / max-over-ranks / sum of work counters of the benchmark, and
  * the shared-pose consensus named by BASELINE.json's north_star: boundary keyframes that are duplicated in
    neighbouring submaps are averaged (positions arithmetically, orientations by the quaternion average) — a
    latency-bound all-reduce of a few doubles per shared block — and, in consensus_by_marginals, the submaps'
    small marginal priors on those blocks are exchanged so that the rounds converge to the merged optimum.
The reference has no counterpart (it runs its two smoothers as separate ROS processes,
beam_slam_launch/launch/vio.launch:19-30, and refines submaps in a serial loop,
bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115).
"""
import numpy as np


def assign_windows(n_windows, world_size, rank):
    """Static round-robin shard of independent windows/submaps over ranks (weak scaling: BASELINE
    config 5 uses n_windows == world_size, one 200KF x 50k window per GPU)."""
    return list(range(rank, n_windows, world_size))


def window_seed(base_seed, window):
    """BASELINE config 5: seeds +10 .. +17 for the 8 independent C2 instances."""
    return base_seed + 10 + int(window)


def aggregate(dist, work_units, seconds, device="cpu"):
    """Whole-job throughput: sum of work over ranks / max time over ranks."""
    import torch
    w = torch.tensor([float(work_units)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(w.item()), float(t.item())


def per_rank_rates(dist, work_units, seconds, device="cpu"):
    """(number of ranks that reported, every rank's own work / time): one all-reduce of a vector with one slot per rank + a count of
    ones — a run whose ranks did not all arrive shows it in the first number instead of in a plausible-looking aggregate."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    v = torch.zeros(world + 1, dtype=torch.float64, device=device)
    v[rank] = float(work_units) / max(float(seconds), 1e-30)
    v[world] = 1.0
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return int(round(float(v[world].item()))), [round(float(x), 2) for x in v[:world].tolist()]


def _quat_mul(a, b):
    import torch
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def _quat_log(q):
    import torch
    v = q[..., 1:]
    s = v.norm(dim=-1, keepdim=True)
    w = q[..., :1]
    two_theta = 2.0 * torch.where(w < 0, torch.atan2(-s, -w), torch.atan2(s, w))
    k = torch.where(s > 0, two_theta / s.clamp_min(1e-300), torch.full_like(s, 2.0))
    return v * k


def _quat_exp(aa):
    import torch
    th = aa.norm(dim=-1, keepdim=True)
    half = 0.5 * th
    k = torch.where(th > 0, torch.sin(half) / th.clamp_min(1e-300), torch.full_like(th, 0.5))
    return torch.cat([torch.cos(half), aa * k], -1)


def consensus_poses(dist, positions, quaternions, weights=None):
    """Shared-pose consensus over all ranks.

    positions (n,3), quaternions (n,4 wxyz): this rank's estimates of the n shared boundary poses (torch tensors on the
    collective's device).  weights (n,) optional per-rank confidence (0 = this rank does not hold pose i: whatever it passes for
    that pose is ignored).  Returns the consensus (positions, quaternions), identical on every rank: the weighted mean of the
    positions and the weighted quaternion average of the orientations — the principal eigenvector of sum_i w_i q_i q_i^T, which
    needs no reference orientation (no rank's value is singled out, so a placeholder on a rank that does not hold the pose cannot
    leak in) and does not care about the sign of q.  One all-reduce of n x 14 doubles."""
    import torch
    n = positions.shape[0]
    w = torch.ones(n, dtype=positions.dtype, device=positions.device) if weights is None else weights
    q = torch.where((w > 0)[:, None], quaternions, torch.zeros_like(quaternions))     # (a NaN placeholder times 0 would stay NaN)
    p = torch.where((w > 0)[:, None], positions, torch.zeros_like(positions))
    outer = (q[:, :, None] * q[:, None, :]).reshape(n, 16)
    iu = torch.triu_indices(4, 4)
    buf = torch.cat([p * w[:, None], outer[:, iu[0] * 4 + iu[1]] * w[:, None], w[:, None]], dim=1)   # n x (3 + 10 + 1)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    tot = buf[:, 13:14].clamp_min(1e-300)
    pc = buf[:, 0:3] / tot
    M = torch.zeros(n, 4, 4, dtype=buf.dtype, device=buf.device)
    M[:, iu[0], iu[1]] = buf[:, 3:13]
    M = M + M.transpose(1, 2) - torch.diag_embed(torch.diagonal(M, dim1=1, dim2=2))
    evals, evecs = torch.linalg.eigh(M.cpu())
    qc = evecs[:, :, -1].to(buf.device)
    qc = torch.where(qc[:, :1] < 0, -qc, qc)
    return pc, qc / qc.norm(dim=-1, keepdim=True)


# ---------------------------------------------------------------------------------------------------------------------------
# Shared-pose consensus wired to the solver: overlapping windows / submaps, one per GPU, each solved by its own context.  What the
# windows share (the states of their boundary keyframes) is brought to agreement — and to the optimum of the MERGED graph — by
# rounds of
#     1. every rank solves its window: its own factors + the other windows' marginal priors on the shared blocks   (bsgpu_solve)
#     2. consensus: ONE all-reduce averages the ranks' values of the shared blocks (n_shared x 21 doubles: latency-bound; RCCL over
#        xGMI on the GPU box) and every rank moves its copy there
#     3. every rank marginalises its PRIVATE blocks at that point (bsgpu_marginalize: the Schur complement of its own factors
#        onto the shared blocks, a fuse_constraints::MarginalConstraint payload) and the ranks exchange these small dense priors
#        (n_shared_dims^2 doubles each)
# At the fixed point every window minimises  own factors + exact linearised summary of all the others  =  the merged objective,
# so the rounds converge like Gauss-Newton on the merged graph (a handful of rounds), not like a first-order consensus scheme —
# a quadratic-penalty / ADMM variant of this loop needed hundreds of rounds for the bias blocks' 1e6-weighted directions.
# The reference refines its submaps one after the other and exchanges nothing (submap_refinement.cpp:35-115).
# ---------------------------------------------------------------------------------------------------------------------------
def _np_quat_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def _np_quat_log(q):
    v, w = q[..., 1:], q[..., :1]
    s = np.linalg.norm(v, axis=-1, keepdims=True)
    two_theta = 2.0 * np.where(w < 0, np.arctan2(-s, -w), np.arctan2(s, w))
    k = np.where(s > 0, two_theta / np.maximum(s, 1e-300), 2.0)
    return v * k


def _np_quat_exp(aa):
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = np.where(th > 0, np.sin(0.5 * th) / np.maximum(th, 1e-300), 0.5)
    return np.concatenate([np.cos(0.5 * th), aa * k], -1)


def _boxminus(x, z, is_quat):
    """x [-] z per block (rows): tangent 3-vectors."""
    out = np.empty((x.shape[0], 3))
    out[~is_quat] = x[~is_quat, :3] - z[~is_quat, :3]
    if is_quat.any():
        zi = z[is_quat] * np.array([1.0, -1.0, -1.0, -1.0])
        out[is_quat] = _np_quat_log(_np_quat_mul(zi, x[is_quat]))
    return out


def _boxplus(z, d, is_quat):
    out = z.copy()
    out[~is_quat, :3] = z[~is_quat, :3] + d[~is_quat]
    if is_quat.any():
        q = _np_quat_mul(z[is_quat], _np_quat_exp(d[is_quat]))
        out[is_quat] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return out


class SubWindow:
    """One rank's part of a partitioned graph: a local Problem, the global block of every local block, and its shared blocks."""

    def __init__(self, problem, global_of_local, shared_local, shared_id):
        self.problem, self.global_of_local = problem, np.asarray(global_of_local, np.int64)
        self.shared_local, self.shared_id = np.asarray(shared_local, np.int64), np.asarray(shared_id, np.int64)


def partition_problem(pr, part_of_factor, n_parts):
    """Splits Problem `pr` by FACTORS: part_of_factor[(type, chunk)] is an int array (one part per factor row).  A block belongs to
    every part whose factors touch it; blocks touched from more than one part are the shared ones.  Returns ([SubWindow], n_shared,
    is_quat (n_shared,), holders (n_shared, n_parts) bool).  The parts' objectives add up to the merged objective exactly."""
    from . import capi
    from .problem import Problem, NIDX
    cam_types = (capi.F_REPROJ, capi.F_REPROJ_ONLINE_CALIB, capi.F_IDP_REPROJ, capi.F_IDP_REPROJ_UNARY)
    nb = pr.n_blocks
    touched = np.zeros((nb, n_parts), bool)
    for t, chunks in pr.factors.items():
        nvar = NIDX[t] - (1 if t in cam_types else 0)
        for ci, (idx, consts, lk, la) in enumerate(chunks):
            parts = np.asarray(part_of_factor[(t, ci)])
            for p in range(n_parts):
                rows = parts == p
                if rows.any():
                    touched[np.unique(idx[rows, :nvar]), p] = True
    if pr.marginals:
        raise ValueError("partition_problem: dense marginal factors are not split")
    shared_global = np.flatnonzero(touched.sum(axis=1) > 1)
    shared_id_of = -np.ones(nb, np.int64)
    shared_id_of[shared_global] = np.arange(shared_global.size)
    is_quat = np.array([pr.manifold[b] == capi.MANIFOLD_QUAT_RIGHT for b in shared_global], bool)
    subs = []
    for p in range(n_parts):
        glob = np.flatnonzero(touched[:, p])
        local_of = -np.ones(nb, np.int64)
        local_of[glob] = np.arange(glob.size)
        sp = Problem()
        for b in glob:
            lb = sp.add_block(pr.block(int(b)), const=bool(pr.is_const[b]))
            sp.manifold[lb] = pr.manifold[b]
        sp.cameras = list(pr.cameras)
        for t, chunks in pr.factors.items():
            nvar = NIDX[t] - (1 if t in cam_types else 0)
            for ci, (idx, consts, lk, la) in enumerate(chunks):
                rows = np.asarray(part_of_factor[(t, ci)]) == p
                if not rows.any():
                    continue
                li = idx[rows].copy()
                li[:, :nvar] = local_of[li[:, :nvar]]
                sp.add_factors(t, li, consts[rows], lk[rows], la[rows])
        sl = np.flatnonzero(shared_id_of[glob] >= 0)
        subs.append(SubWindow(sp, glob, sl, shared_id_of[glob[sl]]))
    holders = touched[shared_global]
    return subs, int(shared_global.size), is_quat, holders


def _shared_values(sub, n_shared):
    x = np.zeros((n_shared, 4))
    for l, s in zip(sub.shared_local, sub.shared_id):
        v = sub.problem.block(int(l))
        x[s, :v.size] = v
    return x


def _mean_shared(xs, subs, n_shared, is_quat, red):
    """Mean of the holders' values per shared block: arithmetic for vectors, the quaternion average (principal eigenvector of
    sum q q^T) for orientations.  No rank's value is singled out as a reference: a rank that does not hold a block contributes
    nothing."""
    z = np.zeros((n_shared, 4)); w = np.zeros((n_shared, 1)); M = np.zeros((n_shared, 16))
    for x, sub in zip(xs, subs):
        xi = x[sub.shared_id]
        z[sub.shared_id] += xi; w[sub.shared_id] += 1.0
        M[sub.shared_id] += (xi[:, :, None] * xi[:, None, :]).reshape(-1, 16)
    packed = red(np.concatenate([z, w, M], axis=1))        # the consensus all-reduce
    z = packed[:, :4] / np.maximum(packed[:, 4:5], 1.0)
    for s in np.flatnonzero(is_quat):                       # orientations: principal eigenvector of sum q q^T (sign- and reference-free)
        _, vec = np.linalg.eigh(packed[s, 5:].reshape(4, 4))
        q = vec[:, -1]
        z[s] = q if q[0] >= 0 else -q
    return z


def consensus_by_marginals(solvers, subs, n_shared, is_quat, rounds, options=None, all_reduce=None, part_ids=None, n_parts=None,
                           tol=1e-10, on_round=None):
    """The loop above over the sub-windows THIS process holds: `solvers[i]` solves `subs[i]`, whose global part number is
    part_ids[i] (default: 0..len-1 with n_parts = len — one process holding every part, as in the single-GPU test; under
    torch.distributed every rank passes its one window and all_reduce(array) -> array, a float64 SUM over all ranks).
    Returns (z (n_shared, 4): the consensus values of the shared blocks, history [(round, |dz|, cost of this process' windows)]);
    the sub-problems' values are left at the last solution with the shared blocks AT the consensus."""
    red = (lambda a: a) if all_reduce is None else all_reduce
    part_ids = list(range(len(subs))) if part_ids is None else list(part_ids)
    n_parts = len(subs) if n_parts is None else int(n_parts)
    # LIMITATION: every part marginalises its own factors onto ALL of its shared blocks and a receiver can only apply
    # a prior whose blocks it holds — exact when every window holds every shared block (two windows; a star around one key frame),
    # NOT for a chain A - B - C, where A would never hear of C.  Chains and trees of windows: MessagePassing below.
    for sub in subs:
        if len(sub.shared_id) != n_shared:
            raise NotImplementedError("consensus_by_marginals: a window does not hold every shared block (a chain / tree of windows): "
                                      "use sharding.MessagePassing + message_passing_rounds")
        if any(sub.problem.size[int(l)] not in (3, 4) for l in sub.shared_local):
            raise ValueError("consensus_by_marginals: shared blocks must be 3-vectors or unit quaternions")
    dims = np.where(is_quat, 3, 3)                          # tangent width of every shared block (checked above)
    col0 = np.concatenate([[0], np.cumsum(dims)])
    nd = int(col0[-1])
    amb = np.where(is_quat, 4, 3)
    # marginal priors of ALL parts, in a fixed layout every rank fills its own slots of (the exchange is an all-reduce SUM of
    # one-hot slots: an all-gather): per part  A (nd x nd, upper-triangular factor rows, zero-padded), b (nd), xbar (n_shared x 4),
    # mask (n_shared): which shared blocks the prior covers
    slot = nd * nd + nd + 4 * n_shared + n_shared
    priors = np.zeros((n_parts, slot))
    z = None
    history = []
    for rnd in range(rounds):
        xs, cost = [], 0.0
        for g, sub, pid in zip(solvers, subs, part_ids):
            sub.problem.load(g)
            local_of_shared = {int(s): int(l) for l, s in zip(sub.shared_local, sub.shared_id)}
            for other in range(n_parts):
                if other == pid:
                    continue
                A = priors[other, :nd * nd].reshape(nd, nd)
                b = priors[other, nd * nd:nd * nd + nd]
                xbar = priors[other, nd * nd + nd:nd * nd + nd + 4 * n_shared].reshape(n_shared, 4)
                mask = priors[other, nd * nd + nd + 4 * n_shared:] > 0.5
                ids = [int(s) for s in np.flatnonzero(mask)]
                if not ids or any(s not in local_of_shared for s in ids):
                    continue                                  # nothing received yet / a prior over blocks this window does not hold
                cols = np.concatenate([np.arange(col0[s], col0[s + 1]) for s in ids])
                rows = np.flatnonzero(np.abs(A[:, cols]).sum(axis=1) + np.abs(b) > 0)
                if rows.size == 0:
                    continue
                g.add_marginal([local_of_shared[s] for s in ids], A[np.ix_(rows, cols)], b[rows],
                               np.concatenate([xbar[s, :amb[s]] for s in ids]))
            summ = g.solve(options)
            cost += summ.final_cost
            sub.problem.values = g.get_blocks()
            xs.append(_shared_values(sub, n_shared))
        z_new = _mean_shared(xs, subs, n_shared, is_quat, red)
        dz = float(np.abs(_boxminus(z_new, z, is_quat)).max()) if z is not None else float("inf")
        z = z_new
        history.append((rnd, dz, cost))
        if on_round is not None:
            on_round(rnd, z, dz)
        # every window moves its copy of the shared blocks to the consensus and summarises its own factors there
        mine = np.zeros((n_parts, slot))
        for g, sub, pid in zip(solvers, subs, part_ids):
            v = sub.problem.values.copy()
            for l, s in zip(sub.shared_local, sub.shared_id):
                o = sub.problem.offset[int(l)]
                v[o:o + amb[s]] = z[s, :amb[s]]
            sub.problem.values = v
            if dz < tol or rnd == rounds - 1:
                continue
            sub.problem.load(g)
            shared_set = set(int(l) for l in sub.shared_local)
            private = [b for b in range(sub.problem.n_blocks) if b not in shared_set and not sub.problem.is_const[b]]
            kept, A, b, xbar = g.marginalize(private, sub.problem.size)
            sid_of_local = {int(l): int(s) for l, s in zip(sub.shared_local, sub.shared_id)}
            Afull = np.zeros((nd, nd)); bfull = np.zeros(nd); xb = np.zeros((n_shared, 4)); mask = np.zeros(n_shared)
            c = 0; xo = 0
            nrow = A.shape[0]
            assert nrow <= nd
            for kb in kept:
                sid = sid_of_local[int(kb)]
                Afull[:nrow, col0[sid]:col0[sid + 1]] = A[:, c:c + dims[sid]]
                xb[sid, :amb[sid]] = xbar[xo:xo + amb[sid]]
                mask[sid] = 1.0
                c += dims[sid]; xo += amb[sid]
            bfull[:nrow] = b
            mine[pid] = np.concatenate([Afull.ravel(), bfull, xb.ravel(), mask])
        if dz < tol:
            break
        priors = red(mine)                                    # the exchange of the marginal priors
    return z, history


# ---------------------------------------------------------------------------------------------------------------------------
# Consensus by MESSAGES (round 3): the same goal — overlapping windows / submaps, one per GPU, converge to the optimum of the
# merged graph exchanging nothing but small summaries of their shared key frames — for any TREE of windows (a chain of submaps:
# A - B - C ...), at window scale, with everything but the messages staying on the device:
#     1. every rank solves its window: own factors + one dense prior per neighbour (the neighbour's message)           bsgpu_solve,
#        warm-started; the context is finalized ONCE, the priors' contents are replaced in place                  bsgpu_update_marginal
#     2. its BELIEF on the key frames it shares: their joint marginal covariance at the solution (one undamped
#        factorisation, unit vectors in the rhs tile)                                                            bsgpu_covariance_joint
#     3. the message to neighbour q = belief on what is shared with q, divided by q's own message (information form:
#        Lambda_out = Lambda_belief - Lambda_in) — i.e. the marginal of own factors + EVERY OTHER neighbour's message, which is what
#        makes a chain A - B - C exact: B's message to A carries what C told B
#     4. the messages travel in ONE all-reduce (one slot per directed pair of neighbours; RCCL over xGMI on the GPU box)
# On a tree this is Gaussian belief propagation with re-linearisation: exact for linear-Gaussian graphs after diameter-many rounds,
# Gauss-Newton-like on ours.  consensus_by_marginals above averages the shared values and lets every part marginalise its own
# factors only: exact for two parts, but a part that does not hold a prior's blocks skips it, so in a chain the ends never hear of
# each other.  A block shared by MORE than two windows is not a tree edge: refused.
# The reference refines its submaps one after the other and exchanges nothing (submap_refinement.cpp:35-115).
# ---------------------------------------------------------------------------------------------------------------------------
class MessagePassing:
    """One window of the tree (this rank's).  `problem`: its local Problem; `neighbours`: {part id: [local blocks shared with it]}, the
    blocks in the SAME order on both sides; `solver`: a capi.Solver (libbsgpu context, or the test oracle)."""

    WEAK = 1e-3          # sqrt-information of the placeholder priors before the first messages arrive

    def __init__(self, solver, problem, part_id, neighbours, options=None):
        from . import capi
        self.g, self.pr, self.pid, self.opt = solver, problem, int(part_id), options
        self.nbr = sorted(int(q) for q in neighbours)
        self.blocks = {q: [int(b) for b in neighbours[q]] for q in self.nbr}
        self.is_quat = {q: np.array([problem.manifold[b] == capi.MANIFOLD_QUAT_RIGHT for b in self.blocks[q]]) for q in self.nbr}
        self.amb = {q: np.array([problem.size[b] for b in self.blocks[q]]) for q in self.nbr}
        self.D = {q: 3 * len(self.blocks[q]) for q in self.nbr}
        for q in self.nbr:
            if any(problem.size[b] not in (3, 4) for b in self.blocks[q]):
                raise ValueError("MessagePassing: shared blocks must be 3-vectors or unit quaternions")
        seen = [b for q in self.nbr for b in self.blocks[q]]
        if len(seen) != len(set(seen)):
            raise NotImplementedError("MessagePassing: a block shared with more than one neighbour is not an edge of a tree of windows")
        # incoming messages (xbar ambient rows, Lambda, m): placeholders, weak and centred at the initial values
        self.msg_in = {}
        for q in self.nbr:
            xbar = self._values_of(q, problem.values)
            self.msg_in[q] = (xbar, (self.WEAK ** 2) * np.eye(self.D[q]), np.zeros(self.D[q]))
        self._resident = solver.has("update_marginal") and solver.has("covariance_joint")
        self._loaded = False
        self.last_summary = None
        self._x0 = problem.values.copy()

    def reset(self):
        """Back to the initial values and the placeholder messages (the device tables stay: a benchmark repeats the rounds)."""
        self.pr.values = self._x0.copy()
        for q in self.nbr:
            self.msg_in[q] = (self._values_of(q, self._x0), (self.WEAK ** 2) * np.eye(self.D[q]), np.zeros(self.D[q]))
        if self._loaded:
            self.g.set_values(self._x0)

    # -- helpers -------------------------------------------------------------------------------------------------------------
    def _values_of(self, q, values):
        x = np.zeros((len(self.blocks[q]), 4))
        for i, b in enumerate(self.blocks[q]):
            v = self.pr.block(b, values)
            x[i, :v.size] = v
        return x

    @staticmethod
    def _factor(Lam, m):
        """(A, b) of the prior r = b + A d with A^T A = Lambda (projected onto the positive semi-definite cone), b = -A m."""
        w, V = np.linalg.eigh(0.5 * (Lam + Lam.T))
        A = (np.sqrt(np.maximum(w, 0.0))[:, None]) * V.T
        return A, -A @ m

    def _payload(self, q):
        xbar, Lam, m = self.msg_in[q]
        A, b = self._factor(Lam, m)
        return A, b, np.concatenate([xbar[i, :self.amb[q][i]] for i in range(xbar.shape[0])])

    def _load(self):
        self.pr.marginals = []
        for q in self.nbr:
            A, b, xb = self._payload(q)
            self.pr.add_marginal(self.blocks[q], A, b, xb)
        self.pr.load(self.g)
        self._loaded = True

    def _push_messages(self):
        if self._resident and self._loaded:
            for i, q in enumerate(self.nbr):
                self.g.update_marginal(i, *self._payload(q))
        else:
            self._load()

    def _joint_covariance(self):
        blocks = [b for q in self.nbr for b in self.blocks[q]]
        if self.g.has("covariance_joint"):
            return self.g.covariance_joint(blocks, [3] * len(blocks))
        n = 3 * len(blocks)                                   # (the test oracle: pairwise blocks)
        S = np.zeros((n, n))
        for i, bi in enumerate(blocks):
            for j, bj in enumerate(blocks):
                if j >= i:
                    S[3 * i:3 * i + 3, 3 * j:3 * j + 3] = self.g.covariance(bi, bj)
                    S[3 * j:3 * j + 3, 3 * i:3 * i + 3] = S[3 * i:3 * i + 3, 3 * j:3 * j + 3].T
        return S

    # -- one round, this rank's half ---------------------------------------------------------------------------------------------
    def solve_and_summarise(self):
        """Steps 1-3.  Returns {q: (xbar, Lambda_out, m_out)}: the messages to the neighbours."""
        self._push_messages()
        self.last_summary = self.g.solve(self.opt)
        values = self.g.get_blocks()
        self.pr.values = values
        out = {}
        if not self.nbr:
            return out
        S = self._joint_covariance()
        c = 0
        for q in self.nbr:
            D = self.D[q]
            Lam_b = np.linalg.inv(S[c:c + D, c:c + D])
            c += D
            x = self._values_of(q, values)
            xbar_in, Lam_in, m_in = self.msg_in[q]
            shift = _boxminus(x, xbar_in, self.is_quat[q]).ravel()    # the incoming message, re-centred at this window's solution
            m_rel = m_in - shift
            Lam_out = Lam_b - Lam_in
            w, V = np.linalg.eigh(0.5 * (Lam_out + Lam_out.T))
            w = np.maximum(w, 0.0)
            Lam_out = (V * w) @ V.T
            winv = np.where(w > 1e-12 * max(w.max(), 1e-300), 1.0 / np.maximum(w, 1e-300), 0.0)
            m_out = -((V * winv) @ V.T) @ (Lam_in @ m_rel)
            out[q] = (x, Lam_out, m_out)
        return out

    def own_cost(self):
        """Cost of this window's OWN factors at its current values (the solver's cost minus the priors' energies)."""
        values = self.pr.values
        e = 0.0
        for q in self.nbr:
            xbar, Lam, m = self.msg_in[q]
            A, b = self._factor(Lam, m)
            d = _boxminus(self._values_of(q, values), xbar, self.is_quat[q]).ravel()
            r = b + A @ d
            e += 0.5 * float(r @ r)
        return self.last_summary.final_cost - e

    def shared_values(self, q):
        return self._values_of(q, self.pr.values)


def message_passing_rounds(windows, rounds, all_reduce=None, n_parts=None, tol=1e-9, on_round=None):
    """Runs the rounds for the windows THIS process holds (`windows`: list of MessagePassing; one per rank under torch.distributed,
    all of them in a single-process test).  all_reduce(array) -> array: float64 SUM over all ranks (None: single process).
    Returns history [(round, max |change of a shared value| over all windows, sum of own costs of this process' windows)]."""
    red = (lambda a: a) if all_reduce is None else all_reduce
    n_parts = (max(w.pid for w in windows) + 1) if n_parts is None else int(n_parts)
    Dmax = 15
    for w in windows:
        for q in w.nbr:
            Dmax = max(Dmax, w.D[q])
    nblk = Dmax // 3
    slot = 4 * nblk + Dmax * Dmax + Dmax + 1                 # xbar | Lambda | m | present
    history = []
    prev = {}
    for rnd in range(rounds):
        buf = np.zeros((n_parts, n_parts, slot))             # [sender, receiver]
        stat = np.zeros(2 * n_parts)                          # [own cost per part | max shared change per part]
        for w in windows:
            msgs = w.solve_and_summarise()
            for q, (x, Lam, m) in msgs.items():
                D, nb = w.D[q], len(w.blocks[q])
                s = np.zeros(slot)
                s[:4 * nb] = x.ravel()
                L = np.zeros((Dmax, Dmax)); L[:D, :D] = Lam
                s[4 * nblk:4 * nblk + Dmax * Dmax] = L.ravel()
                s[4 * nblk + Dmax * Dmax:4 * nblk + Dmax * Dmax + D] = m
                s[-1] = 1.0
                buf[w.pid, q] = s
                key = (w.pid, q)
                if key in prev:
                    stat[n_parts + w.pid] = max(stat[n_parts + w.pid], float(np.abs(_boxminus(x, prev[key], w.is_quat[q])).max()))
                else:
                    stat[n_parts + w.pid] = np.inf
                prev[key] = x
            stat[w.pid] = w.own_cost()
        buf = red(buf.ravel()).reshape(n_parts, n_parts, slot)  # the exchange: every rank filled its own rows only
        stat_max = stat.copy(); stat_max[:n_parts] = 0.0
        cost = red(np.concatenate([stat[:n_parts], np.zeros(n_parts)]))[:n_parts].sum()
        # (the max over ranks as a sum of one-hot entries: every part's change sits in its own slot)
        fin = np.where(np.isfinite(stat[n_parts:]), stat[n_parts:], 1e300)
        change = red(np.concatenate([np.zeros(n_parts), fin]))[n_parts:]
        dz = float(change.max()) if change.size else 0.0
        dz = float("inf") if dz >= 1e299 else dz
        for w in windows:
            for q in w.nbr:
                s = buf[q, w.pid]
                if s[-1] < 0.5:
                    continue
                D, nb = w.D[q], len(w.blocks[q])
                x = s[:4 * nb].reshape(nb, 4)
                Lam = s[4 * nblk:4 * nblk + Dmax * Dmax].reshape(Dmax, Dmax)[:D, :D]
                m = s[4 * nblk + Dmax * Dmax:4 * nblk + Dmax * Dmax + D]
                w.msg_in[q] = (x.copy(), Lam.copy(), m.copy())
        history.append((rnd, dz, float(cost)))
        if on_round is not None:
            on_round(rnd, dz, float(cost))
        if dz < tol:
            break
    return history
