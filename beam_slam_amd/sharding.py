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
    dims = np.where(is_quat, 3, 3)                          # tangent width of every shared block
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
