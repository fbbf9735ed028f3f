"""Multi-GPU layer of the solve path: independent sliding windows / submap refinements
shard one-per-GPU (one process per GPU, launched by torch.distributed.run); there is NO data-path
collective.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests) carries only
  * the timing barrier / max-over-ranks / sum of work counters of the benchmark, and
  * the optional shared-pose consensus named by BASELINE.json's north_star: boundary keyframes that
    are duplicated in neighbouring submaps are averaged (positions arithmetically, orientations in the
    tangent space of a common reference) — a latency-bound all-reduce of n_shared x 6 doubles.
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

    positions (n,3), quaternions (n,4 wxyz): this rank's estimates of the n shared boundary poses
    (torch tensors on the collective's device).  weights (n,) optional per-rank confidence (0 = this
    rank does not hold pose i).  Returns the consensus (positions, quaternions), identical on every
    rank: weighted mean of positions; orientations averaged in the tangent space of rank 0's estimate
    (right perturbation, like the solver's manifold)."""
    import torch
    n = positions.shape[0]
    w = torch.ones(n, dtype=positions.dtype, device=positions.device) if weights is None else weights
    ref = quaternions.clone()
    dist.broadcast(ref, src=0)
    ref_inv = ref * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=ref.dtype, device=ref.device)
    tang = _quat_log(_quat_mul(ref_inv, quaternions))
    buf = torch.cat([positions * w[:, None], tang * w[:, None], w[:, None]], dim=1)   # n x 7, one all-reduce
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    tot = buf[:, 6:7].clamp_min(1e-300)
    p = buf[:, 0:3] / tot
    q = _quat_mul(ref, _quat_exp(buf[:, 3:6] / tot))
    return p, q / q.norm(dim=-1, keepdim=True)
