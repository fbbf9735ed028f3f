"""world_size-2 CPU (gloo) coverage of the N>1 layer: window sharding, whole-job aggregation and the
shared-pose consensus all-reduce (beam_slam_amd/sharding.py).  The GPU box runs the same code over
RCCL (backend "nccl")."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from beam_slam_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    wins = sharding.assign_windows(5, world, rank)
    seeds = [sharding.window_seed(20250620, w) for w in wins]
    work, t = sharding.aggregate(dist, 10 * len(wins), 0.5 + 0.25 * rank)
    rng = np.random.default_rng(1)
    p_true = rng.normal(size=(4, 3)); aa = rng.normal(size=(4, 3)) * 0.3
    q_true = sharding._quat_exp(torch.tensor(aa))
    prng = np.random.default_rng(100 + rank)
    p = torch.tensor(p_true + 0.01 * prng.normal(size=(4, 3)))
    q = sharding._quat_mul(q_true, sharding._quat_exp(torch.tensor(0.01 * prng.normal(size=(4, 3)))))
    w = torch.ones(4, dtype=torch.float64)
    if rank == 1: w[3] = 0.0          # rank 1 does not hold pose 3 ...
    if rank == 1: q[3] = float("nan"); p[3] = float("nan")     # ... and passes a placeholder there
    if rank == 0: w[2] = 0.0; q[2] = 0.0                       # rank 0 (no special role) does not hold pose 2: a zero quaternion
    pc, qc = sharding.consensus_poses(dist, p, q, w)
    # one file per rank: both ranks share the launcher's stdout pipe and long lines written to it can interleave
    with open(os.path.join(sys.argv[1], "result_%%d.json" %% rank), "w") as fh:
        json.dump(dict(rank=rank, wins=wins, seeds=seeds, work=work, t=t, pc=pc.tolist(), qc=qc.tolist(), p=p.tolist(), q=q.tolist()), fh)
    dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_aggregation_and_consensus(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = [json.loads((tmp_path / ("result_%d.json" % r)).read_text()) for r in range(2)]
    # sharding: disjoint cover of the 5 windows, BASELINE config-5 seeds
    assert res[0]["wins"] == [0, 2, 4] and res[1]["wins"] == [1, 3]
    assert res[0]["seeds"] == [20250630, 20250632, 20250634]
    # aggregation: sum of work, max of time, same on both ranks
    for r in res:
        assert r["work"] == 50.0 and r["t"] == 0.75
    # consensus identical on both ranks and equal to the numpy reference
    pc0, pc1 = np.array(res[0]["pc"]), np.array(res[1]["pc"])
    qc0, qc1 = np.array(res[0]["qc"]), np.array(res[1]["qc"])
    assert np.array_equal(pc0, pc1) and np.array_equal(qc0, qc1)
    p0, p1 = np.array(res[0]["p"]), np.array(res[1]["p"])
    exp_p = np.vstack([(p0[:2] + p1[:2]) / 2, p1[2:3], p0[3:4]])
    assert np.abs(pc0 - exp_p).max() < 1e-14
    q0, q1 = np.array(res[0]["q"]), np.array(res[1]["q"])
    same = lambda a, b: min(np.abs(a - b).max(), np.abs(a + b).max())
    assert same(qc0[3], q0[3]) < 1e-14                    # pose 3 only known to rank 0 (rank 1 passed NaN)
    assert same(qc0[2], q1[2]) < 1e-14                    # pose 2 only known to rank 1 (rank 0 passed a zero quaternion)
    assert np.isfinite(qc0).all() and np.isfinite(pc0).all()
    assert np.abs(np.linalg.norm(qc0, axis=1) - 1).max() < 1e-14
    # the consensus orientation lies between the two estimates
    for i in range(2):
        d01 = 1 - abs(q0[i] @ q1[i]); d0c = 1 - abs(q0[i] @ qc0[i]); d1c = 1 - abs(q1[i] @ qc0[i])
        assert d0c < d01 and d1c < d01
