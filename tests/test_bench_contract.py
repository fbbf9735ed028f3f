"""bench.py's own contract, executed on the GPU box: the single-process line, and the torch.distributed (RCCL) path that the
driver uses for N > 1 — forced at world size 1 (BSGPU_BENCH_FORCE_DIST=1: init_process_group("nccl"), barriers, the max-over-ranks
all-reduce), so that the multi-GPU code is at least executed where one GPU exists."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    # (--no-other-configs: the C3 / C4 / batch / host-cycle legs of the default run take half a minute and are the driver's bench run, not this contract)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--sustained-seconds", "0.3", *args],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_line_small_window():
    r = _bench({}, "--n-kf", "40", "--n-lm", "4000")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["value"] > 0 and r["dtype"] == "f64"
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1.5


def test_bench_rccl_path_world_size_one():
    env = {"BSGPU_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
           "MASTER_PORT": "29533", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = _bench(env, "--n-kf", "40", "--n-lm", "4000")
    assert r["n_gpus"] == 1 and r["value"] > 0
    single = _bench({}, "--n-kf", "40", "--n-lm", "4000")
    assert abs(r["config"]["final_cost"] - single["config"]["final_cost"]) <= 1e-9 * single["config"]["final_cost"]


def test_bench_consensus_two_ranks_on_one_gpu():
    """bench.py --consensus as the driver launches it for N = 2 (torch.distributed.run, one process per window), here with both
    ranks on the one GPU of the box (gloo carries the messages: RCCL refuses two ranks on one device): the windows are consecutive
    submaps sharing a key frame, the line reports the rounds, the time of each and the merged graph's cost."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, BSGPU_BENCH_BACKEND="gloo", BSGPU_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--consensus", "--n-kf", "40", "--n-lm", "4000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["value"] > 0 and "consensus" in r
    c = r["consensus"]
    assert 2 <= c["rounds"] <= 8 and len(c["ms_per_round"]) == c["rounds"] and c["last_change_of_a_shared_value"] < 1e-7
    assert c["merged_graph_cost"] > 0


def test_bench_windows_per_gpu_two_ranks_on_one_gpu():
    """bench.py --gpus 2 --windows-per-gpu 3 as the driver launches it (torch.distributed.run), both ranks on the one GPU of the box (gloo: RCCL
    refuses two ranks on one device): every rank advances its three windows with one bsgpu_solve_batch per step; the line says how many ranks
    reported and what each did, and the aggregate is their sum over the slowest rank's time."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, BSGPU_BENCH_BACKEND="gloo", BSGPU_BENCH_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows-per-gpu", "3", "--n-kf", "30", "--n-lm", "2000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
    c = r["config"]
    assert c["n_ranks_seen"] == 2 and len(c["per_rank_lm_iterations_per_s"]) == 2 and min(c["per_rank_lm_iterations_per_s"]) > 0
    assert "3 per GPU" in c["workload"] and "6 independent" in c["workload"]
    # the same flag on one rank
    r1 = _bench({}, "--windows-per-gpu", "3", "--n-kf", "30", "--n-lm", "2000")
    assert r1["n_gpus"] == 1 and r1["config"]["n_ranks_seen"] == 1 and r1["value"] > 0


def test_bench_plain_gpus_two_launches_two_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the documented usage): bench.py starts its two ranks itself under
    torch.distributed.run and the line says n_gpus 2 with two ranks reporting.  Both ranks share the one GPU of the box (gloo carries the
    barriers: RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BSGPU_BENCH_BACKEND="gloo", BSGPU_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-other-configs", "--sustained-seconds", "0", "--n-kf", "30", "--n-lm", "2000"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # ONE JSON line for the whole job
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["n_ranks_seen"] == 2 and len(r["config"]["per_rank_lm_iterations_per_s"]) == 2
    assert "2 independent" in r["config"]["workload"] and r["value"] > 0
