"""bench.py's launcher logic, without a GPU: `--gpus N` must never silently run a different job than the one asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BSGPU_BENCH_SAME_DEVICE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_more_gpus_than_devices_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    if n == 1:
        n = 2
    out = _run({}, "--gpus", str(n), "--steps", "1", "--warmup", "0")
    assert out.returncode != 0
    assert "GPU(s)" in out.stderr and "--gpus %d" % n in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]      # no line at all rather than a line about another job


def test_world_size_must_equal_gpus():
    out = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "4", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
