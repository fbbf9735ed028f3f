"""The fused Cholesky of the reduced camera system (beam_slam_amd/csrc/dense_plan.h, chol_chain.h, k_chol.hip).

CPU: the planner's ticket list is replayed in order on random tile structures (tests/plan/test_plan.cpp): every counter a
task waits for has been advanced by EARLIER tasks (the dead-lock-freedom argument of chol_fused_kernel), every tile is
factored exactly once, the tile-level replay reproduces a dense Cholesky.
GPU (-m gpu): the in-workgroup chain factorisation alone (scripts/chain_probe.hip: one workgroup, chains of 1-3 tiles, a
partial last tile) against a host Cholesky — factor, tile inverses, flags."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ticket_list_replay(tmp_path):
    exe = str(tmp_path / "test_plan")
    out = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "beam_slam_amd", "csrc"),
                          os.path.join(ROOT, "tests", "plan", "test_plan.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-4000:]
    assert "400 cases, 0 failures" in run.stdout
    assert "400 ordered cases, 0 failures" in run.stdout   # plans on the per-dimension order (dim_order.h)


@pytest.mark.gpu
def test_chain_factorisation_in_one_workgroup(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "chain_probe")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value",
                          os.path.join(ROOT, "scripts", "chain_probe.hip"), "-o", exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "all ok" in run.stdout, run.stdout[-4000:]
