"""C++ host side (beam_slam_amd/host/: fuse_core / bs_constraints / bs_optimizers mirrors) — the tests
are a C++ program written like the reference's gtests (tests/host/test_host.cpp).

CPU run: the HOST LOGIC (deterministic block order, pack(), transactions, lag window,
pseudo-marginalisation, KATs through GpuGraph::optimize) is exercised with the header compiled against
the test oracle (tests/host/oracle_backend.h force-included), because no GPU exists here.
GPU run (-m gpu): the same program linked against the product library libbsgpu.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_host.cpp")


def _build(tmp_path, extra):
    exe = str(tmp_path / "test_host")
    cmd = ["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wno-unused-function", SRC, "-o", exe] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    return exe


def _run(exe):
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "ALL HOST TESTS PASSED" in out.stdout
    return out.stdout


def test_host_logic_against_oracle_backend(tmp_path):
    from oracle import build
    build()
    odir = os.path.join(ROOT, "oracle")
    exe = _build(tmp_path, ["-include", os.path.join(ROOT, "tests", "host", "oracle_backend.h"), "-L" + odir, "-lbs_oracle", "-Wl,-rpath," + odir])
    out = _run(exe)
    assert "FixedLagSmootherWindow" in out


@pytest.mark.gpu
def test_host_on_gpu_through_libbsgpu(tmp_path):
    cdir = os.path.join(ROOT, "beam_slam_amd", "csrc")
    exe = _build(tmp_path, ["-L" + cdir, "-lbsgpu", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + cdir, "-Wl,-rpath,/opt/rocm/lib"])
    _run(exe)


def test_cholesky_plan_executed_on_the_host(tmp_path):
    """beam_slam_amd/csrc/dense_plan.h (tile order, symbolic fill, concurrent-panel schedule with shared tiles, look-ahead,
    back-substitution plan) executed with the device kernels' semantics by tests/host/test_plan.cpp."""
    exe = str(tmp_path / "test_plan")
    out = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(ROOT, "tests", "host", "test_plan.cpp"), "-o", exe],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ALL PLAN TESTS PASSED" in run.stdout, run.stdout[-4000:]
