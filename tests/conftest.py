import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_cls():
    from oracle import Oracle  # oracle/oracle.py — the CPU checker (test infrastructure)
    return Oracle


@pytest.fixture(scope="session")
def gpu_solver_cls():
    from beam_slam_amd.gpu import GpuSolver
    return GpuSolver
