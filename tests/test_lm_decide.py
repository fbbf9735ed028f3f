"""The trust-region decision as the device takes it (beam_slam_amd/csrc/lm_decide.h) against LmState::advance (lm_state.h), on the CPU:
tests/plan/test_lm_decide.cpp — two million random steps (accepted, rejected, invalid, every tolerance), the same decision and the same
radius bit for bit; the shared cube against libm's pow(t, 3)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_decision_is_lmstate_advance(tmp_path):
    exe = str(tmp_path / "test_lm_decide")
    out = subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "beam_slam_amd", "csrc"),
                          os.path.join(ROOT, "tests", "plan", "test_lm_decide.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "all ok" in run.stdout, run.stdout[-4000:]
    assert ", 0 mismatches" in run.stdout
