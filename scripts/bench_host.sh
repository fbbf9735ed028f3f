#!/bin/bash
# builds and runs tests/host/bench_host.cpp against libbsgpu (GPU box)
set -e
cd "$(dirname "$0")/.."
g++ -O2 -std=c++17 tests/host/bench_host.cpp -o /tmp/bench_host -Lbeam_slam_amd/csrc -lbsgpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/beam_slam_amd/csrc -Wl,-rpath,/opt/rocm/lib
/tmp/bench_host "$@"
