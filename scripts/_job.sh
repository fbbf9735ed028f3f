python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\[bsgpu\]" | tail -25 > gpurun_out/r6_tests_full.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r6b.json 2> gpurun_out/bench_r6b.err
bash scripts/profile_all.sh r06 > gpurun_out/profile_all_r06.log 2>&1
bash scripts/pmc_calib.sh gpurun_out/prof_r06/r06_pmc_calib.csv > gpurun_out/pmc_calib.log 2>&1
tail -5 gpurun_out/r6_tests_full.log
