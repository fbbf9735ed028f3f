python -m pytest tests/test_gpu_band.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6_tests_h.log
bash scripts/kstats.sh c2 9 > gpurun_out/kstats_lz.log 2>&1
bash scripts/pmc_kernel.sh pairs_band "FETCH_SIZE" >> gpurun_out/kstats_lz.log 2>&1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'])" >> gpurun_out/kstats_lz.log; done
cat gpurun_out/r6_tests_h.log gpurun_out/kstats_lz.log
