python -m pytest tests/test_gpu_band.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_env_paths.py tests/test_gpu_assembly_ahead.py tests/test_marginalization.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r6_tests_i.log
for v in 0 1 0 1; do echo "== BSGPU_BAND_LOWER=$v" >> gpurun_out/ab_lower.log; BSGPU_BAND_LOWER=$v bash scripts/kstats.sh c2 6 2>&1 | grep -i "pairs_band\|chol_fused_kernel<\|value" >> gpurun_out/ab_lower.log
BSGPU_BAND_LOWER=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'cost %.12e' % d['config']['final_cost'])" >> gpurun_out/ab_lower.log; done
cat gpurun_out/r6_tests_i.log gpurun_out/ab_lower.log
