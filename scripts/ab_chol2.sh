B="python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'))" "$1"; }
for beta in 0 0.25 0.5 1; do BSGPU_TICKET_BETA=$beta $B 2>/dev/null | ex beta=$beta; done
$B --workload c3 2>/dev/null | ex c3
BSGPU_CHOL_PROBE=gpurun_out/chol_probe_dim2.txt python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --sustained-seconds 0 --no-past-l3 > /dev/null 2>&1; python scripts/chol_probe.py gpurun_out/chol_probe_dim2.txt > gpurun_out/chol_probe_dim2_path.txt 2>&1
python scripts/small_window.py 2>&1 | tail -4
