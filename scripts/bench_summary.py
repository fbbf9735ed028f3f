"""One screen of a default bench line:  python bench.py > b.json; python scripts/bench_summary.py b.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C2", d["value"], "LM it/s,", d["ms_per_step"], "ms per solve; roofline frac", d["roofline"]["frac"], "sustained", (d.get("sustained") or {}).get("value"))
for k, v in (d.get("other_configs") or {}).items():
    one = v.get("one_window_alone")
    print(" ", k, v.get("value"), v.get("host_cycle_ms"), one.get("value") if isinstance(one, dict) else one)
cb = d.get("cpu_baseline") or {}
print("cpu", cb.get("value"), (cb.get("reference_thread_count") or {}).get("value"))
