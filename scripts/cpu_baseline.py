"""CPU baseline of BASELINE.md §3: the oracle (own restatement, test infrastructure) on C2 with the reference's shipped
thread count (vio.yaml:11: 6) and with all usable cores.  Prints one line per setting."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from beam_slam_amd import synthetic
from oracle import Oracle, usable_cpus

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
pr = getattr(synthetic, which)()
try:
    model = [l.split(":", 1)[1].strip() for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines() if l.startswith("Model name")][0]
except Exception:
    model = "?"
print("host CPU:", model, "| usable cores:", usable_cpus())
for threads in (6, usable_cpus()):
    o = Oracle(threads=threads)
    pr.load(o)
    opt = o.options_vio()
    opt.max_solver_time_in_seconds = 1e9
    o.finalize()
    t0 = time.perf_counter()
    s = o.solve(opt)
    dt = time.perf_counter() - t0
    print(f"{which} oracle threads={threads}: {s.num_linear_solves / dt:.2f} LM it/s, {1e3 * dt:.0f} ms / solve ({s.num_linear_solves} it), cost {s.initial_cost:.6e} -> {s.final_cost:.10e}", flush=True)
