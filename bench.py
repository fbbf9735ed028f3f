#!/usr/bin/env python
"""bench.py — LM iterations/s and ms per graph solve on the 200-keyframe x 50k-landmark VIO window.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full graph solve (bsgpu_solve) of BASELINE.json config 2 — the 200 KF x 50 k
landmark synthetic visual-inertial window of SURVEY.md §8d — from its device-resident initial guess
with the solver options the reference ships (beam_slam_launch/config/vio.yaml:7-17: <= 10 LM
iterations, tolerances 1.5e-7; the 0.05 s wall-clock clip is lifted so every step does the same
work).  Inputs are resident in HBM before the timed region (bsgpu_finalize + bsgpu_reset_values
are device-side).  value = LM iterations (trust-region steps computed) of all ranks / wall time.

N > 1: one process per GPU (launched by torch.distributed.run); every rank solves its own
independent window (BASELINE config 5: seeds +10 + rank) — the path shards by window, there is no
data-path collective; torch.distributed (RCCL) is used for the barriers and the max-over-ranks.

The JSON line also carries
  roofline     — the reprojection Jacobian-evaluation kernel: algorithmic bytes / HIP-event time measured in situ (inside real LM
                 steps), plus the same kernel on a working set above the 256 MiB Infinity Cache (past_l3)
  roofline_mfma, kernels, phases_us_per_lm_step — the Cholesky of the reduced camera system against the FP64 MFMA peak, the other
                 HBM-bound kernels of a step, and where an LM step's time goes (bsgpu_profile_step)
  cpu_baseline — the CPU oracle (own restatement, kind "port") on the same window, rank 0, N=1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_COPY_GBS = 6290.0      # ... measured copy peak (the achievable streaming rate)
PROFILE_ROUND = "r06"      # profiles/<round>_*_pmc_hbm.csv: the PMC passes `traffic` is read from (bench.py cannot collect counters itself)
MFMA_F64_PEAK_TF = 78.6    # ... dense FP64 MFMA (v_mfma_f64_16x16x4_f64: 64 cycles / instruction / SIMD)
MFMA_NOTE = ("latency-bound, not MFMA-bound: the critical path is the pivot chain of the chains on it (a leaf piece + one separator per "
             "level, 16 pivots at a time inside one workgroup: chol_chain.h) plus one hand-over per level; DESIGN.md 3.2")


def _attach_traffic(roofline, csv_name, kernel_prefix, nbytes, check_bytes=True):
    """HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md; bench.py itself cannot collect counters).  The CSV row carries the algorithmic byte count the pass was
    taken at: a different count now means the kernel or the workload changed, and the stale figure is not reported."""
    import csv
    pmc = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(pmc):
        return
    for row in csv.reader(open(pmc)):
        if row and kernel_prefix in row[0] and len(row) >= 4:
            try:
                fetch_kb, write_kb = float(row[2]), float(row[3])
                recorded = int(float(row[4])) if len(row) >= 5 and row[4] else None
            except ValueError:
                continue
            if check_bytes and recorded is not None and abs(recorded - nbytes) > 0.01 * nbytes:
                roofline["traffic_stale"] = "profiles/%s was taken at %d algorithmic bytes per launch, this run moves %d" % (csv_name, recorded, int(nbytes))
                return
            roofline["traffic"] = int((2.0 * fetch_kb + write_kb) * 1024)
            roofline["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % csv_name
            return


def short_leg(which, device):
    """C3 / C4 for a few steps inside the default run (the driver times them with everything else): same step definition as the
    headline — full solves from the device-resident initial guess, the reference's options for that path — plus the roofline of
    the configuration's Jacobian evaluation (relative-pose factors: SURVEY.md 8(d))."""
    from beam_slam_amd import synthetic
    from beam_slam_amd.gpu import GpuSolver
    pr = synthetic.c3() if which == "c3" else synthetic.c4()
    g = GpuSolver(device)
    pr.load(g)
    g.finalize()
    if which == "c3":
        opt = g.options_vio()
        opt.max_solver_time_in_seconds = 0.0
    else:
        opt = g.options_default()
        opt.max_num_iterations = 10
    steps, warmup = 6, 2
    for _ in range(warmup):
        g.reset_values(); s = g.solve(opt)
    t0 = time.perf_counter()
    n_it = 0
    for _ in range(steps):
        g.reset_values(); s = g.solve(opt)
        n_it += s.num_linear_solves
    dt = time.perf_counter() - t0
    ms_e, nb_e = g.time_eval_ms(20), g.eval_bytes()
    ach = nb_e / (ms_e * 1e-3) / 1e9
    leg = {"workload": {"c3": "C3: LIO window, 100 keyframes, 20000 relative-pose(+extrinsics) + 99 IMU factors",
                        "c4": "C4: global-mapper pose graph, 5000 poses, 50000 constraints (block-sparse PCG path)"}[which],
           "value": round(n_it / dt, 2), "unit": "LM iterations/s", "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3),
           "lm_iterations_per_solve": round(n_it / steps, 2), "pcg_iterations_per_solve": int(s.num_inner_iterations),
           "pcg_relative_tolerance": opt.pcg_tolerance if which == "c4" else None,
           "final_cost": s.final_cost, "initial_cost": s.initial_cost,
           "roofline": {"bound": "hbm", "kernel": "relative-pose (+ IMU) Jacobian evaluation", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_per_launch": int(nb_e), "ms_per_launch": round(ms_e, 5),
                        "timing": "20 back-to-back evaluations between two HIP events on the solver's stream"}}
    if which == "c3":
        prof = g.profile_step(opt, reps=10)
        leg["phases_us_per_lm_step"] = {k: round(1e3 * v[0], 2) for k, v in prof.items()}
    if which == "c4":
        # the same solves with the inner tolerance a caller may choose instead of the reference-equivalent default: labelled, never the leg's value
        opt.pcg_tolerance = 1e-6
        g.reset_values(); s2 = g.solve(opt)
        t0 = time.perf_counter(); n2 = 0
        for _ in range(steps):
            g.reset_values(); s2 = g.solve(opt)
            n2 += s2.num_linear_solves
        dt2 = time.perf_counter() - t0
        leg["inexact_inner_tolerance_1e-6"] = {"value": round(n2 / dt2, 2), "ms_per_step": round(1e3 * dt2 / steps, 3), "pcg_iterations_per_solve": int(s2.num_inner_iterations),
                                               "final_cost": s2.final_cost, "final_cost_rel_to_default": abs(s2.final_cost - s.final_cost) / s.final_cost,
                                               "note": "pcg_tolerance set explicitly by the caller; the default (1e-10) is the reference-equivalent step"}
    g.close()
    return leg


def batch_leg(problems, device, steps, label, opt_of=None, lone_steps=0):
    """The windows `problems` on ONE device through bsgpu_solve_batch (one set of launches per LM iteration for all of them,
    csrc/bsgpu_batch.cpp): BASELINE config 5's workload folded onto a single GPU, the reference's own window sizes (vio.yaml:3,56),
    its lidar-inertial windows (lio.yaml) or its submap pose graphs (submap_pose_graph_optimization.cpp:22-150).  lone_steps > 0: the
    first window is also solved alone that many times, so that the line carries the ratio."""
    from beam_slam_amd.gpu import GpuSolver
    gs = []
    for pr in problems:
        g = GpuSolver(device); pr.load(g)
        g.set_plan_preference(True)   # (BSGPU_PLAN_THROUGHPUT: the window is one of many — include/bsgpu.h bsgpu_set_plan_preference)
        g.finalize(); gs.append(g)
    n_win = len(gs)
    opt = opt_of(gs[0]) if opt_of else gs[0].options_vio()
    opt.max_solver_time_in_seconds = 0.0
    for _ in range(2):
        for g in gs: g.reset_values()
        GpuSolver.solve_batch(gs, opt)
    w0, r0 = GpuSolver.batch_stats()
    t0 = time.perf_counter()
    n_it = 0
    for _ in range(steps):
        for g in gs: g.reset_values()
        n_it += sum(s.num_linear_solves for s in GpuSolver.solve_batch(gs, opt))
    dt = time.perf_counter() - t0
    w1, r1 = GpuSolver.batch_stats()
    leg = {"workload": label, "windows": n_win, "value": round(n_it / dt, 1), "unit": "LM iterations/s (aggregate)", "steps": steps,
           "ms_per_step": round(1e3 * dt / steps, 3), "lm_iterations_per_window_and_solve": round(n_it / steps / n_win, 2),
           "windows_on_the_batched_launches": (w1 - w0) // max(1, steps), "us_per_round_of_launches": round(1e6 * dt / max(1, r1 - r0), 1)}
    if lone_steps > 0:
        g = GpuSolver(device); problems[0].load(g); g.finalize()   # (by itself: a context of its own, planned for latency — the default)
        gs.append(g)
        for _ in range(3):
            g.reset_values(); g.solve(opt)
        t0 = time.perf_counter()
        n1 = 0
        for _ in range(lone_steps):
            g.reset_values(); n1 += g.solve(opt).num_linear_solves
        dt1 = time.perf_counter() - t0
        leg["one_window_alone"] = {"value": round(n1 / dt1, 1), "unit": "LM iterations/s", "us_per_lm_iteration": round(1e6 * dt1 / max(1, n1), 1), "steps": lone_steps}
        leg["aggregate_over_one_window_alone"] = round(leg["value"] / max(1e-9, n1 / dt1), 2)
    for g in gs: g.close()
    return leg


def host_cycle_leg():
    """One optimisation cycle of the host mirror (beam_slam_amd/host/: GpuGraph::optimize, clone, release of the previous snapshot,
    update with a sliding-window transaction; fixed_lag_smoother.cpp:220,274,308) through the C++ program tests/host/bench_host.cpp that
    __graft_entry__.build() compiles against libbsgpu: at C2's size (the top-level keys), and at the sizes the reference itself runs at
    14-25 Hz — a 20-key-frame x 500-landmark visual-inertial window (vio.yaml:2-3) and a 20-key-frame lidar-inertial window (lio.yaml:2)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "host", "bench_host.bin")
    if not os.path.exists(exe):
        return {"skipped": "tests/host/bench_host.bin is not built (__graft_entry__.build())"}

    def run(*argv):
        try:
            out = subprocess.run([exe, *argv], capture_output=True, text=True, timeout=240).stdout
        except Exception as e:   # noqa: BLE001
            return {"skipped": "bench_host failed: %r" % (e,)}
        for line in out.splitlines():
            if line.startswith("HOST_CYCLE_JSON "):
                return json.loads(line[len("HOST_CYCLE_JSON "):])
        return {"skipped": "no summary line", "tail": out[-400:]}
    leg = run()
    leg["reference_sized_vio_window"] = run("20", "500")
    leg["reference_sized_lidar_inertial_window"] = run("20", "15", "lio")
    return leg


def _relaunch(n):
    """Plain `python bench.py --gpus N` (no launcher): re-exec under torch.distributed.run with one rank per GPU on 127.0.0.1."""
    import socket
    import subprocess
    if not os.environ.get("BSGPU_BENCH_SAME_DEVICE"):
        import torch
        have = torch.cuda.device_count()
        if n > have:
            sys.exit("bench.py: --gpus %d but this node shows %d GPU(s); one rank per GPU is the only layout this bench runs" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-kf", type=int, default=200)
    ap.add_argument("--n-lm", type=int, default=50000)
    ap.add_argument("--no-past-l3", dest="past_l3", action="store_false",
                    help="skip the Jacobian-evaluation measurement on the 800 KF x 300k-landmark window (working set above the Infinity Cache)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"],
                    help="c2 (default) is the headline; c3 / c4 are the other BASELINE configs, for BASELINE.md")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip the short C3 and C4 legs the default (C2, one GPU) run appends under \"other_configs\"")
    ap.add_argument("--sustained-seconds", type=float, default=2.5,
                    help="after the timed region: back-to-back C2 solves for this long (\"sustained\"; 0 = skip)")
    ap.add_argument("--windows-per-gpu", type=int, default=0,
                    help="W > 0: every rank advances W independent C2-shaped windows per step with ONE bsgpu_solve_batch (what an N-GPU run of "
                         "BASELINE config 5 multiplies when a GPU holds more than one submap); value = LM iterations of all windows of all ranks / "
                         "the slowest rank's time")
    ap.add_argument("--consensus", action="store_true",
                    help="C5 with shared-pose consensus: the N windows are consecutive submaps of ONE trajectory, neighbours share their boundary "
                         "key frame, and a step is the whole message-passing solve of the merged graph (beam_slam_amd/sharding.py); the messages "
                         "are the only collective (RCCL all-reduce of a few KB per round)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as documented: start the N ranks ourselves (one process per GPU under torch.distributed.run, the
        # same command line the driver uses) and hand their single JSON line through
        return _relaunch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or plain `python bench.py --gpus N`)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BSGPU_BENCH_FORCE_DIST"):   # (the env knob exercises the RCCL path on a 1-GPU box)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BSGPU_BENCH_BACKEND", "nccl")   # ("gloo": the tests run two ranks on ONE GPU, which RCCL refuses)
        if os.environ.get("BSGPU_BENCH_SAME_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from beam_slam_amd import capi, sharding, synthetic
    from beam_slam_amd.gpu import GpuSolver

    # ---- workload: C2 at N=1, C5 instances (independent C2-shaped windows) at N>1 ----------------
    if args.workload == "c3":
        pr = synthetic.c3()
        workload = "C3: LIO window, 100 keyframes, 20000 relative-pose(+extrinsics) + 99 IMU factors"
    elif args.workload == "c4":
        pr = synthetic.c4()
        workload = "C4: global-mapper pose graph, 5000 poses, 50000 constraints (block-sparse PCG path)"
    elif args.consensus:
        pr = synthetic.chain_window(rank, world, n_kf=args.n_kf, n_lm=args.n_lm, seed=20250620)
        workload = "C5 with shared-pose consensus: %d consecutive submaps (%d KF x %d landmarks each) of one trajectory, neighbours share a key frame" % (world, args.n_kf, args.n_lm)
    elif world == 1:
        pr = synthetic.vio_window(n_kf=args.n_kf, n_lm=args.n_lm, seed=20250620)
        workload = "C2: %d-keyframe x %d-landmark VIO window" % (args.n_kf, args.n_lm)
    else:
        (window,) = sharding.assign_windows(world, world, rank)     # one window per GPU
        pr = synthetic.vio_window(n_kf=args.n_kf, n_lm=args.n_lm, seed=sharding.window_seed(20250620, window))
        workload = "C5: %d independent C2 windows (%d KF x %d landmarks each)" % (world, args.n_kf, args.n_lm)
    g = GpuSolver(local_rank)
    pr.load(g)
    g.finalize()
    opt = g.options_vio()
    opt.max_solver_time_in_seconds = 0.0  # every step does the full <= 10 iterations
    wins = None
    if args.windows_per_gpu > 0 and args.workload == "c2" and not args.consensus:
        # W windows per rank, all of them through ONE bsgpu_solve_batch per step (window ids rank * W + w: every window of the job is distinct)
        W = args.windows_per_gpu
        wins = [g]
        if world > 1 or W > 1:
            g.close()
            wins = []
            for w in range(W):
                prw = synthetic.vio_window(n_kf=args.n_kf, n_lm=args.n_lm, seed=sharding.window_seed(20250620, rank * W + w))
                gw = GpuSolver(local_rank); prw.load(gw); gw.finalize(); wins.append(gw)
            g, pr = wins[0], prw
        workload = "C5: %d independent C2 windows (%d KF x %d landmarks each), %d per GPU through one bsgpu_solve_batch per step" % (world * W, args.n_kf, args.n_lm, W)
    if args.workload == "c4":             # the global mapper passes default ceres options
        opt = g.options_default()
        opt.max_num_iterations = 10

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    cons = None
    if args.consensus:
        import numpy as np
        opt.function_tolerance = 1e-12; opt.gradient_tolerance = 1e-12; opt.parameter_tolerance = 1e-12   # (the rounds cannot agree better than the windows are solved)
        opt.max_num_iterations = 30
        mp_win = sharding.MessagePassing(g, pr, rank, pr.meta["shared"], opt)
        cons = {"rounds": 0, "round_ms": [], "lm_iterations": 0, "cost": 0.0, "dz": 0.0}
        if dist is not None:
            import torch
            on_gpu = dist.get_backend() == "nccl"

            def all_reduce(a):   # the exchange of the messages: ONE all-reduce (RCCL over xGMI on the GPU box)
                t = torch.from_numpy(np.ascontiguousarray(a, np.float64).copy())
                if on_gpu:
                    t = t.cuda()
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return t.cpu().numpy()
        else:
            all_reduce = None

    def one_step():
        if cons is not None:
            mp_win.reset()
            t_r = [time.perf_counter()]
            its = [0]

            def on_round(rnd, dz, cost):
                t_r.append(time.perf_counter())
                its[0] += mp_win.last_summary.num_linear_solves
            hist = sharding.message_passing_rounds([mp_win], 12, all_reduce=all_reduce, n_parts=world, tol=1e-8, on_round=on_round)
            cons["rounds"] = len(hist); cons["cost"] = hist[-1][2]; cons["dz"] = hist[-1][1]
            cons["round_ms"] = [round(1e3 * (b - a), 3) for a, b in zip(t_r[:-1], t_r[1:])]
            cons["lm_iterations"] = its[0]
            sm = mp_win.last_summary
            sm.num_linear_solves = its[0]
            return sm
        if wins is not None:
            for gw in wins: gw.reset_values()
            sums = GpuSolver.solve_batch(wins, opt)
            sm = sums[0]
            sm.num_linear_solves = sum(x.num_linear_solves for x in sums)
            return sm
        g.reset_values()
        return g.solve(opt)

    for _ in range(args.warmup):
        s = one_step()
    barrier()
    t0 = time.perf_counter()
    n_it = 0
    dev_s = 0.0
    for _ in range(args.steps):
        s = one_step()           # bsgpu_solve returns after its final stream synchronisation
        n_it += s.num_linear_solves
        dev_s += s.device_time_in_seconds
    barrier()
    dt = time.perf_counter() - t0

    tot_it, max_dt = float(n_it), dt
    ranks_seen, per_rank = 1, [round(n_it / dt, 2)]
    if dist is not None:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tot_it, max_dt = sharding.aggregate(dist, n_it, dt, device=dev)
        ranks_seen, per_rank = sharding.per_rank_rates(dist, n_it, dt, device=dev)

    if ranks_seen != args.gpus:
        sys.exit("bench.py: --gpus %d but %d rank(s) reported: the line would not describe the job asked for" % (args.gpus, ranks_seen))
    out = None
    if rank == 0:
        # ---- rooflines, all measured IN SITU: HIP events at the phase boundaries of real LM steps on the solver's stream
        # (bsgpu_profile_step); the kernel-trace average of the same kernels is in profiles/r02_*_kernel_stats.csv
        has_vis = pr.n_factors(0) > 0 and not args.consensus and wins is None
        roofline = roofline_mfma = kernels = phases = None
        if args.workload in ("c3", "c4"):
            # no reprojection factors: the roofline is that of the relative-pose evaluation (~990 B per factor),
            # timed with HIP events on the solver's stream (20 back-to-back evaluations of every factor type of the window)
            ms_e, nb_e = g.time_eval_ms(20), g.eval_bytes()
            ach = nb_e / (ms_e * 1e-3) / 1e9
            rel_kernel = "relpose_imu_eval_kernel" if (args.workload == "c3" and os.environ.get("BSGPU_EVAL_MERGE", "2") == "2") else "small_eval_set_kernel"
            roofline = {"bound": "hbm", "kernel": rel_kernel + (" (relative-pose factors + the window's IMU factors as its first workgroups)" if rel_kernel != "small_eval_set_kernel" else " (the relative-pose factors and the prior on the first pose in one launch)"),
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_per_launch": int(nb_e), "ms_per_launch": round(ms_e, 5),
                        "timing": "20 back-to-back evaluations (residuals + Jacobians of every factor type) between two HIP events on the solver's stream",
                        "working_set_mb": round(nb_e / 1e6, 1), "cache_residency": "below the 256 MiB Infinity Cache", "traffic": None}
            _attach_traffic(roofline, "%s_%s_pmc_hbm.csv" % (PROFILE_ROUND, args.workload), rel_kernel + "<true", nb_e)
        if args.workload == "c3":          # dense Schur path without landmarks: phases and the factorisation's figure
            prof = g.profile_step(opt, reps=20)
            phases = {k: round(1e3 * v[0], 2) for k, v in prof.items()}
            ms_f, flops = prof["factor"]
            tf = flops / (ms_f * 1e-3) / 1e12
            roofline_mfma = {"bound": "mfma", "kernel": "chol_fused_kernel (+ chol_backsolve_fused_kernel: %.1f us)" % (1e3 * prof["backsolve"][0]),
                             "achieved": round(tf, 3), "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F64_PEAK_TF, 4),
                             "flops_per_launch": int(flops), "ms_per_launch": round(ms_f, 5), "note": MFMA_NOTE}
            kernels = []
        if args.workload == "c4":          # block-sparse PCG path: what an inner iteration moves
            nbr, nnzb = g.bsr_info()
            it_bytes = nnzb * (72 + 4) + 6 * 3 * nbr * 8      # the 3x3 blocks + their column indices, and x, r, p, q, z, M^-1 once each
            pcg = {"block_rows": nbr, "nnz_blocks_3x3": nnzb, "bytes_per_inner_iteration": int(it_bytes),
                   "inner_iterations_per_solve": int(s.num_inner_iterations),
                   "note": "pcg_persistent_kernel keeps its blocks in registers: an inner iteration is latency (two cross-XCD reductions), not bandwidth"}
        if has_vis and not (args.workload == "c4"):
            prof = g.profile_step(opt, reps=20)
            phases = {k: round(1e3 * v[0], 2) for k, v in prof.items()}          # microseconds per LM step
            ms, nbytes = prof["eval_reproj"]
            achieved = nbytes / (ms * 1e-3) / 1e9
            working_set_mb = nbytes / 1e6
            # (a window that also has IMU factors evaluates them as the first workgroups of the same launch: k_small.hip)
            merged_imu = pr.n_factors(capi.F_IMU_DELTA) + pr.n_factors(capi.F_IMU_PRIOR) > 0 and os.environ.get("BSGPU_EVAL_MERGE", "2") == "2"
            eval_kernel = "visual_imu_eval_kernel<true>" if merged_imu else "reproj_eval_kernel<true>"
            roofline = {"bound": "hbm", "kernel": eval_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBS, 4),
                        "bytes_per_launch": int(nbytes), "ms_per_launch": round(ms, 5),
                        "timing": "in situ: HIP events around the launch inside 20 full LM steps (bsgpu_profile_step)",
                        "working_set_mb": round(working_set_mb, 1),
                        "cache_residency": "below the 256 MiB Infinity Cache: the stream is MALL/fabric traffic, see past_l3" if working_set_mb < 256 else "above the 256 MiB Infinity Cache",
                        "traffic": None}
            if merged_imu:
                roofline["kernel_note"] = ("the reprojection factors' evaluation (the bytes counted) plus the window's %d IMU factors as the launch's first workgroups "
                                           "(~6.1 KB each, not counted)" % (pr.n_factors(capi.F_IMU_DELTA) + pr.n_factors(capi.F_IMU_PRIOR)))
            if world == 1 and args.workload == "c2" and args.n_kf == 200 and args.n_lm == 50000:
                _attach_traffic(roofline, "%s_c2_pmc_hbm.csv" % PROFILE_ROUND, eval_kernel, nbytes)
            ms_f, flops = prof["factor"]
            tf = flops / (ms_f * 1e-3) / 1e12
            roofline_mfma = {"bound": "mfma", "kernel": "chol_fused_kernel (+ chol_backsolve_fused_kernel: %.1f us)" % (1e3 * prof["backsolve"][0]),
                             "achieved": round(tf, 3), "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F64_PEAK_TF, 4),
                             "flops_per_launch": int(flops), "ms_per_launch": round(ms_f, 5),
                             "note": MFMA_NOTE}
            kernels = []
            band = os.environ.get("BSGPU_PAIRS_BAND", "1") != "0" and pr.n_factors(0) >= 150000
            for name, kern, pmc_name in (("landmark", "landmark_kernel (+ the step's clearing of S, gradient and diag(H))", "bsg::landmark_kernel"),
                                         ("pairs", "pairs_band_nocr_kernel / pairs_band_kernel (WRITE_SIZE counts its FP64 atomic adds at 32 B each)" if band else "pairs_kernel", "bsg::pairs_band" if band else "bsg::pairs_kernel"),
                                         ("backsub", "backsub_mcc_kernel (+ small_mcc)", "bsg::backsub_mcc_kernel"),
                                         ("candidate", "update + visual_imu_eval_kernel<false> / reproj_eval_kernel<false> + reduction", None)):
                ms_k, by = prof[name]
                k = {"phase": name, "kernel": kern, "us": round(1e3 * ms_k, 2), "algorithmic_bytes": int(by),
                     "achieved_gbs": round(by / (ms_k * 1e-3) / 1e9, 1), "frac_hbm": round(by / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                if pmc_name and world == 1 and args.workload == "c2" and args.n_kf == 200 and args.n_lm == 50000:
                    tr = {"traffic": None}
                    _attach_traffic(tr, "%s_c2_pmc_hbm.csv" % PROFILE_ROUND, pmc_name, by, check_bytes=False)
                    if tr.get("traffic"):
                        k["traffic"] = tr["traffic"]; k["traffic_over_algorithmic"] = round(tr["traffic"] / by, 3); k["traffic_source"] = tr["traffic_source"]
                kernels.append(k)
            if args.past_l3 and world == 1 and args.workload == "c2":
                # the same kernel on a working set the Infinity Cache cannot hold: 800 keyframes x 300 000 landmarks, ~2.4 M observations
                # (J + r ~ 385 MB per evaluation); Jacobian evaluation only (50 launches, HIP events) — nothing is solved
                big = synthetic.vio_window(n_kf=800, n_lm=300000, seed=20250621)
                gb = GpuSolver(local_rank)
                big.load(gb)
                gb.finalize()
                ms_b = gb.time_reproj_jacobian_ms(20)
                nb_b = gb.reproj_jacobian_bytes()
                ach_b = nb_b / (ms_b * 1e-3) / 1e9
                roofline["past_l3"] = {"workload": "800 KF x 300000 landmarks, %d observations" % big.n_factors(0), "bytes_per_launch": int(nb_b),
                                       "ms_per_launch": round(ms_b, 5), "achieved": round(ach_b, 1), "frac": round(ach_b / HBM_PEAK_GBS, 4),
                                       "frac_of_measured_copy_peak": round(ach_b / HBM_COPY_GBS, 4),
                                       "timing": "20 back-to-back launches between two HIP events (each launch streams more than the cache holds)"}
                gb.close()
        metric = {"c2": "LM iterations/sec + ms/graph-solve, 200KF x 50k-landmark VIO window",
                  "c3": "LM iterations/sec + ms/graph-solve, LIO fixed-lag window (100 KF, 20k relative-pose + IMU factors)",
                  "c4": "LM iterations/sec + ms/graph-solve, global-mapper pose graph (5k poses, 50k constraints)"}[args.workload]
        solver_options = {"c2": "vio.yaml:7-17 (<= 10 iterations, tolerances 1.5e-7), max_solver_time lifted",
                          "c3": "vio.yaml:7-17 (<= 10 iterations, tolerances 1.5e-7), max_solver_time lifted",
                          "c4": "default ceres::Solver::Options as the global mapper passes them (SURVEY.md §3.4), max_num_iterations 10"}[args.workload]
        if args.consensus:
            solver_options = {"c2": "default ceres::Solver::Options with tolerances 1e-12 and <= 30 iterations per window and round (the rounds cannot agree better than the windows are solved)"}
        out = {
            "metric": metric,
            "value": round(tot_it / max_dt, 2), "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * max_dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n_obs": int(pr.meta.get("n_obs", 0)), "n_imu_factors": int(pr.meta.get("n_imu", 0)),
                       "pcg_iterations_per_solve": s.num_inner_iterations,
                       "lm_iterations_per_solve": round(n_it / args.steps, 2),
                       "solver_options": solver_options,
                       "final_cost": s.final_cost, "initial_cost": s.initial_cost,
                       "device_ms_per_solve": round(1e3 * dev_s / args.steps, 3),
                       "parallelism": ("%d window%s per GPU, no collective" % (max(1, args.windows_per_gpu), "s" if args.windows_per_gpu > 1 else "")) if (world > 1 or wins is not None) else "single GPU",
                       "n_ranks_seen": ranks_seen, "per_rank_lm_iterations_per_s": per_rank},
            "roofline": roofline,
        }
        if phases and not roofline_mfma:
            out["phases_us_per_lm_step"] = phases
        if roofline_mfma:
            out["roofline_mfma"] = roofline_mfma
            if kernels:
                out["kernels"] = kernels
            out["phases_us_per_lm_step"] = phases
            out["phases_note"] = ("bsgpu_profile_step enqueues accepted steps WITHOUT an assembly ahead: 'candidate' holds the cost-only pass at the candidate, which a solve of a "
                                  "window of >= 150 000 reprojection factors no longer runs on accepted steps (the decision is also taken on the device and the candidate is evaluated "
                                  "once, with Jacobians: DESIGN.md 9 (4)); a solve's own per-iteration time is ms_per_step / lm_iterations_per_solve")
        if args.workload == "c4":
            out["pcg"] = pcg
        if cons is not None:
            out["consensus"] = {"rounds": cons["rounds"], "ms_per_round": cons["round_ms"], "merged_graph_cost": cons["cost"],
                                "last_change_of_a_shared_value": cons["dz"], "lm_iterations_per_window": cons["lm_iterations"],
                                "exchange": "one all-reduce per round: a 15 x 15 information matrix, its mean and the key frame's value per directed pair of neighbours",
                                "note": "a step is the whole consensus solve from the initial values; value counts the LM iterations of all windows and rounds"}
        default_c2 = world == 1 and args.workload == "c2" and not args.consensus and wins is None
        # ---- sustained: back-to-back solves of the same window for a few seconds (the timed region above is 0.1 s of GPU work: too
        # short for an external utilisation sampler to see; this is the same loop, longer)
        if default_c2 and args.sustained_seconds > 0:
            t1 = time.perf_counter()
            n_s, it_s = 0, 0
            while time.perf_counter() - t1 < args.sustained_seconds:
                s2 = one_step()
                n_s += 1
                it_s += s2.num_linear_solves
            dt_s = time.perf_counter() - t1
            out["sustained"] = {"seconds": round(dt_s, 2), "solves": n_s, "value": round(it_s / dt_s, 2), "unit": "LM iterations/s",
                                "note": "back-to-back solves of the headline window after the timed region, same options"}
        # ---- the other single-GPU configurations of BASELINE.json, a few steps each (their own full runs: --workload c3 / c4)
        if default_c2 and args.other_configs:
            g.close()
            out["other_configs"] = {"c3": short_leg("c3", local_rank), "c4": short_leg("c4", local_rank)}
            # several windows per GPU (what an N-GPU run multiplies): 8 C2-shaped windows — config 5 folded onto one device — and 32 windows of
            # the reference's own size, each call one bsgpu_solve_batch
            seed5 = sharding.window_seed(20250620, 0)
            out["other_configs"]["c5_on_one_gpu"] = batch_leg([synthetic.vio_window(n_kf=args.n_kf, n_lm=args.n_lm, seed=seed5 + w) for w in range(8)], local_rank, 5,
                                                              "8 independent C2 windows (BASELINE config 5) on ONE GPU, one bsgpu_solve_batch per step")
            out["other_configs"]["reference_sized_windows"] = batch_leg([synthetic.vio_window(n_kf=20, n_lm=500, seed=20250700 + w) for w in range(32)], local_rank, 20,
                                                                        "32 independent windows of 20 key frames x 500 landmarks (the reference's own window size, vio.yaml:3,56), one bsgpu_solve_batch per step; "
                                                                        "one_window_alone = the same window by itself (the latency case)", lone_steps=40)
            out["other_configs"]["lidar_inertial_windows"] = batch_leg([synthetic.lio_window(n_kf=20, n_rel=300, seed=20250800 + w) for w in range(32)], local_rank, 20,
                                                                       "32 independent lidar-inertial windows of 20 key frames x 300 scan-registration factors + IMU factors (lio.yaml, "
                                                                       "scan_to_map_registration.cpp:74-78), one bsgpu_solve_batch per step", lone_steps=40)

            def opt_pg(g):
                o = g.options_default(); o.max_num_iterations = 10
                return o
            out["other_configs"]["submap_pose_graphs"] = batch_leg([synthetic.pose_graph(n_pose=200, n_loop=300, seed=20250900 + w) for w in range(16)], local_rank, 10,
                                                                   "16 independent pose graphs of 200 poses on the dense exact path (submap_pose_graph_optimization.cpp:22-150), "
                                                                   "one bsgpu_solve_batch per step", opt_of=opt_pg, lone_steps=20)
            out["other_configs"]["host_cycle"] = host_cycle_leg()
        # ---- CPU baseline: the oracle on the same window (bounded samples, same options): with every usable core, and with the six
        # threads the reference's own configuration gives Ceres (beam_slam_launch/config/vio.yaml:11 num_threads: 6)
        if world == 1 and not args.no_cpu_baseline and args.workload != "c4" and not args.consensus and wins is None:   # (C4 at full size: the oracle's dense solve does not finish in bench time)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle import Oracle, usable_cpus

            def cpu_leg(threads, budget_s, max_solves):
                o = Oracle(threads)
                pr.load(o)
                cpu_dt, n_solves, n_its = 0.0, 0, 0
                while cpu_dt < budget_s and n_solves < max_solves:
                    o.reset_values()
                    t1 = time.perf_counter()
                    so = o.solve(opt)
                    cpu_dt += time.perf_counter() - t1
                    n_solves += 1
                    n_its += so.num_linear_solves
                leg = {"value": round(n_its / cpu_dt, 3), "unit": "LM iterations/s", "cores": o.threads,
                       "kind": "port", "ms_per_solve": round(1e3 * cpu_dt / n_solves, 1), "final_cost": so.final_cost,
                       "sample": "%d full solves (%d LM iterations each) of the same window, OpenMP oracle, %.1f s of CPU wall time"
                                 % (n_solves, so.num_linear_solves, cpu_dt)}
                o.close()
                return leg
            out["cpu_baseline"] = cpu_leg(None, 10.0, 8)
            if usable_cpus() >= 6 and out["cpu_baseline"]["cores"] != 6:
                ref = cpu_leg(6, 8.0, 4)
                ref["note"] = "the thread count of the reference's own configuration (vio.yaml:11 num_threads: 6)"
                out["cpu_baseline"]["reference_thread_count"] = ref
            out["config"]["final_cost_rel_diff_vs_cpu"] = abs(s.final_cost - out["cpu_baseline"]["final_cost"]) / out["cpu_baseline"]["final_cost"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
