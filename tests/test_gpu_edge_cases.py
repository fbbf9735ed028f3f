"""Degenerate shapes of the solve path, HIP vs oracle: empty / ragged inputs, everything constant, a single factor,
landmarks seen once (rank-deficient H_ll kept solvable only by the LM diagonal), zero iterations, non-finite input,
tile boundaries of the reduced system."""
import numpy as np
import pytest

from beam_slam_amd import capi, synthetic
from beam_slam_amd.problem import Problem
from helpers import mixed_problem

pytestmark = pytest.mark.gpu


def _both(pr, oracle_cls, gpu_solver_cls):
    g, o = gpu_solver_cls(0), oracle_cls()
    pr.load(g); pr.load(o)
    return g, o


def _same_solve(g, o, opt=None, tol_x=1e-7):
    sg, so = g.solve(opt), o.solve(opt)
    assert sg.termination_type == so.termination_type
    assert [i.step_is_successful for i in g.iterations()] == [i.step_is_successful for i in o.iterations()]
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * max(so.final_cost, 1e-30)
    assert np.abs(g.get_blocks() - o.get_blocks()).max() <= tol_x
    return sg, so


def test_blocks_without_factors_is_an_error_free_noop(oracle_cls, gpu_solver_cls):
    pr = Problem()
    pr.add_block([1.0, 2.0, 3.0]); pr.add_quat([1.0, 0, 0, 0])
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    sg, so = g.solve(), o.solve()
    assert sg.termination_type == so.termination_type == capi.CONVERGENCE
    assert sg.initial_cost == sg.final_cost == 0.0
    assert np.array_equal(g.get_blocks(), pr.values)


def test_blocks_no_factor_touches_are_left_out_of_the_problem(oracle_cls, gpu_solver_cls):
    """A sliding window leaves variables behind that no constraint touches any more — landmarks whose last observation was
    marginalised, states of sensor models that are switched off.  [EXT] Ceres drops such parameter blocks from the reduced
    program; here they get no tangent columns (like constant blocks), keep their values, and neither make the covariance
    singular nor cost tiles of the reduced system."""
    pr = synthetic.vio_window(n_kf=6, n_lm=40, seed=9)
    n0 = pr.n_blocks
    orphan_lm = [pr.add_block([50.0 + i, -3.0, 7.0]) for i in range(5)]          # landmark-shaped blocks without observations
    orphan_q = pr.add_quat([0.5, 0.5, 0.5, 0.5])                                 # an unused pose
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    g.finalize(); o.finalize()
    offs = [g.tangent_offset(b) for b in range(pr.n_blocks)]
    assert offs == [o.tangent_offset(b) for b in range(pr.n_blocks)]
    assert all(offs[b] == -1 for b in orphan_lm + [orphan_q]) and g.num_parameters_tangent() == o.num_parameters_tangent()
    ref = synthetic.vio_window(n_kf=6, n_lm=40, seed=9)                          # the same window without the orphans
    r = gpu_solver_cls(0)
    ref.load(r)
    r.finalize()
    assert g.num_parameters_tangent() == r.num_parameters_tangent()
    sg, so = _same_solve(g, o)
    sr = r.solve()
    assert sg.num_iterations == sr.num_iterations and abs(sg.final_cost - sr.final_cost) <= 1e-12 * sr.final_cost
    x = g.get_blocks()
    for b in orphan_lm + [orphan_q]:
        assert np.array_equal(pr.block(b, x), pr.block(b))                        # untouched
    assert np.abs(x[:ref.values.size] - r.get_blocks()).max() <= 1e-12
    kf = pr.meta["kf_blocks"]
    cg, co = g.covariance(int(kf[2, 1]), int(kf[3, 1])), o.covariance(int(kf[2, 1]), int(kf[3, 1]))
    assert np.all(np.isfinite(cg)) and np.abs(cg - co).max() <= 1e-6 * np.abs(co).max()
    with pytest.raises(capi.SolverError):
        g.covariance(orphan_lm[0], orphan_lm[0])                                  # not part of the problem
    assert n0 + 6 == pr.n_blocks


def test_no_blocks_at_all_is_rejected(gpu_solver_cls):
    g = gpu_solver_cls(0)
    with pytest.raises(capi.SolverError) as e:
        g.finalize()
    assert e.value.code == capi.ERR_INVALID


def test_all_poses_constant_pure_triangulation(oracle_cls, gpu_solver_cls):
    """Every pose block held constant: the reduced camera system is EMPTY (n_pose = 0), only landmarks move."""
    pr = synthetic.vio_window(n_kf=5, n_lm=30, seed=9, track_min=3, track_max=5, with_imu=False)
    for b in pr.meta["kf_blocks"].ravel():
        pr.is_const[int(b)] = 1
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    g.finalize(); o.finalize()
    assert g.num_parameters_tangent() == o.num_parameters_tangent() == 3 * 30
    sg, so = _same_solve(g, o)
    assert sg.final_cost < sg.initial_cost     # (the fixed, perturbed poses bound what triangulation alone can gain)
    x = g.get_blocks()
    for b in pr.meta["kf_blocks"].ravel():
        assert np.array_equal(pr.block(int(b), x), pr.block(int(b)))


def test_everything_constant_gives_fixed_cost_only(oracle_cls, gpu_solver_cls):
    pr = mixed_problem(2, n_state=3, n_lm=6, consistent=True)
    pr.is_const = [1] * pr.n_blocks
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    cg, co = g.evaluate(gradient=False)[0], o.evaluate(gradient=False)[0]
    assert abs(cg - co) <= 1e-12 * co
    sg, so = g.solve(), o.solve()
    assert sg.num_parameters_tangent == so.num_parameters_tangent == 0
    assert abs(sg.final_cost - so.final_cost) <= 1e-12 * so.final_cost and abs(sg.fixed_cost - so.fixed_cost) <= 1e-12 * so.fixed_cost
    assert np.array_equal(g.get_blocks(), pr.values)


def test_single_factor(oracle_cls, gpu_solver_cls):
    pr = Problem()
    b = pr.add_block([0.3, -0.2, 0.9])
    A = synthetic.sqrt_information_upper(np.diag([0.1, 0.2, 0.3]))
    pr.add_factors(capi.F_ABS_VEC3, [[b]], [np.concatenate([[1.0, 2.0, 3.0], A.ravel()])])
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    _same_solve(g, o, tol_x=1e-9)
    assert np.abs(g.get_blocks() - [1.0, 2.0, 3.0]).max() < 1e-6


def test_landmarks_seen_once_and_twice(oracle_cls, gpu_solver_cls):
    """A landmark with ONE observation has a rank-2 H_ll: only the LM diagonal (min_lm_diagonal 1e-6) keeps its 3x3 system
    solvable, exactly as in Ceres; ragged track lengths 1..5 in one window."""
    pr = synthetic.vio_window(n_kf=6, n_lm=50, seed=21, track_min=1, track_max=5)
    idx = np.concatenate([c[0] for c in pr.factors[capi.F_REPROJ]])
    counts = np.bincount(idx[:, 2])[pr.meta["lm_blocks"]]
    assert counts.min() == 1 and counts.max() >= 4
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    sg, so = g.solve(), o.solve()
    assert sg.is_solution_usable == so.is_solution_usable == 1
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.final_cost


def test_zero_iterations(oracle_cls, gpu_solver_cls):
    pr = mixed_problem(1, consistent=True)
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    opt = g.options_default()
    opt.max_num_iterations = 0
    sg, so = g.solve(opt), o.solve(opt)
    assert sg.termination_type == so.termination_type == capi.NO_CONVERGENCE
    assert sg.num_iterations == so.num_iterations == 0
    assert abs(sg.final_cost - so.final_cost) <= 1e-12 * so.final_cost and sg.final_cost == sg.initial_cost
    assert np.array_equal(g.get_blocks(), pr.values)


def test_non_finite_input_fails_loudly(oracle_cls, gpu_solver_cls):
    """NaN in a parameter block: ceres::Solve returns FAILURE / unusable solution (the reference's fatal path,
    fixed_lag_smoother.cpp:286-295) — never a silent 'converged'."""
    pr = mixed_problem(1, consistent=True)
    v = pr.values.copy()
    v[pr.offset[int(pr.meta["states"][1, 1])]] = np.nan
    pr.values = v
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    sg, so = g.solve(), o.solve()
    assert sg.termination_type == so.termination_type == capi.FAILURE
    assert sg.is_solution_usable == so.is_solution_usable == 0


@pytest.mark.parametrize("n_kf", [4, 5, 9])
def test_reduced_system_at_tile_boundaries(oracle_cls, gpu_solver_cls, n_kf):
    """15 tangent dims per keyframe: n_pose = 60 (inside one 64-wide tile), 75 (one full + one partial tile) and 135."""
    pr = synthetic.vio_window(n_kf=n_kf, n_lm=40, seed=30 + n_kf, track_min=2, track_max=min(4, n_kf))
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    g.finalize()
    assert g.tangent_offset(int(pr.meta["lm_blocks"][0])) == 15 * n_kf
    _same_solve(g, o, tol_x=1e-6)


def test_exactly_one_tile(oracle_cls, gpu_solver_cls):
    """n_pose == 64 exactly: the last real column is the last column of a tile."""
    pr = synthetic.vio_window(n_kf=4, n_lm=40, seed=77, track_min=2, track_max=4)      # 60 dims ...
    extra = pr.add_block([0.1, 0.2, 0.3]); one = pr.add_block([0.5])                    # ... + 3 + 1
    A3, A1 = synthetic.sqrt_information_upper(0.1 * np.eye(3)), np.array([[2.0]])
    pr.add_factors(capi.F_ABS_VEC3, [[extra]], [np.concatenate([[0.0, 0.0, 0.0], A3.ravel()])])
    # a 1-d block enters through a marginal (dense linear) prior: r = 2 (x - 0.25)
    pr.add_marginal([one], A1, [0.0], [0.25])
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    g.finalize()
    assert g.tangent_offset(int(pr.meta["lm_blocks"][0])) == 64
    _same_solve(g, o, tol_x=1e-6)
    x = g.get_blocks()
    assert abs(x[pr.offset[one]] - 0.25) < 1e-9


@pytest.mark.parametrize("make", ["vio", "const_landmarks", "held_poses"])
def test_device_and_host_flattening_build_the_same_tables(oracle_cls, gpu_solver_cls, make, monkeypatch):
    """k_flatten.hip (rocPRIM sorts / scans on the raw factor table) against the host loops of finalize(): same factor
    order, camera-pose ids, pair entries and tile adjacency => bit-identical evaluation and LM trajectory."""
    pr = synthetic.vio_window(n_kf=9, n_lm=300, seed=41, track_min=1, track_max=6)
    if make == "const_landmarks":
        for b in pr.meta["lm_blocks"][::7]:
            pr.is_const[int(b)] = 1
    if make == "held_poses":
        for b in pr.meta["kf_blocks"][0]:
            pr.is_const[int(b)] = 1
    out = {}
    for mode in ("host", "device"):
        monkeypatch.setenv("BSGPU_FLATTEN", mode)
        g = gpu_solver_cls(0)
        pr.load(g)
        cost, r, grad, J = g.evaluate(jacobian=True)
        err = g.reprojection_errors(pr.n_factors(capi.F_REPROJ))
        s = g.solve()
        out[mode] = (cost, r, grad, J, [i.cost for i in g.iterations()], g.get_blocks(), err)
    monkeypatch.delenv("BSGPU_FLATTEN")
    h, d = out["host"], out["device"]
    assert h[0] == d[0] and np.array_equal(h[1], d[1]) and np.array_equal(h[3], d[3])          # evaluation: bit-identical
    assert np.allclose(h[2], d[2], rtol=1e-13, atol=1e-9)                                       # (host-side sum of J^T r)
    assert len(h[4]) == len(d[4]) and np.allclose(h[4], d[4], rtol=1e-10)                       # atomics in the Cholesky: not bitwise
    assert np.abs(h[5] - d[5]).max() < 1e-8
    assert np.array_equal(h[6], d[6])
    o = oracle_cls()
    pr.load(o)
    assert abs(o.solve().final_cost - d[4][-1]) <= 1e-6 * d[4][-1]


def test_device_flattening_declines_what_it_does_not_cover(oracle_cls, gpu_solver_cls, monkeypatch):
    """Landmark blocks shared with another factor and online-calibration factors take the host path even when the device
    path is forced; the result is the oracle's either way."""
    monkeypatch.setenv("BSGPU_FLATTEN", "device")
    pr = mixed_problem(9, n_state=4, n_lm=16, consistent=True)        # has online-calibration factors
    lm0 = int(pr.meta["landmarks"][0])
    A = synthetic.sqrt_information_upper(0.01 * np.eye(3))
    pr.add_factors(capi.F_ABS_VEC3, [[lm0]], [np.concatenate([pr.block(lm0), A.ravel()])])
    g, o = _both(pr, oracle_cls, gpu_solver_cls)
    _same_solve(g, o)
    pr2 = synthetic.vio_window(n_kf=6, n_lm=80, seed=42, track_min=2, track_max=5)   # plain reprojection + shared landmark
    lm1 = int(pr2.meta["lm_blocks"][3])
    pr2.add_factors(capi.F_ABS_VEC3, [[lm1]], [np.concatenate([pr2.block(lm1), A.ravel()])])
    g2, o2 = _both(pr2, oracle_cls, gpu_solver_cls)
    _same_solve(g2, o2, tol_x=1e-6)


def test_contexts_release_their_device_memory(gpu_solver_cls):
    """create / describe / solve / covariance / marginalise / destroy, many times: device memory in use does not creep (pooled
    buffers, pinned scalars, events and streams all go back)."""
    import ctypes
    import os
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("reads the device-wide free memory: meaningless next to other test processes (-n)")
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")      # the runtime libbsgpu itself is linked against (torch brings its own)

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    pr = synthetic.vio_window(n_kf=12, n_lm=300, seed=4)
    kf = pr.meta["kf_blocks"]

    def cycle():
        g = gpu_solver_cls(0)
        pr.load(g)
        g.solve()
        g.covariance(int(kf[3, 1]), int(kf[3, 1]))
        g.marginalize([int(b) for b in kf[0]], pr.size)
        pr.load(g)              # re-describe and re-flatten in the same context (pool reuse)
        g.solve()
        g.close()

    for _ in range(3):
        cycle()
    free0 = free_bytes()
    for _ in range(40):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 32 << 20, "device memory in use grew by %.1f MB over 40 context lifetimes" % ((free0 - free1) / 2**20)


def test_solve_batch_equals_lone_solves(gpu_solver_cls):
    """bsgpu_solve_batch: windows of different kinds (Schur, pose-only dense, PCG, inverse-depth) in one call give what each gives
    alone; per-window options; a failing window reports its own error and the others still finish."""
    cases = [synthetic.vio_window(n_kf=40, n_lm=900, seed=1), synthetic.lio_window(n_kf=60, n_rel=500, seed=2),
             synthetic.pose_graph(n_pose=2200, n_loop=3000, seed=3), synthetic.idp_window(n_kf=8, n_lm=60, seed=6),
             synthetic.vio_window(n_kf=12, n_lm=200, seed=5, cauchy_a=None)]
    def fresh():
        out = []
        for pr in cases:
            g = gpu_solver_cls(0); pr.load(g); out.append(g)
        return out
    alone = fresh()
    opts = []
    for i, g in enumerate(alone):
        o = g.options_default(); o.max_num_iterations = 4 + i
        opts.append(o)
    lone = [g.solve(o) for g, o in zip(alone, opts)]
    batch = fresh()
    sums = gpu_solver_cls.solve_batch(batch, opts)
    for g0, s0, g1, s1 in zip(alone, lone, batch, sums):
        assert s1.num_iterations == s0.num_iterations and s1.termination_type == s0.termination_type
        assert s1.linear_solver_used == s0.linear_solver_used
        assert abs(s1.final_cost - s0.final_cost) <= 1e-9 * abs(s0.final_cost)
        assert np.abs(g1.get_blocks() - g0.get_blocks()).max() < 1e-8
    # one shared options entry
    again = fresh()
    shared = again[0].options_default(); shared.max_num_iterations = 3
    for s in gpu_solver_cls.solve_batch(again, shared):
        assert s.num_iterations <= 3
    # a window the exact path must refuse (pose graph above the dense limit): its error, the others solved
    mixed = fresh()
    o_exact = mixed[0].options_default(); o_exact.linear_solver_type = capi.LINEAR_SCHUR_CHOLESKY; o_exact.max_num_iterations = 3
    with pytest.raises(capi.SolverError) as e:
        gpu_solver_cls.solve_batch(mixed, o_exact)
    assert "limit" in str(e.value)
    assert mixed[0].iterations() and mixed[1].iterations()
    with pytest.raises(capi.SolverError):
        gpu_solver_cls.solve_batch([mixed[0], mixed[0]], o_exact)


def test_solve_batch_one_set_of_launches_equals_lone_solves(gpu_solver_cls):
    """The windows bsgpu_solve_batch advances TOGETHER (csrc/bsgpu_batch.cpp: one launch per kernel of the LM step for all of them,
    blockIdx.y = window) — visual-inertial windows of different sizes, with and without a robust loss, different iteration budgets so
    that windows drop out at different rounds — against their lone solves: the same decisions, costs, radii and final values (every
    window's tables and partial sums are laid out as in its lone solve), and the call really took the batched path.  A pose graph on
    the dense path rides in the same launches (round 5; tests/test_gpu_batch_kinds.py has the other kinds)."""
    cases = [synthetic.vio_window(n_kf=20, n_lm=500, seed=11), synthetic.vio_window(n_kf=30, n_lm=2000, seed=12),
             synthetic.vio_window(n_kf=12, n_lm=200, seed=5, cauchy_a=None), synthetic.vio_window(n_kf=60, n_lm=3000, seed=13),
             synthetic.vio_window(n_kf=20, n_lm=500, seed=14), synthetic.pose_graph(n_pose=300, n_loop=400, seed=3),
             synthetic.vio_window(n_kf=25, n_lm=800, seed=15)]
    for b in cases[4].meta["lm_blocks"][::17]:     # some landmarks held constant: their factors take the kernels' tail paths (no elimination)
        cases[4].is_const[int(b)] = 1
    def fresh():
        out = []
        for pr in cases:
            g = gpu_solver_cls(0); pr.load(g); out.append(g)
        return out
    alone = fresh()
    opts = []
    for i, g in enumerate(alone):
        o = g.options_vio(); o.max_solver_time_in_seconds = 0.0; o.max_num_iterations = 3 + 2 * i
        opts.append(o)
    lone = [g.solve(o) for g, o in zip(alone, opts)]
    w0, r0 = gpu_solver_cls.batch_stats()
    batch = fresh()
    sums = gpu_solver_cls.solve_batch(batch, opts)
    w1, r1 = gpu_solver_cls.batch_stats()
    assert w1 - w0 == 7 and r1 - r0 >= 4      # all seven went through the batched launches (round 5: the pose graph on the dense path too)
    for g0, s0, g1, s1 in zip(alone, lone, batch, sums):
        assert s1.num_iterations == s0.num_iterations and s1.termination_type == s0.termination_type
        assert s1.num_successful_steps == s0.num_successful_steps and s1.num_unsuccessful_steps == s0.num_unsuccessful_steps
        assert s1.num_linear_solves == s0.num_linear_solves and s1.is_solution_usable == s0.is_solution_usable
        i0, i1 = g0.iterations(), g1.iterations()
        assert len(i0) == len(i1)
        for a, b in zip(i0, i1):
            assert a.step_is_successful == b.step_is_successful and a.step_is_valid == b.step_is_valid
            assert abs(a.cost - b.cost) <= 1e-9 * abs(a.cost)
            assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-6 * a.trust_region_radius
            assert abs(a.gradient_max_norm - b.gradient_max_norm) <= 1e-6 * (a.gradient_max_norm + 1e-300)   # (sums of FP64 atomics, cancelling near the optimum: two lone runs differ as much)
        assert abs(s1.final_cost - s0.final_cost) <= 1e-9 * abs(s0.final_cost)
        assert np.abs(g1.get_blocks() - g0.get_blocks()).max() < 1e-8
    # a second call on the same contexts from the same start (cached argument tables), and a lone solve after a batched one
    for g in batch: g.reset_values()
    again = gpu_solver_cls.solve_batch(batch, opts)
    for s0, s1 in zip(lone, again):
        assert s1.num_iterations == s0.num_iterations and abs(s1.final_cost - s0.final_cost) <= 1e-9 * abs(s0.final_cost)
    batch[1].reset_values()
    s_l = batch[1].solve(opts[1])
    assert s_l.num_iterations == lone[1].num_iterations and abs(s_l.final_cost - lone[1].final_cost) <= 1e-9 * abs(lone[1].final_cost)
    for g in batch: g.reset_values()
    third = gpu_solver_cls.solve_batch(batch, opts)       # (the lone solve swapped that window's buffers: the tables are rebuilt)
    for s0, s1 in zip(lone, third):
        assert s1.num_iterations == s0.num_iterations and abs(s1.final_cost - s0.final_cost) <= 1e-9 * abs(s0.final_cost)


def test_solve_batch_more_windows_than_one_round_holds(gpu_solver_cls):
    """70 small windows in one bsgpu_solve_batch call: the batched launches take at most 64 windows at a time (BatchDyn's lists), the
    rest follow in a second set — every window still ends where its lone solve ends."""
    cases = [synthetic.vio_window(n_kf=6 + (i % 5), n_lm=40 + 7 * (i % 9), seed=300 + i) for i in range(70)]
    def fresh():
        out = []
        for pr in cases:
            g = gpu_solver_cls(0); pr.load(g); out.append(g)
        return out
    alone = fresh()
    opt = alone[0].options_vio(); opt.max_solver_time_in_seconds = 0.0; opt.max_num_iterations = 8
    lone = [g.solve(opt) for g in alone]
    w0, _ = gpu_solver_cls.batch_stats()
    batch = fresh()
    sums = gpu_solver_cls.solve_batch(batch, opt)
    w1, _ = gpu_solver_cls.batch_stats()
    assert w1 - w0 == 70
    for g0, s0, g1, s1 in zip(alone, lone, batch, sums):
        assert s1.num_iterations == s0.num_iterations and s1.termination_type == s0.termination_type
        assert abs(s1.final_cost - s0.final_cost) <= 1e-9 * abs(s0.final_cost) + 1e-18
        assert np.abs(g1.get_blocks() - g0.get_blocks()).max() < 1e-8


def test_contexts_on_concurrent_host_threads(gpu_solver_cls):
    """One context per host thread, all on one device (the reference runs its local smoother, global mapper and submap
    refinement side by side, submap_refinement.cpp:35-115): create, load, finalize, solve, read back and destroy concurrently;
    every thread must get what a lone context gets for the same window — nothing in the library may be shared between
    contexts except the device."""
    import threading
    cases = [synthetic.vio_window(n_kf=40, n_lm=900, seed=1),            # Schur path, several tiles
             synthetic.lio_window(n_kf=60, n_rel=500, seed=2),           # pose-only dense path
             synthetic.pose_graph(n_pose=2200, n_loop=3000, seed=3),     # above the dense limit: PCG path
             mixed_problem(4, n_state=5, n_lm=30, consistent=True, with_losses=True, hold_first=True),
             synthetic.vio_window(n_kf=12, n_lm=200, seed=5, cauchy_a=None),
             synthetic.idp_window(n_kf=8, n_lm=60, seed=6)]

    def run(pr, rounds):
        out = []
        for _ in range(rounds):
            g = gpu_solver_cls(0)
            pr.load(g); g.finalize()
            opt = g.options_default(); opt.max_num_iterations = 5
            s = g.solve(opt)
            out.append((s.termination_type, [i.step_is_successful for i in g.iterations()], s.final_cost, g.get_blocks()))
            del g
        return out

    alone = [run(pr, 1)[0] for pr in cases]
    results, errors = [None] * len(cases), []

    def work(i):
        try:
            results[i] = run(cases[i], 4)
        except Exception as e:      # noqa: BLE001 — reported below, from the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for i, (term, acc, cost, x) in enumerate(alone):
        for term_c, acc_c, cost_c, x_c in results[i]:
            assert term_c == term and acc_c == acc
            # (run-to-run: FP64 atomics, and the PCG window stops its inner solves at a 1e-10 relative residual — DESIGN.md §4)
            assert abs(cost_c - cost) <= 1e-7 * max(abs(cost), 1e-30), (i, cost_c, cost)
            # (the unconverged, ill-conditioned PCG window alone repeats its values to ~1e-6 only, sequentially as well)
            assert np.abs(x_c - x).max() <= (1e-4 if i == 2 else 1e-6), i


def test_values_survive_a_change_of_the_problem_description(oracle_cls, gpu_solver_cls):
    """solve, add_factors, get_blocks / solve: "on return the best accepted point is the context's current value set" (bsgpu.h) also
    when the description changes afterwards — the next finalize() must start from the optimised point, not from the values that
    were handed over before the first solve (the host copy was never refreshed)."""
    pr = synthetic.vio_window(n_kf=6, n_lm=60, seed=4)
    g = gpu_solver_cls(0)
    pr.load(g)
    s1 = g.solve()
    x1 = g.get_blocks()
    assert s1.final_cost < s1.initial_cost and np.abs(x1 - pr.values).max() > 1e-6
    # a prior on a keyframe position, centred on the solved value, re-opens the description (finalized -> false)
    blk = int(pr.meta["kf_blocks"][2, 1])
    consts = np.concatenate([pr.block(blk, x1), np.eye(3).ravel()])[None, :]
    g.add_factors(capi.F_ABS_VEC3, np.array([[blk]], np.int32), consts)
    assert np.array_equal(g.get_blocks(), x1)                   # not finalized: the host copy — which must be the solved point
    s2 = g.solve()
    assert abs(s2.initial_cost - s1.final_cost) <= 1e-9 * s1.final_cost     # the prior is satisfied at x1: same cost, same point
    assert np.abs(g.get_blocks() - x1).max() < 1e-6
