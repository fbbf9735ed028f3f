"""pairs_band_kernel (csrc/k_band.hip): the landmarks of feature tracks — all observations within 13 consecutive camera poses, one per
camera pose — are taken through the reduced system as Z Z^T on the matrix cores; everything else keeps its pair entries (pairs_kernel).
Both forms against each other (BSGPU_PAIRS_BAND=0: every pair by entries) and against the oracle, on windows that mix them:
tracks with gaps, tracks longer than the band, two observations of a landmark from one camera pose, constant landmarks, held poses,
every flattening path (bs_optimizers/src/fixed_lag_smoother.cpp:281 is what either form serves)."""
import numpy as np
import pytest

from beam_slam_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def _drop_and_duplicate(pr, rng, drop=0.15, dup=12):
    """gaps in the tracks (observations dropped at random) and a few landmarks seen twice from one camera pose"""
    idx, consts, lk, la = pr.factors[capi.F_REPROJ][0]
    keep = rng.random(idx.shape[0]) >= drop
    idx, consts, lk, la = idx[keep], consts[keep], lk[keep], la[keep]
    pick = rng.choice(idx.shape[0], size=dup, replace=False)
    c2 = consts[pick].copy()
    c2[:, :2] += rng.normal(0, 1.0, (dup, 2)).round()
    pr.factors[capi.F_REPROJ] = [(np.concatenate([idx, idx[pick]]), np.concatenate([consts, c2]), np.concatenate([lk, lk[pick]]),
                                  np.concatenate([la, la[pick]]))]
    return pr


def _window(kind):
    rng = np.random.default_rng(77)
    if kind == "tracks":            # every landmark qualifies
        return synthetic.vio_window(n_kf=30, n_lm=900, seed=51, track_min=2, track_max=12)
    if kind == "long_tracks":       # spans of up to 20 key frames: the long ones keep their pair entries
        return synthetic.vio_window(n_kf=40, n_lm=900, seed=52, track_min=2, track_max=20)
    if kind == "gaps_and_stereo":
        return _drop_and_duplicate(synthetic.vio_window(n_kf=26, n_lm=700, seed=53, track_min=3, track_max=16), rng)
    if kind == "const_and_held":
        pr = synthetic.vio_window(n_kf=24, n_lm=600, seed=54, track_min=2, track_max=13)
        for b in pr.meta["lm_blocks"][::9]:
            pr.is_const[int(b)] = 1
        for b in pr.meta["kf_blocks"][5][:2]:     # a held pose in the middle of the tracks: its rows are not in the reduced system
            pr.is_const[int(b)] = 1
        return pr
    if kind == "no_imu":            # visual-only window: no pose-only factors ride in the launch
        return synthetic.vio_window(n_kf=16, n_lm=500, seed=55, with_imu=False)
    raise ValueError(kind)


def _run(pr, gpu_solver_cls, iters=6):
    g = gpu_solver_cls(0)
    pr.load(g)
    o = g.options_vio()
    o.max_num_iterations = iters
    o.max_solver_time_in_seconds = 0.0
    s = g.solve(o)
    return s, [(i.cost, i.gradient_max_norm, i.step_norm, i.trust_region_radius) for i in g.iterations()], g.get_blocks()


@pytest.mark.parametrize("kind", ["tracks", "long_tracks", "gaps_and_stereo", "const_and_held", "no_imu"])
@pytest.mark.parametrize("flatten", ["host", "device"])
def test_band_kernel_equals_pair_entries_and_oracle(oracle_cls, gpu_solver_cls, monkeypatch, kind, flatten):
    pr = _window(kind)
    monkeypatch.setenv("BSGPU_FLATTEN", flatten)
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "0")
    s0, it0, x0 = _run(pr, gpu_solver_cls)
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "1")      # (whatever the size: by itself the library takes the band form from 150 000 factors)
    s1, it1, x1 = _run(pr, gpu_solver_cls)
    assert s0.num_iterations == s1.num_iterations and s0.num_successful_steps == s1.num_successful_steps
    a, b = np.array(it0), np.array(it1)
    # the two forms add the same terms in another order: the trajectories agree to rounding (cost, gradient norm, step norm, radius)
    assert np.allclose(a[:, 0], b[:, 0], rtol=1e-10) and np.allclose(a[:, 3], b[:, 3], rtol=1e-12)
    assert np.allclose(a[:, 1], b[:, 1], rtol=1e-6, atol=1e-9) and np.allclose(a[:, 2], b[:, 2], rtol=1e-6, atol=1e-12)
    assert np.abs(x0 - x1).max() < 1e-8
    o = oracle_cls()
    pr.load(o)
    oo = o.options_vio()
    oo.max_num_iterations = 6
    oo.max_solver_time_in_seconds = 0.0
    so = o.solve(oo)
    assert abs(so.final_cost - s1.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(o.get_blocks() - x1).max() < 1e-6


def test_gradient_only_step_and_iteration_budget(oracle_cls, gpu_solver_cls, monkeypatch):
    """the last iteration of a budget is a gradient-only step: the band launch then forms the per-camera sums alone (no products)"""
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "1")
    pr = _window("tracks")
    for iters in (1, 2):
        s, it, x = _run(pr, gpu_solver_cls, iters=iters)
        o = oracle_cls()
        pr.load(o)
        oo = o.options_vio()
        oo.max_num_iterations = iters
        oo.max_solver_time_in_seconds = 0.0
        so = o.solve(oo)
        io = o.iterations()
        assert len(io) == len(it)
        assert np.allclose([i.cost for i in io], [i[0] for i in it], rtol=1e-10)
        assert np.allclose([i.gradient_max_norm for i in io], [i[1] for i in it], rtol=1e-7)


def test_band_units_of_several_parts(oracle_cls, gpu_solver_cls, monkeypatch):
    """BSGPU_BAND_PART: a first camera pose's landmarks cut into several units (what a window of many landmarks per key frame gets)"""
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "1")
    pr = synthetic.vio_window(n_kf=12, n_lm=1500, seed=56, track_min=2, track_max=9)
    s_ref, it_ref, x_ref = _run(pr, gpu_solver_cls)
    for part in ("16", "40"):
        monkeypatch.setenv("BSGPU_BAND_PART", part)
        s, it, x = _run(pr, gpu_solver_cls)
        assert np.allclose([i[0] for i in it], [i[0] for i in it_ref], rtol=1e-10) and np.abs(x - x_ref).max() < 1e-8


@pytest.mark.parametrize("flatten", ["host", "device"])
def test_no_c_rows(oracle_cls, gpu_solver_cls, monkeypatch, flatten):
    """A window whose landmarks are all band landmarks keeps no C rows (Visual::no_cr): the band kernel and the landmark back-substitution form
    C = B Linv^T and rho = r - C z themselves.  Against the same solve with the rows kept (BSGPU_NO_CR=0) and against the oracle — with robust losses,
    a held pose, a gradient-only last step, and a window of one more kind of factor (so that the pose-only riders are in the launches)."""
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "1")
    monkeypatch.setenv("BSGPU_FLATTEN", flatten)
    pr = synthetic.vio_window(n_kf=22, n_lm=1200, seed=59, track_min=2, track_max=12)
    for b in pr.meta["kf_blocks"][7][:2]:
        pr.is_const[int(b)] = 1
    monkeypatch.setenv("BSGPU_NO_CR", "0")
    s0, it0, x0 = _run(pr, gpu_solver_cls, iters=7)
    monkeypatch.delenv("BSGPU_NO_CR")
    s1, it1, x1 = _run(pr, gpu_solver_cls, iters=7)
    assert len(it0) == len(it1)
    assert np.allclose([i[0] for i in it0], [i[0] for i in it1], rtol=1e-11) and np.allclose([i[1] for i in it0], [i[1] for i in it1], rtol=1e-7)
    assert np.abs(x0 - x1).max() < 1e-7, np.abs(x0 - x1).max()   # (the two round C differently; weakly observed landmark depths carry it)
    o = oracle_cls(); pr.load(o)
    opt = o.options_default(); opt.max_num_iterations = 7
    g = gpu_solver_cls(0); pr.load(g)
    sg, so = g.solve(opt), o.solve(opt)
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost and np.abs(g.get_blocks() - o.get_blocks()).max() < 1e-7
    for a, b in zip(g.iterations(), o.iterations()):
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-9 * b.cost


def test_no_c_rows_lone_and_batched_steps_on_the_same_contexts(gpu_solver_cls, monkeypatch):
    """The lone launches of an all-band window keep no C rows, the batched launches do (they pass the rows' buffer): contexts that go through a lone solve,
    then a batched one, then a lone one again give what fresh contexts give for each — every step's landmark launch is the one its consumers expect."""
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "1")
    cases = [synthetic.vio_window(n_kf=16, n_lm=700, seed=60 + i, track_min=2, track_max=10) for i in range(3)]
    def fresh():
        out = []
        for pr in cases:
            g = gpu_solver_cls(0); pr.load(g); out.append(g)
        return out
    gs = fresh()
    opt = gs[0].options_vio(); opt.max_num_iterations = 4; opt.max_solver_time_in_seconds = 0.0
    ref_lone = [(g.solve(opt).final_cost, g.get_blocks()) for g in fresh()]
    ref_b = fresh()
    ref_batch = [(s.final_cost, g.get_blocks()) for s, g in zip(gpu_solver_cls.solve_batch(ref_b, opt), ref_b)]
    for (c0, x0), (c1, x1) in zip(ref_lone, ref_batch):
        assert abs(c0 - c1) <= 1e-10 * c0 and np.abs(x0 - x1).max() < 1e-7
    for rounds in range(2):
        for g, (c0, x0) in zip(gs, ref_lone):            # lone (no C rows)
            g.reset_values()
            s = g.solve(opt)
            assert abs(s.final_cost - c0) <= 1e-10 * c0 and np.abs(g.get_blocks() - x0).max() < 1e-7
        for g in gs: g.reset_values()
        sums = gpu_solver_cls.solve_batch(gs, opt)        # batched (C rows)
        for g, s, (c1, x1) in zip(gs, sums, ref_batch):
            assert abs(s.final_cost - c1) <= 1e-10 * c1 and np.abs(g.get_blocks() - x1).max() < 1e-7


def test_size_rule(gpu_solver_cls, monkeypatch):
    """by itself the library takes the band form from kBandMinFactors reprojection factors on: the same solve either way at a size above it"""
    monkeypatch.delenv("BSGPU_PAIRS_BAND", raising=False)
    pr = synthetic.vio_window(n_kf=60, n_lm=20000, seed=57)     # ~160 000 factors
    s1, it1, x1 = _run(pr, gpu_solver_cls, iters=4)
    monkeypatch.setenv("BSGPU_PAIRS_BAND", "0")
    s0, it0, x0 = _run(pr, gpu_solver_cls, iters=4)
    assert np.allclose([i[0] for i in it0], [i[0] for i in it1], rtol=1e-10) and np.abs(x0 - x1).max() < 1e-8
