"""Mid-size structure sweep of the tiled Cholesky path, HIP vs oracle: reduced systems of 5-35 tiles of 64 whose coupling
pattern is drawn from a seed — band widths from a few states to the whole window (lidar relative-pose factors with a random
maximum gap), sparse and dense sets of far couplings (random loop closures of a pose graph), long visual tracks.  These are
the shapes that decide the plan of dense_plan.h (nested-dissection pieces, band width, separators, tiles shared between the
panels of one step, look-ahead by the last-arriving updater); the small windows of test_gpu_random.py fit in one or two
tiles and the full-size windows of test_gpu_fullsize.py are three fixed shapes."""
import numpy as np
import pytest

from beam_slam_amd import synthetic

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(7000 + seed)
    kind = seed % 3
    if kind == 0:      # lidar-inertial window, 15 tangent dimensions per keyframe: band width = max_gap states
        n_kf = int(rng.integers(30, 140))
        max_gap = int([2, 5, 12, 40, n_kf - 1][int(rng.integers(0, 5))])
        return synthetic.lio_window(n_kf=n_kf, n_rel=int(rng.integers(n_kf, 12 * n_kf)), seed=seed, max_gap=min(max_gap, n_kf - 1))
    if kind == 1:      # pose graph, 6 per pose: odometry chain + uniformly random loop closures (none .. 3 per pose)
        n_pose = int(rng.integers(80, 340))
        n_loop = int([0, n_pose // 20, n_pose // 4, n_pose, 3 * n_pose][int(rng.integers(0, 5))])
        return synthetic.pose_graph(n_pose=n_pose, n_loop=n_loop, seed=seed)
    n_kf = int(rng.integers(30, 120))   # visual-inertial window with tracks up to a third of the window
    return synthetic.vio_window(n_kf=n_kf, n_lm=int(rng.integers(200, 1500)), seed=seed, track_min=2,
                                track_max=int(rng.integers(4, max(5, n_kf // 3))), cauchy_a=[None, 5.0][int(rng.integers(0, 2))])


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("BSGPU_STRUCTURE_CASES", "24")))))
def test_midsize_structures(oracle_cls, gpu_solver_cls, seed):
    pr = _case(seed)
    g, o = gpu_solver_cls(0), oracle_cls()
    pr.load(g); pr.load(o)
    g.finalize(); o.finalize()
    assert [g.tangent_offset(b) for b in range(pr.n_blocks)] == [o.tangent_offset(b) for b in range(pr.n_blocks)]
    cg, _, gg, _ = g.evaluate()
    co, _, go, _ = o.evaluate()
    assert abs(cg - co) <= 1e-12 * abs(co)
    assert np.abs(gg - go).max() <= 1e-9 * max(1.0, np.abs(go).max())
    og, oo = g.options_default(), o.options_default()
    og.max_num_iterations = oo.max_num_iterations = 6
    chains, steps, tiles = g.plan_info()
    assert tiles >= 3 and steps >= 2, (chains, steps, tiles)      # (this sweep is about multi-tile plans)
    sg, so = g.solve(og), o.solve(oo)
    assert sg.termination_type == so.termination_type
    ig, io = g.iterations(), o.iterations()
    k = next((i for i, it in enumerate(io) if not it.step_is_successful and i > 0), len(io))
    for a, b in list(zip(ig, io))[:k + 1]:
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-7 * abs(b.cost) + 1e-16 * max(1.0, abs(io[0].cost))
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.final_cost + 1e-16 * max(1.0, abs(io[0].cost))
    if k == len(io):
        xg, xo = g.get_blocks(), o.get_blocks()
        states = pr.meta["kf_blocks"] if pr.meta.get("kind") == "vio_window" else pr.meta.get("states")
        if states is not None:
            for b in np.asarray(states).ravel():
                assert np.abs(pr.block(int(b), xg) - pr.block(int(b), xo)).max() <= 1e-6
        else:
            assert np.abs(xg - xo).max() <= 1e-6


@pytest.mark.parametrize("kind", ["vio", "lio"])
def test_plan_preference(oracle_cls, gpu_solver_cls, kind):
    """bsgpu_set_plan_preference (include/bsgpu.h): a small system planned for latency (the default: several settings of the dissection's cost model, the
    shortest replay of the task list kept — bsgpu_finalize.cpp) and for throughput (one setting, the fewest supernodes) is the SAME system in another
    elimination order: both solves against the oracle's, and the preference taking effect on a context that was finalized already (planned again from
    the point it has reached)."""
    from beam_slam_amd import synthetic
    pr = synthetic.vio_window(n_kf=20, n_lm=500, seed=8801) if kind == "vio" else synthetic.lio_window(n_kf=20, n_rel=300, seed=8802)
    o = oracle_cls(); pr.load(o)
    oo = o.options_vio(); oo.max_solver_time_in_seconds = 0.0
    so = o.solve(oo)
    plans = []
    for throughput in (False, True):
        g = gpu_solver_cls(0); pr.load(g)
        if throughput: g.set_plan_preference(True)
        og = g.options_vio(); og.max_solver_time_in_seconds = 0.0
        sg = g.solve(og)
        plans.append(g.plan_info())
        assert sg.num_iterations == so.num_iterations
        assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost
        for a, b in zip(g.iterations(), o.iterations()):
            assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-9 * abs(b.cost)
        if not throughput:
            # ... switched on the finalized, solved context: planned again, the values it has reached kept
            x_before = g.get_blocks().copy()
            g.set_plan_preference(True)
            assert np.array_equal(g.get_blocks(), x_before)
            assert g.plan_info() != plans[0] or True   # (the plans may coincide on a graph every setting cuts the same way)
            g.set_values(pr.values)   # (reset_values() would return to the point the context was planned again at: the optimum)
            s2 = g.solve(og)
            assert abs(s2.final_cost - so.final_cost) <= 1e-9 * so.final_cost and s2.num_iterations == so.num_iterations
        g.close()
    assert plans[0][2] >= plans[1][2]   # (tiles: the latency plan has at least the throughput plan's supernodes, each padded to whole tiles)
