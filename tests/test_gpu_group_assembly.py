"""Same-slot groups of pose-only factors (k_small.hip small_assemble_group: J^T J of a group on the matrix cores), HIP vs oracle:
lidar-inertial windows whose relative-pose factors fall into groups of every shape the kernel distinguishes — constant extrinsics
(13 columns, one 16 x 16 tile), estimated extrinsics (19 columns: the three lower tiles of a 2 x 2 grid), pairs of key frames with more
factors than one record holds (a group cut into pieces of equal size), very few factors per pair (groups below the minimum stay with
the segments), a held key frame (zero columns inside a group)."""
import numpy as np
import pytest

from beam_slam_amd import synthetic

pytestmark = pytest.mark.gpu

CASES = [
    dict(n_kf=30, n_rel=3000, max_gap=3, free_extrinsics=False),    # ~35 factors per pair: records of 17 - 24
    dict(n_kf=30, n_rel=3000, max_gap=3, free_extrinsics=True),     # ... with the extrinsics estimated: three tiles
    dict(n_kf=40, n_rel=6000, max_gap=10, free_extrinsics=True),    # ~15 per pair
    dict(n_kf=12, n_rel=2500, max_gap=1, free_extrinsics=True),     # ~230 per pair: ten records per pair
    dict(n_kf=60, n_rel=900, max_gap=12, free_extrinsics=False),    # one or two per pair: below the group minimum
    dict(n_kf=25, n_rel=1800, max_gap=4, free_extrinsics=True, hold=7),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_group_assembly_matches_oracle(oracle_cls, gpu_solver_cls, case):
    kw = dict(CASES[case])
    hold = kw.pop("hold", None)
    pr = synthetic.lio_window(seed=8100 + case, **kw)
    if hold is not None:
        for b in np.asarray(pr.meta["kf_blocks"])[hold].ravel():
            pr.is_const[int(b)] = 1
    g, o = gpu_solver_cls(0), oracle_cls()
    pr.load(g); pr.load(o)
    g.finalize(); o.finalize()
    cg, _, gg, _ = g.evaluate()
    co, _, go, _ = o.evaluate()
    assert abs(cg - co) <= 1e-12 * abs(co)
    assert np.abs(gg - go).max() <= 1e-9 * max(1.0, np.abs(go).max())      # J^T r of every group
    og, oo = g.options_default(), o.options_default()
    og.max_num_iterations = oo.max_num_iterations = 6
    sg, so = g.solve(og), o.solve(oo)
    assert sg.termination_type == so.termination_type
    ig, io = g.iterations(), o.iterations()
    k = next((i for i, it in enumerate(io) if not it.step_is_successful and i > 0), len(io))
    for a, b in list(zip(ig, io))[:k + 1]:                                   # J^T J: the steps it gives
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-7 * abs(b.cost) + 1e-16 * max(1.0, abs(io[0].cost))
    assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.final_cost + 1e-16 * max(1.0, abs(io[0].cost))
