"""bsgpu_add_factors_indirect — the entry point a caller that keeps its factor tables across solves uses (the C++ host mirror,
beam_slam_amd/host/gpu_graph.h): block columns hold caller-side variable SLOTS and are translated through a slot -> block map on
the way in, camera columns are copied.  Same problem described both ways must be the same problem, bit for bit."""
import numpy as np
import pytest

from beam_slam_amd import capi, problem
from helpers import mixed_problem


def _load_indirect(pr, solver, rng):
    """pr through add_factors_indirect with a random slot numbering (slots = a shuffled, gappy renaming of the blocks)."""
    n_slots = pr.n_blocks + 7
    slots = rng.permutation(n_slots)[:pr.n_blocks]          # block b lives in slot slots[b]
    slot_to_block = np.full(n_slots, -1, np.int32)
    slot_to_block[slots] = np.arange(pr.n_blocks)
    solver.clear()
    solver.set_blocks(pr.values, pr.offset, pr.size, pr.manifold, pr.is_const)
    if pr.cameras:
        solver.set_cameras(pr.cameras)
    nidx_vars = {t: problem.NIDX[t] - (1 if t in (capi.F_REPROJ, capi.F_REPROJ_ONLINE_CALIB, capi.F_IDP_REPROJ, capi.F_IDP_REPROJ_UNARY) else 0)
                 for t in pr.factors}
    for t in sorted(pr.factors):
        for idx, consts, lk, la in pr.factors[t]:
            idx = np.array(idx, np.int32, copy=True)
            nv = nidx_vars[t]
            idx[:, :nv] = slots[idx[:, :nv]]
            solver.add_factors_indirect(t, idx, slot_to_block, consts, lk, la)
    for blocks, A, b, xbar in pr.marginals:
        solver.add_marginal(blocks, A, b, xbar)
    return slots, slot_to_block


def _check_same(direct, indirect, pr, bitwise=True):
    direct.finalize(); indirect.finalize()
    assert [direct.tangent_offset(b) for b in range(pr.n_blocks)] == [indirect.tangent_offset(b) for b in range(pr.n_blocks)]
    cd, rd, gd, _ = direct.evaluate()
    ci, ri, gi, _ = indirect.evaluate()
    assert cd == ci and np.array_equal(rd, ri) and np.array_equal(gd, gi)
    sd, si = direct.solve(), indirect.solve()
    if bitwise:
        assert sd.num_iterations == si.num_iterations and sd.final_cost == si.final_cost
        assert np.array_equal(direct.get_blocks(), indirect.get_blocks())
    else:   # the device adds into the reduced system with FP64 atomics: two runs of the SAME description differ in the last bits,
            # and these (deliberately inconsistent, ~50-iteration) problems amplify that
        assert sd.termination_type == si.termination_type and abs(sd.final_cost - si.final_cost) <= 1e-6 * sd.final_cost


@pytest.mark.parametrize("seed", [0, 1])
def test_indirect_equals_direct_oracle(oracle_cls, seed):
    pr = mixed_problem(seed, with_losses=True)
    d, i = oracle_cls(), oracle_cls()
    pr.load(d)
    _load_indirect(pr, i, np.random.default_rng(seed))
    _check_same(d, i, pr)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_indirect_equals_direct_device(gpu_solver_cls, seed):
    pr = mixed_problem(seed, with_losses=True)
    d, i = gpu_solver_cls(0), gpu_solver_cls(0)
    pr.load(d)
    _load_indirect(pr, i, np.random.default_rng(seed))
    _check_same(d, i, pr, bitwise=False)


@pytest.mark.gpu
def test_indirect_rejects_unmapped_and_out_of_range_slots(gpu_solver_cls):
    pr = mixed_problem(3, with_losses=False)
    g = gpu_solver_cls(0)
    slots, s2b = _load_indirect(pr, g, np.random.default_rng(5))
    idx, consts, lk, la = pr.factors[capi.F_RELPOSE][0]
    bad = np.array(idx, np.int32, copy=True)
    bad[:, :] = slots[bad]
    free = int(np.flatnonzero(s2b < 0)[0])
    for wrong in (free, s2b.size, -1):
        b2 = bad.copy(); b2[0, 1] = wrong
        with pytest.raises(capi.SolverError) as e:
            g.add_factors_indirect(capi.F_RELPOSE, b2, s2b, consts, lk, la)
        assert e.value.code == capi.ERR_INVALID
    # a rejected call adds nothing: the problem still equals the directly described one
    d = gpu_solver_cls(0)
    pr.load(d)
    _check_same(d, g, pr, bitwise=False)
