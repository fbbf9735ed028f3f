"""Shared helpers for the tests: small random problems covering every factor type."""
import numpy as np

from beam_slam_amd import capi, synthetic
from beam_slam_amd.problem import Problem
from beam_slam_amd.synthetic import (quat_from_aa, quat_mul, quat_to_rot, rot_to_quat, so3_exp,
                                     sqrt_information_upper)


def rand_quat(rng, scale=1.0):
    return quat_from_aa(rng.normal(0, scale, 3))


def rand_spd(rng, n, lo=0.5, hi=2.0):
    M = rng.normal(0, 1, (n, n))
    Q, _ = np.linalg.qr(M)
    return Q @ np.diag(rng.uniform(lo, hi, n)) @ Q.T


def mixed_problem(seed=0, n_state=4, n_lm=12, with_losses=True, hold_first=False, consistent=False):
    """A small graph with every factor type of include/bsgpu.h.

    consistent=False: measurements are random (far from any optimum): evaluation / Jacobian checks at
    arbitrary points, and a wild LM path with rejected steps.
    consistent=True: the states form an IMU-consistent chain and every measurement is generated from
    them with small noise, so the problem is well conditioned and LM converges in a few iterations."""
    if consistent:
        return _consistent_problem(seed, n_state, n_lm, with_losses, hold_first)
    rng = np.random.default_rng(seed)
    pr = Problem()
    R_cb, t_cb = synthetic._t_cam_baselink()
    cam = pr.add_camera(synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, R_cb, t_cb)
    st = []
    q_true, p_true = [], []
    for i in range(n_state):
        q = rand_quat(rng, 0.3)
        p = rng.normal(0, 1.0, 3)
        q_true.append(q); p_true.append(p)
        b = [pr.add_quat(quat_mul(q, quat_from_aa(rng.normal(0, 0.02, 3))), const=(hold_first and i == 0)),
             pr.add_block(p + rng.normal(0, 0.05, 3), const=(hold_first and i == 0)),
             pr.add_block(rng.normal(0, 1, 3)), pr.add_block(rng.normal(0, 0.01, 3)), pr.add_block(rng.normal(0, 0.05, 3))]
        st.append(b)
    st = np.array(st, np.int32)
    # landmarks in front of the cameras
    lm = []
    obs_idx, obs_c = [], []
    for j in range(n_lm):
        k = int(rng.integers(0, n_state))
        R = quat_to_rot(q_true[k])
        Pc = np.array([rng.uniform(-1, 1), rng.uniform(-0.7, 0.7), rng.uniform(4, 9)])
        Pw = R @ (R_cb.T @ (Pc - t_cb)) + p_true[k]
        b = pr.add_block(Pw + rng.normal(0, 0.05, 3))
        lm.append(b)
        for kk in range(n_state):
            Rk = quat_to_rot(q_true[kk])
            Pck = R_cb @ (Rk.T @ (Pw - p_true[kk])) + t_cb
            if Pck[2] < 1.0:
                continue
            uv = np.array([synthetic.FX * Pck[0] / Pck[2] + synthetic.CX, synthetic.FY * Pck[1] / Pck[2] + synthetic.CY])
            obs_idx.append([st[kk, 0], st[kk, 1], b, cam])
            obs_c.append([*(uv + rng.normal(0, 1.0, 2)), rng.uniform(0.5, 2.0)])
    pr.add_factors(capi.F_REPROJ, obs_idx, obs_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 5.0)
    # online-calib reprojection with constant extrinsic blocks (T_BASELINK_CAM)
    T_bc = synthetic.T_IMU_CAM
    qe = pr.add_quat(rot_to_quat(T_bc[:3, :3]), const=True)
    pe = pr.add_block(T_bc[:3, 3], const=True)
    n_oc = max(2, len(obs_idx) // 4)
    sel = rng.choice(len(obs_idx), n_oc, replace=False)
    oc_idx = [[obs_idx[s][0], obs_idx[s][1], obs_idx[s][2], qe, pe, cam] for s in sel]
    oc_c = [[obs_c[s][0] + 0.3, obs_c[s][1] - 0.2, 1.0] for s in sel]
    pr.add_factors(capi.F_REPROJ_ONLINE_CALIB, oc_idx, oc_c, capi.LOSS_HUBER if with_losses else capi.LOSS_TRIVIAL, 1.5)
    # IMU delta factors with random (valid) pre-integration constants
    imu_idx, imu_c = [], []
    for i in range(n_state - 1):
        pre = synthetic.PreIntegrator()
        n = 11
        t = np.arange(n) * 0.01
        pre.integrate(t, rng.normal(0, 0.3, (n, 3)), rng.normal(0, 1.0, (n, 3)) + np.array([0, 0, 9.8]), t[-1],
                      np.zeros(3), np.zeros(3))
        imu_idx.append(np.concatenate([st[i], st[i + 1]]))
        imu_c.append(pre.pack(rng.normal(0, 0.005, 3), rng.normal(0, 0.02, 3), 0.7))
    pr.add_factors(capi.F_IMU_DELTA, imu_idx, imu_c)
    # IMU prior on state 0
    mean = np.concatenate([rand_quat(rng, 0.3), rng.normal(0, 1, 12)])
    A = sqrt_information_upper(rand_spd(rng, 15, 0.01, 0.1))
    pr.add_factors(capi.F_IMU_PRIOR, st[0][None, :], np.concatenate([mean, A.ravel()])[None, :])
    # relative pose with (constant) extrinsics and without
    pxe = pr.add_block(np.array([0.1, -0.05, 0.2]), const=True)
    qxe = pr.add_quat(rand_quat(rng, 0.2), const=True)
    re_idx, re_c, r_idx, r_c = [], [], [], []
    for i in range(n_state - 1):
        d = np.concatenate([rng.normal(0, 0.5, 3), rand_quat(rng, 0.2)])
        A6 = sqrt_information_upper(rand_spd(rng, 6, 0.01, 0.05))
        re_idx.append([st[i, 1], st[i, 0], st[i + 1, 1], st[i + 1, 0], pxe, qxe])
        re_c.append(np.concatenate([d, A6.ravel()]))
        d2 = np.concatenate([rng.normal(0, 0.5, 3), rand_quat(rng, 0.2)])
        r_idx.append([st[i, 1], st[i, 0], st[(i + 2) % n_state, 1], st[(i + 2) % n_state, 0]])
        r_c.append(np.concatenate([d2, sqrt_information_upper(rand_spd(rng, 6, 0.01, 0.05)).ravel()]))
    pr.add_factors(capi.F_RELPOSE_EXT, re_idx, re_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 1.0)
    pr.add_factors(capi.F_RELPOSE, r_idx, r_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 1.0)
    # absolute pose prior
    b = np.concatenate([rng.normal(0, 1, 3), rand_quat(rng, 0.3)])
    pr.add_factors(capi.F_ABSPOSE, [[st[1, 1], st[1, 0]]], [np.concatenate([b, sqrt_information_upper(rand_spd(rng, 6, 0.05, 0.2)).ravel()])])
    # 3-vector absolute / relative
    pr.add_factors(capi.F_ABS_VEC3, [[st[1, 2]], [st[2, 3]]],
                   [np.concatenate([rng.normal(0, 1, 3), sqrt_information_upper(rand_spd(rng, 3)).ravel()]) for _ in range(2)])
    pr.add_factors(capi.F_REL_VEC3, [[st[0, 2], st[1, 2]], [st[1, 4], st[2, 4]]],
                   [np.concatenate([rng.normal(0, 1, 3), sqrt_information_upper(rand_spd(rng, 3)).ravel()]) for _ in range(2)])
    # gravity alignment
    pr.add_factors(capi.F_GRAVITY, [[st[2, 0]]], [np.concatenate([[0.1, -0.2, -9.7], (10.0 * np.eye(2)).ravel()])])
    pr.meta = dict(states=st, landmarks=np.array(lm, np.int32))
    return pr


def _consistent_problem(seed, n_state, n_lm, with_losses, hold_first):
    rng = np.random.default_rng(seed)
    pr = Problem()
    R_cb, t_cb = synthetic._t_cam_baselink()
    cam = pr.add_camera(synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, R_cb, t_cb)
    g = synthetic.GRAVITY_WORLD
    # ---- IMU-consistent chain of true states (imu_preintegration.cpp:225-243 PredictState)
    q_t, p_t, v_t = [rand_quat(rng, 0.3)], [rng.normal(0, 1, 3)], [rng.normal(0, 0.5, 3)]
    bg_t, ba_t = rng.normal(0, 0.002, 3), rng.normal(0, 0.02, 3)
    pres = []
    for i in range(n_state - 1):
        pre = synthetic.PreIntegrator()
        n = 21
        t = np.arange(n) * 0.005
        pre.integrate(t, rng.normal(0, 0.3, (n, 3)), rng.normal(0, 0.5, (n, 3)) + np.array([0, 0, 9.8]), t[-1], np.zeros(3), np.zeros(3))
        R = quat_to_rot(q_t[-1])
        dt = pre.t
        # bias-corrected deltas at the true biases (first order, like the functor)
        dq = quat_mul(pre.q, np.array([1.0, *(0.5 * pre.dq_dbg @ bg_t)]))
        dp = pre.p + pre.dp_dbg @ bg_t + pre.dp_dba @ ba_t
        dv = pre.v + pre.dv_dbg @ bg_t + pre.dv_dba @ ba_t
        p_t.append(p_t[-1] + v_t[-1] * dt + 0.5 * g * dt * dt + R @ dp)
        v_t.append(v_t[-1] + g * dt + R @ dv)
        q = quat_mul(q_t[-1], dq / np.linalg.norm(dq))
        q_t.append(q / np.linalg.norm(q))
        pres.append(pre)
    st = []
    for i in range(n_state):
        c = hold_first and i == 0
        b = [pr.add_quat(q_t[i] if c else quat_mul(q_t[i], quat_from_aa(rng.normal(0, 0.01, 3))), const=c),
             pr.add_block(p_t[i] if c else p_t[i] + rng.normal(0, 0.02, 3), const=c),
             pr.add_block(v_t[i] + rng.normal(0, 0.02, 3)), pr.add_block(np.zeros(3)), pr.add_block(np.zeros(3))]
        st.append(b)
    st = np.array(st, np.int32)
    lm, obs_idx, obs_c = [], [], []
    for j in range(n_lm):
        k = int(rng.integers(0, n_state))
        R = quat_to_rot(q_t[k])
        Pc = np.array([rng.uniform(-1, 1), rng.uniform(-0.7, 0.7), rng.uniform(4, 9)])
        Pw = R @ (R_cb.T @ (Pc - t_cb)) + p_t[k]
        b = pr.add_block(Pw + rng.normal(0, 0.05, 3))
        lm.append(b)
        for kk in range(n_state):
            Rk = quat_to_rot(q_t[kk])
            Pck = R_cb @ (Rk.T @ (Pw - p_t[kk])) + t_cb
            if Pck[2] < 1.0:
                continue
            uv = np.array([synthetic.FX * Pck[0] / Pck[2] + synthetic.CX, synthetic.FY * Pck[1] / Pck[2] + synthetic.CY])
            obs_idx.append([st[kk, 0], st[kk, 1], b, cam])
            obs_c.append([*(uv + rng.normal(0, 1.0, 2)), 1.0])
    pr.add_factors(capi.F_REPROJ, obs_idx, obs_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 5.0)
    T_bc = np.linalg.inv(np.block([[R_cb, t_cb[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]))
    qe = pr.add_quat(rot_to_quat(T_bc[:3, :3]), const=True)
    pe = pr.add_block(T_bc[:3, 3], const=True)
    sel = rng.choice(len(obs_idx), max(2, len(obs_idx) // 4), replace=False)
    pr.add_factors(capi.F_REPROJ_ONLINE_CALIB, [[obs_idx[s][0], obs_idx[s][1], obs_idx[s][2], qe, pe, cam] for s in sel],
                   [[obs_c[s][0] + rng.normal(0, 0.5), obs_c[s][1] + rng.normal(0, 0.5), 1.0] for s in sel],
                   capi.LOSS_HUBER if with_losses else capi.LOSS_TRIVIAL, 1.5)
    pr.add_factors(capi.F_IMU_DELTA, [np.concatenate([st[i], st[i + 1]]) for i in range(n_state - 1)],
                   [pres[i].pack(np.zeros(3), np.zeros(3), 1.0) for i in range(n_state - 1)])
    mean = np.concatenate([q_t[0], p_t[0], v_t[0], bg_t, ba_t])
    pr.add_factors(capi.F_IMU_PRIOR, st[0][None, :], np.concatenate([mean, sqrt_information_upper(1e-4 * np.eye(15)).ravel()])[None, :])
    pxe = pr.add_block(np.array([0.1, -0.05, 0.2]), const=True)
    qx = rand_quat(rng, 0.2)
    qxe = pr.add_quat(qx, const=True)
    Rx = quat_to_rot(qx)
    A6 = sqrt_information_upper(np.diag([1e-4] * 3 + [1e-5] * 3))
    re_idx, re_c, r_idx, r_c = [], [], [], []
    for i in range(n_state - 1):
        j = i + 1
        R1, R2 = quat_to_rot(q_t[i]), quat_to_rot(q_t[j])
        R1s, R2s = R1 @ Rx, R2 @ Rx
        p1s, p2s = R1 @ np.array([0.1, -0.05, 0.2]) + p_t[i], R2 @ np.array([0.1, -0.05, 0.2]) + p_t[j]
        d = np.concatenate([R1s.T @ (p2s - p1s) + rng.normal(0, 0.01, 3), rot_to_quat(R1s.T @ R2s @ so3_exp(rng.normal(0, 0.003, 3)))])
        re_idx.append([st[i, 1], st[i, 0], st[j, 1], st[j, 0], pxe, qxe]); re_c.append(np.concatenate([d, A6.ravel()]))
        j2 = (i + 2) % n_state
        R3 = quat_to_rot(q_t[j2])
        d2 = np.concatenate([R1.T @ (p_t[j2] - p_t[i]) + rng.normal(0, 0.01, 3), rot_to_quat(R1.T @ R3 @ so3_exp(rng.normal(0, 0.003, 3)))])
        r_idx.append([st[i, 1], st[i, 0], st[j2, 1], st[j2, 0]]); r_c.append(np.concatenate([d2, A6.ravel()]))
    pr.add_factors(capi.F_RELPOSE_EXT, re_idx, re_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 1.0)
    pr.add_factors(capi.F_RELPOSE, r_idx, r_c, capi.LOSS_CAUCHY if with_losses else capi.LOSS_TRIVIAL, 1.0)
    pr.add_factors(capi.F_ABSPOSE, [[st[1, 1], st[1, 0]]],
                   [np.concatenate([p_t[1] + rng.normal(0, 0.01, 3), quat_mul(q_t[1], quat_from_aa(rng.normal(0, 0.003, 3))), A6.ravel()])])
    I3 = sqrt_information_upper(1e-2 * np.eye(3)).ravel()
    pr.add_factors(capi.F_ABS_VEC3, [[st[1, 2]], [st[2, 3]]], [np.concatenate([v_t[1], I3]), np.concatenate([bg_t, I3])])
    pr.add_factors(capi.F_REL_VEC3, [[st[0, 2], st[1, 2]], [st[1, 4], st[2, 4]]],
                   [np.concatenate([v_t[1] - v_t[0], I3]), np.concatenate([np.zeros(3), I3])])
    g_b = quat_to_rot(q_t[2]).T @ np.array([0, 0, -9.80665])
    pr.add_factors(capi.F_GRAVITY, [[st[2, 0]]], [np.concatenate([g_b, (10.0 * np.eye(2)).ravel()])])
    pr.meta = dict(states=st, landmarks=np.array(lm, np.int32))
    return pr


def manifold_plus(pr, values, delta, toff_of):
    """x (+) delta over all non-constant blocks, numpy restatement for finite differences."""
    out = values.copy()
    for b in range(pr.n_blocks):
        to = toff_of(b)
        if to < 0:
            continue
        o, s = pr.offset[b], pr.size[b]
        if pr.manifold[b] == capi.MANIFOLD_QUAT_RIGHT:
            out[o:o + 4] = quat_mul(values[o:o + 4], quat_from_aa(delta[to:to + 3]))
        else:
            out[o:o + s] = values[o:o + s] + delta[to:to + s]
    return out
