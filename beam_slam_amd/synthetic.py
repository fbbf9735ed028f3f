"""Deterministic synthetic workloads C1..C5 of BASELINE.json as flat IR.

Workload generation only (bench.py and tests use it to feed identical inputs to the HIP path and
to the checker).  Includes a numpy restatement of the reference's IMU pre-integrator
(bs_common/src/bs_common/preintegrator.cpp:26-144), which in the reference also runs on the host
when a factor is created (bs_models/src/lib/imu/imu_preintegration.cpp:245-318): it produces the
constants of the IMU factor, it is not on the solve path.
"""
import numpy as np

from . import capi
from .problem import Problem

GRAVITY_WORLD = np.array([0.0, 0.0, -9.80665])  # bs_common/include/bs_common/utils.h:20-24

# bs_models/tests/reprojection_test.cpp:21-25
T_IMU_CAM = np.array([
    [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
    [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
    [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
    [0.0, 0.0, 0.0, 1.0]])
# bs_models/tests/data/intrinsics.json:5-12 (distortion ignored: factors use rectified pixels)
FX, FY, CX, CY, IMG_W, IMG_H = 458.654, 457.296, 367.215, 248.375, 752, 480
# beam_slam_launch/calibrations/ig2/imu.json:5-8
COV_GYRO_NOISE, COV_ACCEL_NOISE, COV_GYRO_BIAS, COV_ACCEL_BIAS = 5.7e-4, 9.4e-4, 3.7e-6, 2.4e-6


# ---------------------------------------------------------------------------------------------
# small SO(3) toolbox (quaternions are (w, x, y, z) like everywhere in the reference)
# ---------------------------------------------------------------------------------------------
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def so3_exp(w):  # [EXT] beam::LieAlgebraToR
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def so3_log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    return th / (2 * np.sin(th)) * v


def so3_right_jacobian(w):  # [EXT] beam::RightJacobianOfSO3
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-8:
        return np.eye(3) - 0.5 * K + K @ K / 6.0
    return np.eye(3) - (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K


def quat_from_aa(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        q = np.array([1.0, *(0.5 * w)])
        return q / np.linalg.norm(q)
    return np.array([np.cos(th / 2), *(np.sin(th / 2) / th * w)])


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quats_to_rots(q):
    """(n,4) -> (n,3,3)"""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def sqrt_information_upper(cov):
    """cov.inverse().llt().matrixU() — U upper with U^T U = cov^-1
    (absolute_imu_state_3d_stamped_constraint.cpp:22, preintegrator.cpp:135-138)."""
    return np.linalg.cholesky(np.linalg.inv(cov)).T.copy()


# ---------------------------------------------------------------------------------------------
# A9: PreIntegrator restated (bs_common/src/bs_common/preintegrator.cpp)
# ---------------------------------------------------------------------------------------------
class PreIntegrator:
    ES_Q, ES_P, ES_V, ES_BG, ES_BA = 0, 3, 6, 9, 12  # preintegrator.h:10-17

    def __init__(self, cov_w=COV_GYRO_NOISE, cov_a=COV_ACCEL_NOISE, cov_bg=COV_GYRO_BIAS, cov_ba=COV_ACCEL_BIAS):
        self.cov_w, self.cov_a = np.eye(3) * cov_w, np.eye(3) * cov_a
        self.cov_bg, self.cov_ba = np.eye(3) * cov_bg, np.eye(3) * cov_ba
        self.cov_tol, self.bias_cov_tol, self.invalid_inv_cov_weight = 1e-5, 1e-9, 1e-4  # preintegrator.h:129-143
        self.reset()

    def reset(self):  # :7-20
        self.t = 0.0
        self.q = np.array([1.0, 0, 0, 0])
        self.p, self.v = np.zeros(3), np.zeros(3)
        self.cov = np.zeros((15, 15))
        self.sqrt_inv_cov = np.zeros((15, 15))
        self.dq_dbg, self.dp_dbg, self.dp_dba = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3))
        self.dv_dbg, self.dv_dba = np.zeros((3, 3)), np.zeros((3, 3))

    def increment(self, dt, w_m, a_m, bg, ba):  # :26-89
        w, a = w_m - bg, a_m - ba
        R_full = so3_exp(w * dt)
        Rdq = quat_to_rot(self.q)
        Sa = skew(a)
        A = np.eye(9)
        A[0:3, 0:3] = R_full.T
        A[6:9, 0:3] = -dt * Rdq @ Sa
        A[3:6, 0:3] = -0.5 * dt * dt * Rdq @ Sa
        A[3:6, 6:9] = dt * np.eye(3)
        B = np.zeros((9, 6))
        Jr = so3_right_jacobian(w * dt)
        B[0:3, 0:3] = dt * Jr
        B[6:9, 3:6] = dt * Rdq
        B[3:6, 3:6] = 0.5 * dt * dt * Rdq
        inv_dt = 1.0 / max(dt, 1e-7)
        Q = np.zeros((6, 6))
        Q[0:3, 0:3] = self.cov_w * inv_dt
        Q[3:6, 3:6] = self.cov_a * inv_dt
        self.cov[0:9, 0:9] = A @ self.cov[0:9, 0:9] @ A.T + B @ Q @ B.T
        self.cov[9:12, 9:12] += self.cov_bg * dt
        self.cov[12:15, 12:15] += self.cov_ba * dt
        # bias jacobians (:69-80; order matters)
        self.dp_dbg = self.dp_dbg + dt * self.dv_dbg - 0.5 * dt * dt * Rdq @ Sa @ self.dq_dbg
        self.dp_dba = self.dp_dba + dt * self.dv_dba - 0.5 * dt * dt * Rdq
        self.dv_dbg = self.dv_dbg - dt * Rdq @ Sa @ self.dq_dbg
        self.dv_dba = self.dv_dba - dt * Rdq
        self.dq_dbg = R_full.T @ self.dq_dbg - dt * Jr
        # state (:82-88)
        q_half = quat_from_aa(0.5 * w * dt)
        a_mid = quat_to_rot(quat_mul(self.q, q_half)) @ a
        self.t += dt
        self.p = self.p + dt * self.v + 0.5 * dt * dt * a_mid
        self.v = self.v + dt * a_mid
        q = quat_mul(self.q, quat_from_aa(w * dt))
        self.q = q / np.linalg.norm(q)

    def integrate(self, times, w_m, a_m, t_end, bg, ba):  # :91-115
        self.reset()
        for i in range(len(times) - 1):
            if times[i + 1] > t_end + 1e-12:
                break
            self.increment(times[i + 1] - times[i], w_m[i], a_m[i], bg, ba)
        dt = t_end - times[-1]
        if dt > 1e-12:
            self.increment(dt, w_m[-1], a_m[-1], bg, ba)
        self.compute_sqrt_inv_cov()

    def compute_sqrt_inv_cov(self):  # :117-143
        if np.linalg.norm(self.cov[0:9, 0:9]) < self.cov_tol:
            self.cov[0:9, 0:9] = np.eye(9) * self.cov_tol
        if np.linalg.norm(self.cov[9:15, 9:15]) < self.bias_cov_tol:
            self.cov[9:15, 9:15] = np.eye(6) * self.bias_cov_tol
        try:
            U = sqrt_information_upper(self.cov)
            ok = np.all(np.isfinite(U))
        except np.linalg.LinAlgError:
            ok = False
        self.sqrt_inv_cov = U if ok else np.eye(15) * self.invalid_inv_cov_weight

    def pack(self, bg_lin, ba_lin, info_weight=1.0):
        """consts of BSGPU_F_IMU_DELTA (include/bsgpu.h)."""
        return np.concatenate([[self.t], self.q, self.p, self.v, self.dq_dbg.ravel(), self.dp_dbg.ravel(),
                               self.dp_dba.ravel(), self.dv_dbg.ravel(), self.dv_dba.ravel(), bg_lin, ba_lin,
                               (info_weight * self.sqrt_inv_cov).ravel()])


# ---------------------------------------------------------------------------------------------
# trajectory
# ---------------------------------------------------------------------------------------------
class Lissajous:
    """Smooth 3-D Lissajous (radius 5 m, +-0.5 m in z), body x forward."""

    def __init__(self, duration):
        # one figure per 20 s (the C2 window); shorter windows ride an arc of the same curve
        self.a = 2 * np.pi / max(duration, 20.0)
        self.phi = 0.3

    def pos(self, t):
        a = self.a
        return np.array([5 * np.sin(a * t), 5 * np.sin(2 * a * t + self.phi), 0.5 * np.sin(3 * a * t)])

    def vel(self, t):
        a = self.a
        return np.array([5 * a * np.cos(a * t), 10 * a * np.cos(2 * a * t + self.phi), 1.5 * a * np.cos(3 * a * t)])

    def acc(self, t):
        a = self.a
        return np.array([-5 * a * a * np.sin(a * t), -20 * a * a * np.sin(2 * a * t + self.phi),
                         -4.5 * a * a * np.sin(3 * a * t)])

    def rot(self, t):
        x = self.vel(t)
        x = x / np.linalg.norm(x)
        up = np.array([0.0, 0, 1])
        z = up - (up @ x) * x
        z = z / np.linalg.norm(z)
        y = np.cross(z, x)
        return np.stack([x, y, z], axis=1)

    def omega_body(self, t, h=1e-4):
        return so3_log(self.rot(t - h).T @ self.rot(t + h)) / (2 * h)


class RandomWalkLoop:
    """3-D random-walk pose chain (C4)."""

    def __init__(self, n, rng):
        self.R = [np.eye(3)]
        self.p = [np.zeros(3)]
        for _ in range(n - 1):
            dR = so3_exp(rng.normal(0, 0.1, 3))
            dp = np.array([1.0, 0, 0]) + rng.normal(0, 0.1, 3)
            self.p.append(self.p[-1] + self.R[-1] @ dp)
            self.R.append(self.R[-1] @ dR)


def _t_cam_baselink():
    T = np.linalg.inv(T_IMU_CAM)
    return T[:3, :3].copy(), T[:3, 3].copy()


def _perturb_quat(q, rng, sigma):
    return quat_mul(q, quat_from_aa(rng.normal(0, sigma, 3)))


# ---------------------------------------------------------------------------------------------
# visual-inertial window (C1, C2, C5 instances)
# ---------------------------------------------------------------------------------------------
def vio_window(n_kf=200, n_lm=50000, seed=20250620, kf_rate=10.0, imu_rate=200.0, track_min=4, track_max=12,
               pixel_sigma=1.0, w_reproj=1.0, cauchy_a=5.0, w_inertial=1.0, with_imu=True, sigma_rot=0.02,
               sigma_pos=0.05, sigma_vel=0.05, sigma_lm=0.1, kf0=0, first_prior=True, bias_seed=None):
    """SURVEY.md §8d "C2 synthetic input".  Returns a Problem; p.meta holds ground truth.
    kf0 / first_prior / bias_seed: the window starts at key frame kf0 of the (periodic) trajectory, has no prior on its first state,
    and takes the IMU biases of another seed — neighbouring submaps of one trajectory that share a boundary key frame (chain_windows)."""
    rng = np.random.default_rng(seed)
    dt_kf = 1.0 / kf_rate
    traj = Lissajous(duration=n_kf * dt_kf)
    t_kf = (kf0 + np.arange(n_kf)) * dt_kf
    R_true = np.stack([traj.rot(t) for t in t_kf])
    q_true = np.stack([rot_to_quat(R) for R in R_true])
    p_true = np.stack([traj.pos(t) for t in t_kf])
    v_true = np.stack([traj.vel(t) for t in t_kf])
    rb = rng if bias_seed is None else np.random.default_rng(bias_seed)
    bg_true = rb.normal(0, 0.002, 3)
    ba_true = rb.normal(0, 0.02, 3)
    R_cb, t_cb = _t_cam_baselink()

    # ---- landmarks + observations -----------------------------------------------------------
    track_max = min(track_max, n_kf)
    track_min = min(track_min, track_max)
    L = rng.integers(track_min, track_max + 1, n_lm)
    k0 = (rng.random(n_lm) * (n_kf - L + 1)).astype(np.int64)
    u0 = rng.random(n_lm) * IMG_W
    v0 = rng.random(n_lm) * IMG_H
    depth = rng.uniform(3.0, 15.0, n_lm)
    Pc = np.stack([(u0 - CX) / FX * depth, (v0 - CY) / FY * depth, depth], axis=1)
    Pb = (Pc - t_cb) @ R_cb  # R_cb^T (Pc - t_cb)
    P_true = np.einsum('nij,nj->ni', R_true[k0], Pb) + p_true[k0]
    obs_l = np.repeat(np.arange(n_lm), L)
    obs_k = (np.repeat(k0, L) + (np.arange(L.sum()) - np.repeat(np.cumsum(L) - L, L))).astype(np.int64)
    Pb_o = np.einsum('nji,nj->ni', R_true[obs_k], P_true[obs_l] - p_true[obs_k])
    Pc_o = Pb_o @ R_cb.T + t_cb
    keep = Pc_o[:, 2] > 0.5
    obs_l, obs_k, Pc_o = obs_l[keep], obs_k[keep], Pc_o[keep]
    uv = np.stack([FX * Pc_o[:, 0] / Pc_o[:, 2] + CX, FY * Pc_o[:, 1] / Pc_o[:, 2] + CY], axis=1)
    uv = np.rint(uv + rng.normal(0, pixel_sigma, uv.shape))  # rounded to int: visual_map.cpp:188-192

    # ---- IMU ---------------------------------------------------------------------------------
    dt_imu = 1.0 / imu_rate
    per = int(round(imu_rate / kf_rate))
    n_imu = (n_kf - 1) * per + 1
    imu_fac = []
    if with_imu and n_kf > 1:
        t_imu = t_kf[0] + np.arange(n_imu) * dt_imu
        w_m = np.empty((n_imu, 3))
        a_m = np.empty((n_imu, 3))
        sg, sa = np.sqrt(COV_GYRO_NOISE / dt_imu), np.sqrt(COV_ACCEL_NOISE / dt_imu)
        for i, t in enumerate(t_imu):
            R = traj.rot(t)
            w_m[i] = traj.omega_body(t) + bg_true + rng.normal(0, sg, 3)
            a_m[i] = R.T @ (traj.acc(t) - GRAVITY_WORLD) + ba_true + rng.normal(0, sa, 3)
        pre = PreIntegrator()
        zero = np.zeros(3)
        for i in range(n_kf - 1):
            s = slice(i * per, (i + 1) * per + 1)
            pre.integrate(t_imu[s], w_m[s], a_m[s], t_kf[i + 1], zero, zero)
            imu_fac.append(pre.pack(zero, zero, w_inertial))

    # ---- IR: A17 ordering — keyframes ascending, (q,p,v,bg,ba) each; landmarks ascending ----
    pr = Problem()
    kf_blocks = np.empty((n_kf, 5), np.int32)
    x0 = {}
    for i in range(n_kf):
        q0 = _perturb_quat(q_true[i], rng, sigma_rot)
        kf_blocks[i, 0] = pr.add_quat(q0)
        kf_blocks[i, 1] = pr.add_block(p_true[i] + rng.normal(0, sigma_pos, 3))
        kf_blocks[i, 2] = pr.add_block(v_true[i] + rng.normal(0, sigma_vel, 3))
        kf_blocks[i, 3] = pr.add_block(np.zeros(3))
        kf_blocks[i, 4] = pr.add_block(np.zeros(3))
    lm_blocks = pr.add_blocks(P_true + rng.normal(0, sigma_lm, P_true.shape))
    cam = pr.add_camera(FX, FY, CX, CY, R_cb, t_cb)
    n_obs = obs_l.size
    idx = np.stack([kf_blocks[obs_k, 0], kf_blocks[obs_k, 1], lm_blocks[obs_l], np.full(n_obs, cam, np.int32)], axis=1)
    consts = np.concatenate([uv, np.full((n_obs, 1), w_reproj)], axis=1)
    if cauchy_a is None:
        pr.add_factors(capi.F_REPROJ, idx, consts)
    else:
        pr.add_factors(capi.F_REPROJ, idx, consts, capi.LOSS_CAUCHY, cauchy_a * w_reproj)  # a = 5 w: visual_odometry_params.h:77-80
    if imu_fac:
        idx = np.concatenate([kf_blocks[:-1], kf_blocks[1:]], axis=1)
        pr.add_factors(capi.F_IMU_DELTA, idx, np.stack(imu_fac))
    # pseudo-marginalisation prior on the first state at its current estimate, cov 1e-5 I
    # (bs_optimizers/src/fixed_lag_smoother.cpp:244-268)
    if first_prior:
        vals = pr.values
        mean = np.concatenate([pr.block(int(b), vals) for b in kf_blocks[0]])
        A = sqrt_information_upper(1e-5 * np.eye(15))
        pr.add_factors(capi.F_IMU_PRIOR, kf_blocks[0][None, :], np.concatenate([mean, A.ravel()])[None, :])
    pr.meta = dict(kind="vio_window", n_kf=n_kf, n_lm=n_lm, n_obs=int(n_obs), n_imu=len(imu_fac), seed=seed, kf0=kf0,
                   kf_blocks=kf_blocks, lm_blocks=lm_blocks, q_true=q_true, p_true=p_true, v_true=v_true,
                   P_true=P_true, bg_true=bg_true, ba_true=ba_true)
    return pr


def c1(seed=20250620):
    """BASELINE config 1: 20 keyframes x 500 landmarks (plumbing case)."""
    return vio_window(n_kf=20, n_lm=500, seed=seed)


def c2(seed=20250620):
    """BASELINE config 2: 200 keyframes x 50k landmarks, ~400k reprojection factors."""
    return vio_window(n_kf=200, n_lm=50000, seed=seed)


def chain_window(rank, n_windows, n_kf=200, n_lm=50000, seed=20250620):
    """Window `rank` of a chain of n_windows submaps of ONE trajectory: consecutive windows share their boundary key frame (the last
    one of window r is the first one of window r + 1: its five blocks q, p, v, bg, ba exist in both) and nothing else — landmarks are
    local to a submap, as in the reference's global map.  Only window 0 carries the prior on its first state.
    meta["shared"]: {neighbour rank: the five LOCAL blocks of the key frame shared with it}."""
    pr = vio_window(n_kf=n_kf, n_lm=n_lm, seed=seed + 10 + rank, kf0=rank * (n_kf - 1), first_prior=(rank == 0), bias_seed=seed)
    kf = pr.meta["kf_blocks"]
    shared = {}
    if rank > 0:
        shared[rank - 1] = [int(b) for b in kf[0]]
    if rank + 1 < n_windows:
        shared[rank + 1] = [int(b) for b in kf[-1]]
    pr.meta["shared"] = shared
    return pr


def merge_chain(windows):
    """The graph the windows of a chain describe TOGETHER: one Problem in which every shared key frame exists once (with the value the
    earlier window holds).  Returns (merged Problem, [local block -> merged block per window])."""
    from .problem import NIDX
    cam_types = (capi.F_REPROJ, capi.F_REPROJ_ONLINE_CALIB, capi.F_IDP_REPROJ, capi.F_IDP_REPROJ_UNARY)
    mp = Problem()
    maps = []
    for r, w in enumerate(windows):
        m = -np.ones(w.n_blocks, np.int64)
        if r > 0:
            prev_last = windows[r - 1].meta["kf_blocks"][-1]
            for lb, pb in zip(w.meta["kf_blocks"][0], prev_last):
                m[int(lb)] = maps[r - 1][int(pb)]
        vals = w.values
        for b in range(w.n_blocks):
            if m[b] >= 0:
                continue
            nb = mp.add_block(w.block(b, vals), const=bool(w.is_const[b]))
            mp.manifold[nb] = w.manifold[b]
            m[b] = nb
        if r == 0:
            mp.cameras = list(w.cameras)
        for t, chunks in w.factors.items():
            nvar = NIDX[t] - (1 if t in cam_types else 0)
            for idx, consts, lk, la in chunks:
                li = idx.copy()
                li[:, :nvar] = m[li[:, :nvar]]
                mp.add_factors(t, li, consts, lk, la)
        maps.append(m)
    return mp, maps


# ---------------------------------------------------------------------------------------------
# C3: lidar-inertial window — relative-pose factors with (constant) extrinsics + IMU factors
# ---------------------------------------------------------------------------------------------
def idp_window(n_kf=8, n_lm=60, seed=20250623, kf_rate=10.0, track_min=3, track_max=6, pixel_sigma=0.5,
               w_reproj=1.0, cauchy_a=5.0, sigma_rot=0.01, sigma_pos=0.03, sigma_rho=0.02, with_unary=True,
               unit_bearing=True):
    """Visual window with inverse-depth landmarks (use_idp: true, config/vo/vo_params.json:4): every landmark is a
    scalar rho anchored at the first keyframe that sees it, with a constant bearing m in the anchor camera
    (point in the anchor camera = m / rho, bs_variables/include/bs_variables/inverse_depth_landmark.h:41-45);
    the anchor observation is the unary constraint, every later one the binary constraint.
    The first keyframe carries a pose prior and the second a position prior, both at the generating poses
    (they fix the gauge and the scale)."""
    rng = np.random.default_rng(seed)
    dt_kf = 1.0 / kf_rate
    traj = Lissajous(duration=max(n_kf * dt_kf, 2.0))
    t_kf = np.arange(n_kf) * dt_kf
    R_true = np.stack([traj.rot(t) for t in t_kf])
    q_true = np.stack([rot_to_quat(R) for R in R_true])
    p_true = np.stack([traj.pos(t) for t in t_kf])
    R_cb, t_cb = _t_cam_baselink()
    track_max = min(track_max, n_kf)
    track_min = min(track_min, track_max)
    L = rng.integers(track_min, track_max + 1, n_lm)
    k0 = (rng.random(n_lm) * (n_kf - L + 1)).astype(np.int64)
    u0 = rng.uniform(0.2, 0.8, n_lm) * IMG_W
    v0 = rng.uniform(0.2, 0.8, n_lm) * IMG_H
    depth = rng.uniform(3.0, 12.0, n_lm)
    ray = np.stack([(u0 - CX) / FX, (v0 - CY) / FY, np.ones(n_lm)], axis=1)
    Pc = ray * depth[:, None]
    if unit_bearing:
        bearing = ray / np.linalg.norm(ray, axis=1, keepdims=True)   # unit direction, rho = 1 / range
        rho_true = 1.0 / np.linalg.norm(Pc, axis=1)
    else:
        bearing = ray                                                # [mx, my, 1], rho = 1 / z
        rho_true = 1.0 / depth
    Pb = (Pc - t_cb) @ R_cb
    P_true = np.einsum('nij,nj->ni', R_true[k0], Pb) + p_true[k0]

    pr = Problem()
    kf_blocks = np.empty((n_kf, 2), np.int32)
    for i in range(n_kf):
        kf_blocks[i, 0] = pr.add_quat(_perturb_quat(q_true[i], rng, sigma_rot))
        kf_blocks[i, 1] = pr.add_block(p_true[i] + rng.normal(0, sigma_pos, 3))
    rho_blocks = np.array([pr.add_block([r * (1.0 + rng.normal(0, sigma_rho))]) for r in rho_true], np.int32)
    cam = pr.add_camera(FX, FY, CX, CY, R_cb, t_cb)
    bin_idx, bin_c, un_idx, un_c = [], [], [], []
    for l in range(n_lm):
        for k in range(k0[l], k0[l] + L[l]):
            Pck = R_cb @ (R_true[k].T @ (P_true[l] - p_true[k])) + t_cb
            if Pck[2] < 0.5:
                continue
            uv = np.array([FX * Pck[0] / Pck[2] + CX, FY * Pck[1] / Pck[2] + CY]) + rng.normal(0, pixel_sigma, 2)
            c = np.concatenate([uv, [w_reproj], bearing[l]])
            if k == k0[l]:
                if with_unary:
                    un_idx.append([kf_blocks[k, 0], kf_blocks[k, 1], rho_blocks[l], cam]); un_c.append(c)
            else:
                bin_idx.append([kf_blocks[k0[l], 0], kf_blocks[k0[l], 1], kf_blocks[k, 0], kf_blocks[k, 1], rho_blocks[l], cam])
                bin_c.append(c)
    loss = (capi.LOSS_TRIVIAL, 1.0) if cauchy_a is None else (capi.LOSS_CAUCHY, cauchy_a * w_reproj)
    pr.add_factors(capi.F_IDP_REPROJ, np.array(bin_idx, np.int32), np.array(bin_c), *loss)
    if un_idx:
        pr.add_factors(capi.F_IDP_REPROJ_UNARY, np.array(un_idx, np.int32), np.array(un_c), *loss)
    A = sqrt_information_upper(1e-4 * np.eye(6))
    b = np.concatenate([p_true[0], q_true[0]])
    pr.add_factors(capi.F_ABSPOSE, [[kf_blocks[0, 1], kf_blocks[0, 0]]], [np.concatenate([b, A.ravel()])])
    if n_kf > 1:
        A3 = sqrt_information_upper(1e-6 * np.eye(3))
        pr.add_factors(capi.F_ABS_VEC3, [[kf_blocks[1, 1]]], [np.concatenate([p_true[1], A3.ravel()])])
    pr.meta = dict(kind="idp_window", n_kf=n_kf, n_lm=n_lm, seed=seed, kf_blocks=kf_blocks, rho_blocks=rho_blocks,
                   q_true=q_true, p_true=p_true, rho_true=rho_true, n_binary=len(bin_idx), n_unary=len(un_idx))
    return pr


def lio_window(n_kf=100, n_rel=20000, seed=20250621, w_lidar=1.0, w_inertial=1e-2, max_gap=10, free_extrinsics=False):
    rng = np.random.default_rng(seed)
    base = vio_window(n_kf=n_kf, n_lm=0, seed=seed, w_inertial=w_inertial)
    pr = base
    pr.factors.pop(capi.F_REPROJ, None)
    kf = pr.meta["kf_blocks"]
    q_true, p_true = pr.meta["q_true"], pr.meta["p_true"]
    # constant extrinsic pair (bs_variables::Position3D / Orientation3D: holdConstant() == true)
    q_bs = quat_from_aa(np.array([0.02, -0.01, 0.03]))
    p_bs = np.array([0.1, -0.05, 0.2])
    # (free_extrinsics: online calibration — the pair is estimated with the window, every relative-pose factor names it)
    b_pe = pr.add_block(p_bs, const=not free_extrinsics)
    b_qe = pr.add_quat(q_bs, const=not free_extrinsics)
    R_bs = quat_to_rot(q_bs)
    i = rng.integers(0, n_kf - 1, n_rel)
    gap = rng.integers(1, max_gap + 1, n_rel)
    j = np.minimum(i + gap, n_kf - 1)
    sig_p, sig_r = 0.01, np.deg2rad(0.2)
    cov = np.diag([sig_p ** 2] * 3 + [sig_r ** 2] * 3) / (w_lidar ** 2)
    A = sqrt_information_upper(cov)
    consts = np.empty((n_rel, 43))
    R_all = quats_to_rots(q_true)
    for k in range(n_rel):
        R1s, R2s = R_all[i[k]] @ R_bs, R_all[j[k]] @ R_bs
        p1s, p2s = R_all[i[k]] @ p_bs + p_true[i[k]], R_all[j[k]] @ p_bs + p_true[j[k]]
        dR = R1s.T @ R2s @ so3_exp(rng.normal(0, sig_r, 3))
        dp = R1s.T @ (p2s - p1s) + rng.normal(0, sig_p, 3)
        consts[k, 0:3] = dp
        consts[k, 3:7] = rot_to_quat(dR)
        consts[k, 7:] = A.ravel()
    idx = np.stack([kf[i, 1], kf[i, 0], kf[j, 1], kf[j, 0], np.full(n_rel, b_pe), np.full(n_rel, b_qe)], axis=1)
    pr.add_factors(capi.F_RELPOSE_EXT, idx, consts, capi.LOSS_CAUCHY, 1.0)  # pose_3d_stamped_transaction.cpp:18,80
    pr.meta.update(kind="lio_window", n_rel=n_rel)
    return pr


def c3(seed=20250621):
    return lio_window(100, 20000, seed)


# ---------------------------------------------------------------------------------------------
# C4: global-mapper pose graph
# ---------------------------------------------------------------------------------------------
def pose_graph(n_pose=5000, n_loop=45001, seed=20250622):
    rng = np.random.default_rng(seed)
    walk = RandomWalkLoop(n_pose, rng)
    R_true, p_true = np.stack(walk.R), np.stack(walk.p)
    pr = Problem()
    blocks = np.empty((n_pose, 2), np.int32)
    for k in range(n_pose):
        blocks[k, 0] = pr.add_block(p_true[k] + rng.normal(0, 0.05, 3))
        blocks[k, 1] = pr.add_quat(_perturb_quat(rot_to_quat(R_true[k]), rng, 0.02))
    A_odom = sqrt_information_upper(1e-3 * np.eye(6))  # global_map.json:6-21
    A_loop = sqrt_information_upper(1e-5 * np.eye(6))
    i = np.concatenate([np.arange(n_pose - 1), rng.integers(0, n_pose, n_loop)])
    j = np.concatenate([np.arange(1, n_pose), rng.integers(0, n_pose, n_loop)])
    same = i == j
    j[same] = (j[same] + 1) % n_pose
    n = i.size
    consts = np.empty((n, 43))
    for k in range(n):
        odom = k < n_pose - 1
        sp, sr = (np.sqrt(1e-3), np.sqrt(1e-3)) if odom else (np.sqrt(1e-5), np.sqrt(1e-5))
        dR = R_true[i[k]].T @ R_true[j[k]] @ so3_exp(rng.normal(0, sr, 3))
        dp = R_true[i[k]].T @ (p_true[j[k]] - p_true[i[k]]) + rng.normal(0, sp, 3)
        consts[k, 0:3] = dp
        consts[k, 3:7] = rot_to_quat(dR)
        consts[k, 7:] = (A_odom if odom else A_loop).ravel()
    idx = np.stack([blocks[i, 0], blocks[i, 1], blocks[j, 0], blocks[j, 1]], axis=1)
    pr.add_factors(capi.F_RELPOSE, idx, consts, capi.LOSS_CAUCHY, 1.0)
    # prior cov 1e-9 I on pose 0 (submap_pose_graph_optimization.h:40)
    A0 = sqrt_information_upper(1e-9 * np.eye(6))
    b = np.concatenate([pr.block(int(blocks[0, 0])), pr.block(int(blocks[0, 1]))])
    pr.add_factors(capi.F_ABSPOSE, blocks[0][None, :], np.concatenate([b, A0.ravel()])[None, :])
    pr.meta = dict(kind="pose_graph", n_pose=n_pose, n_rel=int(n), seed=seed, blocks=blocks, R_true=R_true, p_true=p_true)
    return pr


def pose_graph_local(n_pose=5000, n_loop=45001, row_len=25, seed=20250623):
    """A pose graph whose loop closures are SPATIALLY LOCAL, as a mapper's are (a place is re-observed from the poses that pass near
    it: bs_models/src/lib/global_mapping/submap_pose_graph_optimization.cpp:96-141 matches a submap against its spatial neighbours):
    a boustrophedon sweep of rows of `row_len` poses, odometry along the path, every loop closure between a pose and one within two
    rows and two columns of it.  C4's sizes and weights; what differs from `pose_graph` is only WHICH pairs the loops join — there
    uniformly random pairs (the reduced system fills in completely), here pairs at most ~4 rows apart in time (a band)."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_pose)
    row, col = k // row_len, k % row_len
    col = np.where(row % 2 == 0, col, row_len - 1 - col)
    yaw = np.where(row % 2 == 0, 0.0, np.pi) + 0.05 * np.sin(0.3 * k)
    p_true = np.stack([1.0 * col, 1.0 * row, 0.1 * np.sin(0.05 * k)], axis=1)
    R_true = np.stack([so3_exp(np.array([0.02 * np.sin(0.1 * t), 0.02 * np.cos(0.07 * t), y])) for t, y in zip(k, yaw)])
    pr = Problem()
    blocks = np.empty((n_pose, 2), np.int32)
    for t in range(n_pose):
        blocks[t, 0] = pr.add_block(p_true[t] + rng.normal(0, 0.05, 3))
        blocks[t, 1] = pr.add_quat(_perturb_quat(rot_to_quat(R_true[t]), rng, 0.02))
    A_odom = sqrt_information_upper(1e-3 * np.eye(6))
    A_loop = sqrt_information_upper(1e-5 * np.eye(6))
    li = rng.integers(0, n_pose, n_loop)
    drow, dcol = rng.integers(-2, 3, n_loop), rng.integers(-2, 3, n_loop)
    r2 = np.clip(row[li] + drow, 0, (n_pose - 1) // row_len)
    c2_ = np.clip(col[li] + dcol, 0, row_len - 1)
    lj = r2 * row_len + np.where(r2 % 2 == 0, c2_, row_len - 1 - c2_)
    lj = np.clip(lj, 0, n_pose - 1)
    same = li == lj
    lj[same] = (lj[same] + 1) % n_pose
    i = np.concatenate([np.arange(n_pose - 1), li])
    j = np.concatenate([np.arange(1, n_pose), lj])
    n = i.size
    consts = np.empty((n, 43))
    for t in range(n):
        odom = t < n_pose - 1
        sg = np.sqrt(1e-3) if odom else np.sqrt(1e-5)
        dR = R_true[i[t]].T @ R_true[j[t]] @ so3_exp(rng.normal(0, sg, 3))
        dp = R_true[i[t]].T @ (p_true[j[t]] - p_true[i[t]]) + rng.normal(0, sg, 3)
        consts[t, 0:3] = dp
        consts[t, 3:7] = rot_to_quat(dR)
        consts[t, 7:] = (A_odom if odom else A_loop).ravel()
    idx = np.stack([blocks[i, 0], blocks[i, 1], blocks[j, 0], blocks[j, 1]], axis=1)
    pr.add_factors(capi.F_RELPOSE, idx, consts, capi.LOSS_CAUCHY, 1.0)
    A0 = sqrt_information_upper(1e-9 * np.eye(6))
    b = np.concatenate([pr.block(int(blocks[0, 0])), pr.block(int(blocks[0, 1]))])
    pr.add_factors(capi.F_ABSPOSE, blocks[0][None, :], np.concatenate([b, A0.ravel()])[None, :])
    pr.meta = dict(kind="pose_graph_local", n_pose=n_pose, n_rel=int(n), seed=seed, blocks=blocks, R_true=R_true, p_true=p_true, row_len=row_len)
    return pr


def c4(seed=20250622):
    return pose_graph(5000, 45001, seed)


def c5_instance(rank):
    """BASELINE config 5: 8 independent C2 windows, seeds +10..+17, one per GPU."""
    return c2(seed=20250620 + 10 + int(rank))
