// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md).  Never linked into,
// imported by or executed from the product path (beam_slam_amd/).
//
// Restatement of the reference's cost functors (templated on the scalar so they run on doubles
// and on Jets exactly like ceres::AutoDiffCostFunction would run the originals), plus the pieces
// of Ceres' rotation.h, Eigen's Quaternion and fuse's NormalDelta/NormalPrior functors they call.
// Every function cites the reference line (relative to /root/reference) or the [EXT] dependency
// whose published algorithm it restates.
#pragma once
#include "jet.h"

namespace bso {

// ---------------------------------------------------------------------------------------------
// [EXT] ceres/rotation.h (Ceres Solver 1.14 / 2.x, unpinned: bs_constraints/package.xml:15)
// ---------------------------------------------------------------------------------------------
template <typename T> inline void QuaternionProduct(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

template <typename T> inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  const T t2 = q[0] * q[1];
  const T t3 = q[0] * q[2];
  const T t4 = q[0] * q[3];
  const T t5 = -q[1] * q[1];
  const T t6 = q[1] * q[2];
  const T t7 = q[1] * q[3];
  const T t8 = -q[2] * q[2];
  const T t9 = q[2] * q[3];
  const T t1 = -q[3] * q[3];
  result[0] = T(2.0) * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
  result[1] = T(2.0) * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
  result[2] = T(2.0) * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}

template <typename T> inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  // 'scale' is 1 / norm(q).
  const T scale = T(1.0) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuaternionRotatePoint(unit, pt, result);
}

template <typename T> inline void AngleAxisToQuaternion(const T aa[3], T q[4]) {
  const T theta_squared = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta_squared > 0.0) {
    const T theta = sqrt(theta_squared);
    const T half_theta = theta * T(0.5);
    const T k = sin(half_theta) / theta;
    q[0] = cos(half_theta);
    q[1] = aa[0] * k;
    q[2] = aa[1] * k;
    q[3] = aa[2] * k;
  } else {
    const T k(0.5);
    q[0] = T(1.0);
    q[1] = aa[0] * k;
    q[2] = aa[1] * k;
    q[3] = aa[2] * k;
  }
}

template <typename T> inline void QuaternionToAngleAxis(const T q[4], T aa[3]) {
  const T& q1 = q[1];
  const T& q2 = q[2];
  const T& q3 = q[3];
  const T sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
  if (sin_squared_theta > 0.0) {
    const T sin_theta = sqrt(sin_squared_theta);
    const T& cos_theta = q[0];
    // angle in (-pi, pi]: q and -q give the same rotation vector
    const T two_theta = T(2.0) * ((cos_theta < 0.0) ? atan2(-sin_theta, -cos_theta)
                                                     : atan2(sin_theta, cos_theta));
    const T k = two_theta / sin_theta;
    aa[0] = q1 * k;
    aa[1] = q2 * k;
    aa[2] = q3 * k;
  } else {
    const T k(2.0);
    aa[0] = q1 * k;
    aa[1] = q2 * k;
    aa[2] = q3 * k;
  }
}

// ---------------------------------------------------------------------------------------------
// [EXT] Eigen::Quaternion<T> (w,x,y,z kept in that order here)
// ---------------------------------------------------------------------------------------------
// Quaternion::toRotationMatrix() — does NOT normalise.
template <typename T> inline void EigenQuatToRot(const T q[4], T R[9]) {
  const T tx = T(2.0) * q[1], ty = T(2.0) * q[2], tz = T(2.0) * q[3];
  const T twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const T txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const T tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = T(1.0) - (tyy + tzz); R[1] = txy - twz;            R[2] = txz + twy;
  R[3] = txy + twz;            R[4] = T(1.0) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;            R[7] = tyz + twx;            R[8] = T(1.0) - (txx + tyy);
}
// Quaternion::operator*(Vector3) (_transformVector): v + w*uv + u x uv, uv = 2 u x v.
template <typename T> inline void EigenQuatTransform(const T q[4], const T v[3], T out[3]) {
  T uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  out[0] = v[0] + q[0] * uv[0] + (q[2] * uv[2] - q[3] * uv[1]);
  out[1] = v[1] + q[0] * uv[1] + (q[3] * uv[0] - q[1] * uv[2]);
  out[2] = v[2] + q[0] * uv[2] + (q[1] * uv[1] - q[2] * uv[0]);
}
template <typename T> inline void EigenQuatConj(const T q[4], T c[4]) {
  c[0] = q[0]; c[1] = -q[1]; c[2] = -q[2]; c[3] = -q[3];
}
// Quaternion::inverse(): conjugate / squaredNorm
template <typename T> inline void EigenQuatInverse(const T q[4], T c[4]) {
  const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  c[0] = q[0] / n2; c[1] = -q[1] / n2; c[2] = -q[2] / n2; c[3] = -q[3] / n2;
}

template <typename T> inline void Mat3Vec(const T M[9], const T v[3], T o[3]) {
  o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
template <typename T> inline void Mat3TVec(const T M[9], const T v[3], T o[3]) {
  o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}

struct Camera {  // K (as used: full 3x3 with zero skew) and T_cam_baselink
  double fx, fy, cx, cy;
  double R_cb[9];
  double t_cb[3];
};

// ---------------------------------------------------------------------------------------------
// A6 twin: bs_constraints/include/bs_constraints/visual/euclidean_reprojection_functor.h:16-81
// (same residual as the analytic EuclideanReprojection, euclidean_reprojection_function.h:66-94)
//   blocks: q_WORLD_BASELINK[4], t_WORLD_BASELINK[3], P_WORLD[3]
// ---------------------------------------------------------------------------------------------
template <typename T>
inline void ReprojResidual(const Camera& cam, const double pix[2], double w, const T* q, const T* t,
                           const T* P, T r[2]) {
  T R[9];
  EigenQuatToRot(q, R);                       // q.toRotationMatrix()  (function.h:68-70)
  T a[3], b[3], Pb[3];
  Mat3TVec(R, P, a);                          // R_BASELINK_WORLD * P_WORLD
  Mat3TVec(R, t, b);                          // R_BASELINK_WORLD * t_WORLD_BASELINK
  Pb[0] = a[0] - b[0]; Pb[1] = a[1] - b[1]; Pb[2] = a[2] - b[2];      // (function.h:81-82)
  T Pc[3];
  for (int i = 0; i < 3; ++i)
    Pc[i] = T(cam.R_cb[3 * i]) * Pb[0] + T(cam.R_cb[3 * i + 1]) * Pb[1] + T(cam.R_cb[3 * i + 2]) * Pb[2] +
            T(cam.t_cb[i]);                   // (function.h:85)
  // (K * P_CAMERA).hnormalized()  (function.h:88)
  const T hx = T(cam.fx) * Pc[0] + T(cam.cx) * Pc[2];
  const T hy = T(cam.fy) * Pc[1] + T(cam.cy) * Pc[2];
  const T u = hx / Pc[2];
  const T v = hy / Pc[2];
  r[0] = T(w) * (T(pix[0]) - u);              // information_matrix (= w*I2) * (z - u) (function.h:91-94)
  r[1] = T(w) * (T(pix[1]) - v);
}

// ---------------------------------------------------------------------------------------------
// A6: bs_constraints/include/bs_constraints/visual/euclidean_reprojection_functor_online_calib.h:16-83
//   blocks: q_WB[4], p_WB[3], P[3], q_BASELINK_CAM[4], p_BASELINK_CAM[3]
// ---------------------------------------------------------------------------------------------
template <typename T>
inline void ReprojOnlineCalibResidual(const Camera& cam, const double pix[2], double w, const T* q,
                                      const T* p, const T* P, const T* qe, const T* pe, T r[2]) {
  T Rwb[9], Rbc[9];
  EigenQuatToRot(q, Rwb);    // helpers.h:14-25
  EigenQuatToRot(qe, Rbc);
  // T_BASELINK_WORLD * P = Rwb^T P - Rwb^T p   (helpers.h:27-35)
  T a[3], b[3], Pb[3];
  Mat3TVec(Rwb, P, a);
  Mat3TVec(Rwb, p, b);
  for (int i = 0; i < 3; ++i) Pb[i] = a[i] - b[i];
  // T_CAM_BASELINK * Pb = Rbc^T Pb - Rbc^T pe
  T c[3], d[3], Pc[3];
  Mat3TVec(Rbc, Pb, c);
  Mat3TVec(Rbc, pe, d);
  for (int i = 0; i < 3; ++i) Pc[i] = c[i] - d[i];
  const T hx = T(cam.fx) * Pc[0] + T(cam.cx) * Pc[2];
  const T hy = T(cam.fy) * Pc[1] + T(cam.cy) * Pc[2];
  r[0] = T(w) * (T(pix[0]) - hx / Pc[2]);
  r[1] = T(w) * (T(pix[1]) - hy / Pc[2]);
}

// ---------------------------------------------------------------------------------------------
// A8: bs_constraints/include/bs_constraints/inertial/normal_delta_imu_state_3d_cost_functor.h:59-141
//   consts layout (bsgpu.h BSGPU_F_IMU_DELTA): dt, dq[4], dp[3], dv[3], dq_dbg[9], dp_dbg[9],
//   dp_dba[9], dv_dbg[9], dv_dba[9], bg_lin[3], ba_lin[3], A[225]
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// A7: bs_constraints/include/bs_constraints/visual/inversedepth_reprojection_functor.h:57-125
//   blocks: q_WORLD_BASELINKa[4], p_WORLD_BASELINKa[3], q_WORLD_BASELINKm[4], p_WORLD_BASELINKm[3], rho[1]
//   consts: pixel u,v; w (sqrt information = w*I2, inversedepth_reprojection_constraint.cpp:33-34); bearing m[3]
// Rigid transforms are kept as (R row-major 3x3, t) pairs; compose/invert follow helpers.h:14-35.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Rigid { T R[9]; T t[3]; };
template <typename T> inline Rigid<T> RigidCompose(const Rigid<T>& A, const Rigid<T>& B) {  // A * B
  Rigid<T> C;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * i] * B.R[j] + A.R[3 * i + 1] * B.R[3 + j] + A.R[3 * i + 2] * B.R[6 + j];
    C.t[i] = A.R[3 * i] * B.t[0] + A.R[3 * i + 1] * B.t[1] + A.R[3 * i + 2] * B.t[2] + A.t[i];
  }
  return C;
}
template <typename T> inline Rigid<T> RigidInverse(const Rigid<T>& A) {  // helpers.h:27-35
  Rigid<T> C;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C.R[3 * i + j] = A.R[3 * j + i];
    C.t[i] = -(A.R[i] * A.t[0] + A.R[3 + i] * A.t[1] + A.R[6 + i] * A.t[2]);
  }
  return C;
}
template <typename T> inline Rigid<T> RigidFromQP(const T* q, const T* p) {  // helpers.h:14-25
  Rigid<T> C;
  EigenQuatToRot(q, C.R);
  C.t[0] = p[0]; C.t[1] = p[1]; C.t[2] = p[2];
  return C;
}
template <typename T> inline Rigid<T> CameraExtrinsic(const Camera& cam) {  // T_cam_baselink as T
  Rigid<T> C;
  for (int i = 0; i < 9; ++i) C.R[i] = T(cam.R_cb[i]);
  for (int i = 0; i < 3; ++i) C.t[i] = T(cam.t_cb[i]);
  return C;
}
// (K [R|t] (m; rho)).hnormalized(), residual = w (pixel - reproj)   (functor.h:107-121)
template <typename T>
inline void ProjectBearing(const Camera& cam, const double* k, const Rigid<T>& T_cm_ca, const T& rho, T r[2]) {
  T c[3];
  for (int i = 0; i < 3; ++i)
    c[i] = T_cm_ca.R[3 * i] * T(k[3]) + T_cm_ca.R[3 * i + 1] * T(k[4]) + T_cm_ca.R[3 * i + 2] * T(k[5]) + T_cm_ca.t[i] * rho;
  const T hx = T(cam.fx) * c[0] + T(cam.cx) * c[2];
  const T hy = T(cam.fy) * c[1] + T(cam.cy) * c[2];
  r[0] = T(k[2]) * (T(k[0]) - hx / c[2]);
  r[1] = T(k[2]) * (T(k[1]) - hy / c[2]);
}
template <typename T>
inline void InverseDepthReprojResidual(const Camera& cam, const double* k, const T* qa, const T* pa, const T* qm,
                                       const T* pm, const T* rho, T r[2]) {
  const Rigid<T> T_BASELINK_CAM = RigidInverse(CameraExtrinsic<T>(cam));                    // functor.h:64-65
  const Rigid<T> T_WORLD_CAMERAa = RigidCompose(RigidFromQP(qa, pa), T_BASELINK_CAM);       // :68-71
  const Rigid<T> T_WORLD_CAMERAm = RigidCompose(RigidFromQP(qm, pm), T_BASELINK_CAM);       // :74-77
  const Rigid<T> T_CAMERAm_CAMERAa = RigidCompose(RigidInverse(T_WORLD_CAMERAm), T_WORLD_CAMERAa);  // :80-81
  ProjectBearing(cam, k, T_CAMERAm_CAMERAa, rho[0], r);
}
// A7 unary: inversedepth_reprojection_functor_unary.h:36-72 — the observation made in the anchor frame itself:
// T_CAMERAm_CAMERAa = I, so the residual does not depend on any parameter block (all Jacobians are zero).
template <typename T>
inline void InverseDepthReprojUnaryResidual(const Camera& cam, const double* k, const T* /*qa*/, const T* /*pa*/,
                                            const T* rho, T r[2]) {
  Rigid<T> I;
  for (int i = 0; i < 9; ++i) I.R[i] = T(i % 4 == 0 ? 1.0 : 0.0);
  for (int i = 0; i < 3; ++i) I.t[i] = T(0.0);
  ProjectBearing(cam, k, I, rho[0], r);
}

static const double kGravityWorld[3] = {0.0, 0.0, -9.80665};  // bs_common/include/bs_common/utils.h:20-24

template <typename T>
inline void ImuDeltaResidual(const double* c, const T* q_i, const T* p_i, const T* v_i, const T* bg_i,
                             const T* ba_i, const T* q_j, const T* p_j, const T* v_j, const T* bg_j,
                             const T* ba_j, T r[15]) {
  const double dt = c[0];
  const double* dq = c + 1;
  const double* dp = c + 5;
  const double* dv = c + 8;
  const double* dq_dbg = c + 11;
  const double* dp_dbg = c + 20;
  const double* dp_dba = c + 29;
  const double* dv_dbg = c + 38;
  const double* dv_dba = c + 47;
  const double* bg_lin = c + 56;
  const double* ba_lin = c + 59;
  const double* A = c + 62;

  T dbg[3], dba[3];
  for (int k = 0; k < 3; ++k) { dbg[k] = bg_i[k] - T(bg_lin[k]); dba[k] = ba_i[k] - T(ba_lin[k]); }  // :86-87

  // q_tmp = dq_dbg * dbg; q_corrected = dq * DeltaQ(q_tmp)   (:96-97; DeltaQ: utils.h:28-38, NOT normalised)
  T qtmp[3];
  for (int i = 0; i < 3; ++i)
    qtmp[i] = T(dq_dbg[3 * i]) * dbg[0] + T(dq_dbg[3 * i + 1]) * dbg[1] + T(dq_dbg[3 * i + 2]) * dbg[2];
  T dQ[4] = {T(1.0), qtmp[0] / T(2.0), qtmp[1] / T(2.0), qtmp[2] / T(2.0)};
  T dqT[4] = {T(dq[0]), T(dq[1]), T(dq[2]), T(dq[3])};
  T q_corr[4];
  QuaternionProduct(dqT, dQ, q_corr);  // Hamilton product == Eigen operator*
  T p_corr[3], v_corr[3];
  for (int i = 0; i < 3; ++i) {        // :98-99
    p_corr[i] = T(dp[i]) + (T(dp_dbg[3 * i]) * dbg[0] + T(dp_dbg[3 * i + 1]) * dbg[1] + T(dp_dbg[3 * i + 2]) * dbg[2]) +
                (T(dp_dba[3 * i]) * dba[0] + T(dp_dba[3 * i + 1]) * dba[1] + T(dp_dba[3 * i + 2]) * dba[2]);
    v_corr[i] = T(dv[i]) + (T(dv_dbg[3 * i]) * dbg[0] + T(dv_dbg[3 * i + 1]) * dbg[1] + T(dv_dbg[3 * i + 2]) * dbg[2]) +
                (T(dv_dba[3 * i]) * dba[0] + T(dv_dba[3 * i + 1]) * dba[1] + T(dv_dba[3 * i + 2]) * dba[2]);
  }
  // res_q = 2 * (q_corrected.inverse() * (q_i.inverse() * q_j)).vec()   (:103-104)
  T qc_inv[4], qi_inv[4], qij[4], e[4];
  EigenQuatInverse(q_corr, qc_inv);
  EigenQuatInverse(q_i, qi_inv);
  QuaternionProduct(qi_inv, q_j, qij);
  QuaternionProduct(qc_inv, qij, e);
  T res[15];
  res[0] = T(2.0) * e[1]; res[1] = T(2.0) * e[2]; res[2] = T(2.0) * e[3];
  // res_p = q_i.conjugate() * (p_j - p_i - dt v_i - 0.5 dt^2 G) - p_corrected   (:107-110)
  T qi_c[4];
  EigenQuatConj(q_i, qi_c);
  T a[3], ra[3];
  for (int k = 0; k < 3; ++k) a[k] = p_j[k] - p_i[k] - T(dt) * v_i[k] - T(0.5 * dt * dt * kGravityWorld[k]);
  EigenQuatTransform(qi_c, a, ra);
  for (int k = 0; k < 3; ++k) res[3 + k] = ra[k] - p_corr[k];
  // res_v = q_i.conjugate() * (v_j - v_i - dt G) - v_corrected   (:113-114)
  for (int k = 0; k < 3; ++k) a[k] = v_j[k] - v_i[k] - T(dt * kGravityWorld[k]);
  EigenQuatTransform(qi_c, a, ra);
  for (int k = 0; k < 3; ++k) res[6 + k] = ra[k] - v_corr[k];
  for (int k = 0; k < 3; ++k) { res[9 + k] = bg_j[k] - bg_i[k]; res[12 + k] = ba_j[k] - ba_i[k]; }  // :116-117
  // residual = A * residual   (:137-138)
  for (int i = 0; i < 15; ++i) {
    T s(0.0);
    for (int k = 0; k < 15; ++k) s = s + T(A[15 * i + k]) * res[k];
    r[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// [EXT] fuse_constraints::NormalPriorOrientation3DCostFunctor with A = I3:
//   r = QuaternionToAngleAxis(b^-1 (x) q)
// ---------------------------------------------------------------------------------------------
template <typename T> inline void PriorOrientation(const double b[4], const T* q, T r[3]) {
  T obs_inv[4] = {T(b[0]), T(-b[1]), T(-b[2]), T(-b[3])};
  T qq[4] = {q[0], q[1], q[2], q[3]};
  T diff[4];
  QuaternionProduct(obs_inv, qq, diff);
  QuaternionToAngleAxis(diff, r);
}

// A10: bs_constraints/include/bs_constraints/inertial/normal_prior_imu_state_3d_cost_functor.h:57-88
//   consts: b[16] (q wxyz, p, v, bg, ba), A[225]
template <typename T>
inline void ImuPriorResidual(const double* c, const T* q, const T* p, const T* v, const T* bg, const T* ba,
                             T r[15]) {
  const double* b = c;
  const double* A = c + 16;
  T res[15];
  PriorOrientation(b, q, res);                              // :63
  for (int k = 0; k < 3; ++k) {
    res[3 + k] = p[k] - T(b[4 + k]);                        // :65-67
    res[6 + k] = v[k] - T(b[7 + k]);                        // :69-71
    res[9 + k] = bg[k] - T(b[10 + k]);                      // :73-75
    res[12 + k] = ba[k] - T(b[13 + k]);                     // :77-79
  }
  for (int i = 0; i < 15; ++i) {                            // :81-82
    T s(0.0);
    for (int k = 0; k < 15; ++k) s = s + T(A[15 * i + k]) * res[k];
    r[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// [EXT] fuse_constraints::NormalDeltaPose3DCostFunctor(A, d):
//   r_p = R(q1)^-1 (p2 - p1) - d_p ;  r_q = AngleAxis(d_q^-1 (x) q1^-1 (x) q2) ;  r = A [r_p; r_q]
//   consts: d[7] (x,y,z,qw,qx,qy,qz), A[36]
// ---------------------------------------------------------------------------------------------
template <typename T>
inline void DeltaPoseResidual(const double* c, const T* p1, const T* q1, const T* p2, const T* q2, T r[6]) {
  const double* d = c;
  const double* A = c + 7;
  T q1_inv[4] = {q1[0], -q1[1], -q1[2], -q1[3]};
  T dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  T dpr[3];
  QuaternionRotatePoint(q1_inv, dp, dpr);
  T res[6];
  res[0] = dpr[0] - T(d[0]); res[1] = dpr[1] - T(d[1]); res[2] = dpr[2] - T(d[2]);
  T obs_inv[4] = {T(d[3]), T(-d[4]), T(-d[5]), T(-d[6])};
  T q2c[4] = {q2[0], q2[1], q2[2], q2[3]};
  T diff[4], err[4];
  QuaternionProduct(q1_inv, q2c, diff);
  QuaternionProduct(obs_inv, diff, err);
  QuaternionToAngleAxis(err, res + 3);
  for (int i = 0; i < 6; ++i) {
    T s(0.0);
    for (int k = 0; k < 6; ++k) s = s + T(A[6 * i + k]) * res[k];
    r[i] = s;
  }
}

// A11: bs_constraints/include/bs_constraints/relative_pose/delta_pose_3d_with_extrinsics_cost_functor.h:65-109
template <typename T>
inline void TransformPoseToSensorFrame(const T* p_wb, const T* o_wb, const T* p_bs, const T* o_bs, T* p_ws,
                                       T* o_ws) {
  QuaternionProduct(o_wb, o_bs, o_ws);          // :101
  T t[3];
  QuaternionRotatePoint(o_wb, p_bs, t);         // :103-104
  p_ws[0] = t[0] + p_wb[0]; p_ws[1] = t[1] + p_wb[1]; p_ws[2] = t[2] + p_wb[2];  // :105-107
}
template <typename T>
inline void DeltaPoseExtResidual(const double* c, const T* p1, const T* q1, const T* p2, const T* q2,
                                 const T* pe, const T* qe, T r[6]) {
  T ps1[3], qs1[4], ps2[3], qs2[4];
  TransformPoseToSensorFrame(p1, q1, pe, qe, ps1, qs1);   // :73-77
  TransformPoseToSensorFrame(p2, q2, pe, qe, ps2, qs2);   // :79-83
  DeltaPoseResidual(c, ps1, qs1, ps2, qs2, r);            // :85-87
}

// [EXT] fuse_constraints::NormalPriorPose3DCostFunctor(A, b): r = A [p - b_p ; AngleAxis(b_q^-1 (x) q)]
// (bs_constraints/src/global/absolute_pose_3d_constraint.cpp:45-50)   consts: b[7], A[36]
template <typename T> inline void PriorPoseResidual(const double* c, const T* p, const T* q, T r[6]) {
  const double* b = c;
  const double* A = c + 7;
  T res[6];
  res[0] = p[0] - T(b[0]); res[1] = p[1] - T(b[1]); res[2] = p[2] - T(b[2]);
  PriorOrientation(b + 3, q, res + 3);
  for (int i = 0; i < 6; ++i) {
    T s(0.0);
    for (int k = 0; k < 6; ++k) s = s + T(A[6 * i + k]) * res[k];
    r[i] = s;
  }
}

// [EXT] fuse_constraints::NormalPriorCostFunctor / AbsoluteConstraint<V>: r = A (x - b); consts b[3], A[9]
template <typename T> inline void AbsVec3Residual(const double* c, const T* x, T r[3]) {
  T e[3] = {x[0] - T(c[0]), x[1] - T(c[1]), x[2] - T(c[2])};
  const double* A = c + 3;
  for (int i = 0; i < 3; ++i) r[i] = T(A[3 * i]) * e[0] + T(A[3 * i + 1]) * e[1] + T(A[3 * i + 2]) * e[2];
}
// [EXT] fuse_constraints::NormalDeltaCostFunctor / RelativeConstraint<V>: r = A ((x2 - x1) - d)
template <typename T> inline void RelVec3Residual(const double* c, const T* x1, const T* x2, T r[3]) {
  T e[3] = {x2[0] - x1[0] - T(c[0]), x2[1] - x1[1] - T(c[1]), x2[2] - x1[2] - T(c[2])};
  const double* A = c + 3;
  for (int i = 0; i < 3; ++i) r[i] = T(A[3 * i]) * e[0] + T(A[3 * i + 1]) * e[1] + T(A[3 * i + 2]) * e[2];
}

// A14: bs_constraints/include/bs_constraints/global/gravity_alignment_cost_functor.h:50-63
//   consts: g_b[3], A[4]
template <typename T> inline void GravityResidual(const double* c, const T* q, T r[2]) {
  T g[3] = {T(c[0]), T(c[1]), T(c[2])};
  T qq[4] = {q[0], q[1], q[2], q[3]};
  T gw[3];
  QuaternionRotatePoint(qq, g, gw);             // :58
  r[0] = T(c[3]) * gw[0] + T(c[4]) * gw[1];     // :60
  r[1] = T(c[5]) * gw[0] + T(c[6]) * gw[1];     // :61
}

}  // namespace bso
