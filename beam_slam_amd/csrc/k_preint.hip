// IMU pre-integration for a batch of keyframe intervals — the factor producer on the input side of the solve
// (SURVEY.md §8f rank 4).  Restates bs_common::PreIntegrator (bs_common/src/bs_common/preintegrator.cpp):
//   Increment  :26-89   covariance propagation A P A^T + B Q B^T on the 9-d error state (q, p, v), bias random walk,
//                        bias Jacobians (order of the updates matters), mid-point state integration
//   Integrate  :91-115  consecutive samples up to t_end, then the remainder with the last sample
//   ComputeSqrtInvCov :117-143  norm guards, sqrt information = cov^-1 .llt().matrixU(), fallback weight
// and writes, per interval, the constant payload of BSGPU_F_IMU_DELTA (include/bsgpu.h): dt, dq, dp, dv, the five
// bias Jacobians, the bias linearisation point and A = info_weight * sqrt_inv_cov.
// One lane per interval: an interval is a strictly sequential recursion over its samples (20 at 200 Hz / 10 Hz),
// a window has a few hundred intervals, so this is latency work — it is here to keep the IMU samples and the
// factor constants on the device, not for throughput.
#include "bsgpu_device.h"

namespace bsg {

namespace {

struct M3 { double m[9]; };
BSG_DEV M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.0; return r; }
BSG_DEV M3 m3_eye() { M3 r = m3_zero(); r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
BSG_DEV M3 m3_mul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
BSG_DEV M3 m3_t(const M3& a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * j + i]; return r; }
BSG_DEV M3 m3_axpy(double s, const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = s * a.m[i] + b.m[i]; return r; }
BSG_DEV M3 m3_skew(const double v[3]) { M3 r = m3_zero(); r.m[1] = -v[2]; r.m[2] = v[1]; r.m[3] = v[2]; r.m[5] = -v[0]; r.m[6] = -v[1]; r.m[7] = v[0]; return r; }
// [EXT] beam::LieAlgebraToR (Rodrigues)
BSG_DEV M3 so3_exp(const double w[3]) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const M3 K = m3_skew(w), K2 = m3_mul(K, K);
  const double a = th < 1e-10 ? 1.0 : sin(th) / th, b = th < 1e-10 ? 0.5 : (1.0 - cos(th)) / (th * th);
  M3 r = m3_eye();
  for (int i = 0; i < 9; ++i) r.m[i] += a * K.m[i] + b * K2.m[i];
  return r;
}
// [EXT] beam::RightJacobianOfSO3
BSG_DEV M3 so3_jr(const double w[3]) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const M3 K = m3_skew(w), K2 = m3_mul(K, K);
  const double a = th < 1e-8 ? 0.5 : (1.0 - cos(th)) / (th * th), b = th < 1e-8 ? 1.0 / 6.0 : (th - sin(th)) / (th * th * th);
  M3 r = m3_eye();
  for (int i = 0; i < 9; ++i) r.m[i] += -a * K.m[i] + b * K2.m[i];
  return r;
}
BSG_DEV void quat_from_aa_unit(const double w[3], double q[4]) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (th < 1e-12) {
    q[0] = 1.0; q[1] = 0.5 * w[0]; q[2] = 0.5 * w[1]; q[3] = 0.5 * w[2];
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
  } else {
    const double s = sin(0.5 * th) / th;
    q[0] = cos(0.5 * th); q[1] = s * w[0]; q[2] = s * w[1]; q[3] = s * w[2];
  }
}

// in-place lower Cholesky of an n x n SPD matrix (row-major, pitch n); false on a non-positive pivot
template <int N> BSG_DEV bool chol_lower(double* A) {
  for (int j = 0; j < N; ++j) {
    double d = A[j * N + j];
    for (int k = 0; k < j; ++k) d -= A[j * N + k] * A[j * N + k];
    if (!(d > 0.0)) return false;
    const double l = sqrt(d);
    A[j * N + j] = l;
    for (int i = j + 1; i < N; ++i) {
      double s = A[i * N + j];
      for (int k = 0; k < j; ++k) s -= A[i * N + k] * A[j * N + k];
      A[i * N + j] = s / l;
    }
  }
  return true;
}

}  // namespace

__global__ __launch_bounds__(64) void preintegrate_kernel(int n_int, const int* __restrict__ sample_start, const double* __restrict__ ts,
                                                          const double* __restrict__ wm, const double* __restrict__ am,
                                                          const double* __restrict__ t_end, const double* __restrict__ bgs,
                                                          const double* __restrict__ bas, const double* __restrict__ covs /* cov_w, cov_a, cov_bg, cov_ba: 4 x 9 */,
                                                          double info_weight, double* __restrict__ out /* n_int x 287 */) {
  const int iv = blockIdx.x * 64 + threadIdx.x;
  if (iv >= n_int) return;
  const int s0 = sample_start[iv], s1 = sample_start[iv + 1];
  const double bg[3] = {bgs[3 * iv], bgs[3 * iv + 1], bgs[3 * iv + 2]}, ba[3] = {bas[3 * iv], bas[3 * iv + 1], bas[3 * iv + 2]};
  double dt_tot = 0.0, q[4] = {1, 0, 0, 0}, p[3] = {0, 0, 0}, v[3] = {0, 0, 0};
  double cov[15 * 15];
  for (int i = 0; i < 225; ++i) cov[i] = 0.0;
  M3 dq_dbg = m3_zero(), dp_dbg = m3_zero(), dp_dba = m3_zero(), dv_dbg = m3_zero(), dv_dba = m3_zero();
  M3 Cw, Ca, Cbg, Cba;
  for (int i = 0; i < 9; ++i) { Cw.m[i] = covs[i]; Ca.m[i] = covs[9 + i]; Cbg.m[i] = covs[18 + i]; Cba.m[i] = covs[27 + i]; }

  auto increment = [&](double dt, const double* wraw, const double* araw) {   // preintegrator.cpp:26-89
    const double w[3] = {wraw[0] - bg[0], wraw[1] - bg[1], wraw[2] - bg[2]}, a[3] = {araw[0] - ba[0], araw[1] - ba[1], araw[2] - ba[2]};
    const double wdt[3] = {w[0] * dt, w[1] * dt, w[2] * dt}, whalf[3] = {0.5 * wdt[0], 0.5 * wdt[1], 0.5 * wdt[2]};
    const M3 R_full = so3_exp(wdt), Jr = so3_jr(wdt), Sa = m3_skew(a);
    M3 Rdq; quat_to_rot(q, Rdq.m);
    const M3 RS = m3_mul(Rdq, Sa);
    // A (9x9) and B (9x6) of the error-state propagation; error-state order q(0) p(3) v(6)
    double A[81], B[54];
    for (int i = 0; i < 81; ++i) A[i] = 0.0;
    for (int i = 0; i < 54; ++i) B[i] = 0.0;
    for (int i = 0; i < 9; ++i) A[10 * i] = 1.0;
    const M3 Rt = m3_t(R_full);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      A[(0 + i) * 9 + j] = Rt.m[3 * i + j];
      A[(6 + i) * 9 + j] = -dt * RS.m[3 * i + j];
      A[(3 + i) * 9 + j] = -0.5 * dt * dt * RS.m[3 * i + j];
      A[(3 + i) * 9 + 6 + j] = (i == j) ? dt : 0.0;
      B[(0 + i) * 6 + j] = dt * Jr.m[3 * i + j];
      B[(6 + i) * 6 + 3 + j] = dt * Rdq.m[3 * i + j];
      B[(3 + i) * 6 + 3 + j] = 0.5 * dt * dt * Rdq.m[3 * i + j];
    }
    const double inv_dt = 1.0 / fmax(dt, 1.0e-7);
    // P9 <- A P9 A^T + B Q B^T,  Q = blkdiag(cov_w, cov_a) / dt
    double AP[81], P9[81];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { double s = 0.0; for (int k = 0; k < 9; ++k) s += A[i * 9 + k] * cov[k * 15 + j]; AP[i * 9 + j] = s; }
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { double s = 0.0; for (int k = 0; k < 9; ++k) s += AP[i * 9 + k] * A[j * 9 + k]; P9[i * 9 + j] = s; }
    double BQ[54];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 6; ++j) {
      double s = 0.0;
      const M3& Qb = j < 3 ? Cw : Ca;
      const int o = j < 3 ? 0 : 3;
      for (int k = 0; k < 3; ++k) s += B[i * 6 + o + k] * (Qb.m[3 * k + (j - o)] * inv_dt);
      BQ[i * 6 + j] = s;
    }
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { double s = 0.0; for (int k = 0; k < 6; ++k) s += BQ[i * 6 + k] * B[j * 6 + k]; cov[i * 15 + j] = P9[i * 9 + j] + s; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { cov[(9 + i) * 15 + 9 + j] += dt * Cbg.m[3 * i + j]; cov[(12 + i) * 15 + 12 + j] += dt * Cba.m[3 * i + j]; }
    // bias Jacobians (:69-80, this order)
    const M3 RSdq = m3_mul(RS, dq_dbg);
    dp_dbg = m3_axpy(-0.5 * dt * dt, RSdq, m3_axpy(dt, dv_dbg, dp_dbg));
    dp_dba = m3_axpy(-0.5 * dt * dt, Rdq, m3_axpy(dt, dv_dba, dp_dba));
    dv_dbg = m3_axpy(-dt, RSdq, dv_dbg);
    dv_dba = m3_axpy(-dt, Rdq, dv_dba);
    dq_dbg = m3_axpy(-dt, Jr, m3_mul(Rt, dq_dbg));
    // state (:82-88)
    double qh[4], qm[4], qf[4], qn[4];
    quat_from_aa_unit(whalf, qh);
    quat_mul(q, qh, qm);
    M3 Rm; quat_to_rot(qm, Rm.m);
    double amid[3];
    mat3_vec(Rm.m, a, amid);
    dt_tot += dt;
    for (int i = 0; i < 3; ++i) p[i] += dt * v[i] + 0.5 * dt * dt * amid[i];
    for (int i = 0; i < 3; ++i) v[i] += dt * amid[i];
    quat_from_aa_unit(wdt, qf);
    quat_mul(q, qf, qn);
    const double nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; ++i) q[i] = qn[i] / nn;
  };

  const double te = t_end[iv];
  for (int s = s0; s + 1 < s1; ++s) {                       // :96-102
    if (ts[s + 1] > te + 1e-12) break;
    increment(ts[s + 1] - ts[s], wm + 3 * s, am + 3 * s);
  }
  if (s1 > s0) {                                            // :104-108 remainder with the last sample
    const double dt = te - ts[s1 - 1];
    if (dt > 1e-12) increment(dt, wm + 3 * (s1 - 1), am + 3 * (s1 - 1));
  }
  // ComputeSqrtInvCov (:117-143)
  double n9 = 0.0, n6 = 0.0;
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) n9 += cov[i * 15 + j] * cov[i * 15 + j];
  for (int i = 9; i < 15; ++i) for (int j = 9; j < 15; ++j) n6 += cov[i * 15 + j] * cov[i * 15 + j];
  if (sqrt(n9) < 1e-5) for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) cov[i * 15 + j] = (i == j) ? 1e-5 : 0.0;
  if (sqrt(n6) < 1e-9) for (int i = 9; i < 15; ++i) for (int j = 9; j < 15; ++j) cov[i * 15 + j] = (i == j) ? 1e-9 : 0.0;
  // info = cov^-1 through cov = L L^T;  U = chol(info)^T  (cov.inverse().llt().matrixU())
  double L[225], Li[225];
  for (int i = 0; i < 225; ++i) L[i] = cov[i];
  bool ok = chol_lower<15>(L);
  double U[225];
  for (int i = 0; i < 225; ++i) U[i] = 0.0;
  if (ok) {
    for (int i = 0; i < 225; ++i) Li[i] = 0.0;
    for (int c = 0; c < 15; ++c)                      // Li = L^-1 (lower)
      for (int i = c; i < 15; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) s -= L[i * 15 + k] * Li[k * 15 + c];
        Li[i * 15 + c] = s / L[i * 15 + i];
      }
    double info[225];
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { double s = 0.0; for (int k = (i > j ? i : j); k < 15; ++k) s += Li[k * 15 + i] * Li[k * 15 + j]; info[i * 15 + j] = s; }
    ok = chol_lower<15>(info);
    if (ok) for (int i = 0; i < 15; ++i) for (int j = 0; j <= i; ++j) { U[j * 15 + i] = info[i * 15 + j]; if (!isfinite(info[i * 15 + j])) ok = false; }
  }
  if (!ok) for (int i = 0; i < 225; ++i) U[i] = (i % 16 == 0) ? 1e-4 : 0.0;    // invalid_inv_cov_weight_ * I
  double* o = out + (size_t)iv * 287;
  o[0] = dt_tot;
  for (int i = 0; i < 4; ++i) o[1 + i] = q[i];
  for (int i = 0; i < 3; ++i) { o[5 + i] = p[i]; o[8 + i] = v[i]; }
  for (int i = 0; i < 9; ++i) { o[11 + i] = dq_dbg.m[i]; o[20 + i] = dp_dbg.m[i]; o[29 + i] = dp_dba.m[i]; o[38 + i] = dv_dbg.m[i]; o[47 + i] = dv_dba.m[i]; }
  for (int i = 0; i < 3; ++i) { o[56 + i] = bg[i]; o[59 + i] = ba[i]; }
  for (int i = 0; i < 225; ++i) o[62 + i] = info_weight * U[i];
}

void launch_preintegrate(hipStream_t s, int n_int, const int* sample_start, const double* ts, const double* wm, const double* am,
                         const double* t_end, const double* bg, const double* ba, const double* covs, double info_weight, double* out) {
  if (n_int > 0)
    hipLaunchKernelGGL(preintegrate_kernel, dim3((n_int + 63) / 64), dim3(64), 0, s, n_int, sample_start, ts, wm, am, t_end, bg, ba, covs,
                       info_weight, out);
}

}  // namespace bsg
