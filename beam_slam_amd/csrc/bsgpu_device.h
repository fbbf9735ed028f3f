// Device-side math shared by the HIP kernels: quaternion / SO(3) helpers, robust losses, reductions.
// All double precision (the reference path is IEEE double end to end, SURVEY.md "Conventions").
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bsgpu.h"
#include "bsgpu_internal.h"
#include "lm_decide.h"

namespace bsg {

#define BSG_DEV __device__ __forceinline__

// Eigen::Quaternion::toRotationMatrix() (no normalisation) — what the reprojection factor uses
// (euclidean_reprojection_function.h:68-70).  R row-major.
BSG_DEV void quat_to_rot(const double q[4], double R[9]) {
  const double tx = 2.0 * q[1], ty = 2.0 * q[2], tz = 2.0 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
// rotation matrix of the normalised quaternion (ceres::QuaternionRotatePoint semantics)
BSG_DEV void quat_to_rot_normalized(const double q[4], double R[9]) {
  const double s = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double u[4] = {q[0] * s, q[1] * s, q[2] * s, q[3] * s};
  quat_to_rot(u, R);
}
BSG_DEV void quat_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
BSG_DEV void mat3_vec(const double M[9], const double v[3], double o[3]) {
  o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
BSG_DEV void mat3t_vec(const double M[9], const double v[3], double o[3]) {
  o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}
BSG_DEV void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// ceres::AngleAxisToQuaternion
BSG_DEV void angle_axis_to_quat(const double aa[3], double q[4]) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > 0.0) {
    const double th = sqrt(th2), half = th * 0.5;
    double sn, cs;
    sincos(half, &sn, &cs);
    const double k = sn / th;
    q[0] = cs; q[1] = aa[0] * k; q[2] = aa[1] * k; q[3] = aa[2] * k;
  } else {
    q[0] = 1.0; q[1] = aa[0] * 0.5; q[2] = aa[1] * 0.5; q[3] = aa[2] * 0.5;
  }
}
// ceres::QuaternionToAngleAxis (angle in (-pi, pi]; q and -q agree; scale invariant)
BSG_DEV void quat_to_angle_axis(const double q[4], double aa[3]) {
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double sn = sqrt(s2), cs = q[0];
    const double two_theta = 2.0 * ((cs < 0.0) ? atan2(-sn, -cs) : atan2(sn, cs));
    const double k = two_theta / sn;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
  }
}
// inverse right Jacobian of SO(3): Log(Exp(e) Exp(d)) ~= e + Jr^-1(e) d.   Row-major 3x3.
BSG_DEV void so3_jr_inv(const double e[3], double J[9]) {
  const double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  double c;
  if (th2 < 1e-6) {
    c = 1.0 / 12.0 + th2 / 720.0;
  } else {
    const double th = sqrt(th2);
    double sn, cs;
    sincos(th, &sn, &cs);
    c = 1.0 / th2 - (1.0 + cs) / (2.0 * th * sn);
  }
  // I + 0.5 [e]x + c [e]x^2, [e]x^2 = e e^T - th2 I
  J[0] = 1.0 + c * (e[0] * e[0] - th2); J[1] = -0.5 * e[2] + c * e[0] * e[1]; J[2] = 0.5 * e[1] + c * e[0] * e[2];
  J[3] = 0.5 * e[2] + c * e[1] * e[0];  J[4] = 1.0 + c * (e[1] * e[1] - th2); J[5] = -0.5 * e[0] + c * e[1] * e[2];
  J[6] = -0.5 * e[1] + c * e[2] * e[0]; J[7] = 0.5 * e[0] + c * e[2] * e[1];  J[8] = 1.0 + c * (e[2] * e[2] - th2);
}

// ceres::LossFunction::Evaluate: returns rho(s), sets *rho1 = rho'(s)
BSG_DEV double loss_eval(const DevLoss& L, double s, double* rho1) {
  if (L.kind == BSGPU_LOSS_CAUCHY) {
    const double b = L.a * L.a, c = 1.0 / b;
    const double sum = 1.0 + s * c, inv = 1.0 / sum;
    *rho1 = fmax(2.2250738585072014e-308, inv);
    return b * log(sum);
  } else if (L.kind == BSGPU_LOSS_HUBER) {
    const double b = L.a * L.a;
    if (s > b) {
      const double r = sqrt(s);
      *rho1 = fmax(2.2250738585072014e-308, L.a / r);
      return 2.0 * L.a * r - b;
    }
  }
  *rho1 = 1.0;
  return s;
}

// LM diagonal of pose-side column at solver position i (k_dense.hip pose_diag_kernel, and fused with the gradient norms in
// k_misc.hip): Jacobi scale / clamped diagonal where H_jj is known, unit pivots on the rhs row and the padding
BSG_DEV void pose_diag_element(int i, int n_pose, double* __restrict__ S, int ld, const double* __restrict__ hdiag, double inv_radius,
                               int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* __restrict__ scale,
                               double* __restrict__ dcl, const int* __restrict__ iperm) {
  const int j = iperm[i];                                 // tangent index, or -1 on the padding and the rhs tile
  if (j >= 0) {
    const double h = hdiag[j];
    double sc = compute_scale ? (jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0) : scale[j];
    double d = compute_dcl ? fmin(fmax(sc * sc * h, lm_lo), lm_hi) / (sc * sc) : dcl[j];
    if (compute_scale) scale[j] = sc;
    if (compute_dcl) dcl[j] = d;
    S[(size_t)i * ld + i] += d * inv_radius;
  } else {
    S[(size_t)i * ld + i] = 1.0;  // rhs row / padding: unit pivot, never used as a real pivot
  }
}

// the LM diagonal of tangent column j, times 1 / radius (pose_diag_element without the store into S); writes the Jacobi scale / clamped
// diagonal when this step (re)computes them
BSG_DEV double lm_diag_value(int j, const LmDiag& D) {
  const double h = D.hdiag[j];
  const double sc = D.compute_scale ? (D.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0) : D.scale[j];
  const double d = D.compute_dcl ? fmin(fmax(sc * sc * h, D.lm_lo), D.lm_hi) / (sc * sc) : D.dcl[j];
  if (D.compute_scale) D.scale[j] = sc;
  if (D.compute_dcl) D.dcl[j] = d;
  return d * D.inv_radius;
}
BSG_DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// 64 wave-wide sums at once: on entry every lane holds its own v[0..63]; on exit lane l holds in v[0] the sum over all lanes of
// v[l].  Each butterfly step exchanges the half of the values the lane does not keep, so the whole reduction costs 63
// exchange-adds per lane instead of 64 x 6 for 64 separate wave_sum()s — and pairs the lanes exactly as wave_sum() does
// (l with l ^ 32, then ^ 16, ...), i.e. it produces the same bits.
BSG_DEV void wave_sum_transpose64(double (&v)[64]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 32, h = 32; o > 0; o >>= 1, h >>= 1) {
    const bool upper = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const double lo = v[i], hi = v[i + h];
      const double recv = __shfl_xor(upper ? lo : hi, o, 64);
      v[i] = (upper ? hi : lo) + recv;
    }
  }
}
// x (+) delta on one block: fuse's Orientation3DLocalParameterization::Plus on quaternion blocks (bs_constraints/src/jacobians.cpp:24-35:
// x (x) AngleAxisToQuaternion(delta), right perturbation), a sum elsewhere
BSG_DEV void block_plus(int manifold, int size, const double* x, const double* d, double* out) {
  if (manifold == BSGPU_MANIFOLD_QUAT_RIGHT) {
    double qd[4];
    angle_axis_to_quat(d, qd);
    const double q[4] = {x[0], x[1], x[2], x[3]};
    quat_mul(q, qd, out);
  } else {
    for (int i = 0; i < size; ++i) out[i] = x[i] + d[i];
  }
}
// the candidate of block b (a constant block: a copy) and its terms of |x_cand - x|^2 and |x|^2
BSG_DEV void update_block(int b, const int* __restrict__ xoff, const int* __restrict__ toff, const unsigned char* __restrict__ size,
                          const unsigned char* __restrict__ manifold, const double* __restrict__ x, const double* __restrict__ delta,
                          double* __restrict__ x_cand, double& d2, double& x2) {
  const int o = xoff[b], t = toff[b], sz = size[b];
  if (t >= 0) {
    double out[4];
    double xin[4] = {0, 0, 0, 0}, din[4] = {0, 0, 0, 0};
    const int ts = (manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : sz;
    for (int i = 0; i < sz && i < 4; ++i) xin[i] = x[o + i];
    for (int i = 0; i < ts && i < 4; ++i) din[i] = delta[t + i];
    block_plus(manifold[b], sz, xin, din, out);
    for (int i = 0; i < sz && i < 4; ++i) {
      x_cand[o + i] = out[i];
      const double df = xin[i] - out[i];
      d2 += df * df; x2 += xin[i] * xin[i];
    }
  } else {
    for (int i = 0; i < sz; ++i) x_cand[o + i] = x[o + i];
  }
}
// sum over a 256-thread block; result valid in thread 0
BSG_DEV double block_sum_256(double v, double* smem /* >= 4 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = smem[0] + smem[1] + smem[2] + smem[3];
  __syncthreads();
  return t;
}

// The end-of-step reduction (k_misc.hip final_reduce_kernel): unit `slot` adds up, in a fixed order, every partial array registered for that
// scalar; unit n_slots mirrors the scalars earlier kernels of the step produced.  The host does not wait for an event behind the launch
// (recording one costs the next kernel ~6 us of dispatch bubble): the LAST unit to finish — all host-side writes of a unit are thread 0's,
// fenced at system scope before it takes its ticket — stamps the mirror with the launch's sequence number, which the host polls
// (bsgpu_solve.cpp: fetch_scalars).
// LmState::advance's decision for the step this reduction closes: lm_decide.h (host- and device-compilable; tested on the CPU against LmState itself)
BSG_DEV void lm_load_scalars(const double* scal, LmScal& v) {
  auto ld = [&](int i) { return __hip_atomic_load(&scal[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  v.mcc = ld(SC_MCC); v.sn2 = ld(SC_STEP_NORM2); v.xn2 = ld(SC_X_NORM2); v.cand = ld(SC_COST_CAND); v.cost_x = ld(SC_COST_X); v.gmax = ld(SC_GRAD_MAX);
  v.chol_fail = ld(SC_CHOL_FAIL_SEEN);
}
// ... and where it goes: the workgroups of this launch that wait for it and the launches behind it read ONE word per copy (the next radius,
// negative when the step was not accepted; a waiting wave looks for anything but zero) — relaxed write-through stores and no fence (a release
// at agent scope writes the L2 back: 64 of them made a C2 iteration 1.6 ms longer); the other bank, the next deciding launch's, is cleared
// for it; the host's mirror gets both answers ahead of the stamp
BSG_DEV double lm_decide_word(const ReduceRide& R, const LmScal& v) {
  double radius;
  const int go = lm_decide(R.lmd, v, &radius);
  if (R.host_scal) { R.host_scal[SC_DEC_GO] = go ? 1.0 : 0.0; R.host_scal[SC_DEC_RADIUS] = radius; }
  return go ? radius : -R.lmd.radius;
}
// lanes [lane0, lane0 + n_lanes) store the copies
BSG_DEV void lm_publish_word(const ReduceRide& R, double word, int lane, int n_lanes) {
  for (int k = lane; k < kDecSlots; k += n_lanes) __hip_atomic_store(&R.dec[(size_t)k * kDecStride], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int k = lane; k < kDecSlots; k += n_lanes) __hip_atomic_store(&R.dec_next[(size_t)k * kDecStride], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// decided: this unit was the reduction's last by construction (ReduceRide::only_slot: every other unit ran in an earlier launch), held the
// step's scalars already and has decided before anything else (final_reduce_unit)
BSG_DEV void final_reduce_done(const ReduceRide& R, int n_units, bool decided = false) {
  double* host_scal = R.host_scal;
  int* counter = R.counter;
  if (!counter) return;
  const bool decides = R.lmd.on && R.dec != nullptr && !decided;
  __threadfence_system();
  const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (prev == n_units - 1) {
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (decides) {   // (every unit's scalar is in R.scal: each fenced before it took its ticket)
      LmScal v;
      lm_load_scalars(R.scal, v);
      lm_publish_word(R, lm_decide_word(R, v), 0, 1);
    }
    __threadfence_system();
    if (host_scal) __hip_atomic_store(&host_scal[SC_SEQ], R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// a wave of an assembly ahead waits for the decision of the reduction riding in its launch (copy `wave & 63`): the radius, or 0 when the
// step was not accepted (or nothing came: bounded by the wall clock)
BSG_DEV double wait_decision(const double* dec, int copy) {
  const double* w = dec + (size_t)(copy & (kDecSlots - 1)) * kDecStride;
  const unsigned long long t0 = wall_clock64();
  for (int spin = 0;; ++spin) {
    const double v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != 0.0) return v > 0.0 ? v : 0.0;
    __builtin_amdgcn_s_sleep(4);
    if ((spin & 255) == 255 && wall_clock64() - t0 > 200000000ull) return 0.0;   // (2 s at 100 MHz)
  }
}
// NT threads (1024, or 256 = a rider of a 256-thread launch): a thread plays 1024 / NT of the kernel's 1024 threads one after the other,
// and the waves' sums land in the same sixteen slots — the same additions in the same order, the same bits, whatever NT.  sred: 16 doubles.
template <int NT>
BSG_DEV void final_reduce_unit(int slot, int tid, const ReduceRide& R, int n_units, double* sred) {
  constexpr int VT = 1024 / NT;
  if (slot == R.defer_slot) return;   // (a later launch runs this unit — the last of the reduction: ReduceRide::only_slot there)
  // (that launch's unit: it will decide, and asks for the other units' scalars — an earlier launch's — with its first loads)
  // (they wait in LDS: held in registers over the sums they cost the carrying kernel a wave of occupancy per SIMD)
  const bool early = R.lmd.on && R.dec != nullptr && R.only_slot >= 0 && R.counter != nullptr;
  __shared__ LmScal s_pre;
  if (early && tid == 0) { LmScal pre; lm_load_scalars(R.scal, pre); s_pre = pre; }
  if (slot == R.n_slots) {   // the scalars earlier kernels of the step produced (gradient norms, Cholesky flag, ...) -> host mirror
    if (tid == 0) {
      if (R.host_scal) for (int i = R.n_slots; i < SC_SEQ; ++i) R.host_scal[i] = R.scal[i];
      // (mirrored: the factorisation's flag of this step is cleared here for the next one — its clearing may have run already, in the
      //  launch that carried the candidate update, bsgpu_solve.cpp)
      if (R.lmd.on) R.scal[SC_CHOL_FAIL_SEEN] = R.scal[SC_CHOL_FAIL];
      if (R.host_scal) R.scal[SC_CHOL_FAIL] = 0.0;
      final_reduce_done(R, n_units);
    }
    return;
  }
  if (slot == R.skip_slot) { if (tid == 0) final_reduce_done(R, n_units); return; }
  double acc[VT];
#pragma unroll
  for (int q = 0; q < VT; ++q) acc[q] = 0.0;
  bool any = false, is_max = false;
  // (the table of partial arrays goes through LDS in one trip: walked in memory it was a dependent scalar load per entry — thirteen on C3 —
  //  in front of the host's stamp)
  constexpr int kEntMax = 64;
  static_assert(sizeof(ReduceEntry) == 32, "ReduceEntry");
  __shared__ __attribute__((aligned(16))) unsigned long long s_ent_raw[kEntMax * 4];   // (ReduceEntry has a member initialiser: raw storage)
  ReduceEntry* s_ent = reinterpret_cast<ReduceEntry*>(s_ent_raw);
  // The deciding unit of a split reduction (ReduceRide::only_slot) with its arrays in the arguments: every load it needs leaves NOW — the seven
  // scalars above, the one larger array's sixteen strides, a value of each small array — and the sums are formed as the walk below forms them
  // (an array of <= 4 096 values: pairwise / in turn per virtual thread; of <= NT values: the value itself, to virtual thread `tid`).  With the
  // table staged and three arrays walked one after the other the word left 14 us into a 16 us landmark launch: four round trips in the
  // launch's own 52 MB.
  const bool from_args = early && R.n_early > 0 && R.n_early <= 4;
  if (from_args) {
    double pv[VT][4], tv[4];
    int big = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tv[k] = 0.0;
      if (k >= R.n_early) continue;
      const ReduceEntry& en = R.early[k];
      if (en.n > NT) {
        big = k;
#pragma unroll
        for (int q = 0; q < VT; ++q)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = tid + NT * q + 1024 * u;
            pv[q][u] = idx < en.n ? en.ptr[(size_t)idx * en.stride + en.offset] : 0.0;
          }
      } else
        tv[k] = tid < en.n ? en.ptr[(size_t)tid * en.stride + en.offset] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= R.n_early) continue;
      if (k == big) {
        const int nb = R.early[k].n;
#pragma unroll
        for (int q = 0; q < VT; ++q) {
          const bool all4 = tid + NT * q + 3 * 1024 < nb;
          const double pairwise = (pv[q][0] + pv[q][1]) + (pv[q][2] + pv[q][3]), in_turn = (pv[q][0] + pv[q][1]) + pv[q][2];
          acc[q] += all4 ? pairwise : in_turn;
        }
      } else
        acc[0] += tv[k];
    }
    any = true;
  }
  const bool staged = !from_args && R.n_entries <= kEntMax;
  if (staged) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(R.entries);
    for (int e = tid; e < 4 * R.n_entries; e += NT) s_ent_raw[e] = src[e];
    __syncthreads();
  }
  for (int e = 0; e < (from_args ? 0 : R.n_entries); ++e) {
    const ReduceEntry en = staged ? s_ent[e] : R.entries[e];
    if (en.slot != slot) continue;
    any = true;
    if (en.op == 1) is_max = true;
    if (en.op == 0 && en.n <= 4 * 1024) {
      // (the usual size — a partial sum per workgroup of the producing launch: every stride of the 1024 virtual threads asked for at once, ONE
      //  memory round trip per array instead of one per stride and virtual thread (seven in a row for C2's 1 566 cost partials), and the sums
      //  formed exactly as the loops below form them: four strides pairwise when all four exist, else one after the other)
      double pv[VT][4];
#pragma unroll
      for (int q = 0; q < VT; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = tid + NT * q + 1024 * u;
          pv[q][u] = idx < en.n ? en.ptr[(size_t)idx * en.stride + en.offset] : 0.0;
        }
#pragma unroll
      for (int q = 0; q < VT; ++q) {
        const bool all4 = tid + NT * q + 3 * 1024 < en.n;
        const double pairwise = (pv[q][0] + pv[q][1]) + (pv[q][2] + pv[q][3]), in_turn = (pv[q][0] + pv[q][1]) + pv[q][2];
        acc[q] += all4 ? pairwise : in_turn;
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < VT; ++q) {
      // 1024 (virtual) threads, four independent partial sums each: with one, every load waits for the previous add — the per-factor array
      // of a 20 000-factor group then costs 80 dependent round trips on 256 threads (115 us on C3) instead of 5 (a 70 000-factor
      // inverse-depth group: 22 us on 256 threads with eight partial sums)
      double a[4] = {0, 0, 0, 0};
      int i = tid + NT * q;
      for (; i + 3 * 1024 < en.n; i += 4 * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += en.ptr[(size_t)(i + 1024 * u) * en.stride + en.offset];
      }
      if (en.op == 1) {   // (a maximum: one entry per slot)
        for (; i < en.n; i += 1024) acc[q] = fmax(acc[q], en.ptr[(size_t)i * en.stride + en.offset]);
        continue;
      }
      for (; i < en.n; i += 1024) a[0] += en.ptr[(size_t)i * en.stride + en.offset];
      acc[q] += (a[0] + a[1]) + (a[2] + a[3]);
    }
  }
#pragma unroll
  for (int q = 0; q < VT; ++q) {
    double v = acc[q];
    if (is_max) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    } else {
      v = wave_sum(v);
    }
    if ((tid & 63) == 0) sred[(tid >> 6) + (NT / 64) * q] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t = is_max ? fmax(t, sred[w]) : t + sred[w];
    if (any) R.scal[slot] = t;
    // mirror in pinned host memory: the host reads the step's scalars right after the stream drains, without a
    // device-to-host copy of its own on the dependent path
    if (R.host_scal) R.host_scal[slot] = any ? t : R.scal[slot];
    if (early) {
      if (any && slot == SC_COST_CAND) s_pre.cand = t;
      if (any && slot == SC_COST_X) s_pre.cost_x = t;
      s_pre.word = lm_decide_word(R, s_pre);
    } else
      final_reduce_done(R, n_units);
  }
  if (early) {   // (the copies leave from a whole wave, then the mirror's fences and stamp)
    __syncthreads();
    if (tid < 64) lm_publish_word(R, s_pre.word, tid, 64);
    if (tid == 0) final_reduce_done(R, n_units, true);
  }
}

// One unit of 256 blocks of the gradient norms (grad_norms_kernel): max |x (+) (-g) - x| and its squared sum over the unit's blocks into
// gpart[2 unit], gpart[2 unit + 1].  NT threads (a multiple of 64, >= 256; the threads beyond 256 only take part in the reduction);
// sred / smax: NT / 64 doubles of LDS each.  Ends with every thread past a __syncthreads().
template <int NT>
BSG_DEV void grad_norms_unit(int unit, int tid, const GradNormRide& G, double* sred, double* smax) {
  const int b = unit * 256 + tid;
  double mx = 0.0, s2 = 0.0;
  if (tid < 256 && b < G.nb) {
    const int o = G.xoff[b], t = G.toff[b], sz = G.size[b];
    if (t >= 0) {
      const int mf = G.manifold[b];
      const int ts = (mf == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : sz;
      double xin[4] = {0, 0, 0, 0}, din[4] = {0, 0, 0, 0}, out[4];
      for (int i = 0; i < sz && i < 4; ++i) xin[i] = G.x[o + i];
      for (int i = 0; i < ts && i < 4; ++i) din[i] = -G.grad[t + i];
      block_plus(mf, sz, xin, din, out);
      for (int i = 0; i < sz && i < 4; ++i) {
        const double df = fabs(xin[i] - out[i]);
        mx = fmax(mx, df); s2 += df * df;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) { smax[tid >> 6] = mx; sred[tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    // (fixed order: the waves of the first 256 threads hold everything, as in grad_norms_kernel — the same bits)
    double m = 0.0;
    for (int w = 0; w < 4; ++w) m = fmax(m, smax[w]);
    G.gpart[2 * unit] = m;
    G.gpart[2 * unit + 1] = sred[0] + sred[1] + sred[2] + sred[3];
  }
  __syncthreads();
}

// assembly of ONE factor of a pose-only group into the dense reduced system by the calling workgroup (`nthr` threads): J staged in LDS
// (sJ >= 15 * 30 doubles, sr >= 15, st >= 60 ints), lanes stride over the (column a, column b) pairs; FP64 atomics into S, the rhs row,
// grad and hdiag.  `unit` = workgroup index over the set (SmallGroupSet::first).  Shared by small_assemble_kernel (a launch of its own)
// and pairs_kernel (whose extra workgroups do this work underneath the camera pairs: one launch less on the dependent path).
// (part / parts: the factor's column pairs are shared out among `parts` workgroups — the riders of the pair launch are single waves, and a
//  window of the reference's size waits for the slowest of them: 900 column pairs of an IMU factor on 64 lanes)
BSG_DEV void small_assemble_unit(const SmallGroupSet& set, int unit, int lane, int nthr, double* sJ, double* sr, int* st /* 60 ints */, double* __restrict__ S, int ld,
                                 int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int part = 0, int parts = 1) {
  if (unit >= set.first[set.n]) return;   // (padding of the caller's grid)
  int gi = 0;
  while (gi + 1 < set.n && unit >= set.first[gi + 1]) ++gi;
  const SmallGroup& g = set.g[gi];
  const int f = unit - set.first[gi];
  const int m = g.m, tw = 3 * g.nv, mt = m * tw;
  // every load of the factor is asked for before any is waited for (a staging loop of load, wait, store per pass was eight dependent trips
  // for a single wave, and a trip through perm[] per column pair after it): the rows, the residual, the flag; then the columns' tangent
  // indices and their places in the reduced system, which go to LDS with the rows — the sums below touch no table in memory
  const unsigned char act = g.active[f];
  const double* J = g.J + (size_t)f * mt;
  constexpr int kLoads = 8;   // (15 x 30 doubles on 64 lanes)
  double v[kLoads];
#pragma unroll
  for (int it = 0; it < kLoads; ++it) v[it] = J[min(lane + it * nthr, mt - 1)];
  const double rv = g.r[(size_t)f * m + min(lane, m - 1)];
  const int wcut = 3 * (g.nv - 1) + g.w_last;   // columns >= wcut are the padding of a narrow last slot
  int ta = -1, pa = 0;
  if (lane < tw) {
    const int t = g.toff[(size_t)f * g.nv + lane / 3];
    ta = (t < 0 || lane >= wcut) ? -1 : t + lane % 3;
    pa = ta >= 0 ? perm[ta] : 0;
  }
  if (!act) return;
#pragma unroll
  for (int it = 0; it < kLoads; ++it) { const int i = lane + it * nthr; if (i < mt) sJ[i] = v[it]; }
  if (lane < m) sr[lane] = rv;
  if (lane < tw) { st[lane] = ta; st[30 + lane] = pa; }
  __syncthreads();
  for (int p = lane + nthr * part; p < tw * tw; p += nthr * parts) {   // (an IMU factor has 900 column pairs)
    const int a = p / tw, b = p % tw;
    if (st[a] < 0 || st[b] < 0) continue;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += sJ[k * tw + a] * sJ[k * tw + b];
    atomicAdd(&S[(size_t)st[30 + a] * ld + st[30 + b]], acc);
  }
  if (part != 0) return;
  for (int a = lane; a < wcut; a += nthr) {
    const int ra = st[a];
    if (ra < 0) continue;
    double gs = 0.0, hs = 0.0;
    for (int k = 0; k < m; ++k) { const double j = sJ[k * tw + a]; gs += j * sr[k]; hs += j * j; }
    atomicAdd(&S[(size_t)rhs_row * ld + st[30 + a]], gs);
    atomicAdd(&grad[ra], gs);
    atomicAdd(&hdiag[ra], hs);
  }
}

// model cost change term of pose-only groups, one lane per residual row: part[unit] = sum over the unit's 128 rows of
// -(J_k d) (r_k + J_k d / 2).  `unit` = 128-row unit over the set (SmallGroupSet::first), t128 = thread within the unit, s2 = 2 doubles
// of LDS private to the unit's two waves.  Shared by small_mcc_kernel and backsub_mcc_kernel (which runs the units as extra workgroups).
BSG_DEV void small_mcc_unit(const SmallGroupSet& set, int unit, int t128, const double* __restrict__ delta, double* s2) {
  int gi = 0;
  while (gi + 1 < set.n && unit >= set.first[gi + 1]) ++gi;
  const SmallGroup& g = set.g[gi];
  double* part = set.part[gi];
  const int wg = unit - set.first[gi];
  const int id = wg * 128 + t128;
  const int m = g.m, nv = g.nv, tw = 3 * nv;
  // The loads are asked for in rounds, not one slot after the other (flag, then per slot: offset, branch, three entries of J and of the
  // step — 13 dependent trips for a relative-pose row with extrinsics, 21 for an IMU row): the flag, the residual and every slot's
  // offset first; then, three slots at a time, nine entries of the row and the nine of the step they multiply.  Indices are clamped to
  // entries that exist and the values masked, so that no load sits behind a lane's branch.
  const bool valid = id < g.n * m;
  const int idc = valid ? id : 0;
  const int f = idc / m, k = idc - f * m;
  const unsigned char act = g.active[f];
  const int* __restrict__ to = g.toff + (size_t)f * nv;
  constexpr int kSlots = 12;   // (an IMU factor has ten)
  int t[kSlots];
#pragma unroll
  for (int sl = 0; sl < kSlots; ++sl) t[sl] = to[min(sl, nv - 1)];
  const double rk = g.r[(size_t)f * m + k];
  const double* __restrict__ J = g.J + ((size_t)f * m + k) * tw;
  double jv = 0.0;
#pragma unroll
  for (int c = 0; c < kSlots / 3; ++c) {
    if (3 * c >= nv) break;   // (uniform)
    double Jc[9], dc[9];
    bool on[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sl = 3 * c + j;
      const bool slot_on = sl < nv && t[sl] >= 0;
      const int w = sl == nv - 1 ? g.w_last : 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        on[3 * j + i] = slot_on && i < w;
        Jc[3 * j + i] = J[min(3 * sl + i, tw - 1)];
        dc[3 * j + i] = delta[on[3 * j + i] ? t[sl] + i : 0];
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      jv += (on[3 * j] ? Jc[3 * j] * dc[3 * j] : 0.0) + (on[3 * j + 1] ? Jc[3 * j + 1] * dc[3 * j + 1] : 0.0) + (on[3 * j + 2] ? Jc[3 * j + 2] * dc[3 * j + 2] : 0.0);
  }
  const double acc = (valid && act) ? -jv * (rk + 0.5 * jv) : 0.0;
  const double w = wave_sum(acc);
  if ((t128 & 63) == 0) s2[t128 >> 6] = w;
  __syncthreads();
  if (t128 == 0) part[wg] = s2[0] + s2[1];
}

}  // namespace bsg
