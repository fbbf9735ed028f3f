// The reprojection factor's evaluation as a device function: reproj_eval_kernel (k_reproj.hip) and the launch that also carries the IMU
// factors of a visual-inertial window (visual_imu_eval_kernel, k_small.hip) both run it, 256 threads per block of 256 factors.
#pragma once
#include "bsgpu_device.h"

namespace bsg {

// ---------------------------------------------------------------------------------------------------
// reprojection residual + Jacobian.  One factor per lane; the 2x9 Jacobian of a wave's 64 factors is
// transposed through LDS so that the AoS rows leave as contiguous 16-byte-per-lane stores.
// Algorithmic bytes per factor: 16 (idx+meta) + 16 (pixel) + 8 (w) in, 16 (r) + 144 (J) out = 200.
// ---------------------------------------------------------------------------------------------------
// (the staging area is the caller's: the launch that also carries IMU factors lends the same bytes to their workgroups)
template <bool WITH_J> constexpr int kReprojStage = WITH_J ? 4 * 64 * 18 : 4;
template <bool WITH_J>
__device__ __forceinline__ void reproj_eval_body(const int block, int n, const int4* __restrict__ fac, const double2* __restrict__ pix,
                                                 const double* __restrict__ wgt, const double* __restrict__ x,
                                                 const DevCamera* __restrict__ cams, const DevLoss* __restrict__ losses,
                                                 double2* __restrict__ r_out, double* __restrict__ J_out, double* __restrict__ JB_out,
                                                 double* __restrict__ cost_part, int count_inactive, double* sJ /* kReprojStage<WITH_J> doubles of LDS, 16-byte aligned */) {
  __shared__ double sred[4];
  const int f = block * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double cost = 0.0;
  double J[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) J[i] = 0.0;
  if (f < n) {
    const int4 fc = fac[f];
    const double2 z = pix[f];
    const double w = wgt[f];
    const int cam_id = fc.w & ((1 << kMetaCamBits) - 1);
    const int loss_id = (fc.w >> kMetaCamBits) & ((1 << kMetaLossBits) - 1);
    const int flags = fc.w >> (kMetaCamBits + kMetaLossBits);
    const double* qp = x + fc.x;
    const double* tp = x + fc.y;
    const double* Pp = x + fc.z;
    const double q[4] = {qp[0], qp[1], qp[2], qp[3]};
    const double t[3] = {tp[0], tp[1], tp[2]};
    const double P[3] = {Pp[0], Pp[1], Pp[2]};
    const DevCamera cam = cams[cam_id];
    double R[9];
    quat_to_rot(q, R);
    // P_b = R^T P - R^T t  (function.h:81-82)
    double a[3], b[3];
    mat3t_vec(R, P, a);
    mat3t_vec(R, t, b);
    const double Pb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double Pc[3];
    mat3_vec(cam.R, Pb, Pc);
    Pc[0] += cam.t[0]; Pc[1] += cam.t[1]; Pc[2] += cam.t[2];
    // (K P_c).hnormalized()
    const double iz = 1.0 / Pc[2];
    const double u = (cam.fx * Pc[0] + cam.cx * Pc[2]) * iz;
    const double v = (cam.fy * Pc[1] + cam.cy * Pc[2]) * iz;
    double r0 = w * (z.x - u), r1 = w * (z.y - v);
    const double s = r0 * r0 + r1 * r1;
    double rho1;
    const double rho = loss_eval(losses[loss_id], s, &rho1);
    const double sc = sqrt(rho1);
    const bool active = flags != (kFlagQConst | kFlagPConst | kFlagLConst);
    cost = (active != (count_inactive != 0)) ? 0.5 * rho : 0.0;   // count_inactive: the fixed-cost pass (all three blocks constant)
    if (WITH_J) {   // (a cost-only pass must not disturb r of the current point)
      typedef double d2_t __attribute__((ext_vector_type(2)));
      const d2_t rv = {r0 * sc, r1 * sc};
      __builtin_nontemporal_store(rv, reinterpret_cast<d2_t*>(r_out) + f);
    }
    if (WITH_J) {
      // Jpi (jacobians.cpp:202-214), M = Jpi R_cb (2x3), scaled by the corrector and the weight
      const double jx0 = cam.fx * iz, jx2 = -cam.fx * Pc[0] * iz * iz;
      const double jy1 = cam.fy * iz, jy2 = -cam.fy * Pc[1] * iz * iz;
      const double ws = w * sc;
      double M[6];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        M[j] = ws * (jx0 * cam.R[j] + jx2 * cam.R[6 + j]);
        M[3 + j] = ws * (jy1 * cam.R[3 + j] + jy2 * cam.R[6 + j]);
      }
      // d/dtheta = -M [P_b]x
      if (!(flags & kFlagQConst)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double m0 = M[3 * i], m1 = M[3 * i + 1], m2 = M[3 * i + 2];
          J[9 * i + 0] = -(m1 * Pb[2] - m2 * Pb[1]);
          J[9 * i + 1] = -(m2 * Pb[0] - m0 * Pb[2]);
          J[9 * i + 2] = -(m0 * Pb[1] - m1 * Pb[0]);
        }
      }
      // d/dt = +M R^T ; d/dP = -M R^T
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const double mr = M[3 * i] * R[3 * j] + M[3 * i + 1] * R[3 * j + 1] + M[3 * i + 2] * R[3 * j + 2];
          J[9 * i + 3 + j] = (flags & kFlagPConst) ? 0.0 : mr;
          J[9 * i + 6 + j] = (flags & kFlagLConst) ? 0.0 : -mr;
        }
    }
  }
  const double tot = block_sum_256(cost, sred);
  if (threadIdx.x == 0) cost_part[block] = tot;
  if (WITH_J) {
    double* sw = sJ + wave * (64 * 18);
#pragma unroll
    // stored row: [A row 0 (q, p: 6) | A row 1 (6) | B row 0 (landmark: 3) | B row 1 (3)] — the pose part contiguous for the pair
    // kernel and the back-substitution, the landmark part contiguous for the landmark kernel (J[] above is [q p l | q p l])
    for (int i = 0; i < 18; ++i) { const int rw = i / 9, cl = i % 9; sw[lane * 18 + (cl < 6 ? 6 * rw + cl : 12 + 3 * rw + (cl - 6))] = J[i]; }
    __syncthreads();
    const int fb = block * 256 + wave * 64;
    const int cnt = min(64, n - fb);
    if (cnt > 0) {
      // two contiguous streams per wave: 96 B of pose part and 48 B of landmark part per factor, 16 B per lane and store
      typedef double d2_t __attribute__((ext_vector_type(2)));
      d2_t* dstA = reinterpret_cast<d2_t*>(J_out + (size_t)fb * kJAStride);
      d2_t* dstB = reinterpret_cast<d2_t*>(JB_out + (size_t)fb * 6);
      const d2_t* src = reinterpret_cast<const d2_t*>(sw);
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int e = it * 64 + lane;
        if (e < cnt * 6) __builtin_nontemporal_store(src[(e / 6) * 9 + (e % 6)], &dstA[(e / 6) * (kJAStride / 2) + (e % 6)]);   // streamed: no cache holds 58 MB until the next kernel
      }
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int e = it * 64 + lane;
        if (e < cnt * 3) __builtin_nontemporal_store(src[(e / 3) * 9 + 6 + (e % 3)], &dstB[e]);
      }
    }
  }
}


}  // namespace bsg
