// Device bodies of the dense linear prior factor ([EXT] fuse_constraints::MarginalConstraint; k_marg.hip has the arithmetic's description
// and the launches).  In a header since round 5: a window that carries such a prior evaluates, assembles and takes the model-cost terms of it
// inside the launches of its other pose-only factors (k_small.hip small_eval_set_marg_kernel, small_assemble_seg_marg_kernel; k_reproj.hip
// backsub_mcc_marg_kernel) — four dependent launches less per LM iteration, 4.6 - 7.4 us each on a window of the reference's size.
#pragma once
#include "bsgpu_device.h"

namespace bsg {

// one wave per residual row: wave w of an NT-thread workgroup takes row first_row + w.  The workgroup first forms delta = x [-] xbar and, for
// quaternion blocks, the scale |x|^2 of the tangent Jacobian itself, in LDS (a few dozen blocks: cheaper than a launch of its own in front of
// every evaluation), then r = b + A_row . delta, the cost term, and J_row = A_row D (constant columns zero)
constexpr int kMargColsLds = 1024;   // columns / blocks kept in LDS per workgroup; larger factors read the arrays a separate launch left
BSG_DEV void marg_delta_block(const MargDev& m, int i, const double* __restrict__ x, double* delta, double* D) {
  const int xo = m.blk_xoff[i], sz = m.blk_size[i], ct = m.blk_col[i], ca = m.blk_amb[i];
  if (m.blk_quat[i]) {
    const double* xb = m.xbar + ca;
    const double cj[4] = {xb[0], -xb[1], -xb[2], -xb[3]};   // QuaternionInverse = conjugate (jacobians.cpp:3-8)
    const double q[4] = {x[xo], x[xo + 1], x[xo + 2], x[xo + 3]};
    double e[4], aa[3];
    quat_mul(cj, q, e);
    quat_to_angle_axis(e, aa);
    delta[ct] = aa[0]; delta[ct + 1] = aa[1]; delta[ct + 2] = aa[2];
    D[i] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];   // MinusJacobian(x) PlusJacobian(x) = |x|^2 I
  } else {
    for (int k = 0; k < sz; ++k) delta[ct + k] = x[xo + k] - m.xbar[ca + k];
    D[i] = 1.0;
  }
}
template <bool WITH_J, bool IN_LDS, int NT>
__device__ __forceinline__ void marg_eval_kernel_body(const int first_row, const MargDev& m, const double* __restrict__ x, double* __restrict__ cost_part) {
  __shared__ double s_delta[IN_LDS ? kMargColsLds : 1];
  __shared__ double s_D[IN_LDS ? kMargColsLds : 1];
  const int row = first_row + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double* delta = m.delta;
  const double* D = m.D;
  if (IN_LDS) {
    for (int i = threadIdx.x; i < m.nblk; i += NT) marg_delta_block(m, i, x, s_delta, s_D);
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    delta = s_delta; D = s_D;
  }
  if (row >= m.rows) return;
  const double* Ar = m.A + (size_t)row * m.cols;
  double acc = 0.0;
  for (int k = lane; k < m.cols; k += 64) acc = fma(Ar[k], delta[k], acc);
  acc = wave_sum(acc);
  const double r = m.b[row] + acc;
  if (lane == 0) {
    cost_part[row] = 0.5 * r * r;
    if (WITH_J) m.r[row] = r;
  }
  if (!WITH_J) return;
  double* Jr = m.J + (size_t)row * m.cols;
  for (int k = lane; k < m.cols; k += 64) {
    const int bi = m.col_blk[k];
    double v;
    if (m.col_t[k] < 0) v = 0.0;
    else if (m.blk_quat[bi]) v = Ar[k] * D[bi];
    else v = Ar[k];
    Jr[k] = v;
  }
}
// gradient J^T r (also into the rhs row) and diag(J^T J): the workgroups of row blockIdx.y == gridDim.y - 1 of the assembly launch,
// sixteen columns each, the rows split over the sixteen thread rows
BSG_DEV void marg_grad_block(const MargDev& m, int a0, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                             double* __restrict__ hdiag, const int* __restrict__ perm, double (*sG)[17], double (*sH)[17]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int a = a0 + tx;
  double gs = 0.0, hs = 0.0;
  if (a < m.cols)
    for (int k = ty; k < m.rows; k += 16) { const double j = m.J[(size_t)k * m.cols + a]; gs = fma(j, m.r[k], gs); hs = fma(j, j, hs); }
  sG[ty][tx] = gs; sH[ty][tx] = hs;
  __syncthreads();
  if (ty != 0 || a >= m.cols) return;
  const int ta = m.col_t[a];
  if (ta < 0) return;
  double g = 0.0, h = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { g += sG[q][tx]; h += sH[q][tx]; }
  atomicAdd(&S[(size_t)rhs_row * ld + perm[ta]], g);
  atomicAdd(&grad[ta], g);
  atomicAdd(&hdiag[ta], h);
}

// S += J^T J (16 x 16 output tile per workgroup, rows staged through LDS), FP64 atomics because the blocks of a
// marginal factor are scattered over the reduced system
__device__ __forceinline__ void marg_assemble_kernel_body(const int bsg_bx, const int bsg_by, const int bsg_gy, const MargDev& m, double* __restrict__ S, int ld,
                                                          const int* __restrict__ perm, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag) {
  __shared__ double sA[16][17], sB[16][17];
  if (bsg_by == bsg_gy - 1) { marg_grad_block(m, bsg_bx * 16, S, ld, rhs_row, grad, hdiag, perm, sA, sB); return; }
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int a0 = bsg_by * 16, b0 = bsg_bx * 16;
  double acc = 0.0;
  for (int k0 = 0; k0 < m.rows; k0 += 16) {
    const int k = k0 + ty;
    sA[ty][tx] = (k < m.rows && a0 + tx < m.cols) ? m.J[(size_t)k * m.cols + a0 + tx] : 0.0;
    sB[ty][tx] = (k < m.rows && b0 + tx < m.cols) ? m.J[(size_t)k * m.cols + b0 + tx] : 0.0;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = fma(sA[kk][ty], sB[kk][tx], acc);
    __syncthreads();
  }
  const int a = a0 + ty, b = b0 + tx;
  if (a >= m.cols || b >= m.cols) return;
  const int ta = m.col_t[a], tb = m.col_t[b];
  if (ta < 0 || tb < 0) return;
  atomicAdd(&S[(size_t)perm[ta] * ld + perm[tb]], acc);
}
// one wave per row (wave w of the workgroup: row first_row + w): model-cost-change term -(J d)(r + J d / 2)
__device__ __forceinline__ void marg_mcc_kernel_body(const int first_row, const MargDev& m, const double* __restrict__ delta_tan, double* __restrict__ part) {
  const int row = first_row + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m.rows) return;
  const double* Jr = m.J + (size_t)row * m.cols;
  double jv = 0.0;
  for (int k = lane; k < m.cols; k += 64) { const int t = m.col_t[k]; if (t >= 0) jv = fma(Jr[k], delta_tan[t], jv); }
  jv = wave_sum(jv);
  if (lane == 0) part[row] = -jv * (m.r[row] + 0.5 * jv);
}
// whether a prior fits the single-launch evaluation (delta and D in LDS)
inline bool marg_fits_lds(const MargDev& m) { return m.cols <= kMargColsLds && m.nblk <= kMargColsLds; }

}  // namespace bsg
