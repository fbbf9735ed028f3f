// Dense linear prior factors — [EXT] fuse_constraints::MarginalConstraint, the product of
// fuse_constraints::marginalizeVariables (bs_optimizers/src/fixed_lag_smoother.cpp:270-271):
//   r = b + sum_i A_i (x_i [-] xbar_i)      [-] = LocalParameterization::Minus(xbar_i, x_i)
//     quaternion blocks: QuaternionToAngleAxis(xbar^-1 (x) x)   (bs_constraints/src/jacobians.cpp:37-50)
//   tangent Jacobian, the way fuse's MarginalCostFunction builds it: A_i MinusJacobian(x_i) PlusJacobian(x_i)
//     (ComputeMinusJacobian "evaluated at x1 = x2 = x", at the current parameter) = A_i |x_i|^2
//     (jacobians.cpp:144-174 multiplied out; exactly A_i for unit quaternions)
// A marginal factor is a few hundred rows / columns at most and there is usually exactly one in a window, so
// the kernels are plain: one launch per factor and step, rows x cols work spread over the chip.
#include "bsgpu_device.h"

namespace bsg {

// thread per block: delta segment and, for quaternion blocks, the scale |x|^2 of the tangent Jacobian
__global__ void marg_delta_kernel(MargDev m, const double* __restrict__ x, int with_J) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= m.nblk) return;
  const int xo = m.blk_xoff[i], sz = m.blk_size[i], ct = m.blk_col[i], ca = m.blk_amb[i];
  if (m.blk_quat[i]) {
    const double* xb = m.xbar + ca;
    const double cj[4] = {xb[0], -xb[1], -xb[2], -xb[3]};   // QuaternionInverse = conjugate (jacobians.cpp:3-8)
    const double q[4] = {x[xo], x[xo + 1], x[xo + 2], x[xo + 3]};
    double e[4], aa[3];
    quat_mul(cj, q, e);
    quat_to_angle_axis(e, aa);
    m.delta[ct] = aa[0]; m.delta[ct + 1] = aa[1]; m.delta[ct + 2] = aa[2];
    if (with_J) m.D[i] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];   // MinusJacobian(x) PlusJacobian(x) = |x|^2 I
  } else {
    for (int k = 0; k < sz; ++k) m.delta[ct + k] = x[xo + k] - m.xbar[ca + k];
  }
}

// one wave per residual row: r = b + A_row . delta, cost term, and J_row = A_row D (constant columns zero)
template <bool WITH_J>
__global__ __launch_bounds__(64) void marg_eval_kernel(MargDev m, double* __restrict__ cost_part) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const double* Ar = m.A + (size_t)row * m.cols;
  double acc = 0.0;
  for (int k = lane; k < m.cols; k += 64) acc = fma(Ar[k], m.delta[k], acc);
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
    else if (m.blk_quat[bi]) v = Ar[k] * m.D[bi];
    else v = Ar[k];
    Jr[k] = v;
  }
}

// S += J^T J (16 x 16 output tile per workgroup, rows staged through LDS), FP64 atomics because the blocks of a
// marginal factor are scattered over the reduced system
__global__ __launch_bounds__(256) void marg_assemble_kernel(MargDev m, double* __restrict__ S, int ld, const int* __restrict__ perm) {
  __shared__ double sA[16][17], sB[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
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
  atomicAdd(&S[(size_t)(perm[ta >> 6] * 64 + (ta & 63)) * ld + perm[tb >> 6] * 64 + (tb & 63)], acc);
}

// thread per column: gradient J^T r (also into the rhs row) and diag(J^T J)
__global__ void marg_grad_kernel(MargDev m, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                 double* __restrict__ hdiag, const int* __restrict__ perm) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= m.cols) return;
  const int ta = m.col_t[a];
  if (ta < 0) return;
  double gs = 0.0, hs = 0.0;
  for (int k = 0; k < m.rows; ++k) { const double j = m.J[(size_t)k * m.cols + a]; gs = fma(j, m.r[k], gs); hs = fma(j, j, hs); }
  atomicAdd(&S[(size_t)rhs_row * ld + perm[ta >> 6] * 64 + (ta & 63)], gs);
  atomicAdd(&grad[ta], gs);
  atomicAdd(&hdiag[ta], hs);
}

// one wave per row: model-cost-change term -(J d)(r + J d / 2)
__global__ __launch_bounds__(64) void marg_mcc_kernel(MargDev m, const double* __restrict__ delta_tan, double* __restrict__ part) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const double* Jr = m.J + (size_t)row * m.cols;
  double jv = 0.0;
  for (int k = lane; k < m.cols; k += 64) { const int t = m.col_t[k]; if (t >= 0) jv = fma(Jr[k], delta_tan[t], jv); }
  jv = wave_sum(jv);
  if (lane == 0) part[row] = -jv * (m.r[row] + 0.5 * jv);
}

void launch_marg_eval(hipStream_t s, const MargDev& m, const double* x, bool with_J, double* cost_part) {
  hipLaunchKernelGGL(marg_delta_kernel, dim3((m.nblk + 63) / 64), dim3(64), 0, s, m, x, with_J ? 1 : 0);
  if (with_J) hipLaunchKernelGGL(marg_eval_kernel<true>, dim3(m.rows), dim3(64), 0, s, m, cost_part);
  else hipLaunchKernelGGL(marg_eval_kernel<false>, dim3(m.rows), dim3(64), 0, s, m, cost_part);
}
void launch_marg_assemble(hipStream_t s, const MargDev& m, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm) {
  const int g = (m.cols + 15) / 16;
  hipLaunchKernelGGL(marg_assemble_kernel, dim3(g, g), dim3(256), 0, s, m, S, ld, perm);
  hipLaunchKernelGGL(marg_grad_kernel, dim3((m.cols + 63) / 64), dim3(64), 0, s, m, S, ld, rhs_row, grad, hdiag, perm);
}
void launch_marg_mcc(hipStream_t s, const MargDev& m, const double* delta_tan, double* part) {
  hipLaunchKernelGGL(marg_mcc_kernel, dim3(m.rows), dim3(64), 0, s, m, delta_tan, part);
}

}  // namespace bsg
