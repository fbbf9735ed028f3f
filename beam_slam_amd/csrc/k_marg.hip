// Dense linear prior factors — [EXT] fuse_constraints::MarginalConstraint, the product of
// fuse_constraints::marginalizeVariables (bs_optimizers/src/fixed_lag_smoother.cpp:270-271):
//   r = b + sum_i A_i (x_i [-] xbar_i)      [-] = LocalParameterization::Minus(xbar_i, x_i)
//     quaternion blocks: QuaternionToAngleAxis(xbar^-1 (x) x)   (bs_constraints/src/jacobians.cpp:37-50)
//   tangent Jacobian, the way fuse's MarginalCostFunction builds it: A_i MinusJacobian(x_i) PlusJacobian(x_i)
//     (ComputeMinusJacobian "evaluated at x1 = x2 = x", at the current parameter) = A_i |x_i|^2
//     (jacobians.cpp:144-174 multiplied out; exactly A_i for unit quaternions)
// A marginal factor is a few hundred rows / columns at most and there is usually exactly one in a window, so
// the kernels are plain, rows x cols work spread over the chip: per factor and step one launch for the evaluation (delta formed by
// every row's wave in LDS), one for the assembly (the gradient in its last row of workgroups), one for the model cost change.
#include "bsgpu_device.h"

namespace bsg {

// one wave per residual row.  Every wave first forms delta = x [-] xbar and, for quaternion blocks, the scale |x|^2 of the tangent
// Jacobian itself, in LDS (a few dozen blocks: cheaper than a launch of its own in front of every evaluation), then
// r = b + A_row . delta, the cost term, and J_row = A_row D (constant columns zero)
constexpr int kMargColsLds = 1024;   // columns / blocks kept in LDS per wave; larger factors read the arrays a separate launch left
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
// (factors too wide for the LDS copy: delta and D through global memory, a launch in front)
__global__ void marg_delta_kernel(MargDev m, const double* __restrict__ x) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < m.nblk) marg_delta_block(m, i, x, m.delta, m.D);
}
template <bool WITH_J, bool IN_LDS>
__device__ __forceinline__ void marg_eval_kernel_body(const int bsg_bx, const MargDev& m, const double* __restrict__ x, double* __restrict__ cost_part) {
  __shared__ double s_delta[IN_LDS ? kMargColsLds : 1];
  __shared__ double s_D[IN_LDS ? kMargColsLds : 1];
  const int row = bsg_bx, lane = threadIdx.x;
  const double* delta = m.delta;
  const double* D = m.D;
  if (IN_LDS) {
    for (int i = lane; i < m.nblk; i += 64) marg_delta_block(m, i, x, s_delta, s_D);
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    delta = s_delta; D = s_D;
  }
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
template <bool WITH_J, bool IN_LDS>
__global__ __launch_bounds__(64) void marg_eval_kernel(MargDev m, const double* __restrict__ x, double* __restrict__ cost_part) {
  marg_eval_kernel_body<WITH_J, IN_LDS>((int)blockIdx.x, m, x, cost_part);
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
__global__ __launch_bounds__(256) void marg_assemble_kernel(MargDev m, double* __restrict__ S, int ld, const int* __restrict__ perm, int rhs_row,
                                                            double* __restrict__ grad, double* __restrict__ hdiag) {
  marg_assemble_kernel_body((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, m, S, ld, perm, rhs_row, grad, hdiag);
}

// one wave per row: model-cost-change term -(J d)(r + J d / 2)
__device__ __forceinline__ void marg_mcc_kernel_body(const int bsg_bx, const MargDev& m, const double* __restrict__ delta_tan, double* __restrict__ part) {
  const int row = bsg_bx, lane = threadIdx.x;
  const double* Jr = m.J + (size_t)row * m.cols;
  double jv = 0.0;
  for (int k = lane; k < m.cols; k += 64) { const int t = m.col_t[k]; if (t >= 0) jv = fma(Jr[k], delta_tan[t], jv); }
  jv = wave_sum(jv);
  if (lane == 0) part[row] = -jv * (m.r[row] + 0.5 * jv);
}
__global__ __launch_bounds__(64) void marg_mcc_kernel(MargDev m, const double* __restrict__ delta_tan, double* __restrict__ part) {
  marg_mcc_kernel_body((int)blockIdx.x, m, delta_tan, part);
}

void launch_marg_eval(hipStream_t s, const MargDev& m, const double* x, bool with_J, double* cost_part) {
  if (m.cols <= kMargColsLds && m.nblk <= kMargColsLds) {
    if (with_J) hipLaunchKernelGGL((marg_eval_kernel<true, true>), dim3(m.rows), dim3(64), 0, s, m, x, cost_part);
    else hipLaunchKernelGGL((marg_eval_kernel<false, true>), dim3(m.rows), dim3(64), 0, s, m, x, cost_part);
    return;
  }
  hipLaunchKernelGGL(marg_delta_kernel, dim3((m.nblk + 63) / 64), dim3(64), 0, s, m, x);
  if (with_J) hipLaunchKernelGGL((marg_eval_kernel<true, false>), dim3(m.rows), dim3(64), 0, s, m, x, cost_part);
  else hipLaunchKernelGGL((marg_eval_kernel<false, false>), dim3(m.rows), dim3(64), 0, s, m, x, cost_part);
}
void launch_marg_assemble(hipStream_t s, const MargDev& m, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm) {
  const int g = (m.cols + 15) / 16;
  hipLaunchKernelGGL(marg_assemble_kernel, dim3(g, g + 1), dim3(256), 0, s, m, S, ld, perm, rhs_row, grad, hdiag);   // (row g: gradient and diagonal)
}
void launch_marg_mcc(hipStream_t s, const MargDev& m, const double* delta_tan, double* part) {
  hipLaunchKernelGGL(marg_mcc_kernel, dim3(m.rows), dim3(64), 0, s, m, delta_tan, part);
}

// ---- the same launches over several windows (bsgpu_batch.cpp): a window after a slide with true marginalisation (fixed_lag_smoother.cpp:269-272)
// carries ONE dense prior; entry w = what its lone launches pass (a zero grid: the window has none).  The assembly's 2-d grid is flattened.
struct marg_eval_Args { int bsg_grid; MargDev m; const double* x; double* cost_part; };
template <bool WITH_J>
__global__ __launch_bounds__(64) void marg_eval_kernel_batch(const marg_eval_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const marg_eval_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  marg_eval_kernel_body<WITH_J, true>((int)blockIdx.x, a.m, a.x, a.cost_part);
}
bool batchargs_marg_eval(BatchArgTable& t, const MargDev* m, const double* x, double* cost_part) {
  marg_eval_Args a;
  a.m = m ? *m : MargDev(); a.x = x; a.cost_part = cost_part; a.bsg_grid = m ? m->rows : 0;
  if (m && !(m->cols <= kMargColsLds && m->nblk <= kMargColsLds)) return false;   // (wider priors: the two-launch form, not batched)
  t.push(a);
  return true;
}
void launch_marg_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J) {
  if (n <= 0 || t.max_grid <= 0) return;
  const auto* A = static_cast<const marg_eval_Args*>(t.dev);
  if (with_J) hipLaunchKernelGGL(marg_eval_kernel_batch<true>, dim3(t.max_grid, n), dim3(64), 0, s, A, dyn, list);
  else hipLaunchKernelGGL(marg_eval_kernel_batch<false>, dim3(t.max_grid, n), dim3(64), 0, s, A, dyn, list);
}
struct marg_assemble_Args { int bsg_grid; int g; MargDev m; double* S; int ld; const int* perm; int rhs_row; double* grad; double* hdiag; };
__global__ __launch_bounds__(256) void marg_assemble_kernel_batch(const marg_assemble_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const marg_assemble_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  marg_assemble_kernel_body((int)blockIdx.x % a.g, (int)blockIdx.x / a.g, a.g + 1, a.m, a.S, a.ld, a.perm, a.rhs_row, a.grad, a.hdiag);
}
void batchargs_marg_assemble(BatchArgTable& t, const MargDev* m, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm) {
  marg_assemble_Args a;
  a.m = m ? *m : MargDev(); a.S = S; a.ld = ld; a.perm = perm; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag;
  a.g = m ? (m->cols + 15) / 16 : 1;
  a.bsg_grid = m ? a.g * (a.g + 1) : 0;
  t.push(a);
}
void launch_marg_assemble_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(marg_assemble_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const marg_assemble_Args*>(t.dev), dyn, list);
}
struct marg_mcc_Args { int bsg_grid; MargDev m; const double* delta_tan; double* part; };
__global__ __launch_bounds__(64) void marg_mcc_kernel_batch(const marg_mcc_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const marg_mcc_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  marg_mcc_kernel_body((int)blockIdx.x, a.m, a.delta_tan, a.part);
}
void batchargs_marg_mcc(BatchArgTable& t, const MargDev* m, const double* delta_tan, double* part) {
  marg_mcc_Args a;
  a.m = m ? *m : MargDev(); a.delta_tan = delta_tan; a.part = part; a.bsg_grid = m ? m->rows : 0;
  t.push(a);
}
void launch_marg_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(marg_mcc_kernel_batch, dim3(t.max_grid, n), dim3(64), 0, s, static_cast<const marg_mcc_Args*>(t.dev), dyn, list);
}

}  // namespace bsg
