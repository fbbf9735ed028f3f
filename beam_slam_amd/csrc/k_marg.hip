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
#include "marg_body.h"

namespace bsg {

// (factors too wide for the LDS copy: delta and D through global memory, a launch in front)
__global__ void marg_delta_kernel(MargDev m, const double* __restrict__ x) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < m.nblk) marg_delta_block(m, i, x, m.delta, m.D);
}
template <bool WITH_J, bool IN_LDS>
__global__ __launch_bounds__(64) void marg_eval_kernel(MargDev m, const double* __restrict__ x, double* __restrict__ cost_part) {
  marg_eval_kernel_body<WITH_J, IN_LDS, 64>((int)blockIdx.x, m, x, cost_part);
}
__global__ __launch_bounds__(256) void marg_assemble_kernel(MargDev m, double* __restrict__ S, int ld, const int* __restrict__ perm, int rhs_row,
                                                            double* __restrict__ grad, double* __restrict__ hdiag) {
  marg_assemble_kernel_body((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, m, S, ld, perm, rhs_row, grad, hdiag);
}
__global__ __launch_bounds__(64) void marg_mcc_kernel(MargDev m, const double* __restrict__ delta_tan, double* __restrict__ part) {
  marg_mcc_kernel_body((int)blockIdx.x, m, delta_tan, part);
}

void launch_marg_eval(hipStream_t s, const MargDev& m, const double* x, bool with_J, double* cost_part) {
  if (marg_fits_lds(m)) {
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
  marg_eval_kernel_body<WITH_J, true, 64>((int)blockIdx.x, a.m, a.x, a.cost_part);
}
bool batchargs_marg_eval(BatchArgTable& t, const MargDev* m, const double* x, double* cost_part) {
  marg_eval_Args a;
  a.m = m ? *m : MargDev(); a.x = x; a.cost_part = cost_part; a.bsg_grid = m ? m->rows : 0;
  if (m && !marg_fits_lds(*m)) return false;   // (wider priors: the two-launch form, not batched)
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
