// Glue between the assembled reduced camera system and its tiled Cholesky (k_chol.hip):
//   pose_diag    LM diagonal (Ceres' Jacobi scaling folded in) onto the diagonal of S, unit pivots on the
//                padding / rhs positions.  S is in SOLVER order: position = DensePlan::dpos[tangent index]
//                (dense_plan.h); hdiag / scale / dcl stay in tangent order.
//   marg_*       gather of the reduced system into [marginalised | kept] order and its positive-SEMI-definite
//                Cholesky: Schur complement onto the kept variables and the factor of the marginal prior
//                (bsgpu_marginalize)
//   cov_*        marginal covariance blocks from forward-substituted unit vectors (bsgpu_covariance)
// Stands in, together with k_chol.hip, for the reference's SPARSE_NORMAL_CHOLESKY step ([EXT] Ceres,
// beam_slam_launch/config/vio.yaml:9) after Schur elimination of the landmarks.
#include "bsgpu_device.h"

namespace bsg {

__global__ void pose_diag_kernel(int n_pose, double* __restrict__ S, int ld, const double* __restrict__ hdiag,
                                 const double* __restrict__ radius_ptr, int compute_scale, int compute_dcl, int jacobi,
                                 double lm_lo, double lm_hi, double* __restrict__ scale, double* __restrict__ dcl, int npad,
                                 const int* __restrict__ iperm, double radius_val) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // solver position
  if (i >= npad) return;
  pose_diag_element(i, n_pose, S, ld, hdiag, 1.0 / (radius_ptr ? radius_ptr[0] : radius_val), compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, iperm);
}

void launch_pose_diag(hipStream_t s, int n_pose, double* S, int ld, const double* hdiag, const double* radius_ptr,
                      int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale,
                      double* dcl, int npad, const int* iperm, double radius_val) {
  hipLaunchKernelGGL(pose_diag_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, n_pose, S, ld, hdiag, radius_ptr,
                     compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, npad, iperm, radius_val);
}

// ---- true marginalisation (fuse_constraints::marginalizeVariables, fixed_lag_smoother.cpp:270-271) --------------
// M (n x n, row-major) = S[order, order] in natural tangent numbering, g (n) = rhs[order]
__global__ void marg_gather_kernel(const double* __restrict__ S, int ld, int rhs_row, const int* __restrict__ spos, int n,
                                   double* __restrict__ M, double* __restrict__ g, double* __restrict__ diag0) {
  const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const double v = S[(size_t)spos[i] * ld + spos[j]];
  M[(size_t)i * n + j] = v;
  if (i == j) diag0[i] = v;
  if (i == 0) g[j] = S[(size_t)rhs_row * ld + spos[j]];
}
// One workgroup, right-looking Cholesky of M with the rhs carried along.  The first m columns (the marginalised
// pose-side variables) must be positive definite; for the kept columns a pivot at round-off level means the prior
// has no information in that direction (gauge freedom of the marginalised factors): the column is dropped, which
// is what the rank-revealing QR of fuse's marginalizeVariables leaves as a zero row.
// status[0] = number of kept pivots, status[1] > 0 on a non-positive pivot inside the first m columns.
__global__ __launch_bounds__(1024) void marg_psd_chol_kernel(double* __restrict__ M, double* __restrict__ g, const double* __restrict__ diag0,
                                                             int n, int m, double rel_tol, int* __restrict__ pivot_ok, double* __restrict__ status) {
  extern __shared__ double s_col[];   // column j of L (n doubles)
  __shared__ double s_l, s_gj;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  int kept = 0;
  bool bad = false;
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      const double d = M[(size_t)j * n + j];
      const bool ok = d > rel_tol * fmax(diag0[j], 1e-300) && isfinite(d);
      s_ok = ok ? 1 : 0;
      s_l = ok ? sqrt(d) : 0.0;
      s_gj = ok ? g[j] / s_l : 0.0;
      g[j] = s_gj;
      pivot_ok[j] = s_ok;
      M[(size_t)j * n + j] = s_l;
    }
    __syncthreads();
    const int ok = s_ok;
    const double l = s_l, gj = s_gj;
    if (!ok && j < m) bad = true;
    if (ok && j >= m) ++kept;
    // scale column j below the diagonal (stored in row j of the upper part as well: L(i,j) kept at M[i][j])
    const double inv = ok ? 1.0 / l : 0.0;
    for (int i = j + 1 + tid; i < n; i += 1024) {
      const double v = M[(size_t)i * n + j] * inv;
      M[(size_t)i * n + j] = v;
      s_col[i] = v;
      g[i] = fma(-v, gj, g[i]);
    }
    __syncthreads();
    if (ok) {
      // trailing update of the lower triangle: M[i][k] -= L(i,j) L(k,j), j < k <= i
      const int rem = n - j - 1;
      for (int e = tid; e < rem * rem; e += 1024) {
        const int i = j + 1 + e / rem, k = j + 1 + e % rem;
        if (k <= i) M[(size_t)i * n + k] = fma(-s_col[i], s_col[k], M[(size_t)i * n + k]);
      }
    }
    __syncthreads();
  }
  if (tid == 0) { status[0] = (double)kept; status[1] = bad ? 1.0 : 0.0; }
}
// A = rows of L_kk^T with a non-zero pivot, b = the matching entries of the forward-substituted rhs
__global__ void marg_extract_kernel(const double* __restrict__ M, const double* __restrict__ g, const int* __restrict__ pivot_ok,
                                    int n, int m, double* __restrict__ A, double* __restrict__ b) {
  const int k = n - m;
  const int rr = blockIdx.x;                 // candidate row = kept column m + rr
  if (!pivot_ok[m + rr]) return;
  int out = 0;
  for (int q = 0; q < rr; ++q) out += pivot_ok[m + q];
  for (int cidx = threadIdx.x; cidx < k; cidx += 64) A[(size_t)out * k + cidx] = (cidx >= rr) ? M[(size_t)(m + cidx) * n + m + rr] : 0.0;
  if (threadIdx.x == 0) b[out] = g[m + rr];
}
void launch_marg_schur(hipStream_t s, const double* S, int ld, int rhs_row, const int* spos_dev, int n, int m, double rel_tol,
                       double* M, double* g, double* diag0, int* pivot_ok, double* status, double* A, double* b) {
  hipLaunchKernelGGL(marg_gather_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, S, ld, rhs_row, spos_dev, n, M, g, diag0);
  hipLaunchKernelGGL(marg_psd_chol_kernel, dim3(1), dim3(1024), sizeof(double) * n, s, M, g, diag0, n, m, rel_tol, pivot_ok, status);
  if (n > m) hipLaunchKernelGGL(marg_extract_kernel, dim3(n - m), dim3(64), 0, s, M, g, pivot_ok, n, m, A, b);
}

// ---- marginal covariance of pose-side blocks: Sigma = S^-1 (undamped).  Unit vectors ride through the
// factorisation as rows of the rhs tile (z_k = L^-1 e_k lands in the shadow matrix); Sigma(i,j) = z_i . z_j.
__global__ void cov_units_kernel(double* __restrict__ S, int ld, int rhs_row, const int* __restrict__ cols, int n) {
  const int k = threadIdx.x;
  if (k < n) S[(size_t)(rhs_row + k) * ld + cols[k]] = 1.0;
}
__global__ __launch_bounds__(256) void cov_dots_kernel(const double* __restrict__ Lp, int ld, int rhs_row, int n_cols, int row_b0,
                                                       int tb, double* __restrict__ out) {
  __shared__ double sred[4];
  const int i = blockIdx.x / tb, j = blockIdx.x % tb;
  const double* za = Lp + (size_t)(rhs_row + i) * ld;
  const double* zb = Lp + (size_t)(rhs_row + row_b0 + j) * ld;
  double acc = 0.0;
  for (int k = threadIdx.x; k < n_cols; k += 256) acc = fma(za[k], zb[k], acc);
  const double t = block_sum_256(acc, sred);
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}
void launch_cov_units(hipStream_t s, double* S, int ld, int rhs_row, const int* cols_dev, int n) {
  (void)hipMemsetAsync(S + (size_t)rhs_row * ld, 0, sizeof(double) * 64 * (size_t)ld, s);
  hipLaunchKernelGGL(cov_units_kernel, dim3(1), dim3(64), 0, s, S, ld, rhs_row, cols_dev, n);
}
void launch_cov_dots(hipStream_t s, const double* Lp, int ld, int rhs_row, int n_cols, int ta, int row_b0, int tb, double* out) {
  hipLaunchKernelGGL(cov_dots_kernel, dim3(ta * tb), dim3(256), 0, s, Lp, ld, rhs_row, n_cols, row_b0, tb, out);
}

}  // namespace bsg
