// Dense FP64 solve of the reduced camera system on gfx950 — the one MFMA user of the path.
//   pose_diag      LM diagonal (with Ceres' Jacobi scaling folded in) onto S, identity on the padding
//   chol_panel     64-wide panel: diagonal block factorisation in LDS + triangular solve of the rows below
//   chol_update    trailing update A22 -= L21 L21^T with v_mfma_f64_16x16x4_f64, 64x64 tiles from LDS
//   backsolve      L^T y = y' (y' = the rhs row carried through the factorisation as row n_pose)
// S is row-major, full storage, leading dimension ld = npad (multiple of 64); only the lower
// triangle is referenced after assembly.  Stands in for the reference's SPARSE_NORMAL_CHOLESKY
// step ([EXT] Ceres, beam_slam_launch/config/vio.yaml:9) after Schur elimination of the landmarks.
#include "bsgpu_device.h"

namespace bsg {

constexpr int NB = 64;

__global__ void pose_diag_kernel(int n_pose, double* __restrict__ S, int ld, const double* __restrict__ hdiag,
                                 const double* __restrict__ radius_ptr, int compute_scale, int compute_dcl, int jacobi, double lm_lo,
                                 double lm_hi, double* __restrict__ scale, double* __restrict__ dcl, int npad) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= npad) return;
  const double inv_radius = 1.0 / radius_ptr[0];
  if (j < n_pose) {
    const double h = hdiag[j];
    double sc = compute_scale ? (jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0) : scale[j];
    double d = compute_dcl ? fmin(fmax(sc * sc * h, lm_lo), lm_hi) / (sc * sc) : dcl[j];
    if (compute_scale) scale[j] = sc;
    if (compute_dcl) dcl[j] = d;
    S[(size_t)j * ld + j] += d * inv_radius;
  } else {
    S[(size_t)j * ld + j] = 1.0;  // rhs row / padding: unit pivot, never used as a real pivot
  }
}

void launch_pose_diag(hipStream_t s, int n_pose, double* S, int ld, const double* hdiag, const double* radius_ptr,
                      int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale,
                      double* dcl, int npad) {
  hipLaunchKernelGGL(pose_diag_kernel, dim3((npad + 255) / 256), dim3(256), 0, s, n_pose, S, ld, hdiag, radius_ptr,
                     compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, npad);
}

// ---------------------------------------------------------------------------------------------------
// panel k: every workgroup factors the 64x64 diagonal block in LDS (redundantly — it is 90 kflop),
// workgroup 0 writes it back, workgroup t >= 1 solves X L11^T = A(tile row t0 + t, panel k).
// Columns >= n_pose (rhs row, padding) are treated as unit pivots with no coupling.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ S, int ld, int k, int n_pose,
                                                         const int* __restrict__ row_tiles, double* __restrict__ scal) {
  __shared__ double sA[NB][NB + 1];
  __shared__ double sX[NB][NB + 1];
  const int tid = threadIdx.x;
  const int c0 = k * NB;
  for (int i = tid; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    sA[r][c] = (c <= r) ? S[(size_t)(c0 + r) * ld + c0 + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < NB; ++j) {
    const bool real = (c0 + j) < n_pose;
    if (real) {
      const double d = sA[j][j];
      if (tid == 0 && blockIdx.x == 0 && !(d > 0.0 && isfinite(d))) scal[SC_CHOL_FAIL] = 1.0;
      const double dj = sqrt(d);
      __syncthreads();
      if (tid == 0) sA[j][j] = dj;
      for (int i = j + 1 + tid; i < NB; i += 256) sA[i][j] /= dj;
      __syncthreads();
      // trailing update of the block
      const int rem = NB - 1 - j;
      for (int p = tid; p < rem * rem; p += 256) {
        const int i = j + 1 + p / rem, c = j + 1 + p % rem;
        if (c <= i) sA[i][c] -= sA[i][j] * sA[c][j];
      }
      __syncthreads();
    } else {
      __syncthreads();
      if (tid == 0) sA[j][j] = 1.0;
      for (int i = j + 1 + tid; i < NB; i += 256) sA[i][j] = 0.0;
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    for (int i = tid; i < NB * NB; i += 256) {
      const int r = i / NB, c = i % NB;
      if (c <= r) S[(size_t)(c0 + r) * ld + c0 + c] = sA[r][c];
    }
    return;
  }
  const int r0 = row_tiles[blockIdx.x - 1] * NB;
  for (int i = tid; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    sX[r][c] = S[(size_t)(r0 + r) * ld + c0 + c];
  }
  __syncthreads();
  // X L11^T = A: row r is a dependent chain over the columns; 4 lanes share a row and split the
  // inner product (p = part, part+4, ...), reduced with two xor-shuffles.
  {
    const int r = tid >> 2, part = tid & 3;
    for (int j = 0; j < NB; ++j) {
      double s = 0.0;
      for (int p = part; p < j; p += 4) s += sX[r][p] * sA[j][p];
      s += __shfl_xor(s, 1, 4);
      s += __shfl_xor(s, 2, 4);
      if (part == 0) sX[r][j] = (sX[r][j] - s) / sA[j][j];
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int i = tid; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    S[(size_t)(r0 + r) * ld + c0 + c] = sX[r][c];
  }
}

// ---------------------------------------------------------------------------------------------------
// trailing update with FP64 MFMA.  Workgroup (ti, tj), tj <= ti in the list of active row tiles:
//   C(ti, tj) -= L(ti, k) L(tj, k)^T,   64x64x64.   4 waves; wave w owns rows 16w..16w+15, 4 MFMA tiles.
// v_mfma_f64_16x16x4_f64: a = A[lane&15][lane>>4], b = B[lane>>4][lane&15],
//                          d[reg] = D[(lane>>4) + 4 reg][lane&15]
// ---------------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ S, int ld, int k,
                                                          const int* __restrict__ row_tiles, int n_tiles) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  constexpr int LDT = NB + 2;  // 66 doubles: conflict-free ds_read_b64 for the fragment pattern
  __shared__ double sLi[NB * LDT];
  __shared__ double sLj[NB * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ri = row_tiles[bi] * NB, rj = row_tiles[bj] * NB, c0 = k * NB;
  for (int i = tid; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    sLi[r * LDT + c] = S[(size_t)(ri + r) * ld + c0 + c];
    sLj[r * LDT + c] = S[(size_t)(rj + r) * ld + c0 + c];
  }
  __syncthreads();
  double4_t acc[4];
  const int crow = (lane >> 4), ccol = lane & 15;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      acc[t][reg] = S[(size_t)(ri + 16 * w + crow + 4 * reg) * ld + rj + 16 * t + ccol];
  const int arow = 16 * w + (lane & 15), kk0 = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double a = -sLi[arow * LDT + 4 * kk + kk0];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = sLj[(16 * t + (lane & 15)) * LDT + 4 * kk + kk0];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      S[(size_t)(ri + 16 * w + crow + 4 * reg) * ld + rj + 16 * t + ccol] = acc[t][reg];
  (void)n_tiles;
}

void launch_chol_panel(hipStream_t s, double* S, int ld, int k, int n_pose, const int* row_tiles_dev, int n_rows,
                       double* scal) {
  hipLaunchKernelGGL(chol_panel_kernel, dim3(1 + n_rows), dim3(256), 0, s, S, ld, k, n_pose, row_tiles_dev, scal);
}
void launch_chol_update(hipStream_t s, double* S, int ld, int k, const int* row_tiles_dev, int n_rows) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(chol_update_kernel, dim3(n_rows, n_rows), dim3(256), 0, s, S, ld, k, row_tiles_dev, n_rows);
}

// ---------------------------------------------------------------------------------------------------
// backward substitution L^T y = y'.  y is initialised with y' (row n_pose of S).  Step kb (from the
// last real tile down): every workgroup solves the 64x64 diagonal system in LDS (one wave, the
// dependent chain), workgroup 0 writes y_kb, and each workgroup applies y'_j -= sum_r L[r][j] y_kb[r]
// to its 256 columns j < kb*64 (coalesced row reads).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void backsolve_step_kernel(const double* __restrict__ S, int ld, int kb, int n_pose,
                                                             double* __restrict__ y, int col_begin) {
  __shared__ double sL[NB][NB + 1];
  __shared__ double sy[NB];
  const int tid = threadIdx.x;
  const int c0 = kb * NB;
  for (int i = tid; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    sL[r][c] = (c <= r) ? S[(size_t)(c0 + r) * ld + c0 + c] : 0.0;
  }
  if (tid < NB) sy[tid] = (c0 + tid < n_pose) ? y[c0 + tid] : 0.0;
  __syncthreads();
  if (tid < NB) {
    // column-oriented back substitution in registers of one wave: y_j /= L_jj; y_i -= L_ji y_j (i < j)
    double yv = sy[tid];
    for (int j = NB - 1; j >= 0; --j) {
      const double yj = __shfl(yv, j, 64) / sL[j][j];
      if (tid == j) yv = yj;
      if (tid < j) yv -= sL[j][tid] * yj;
    }
    sy[tid] = yv;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < NB && c0 + tid < n_pose) y[c0 + tid] = sy[tid];
  const int j = col_begin + blockIdx.x * 256 + tid;
  if (j < c0) {
    double acc = 0.0;
#pragma unroll 8
    for (int r = 0; r < NB; ++r) acc += S[(size_t)(c0 + r) * ld + j] * sy[r];
    y[j] -= acc;
  }
}

void launch_backsolve_step(hipStream_t s, const double* S, int ld, int kb, int n_pose, double* y, int col_begin) {
  const int cols = kb * NB - col_begin;
  const int grid = cols > 0 ? (cols + 255) / 256 : 1;
  hipLaunchKernelGGL(backsolve_step_kernel, dim3(grid), dim3(256), 0, s, S, ld, kb, n_pose, y, col_begin);
}

}  // namespace bsg
