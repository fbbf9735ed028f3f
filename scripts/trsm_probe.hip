// One-off probe: wall-clock stamps (100 MHz) around the triangular-solve variants of a panel task and its rank-64 update: the LDS
// round-trip substitution (3.3 us), the register-resident transposed substitution the fused kernel uses (3.0 us), and the product form
// X = A W^T with the full tile inverse W (1.6 us, but W itself costs 2.2 us to build on MFMA: not adopted).
//   hipcc --offload-arch=gfx950 -O3 -I include -I beam_slam_amd/csrc scripts/trsm_probe.hip -o scripts/trsm_probe.bin
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../beam_slam_amd/csrc/k_chol.hip"
namespace bsg {
// W = L^-1 of a factored 64x64 tile (full lower-triangular inverse, 16x16 blocks) from L (sL) and the inverses of its diagonal blocks
// (sV), into sW (pitch LDT; blocks above the diagonal are not written).  With W a panel's triangular solves become plain products
// X = A W^T whose four accumulator chains are independent (trsm_gemm_w): 1.6 us per tile against 3.0 / 3.3 for the substitution
// forms (scripts/trsm_probe.hip).  Wave j builds block column j top-down — W_jj = V_j, W_ij = -V_i sum_{k=j..i-1} L_ik W_kj — out of its
// own registers: the MFMA result layout of W_kj is the B-operand layout of the next product.  Wave 3 only copies V_3.
BSG_DEV void tile_inverse_w(const double* sL, const double* sV, double* sW, int lane, int wave) {
  const int n = lane & 15, q = lane >> 4, j = wave;
  double4_t w[4];   // W_kj for k = j .. 3 (index k), result layout: w[k][reg] = W_kj[q + 4 reg][n]
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) w[0][reg] = 0.0;
  // W_jj = V_j
  double4_t wjj;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) wjj[reg] = sV[j * 256 + (q + 4 * reg) * 16 + n];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) if (jj == j) w[jj] = wjj;
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    if (i <= j) continue;   // (wave-uniform)
    double4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < j || k >= i) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(sL[(16 * i + n) * LDT + 16 * k + 4 * kk + q], w[k][kk], t, 0, 0, 0);
    }
    double4_t r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) r = __builtin_amdgcn_mfma_f64_16x16x4f64(-sV[i * 256 + n * 16 + 4 * kk + q], t[kk], r, 0, 0, 0);
    w[i] = r;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < j) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sW[(16 * i + q + 4 * reg) * LDT + 16 * j + n] = w[i][reg];
  }
}
// X strip = A strip W^T: Y_b = X_b^T = sum_{c<=b} W_bc A_c^T — four independent accumulator chains, B operands (A^T) from registers,
// A operands (W) from LDS.  y[b][reg] = X[row n of the strip][16 b + q + 4 reg]: exactly the A-operand layout of the rank-64 update
// (K slice 4 b + reg), so the solved strip feeds C -= X_i X_j^T without touching LDS.
BSG_DEV void trsm_gemm_w(const double* sA, const double* sW, int lane, int wave, double4_t (&y)[4]) {
  const double* rows = sA + (16 * wave) * LDT;
  const int n = lane & 15, q = lane >> 4;
  double4_t at[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) at[c][reg] = rows[n * LDT + 16 * c + q + 4 * reg];   // A_c^T in B-operand layout (K slice reg)
#pragma unroll
  for (int b = 0; b < 4; ++b) y[b] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int b = c; b < 4; ++b)
        y[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sW[(16 * b + n) * LDT + 16 * c + 4 * kk + q], at[c][kk], y[b], 0, 0, 0);
}

__global__ __launch_bounds__(256) void probe(const double* A, const double* L, double* X, long long* ts, int variant) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sX = smem; double* sL = sX + NB * LDT; double* sV = sL + NB * LDT; double* sT = sV + 1024; double* sInvD = sT + 4 * 16 * 17;
  double* sC = sInvD + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; sC[r * LDT + c] = (c <= r) ? L[r * NB + c] : 0.0; sX[r * LDT + c] = A[r * NB + c]; }
  __syncthreads();
  potrf64_lds(sC, sV, sInvD, tid, 64);
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; sL[r * LDT + c] = sC[r * LDT + c]; }
  __syncthreads();
  long long tw0 = 0, tw1 = 0;
  if (variant == 2) {
    __syncthreads();
    tw0 = wall_clock64();
    tile_inverse_w(sL, sV, sC, lane, wave);
    __syncthreads();
    tw1 = wall_clock64();
    if (tid == 0) ts[2] = tw1 - tw0;
  }
  long long t0 = wall_clock64();
  if (variant == 0) trsm_tile(sX, sL, sV, sT, lane, wave);
  else if (variant == 1) trsm_tile_t(sX, sL, sV, lane, wave);
  else {
    double4_t y[4];
    trsm_gemm_w(sX, sC, lane, wave, y);
    const int n = lane & 15, q = lane >> 4;
    double* rows = sX + (16 * wave) * LDT;
    __builtin_amdgcn_wave_barrier();
    for (int b = 0; b < 4; ++b) for (int reg = 0; reg < 4; ++reg) rows[n * LDT + 16 * b + q + 4 * reg] = y[b][reg];
  }
  __syncthreads();
  long long t1 = wall_clock64();
  double4_t acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = double4_t{0, 0, 0, 0};
  for (int t = 0; t < 4; ++t) acc[t] = mfma_abt<64>(acc[t], sX + (16 * wave) * LDT, LDT, sX + (16 * t) * LDT, LDT, -1.0, lane);
  for (int t = 0; t < 4; ++t) store_d(sC + (16 * wave) * LDT + 16 * t, LDT, lane, acc[t]);
  __syncthreads();
  long long t2 = wall_clock64();
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; X[r * NB + c] = sX[r * LDT + c]; }
  if (tid == 0) { ts[0] = t1 - t0; ts[1] = t2 - t1; }
}
}
int main() {
  const int n = 64;
  std::vector<double> A(n * n), L(n * n), X0(n * n), X1(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { L[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j)); A[i * n + j] = sin(i * 0.37 + j * 1.3); }
  double *dA, *dL, *dX; long long* ts;
  hipMalloc(&dA, 8 * n * n); hipMalloc(&dL, 8 * n * n); hipMalloc(&dX, 8 * n * n); hipMalloc(&ts, 64);
  hipMemcpy(dA, A.data(), 8 * n * n, hipMemcpyHostToDevice); hipMemcpy(dL, L.data(), 8 * n * n, hipMemcpyHostToDevice);
  const size_t lds = sizeof(double) * (3 * 64 * 66 + 1024 + 4 * 16 * 17 + 64 + 64 * 66);
  hipFuncSetAttribute(reinterpret_cast<const void*>(bsg::probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int variant = 0; variant < 3; ++variant)
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(bsg::probe, dim3(1), dim3(256), lds, 0, dA, dL, dX, ts, variant);
      hipDeviceSynchronize();
      long long h[3]; hipMemcpy(h, ts, 24, hipMemcpyDeviceToHost);
      if (variant == 2) printf("   tile_inverse_w %lld ticks\n", h[2]);
      hipMemcpy(variant == 0 ? X0.data() : X1.data(), dX, 8 * n * n, hipMemcpyDeviceToHost);
      printf("variant %d rep %d: trsm %lld ticks (10 ns)  update %lld ticks\n", variant, rep, h[0], h[1]);
    }
  double md = 0; for (int i = 0; i < n * n; ++i) md = fmax(md, fabs(X0[i] - X1[i]));
  printf("max |X_lds - X_reg| = %.3e\n", md);
  return 0;
}
