// One-off probe: wall-clock stamps (100 MHz) around the two triangular-solve variants and the rank-64 update of a panel task.
//   hipcc --offload-arch=gfx950 -O3 -I include -I beam_slam_amd/csrc scripts/trsm_probe.hip -o scripts/trsm_probe.bin
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../beam_slam_amd/csrc/k_chol.hip"
namespace bsg {
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
