// One-off probe: wall-clock stamps (100 MHz counter) inside potrf64_lds.  hipcc --offload-arch=gfx950 -O3 -I include -I beam_slam_amd/csrc scripts/potrf_probe.hip -o /tmp/potrf_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "../beam_slam_amd/csrc/k_chol.hip"
namespace bsg {
__global__ __launch_bounds__(256) void probe_kernel(double* S, int ld, long long* ts) {
  __shared__ double sC[NB * LDT];
  __shared__ double sV[4 * 256];
  __shared__ double sInvD[NB];
  const int tid = threadIdx.x;
  long long t0 = wall_clock64();
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; sC[r * LDT + c] = (c <= r) ? S[(size_t)r * ld + c] : 0.0; }
  __syncthreads();
  if (tid == 0) ts[0] = t0;
  potrf64_lds<true>(sC, sV, sInvD, tid, 64, ts + 1);
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; if (c <= r) S[(size_t)r * ld + c] = sC[r * LDT + c]; }
  __syncthreads();
  if (tid == 0) ts[15] = wall_clock64();
}
}
namespace bsg {
// same potrf, but in the LDS environment of the panel-step kernel: dynamic LDS of kPanelStepLds bytes, tile at the sXj offset
__global__ __launch_bounds__(256) void probe_dyn_kernel(double* S, int ld, long long* ts) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sC = smem;
  double* sV = smem + NB * LDT;
  double* sInvD = sV + 4 * 256;
  const int tid = threadIdx.x;
  for (int i = tid; i < NB * NB; i += 256) { const int r = i >> 6, c = i & 63; sC[r * LDT + c] = (c <= r) ? S[(size_t)r * ld + c] : 0.0; }
  __syncthreads();
  potrf64_lds<true>(sC, sV, sInvD, tid, 64, ts);
}
}
static void probe_dyn() {
  const int n = 64;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
  double* d; long long* ts;
  hipMalloc(&d, sizeof(double) * n * n); hipMalloc(&ts, sizeof(long long) * 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(bsg::probe_dyn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t need = sizeof(double) * (64 * 66 + 4 * 256 + 64);
  for (size_t extra : {(size_t)0, (size_t)0, (size_t)16384, (size_t)24576, (size_t)32768, (size_t)49152, (size_t)65536, (size_t)98304, (size_t)120000}) {
    const int rep = (int)(extra / 1024);
    hipMemcpy(d, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bsg::probe_dyn_kernel, dim3(1), dim3(256), need + extra, 0, d, n, ts);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    printf("dynamic-LDS potrf (%zu KB + %d KB):", need / 1024, rep);
    for (int b = 0; b < 4; ++b) printf(" elim%d %lld upd%d %lld |", b, h[1 + 2 * b] - h[2 * b], b, h[2 + 2 * b] - h[1 + 2 * b]);
    printf("\n");
  }
}

static void probe_panel_step() {
  // 3 tiles: panel k = 0 with row tiles {1, 2}; workgroup (0,0) = tile (1,1): trsm + update + look-ahead potrf
  const int T = 3, n = T * 64;
  std::vector<double> A((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (i == j ? 200.0 : 0.0) + 1.0 / (1 + abs(i - j));
  double *S, *Lp, *V, *scal; long long* ts; int *tiles, *nreal, *tile_sync;
  hipMalloc(&S, sizeof(double) * n * n); hipMalloc(&Lp, sizeof(double) * n * n); hipMalloc(&V, sizeof(double) * T * bsg::kVinvStride);
  hipMalloc(&scal, 256); hipMalloc(&ts, 32 * 8); hipMalloc(&tiles, 16); hipMalloc(&nreal, 16); hipMalloc(&tile_sync, 64);
  int h_tiles[1] = {0}, h_nreal[3] = {64, 64, 64};
  int h_sync[6] = {0, 1, 0, 0, 0, 0};   // [expected arrivals per tile | counters]: tile 1 gets its last (and only) update in this step
  hipMemcpy(tile_sync, h_sync, sizeof(h_sync), hipMemcpyHostToDevice);
  hipMemcpy(tiles, h_tiles, 4, hipMemcpyHostToDevice); hipMemcpy(nreal, h_nreal, 12, hipMemcpyHostToDevice);
  bsg::chol_prepare();
  hipFuncSetAttribute(reinterpret_cast<const void*>(bsg::chol_panel_step_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bsg::kPanelStepLds);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(S, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bsg::chol_potrf_tiles_kernel, dim3(1), dim3(256), 0, 0, S, Lp, n, tiles, nreal, V, scal);
    bsg::StepArgs a; memset(&a, 0, sizeof(a));
    a.k[0] = 0; a.n_rows[0] = 2; a.final_mask[0] = 1; a.rows[0][0] = 1; a.rows[0][1] = 2;
    hipLaunchKernelGGL((bsg::chol_panel_step_kernel<true, true>), dim3(2, 2, 1), dim3(256), bsg::kPanelStepLds, 0, S, Lp, n, nullptr, nullptr, nreal, V, scal, tile_sync, a, ts);
    hipDeviceSynchronize();
    long long h[32];
    hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    printf("panel step rep %d (10 ns ticks): loads %lld | trsm %lld | update mfma %lld | to-LDS+mask %lld | potrf %lld | write_factor %lld | total %lld\n", rep,
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
    printf("   potrf inside the panel step:");
    for (int b = 0; b < 4; ++b) printf(" elim%d %lld upd%d %lld |", b, h[9 + 2 * b] - h[8 + 2 * b], b, h[10 + 2 * b] - h[9 + 2 * b]);
    printf("\n");
  }
}

int main() {
  probe_dyn();
  probe_panel_step();
  const int n = 64;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 70.0 : 0.0) + 1.0 / (1 + abs(i - j));
  double* d; long long* ts;
  hipMalloc(&d, sizeof(double) * n * n); hipMalloc(&ts, sizeof(long long) * 16);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(d, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bsg::probe_kernel, dim3(1), dim3(256), 0, 0, d, n, ts);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost);
    printf("rep %d (10 ns ticks since kernel start): load %lld |", rep, h[1] - h[0]);
    for (int b = 0; b < 4; ++b) printf(" elim%d %lld upd%d %lld |", b, h[2 + 2 * b] - h[1 + 2 * b], b, h[3 + 2 * b] - h[2 + 2 * b]);
    printf(" inverses %lld | store %lld | total %lld\n", h[10] - h[9], h[15] - h[10], h[15] - h[0]);
  }
  return 0;
}
