// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of landmark_kernel and pairs_band_kernel
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
// Every kernel below moves a KNOWN number of bytes in one of those patterns; scripts/pmc_calib.sh runs this program under two PMC
// passes and prints counted / known per kernel.  The inputs are produced by a kernel of the same process just before (as the Jacobians
// are by the evaluation launch): a 25.6 MB stream sits in the Infinity Cache, exactly as in the solver.
//
//   calib_wide_copy        16 B per lane, contiguous over the wave (the guide's calibrated case: FETCH_SIZE x 2)
//   calib_rows48           a lane reads its 48-byte row as three 16-byte pieces + a 16-byte r (landmark_kernel pass 1), no stores
//   calib_rows48_twice     the same rows read again after a dependent reduction (landmark_kernel's two passes), 64-byte CR row stored
//   calib_rows48_lds       the two-pass form with the rows kept in LDS for the second pass (what the kernel does from round 6 on)
//   calib_zero_tiles       64 x 64 tiles of a matrix of leading dimension ld zeroed with 16-byte stores (the step's clearing)
//   calib_rows176          a 16-lane group reads an observation's 96 + 64 + 16 bytes as 16-byte pieces (pairs_band_kernel's loaders)
//   calib_atomics          FP64 atomic adds into a 1.5 MB region from every CU (the band units' hand-over of their tiles)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void calib_fill(double* p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + 1e-9 * (double)(i & 1023);
}
__global__ __launch_bounds__(256) void calib_wide_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void calib_rows48(int n, const double* __restrict__ JB, const double2* __restrict__ r, double* __restrict__ sink) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const double2* J2 = reinterpret_cast<const double2*>(JB + (size_t)f * 6);
  const double2 a = J2[0], b = J2[1], c = J2[2], rf = r[f];
  const double s = a.x + a.y + b.x + b.y + c.x + c.y + rf.x + rf.y;
  if (s == 12345.678) sink[0] = s;   // (never true: keeps the loads)
}
template <bool LDS>
__global__ __launch_bounds__(256) void calib_rows48_twice(int n, const double* __restrict__ JB, const double2* __restrict__ r, double* __restrict__ CR) {
  __shared__ double2 keep[LDS ? 256 * 4 : 1];
  const int f = blockIdx.x * 256 + threadIdx.x;
  const bool valid = f < n;
  double h = 0.0;
  const double2* J2 = reinterpret_cast<const double2*>(JB + (size_t)(valid ? f : 0) * 6);
  {
    const double2 a = J2[0], b = J2[1], c = J2[2], rf = r[valid ? f : 0];
    h = a.x * a.x + a.y * b.y + b.x * c.x + c.y * rf.x + rf.y;
    if (LDS) { keep[threadIdx.x] = a; keep[256 + threadIdx.x] = b; keep[512 + threadIdx.x] = c; keep[768 + threadIdx.x] = rf; }
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) h += __shfl_xor(h, o, 8);   // (the per-landmark reduction over 8 lanes: the second pass depends on it)
  const double inv = 1.0 / (1.0 + h * h);
  if (!valid) return;
  double2 a, b, c, rf;
  if (LDS) { a = keep[threadIdx.x]; b = keep[256 + threadIdx.x]; c = keep[512 + threadIdx.x]; rf = keep[768 + threadIdx.x]; }
  else { a = J2[0]; b = J2[1]; c = J2[2]; rf = r[f]; }
  double2* o2 = reinterpret_cast<double2*>(CR + (size_t)f * 8);
  o2[0] = make_double2(a.x * inv, a.y * inv); o2[1] = make_double2(b.x * inv, b.y * inv); o2[2] = make_double2(c.x * inv, c.y * inv);
  o2[3] = make_double2(rf.x - inv, rf.y - inv);
}
__global__ __launch_bounds__(256) void calib_zero_tiles(double* S, int ld, int n_tiles) {
  const int nt = ld >> 6;
  for (int q = blockIdx.x; q < n_tiles; q += gridDim.x) {
    const int ti = q / nt, tj = q - ti * nt;
    double2* base = reinterpret_cast<double2*>(S + (size_t)ti * 64 * ld + (size_t)tj * 64);
    const int r0 = threadIdx.x >> 5, c2 = threadIdx.x & 31;
#pragma unroll
    for (int p = 0; p < 8; ++p) base[(size_t)(r0 + 8 * p) * (ld >> 1) + c2] = make_double2(0.0, 0.0);
  }
}
__global__ __launch_bounds__(512) void calib_rows176(int n, const double* __restrict__ J, const double* __restrict__ CR, const double2* __restrict__ r, double* __restrict__ sink) {
  // sixteen lanes per observation: pieces 0..5 the 96-byte A row, 6..9 the 64-byte C | rho row, 10 the residual (clamped, as the kernel's loaders are)
  const int t = blockIdx.x * 512 + threadIdx.x, f = t >> 4, p = t & 15;
  if (f >= n) return;
  double2 v;
  if (p < 6) v = reinterpret_cast<const double2*>(J + (size_t)f * 12)[p];
  else if (p < 10) v = reinterpret_cast<const double2*>(CR + (size_t)f * 8)[p - 6];
  else v = r[f];
  if (v.x + v.y == 12345.678) sink[0] = v.x;
}
__global__ __launch_bounds__(512) void calib_atomics(double* S, int n_words, int per_unit) {
  // a unit adds per_unit values into a window of the region that overlaps its neighbours' (a band unit's 78 x 78 block overlaps the next pose's)
  const int base = (int)(((size_t)blockIdx.x * 1237) % (size_t)(n_words - 8192));
  for (int i = threadIdx.x; i < per_unit; i += 512) atomicAdd(S + base + (i * 7) % 8192, 1.0);
}

int main() {
  const int n = 400690;                       // C2's observations
  const size_t nJB = (size_t)n * 6, nr = (size_t)n * 2, nCR = (size_t)n * 8, nJ = (size_t)n * 12;
  double *JB, *r, *CR, *J, *sink, *S, *wide_in, *wide_out;
  const int ld = 3776, n_tiles_total = (ld / 64) * (ld / 64);
  CHECK(hipMalloc(&JB, nJB * 8)); CHECK(hipMalloc(&r, nr * 8)); CHECK(hipMalloc(&CR, nCR * 8)); CHECK(hipMalloc(&J, nJ * 8));
  CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&S, (size_t)ld * ld * 8));
  const size_t wide_bytes = (size_t)n * 64;   // 25.6 MB, the size of the landmark kernel's input stream
  CHECK(hipMalloc(&wide_in, wide_bytes)); CHECK(hipMalloc(&wide_out, wide_bytes));
  auto fill_inputs = [&]() {
    hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, JB, nJB, 0.5);
    hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, r, nr, 0.25);
    hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, J, nJ, 0.75);
    hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, CR, nCR, 0.125);
    hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, wide_in, wide_bytes / 8, 1.5);
  };
  std::printf("CALIB known bytes per launch: wide_copy read %zu write %zu | rows48 read %zu | rows48_twice read %zu (once) write %zu | zero_tiles write %zu | rows176 read %zu | atomics %zu adds of 8 B\n",
              wide_bytes, wide_bytes, (size_t)n * 64, (size_t)n * 64, (size_t)n * 64, (size_t)600 * 32768, (size_t)n * 176, (size_t)200 * 6400);
  for (int rep = 0; rep < 5; ++rep) {
    fill_inputs();
    hipLaunchKernelGGL(calib_wide_copy, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const double2*>(wide_in), reinterpret_cast<double2*>(wide_out), wide_bytes / 16);
    fill_inputs();
    hipLaunchKernelGGL(calib_rows48, dim3((n + 255) / 256), dim3(256), 0, 0, n, JB, reinterpret_cast<const double2*>(r), sink);
    fill_inputs();
    hipLaunchKernelGGL(calib_rows48_twice<false>, dim3((n + 255) / 256), dim3(256), 0, 0, n, JB, reinterpret_cast<const double2*>(r), CR);
    fill_inputs();
    hipLaunchKernelGGL(calib_rows48_twice<true>, dim3((n + 255) / 256), dim3(256), 0, 0, n, JB, reinterpret_cast<const double2*>(r), CR);
    hipLaunchKernelGGL(calib_zero_tiles, dim3(600), dim3(256), 0, 0, S, ld, 600 < n_tiles_total ? 600 : n_tiles_total);
    fill_inputs();
    hipLaunchKernelGGL(calib_rows176, dim3((n * 16 + 511) / 512), dim3(512), 0, 0, n, J, CR, reinterpret_cast<const double2*>(r), sink);
    hipLaunchKernelGGL(calib_atomics, dim3(200), dim3(512), 0, 0, S, 187500, 6400);
    CHECK(hipDeviceSynchronize());
  }
  std::printf("CALIB done\n");
  return 0;
}
