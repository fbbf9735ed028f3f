// Manifold update x (+) delta, gradient norms, and the small deterministic reductions of the LM loop.
// Plus on quaternion blocks restates fuse's Orientation3DLocalParameterization::Plus, i.e.
// bs_constraints/src/jacobians.cpp:24-35 (x (x) AngleAxisToQuaternion(delta), right perturbation).
#include <algorithm>
#include <cstdint>

#include "bsgpu_device.h"

namespace bsg {

// (block_plus and the update of one block: bsgpu_device.h — shared with the launch that carries the update, k_reproj.hip)
// x_cand = x (+) delta for every block; per-workgroup partials of |x_cand - x|^2 and |x|^2 over the
// non-constant blocks (ceres: step_norm, x_norm)
__global__ __launch_bounds__(256) void update_kernel(int nb, const int* __restrict__ xoff, const int* __restrict__ toff,
                                                     const unsigned char* __restrict__ size,
                                                     const unsigned char* __restrict__ manifold,
                                                     const double* __restrict__ x, const double* __restrict__ delta,
                                                     double* __restrict__ x_cand, double* __restrict__ part) {
  __shared__ double sred[4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  double d2 = 0.0, x2 = 0.0;
  if (b < nb) update_block(b, xoff, toff, size, manifold, x, delta, x_cand, d2, x2);
  const double a = block_sum_256(d2, sred);
  const double c = block_sum_256(x2, sred);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = c; }
}

void launch_update(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                   const unsigned char* blk_manifold, const double* x, const double* delta, double* x_cand,
                   double* part, int* n_part) {
  const int grid = (nb + 255) / 256;
  *n_part = grid;
  hipLaunchKernelGGL(update_kernel, dim3(grid), dim3(256), 0, s, nb, blk_xoff, blk_toff, blk_size, blk_manifold, x, delta,
                     x_cand, part);
}

// gradient norms the way ceres' TrustRegionMinimizer defines them: |x - Plus(x, -g)|_inf and |.|_2.
// The max is order independent (atomicMax on the bit pattern of a non-negative double).
struct PoseDiagArgs {   // the pose_diag_kernel arguments, for the launch that does both (an accepted / first step)
  int n_pose, ld, compute_scale, compute_dcl, jacobi, npad;
  double* S; const double* hdiag; const double* radius_ptr; double lm_lo, lm_hi; double* scale; double* dcl; const int* iperm; double radius_val;
};
template <bool WITH_DIAG>
__device__ __forceinline__ void grad_norms_kernel_body(const int bsg_bx, const int bsg_gx, int nb, const int* __restrict__ xoff, const int* __restrict__ toff, const unsigned char* __restrict__ size, const unsigned char* __restrict__ manifold, const double* __restrict__ x, const double* __restrict__ grad, double* __restrict__ gpart, PoseDiagArgs pd) {
  __shared__ double sred[4];
  __shared__ double smax[4];
  const int b = bsg_bx * 256 + threadIdx.x;
  if (WITH_DIAG && b < pd.npad)   // independent of the norms: the LM diagonal of the reduced system rides in the same launch
    pose_diag_element(b, pd.n_pose, pd.S, pd.ld, pd.hdiag, 1.0 / (pd.radius_ptr ? pd.radius_ptr[0] : pd.radius_val), pd.compute_scale, pd.compute_dcl, pd.jacobi, pd.lm_lo,
                      pd.lm_hi, pd.scale, pd.dcl, pd.iperm);
  double mx = 0.0, s2 = 0.0;
  if (b < nb) {
    const int o = xoff[b], t = toff[b], sz = size[b];
    if (t >= 0) {
      const int ts = (manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : sz;
      double xin[4] = {0, 0, 0, 0}, din[4] = {0, 0, 0, 0}, out[4];
      for (int i = 0; i < sz && i < 4; ++i) xin[i] = x[o + i];
      for (int i = 0; i < ts && i < 4; ++i) din[i] = -grad[t + i];
      block_plus(manifold[b], sz, xin, din, out);
      for (int i = 0; i < sz && i < 4; ++i) {
        const double df = fabs(xin[i] - out[i]);
        mx = fmax(mx, df); s2 += df * df;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  const double tot = block_sum_256(s2, sred);
  if (threadIdx.x == 0) {
    // per-workgroup partials, reduced in a fixed order by final_reduce_kernel (two same-address atomics per workgroup — 400 of them
    // on C2 — were most of this kernel's 9 us, and made the norm's last bits depend on their order)
    gpart[2 * bsg_bx] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    gpart[2 * bsg_bx + 1] = tot;
  }
}
template <bool WITH_DIAG>
__global__ __launch_bounds__(256) void grad_norms_kernel(int nb, const int* __restrict__ xoff, const int* __restrict__ toff, const unsigned char* __restrict__ size, const unsigned char* __restrict__ manifold, const double* __restrict__ x, const double* __restrict__ grad, double* __restrict__ gpart, PoseDiagArgs pd) {
  grad_norms_kernel_body<WITH_DIAG>((int)blockIdx.x, (int)gridDim.x, nb, xoff, toff, size, manifold, x, grad, gpart, pd);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct grad_norms_kernel_Args {
  int bsg_grid;
  int nb;
  const int* xoff;
  const int* toff;
  const unsigned char* size;
  const unsigned char* manifold;
  const double* x;
  const double* grad;
  double* gpart;
  PoseDiagArgs pd;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct grad_norms_kernel_ArgsG {
  int bsg_grid;
  int nb;
  const int __attribute__((address_space(1)))* xoff;
  const int __attribute__((address_space(1)))* toff;
  const unsigned char __attribute__((address_space(1)))* size;
  const unsigned char __attribute__((address_space(1)))* manifold;
  const double __attribute__((address_space(1)))* x;
  const double __attribute__((address_space(1)))* grad;
  double __attribute__((address_space(1)))* gpart;
  PoseDiagArgs pd;
};
static_assert(sizeof(grad_norms_kernel_ArgsG) == sizeof(grad_norms_kernel_Args), "layout");

template <bool WITH_DIAG>
__global__ __launch_bounds__(256) void grad_norms_kernel_batch(const grad_norms_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const grad_norms_kernel_ArgsG& a = reinterpret_cast<const grad_norms_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  PoseDiagArgs pd = a.pd;
  pd.radius_ptr = nullptr; pd.radius_val = bsg_dyn->radius[bsg_w];
  pd.compute_scale = bsg_dyn->first[bsg_w]; pd.compute_dcl = bsg_dyn->new_J[bsg_w];
  grad_norms_kernel_body<WITH_DIAG>((int)blockIdx.x, a.bsg_grid, a.nb, (const int*)a.xoff, (const int*)a.toff, (const unsigned char*)a.size, (const unsigned char*)a.manifold, (const double*)a.x, (const double*)a.grad, (double*)a.gpart, pd);
}

void launch_grad_norms(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                       const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart) {
  hipLaunchKernelGGL(grad_norms_kernel<false>, dim3((nb + 255) / 256), dim3(256), 0, s, nb, blk_xoff, blk_toff, blk_size,
                     blk_manifold, x, grad, gpart, PoseDiagArgs{});
}
// gradient norms + pose_diag in one launch (both follow the assembly and are independent of each other)
void launch_grad_norms_pose_diag(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                                 const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart, int n_pose, double* S,
                                 int ld, const double* hdiag, const double* radius_ptr, int compute_scale, int compute_dcl, int jacobi,
                                 double lm_lo, double lm_hi, double* scale, double* dcl, int npad, const int* iperm, double radius_val) {
  PoseDiagArgs pd;
  pd.radius_val = radius_val;
  pd.n_pose = n_pose; pd.ld = ld; pd.compute_scale = compute_scale; pd.compute_dcl = compute_dcl; pd.jacobi = jacobi; pd.npad = npad;
  pd.S = S; pd.hdiag = hdiag; pd.radius_ptr = radius_ptr; pd.lm_lo = lm_lo; pd.lm_hi = lm_hi; pd.scale = scale; pd.dcl = dcl; pd.iperm = iperm;
  const int grid = (std::max(nb, npad) + 255) / 256;
  hipLaunchKernelGGL(grad_norms_kernel<true>, dim3(grid), dim3(256), 0, s, nb, blk_xoff, blk_toff, blk_size, blk_manifold, x, grad, gpart, pd);
}

// fixed-order sums of partial arrays (one workgroup): reproducible cost / model-cost-change values
__global__ __launch_bounds__(256) void sum_kernel(const double* __restrict__ part, int n, double* __restrict__ out,
                                                  int accumulate) {
  __shared__ double sred[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  const double t = block_sum_256(a, sred);
  if (threadIdx.x == 0) *out = accumulate ? (*out + t) : t;
}
void launch_sum(hipStream_t s, const double* part, int n, double* out, int accumulate) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, part, n, out, accumulate);
}
// One launch at the end of an LM step: workgroup `slot` adds up, in a fixed order, every partial array registered for that scalar
// (bsgpu_device.h final_reduce_unit — which also runs as the first workgroups of the evaluation launched ahead of the decision, k_small.hip).
__device__ __forceinline__ void final_reduce_kernel_body(const int bsg_bx, const int bsg_gx, const ReduceEntry* __restrict__ entries, int n_entries, int n_slots, double* __restrict__ scal, double* __restrict__ host_scal, int* counter, double seq) {
  __shared__ double sred[16];
  ReduceRide R;
  R.entries = entries; R.n_entries = n_entries; R.n_slots = n_slots; R.scal = scal; R.host_scal = host_scal; R.counter = counter; R.seq = seq;
  final_reduce_unit<1024>(bsg_bx, (int)threadIdx.x, R, bsg_gx, sred);
}
__global__ __launch_bounds__(1024) void final_reduce_kernel(const ReduceEntry* __restrict__ entries, int n_entries, int n_slots, double* __restrict__ scal, double* __restrict__ host_scal, int* counter, double seq) {
  final_reduce_kernel_body((int)blockIdx.x, (int)gridDim.x, entries, n_entries, n_slots, scal, host_scal, counter, seq);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct final_reduce_kernel_Args {
  int bsg_grid;
  const ReduceEntry* entries;
  int n_entries;
  int n_slots;
  double* scal;
  double* host_scal;
  int* counter;
  double seq;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct final_reduce_kernel_ArgsG {
  int bsg_grid;
  const ReduceEntry __attribute__((address_space(1)))* entries;
  int n_entries;
  int n_slots;
  double __attribute__((address_space(1)))* scal;
  double __attribute__((address_space(1)))* host_scal;
  int __attribute__((address_space(1)))* counter;
  double seq;
};
static_assert(sizeof(final_reduce_kernel_ArgsG) == sizeof(final_reduce_kernel_Args), "layout");

__global__ __launch_bounds__(1024) void final_reduce_kernel_batch(const final_reduce_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const final_reduce_kernel_ArgsG& a = reinterpret_cast<const final_reduce_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  final_reduce_kernel_body((int)blockIdx.x, a.bsg_grid, (const ReduceEntry*)a.entries, a.n_entries, a.n_slots, (double*)a.scal, (double*)a.host_scal, (int*)a.counter, bsg_dyn->seq[bsg_w]);
}
void launch_final_reduce(hipStream_t s, const ReduceEntry* entries, int n_entries, int n_slots, double* scal, double* host_scal, int* counter,
                         double seq) {
  if (n_entries > 0)
    hipLaunchKernelGGL(final_reduce_kernel, dim3(n_slots + 1), dim3(1024), 0, s, entries, n_entries, n_slots, scal, host_scal, counter, seq);
}

// plain kernels instead of hipMemsetAsync / hipMemcpyAsync for the buffers of an LM step: the runtime's fill / copy
// paths cost ~5.6 us each on the dependent chain (profiles/), and as graph nodes they are what stalls a captured step
__global__ __launch_bounds__(256) void zero_kernel(double2* __restrict__ p2, int64_t n2, double* __restrict__ tail, int ntail) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) p2[i] = make_double2(0.0, 0.0);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.0;
}
void launch_zero(hipStream_t s, double* p, int64_t n) {
  if (n <= 0) return;
  // p comes from hipMalloc (256-byte aligned) or from an even offset into such a buffer in every caller but the
  // scalar slots: handle a misaligned head by falling back to scalar stores for tiny ranges
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || n < 2) {
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, s, nullptr, 0, p, (int)std::min<int64_t>(n, 256));
    if (n > 256) (void)hipMemsetAsync(p + 256, 0, sizeof(double) * (size_t)(n - 256), s);
    return;
  }
  const int64_t n2 = n / 2;
  const int grid = (int)std::min<int64_t>((n2 + 255) / 256, 2048);
  hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<double2*>(p), n2, p + 2 * n2, (int)(n - 2 * n2));
}
// one launch for what an LM step clears: up to three small arrays and the reduced system — of which only the listed 64x64 tiles
// are ever written (dense_plan.h: touched_tiles): a banded window touches a fraction of the dense square (C2: 30 %; an
// 800-keyframe window: 7 %), the rest stays zero from finalize().  The step's trust-region radius rides in as an argument.
__device__ __forceinline__ void zero_tiles_multi_kernel_body(const int bsg_bx, const int bsg_gx, double* __restrict__ S, int ld, const int* __restrict__ tiles, int n_tiles,
                                                             double* __restrict__ a, int na, double* __restrict__ b, int nb,
                                                             double* __restrict__ c, int nc, double* __restrict__ radius_slot, double radius) {
  const int64_t t = (int64_t)bsg_bx * 256 + threadIdx.x, stride = (int64_t)bsg_gx * 256;
  if (radius_slot && t == 0) *radius_slot = radius;
  const int nt = ld >> 6;
  for (int q = bsg_bx; q < n_tiles; q += bsg_gx) {
    const int ti = tiles[q] / nt, tj = tiles[q] - ti * nt;
    double2* base = reinterpret_cast<double2*>(S + (size_t)ti * 64 * ld + (size_t)tj * 64);
    const int r0 = threadIdx.x >> 5, c2 = threadIdx.x & 31;
#pragma unroll
    for (int p = 0; p < 8; ++p) base[(size_t)(r0 + 8 * p) * (ld >> 1) + c2] = make_double2(0.0, 0.0);
  }
  for (int64_t i = t; i < na; i += stride) a[i] = 0.0;
  for (int64_t i = t; i < nb; i += stride) b[i] = 0.0;
  for (int64_t i = t; i < nc; i += stride) c[i] = 0.0;
}
__global__ __launch_bounds__(256) void zero_tiles_multi_kernel(double* __restrict__ S, int ld, const int* __restrict__ tiles, int n_tiles,
                                                               double* __restrict__ a, int na, double* __restrict__ b, int nb,
                                                               double* __restrict__ c, int nc, double* __restrict__ radius_slot, double radius) {
  zero_tiles_multi_kernel_body((int)blockIdx.x, (int)gridDim.x, S, ld, tiles, n_tiles, a, na, b, nb, c, nc, radius_slot, radius);
}
// one launch over several windows (bsgpu_batch.cpp): the start-of-step clearing of the windows that have no landmark launch to carry it and
// were not cleared at the end of their previous step; zs.c = the step's scalars from SC_GRAD_MAX on (nc = 3), the radius comes per round
struct zero_tiles_multi_kernel_Args {
  int bsg_grid;
  ZeroStep zs;
};
__global__ __launch_bounds__(256) void zero_tiles_multi_kernel_batch(const zero_tiles_multi_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const zero_tiles_multi_kernel_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  ZeroStep zs = a.zs;
  if (!bsg_dyn->new_J[bsg_w]) { zs.c += SC_CHOL_FAIL - SC_GRAD_MAX; zs.nc = 1; }
  zero_tiles_multi_kernel_body((int)blockIdx.x, a.bsg_grid, zs.S, zs.ld, zs.tiles, zs.n_tiles, zs.a, zs.na, zs.b, zs.nb, zs.c, zs.nc, zs.radius_slot, bsg_dyn->radius[bsg_w]);
}
void batchargs_zero_tiles_multi(BatchArgTable& t, const ZeroStep* zs /* null: the window clears elsewhere */) {
  zero_tiles_multi_kernel_Args a;
  a.zs = zs ? *zs : ZeroStep();
  a.bsg_grid = zs ? std::max(1, std::min(zs->n_tiles, 2048)) : 0;
  t.push(a);
}
void launch_zero_tiles_multi_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(zero_tiles_multi_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const zero_tiles_multi_kernel_Args*>(t.dev), dyn, list);
}
void launch_zero_tiles_multi(hipStream_t s, double* S, int ld, const int* tiles_dev, int n_tiles, double* a, int na, double* b, int nb, double* c,
                             int nc, double* radius_slot, double radius) {
  const int grid = std::max(1, std::min(n_tiles, 2048));
  hipLaunchKernelGGL(zero_tiles_multi_kernel, dim3(grid), dim3(256), 0, s, S, ld, tiles_dev, n_tiles, a, na, b, nb, c, nc, radius_slot, radius);
}
// up to four arrays cleared in ONE launch (the block-sparse path clears its matrix values, right-hand side, gradient and diagonal per step:
// four launches of ~5 us each on the dependent path)
__global__ __launch_bounds__(256) void zero4_kernel(double* __restrict__ p0, int64_t n0, double* __restrict__ p1, int64_t n1, double* __restrict__ p2,
                                                    int64_t n2, double* __restrict__ p3, int64_t n3) {
  const int64_t stride = (int64_t)gridDim.x * 256, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t i = t; i < n0; i += stride) p0[i] = 0.0;
  for (int64_t i = t; i < n1; i += stride) p1[i] = 0.0;
  for (int64_t i = t; i < n2; i += stride) p2[i] = 0.0;
  for (int64_t i = t; i < n3; i += stride) p3[i] = 0.0;
}
void launch_zero4(hipStream_t s, double* p0, int64_t n0, double* p1, int64_t n1, double* p2, int64_t n2, double* p3, int64_t n3) {
  const int64_t nmax = std::max(std::max(n0, n1), std::max(n2, n3));
  if (nmax <= 0) return;
  const int grid = (int)std::min<int64_t>((nmax + 255) / 256, 2048);
  hipLaunchKernelGGL(zero4_kernel, dim3(grid), dim3(256), 0, s, p0, n0, p1, n1, p2, n2, p3, n3);
}
__device__ __forceinline__ void copy_kernel_body(const int bsg_bx, const int bsg_gx, const double* __restrict__ src, double* __restrict__ dst, int64_t n, int nzero_after) {
  const int64_t stride = (int64_t)bsg_gx * 256;
  for (int64_t i = (int64_t)bsg_bx * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
  if (bsg_bx == 0 && (int)threadIdx.x < nzero_after) dst[n + threadIdx.x] = 0.0;
}
__global__ __launch_bounds__(256) void copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n, int nzero_after) {
  copy_kernel_body((int)blockIdx.x, (int)gridDim.x, src, dst, n, nzero_after);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct copy_kernel_Args {
  int bsg_grid;
  const double* src;
  double* dst;
  int64_t n;
  int nzero_after;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct copy_kernel_ArgsG {
  int bsg_grid;
  const double __attribute__((address_space(1)))* src;
  double __attribute__((address_space(1)))* dst;
  int64_t n;
  int nzero_after;
};
static_assert(sizeof(copy_kernel_ArgsG) == sizeof(copy_kernel_Args), "layout");

__global__ __launch_bounds__(256) void copy_kernel_batch(const copy_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const copy_kernel_ArgsG& a = reinterpret_cast<const copy_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  copy_kernel_body((int)blockIdx.x, a.bsg_grid, (const double*)a.src, (double*)a.dst, a.n, a.nzero_after);
}
// ---- the same launches over several windows (bsgpu_batch.cpp)
void batchargs_grad_norms_pose_diag(BatchArgTable& t, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                                    const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart, int n_pose, double* S, int ld,
                                    const double* hdiag, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, int npad, const int* iperm) {
  grad_norms_kernel_Args a;
  PoseDiagArgs pd;
  pd.radius_val = 0.0; pd.n_pose = n_pose; pd.ld = ld; pd.compute_scale = 0; pd.compute_dcl = 0; pd.jacobi = jacobi; pd.npad = npad;
  pd.S = S; pd.hdiag = hdiag; pd.radius_ptr = nullptr; pd.lm_lo = lm_lo; pd.lm_hi = lm_hi; pd.scale = scale; pd.dcl = dcl; pd.iperm = iperm;
  a.bsg_grid = (std::max(nb, npad) + 255) / 256;
  a.nb = nb; a.xoff = blk_xoff; a.toff = blk_toff; a.size = blk_size; a.manifold = blk_manifold; a.x = x; a.grad = grad; a.gpart = gpart; a.pd = pd;
  t.push(a);
}
void launch_grad_norms_pose_diag_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(grad_norms_kernel_batch<true>, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const grad_norms_kernel_Args*>(t.dev), dyn, list);
}
void batchargs_final_reduce(BatchArgTable& t, const ReduceEntry* entries, int n_entries, int n_slots, double* scal, double* host_scal, int* counter) {
  final_reduce_kernel_Args a;
  a.bsg_grid = n_entries > 0 ? n_slots + 1 : 0;
  a.entries = entries; a.n_entries = n_entries; a.n_slots = n_slots; a.scal = scal; a.host_scal = host_scal; a.counter = counter; a.seq = 0.0;
  t.push(a);
}
void launch_final_reduce_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(final_reduce_kernel_batch, dim3(t.max_grid, n), dim3(1024), 0, s, static_cast<const final_reduce_kernel_Args*>(t.dev), dyn, list);
}
void batchargs_copy(BatchArgTable& t, const double* src, double* dst, int64_t n) {
  copy_kernel_Args a;
  a.bsg_grid = n > 0 ? (int)std::min<int64_t>((n + 255) / 256, 256) : 0;
  a.src = src; a.dst = dst; a.n = n; a.nzero_after = 0;
  t.push(a);
}
void launch_copy_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(copy_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const copy_kernel_Args*>(t.dev), dyn, list);
}
// dst[0..n) = src[0..n), then nzero_after zeros
void launch_copy(hipStream_t s, const double* src, double* dst, int64_t n, int nzero_after) {
  if (n <= 0) return;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, src, dst, n, nzero_after);
}

__global__ void negate_kernel(int n, const double* __restrict__ y, double* __restrict__ d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = -y[i];
}
void launch_negate_pose(hipStream_t s, int n_pose, const double* y, double* delta) {
  if (n_pose > 0) hipLaunchKernelGGL(negate_kernel, dim3((n_pose + 255) / 256), dim3(256), 0, s, n_pose, y, delta);
}

}  // namespace bsg
