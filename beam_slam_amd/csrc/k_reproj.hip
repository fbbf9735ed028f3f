// Visual (landmark) part of the solve path on gfx950:
//   reproj_eval      residual + tangent Jacobian of EuclideanReprojection (A5/A6)          HBM-bound
//   landmark         per-landmark Hll, its 3x3 factor, C = B M, rho = r - C z               HBM-bound
//   pairs            camera-pair segments: S_ij = [i==j] A^T A - sum A^T C C'^T A', reduced rhs
//   backsub          landmark back-substitution
//   mcc              model cost change -(J d).(r + J d / 2)
// Reference for the arithmetic: bs_constraints/include/bs_constraints/visual/euclidean_reprojection_function.h:66-172
// (residual; the quaternion Jacobian there is a forward difference — here it is the closed form
// -A Jpi R_cb [P_b]x of SURVEY.md Appendix A) and bs_constraints/src/jacobians.cpp:202-214.
#include "bsgpu_device.h"
#include "marg_body.h"
#include "reproj_body.h"

namespace bsg {

// (the body: reproj_body.h — shared with the launch that also carries the IMU factors, k_small.hip)
template <bool WITH_J>
__global__ __launch_bounds__(256) void reproj_eval_kernel(int n, const int4* __restrict__ fac,
                                                          const double2* __restrict__ pix,
                                                          const double* __restrict__ wgt, const double* __restrict__ x,
                                                          const DevCamera* __restrict__ cams,
                                                          const DevLoss* __restrict__ losses,
                                                          double2* __restrict__ r_out, double* __restrict__ J_out,
                                                          double* __restrict__ JB_out, double* __restrict__ cost_part, int count_inactive) {
  __shared__ __attribute__((aligned(16))) double sJ[kReprojStage<WITH_J>];
  reproj_eval_body<WITH_J>((int)blockIdx.x, n, fac, pix, wgt, x, cams, losses, r_out, J_out, JB_out, cost_part, count_inactive, sJ);
}

void launch_reproj_eval(hipStream_t s, const Visual& v, const double* x, const DevCamera* cams,
                        const DevLoss* losses, bool with_J, double* cost_part_out, bool count_inactive) {
  if (v.n == 0) return;
  const int grid = (v.n + 255) / 256;
  if (with_J)
    hipLaunchKernelGGL(reproj_eval_kernel<true>, dim3(grid), dim3(256), 0, s, v.n, v.fac, v.pix, v.w, x, cams, losses,
                       v.r, v.J, v.JB, cost_part_out, count_inactive ? 1 : 0);
  else
    hipLaunchKernelGGL(reproj_eval_kernel<false>, dim3(grid), dim3(256), 0, s, v.n, v.fac, v.pix, v.w, x, cams, losses,
                       v.r, v.J, v.JB, cost_part_out, count_inactive ? 1 : 0);
}
void launch_reproj_jacobian_only(hipStream_t s, const Visual& v, const double* x, const DevCamera* cams,
                                 const DevLoss* losses) {
  launch_reproj_eval(s, v, x, cams, losses, true, v.cost_part, false);
}

// ---------------------------------------------------------------------------------------------------
// plain pixel error |z - pi(T_cam_baselink T_baselink_world P)| of every reprojection factor at x: the screening
// quantity of bs_models/src/visual_odometry.cpp:1247-1272 (ComputeAverageReprojection), un-weighted, no loss.
// Points at or behind the camera give -1.
// ---------------------------------------------------------------------------------------------------
BSG_DEV double pixel_error(const double* __restrict__ x, int xq, int xp, int xl, const DevCamera& cam, double u, double v) {
  const double q[4] = {x[xq], x[xq + 1], x[xq + 2], x[xq + 3]};
  const double t[3] = {x[xp], x[xp + 1], x[xp + 2]};
  const double P[3] = {x[xl], x[xl + 1], x[xl + 2]};
  double R[9], a[3], b[3], Pc[3];
  quat_to_rot(q, R);
  mat3t_vec(R, P, a);
  mat3t_vec(R, t, b);
  const double Pb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  mat3_vec(cam.R, Pb, Pc);
  Pc[0] += cam.t[0]; Pc[1] += cam.t[1]; Pc[2] += cam.t[2];
  if (!(Pc[2] > 0.0)) return -1.0;
  const double du = u - (cam.fx * Pc[0] / Pc[2] + cam.cx), dv = v - (cam.fy * Pc[1] / Pc[2] + cam.cy);
  return sqrt(du * du + dv * dv);
}
__global__ void reproj_error_kernel(int n, const int4* __restrict__ fac, const double2* __restrict__ pix, const double* __restrict__ x,
                                    const DevCamera* __restrict__ cams, double* __restrict__ out) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const int4 fc = fac[f];
  const double2 z = pix[f];
  out[f] = pixel_error(x, fc.x, fc.y, fc.z, cams[fc.w & ((1 << kMetaCamBits) - 1)], z.x, z.y);
}
__global__ void reproj_error_small_kernel(SmallGroup g, const double* __restrict__ x, double* __restrict__ out) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= g.n) return;
  const int* xo = g.xoff + (size_t)f * 3;
  const double* k = g.consts + (size_t)f * 3;
  out[f] = pixel_error(x, xo[0], xo[1], xo[2], g.cams[g.cam[f]], k[0], k[1]);
}
void launch_reproj_errors(hipStream_t s, const Visual& v, const SmallGroup& dense, const double* x, const DevCamera* cams, double* out_vis,
                          double* out_dense) {
  if (v.n) hipLaunchKernelGGL(reproj_error_kernel, dim3((v.n + 255) / 256), dim3(256), 0, s, v.n, v.fac, v.pix, x, cams, out_vis);
  if (dense.n) hipLaunchKernelGGL(reproj_error_small_kernel, dim3((dense.n + 255) / 256), dim3(256), 0, s, dense, x, out_dense);
}

// ---------------------------------------------------------------------------------------------------
// per-landmark: Hll = sum B^T B (+ lambda), g_l = sum B^T r, 3x3 Cholesky, then per factor
// C = B Linv^T (2x3) and rho = r - C z with z = Linv g_l.   8 lanes per landmark.
// Jacobi scaling (Ceres: s = 1/(1+sqrt(H_jj)) from iteration 0) and the LM diagonal are folded into
// lambda_j = clamp(s_j^2 H_jj, lo, hi) / (radius s_j^2) on the unscaled system.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void landmark_kernel_body(const int bsg_bx, const int bsg_gx, int n_lm, const int* __restrict__ lm_start, const double* __restrict__ JB, const double2* __restrict__ r, int n_pose, const double* __restrict__ radius_ptr, int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* __restrict__ scale, double* __restrict__ dcl, double* __restrict__ grad, double* __restrict__ Linv_out, double* __restrict__ z_out, double* __restrict__ CR, int lm_blocks, ZeroStep zs, double radius_val, const double* dec = nullptr) {
  if (bsg_bx >= lm_blocks) {
    // the step's clearing (tiles of S the assembly writes, pose gradient, diag(J^T J), the step's scalars, the radius slot) as extra
    // workgroups of this launch: nothing here is read or written by the landmark workgroups (they get the radius as an argument),
    // and a launch of its own cost 6 us on the dependent path
    const int zb = bsg_bx - lm_blocks, nzb = bsg_gx - lm_blocks;
    const int64_t t = (int64_t)zb * 256 + threadIdx.x, stride = (int64_t)nzb * 256;
    if (zs.radius_slot && t == 0) *zs.radius_slot = zs.radius;
    const int nt = zs.ld >> 6;
    for (int q = zb; q < zs.n_tiles; q += nzb) {
      const int ti = zs.tiles[q] / nt, tj = zs.tiles[q] - ti * nt;
      double2* base = reinterpret_cast<double2*>(zs.S + (size_t)ti * 64 * zs.ld + (size_t)tj * 64);
      const int r0 = threadIdx.x >> 5, c2 = threadIdx.x & 31;
#pragma unroll
      for (int p = 0; p < 8; ++p) base[(size_t)(r0 + 8 * p) * (zs.ld >> 1) + c2] = make_double2(0.0, 0.0);
    }
    for (int64_t i = t; i < zs.na; i += stride) zs.a[i] = 0.0;
    for (int64_t i = t; i < zs.nb; i += stride) zs.b[i] = 0.0;
    for (int64_t i = t; i < zs.nc; i += stride) zs.c[i] = 0.0;
    return;
  }
  const int gid = bsg_bx * 256 + threadIdx.x;
  const int l = gid >> 3, sub = gid & 7;
  const bool valid = l < n_lm;
  int beg = 0, end = 0;
  if (valid) { beg = lm_start[l]; end = lm_start[l + 1]; }
  double h00 = 0, h01 = 0, h02 = 0, h11 = 0, h12 = 0, h22 = 0, b0 = 0, b1 = 0, b2 = 0;
  for (int f = beg + sub; f < end; f += 8) {
    // (16-byte pieces: the 48-byte row is 16-byte aligned; a lane's six 8-byte loads were six look-ups of the same lines)
    const double2* Jf2 = reinterpret_cast<const double2*>(JB + (size_t)f * 6);
    const double2 rf = r[f];
    const double2 ja = Jf2[0], jb = Jf2[1], jc = Jf2[2];
    const double x0 = ja.x, x1 = ja.y, x2 = jb.x, y0 = jb.y, y1 = jc.x, y2 = jc.y;
    h00 += x0 * x0 + y0 * y0; h01 += x0 * x1 + y0 * y1; h02 += x0 * x2 + y0 * y2;
    h11 += x1 * x1 + y1 * y1; h12 += x1 * x2 + y1 * y2; h22 += x2 * x2 + y2 * y2;
    b0 += x0 * rf.x + y0 * rf.y; b1 += x1 * rf.x + y1 * rf.y; b2 += x2 * rf.x + y2 * rf.y;
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    h00 += __shfl_xor(h00, o, 8); h01 += __shfl_xor(h01, o, 8); h02 += __shfl_xor(h02, o, 8);
    h11 += __shfl_xor(h11, o, 8); h12 += __shfl_xor(h12, o, 8); h22 += __shfl_xor(h22, o, 8);
    b0 += __shfl_xor(b0, o, 8); b1 += __shfl_xor(b1, o, 8); b2 += __shfl_xor(b2, o, 8);
  }
  // the radius: an argument; device-resident under graph replay, whose arguments are frozen; or — an assembly ahead of the host's decision —
  // what the reduction riding in this launch's first workgroups decides (bsgpu_device.h lm_decide): waited for here, behind the loads and
  // the sums (the workgroup's first wave looks, the others wait at the barrier), and nothing is written when the step it closes was not accepted
  double radius_now = radius_ptr ? radius_ptr[0] : radius_val;
  if (dec) {
    __shared__ double s_radius;
    if (threadIdx.x < 64) {
      const double rr = wait_decision(dec, bsg_bx);
      if (threadIdx.x == 0) s_radius = rr;
    }
    __syncthreads();
    radius_now = s_radius;
    if (!(radius_now > 0.0)) return;
  }
  if (!valid) return;
  const double inv_radius = 1.0 / radius_now;
  const int to = n_pose + 3 * l;
  const double hd[3] = {h00, h11, h22};
  double lam[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double sc;
    if (compute_scale) sc = jacobi ? 1.0 / (1.0 + sqrt(hd[i])) : 1.0; else sc = scale[to + i];
    double d;
    if (compute_dcl) d = fmin(fmax(sc * sc * hd[i], lm_lo), lm_hi) / (sc * sc); else d = dcl[to + i];
    lam[i] = d * inv_radius;
    if (sub == 0) {
      if (compute_scale) scale[to + i] = sc;
      if (compute_dcl) dcl[to + i] = d;
    }
  }
  if (sub == 0) { grad[to] = b0; grad[to + 1] = b1; grad[to + 2] = b2; }
  // Cholesky of Hll + lambda, Li = L^-1 (lower)
  const double a00 = h00 + lam[0], a11 = h11 + lam[1], a22 = h22 + lam[2];
  const double l00 = sqrt(a00), l10 = h01 / l00, l20 = h02 / l00;
  const double l11 = sqrt(a11 - l10 * l10), l21 = (h12 - l20 * l10) / l11;
  const double l22 = sqrt(a22 - l20 * l20 - l21 * l21);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  const double z0 = i00 * b0, z1 = i10 * b0 + i11 * b1, z2 = i20 * b0 + i21 * b1 + i22 * b2;
  if (sub == 0) {
    double* Lo = Linv_out + (size_t)l * kLmRec;
    Lo[0] = i00; Lo[1] = i10; Lo[2] = i11; Lo[3] = i20; Lo[4] = i21; Lo[5] = i22;
    double* zo = z_out + (size_t)l * kLmRec;
    zo[0] = z0; zo[1] = z1; zo[2] = z2;
  }
  if (CR == nullptr) return;   // (Visual::no_cr: the pair phase and the back-substitution form C and rho themselves)
  for (int f = beg + sub; f < end; f += 8) {
    const double2* Jf2 = reinterpret_cast<const double2*>(JB + (size_t)f * 6);
    const double2 rf = r[f];
    double2* o2 = reinterpret_cast<double2*>(CR + (size_t)f * 8);
    // C[k][j] = sum_i B[k][i] Linv[j][i]
    const double2 ja = Jf2[0], jb = Jf2[1], jc = Jf2[2];
    const double x0 = ja.x, x1 = ja.y, x2 = jb.x, y0 = jb.y, y1 = jc.x, y2 = jc.y;
    const double c00 = x0 * i00, c01 = x0 * i10 + x1 * i11, c02 = x0 * i20 + x1 * i21 + x2 * i22;
    const double c10 = y0 * i00, c11 = y0 * i10 + y1 * i11, c12 = y0 * i20 + y1 * i21 + y2 * i22;
    o2[0] = make_double2(c00, c01); o2[1] = make_double2(c02, c10); o2[2] = make_double2(c11, c12);
    o2[3] = make_double2(rf.x - (c00 * z0 + c01 * z1 + c02 * z2), rf.y - (c10 * z0 + c11 * z1 + c12 * z2));
  }
}
__global__ __launch_bounds__(256) void landmark_kernel(int n_lm, const int* __restrict__ lm_start, const double* __restrict__ JB, const double2* __restrict__ r, int n_pose, const double* __restrict__ radius_ptr, int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* __restrict__ scale, double* __restrict__ dcl, double* __restrict__ grad, double* __restrict__ Linv_out, double* __restrict__ z_out, double* __restrict__ CR, int lm_blocks, ZeroStep zs, double radius_val) {
  landmark_kernel_body((int)blockIdx.x, (int)gridDim.x, n_lm, lm_start, JB, r, n_pose, radius_ptr, compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, grad, Linv_out, z_out, CR, lm_blocks, zs, radius_val);
}
// ... with the end-of-step reduction of the step BEFORE as its first workgroups: the landmark launch of an assembly issued ahead of the host's
// decision (bsgpu_solve.cpp enqueue_step) follows the evaluation of the candidate with Jacobians, which has left the candidate's cost
// partials — the reduction rides here instead of in that evaluation, and the cost-only pass at the candidate is not needed
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void landmark_reduce_kernel(ReduceRide red, int n_lm, const int* __restrict__ lm_start, const double* __restrict__ JB, const double2* __restrict__ r, int n_pose, const double* __restrict__ radius_ptr, int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* __restrict__ scale, double* __restrict__ dcl, double* __restrict__ grad, double* __restrict__ Linv_out, double* __restrict__ z_out, double* __restrict__ CR, int lm_blocks, ZeroStep zs, double radius_val) {
  // (only_slot: the reduction's other units rode in the evaluation in front of this launch; the one whose partial sums that evaluation wrote is here, alone)
  const int n_units = red.only_slot >= 0 ? 1 : red.n_slots + 1;
  if ((int)blockIdx.x < n_units) {
    __shared__ double sred[16];
    final_reduce_unit<256>(red.only_slot >= 0 ? red.only_slot : (int)blockIdx.x, (int)threadIdx.x, red, red.n_slots + 1, sred);
    return;
  }
  landmark_kernel_body((int)blockIdx.x - n_units, (int)gridDim.x - n_units, n_lm, lm_start, JB, r, n_pose, radius_ptr, compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, grad, Linv_out, z_out, CR, lm_blocks, zs, radius_val,
                       (red.lmd.on && red.lmd.on != 2) ? red.dec : nullptr);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct landmark_kernel_Args {
  int bsg_grid;
  int n_lm;
  const int* lm_start;
  const double* JB;
  const double2* r;
  int n_pose;
  const double* radius_ptr;
  int compute_scale;
  int compute_dcl;
  int jacobi;
  double lm_lo;
  double lm_hi;
  double* scale;
  double* dcl;
  double* grad;
  double* Linv_out;
  double* z_out;
  double* CR;
  int lm_blocks;
  ZeroStep zs;
  double radius_val;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct landmark_kernel_ArgsG {
  int bsg_grid;
  int n_lm;
  const int __attribute__((address_space(1)))* lm_start;
  const double __attribute__((address_space(1)))* JB;
  const double2 __attribute__((address_space(1)))* r;
  int n_pose;
  const double __attribute__((address_space(1)))* radius_ptr;
  int compute_scale;
  int compute_dcl;
  int jacobi;
  double lm_lo;
  double lm_hi;
  double __attribute__((address_space(1)))* scale;
  double __attribute__((address_space(1)))* dcl;
  double __attribute__((address_space(1)))* grad;
  double __attribute__((address_space(1)))* Linv_out;
  double __attribute__((address_space(1)))* z_out;
  double __attribute__((address_space(1)))* CR;
  int lm_blocks;
  ZeroStep zs;
  double radius_val;
};
static_assert(sizeof(landmark_kernel_ArgsG) == sizeof(landmark_kernel_Args), "layout");

__global__ __launch_bounds__(256) void landmark_kernel_batch(const landmark_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const landmark_kernel_ArgsG& a = reinterpret_cast<const landmark_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  // what changes from iteration to iteration: the radius, whether the Jacobi scale / LM diagonal are (re)computed, what the step's clearing covers
  const double radius_val = bsg_dyn->radius[bsg_w];
  const int compute_scale = bsg_dyn->first[bsg_w], compute_dcl = bsg_dyn->new_J[bsg_w];
  ZeroStep zs = a.zs;
  zs.radius = radius_val;
  if (!compute_dcl) { zs.c += SC_CHOL_FAIL - SC_GRAD_MAX; zs.nc = 1; }
  landmark_kernel_body((int)blockIdx.x, a.bsg_grid, a.n_lm, (const int*)a.lm_start, (const double*)a.JB, (const double2*)a.r, a.n_pose, nullptr, compute_scale, compute_dcl, a.jacobi, a.lm_lo, a.lm_hi, (double*)a.scale, (double*)a.dcl, (double*)a.grad, (double*)a.Linv_out, (double*)a.z_out, (double*)a.CR, a.lm_blocks, zs, radius_val);
}

// factors whose landmark is constant: C = 0, rho = r
__device__ __forceinline__ void landmark_tail_kernel_body(const int bsg_bx, const int bsg_gx, int first, int n, const double2* __restrict__ r, double* __restrict__ CR) {
  const int f = first + bsg_bx * 256 + threadIdx.x;
  if (f >= n) return;
  double* o = CR + (size_t)f * 8;
  const double2 rf = r[f];
  o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0;
  o[6] = rf.x; o[7] = rf.y;
}
__global__ void landmark_tail_kernel(int first, int n, const double2* __restrict__ r, double* __restrict__ CR) {
  landmark_tail_kernel_body((int)blockIdx.x, (int)gridDim.x, first, n, r, CR);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct landmark_tail_kernel_Args {
  int bsg_grid;
  int first;
  int n;
  const double2* r;
  double* CR;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct landmark_tail_kernel_ArgsG {
  int bsg_grid;
  int first;
  int n;
  const double2 __attribute__((address_space(1)))* r;
  double __attribute__((address_space(1)))* CR;
};
static_assert(sizeof(landmark_tail_kernel_ArgsG) == sizeof(landmark_tail_kernel_Args), "layout");

__global__ void landmark_tail_kernel_batch(const landmark_tail_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const landmark_tail_kernel_ArgsG& a = reinterpret_cast<const landmark_tail_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  landmark_tail_kernel_body((int)blockIdx.x, a.bsg_grid, a.first, a.n, (const double2*)a.r, (double*)a.CR);
}

void launch_landmark(hipStream_t s, const Visual& v, int n_pose, const double* radius_ptr, int compute_scale,
                     int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl,
                     double* grad, const ZeroStep* zero, double radius_val, const ReduceRide* red) {
  if (v.n_lm > 0) {
    const int grid = (v.n_lm * 8 + 255) / 256;
    const int zero_blocks = zero ? std::max(1, std::min(zero->n_tiles, 1024)) : 0;
    if (red && red->n_entries > 0)
      hipLaunchKernelGGL(landmark_reduce_kernel, dim3((red->only_slot >= 0 ? 1 : red->n_slots + 1) + grid + zero_blocks), dim3(256), 0, s, *red, v.n_lm, v.lm_start, v.JB, v.r, n_pose, radius_ptr,
                         compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, grad, v.Linv, v.z, v.no_cr ? nullptr : v.CR, grid, zero ? *zero : ZeroStep(),
                         radius_val);
    else
    hipLaunchKernelGGL(landmark_kernel, dim3(grid + zero_blocks), dim3(256), 0, s, v.n_lm, v.lm_start, v.JB, v.r, n_pose, radius_ptr,
                       compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, grad, v.Linv, v.z, v.no_cr ? nullptr : v.CR, grid, zero ? *zero : ZeroStep(),
                       radius_val);
  }
  if (v.n > v.n_elim) {
    const int grid = (v.n - v.n_elim + 255) / 256;
    hipLaunchKernelGGL(landmark_tail_kernel, dim3(grid), dim3(256), 0, s, v.n_elim, v.n, v.r, v.CR);
  }
}

// ---------------------------------------------------------------------------------------------------
// camera-pair segments.  One wave per (camera pose i, camera pose j >= i) pair; lanes stride over the
// (factor a of i, factor b of j, same landmark) entries:
//   block(i,j) = sum A_a^T ([a==b] I2 - C_a C_b^T) A_b          (6x6)
// diagonal segments also give the reduced rhs sum A^T rho, the raw gradient sum A^T r and diag(A^T A).
// Results are added to the dense reduced system with FP64 atomics (each location is normally owned by
// one segment, so the sums are reproducible).
// ---------------------------------------------------------------------------------------------------
constexpr int kPairRiderParts = 4;   // workgroups (single waves) per pose-only factor riding in the pair launch
__device__ __forceinline__ void pairs_kernel_body(const int bsg_bx, const int bsg_gx, int n_seg, const int* __restrict__ seg_ci, const int* __restrict__ seg_cj, const int* __restrict__ seg_start, const int* __restrict__ ent_fa, const int* __restrict__ ent_fb, const double* __restrict__ J, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only, int n_pair_blocks, const SmallGroupSet& small, int n_small_units) {
  extern __shared__ __attribute__((aligned(16))) double2 slab[];   // [64][6] A_a | [64][6] A_b | [64][4] C_a | [64][3] C_b | 2 x 64 ints
  if (bsg_bx < n_small_units) {
    // the pose-only factors assembled one workgroup per factor (IMU: two or three hundred of them) as extra workgroups of this launch —
    // the FIRST ones, so that they run underneath the pairs, not after them: independent atomics into the same system, and a launch of
    // their own cost ~8 us on the dependent path
    double* sJ = reinterpret_cast<double*>(slab);
    small_assemble_unit(small, bsg_bx / kPairRiderParts, threadIdx.x, 64, sJ, sJ + 15 * 30, reinterpret_cast<int*>(sJ + 15 * 30 + 16), S, ld, rhs_row, grad, hdiag, perm,
                        bsg_bx % kPairRiderParts, kPairRiderParts);
    return;
  }
  // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Segments are
  // ordered by camera pair (ca, cb), and all segments of one ca gather the same J / CR rows, so every XCD takes a
  // CONTIGUOUS range of segments: the rows of a camera are then pulled into one L2 instead of eight.
  const int pb = bsg_bx - n_small_units;   // (n_small_units is a multiple of 8: the XCD round-robin is unchanged)
  const int per_xcd = n_pair_blocks >> 3;
  const int seg = (pb & 7) * per_xcd + (pb >> 3);
  if (seg >= n_seg) return;
  const int lane = threadIdx.x;
  const int ci = seg_ci[seg], cj = seg_cj[seg];
  const bool diag = ci == cj;
  if (grad_only && !diag) return;   // (end of a solve: only the gradient of the accepted point is wanted, bsgpu_solve.cpp)
  double blk[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) blk[i] = 0.0;
  double gr[6], gg[6], hd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { gr[i] = 0.0; gg[i] = 0.0; hd[i] = 0.0; }
  const int beg = seg_start[seg], end = seg_start[seg + 1];
  // The rows are gathered CO-OPERATIVELY: with one lane per entry every 16-byte load of a wave touches 64 different cache lines, and
  // the kernel ran at the rate the L1 looks lines up (19 loads x 64 lines per 64 entries; measured: removing the arithmetic changed
  // nothing, halving the occupancy cost 27 %, rows laid out densely — 48 lines per load — gave exactly 3/4 of the time).  Six lanes
  // now fetch the six pieces of one 96-byte row (one or two lines), four / three lanes a C row, into the wave's LDS slab, and every
  // lane reads its entry's rows back from there: ~330 line look-ups per 64 entries instead of 1 216.
  double2* sAa = slab; double2* sAb = sAa + 64 * 6; double2* sCa = sAb + 64 * 6; double2* sCb = sCa + 64 * 4;
  int* sfa = reinterpret_cast<int*>(sCb + 64 * 3); int* sfb = sfa + 64;
  // (the entry indices of the NEXT stride are requested while this one is gathered and summed: one round trip less per stride)
  int fa_n = ent_fa[min(beg + lane, end - 1)], fb_n = ent_fb[min(beg + lane, end - 1)];
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int e = e0 + lane;
    const bool live = e < end;
    const int fa = fa_n, fb = fb_n;
    if (e0 + 64 < end) { fa_n = ent_fa[min(e + 64, end - 1)]; fb_n = ent_fb[min(e + 64, end - 1)]; }
    sfa[lane] = fa; sfb[lane] = fb;
    __builtin_amdgcn_wave_barrier();
    {
      const double2* J2 = reinterpret_cast<const double2*>(J);
      const double2* C2 = reinterpret_cast<const double2*>(CR);
      double2 va[6], vb[6], vc[4], vd[3];
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int p = it * 64 + lane, row = p / 6, piece = p - 6 * row;
        va[it] = J2[(size_t)sfa[row] * (kJAStride / 2) + piece];
        vb[it] = J2[(size_t)sfb[row] * (kJAStride / 2) + piece];
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = it * 64 + lane, row = p >> 2, piece = p & 3;
        vc[it] = C2[(size_t)sfa[row] * 4 + piece];
      }
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int p = it * 64 + lane, row = p / 3, piece = p - 3 * row;
        vd[it] = C2[(size_t)sfb[row] * 4 + piece];
      }
#pragma unroll
      for (int it = 0; it < 6; ++it) { sAa[it * 64 + lane] = va[it]; sAb[it * 64 + lane] = vb[it]; }
#pragma unroll
      for (int it = 0; it < 4; ++it) sCa[it * 64 + lane] = vc[it];
#pragma unroll
      for (int it = 0; it < 3; ++it) sCb[it * 64 + lane] = vd[it];
    }
    __builtin_amdgcn_wave_barrier();
    const double2* Ja = sAa + lane * 6;
    const double2* Jb = sAb + lane * 6;
    const double2* Ca2 = sCa + lane * 4;
    const double2* Cb2 = sCb + lane * 3;
    double A0[6], A1[6], B0[6], B1[6], Ca[8], Cb[6];
    {
      const double2 a0 = Ja[0], a1 = Ja[1], a2 = Ja[2], a3 = Ja[3], a4 = Ja[4], a5 = Ja[5];
      A0[0] = a0.x; A0[1] = a0.y; A0[2] = a1.x; A0[3] = a1.y; A0[4] = a2.x; A0[5] = a2.y;
      A1[0] = a3.x; A1[1] = a3.y; A1[2] = a4.x; A1[3] = a4.y; A1[4] = a5.x; A1[5] = a5.y;
      const double2 b0 = Jb[0], b1 = Jb[1], b2 = Jb[2], b3 = Jb[3], b4 = Jb[4], b5 = Jb[5];
      B0[0] = b0.x; B0[1] = b0.y; B0[2] = b1.x; B0[3] = b1.y; B0[4] = b2.x; B0[5] = b2.y;
      B1[0] = b3.x; B1[1] = b3.y; B1[2] = b4.x; B1[3] = b4.y; B1[4] = b5.x; B1[5] = b5.y;
      const double2 c0 = Ca2[0], c1 = Ca2[1], c2 = Ca2[2], c3 = Ca2[3];
      Ca[0] = c0.x; Ca[1] = c0.y; Ca[2] = c1.x; Ca[3] = c1.y; Ca[4] = c2.x; Ca[5] = c2.y; Ca[6] = c3.x; Ca[7] = c3.y;
      const double2 d0 = Cb2[0], d1 = Cb2[1], d2 = Cb2[2];
      Cb[0] = d0.x; Cb[1] = d0.y; Cb[2] = d1.x; Cb[3] = d1.y; Cb[4] = d2.x; Cb[5] = d2.y;
    }
    __builtin_amdgcn_wave_barrier();   // (the slab is rewritten by the next stride)
    if (!live) continue;
    const double same = (fa == fb) ? 1.0 : 0.0;
    const double t00 = same - (Ca[0] * Cb[0] + Ca[1] * Cb[1] + Ca[2] * Cb[2]);
    const double t01 = -(Ca[0] * Cb[3] + Ca[1] * Cb[4] + Ca[2] * Cb[5]);
    const double t10 = -(Ca[3] * Cb[0] + Ca[4] * Cb[1] + Ca[5] * Cb[2]);
    const double t11 = same - (Ca[3] * Cb[3] + Ca[4] * Cb[4] + Ca[5] * Cb[5]);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double u0 = t00 * B0[c] + t01 * B1[c];
      const double u1 = t10 * B0[c] + t11 * B1[c];
#pragma unroll
      for (int a = 0; a < 6; ++a) blk[a * 6 + c] += A0[a] * u0 + A1[a] * u1;
    }
    if (fa == fb) {
      const double2 rf = r[fa];
      const double p0 = Ca[6], p1 = Ca[7];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        gr[a] += A0[a] * p0 + A1[a] * p1;
        gg[a] += A0[a] * rf.x + A1[a] * rf.y;
        hd[a] += A0[a] * A0[a] + A1[a] * A1[a];
      }
    }
  }
  // all 54 sums in one transposed butterfly: lane l ends up with the total of value l (0..35 the block, 36.. the three 6-vectors
  // of a diagonal segment) — the same bits as 54 separate wave_sum()s at a sixth of the exchanges, and no LDS round trip
  double v[64];
#pragma unroll
  for (int i = 0; i < 36; ++i) v[i] = blk[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) { v[36 + i] = gr[i]; v[42 + i] = gg[i]; v[48 + i] = hd[i]; }
#pragma unroll
  for (int i = 54; i < 64; ++i) v[i] = 0.0;
  wave_sum_transpose64(v);
  const double total = v[0];
  const int tqi = cp_tq[ci], tpi = cp_tp[ci], tqj = cp_tq[cj], tpj = cp_tp[cj];
  if (lane < 36) {
    const int a = lane / 6, c = lane % 6;
    const int row = (a < 3) ? (tqi < 0 ? -1 : tqi + a) : (tpi < 0 ? -1 : tpi + a - 3);
    const int col = (c < 3) ? (tqj < 0 ? -1 : tqj + c) : (tpj < 0 ? -1 : tpj + c - 3);
    if (row >= 0 && col >= 0) {
      const int sr = perm[row], sc = perm[col];   // solver positions
      atomicAdd(&S[(size_t)sr * ld + sc], total);
      if (!diag) atomicAdd(&S[(size_t)sc * ld + sr], total);
    }
  } else if (diag && lane < 54) {
    const int a = (lane - 36) % 6, which = (lane - 36) / 6;   // 0: reduced rhs, 1: raw gradient, 2: diag(A^T A)
    const int row = (a < 3) ? (tqi < 0 ? -1 : tqi + a) : (tpi < 0 ? -1 : tpi + a - 3);
    if (row >= 0) {
      if (which == 0) atomicAdd(&S[(size_t)rhs_row * ld + perm[row]], total);
      else if (which == 1) atomicAdd(&grad[row], total);
      else atomicAdd(&hdiag[row], total);
    }
  }
}
__global__ __launch_bounds__(64) void pairs_kernel(int n_seg, const int* __restrict__ seg_ci, const int* __restrict__ seg_cj, const int* __restrict__ seg_start, const int* __restrict__ ent_fa, const int* __restrict__ ent_fb, const double* __restrict__ J, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only, int n_pair_blocks, SmallGroupSet small, int n_small_units, GoWord go) {
  if (go.p && !(__hip_atomic_load(go.p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0.0)) return;   // (an assembly ahead whose step was not accepted)
  pairs_kernel_body((int)blockIdx.x, (int)gridDim.x, n_seg, seg_ci, seg_cj, seg_start, ent_fa, ent_fb, J, r, CR, cp_tq, cp_tp, S, ld, rhs_row, grad, hdiag, perm, grad_only, n_pair_blocks, small, n_small_units);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct pairs_kernel_Args {
  int bsg_grid;
  int n_seg;
  const int* seg_ci;
  const int* seg_cj;
  const int* seg_start;
  const int* ent_fa;
  const int* ent_fb;
  const double* J;
  const double2* r;
  const double* CR;
  const int* cp_tq;
  const int* cp_tp;
  double* S;
  int ld;
  int rhs_row;
  double* grad;
  double* hdiag;
  const int* perm;
  int grad_only;
  int n_pair_blocks;
  SmallGroupSet small;
  int n_small_units;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct pairs_kernel_ArgsG {
  int bsg_grid;
  int n_seg;
  const int __attribute__((address_space(1)))* seg_ci;
  const int __attribute__((address_space(1)))* seg_cj;
  const int __attribute__((address_space(1)))* seg_start;
  const int __attribute__((address_space(1)))* ent_fa;
  const int __attribute__((address_space(1)))* ent_fb;
  const double __attribute__((address_space(1)))* J;
  const double2 __attribute__((address_space(1)))* r;
  const double __attribute__((address_space(1)))* CR;
  const int __attribute__((address_space(1)))* cp_tq;
  const int __attribute__((address_space(1)))* cp_tp;
  double __attribute__((address_space(1)))* S;
  int ld;
  int rhs_row;
  double __attribute__((address_space(1)))* grad;
  double __attribute__((address_space(1)))* hdiag;
  const int __attribute__((address_space(1)))* perm;
  int grad_only;
  int n_pair_blocks;
  SmallGroupSet small;
  int n_small_units;
};
static_assert(sizeof(pairs_kernel_ArgsG) == sizeof(pairs_kernel_Args), "layout");

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void pairs_kernel_batch(const pairs_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const pairs_kernel_ArgsG& a = reinterpret_cast<const pairs_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  pairs_kernel_body((int)blockIdx.x, a.bsg_grid, a.n_seg, (const int*)a.seg_ci, (const int*)a.seg_cj, (const int*)a.seg_start, (const int*)a.ent_fa, (const int*)a.ent_fb, (const double*)a.J, (const double2*)a.r, (double*)a.CR, (const int*)a.cp_tq, (const int*)a.cp_tp, (double*)a.S, a.ld, a.rhs_row, (double*)a.grad, (double*)a.hdiag, (const int*)a.perm, bsg_dyn->grad_only[bsg_w], a.n_pair_blocks, a.small, a.n_small_units);
}

constexpr size_t kPairsLds = sizeof(double2) * 64 * (6 + 6 + 4 + 3) + sizeof(int) * 128;
void launch_pairs(hipStream_t s, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm,
                  bool grad_only, const SmallGroupSet* small, int n_small_units, GoWord go) {
  if (v.n_seg == 0) return;
  const int pair_blocks = 8 * ((v.n_seg + 7) / 8);
  SmallGroupSet none;
  none.n = 0;
  const int small_blocks = small ? 8 * ((n_small_units * kPairRiderParts + 7) / 8) : 0;   // (padded: idle workgroups return at once)
  hipLaunchKernelGGL(pairs_kernel, dim3(pair_blocks + small_blocks), dim3(64), kPairsLds, s, v.n_seg, v.seg_ci, v.seg_cj, v.seg_start, v.ent_fa,
                     v.ent_fb, v.J, v.r, v.CR, v.cp_tq, v.cp_tp, S, ld, rhs_row, grad, hdiag, perm, grad_only ? 1 : 0, pair_blocks, small ? *small : none,
                     small_blocks, go);
}

// ---------------------------------------------------------------------------------------------------
// back-substitution: y_l = Linv^T (z - sum_f C_f^T (A_f y_cam(f)));  delta_l = -y_l.  8 lanes / landmark.
// The same lanes then add up the model cost change of the landmark's factors, sum -(J d).(r + J d / 2) (ceres
// TrustRegionMinimizer) with J d = -(A y_cam + B y_l): the rows are still in L1 and a launch of its own is saved.  Factors whose
// landmark is constant (f >= n_elim) are handled one per lane by the workgroups after the landmark ones.
// ---------------------------------------------------------------------------------------------------
BSG_DEV void pose_part(const double* __restrict__ Jf_row, int tq, int tp, const double* __restrict__ y_pose, double& j0, double& j1) {
  // (the 96-byte row as six 16-byte pieces instead of twelve 8-byte loads)
  const double2* J2 = reinterpret_cast<const double2*>(Jf_row);
  const double2 v0 = J2[0], v1 = J2[1], v2 = J2[2], v3 = J2[3], v4 = J2[4], v5 = J2[5];
  const double Jf[12] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y, v4.x, v4.y, v5.x, v5.y};
  j0 = 0.0; j1 = 0.0;
  if (tq >= 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double yv = y_pose[tq + k]; j0 += Jf[k] * yv; j1 += Jf[6 + k] * yv; }
  }
  if (tp >= 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double yv = y_pose[tp + k]; j0 += Jf[3 + k] * yv; j1 += Jf[9 + k] * yv; }
  }
}
__device__ __forceinline__ void backsub_mcc_kernel_body(const int bsg_bx, const int bsg_gx, int n_lm, int n_lm_groups, int n_elim, int n, const int* __restrict__ lm_start, const double* __restrict__ J, const double* __restrict__ JB, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cam_pose, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, const double* __restrict__ Linv, const double* __restrict__ z, int n_pose, const double* __restrict__ y_pose, double* __restrict__ delta, double* __restrict__ mcc_part, int n_vis_blocks, const SmallGroupSet& small, int n_small_units, const UpdateRide& up, int first_update_block) {
  __shared__ double sred[4];
  if (up.n_blocks > 0 && bsg_bx >= first_update_block) {
    // the candidate of every block but the Euclidean landmarks (those: below, by the lanes that compute their step): the pose step is
    // complete when this launch starts, so the update needs no launch of its own behind it
    const int unit = bsg_bx - first_update_block, i = unit * 256 + (int)threadIdx.x;
    double d2 = 0.0, x2 = 0.0;
    if (i < up.n_blocks) update_block(up.blocks[i], up.xoff, up.toff, up.size, up.manifold, up.x, delta, up.x_cand, d2, x2);
    const double a = block_sum_256(d2, sred);
    const double c = block_sum_256(x2, sred);
    if (threadIdx.x == 0) { up.part[2 * unit] = a; up.part[2 * unit + 1] = c; }
    return;
  }
  if (bsg_bx >= n_vis_blocks) {
    // the model-cost terms of the pose-only factors (they need the pose step only) as extra workgroups: two 128-row units each
    const int unit = 2 * (bsg_bx - n_vis_blocks) + ((int)threadIdx.x >> 7);
    if (unit < n_small_units) small_mcc_unit(small, unit, threadIdx.x & 127, delta, sred + 2 * (threadIdx.x >> 7));
    else __syncthreads();
    return;
  }
  double acc = 0.0, upd_d2 = 0.0, upd_x2 = 0.0;
  if (bsg_bx < n_lm_groups) {
    const int gid = bsg_bx * 256 + threadIdx.x;
    const int l = gid >> 3, sub = gid & 7;
    const bool valid = l < n_lm;
    int beg = 0, end = 0;
    if (valid) { beg = lm_start[l]; end = lm_start[l + 1]; }
    // (what depends on the landmark alone is asked for with its range, not after the sums below: two trips off the dependent path)
    const int lc = valid ? l : 0;
    const double* Li = Linv + (size_t)lc * kLmRec;
    const double Li0 = Li[0], Li1 = Li[1], Li2 = Li[2], Li3 = Li[3], Li4 = Li[4], Li5 = Li[5];
    const double* zl = z + (size_t)lc * kLmRec;
    const double zl0 = zl[0], zl1 = zl[1], zl2 = zl[2];
    double a0 = 0, a1 = 0, a2 = 0;
    // (A y_cam of the lane's first two factors stays in registers for the second pass: a track longer than 16 views is rare, and
    // recomputing it costs the row again plus a three-deep chain of dependent gathers — camera pose id, its tangent offsets, y)
    double jk0[2] = {0.0, 0.0}, jk1[2] = {0.0, 0.0};
    int it = 0;
    // (no C rows — Visual::no_cr: the sums are taken with the B rows, sum_a B_a^T (A_a y), and C^T = Linv B^T is applied to them once per landmark below)
    const bool no_cr = CR == nullptr;
    for (int f = beg + sub; f < end; f += 8, ++it) {
      const double2* C2 = no_cr ? reinterpret_cast<const double2*>(JB + (size_t)f * 6) : reinterpret_cast<const double2*>(CR + (size_t)f * 8);
      const double2 ca = C2[0], cb = C2[1], cc = C2[2];
      const double C[6] = {ca.x, ca.y, cb.x, cb.y, cc.x, cc.y};
      const int cp = cam_pose[f];
      double j0, j1;
      pose_part(J + (size_t)f * kJAStride, cp_tq[cp], cp_tp[cp], y_pose, j0, j1);
      if (it == 0) { jk0[0] = j0; jk1[0] = j1; } else if (it == 1) { jk0[1] = j0; jk1[1] = j1; }
      a0 += C[0] * j0 + C[3] * j1; a1 += C[1] * j0 + C[4] * j1; a2 += C[2] * j0 + C[5] * j1;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 8); a1 += __shfl_xor(a1, o, 8); a2 += __shfl_xor(a2, o, 8); }
    if (no_cr) {   // a <- Linv a   (Linv lower triangular: Li0 | Li1 Li2 | Li3 Li4 Li5)
      const double b0 = a0, b1 = a1, b2 = a2;
      a0 = Li0 * b0; a1 = Li1 * b0 + Li2 * b1; a2 = Li3 * b0 + Li4 * b1 + Li5 * b2;
    }
    if (valid) {   // (every one of the 8 lanes holds the sums)
      const double w0 = zl0 - a0, w1 = zl1 - a1, w2 = zl2 - a2;
      // y = Linv^T w
      const double y0 = Li0 * w0 + Li1 * w1 + Li3 * w2;
      const double y1 = Li2 * w1 + Li4 * w2;
      const double y2 = Li5 * w2;
      if (sub == 0) {
        const int to = n_pose + 3 * l;
        delta[to] = -y0; delta[to + 1] = -y1; delta[to + 2] = -y2;
        if (up.n_blocks > 0) {   // the landmark's candidate and its terms of the step / x norms
          const int xo = up.lm_xoff[l];
          const double x0 = up.x[xo], x1 = up.x[xo + 1], x2v = up.x[xo + 2];
          const double c0 = x0 + (-y0), c1 = x1 + (-y1), c2 = x2v + (-y2);
          up.x_cand[xo] = c0; up.x_cand[xo + 1] = c1; up.x_cand[xo + 2] = c2;
          const double e0 = x0 - c0, e1 = x1 - c1, e2 = x2v - c2;
          upd_d2 = e0 * e0 + e1 * e1 + e2 * e2; upd_x2 = x0 * x0 + x1 * x1 + x2v * x2v;
        }
      }
      it = 0;
      for (int f = beg + sub; f < end; f += 8, ++it) {
        const double2* B2 = reinterpret_cast<const double2*>(JB + (size_t)f * 6);
        const double2 ba = B2[0], bb = B2[1], bc = B2[2];
        const double Bf[6] = {ba.x, ba.y, bb.x, bb.y, bc.x, bc.y};
        double j0, j1;
        if (it == 0) { j0 = jk0[0]; j1 = jk1[0]; }
        else if (it == 1) { j0 = jk0[1]; j1 = jk1[1]; }
        else { const int cp = cam_pose[f]; pose_part(J + (size_t)f * kJAStride, cp_tq[cp], cp_tp[cp], y_pose, j0, j1); }
        const double d0 = -(j0 + Bf[0] * y0 + Bf[1] * y1 + Bf[2] * y2), d1 = -(j1 + Bf[3] * y0 + Bf[4] * y1 + Bf[5] * y2);
        const double2 rf = r[f];
        acc -= d0 * (rf.x + 0.5 * d0) + d1 * (rf.y + 0.5 * d1);
      }
    }
  } else {
    const int f = n_elim + (bsg_bx - n_lm_groups) * 256 + (int)threadIdx.x;
    if (f < n) {
      const int cp = cam_pose[f];
      double j0, j1;
      pose_part(J + (size_t)f * kJAStride, cp_tq[cp], cp_tp[cp], y_pose, j0, j1);
      const double2 rf = r[f];
      acc = -((-j0) * (rf.x - 0.5 * j0) + (-j1) * (rf.y - 0.5 * j1));
    }
  }
  const double tot = block_sum_256(acc, sred);
  if (threadIdx.x == 0) mcc_part[bsg_bx] = tot;
  if (up.n_blocks > 0 && bsg_bx < n_lm_groups) {   // (after the update units' pairs in `part`)
    const double a = block_sum_256(upd_d2, sred);
    const double c = block_sum_256(upd_x2, sred);
    const int slot = (up.n_blocks + 255) / 256 + bsg_bx;
    if (threadIdx.x == 0) { up.part[2 * slot] = a; up.part[2 * slot + 1] = c; }
  }
}
__global__ __launch_bounds__(256) void backsub_mcc_kernel(int n_lm, int n_lm_groups, int n_elim, int n, const int* __restrict__ lm_start, const double* __restrict__ J, const double* __restrict__ JB, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cam_pose, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, const double* __restrict__ Linv, const double* __restrict__ z, int n_pose, const double* __restrict__ y_pose, double* __restrict__ delta, double* __restrict__ mcc_part, int n_vis_blocks, SmallGroupSet small, int n_small_units, UpdateRide up, int first_update_block) {
  backsub_mcc_kernel_body((int)blockIdx.x, (int)gridDim.x, n_lm, n_lm_groups, n_elim, n, lm_start, J, JB, r, CR, cam_pose, cp_tq, cp_tp, Linv, z, n_pose, y_pose, delta, mcc_part, n_vis_blocks, small, n_small_units, up, first_update_block);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct backsub_mcc_kernel_Args {
  int bsg_grid;
  int n_lm;
  int n_lm_groups;
  int n_elim;
  int n;
  const int* lm_start;
  const double* J;
  const double* JB;
  const double2* r;
  const double* CR;
  const int* cam_pose;
  const int* cp_tq;
  const int* cp_tp;
  const double* Linv;
  const double* z;
  int n_pose;
  const double* y_pose;
  double* delta;
  double* mcc_part;
  int n_vis_blocks;
  SmallGroupSet small;
  int n_small_units;
  UpdateRide up;
  int first_update_block;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct backsub_mcc_kernel_ArgsG {
  int bsg_grid;
  int n_lm;
  int n_lm_groups;
  int n_elim;
  int n;
  const int __attribute__((address_space(1)))* lm_start;
  const double __attribute__((address_space(1)))* J;
  const double __attribute__((address_space(1)))* JB;
  const double2 __attribute__((address_space(1)))* r;
  const double __attribute__((address_space(1)))* CR;
  const int __attribute__((address_space(1)))* cam_pose;
  const int __attribute__((address_space(1)))* cp_tq;
  const int __attribute__((address_space(1)))* cp_tp;
  const double __attribute__((address_space(1)))* Linv;
  const double __attribute__((address_space(1)))* z;
  int n_pose;
  const double __attribute__((address_space(1)))* y_pose;
  double __attribute__((address_space(1)))* delta;
  double __attribute__((address_space(1)))* mcc_part;
  int n_vis_blocks;
  SmallGroupSet small;
  int n_small_units;
  UpdateRide up;
  int first_update_block;
};
static_assert(sizeof(backsub_mcc_kernel_ArgsG) == sizeof(backsub_mcc_kernel_Args), "layout");

__global__ __launch_bounds__(256) void backsub_mcc_kernel_batch(const backsub_mcc_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const backsub_mcc_kernel_ArgsG& a = reinterpret_cast<const backsub_mcc_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  backsub_mcc_kernel_body((int)blockIdx.x, a.bsg_grid, a.n_lm, a.n_lm_groups, a.n_elim, a.n, (const int*)a.lm_start, (const double*)a.J, (const double*)a.JB, (const double2*)a.r, (double*)a.CR, (const int*)a.cam_pose, (const int*)a.cp_tq, (const int*)a.cp_tp, (const double*)a.Linv, (const double*)a.z, a.n_pose, (const double*)a.y_pose, (double*)a.delta, (double*)a.mcc_part, a.n_vis_blocks, a.small, a.n_small_units, a.up, a.first_update_block);
}

// ---- the same launches over several windows (bsgpu_batch.cpp): one table entry per window, grids as the lone launches compute them
void batchargs_landmark(BatchArgTable& t, BatchArgTable& t_tail, const Visual& v, int n_pose, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl,
                        double* grad, const ZeroStep& zero /* c = the step's scalars from SC_GRAD_MAX on, nc = 3 */) {
  landmark_kernel_Args a;
  const int grid = (v.n_lm * 8 + 255) / 256, zero_blocks = std::max(1, std::min(zero.n_tiles, 1024));
  a.bsg_grid = v.n_lm > 0 ? grid + zero_blocks : 0;
  a.n_lm = v.n_lm; a.lm_start = v.lm_start; a.JB = v.JB; a.r = v.r; a.n_pose = n_pose; a.radius_ptr = nullptr; a.compute_scale = 0; a.compute_dcl = 0;
  a.jacobi = jacobi; a.lm_lo = lm_lo; a.lm_hi = lm_hi; a.scale = scale; a.dcl = dcl; a.grad = grad; a.Linv_out = v.Linv; a.z_out = v.z; a.CR = v.no_cr ? nullptr : v.CR;   // (Visual::no_cr: no second pass, as in the lone launch)
  a.lm_blocks = grid; a.zs = zero; a.radius_val = 0.0;
  t.push(a);
  landmark_tail_kernel_Args b;
  b.bsg_grid = v.n > v.n_elim ? (v.n - v.n_elim + 255) / 256 : 0;
  b.first = v.n_elim; b.n = v.n; b.r = v.r; b.CR = v.CR;
  t_tail.push(b);
}
void launch_landmark_batch(hipStream_t s, const BatchArgTable& t, const BatchArgTable& t_tail, const BatchDyn* dyn, int list, int n) {
  if (n <= 0) return;
  if (t.max_grid > 0) hipLaunchKernelGGL(landmark_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const landmark_kernel_Args*>(t.dev), dyn, list);
  if (t_tail.max_grid > 0) hipLaunchKernelGGL(landmark_tail_kernel_batch, dim3(t_tail.max_grid, n), dim3(256), 0, s, static_cast<const landmark_tail_kernel_Args*>(t_tail.dev), dyn, list);
}
void batchargs_pairs(BatchArgTable& t, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* small,
                     int n_small_units) {
  pairs_kernel_Args a;
  const int pair_blocks = 8 * ((v.n_seg + 7) / 8), small_blocks = small ? 8 * ((n_small_units * kPairRiderParts + 7) / 8) : 0;
  SmallGroupSet none;
  none.n = 0;
  a.bsg_grid = v.n_seg > 0 ? pair_blocks + small_blocks : 0;
  a.n_seg = v.n_seg; a.seg_ci = v.seg_ci; a.seg_cj = v.seg_cj; a.seg_start = v.seg_start; a.ent_fa = v.ent_fa; a.ent_fb = v.ent_fb; a.J = v.J; a.r = v.r; a.CR = v.CR;
  a.cp_tq = v.cp_tq; a.cp_tp = v.cp_tp; a.S = S; a.ld = ld; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag; a.perm = perm; a.grad_only = 0;
  a.n_pair_blocks = pair_blocks; a.small = small ? *small : none; a.n_small_units = small_blocks;
  t.push(a);
}
void launch_pairs_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(pairs_kernel_batch, dim3(t.max_grid, n), dim3(64), kPairsLds, s, static_cast<const pairs_kernel_Args*>(t.dev), dyn, list);
}
int backsub_mcc_groups(const Visual& v);
void batchargs_backsub_mcc(BatchArgTable& t, const Visual& v, int n_pose, const double* y_pose, double* delta, double* mcc_part, const SmallGroupSet* small,
                           int n_small_units, const UpdateRide* upd) {
  backsub_mcc_kernel_Args a;
  const int g_lm = (v.n_lm * 8 + 255) / 256, grid = backsub_mcc_groups(v);
  const int extra = small ? (n_small_units + 1) / 2 : 0;
  const int upd_units = (upd && upd->n_blocks > 0) ? (upd->n_blocks + 255) / 256 : 0;
  a.bsg_grid = grid > 0 ? grid + extra + upd_units : 0;
  a.n_lm = v.n_lm; a.n_lm_groups = g_lm; a.n_elim = v.n_elim; a.n = v.n; a.lm_start = v.lm_start; a.J = v.J; a.JB = v.JB; a.r = v.r; a.CR = v.no_cr ? nullptr : v.CR; a.cam_pose = v.cam_pose;
  a.cp_tq = v.cp_tq; a.cp_tp = v.cp_tp; a.Linv = v.Linv; a.z = v.z; a.n_pose = n_pose; a.y_pose = y_pose; a.delta = delta; a.mcc_part = mcc_part; a.n_vis_blocks = grid;
  a.small = small ? *small : SmallGroupSet(); a.n_small_units = small ? n_small_units : 0; a.up = upd_units ? *upd : UpdateRide(); a.first_update_block = grid + extra;
  t.push(a);
}
void launch_backsub_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(backsub_mcc_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const backsub_mcc_kernel_Args*>(t.dev), dyn, list);
}
int backsub_mcc_groups(const Visual& v) { return (v.n_lm * 8 + 255) / 256 + (v.n - v.n_elim + 255) / 256; }
// ... with the model-cost terms of the window's dense prior (marg_body.h; they read the pose step only: the landmarks a prior names are not
// eliminated) as the launch's last workgroups, four rows each, instead of marg_mcc_kernel behind it (4.6 us)
__global__ __launch_bounds__(256) void backsub_mcc_marg_kernel(int n_lm, int n_lm_groups, int n_elim, int n, const int* __restrict__ lm_start, const double* __restrict__ J, const double* __restrict__ JB, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cam_pose, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, const double* __restrict__ Linv, const double* __restrict__ z, int n_pose, const double* __restrict__ y_pose, double* __restrict__ delta, double* __restrict__ mcc_part, int n_vis_blocks, SmallGroupSet small, int n_small_units, UpdateRide up, int first_update_block, MargDev m, double* __restrict__ marg_part, int first_marg_block) {
  if ((int)blockIdx.x >= first_marg_block) { marg_mcc_kernel_body(4 * ((int)blockIdx.x - first_marg_block), m, delta, marg_part); return; }
  backsub_mcc_kernel_body((int)blockIdx.x, first_marg_block, n_lm, n_lm_groups, n_elim, n, lm_start, J, JB, r, CR, cam_pose, cp_tq, cp_tp, Linv, z, n_pose, y_pose, delta, mcc_part, n_vis_blocks, small, n_small_units, up, first_update_block);
}
bool launch_backsub_mcc(hipStream_t s, const Visual& v, int n_pose, const double* y_pose, double* delta, double* mcc_part,
                        const SmallGroupSet* small, int n_small_units, const UpdateRide* upd, const MargDev* marg, double* marg_part) {
  const int g_lm = (v.n_lm * 8 + 255) / 256, grid = backsub_mcc_groups(v);
  if (grid == 0) return false;
  const int extra = small ? (n_small_units + 1) / 2 : 0;
  const int upd_units = (upd && upd->n_blocks > 0) ? (upd->n_blocks + 255) / 256 : 0;
  if (marg && marg_part && marg->rows > 0) {
    const int own = grid + extra + upd_units;
    hipLaunchKernelGGL(backsub_mcc_marg_kernel, dim3(own + (marg->rows + 3) / 4), dim3(256), 0, s, v.n_lm, g_lm, v.n_elim, v.n, v.lm_start, v.J, v.JB, v.r, v.no_cr ? nullptr : v.CR, v.cam_pose,
                       v.cp_tq, v.cp_tp, v.Linv, v.z, n_pose, y_pose, delta, mcc_part, grid, small ? *small : SmallGroupSet(), small ? n_small_units : 0,
                       upd_units ? *upd : UpdateRide(), grid + extra, *marg, marg_part, own);
    return true;
  }
  hipLaunchKernelGGL(backsub_mcc_kernel, dim3(grid + extra + upd_units), dim3(256), 0, s, v.n_lm, g_lm, v.n_elim, v.n, v.lm_start, v.J, v.JB, v.r, v.no_cr ? nullptr : v.CR, v.cam_pose,
                     v.cp_tq, v.cp_tp, v.Linv, v.z, n_pose, y_pose, delta, mcc_part, grid, small ? *small : SmallGroupSet(), small ? n_small_units : 0,
                     upd_units ? *upd : UpdateRide(), grid + extra);
  return false;
}

}  // namespace bsg
