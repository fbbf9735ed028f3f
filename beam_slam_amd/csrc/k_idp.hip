// Inverse-depth landmarks (A7) eliminated on the landmark side.
// Reference for the factor: bs_constraints/include/bs_constraints/visual/inversedepth_reprojection_functor.h:57-125 (residual and
// its five parameter blocks: anchor orientation / position, measurement orientation / position, the scalar inverse depth); the
// elimination is what [EXT] Ceres' SCHUR solvers do with the landmark blocks of vo_params.json's windows (use_idp: true).
//
// A binary factor f of landmark l has the robustified row block  J_f = [A_a (2x6) | A_m (2x6) | w (2x1)]  (idp_kernel, k_small.hip:
// theta_a, p_a, theta_m, p_m, rho).  What the scalar landmark adds to the pose-pose part J^T J of its factors is
//   h = sum w^T w + lambda,   g = sum w^T r,   linv = h^-1/2,   z = linv g,   c_f = w_f linv
//   u_v = sum_{f sees view v} A_{f,v}^T c_f                                 (6-vector per (landmark, camera pose) view)
//   S(i, j)  -= sum_l u_{l,i} u_{l,j}^T,      rhs(i) -= sum_l u_{l,i} z_l
//   y_l = linv (z - sum_f c_f^T (A_a y_a + A_m y_m)) = linv (z - sum_v u_v . y_cam(v)),   delta_rho = -y_l
// — the scalar case of landmark_kernel / pairs_kernel / backsub_mcc_kernel of k_reproj.hip.  When every binary factor of the window has
// an eliminated landmark (IdpElim::direct) the factors' own pose-pose terms — A_a^T A_a, A_m^T A_m per view, A_a^T A_m per factor,
// the gradient A^T r and diag(A^T A) — are added by the same pair kernel, and the group leaves the generic pose-only assembly (whose
// host-side contribution lists were 50 of the 60 ms of finalize() for a 90 000-factor window); otherwise — some inverse depth constant,
// or shared with another factor — those terms stay with the pose-only groups, the rho slot masked.  Four launches:
//   idp_landmark_kernel   one lane per landmark: h, g, the LM diagonal and Jacobi scale of rho, c of its factors
//   idp_view_kernel       one lane per view: u (and D = sum A^T A, sum A^T r)
//   idp_pairs_kernel      one wave per camera-pose pair segment: 6x6 block sum of -u_a u_b^T (+ the direct terms, the rhs)
//   idp_backsub_kernel    one lane per landmark: the step of rho from the pose step
// Algorithmic bytes: 256 per factor (J 240 + r 16) read by the landmark and view kernels (and by the pair kernel for the factor's cross
// term and for a view of one factor), 64 per view (+ 384 for a view of several factors: the anchor) written once and read once per
// pair entry (diagonal entry); the back-substitution reads the views only.
#include "bsgpu_device.h"

namespace bsg {

__device__ __forceinline__ void idp_landmark_kernel_body(const int bsg_bx, const IdpElim& e, const SmallGroup& g, const double* __restrict__ radius_ptr, double radius_val,
                                                           int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi,
                                                           double* __restrict__ scale, double* __restrict__ dcl, double* __restrict__ grad) {
  const int l = bsg_bx * 128 + threadIdx.x;
  if (l >= e.n_lm) return;
  const int beg = e.lm_start[l], end = e.lm_start[l + 1];
  double h = 0.0, gl = 0.0;
  for (int p = beg; p < end; ++p) {
    const int f = e.order[p];
    const double* J = g.J + (size_t)f * 30;
    const double w0 = J[12], w1 = J[27], r0 = g.r[2 * (size_t)f], r1 = g.r[2 * (size_t)f + 1];
    h += w0 * w0 + w1 * w1;
    gl += w0 * r0 + w1 * r1;
  }
  const int to = e.to0 + l;
  const double inv_radius = 1.0 / (radius_ptr ? radius_ptr[0] : radius_val);
  double sc, d;
  if (compute_scale) { sc = jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0; scale[to] = sc; } else sc = scale[to];
  if (compute_dcl) { d = fmin(fmax(sc * sc * h, lm_lo), lm_hi) / (sc * sc); dcl[to] = d; } else d = dcl[to];
  grad[to] = gl;
  const double linv = 1.0 / sqrt(h + d * inv_radius);
  e.linv[l] = linv; e.z[l] = linv * gl;
  for (int p = beg; p < end; ++p) {
    const double* J = g.J + (size_t)e.order[p] * 30;
    e.C[2 * (size_t)p] = J[12] * linv; e.C[2 * (size_t)p + 1] = J[27] * linv;
  }
}
__global__ __launch_bounds__(128) void idp_landmark_kernel(IdpElim e, SmallGroup g, const double* __restrict__ radius_ptr, double radius_val,
                                                           int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi,
                                                           double* __restrict__ scale, double* __restrict__ dcl, double* __restrict__ grad) {
  idp_landmark_kernel_body((int)blockIdx.x, e, g, radius_ptr, radius_val, compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, grad);
}

// one lane per view: u = sum A^T c over the factors of the landmark that involve the view's camera pose and — when the factors' own
// pose-pose terms are assembled here too (e.direct) — D = sum A^T A (6x6) and the raw gradient sum A^T r.  A factor whose two poses
// are the same camera pose enters with A_a + A_m.
__device__ __forceinline__ void idp_view_kernel_body(const int bsg_bx, const IdpElim& e, const SmallGroup& g) {
  const int v = bsg_bx * 128 + threadIdx.x;
  if (v >= e.n_view) return;
  const int l = e.view_lm[v];
  const int beg = e.lm_start[l], end = e.lm_start[l + 1];
  const double z = e.z[l];
  // (a view of one factor — every measurement view — keeps no D: the pair kernel forms A^T A from the factor's row, 96 bytes instead of 384)
  const int vcode = e.direct ? e.view_code[v] : 0;
  const bool multi = vcode < 0;
  double u[6], gr[6], D[36];
#pragma unroll
  for (int k = 0; k < 6; ++k) { u[k] = 0.0; gr[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < 36; ++k) D[k] = 0.0;
  for (int p = beg; p < end; ++p) {
    const int2 fv = e.fview[p];
    if (fv.x != v && fv.y != v) continue;
    const int f = e.order[p];
    const double* J = g.J + (size_t)f * 30;
    const double ma = fv.x == v ? 1.0 : 0.0, mm = fv.y == v ? 1.0 : 0.0;
    double A0[6], A1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { A0[k] = ma * J[k] + mm * J[6 + k]; A1[k] = ma * J[15 + k] + mm * J[21 + k]; }
    const double c0 = e.C[2 * (size_t)p], c1 = e.C[2 * (size_t)p + 1];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] += A0[k] * c0 + A1[k] * c1;
    if (multi) {
      const double r0 = g.r[2 * (size_t)f], r1 = g.r[2 * (size_t)f + 1];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        gr[a] += A0[a] * r0 + A1[a] * r1;
#pragma unroll
        for (int c = 0; c < 6; ++c) D[a * 6 + c] += A0[a] * A0[c] + A1[a] * A1[c];
      }
    }
  }
  double* uo = e.U + (size_t)v * 8;
#pragma unroll
  for (int k = 0; k < 6; ++k) uo[k] = u[k];
  uo[6] = z; uo[7] = 0.0;
  if (multi) {
    double* o = e.VD + (size_t)(-vcode - 1) * 48;
#pragma unroll
    for (int k = 0; k < 36; ++k) o[k] = D[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[36 + k] = gr[k];
  }
}
__global__ __launch_bounds__(128) void idp_view_kernel(IdpElim e, SmallGroup g) {
  idp_view_kernel_body((int)blockIdx.x, e, g);
}

void launch_idp_landmark(hipStream_t s, const IdpElim& e, const SmallGroup& g, const double* radius_ptr, double radius_val, int compute_scale,
                         int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, double* grad) {
  if (e.n_lm <= 0) return;
  hipLaunchKernelGGL(idp_landmark_kernel, dim3((e.n_lm + 127) / 128), dim3(128), 0, s, e, g, radius_ptr, radius_val, compute_scale, compute_dcl,
                     jacobi, lm_lo, lm_hi, scale, dcl, grad);
  if (e.n_view > 0) hipLaunchKernelGGL(idp_view_kernel, dim3((e.n_view + 127) / 128), dim3(128), 0, s, e, g);
}

// one wave per segment, one lane per entry (view a, view b, code); the 36 + 18 sums leave through one transposed butterfly (as in
// pairs_kernel).  code: -1 = the Schur term only; else (sorted factor position << 2) | (1 = view a is the factor's measurement side) << 1
// | (1 = no Schur term: a further factor on a view pair that already has its entry) — the factor's cross term A_a^T A_m.
__device__ __forceinline__ void idp_pairs_kernel_body(const int bsg_bx, const IdpElim& e, const SmallGroup& g, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                       double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only) {
  const int seg = bsg_bx, lane = threadIdx.x;
  const int ci = e.seg_ci[seg], cj = e.seg_cj[seg];
  const bool diag = ci == cj;
  if (grad_only && !(diag && e.direct)) return;
  const int beg = e.seg_start[seg], end = e.seg_start[seg + 1];
  double v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = 0.0;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  for (int q = beg + lane; q < end; q += 64) {
    const int va = e.ent_va[q], vb = e.ent_vb[q], code = e.ent_code[q];
    if (code < 0 || !(code & 1)) {
      const d2_t* pa = reinterpret_cast<const d2_t*>(e.U + (size_t)va * 8);
      const d2_t* pb = reinterpret_cast<const d2_t*>(e.U + (size_t)vb * 8);
      const d2_t a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3];
      const d2_t b0 = pb[0], b1 = pb[1], b2 = pb[2];
      const double ua[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
      const double ub[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) v[a * 6 + c] -= ua[a] * ub[c];
      if (diag) {
#pragma unroll
        for (int a = 0; a < 6; ++a) v[36 + a] -= ua[a] * a3.x;
      }
    }
    if (e.direct) {
      if (va == vb) {
        const int vc = e.view_code[va];
        if (vc < 0) {
          const double* o = e.VD + (size_t)(-vc - 1) * 48;
#pragma unroll
          for (int k = 0; k < 36; ++k) v[k] += o[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double gk = o[36 + k]; v[36 + k] += gk; v[42 + k] += gk; v[48 + k] += o[7 * k]; }
        } else {   // the view's only factor: (sorted position << 1) | side
          const int f = e.order[vc >> 1], x0 = (vc & 1) ? 6 : 0;
          const double* J = g.J + (size_t)f * 30;
          const double r0 = g.r[2 * (size_t)f], r1 = g.r[2 * (size_t)f + 1];
          double A0[6], A1[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) { A0[k] = J[x0 + k]; A1[k] = J[15 + x0 + k]; }
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            const double gk = A0[a] * r0 + A1[a] * r1;
            v[36 + a] += gk; v[42 + a] += gk; v[48 + a] += A0[a] * A0[a] + A1[a] * A1[a];
#pragma unroll
            for (int c = 0; c < 6; ++c) v[a * 6 + c] += A0[a] * A0[c] + A1[a] * A1[c];
          }
        }
      } else if (code >= 0) {
        const double* J = g.J + (size_t)e.order[code >> 2] * 30;
        const int xa = (code & 2) ? 6 : 0, xb = 6 - xa;   // columns of view a's / view b's pose inside the factor's row
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double p0 = J[xa + a], p1 = J[15 + xa + a];
#pragma unroll
          for (int c = 0; c < 6; ++c) v[a * 6 + c] += p0 * J[xb + c] + p1 * J[15 + xb + c];
        }
      }
    }
  }
  wave_sum_transpose64(v);
  const double total = v[0];
  const int tqi = e.cp_tq[ci], tpi = e.cp_tp[ci], tqj = e.cp_tq[cj], tpj = e.cp_tp[cj];
  if (lane < 36) {
    if (grad_only) return;
    const int a = lane / 6, c = lane % 6;
    const int row = (a < 3) ? (tqi < 0 ? -1 : tqi + a) : (tpi < 0 ? -1 : tpi + a - 3);
    const int col = (c < 3) ? (tqj < 0 ? -1 : tqj + c) : (tpj < 0 ? -1 : tpj + c - 3);
    if (row >= 0 && col >= 0) {
      const int sr = perm[row], sc = perm[col];
      atomicAdd(&S[(size_t)sr * ld + sc], total);
      if (!diag) atomicAdd(&S[(size_t)sc * ld + sr], total);
    }
  } else if (diag && lane < 54) {
    const int a = (lane - 36) % 6, which = (lane - 36) / 6;   // 0: reduced rhs, 1: raw gradient, 2: diag(A^T A)
    const int row = (a < 3) ? (tqi < 0 ? -1 : tqi + a) : (tpi < 0 ? -1 : tpi + a - 3);
    if (row >= 0) {
      if (which == 0) atomicAdd(&S[(size_t)rhs_row * ld + perm[row]], total);
      else if (e.direct && which == 1) atomicAdd(&grad[row], total);
      else if (e.direct) atomicAdd(&hdiag[row], total);
    }
  }
}
__global__ __launch_bounds__(64) void idp_pairs_kernel(IdpElim e, SmallGroup g, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                       double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only) {
  idp_pairs_kernel_body((int)blockIdx.x, e, g, S, ld, rhs_row, grad, hdiag, perm, grad_only);
}

void launch_idp_pairs(hipStream_t s, const IdpElim& e, const SmallGroup& g, double* S, int ld, int rhs_row, double* grad, double* hdiag,
                      const int* perm, bool grad_only) {
  if (e.n_seg <= 0) return;
  hipLaunchKernelGGL(idp_pairs_kernel, dim3(e.n_seg), dim3(64), 0, s, e, g, S, ld, rhs_row, grad, hdiag, perm, grad_only ? 1 : 0);
}

// y_l = linv (z - sum_f c_f^T (A_a y_a + A_m y_m)) = linv (z - sum_v u_v . y_cam(v)): the views carry what the step of rho needs
__device__ __forceinline__ void idp_backsub_kernel_body(const int bsg_bx, const IdpElim& e, const double* __restrict__ y_pose, double* __restrict__ delta) {
  const int l = bsg_bx * 128 + threadIdx.x;
  if (l >= e.n_lm) return;
  double acc = 0.0;
  for (int v = e.view_start[l]; v < e.view_start[l + 1]; ++v) {
    const double* u = e.U + (size_t)v * 8;
    const int cp = e.view_cp[v], tq = e.cp_tq[cp], tp = e.cp_tp[cp];
    if (tq >= 0) acc += u[0] * y_pose[tq] + u[1] * y_pose[tq + 1] + u[2] * y_pose[tq + 2];
    if (tp >= 0) acc += u[3] * y_pose[tp] + u[4] * y_pose[tp + 1] + u[5] * y_pose[tp + 2];
  }
  delta[e.to0 + l] = -(e.linv[l] * (e.z[l] - acc));
}
__global__ __launch_bounds__(128) void idp_backsub_kernel(IdpElim e, const double* __restrict__ y_pose, double* __restrict__ delta) {
  idp_backsub_kernel_body((int)blockIdx.x, e, y_pose, delta);
}

void launch_idp_backsub(hipStream_t s, const IdpElim& e, const double* y_pose, double* delta) {
  if (e.n_lm <= 0) return;
  hipLaunchKernelGGL(idp_backsub_kernel, dim3((e.n_lm + 127) / 128), dim3(128), 0, s, e, y_pose, delta);
}

// ---- the same launches over several windows (bsgpu_batch.cpp): entry w = what window w's lone launches pass (zero grids: no inverse-depth
// landmarks); the radius and the step's flags come per round (BatchDyn)
struct idp_landmark_Args { int bsg_grid; int view_grid; IdpElim e; SmallGroup g; int jacobi; double lm_lo, lm_hi; double* scale; double* dcl; double* grad; };
__global__ __launch_bounds__(128) void idp_landmark_kernel_batch(const idp_landmark_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const idp_landmark_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  idp_landmark_kernel_body((int)blockIdx.x, a.e, a.g, nullptr, bsg_dyn->radius[bsg_w], bsg_dyn->first[bsg_w], bsg_dyn->new_J[bsg_w], a.jacobi, a.lm_lo, a.lm_hi, a.scale, a.dcl, a.grad);
}
__global__ __launch_bounds__(128) void idp_view_kernel_batch(const idp_landmark_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const idp_landmark_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.view_grid) return;
  idp_view_kernel_body((int)blockIdx.x, a.e, a.g);
}
void batchargs_idp_landmark(BatchArgTable& t, BatchArgTable& t_view, const IdpElim& e, const SmallGroup& g, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, double* grad) {
  idp_landmark_Args a;
  a.e = e; a.g = g; a.jacobi = jacobi; a.lm_lo = lm_lo; a.lm_hi = lm_hi; a.scale = scale; a.dcl = dcl; a.grad = grad;
  a.bsg_grid = e.n_lm > 0 ? (e.n_lm + 127) / 128 : 0;
  a.view_grid = (e.n_lm > 0 && e.n_view > 0) ? (e.n_view + 127) / 128 : 0;
  t.push(a);
  idp_landmark_Args b = a;   // (the view launch reads the same entry: its own table only carries its grid for the launch's size)
  b.bsg_grid = a.view_grid;
  t_view.push(b);
}
void launch_idp_landmark_batch(hipStream_t s, const BatchArgTable& t, const BatchArgTable& t_view, const BatchDyn* dyn, int list, int n) {
  if (n <= 0) return;
  if (t.max_grid > 0) hipLaunchKernelGGL(idp_landmark_kernel_batch, dim3(t.max_grid, n), dim3(128), 0, s, static_cast<const idp_landmark_Args*>(t.dev), dyn, list);
  if (t_view.max_grid > 0) hipLaunchKernelGGL(idp_view_kernel_batch, dim3(t_view.max_grid, n), dim3(128), 0, s, static_cast<const idp_landmark_Args*>(t_view.dev), dyn, list);
}
struct idp_pairs_Args { int bsg_grid; IdpElim e; SmallGroup g; double* S; int ld; int rhs_row; double* grad; double* hdiag; const int* perm; };
__global__ __launch_bounds__(64) void idp_pairs_kernel_batch(const idp_pairs_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const idp_pairs_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  idp_pairs_kernel_body((int)blockIdx.x, a.e, a.g, a.S, a.ld, a.rhs_row, a.grad, a.hdiag, a.perm, bsg_dyn->grad_only[bsg_w]);
}
void batchargs_idp_pairs(BatchArgTable& t, const IdpElim& e, const SmallGroup& g, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm) {
  idp_pairs_Args a;
  a.e = e; a.g = g; a.S = S; a.ld = ld; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag; a.perm = perm; a.bsg_grid = e.n_seg > 0 ? e.n_seg : 0;
  t.push(a);
}
void launch_idp_pairs_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(idp_pairs_kernel_batch, dim3(t.max_grid, n), dim3(64), 0, s, static_cast<const idp_pairs_Args*>(t.dev), dyn, list);
}
struct idp_backsub_Args { int bsg_grid; IdpElim e; const double* y_pose; double* delta; };
__global__ __launch_bounds__(128) void idp_backsub_kernel_batch(const idp_backsub_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const idp_backsub_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  idp_backsub_kernel_body((int)blockIdx.x, a.e, a.y_pose, a.delta);
}
void batchargs_idp_backsub(BatchArgTable& t, const IdpElim& e, const double* y_pose, double* delta) {
  idp_backsub_Args a;
  a.e = e; a.y_pose = y_pose; a.delta = delta; a.bsg_grid = e.n_lm > 0 ? (e.n_lm + 127) / 128 : 0;
  t.push(a);
}
void launch_idp_backsub_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(idp_backsub_kernel_batch, dim3(t.max_grid, n), dim3(128), 0, s, static_cast<const idp_backsub_Args*>(t.dev), dyn, list);
}

}  // namespace bsg
