// Pose-only factors of the solve path on gfx950 (residual + analytic tangent Jacobian), their
// J^T J / J^T r assembly into the dense reduced system, and their model-cost-change term.
//   IMU_DELTA   bs_constraints/.../inertial/normal_delta_imu_state_3d_cost_functor.h:59-141   (wave / factor)
//   IMU_PRIOR   bs_constraints/.../inertial/normal_prior_imu_state_3d_cost_functor.h:57-88   (wave / factor)
//   RELPOSE(_EXT) bs_constraints/.../relative_pose/delta_pose_3d_with_extrinsics_cost_functor.h:65-109
//               + [EXT] fuse NormalDeltaPose3DCostFunctor                                     (lane / factor)
//   ABSPOSE     [EXT] fuse NormalPriorPose3DCostFunctor (global/absolute_pose_3d_constraint.cpp:45-50)
//   ABS/REL_VEC3 [EXT] fuse Absolute/RelativeConstraint<V> (global/absolute_constraint.h:10-25)
//   GRAVITY     bs_constraints/.../global/gravity_alignment_cost_functor.h:50-63
// The reference differentiates these with ceres::AutoDiffCostFunction and multiplies by the
// quaternion PlusJacobian; here the tangent Jacobians are closed form,
// checked against the Jet-based oracle in tests/.
#include "bsgpu_device.h"
#include "reproj_body.h"
#include "marg_body.h"

namespace bsg {

__constant__ double kGravity[3] = {0.0, 0.0, -9.80665};  // bs_common/include/bs_common/utils.h:20-24

// Eigen: q.conjugate() * v   (v + w uv + u x uv with uv = 2 u x v, u = -q.vec)
BSG_DEV void eigen_conj_rotate(const double q[4], const double v[3], double o[3]) {
  const double u[3] = {-q[1], -q[2], -q[3]};
  double uv[3];
  cross3(u, v, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  double c[3];
  cross3(u, uv, c);
  o[0] = v[0] + q[0] * uv[0] + c[0];
  o[1] = v[1] + q[0] * uv[1] + c[1];
  o[2] = v[2] + q[0] * uv[2] + c[2];
}

// the cost of a 128-factor block of a lane-per-factor kernel as ONE partial (the relative-pose and inverse-depth groups have tens of
// thousands of factors: a per-factor array kept the end-of-step reduction busy for 9 - 16 us); every thread of the block must call it
BSG_DEV void block_cost_128(double cost, double* __restrict__ cost_part, int block) {
  __shared__ double s_cost[2];
  const double w = wave_sum(cost);
  if ((threadIdx.x & 63) == 0) s_cost[(threadIdx.x >> 6) & 1] = w;
  __syncthreads();
  if ((threadIdx.x & 127) == 0) cost_part[block] = s_cost[0] + s_cost[1];
}
BSG_DEV void finish_small(const SmallGroup& g, int f, const DevLoss* losses, double s, double* sc, double* cost) {
  double rho1;
  const double rho = loss_eval(losses[g.loss[f]], s, &rho1);
  *sc = sqrt(rho1);
  *cost = g.active[f] ? 0.5 * rho : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// IMU delta: one wave per factor.  Lanes 0..14 own a residual row, lanes 0..29 own a Jacobian column.
// ---------------------------------------------------------------------------------------------------
// NW: waves of the calling workgroup (each evaluates a factor of its own: its wave's slice of the LDS stage the caller lends)
constexpr int kImuBPitch = 33;
template <bool WITH_J, int NW> constexpr int kImuStage = WITH_J ? NW * 16 * kImuBPitch : 1;
template <bool WITH_J, int NW = 1>
__device__ __forceinline__ void imu_delta_body(const SmallGroup g, const int f, const double* __restrict__ x,
                                               const DevLoss* __restrict__ losses, double* __restrict__ cost_part, const int lane,
                                               double* s_imuB /* kImuStage<WITH_J, NW> doubles of LDS */) {
  // (round 5) J = sc A Jraw (15 x 15 by 15 x 30) on the matrix core: a lane per COLUMN doing the 225 FMAs of its column with 225 loads of A
  // (the same for every lane) was half of the unit's 2 325 instructions, and the unit — one wave — is what a window of the reference's
  // size waits for in its evaluation launch.  The columns go through LDS as the B operand (16 x 32, zero padding); every lane loads
  // the four entries of A its lane position multiplies.
  constexpr int kBP = kImuBPitch;
  const int* xo = g.xoff + (size_t)f * 10;
  const int* to = g.toff + (size_t)f * 10;
  const double* c = g.consts + (size_t)f * 287;
  const double* A = c + 62;
  double qi[4], qj[4], pi[3], vi[3], bgi[3], bai[3], pj[3], vj[3], bgj[3], baj[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) { qi[k] = x[xo[0] + k]; qj[k] = x[xo[5] + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pi[k] = x[xo[1] + k]; vi[k] = x[xo[2] + k]; bgi[k] = x[xo[3] + k]; bai[k] = x[xo[4] + k];
    pj[k] = x[xo[6] + k]; vj[k] = x[xo[7] + k]; bgj[k] = x[xo[8] + k]; baj[k] = x[xo[9] + k];
  }
  const double dt = c[0];
  const double dq[4] = {c[1], c[2], c[3], c[4]};
  const double* dp = c + 5;
  const double* dv = c + 8;
  const double* dq_dbg = c + 11;
  const double* dp_dbg = c + 20;
  const double* dp_dba = c + 29;
  const double* dv_dbg = c + 38;
  const double* dv_dba = c + 47;
  double dbg[3], dba[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { dbg[k] = bgi[k] - c[56 + k]; dba[k] = bai[k] - c[59 + k]; }
  double tau[3];
  mat3_vec(dq_dbg, dbg, tau);
  const double dd[4] = {1.0, tau[0] / 2.0, tau[1] / 2.0, tau[2] / 2.0};  // DeltaQ: NOT normalised (utils.h:28-38)
  double qc[4];
  quat_mul(dq, dd, qc);
  const double nc = qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3];
  const double u[4] = {qc[0] / nc, -qc[1] / nc, -qc[2] / nc, -qc[3] / nc};   // q_corrected.inverse()
  const double ni = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
  const double qi_inv[4] = {qi[0] / ni, -qi[1] / ni, -qi[2] / ni, -qi[3] / ni};
  double e[4], m[4];
  quat_mul(qi_inv, qj, e);
  quat_mul(u, e, m);
  double res[15];
  res[0] = 2.0 * m[1]; res[1] = 2.0 * m[2]; res[2] = 2.0 * m[3];
  double ap[3], av[3], rap[3], rav[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ap[k] = pj[k] - pi[k] - dt * vi[k] - 0.5 * dt * dt * kGravity[k];
    av[k] = vj[k] - vi[k] - dt * kGravity[k];
  }
  eigen_conj_rotate(qi, ap, rap);
  eigen_conj_rotate(qi, av, rav);
  double t1[3], t2[3];
  mat3_vec(dp_dbg, dbg, t1); mat3_vec(dp_dba, dba, t2);
#pragma unroll
  for (int k = 0; k < 3; ++k) res[3 + k] = rap[k] - (dp[k] + t1[k] + t2[k]);
  mat3_vec(dv_dbg, dbg, t1); mat3_vec(dv_dba, dba, t2);
#pragma unroll
  for (int k = 0; k < 3; ++k) res[6 + k] = rav[k] - (dv[k] + t1[k] + t2[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { res[9 + k] = bgj[k] - bgi[k]; res[12 + k] = baj[k] - bai[k]; }
  // r = A res: lane k < 15 owns row k
  double rk = 0.0;
  if (lane < 15) {
#pragma unroll
    for (int k = 0; k < 15; ++k) rk += A[15 * lane + k] * res[k];
  }
  const double s = wave_sum(rk * rk);
  double sc, cost;
  finish_small(g, f, losses, s, &sc, &cost);
  if (WITH_J && lane < 15) g.r[(size_t)f * 15 + lane] = rk * sc;
  if (lane == 0) cost_part[f] = cost;
  if (!WITH_J) return;
  const int mn = lane & 15, mq = lane >> 4;
  double a_op[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { const int k = 4 * kk + mq; a_op[kk] = (mn < 15 && k < 15) ? A[15 * mn + k] : 0.0; }
  // raw Jacobian column `lane` (< 30): block b, component i
  const int b = lane / 3, i = lane % 3;
  double col[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) col[k] = 0.0;
  double Ri[9];
  quat_to_rot(qi, Ri);
  const double ei[3] = {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0};
  const double RiT_ei[3] = {Ri[3 * i], Ri[3 * i + 1], Ri[3 * i + 2]};  // R_i^T e_i = row i of R_i
  if (lane < 30) switch (b) {
    case 0: {  // theta_i
      const double pe[4] = {0.0, ei[0], ei[1], ei[2]};
      double t[4], w[4];
      quat_mul(u, pe, t);
      quat_mul(t, e, w);
      col[0] = -w[1]; col[1] = -w[2]; col[2] = -w[3];
      double Rap[3], Rav[3];
      mat3t_vec(Ri, ap, Rap); mat3t_vec(Ri, av, Rav);
      double c1[3], c2[3];
      cross3(Rap, ei, c1); cross3(Rav, ei, c2);
      col[3] = c1[0]; col[4] = c1[1]; col[5] = c1[2];
      col[6] = c2[0]; col[7] = c2[1]; col[8] = c2[2];
    } break;
    case 1:  // p_i
      col[3] = -RiT_ei[0]; col[4] = -RiT_ei[1]; col[5] = -RiT_ei[2];
      break;
    case 2:  // v_i
      col[3] = -dt * RiT_ei[0]; col[4] = -dt * RiT_ei[1]; col[5] = -dt * RiT_ei[2];
      col[6] = -RiT_ei[0]; col[7] = -RiT_ei[1]; col[8] = -RiT_ei[2];
      break;
    case 3: {  // bg_i
      // d res_q / d tau_k = 2 vec( du/dtau_k (x) e ),  du/dtau_k = (0,-e_k/2)(x)conj(dq)/nc - u (|dq|^2 tau_k/2)/nc
      const double dqc[4] = {dq[0], -dq[1], -dq[2], -dq[3]};
      const double ndq = dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3];
      const double inv_nc = 1.0 / nc;   // (one reciprocal for the fifteen quotients by nc below)
      double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double pk[4] = {0.0, k == 0 ? -0.5 : 0.0, k == 1 ? -0.5 : 0.0, k == 2 ? -0.5 : 0.0};
        double du[4], w[4];
        quat_mul(pk, dqc, du);
        const double fk = ndq * tau[k] * 0.5 * inv_nc;
#pragma unroll
        for (int a = 0; a < 4; ++a) du[a] = du[a] * inv_nc - u[a] * fk;
        quat_mul(du, e, w);
        const double jk = dq_dbg[3 * k + i];
        acc[0] += 2.0 * w[1] * jk; acc[1] += 2.0 * w[2] * jk; acc[2] += 2.0 * w[3] * jk;
      }
      col[0] = acc[0]; col[1] = acc[1]; col[2] = acc[2];
      col[3] = -dp_dbg[i]; col[4] = -dp_dbg[3 + i]; col[5] = -dp_dbg[6 + i];
      col[6] = -dv_dbg[i]; col[7] = -dv_dbg[3 + i]; col[8] = -dv_dbg[6 + i];
      col[9 + i] = -1.0;
    } break;
    case 4:  // ba_i
      col[3] = -dp_dba[i]; col[4] = -dp_dba[3 + i]; col[5] = -dp_dba[6 + i];
      col[6] = -dv_dba[i]; col[7] = -dv_dba[3 + i]; col[8] = -dv_dba[6 + i];
      col[12 + i] = -1.0;
      break;
    case 5: {  // theta_j : m_w e_i + m_v x e_i
      const double mv[3] = {m[1], m[2], m[3]};
      double cx[3];
      cross3(mv, ei, cx);
      col[0] = m[0] * ei[0] + cx[0]; col[1] = m[0] * ei[1] + cx[1]; col[2] = m[0] * ei[2] + cx[2];
    } break;
    case 6: col[3] = RiT_ei[0]; col[4] = RiT_ei[1]; col[5] = RiT_ei[2]; break;  // p_j
    case 7: col[6] = RiT_ei[0]; col[7] = RiT_ei[1]; col[8] = RiT_ei[2]; break;  // v_j
    case 8: col[9 + i] = 1.0; break;                                              // bg_j
    default: col[12 + i] = 1.0; break;                                            // ba_j
  }
  double* sB = s_imuB + ((threadIdx.x >> 6) % NW) * (16 * kBP);
  if (lane < 32) {   // (lanes 30, 31: zero columns; row 15: zero)
#pragma unroll
    for (int mm = 0; mm < 15; ++mm) sB[mm * kBP + lane] = col[mm];
    sB[15 * kBP + lane] = 0.0;
  }
  __builtin_amdgcn_wave_barrier();
  typedef double imu_d4 __attribute__((ext_vector_type(4)));
  imu_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double b0 = sB[(4 * kk + mq) * kBP + mn], b1 = sB[(4 * kk + mq) * kBP + 16 + mn];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[kk], b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[kk], b1, acc1, 0, 0, 0);
  }
  // the lane holds rows mq + 4 reg of columns mn and 16 + mn
  const double s0 = to[mn / 3] < 0 ? 0.0 : sc, s1 = (mn < 14 && to[(16 + mn) / 3] >= 0) ? sc : 0.0;
  double* Jo = g.J + (size_t)f * 450;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = mq + 4 * reg;
    if (row < 15) {
      Jo[row * 30 + mn] = acc0[reg] * s0;
      if (mn < 14) Jo[row * 30 + 16 + mn] = acc1[reg] * s1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// IMU prior: one wave per factor; 15 rows / 15 columns
// ---------------------------------------------------------------------------------------------------
template <bool WITH_J>
__device__ __forceinline__ void imu_prior_body(const SmallGroup g, const int f, const double* __restrict__ x,
                                               const DevLoss* __restrict__ losses, double* __restrict__ cost_part, const int lane) {
  const int* xo = g.xoff + (size_t)f * 5;
  const int* to = g.toff + (size_t)f * 5;
  const double* c = g.consts + (size_t)f * 241;
  const double* A = c + 16;
  const double q[4] = {x[xo[0]], x[xo[0] + 1], x[xo[0] + 2], x[xo[0] + 3]};
  const double binv[4] = {c[0], -c[1], -c[2], -c[3]};
  double diff[4], res[15];
  quat_mul(binv, q, diff);
  quat_to_angle_axis(diff, res);
#pragma unroll
  for (int s = 1; s < 5; ++s)
#pragma unroll
    for (int k = 0; k < 3; ++k) res[3 * s + k] = x[xo[s] + k] - c[1 + 3 * s + k];
  double rk = 0.0;
  if (lane < 15) {
#pragma unroll
    for (int k = 0; k < 15; ++k) rk += A[15 * lane + k] * res[k];
  }
  const double s2 = wave_sum(rk * rk);
  double sc, cost;
  finish_small(g, f, losses, s2, &sc, &cost);
  if (WITH_J && lane < 15) g.r[(size_t)f * 15 + lane] = rk * sc;
  if (lane == 0) cost_part[f] = cost;
  if (!WITH_J || lane >= 15) return;
  double Jr[9];
  so3_jr_inv(res, Jr);
  const bool is_const = to[lane / 3] < 0;
  double* Jo = g.J + (size_t)f * 225;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    double a;
    if (lane < 3) a = A[15 * k] * Jr[lane] + A[15 * k + 1] * Jr[3 + lane] + A[15 * k + 2] * Jr[6 + lane];
    else a = A[15 * k + lane];
    Jo[k * 15 + lane] = is_const ? 0.0 : a * sc;
  }
}

template <bool WITH_J>
__global__ __launch_bounds__(64) void imu_delta_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
                                                       double* __restrict__ cost_part) {
  __shared__ double sB[kImuStage<WITH_J, 1>];
  imu_delta_body<WITH_J, 1>(g, blockIdx.x, x, losses, cost_part, threadIdx.x, sB);
}
template <bool WITH_J>
__global__ __launch_bounds__(64) void imu_prior_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
                                                       double* __restrict__ cost_part) {
  imu_prior_body<WITH_J>(g, blockIdx.x, x, losses, cost_part, threadIdx.x);
}
// both IMU factor types of a visual-inertial window (n-1 pre-integrated factors, one or two priors) in ONE launch: a launch
// of its own for the single prior costs more in dispatch than in work (see the Makefile note on this file's flags)
template <bool WITH_J>
__global__ __launch_bounds__(64) void imu_eval_kernel(SmallGroup delta, SmallGroup prior, const double* __restrict__ x,
                                                      const DevLoss* __restrict__ losses, double* __restrict__ part_delta,
                                                      double* __restrict__ part_prior) {
  __shared__ double sB[kImuStage<WITH_J, 1>];
  if ((int)blockIdx.x < delta.n) imu_delta_body<WITH_J, 1>(delta, blockIdx.x, x, losses, part_delta, threadIdx.x, sB);
  else imu_prior_body<WITH_J>(prior, blockIdx.x - delta.n, x, losses, part_prior, threadIdx.x);
}
// ... and, in a window that also has reprojection factors, both of them as the first workgroups of the reprojection evaluation (four
// factors, a wave each, per 256-thread workgroup): 5 us (cost only) / 10 us (with Jacobians) of a launch that nothing but the launch
// order made wait for the reprojection factors.
template <bool WITH_J>
__device__ __forceinline__ void visual_imu_eval_kernel_body(const int bsg_bx, const int bsg_gx, SmallGroup delta, SmallGroup prior, double* __restrict__ part_delta, double* __restrict__ part_prior, int n_imu_blocks, int n, const int4* __restrict__ fac, const double2* __restrict__ pix, const double* __restrict__ wgt, const double* __restrict__ x, const DevCamera* __restrict__ cams, const DevLoss* __restrict__ losses, double2* __restrict__ r_out, double* __restrict__ J_out, double* __restrict__ JB_out, double* __restrict__ cost_part, int count_inactive) {
  static_assert(kReprojStage<WITH_J> >= kImuStage<WITH_J, 4>, "the IMU units borrow the reprojection factors' staging area");
  __shared__ __attribute__((aligned(16))) double sJ[kReprojStage<WITH_J>];
  if (bsg_bx < n_imu_blocks) {
    // (the factor index is the same in every lane of the wave: said so, its tables, constants and values are fetched with scalar loads)
    const int f = __builtin_amdgcn_readfirstlane(4 * bsg_bx + ((int)threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (f < delta.n) imu_delta_body<WITH_J, 4>(delta, f, x, losses, part_delta, lane, sJ);
    else if (f < delta.n + prior.n) imu_prior_body<WITH_J>(prior, f - delta.n, x, losses, part_prior, lane);
    return;
  }
  reproj_eval_body<WITH_J>(bsg_bx - n_imu_blocks, n, fac, pix, wgt, x, cams, losses, r_out, J_out, JB_out, cost_part, count_inactive, sJ);
}
template <bool WITH_J>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void visual_imu_eval_kernel(SmallGroup delta, SmallGroup prior, double* __restrict__ part_delta, double* __restrict__ part_prior, int n_imu_blocks, int n, const int4* __restrict__ fac, const double2* __restrict__ pix, const double* __restrict__ wgt, const double* __restrict__ x, const DevCamera* __restrict__ cams, const DevLoss* __restrict__ losses, double2* __restrict__ r_out, double* __restrict__ J_out, double* __restrict__ JB_out, double* __restrict__ cost_part, int count_inactive) {
  visual_imu_eval_kernel_body<WITH_J>((int)blockIdx.x, (int)gridDim.x, delta, prior, part_delta, part_prior, n_imu_blocks, n, fac, pix, wgt, x, cams, losses, r_out, J_out, JB_out, cost_part, count_inactive);
}
// The evaluation launched AHEAD of the accept / reject decision (residuals and Jacobians at the candidate, underneath the host's round trip)
// with the end-of-step reduction of the step just computed as its first workgroups: the host's stamp leaves ~4 us into this launch
// instead of after a launch of its own (7.7 us + the 4.5 us a kernel that wrote host memory takes to retire, on every LM step).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void visual_imu_eval_reduce_kernel(ReduceRide red, SmallGroup delta, SmallGroup prior, double* __restrict__ part_delta, double* __restrict__ part_prior, int n_imu_blocks, int n, const int4* __restrict__ fac, const double2* __restrict__ pix, const double* __restrict__ wgt, const double* __restrict__ x, const DevCamera* __restrict__ cams, const DevLoss* __restrict__ losses, double2* __restrict__ r_out, double* __restrict__ J_out, double* __restrict__ JB_out, double* __restrict__ cost_part, int count_inactive) {
  const int n_units = red.n_slots + 1;
  if ((int)blockIdx.x < n_units) {
    __shared__ double sred[16];
    final_reduce_unit<256>((int)blockIdx.x, (int)threadIdx.x, red, n_units, sred);
    return;
  }
  visual_imu_eval_kernel_body<true>((int)blockIdx.x - n_units, (int)gridDim.x - n_units, delta, prior, part_delta, part_prior, n_imu_blocks, n, fac, pix, wgt, x, cams, losses, r_out, J_out, JB_out, cost_part, count_inactive);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct visual_imu_eval_kernel_Args {
  int bsg_grid;
  SmallGroup delta;
  SmallGroup prior;
  double* part_delta;
  double* part_prior;
  int n_imu_blocks;
  int n;
  const int4* fac;
  const double2* pix;
  const double* wgt;
  const double* x;
  const DevCamera* cams;
  const DevLoss* losses;
  double2* r_out;
  double* J_out;
  double* JB_out;
  double* cost_part;
  int count_inactive;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct visual_imu_eval_kernel_ArgsG {
  int bsg_grid;
  SmallGroup delta;
  SmallGroup prior;
  double __attribute__((address_space(1)))* part_delta;
  double __attribute__((address_space(1)))* part_prior;
  int n_imu_blocks;
  int n;
  const int4 __attribute__((address_space(1)))* fac;
  const double2 __attribute__((address_space(1)))* pix;
  const double __attribute__((address_space(1)))* wgt;
  const double __attribute__((address_space(1)))* x;
  const DevCamera __attribute__((address_space(1)))* cams;
  const DevLoss __attribute__((address_space(1)))* losses;
  double2 __attribute__((address_space(1)))* r_out;
  double __attribute__((address_space(1)))* J_out;
  double __attribute__((address_space(1)))* JB_out;
  double __attribute__((address_space(1)))* cost_part;
  int count_inactive;
};
static_assert(sizeof(visual_imu_eval_kernel_ArgsG) == sizeof(visual_imu_eval_kernel_Args), "layout");

template <bool WITH_J>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void visual_imu_eval_kernel_batch(const visual_imu_eval_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const visual_imu_eval_kernel_ArgsG& a = reinterpret_cast<const visual_imu_eval_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  visual_imu_eval_kernel_body<WITH_J>((int)blockIdx.x, a.bsg_grid, a.delta, a.prior, (double*)a.part_delta, (double*)a.part_prior, a.n_imu_blocks, a.n, (const int4*)a.fac, (const double2*)a.pix, (const double*)a.wgt, (const double*)a.x, (const DevCamera*)a.cams, (const DevLoss*)a.losses, (double2*)a.r_out, (double*)a.J_out, (double*)a.JB_out, (double*)a.cost_part, a.count_inactive);
}
void launch_visual_imu_eval(hipStream_t s, const Visual& v, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevCamera* cams,
                            const DevLoss* losses, bool with_J, double* cost_part_vis, double* part_delta, double* part_prior, const ReduceRide* red) {
  const int n_imu_blocks = (delta.n + prior.n + 3) / 4, grid = n_imu_blocks + (v.n + 255) / 256;
  if (with_J && red && red->n_entries > 0)
    hipLaunchKernelGGL(visual_imu_eval_reduce_kernel, dim3(red->n_slots + 1 + grid), dim3(256), 0, s, *red, delta, prior, part_delta, part_prior, n_imu_blocks, v.n,
                       v.fac, v.pix, v.w, x, cams, losses, v.r, v.J, v.JB, cost_part_vis, 0);
  else if (with_J)
    hipLaunchKernelGGL(visual_imu_eval_kernel<true>, dim3(grid), dim3(256), 0, s, delta, prior, part_delta, part_prior, n_imu_blocks, v.n, v.fac, v.pix, v.w,
                       x, cams, losses, v.r, v.J, v.JB, cost_part_vis, 0);
  else
    hipLaunchKernelGGL(visual_imu_eval_kernel<false>, dim3(grid), dim3(256), 0, s, delta, prior, part_delta, part_prior, n_imu_blocks, v.n, v.fac, v.pix, v.w,
                       x, cams, losses, v.r, v.J, v.JB, cost_part_vis, 0);
}
void batchargs_visual_imu_eval(BatchArgTable& t, const Visual& v, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevCamera* cams,
                               const DevLoss* losses, double* cost_part_vis, double* part_delta, double* part_prior) {
  visual_imu_eval_kernel_Args a;
  a.n_imu_blocks = (delta.n + prior.n + 3) / 4; a.bsg_grid = a.n_imu_blocks + (v.n + 255) / 256;
  a.delta = delta; a.prior = prior; a.part_delta = part_delta; a.part_prior = part_prior; a.n = v.n; a.fac = v.fac; a.pix = v.pix; a.wgt = v.w;
  a.x = x; a.cams = cams; a.losses = losses; a.r_out = v.r; a.J_out = v.J; a.JB_out = v.JB; a.cost_part = cost_part_vis; a.count_inactive = 0;
  t.push(a);
}
void launch_visual_imu_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J) {
  if (n <= 0 || t.max_grid <= 0) return;
  const auto* A = static_cast<const visual_imu_eval_kernel_Args*>(t.dev);
  if (with_J) hipLaunchKernelGGL(visual_imu_eval_kernel_batch<true>, dim3(t.max_grid, n), dim3(256), 0, s, A, dyn, list);
  else hipLaunchKernelGGL(visual_imu_eval_kernel_batch<false>, dim3(t.max_grid, n), dim3(256), 0, s, A, dyn, list);
}
void launch_imu_eval(hipStream_t s, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevLoss* losses, bool with_J,
                     double* part_delta, double* part_prior) {
  if (with_J) hipLaunchKernelGGL(imu_eval_kernel<true>, dim3(delta.n + prior.n), dim3(64), 0, s, delta, prior, x, losses, part_delta, part_prior);
  else hipLaunchKernelGGL(imu_eval_kernel<false>, dim3(delta.n + prior.n), dim3(64), 0, s, delta, prior, x, losses, part_delta, part_prior);
}

// ---------------------------------------------------------------------------------------------------
// relative pose (with / without extrinsics): one lane per factor
// ---------------------------------------------------------------------------------------------------
// doubles of LDS a 128-thread workgroup of the pose-only evaluation lends its body: two rows of the 64 factors of each of its two waves
// (relative-pose factors with extrinsics: 18 columns) or the B operands of two IMU units
template <bool WITH_J> constexpr int kSmallStage = WITH_J ? 2 * 64 * 2 * 18 : 2;
template <bool EXT, bool WITH_J>
__device__ __forceinline__ void relpose_body(const SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
                                             double* __restrict__ cost_part, const int block, double* sJ /* kSmallStage<WITH_J> doubles of LDS, 16-byte aligned */) {
  static_assert(kSmallStage<WITH_J> >= kImuStage<WITH_J, 2>, "stage");
  constexpr int NV = EXT ? 6 : 4;
  constexpr int TW = 3 * NV;
  // (Jacobian rows leave through LDS: a lane per factor storing its 864-byte Jacobian 8 bytes at a time touches 64 cache lines per
  // store instruction — 44 us for C3's 20 000 factors; see the end of the kernel)
  const int f_raw = block * 128 + threadIdx.x;
  const bool live = f_raw < g.n;
  const int f = live ? f_raw : g.n - 1;   // (idle lanes of the last workgroup redo the last factor and store nothing)
  const int* xo = g.xoff + (size_t)f * NV;
  const int* to = g.toff + (size_t)f * NV;
  const double* c = g.consts + (size_t)f * 43;
  const double* A = c + 7;
  double p1[3], q1[4], p2[3], q2[4];
#pragma unroll
  for (int k = 0; k < 3; ++k) { p1[k] = x[xo[0] + k]; p2[k] = x[xo[2] + k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { q1[k] = x[xo[1] + k]; q2[k] = x[xo[3] + k]; }
  double Rb1[9], Rb2[9], Re[9], pe[3] = {0, 0, 0};
  double ps1[3], qs1[4], ps2[3], qs2[4];
  if (EXT) {
    double qe[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) pe[k] = x[xo[4] + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) qe[k] = x[xo[5] + k];
    quat_to_rot_normalized(q1, Rb1);
    quat_to_rot_normalized(q2, Rb2);
    quat_to_rot_normalized(qe, Re);
    quat_mul(q1, qe, qs1);
    quat_mul(q2, qe, qs2);
    double t[3];
    mat3_vec(Rb1, pe, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) ps1[k] = t[k] + p1[k];
    mat3_vec(Rb2, pe, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) ps2[k] = t[k] + p2[k];
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) { ps1[k] = p1[k]; ps2[k] = p2[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { qs1[k] = q1[k]; qs2[k] = q2[k]; }
  }
  double R1[9], R2[9];
  quat_to_rot_normalized(qs1, R1);
  quat_to_rot_normalized(qs2, R2);
  const double dpw[3] = {ps2[0] - ps1[0], ps2[1] - ps1[1], ps2[2] - ps1[2]};
  double dpr[3];
  mat3t_vec(R1, dpw, dpr);
  double e[6];
  e[0] = dpr[0] - c[0]; e[1] = dpr[1] - c[1]; e[2] = dpr[2] - c[2];
  const double q1inv[4] = {qs1[0], -qs1[1], -qs1[2], -qs1[3]};
  const double dinv[4] = {c[3], -c[4], -c[5], -c[6]};
  double diff[4], err[4];
  quat_mul(q1inv, qs2, diff);
  quat_mul(dinv, diff, err);
  quat_to_angle_axis(err, e + 3);
  double r[6], s = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) a += A[6 * i + k] * e[k];
    r[i] = a; s += a * a;
  }
  double sc, cost;
  finish_small(g, f, losses, s, &sc, &cost);
  block_cost_128(live ? cost : 0.0, cost_part, block);
  if (!WITH_J) return;
  if (live) {
#pragma unroll
    for (int i = 0; i < 6; ++i) g.r[(size_t)f * 6 + i] = r[i] * sc;
  }
  // derivatives w.r.t. the SENSOR poses: columns (p_s1, th_s1, p_s2, th_s2)
  double Jr[9];
  so3_jr_inv(e + 3, Jr);
  // G = -Jr R2^T R1
  double R2tR1[9], G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R2tR1[3 * i + j] = R2[i] * R1[j] + R2[3 + i] * R1[3 + j] + R2[6 + i] * R1[6 + j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[3 * i + j] = -(Jr[3 * i] * R2tR1[j] + Jr[3 * i + 1] * R2tR1[3 + j] + Jr[3 * i + 2] * R2tR1[6 + j]);
  // J = sc A Je with A = [A_p | A_q] (6 x 3 each) and the raw Jacobian Je of e w.r.t. the sensor poses — e_p rows: d/dp_s1 = -R1^T,
  // d/dth_s1 = [dpr]x, d/dp_s2 = R1^T; e_q rows: d/dth_s1 = G, d/dth_s2 = Jr — chained to the base-frame / extrinsics columns.  Written out
  // by 3-column blocks (round 5; the generic 6 x 6 by 6 x TW product spent two thirds of its FMAs on structural zeros, which -fno-fast-math
  // keeps, and the kernel is bound by the instruction count of ONE wave: 157 workgroups on 256 compute units).  Row i of
  //   P1 = -A_p R1^T,   M12 = A_p [dpr]x + A_q G,   M3 = A_q Jr            (a^T [p]x = (a x p)^T)
  // gives row i of every block:
  //   no extrinsics   p1: P1   th1: M12   p2: -P1   th2: M3
  //   extrinsics      p_b1: P1   p_b2: -P1   p_e: P1 (Rb1 - Rb2)   th_e: M12 + M3
  //                   th_b1: M12 Re^T + P1 X1,  X1 row k = pe x Rb1_k  (= -Rb1 [pe]x)      th_b2: M3 Re^T + P1 X2,  X2 row k = Rb2_k x pe
  double Dr[9], X1[9], X2[9];
  if (EXT) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cross3(pe, Rb1 + 3 * k, X1 + 3 * k);
      cross3(Rb2 + 3 * k, pe, X2 + 3 * k);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) Dr[k] = Rb1[k] - Rb2[k];
  }
  // a constant block's columns are zero: its scale is (the compare is per block, not per entry)
  double scb[NV];
#pragma unroll
  for (int b = 0; b < NV; ++b) scb[b] = to[b] < 0 ? 0.0 : sc;
  // Two rows of every factor of the wave -> LDS -> 16-byte stores (a lane storing its own 864-byte Jacobian touches 64 lines per store
  // instruction).  The two rows of a factor are 2 TW contiguous doubles of J, so piece p of the wave's 64 TW pieces belongs to factor
  // p / TW and sits p * 16 + (p / TW) * 32 TW bytes behind the wave's first row pair.  Idle lanes of the last workgroup hold the last factor
  // again and store it again: the same bytes to the same place, no lane conditions in the loop.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  d2_t* sw2 = reinterpret_cast<d2_t*>(sJ + wave * (64 * 2 * TW));
  const int f0 = block * 128 + wave * 64;
  unsigned goff[TW];   // in 16-byte pieces, from g.J
#pragma unroll
  for (int it = 0; it < TW; ++it) {
    const int pp = it * 64 + lane, fi = pp / TW;
    const int fg = min(f0 + fi, g.n - 1);
    goff[it] = (unsigned)(fg * (3 * TW) + (pp - fi * TW));
  }
  d2_t* J2 = reinterpret_cast<d2_t*>(g.J);
#pragma unroll
  for (int h = 0; h < 3; ++h) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int i = 2 * h + rr;
      const double* Ai = A + 6 * i;
      double p1[3], m12[3], m3[3], o[TW];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        p1[j] = -(Ai[0] * R1[3 * j] + Ai[1] * R1[3 * j + 1] + Ai[2] * R1[3 * j + 2]);
        m3[j] = Ai[3] * Jr[j] + Ai[4] * Jr[3 + j] + Ai[5] * Jr[6 + j];
      }
      cross3(Ai, dpr, m12);
#pragma unroll
      for (int j = 0; j < 3; ++j) m12[j] += Ai[3] * G[j] + Ai[4] * G[3 + j] + Ai[5] * G[6 + j];
      if (!EXT) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          o[j] = p1[j] * scb[0];
          o[3 + j] = m12[j] * scb[1];
          o[6 + j] = -p1[j] * scb[2];
          o[9 + j] = m3[j] * scb[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          o[j] = p1[j] * scb[0];
          o[6 + j] = -p1[j] * scb[2];
          o[3 + j] = (m12[0] * Re[3 * j] + m12[1] * Re[3 * j + 1] + m12[2] * Re[3 * j + 2] + p1[0] * X1[j] + p1[1] * X1[3 + j] + p1[2] * X1[6 + j]) * scb[1];
          o[9 + j] = (m3[0] * Re[3 * j] + m3[1] * Re[3 * j + 1] + m3[2] * Re[3 * j + 2] + p1[0] * X2[j] + p1[1] * X2[3 + j] + p1[2] * X2[6 + j]) * scb[3];
          o[12 + j] = (p1[0] * Dr[j] + p1[1] * Dr[3 + j] + p1[2] * Dr[6 + j]) * scb[4];
          o[15 + j] = (m12[j] + m3[j]) * scb[5];
        }
      }
#pragma unroll
      for (int k = 0; k < TW / 2; ++k) sw2[lane * TW + rr * (TW / 2) + k] = d2_t{o[2 * k], o[2 * k + 1]};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < TW; ++it) J2[goff[it] + h * TW] = sw2[it * 64 + lane];
    __builtin_amdgcn_wave_barrier();
  }
}
template <bool EXT, bool WITH_J>
__global__ __launch_bounds__(128) void relpose_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
                                                      double* __restrict__ cost_part) {
  __shared__ __attribute__((aligned(16))) double sJ[kSmallStage<WITH_J>];
  relpose_body<EXT, WITH_J>(g, x, losses, cost_part, (int)blockIdx.x, sJ);
}
// a lidar-inertial window: the IMU factors (a wave each) as the first workgroups of the relative-pose evaluation, instead of a launch of
// their own behind it (as visual_imu_eval_kernel does for a visual-inertial window)
// (body shared by the lone launch and the batched one: IMU factors — a wave each, two per workgroup — then 128 relative-pose factors per workgroup)
template <bool EXT, bool WITH_J>
__device__ __forceinline__ void relpose_imu_eval_body(const int bx, const SmallGroup& delta, const SmallGroup& prior, double* __restrict__ part_delta,
                                                      double* __restrict__ part_prior, int n_imu_blocks, const SmallGroup& g, const double* __restrict__ x,
                                                      const DevLoss* __restrict__ losses, double* __restrict__ cost_part, double* sJ /* kSmallStage<WITH_J> doubles of LDS */) {
  if (bx < n_imu_blocks) {
    const int f = __builtin_amdgcn_readfirstlane(2 * bx + ((int)threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (f < delta.n) imu_delta_body<WITH_J, 2>(delta, f, x, losses, part_delta, lane, sJ);
    else if (f < delta.n + prior.n) imu_prior_body<WITH_J>(prior, f - delta.n, x, losses, part_prior, lane);
    return;
  }
  relpose_body<EXT, WITH_J>(g, x, losses, cost_part, bx - n_imu_blocks, sJ);
}
template <bool EXT, bool WITH_J>
__global__ __launch_bounds__(128) void relpose_imu_eval_kernel(SmallGroup delta, SmallGroup prior, double* __restrict__ part_delta,
                                                               double* __restrict__ part_prior, int n_imu_blocks, SmallGroup g,
                                                               const double* __restrict__ x, const DevLoss* __restrict__ losses,
                                                               double* __restrict__ cost_part, ReduceRide red) {
  // (the evaluation launched ahead of the decision carries the end-of-step reduction of the step just computed as its first workgroups,
  //  as visual_imu_eval_reduce_kernel does for a visual-inertial window)
  const int n_units = (WITH_J && red.n_entries > 0) ? red.n_slots + 1 : 0;
  if ((int)blockIdx.x < n_units) {
    __shared__ double sred[16];
    final_reduce_unit<128>((int)blockIdx.x, (int)threadIdx.x, red, n_units, sred);
    return;
  }
  __shared__ __attribute__((aligned(16))) double sJ[kSmallStage<WITH_J>];
  relpose_imu_eval_body<EXT, WITH_J>((int)blockIdx.x - n_units, delta, prior, part_delta, part_prior, n_imu_blocks, g, x, losses, cost_part, sJ);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory.  A
// window without relative-pose factors of this kind has a zero grid in its entry; EXT (the extrinsics slots) is the group's type.
struct relpose_imu_eval_kernel_Args {
  int bsg_grid;
  SmallGroup delta;
  SmallGroup prior;
  double* part_delta;
  double* part_prior;
  int n_imu_blocks;
  SmallGroup g;
  const double* x;
  const DevLoss* losses;
  double* cost_part;
};
template <bool WITH_J>
__global__ __launch_bounds__(128) void relpose_imu_eval_kernel_batch(const relpose_imu_eval_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const relpose_imu_eval_kernel_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  __shared__ __attribute__((aligned(16))) double sJ[kSmallStage<WITH_J>];
  if (a.g.type == BSGPU_F_RELPOSE_EXT) relpose_imu_eval_body<true, WITH_J>((int)blockIdx.x, a.delta, a.prior, a.part_delta, a.part_prior, a.n_imu_blocks, a.g, a.x, a.losses, a.cost_part, sJ);
  else relpose_imu_eval_body<false, WITH_J>((int)blockIdx.x, a.delta, a.prior, a.part_delta, a.part_prior, a.n_imu_blocks, a.g, a.x, a.losses, a.cost_part, sJ);
}
void batchargs_relpose_imu_eval(BatchArgTable& t, const SmallGroup* g /* null: the window has no such launch */, const SmallGroup& delta, const SmallGroup& prior, const double* x,
                                const DevLoss* losses, double* cost_part, double* part_delta, double* part_prior) {
  relpose_imu_eval_kernel_Args a;
  a.delta = delta; a.prior = prior; a.part_delta = part_delta; a.part_prior = part_prior; a.x = x; a.losses = losses; a.cost_part = cost_part;
  if (g) { a.g = *g; a.n_imu_blocks = (delta.n + prior.n + 1) / 2; a.bsg_grid = a.n_imu_blocks + (g->n + 127) / 128; }
  else { a.g = SmallGroup(); a.n_imu_blocks = 0; a.bsg_grid = 0; }
  t.push(a);
}
void launch_relpose_imu_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J) {
  if (n <= 0 || t.max_grid <= 0) return;
  const auto* A = static_cast<const relpose_imu_eval_kernel_Args*>(t.dev);
  if (with_J) hipLaunchKernelGGL(relpose_imu_eval_kernel_batch<true>, dim3(t.max_grid, n), dim3(128), 0, s, A, dyn, list);
  else hipLaunchKernelGGL(relpose_imu_eval_kernel_batch<false>, dim3(t.max_grid, n), dim3(128), 0, s, A, dyn, list);
}
void launch_relpose_imu_eval(hipStream_t s, const SmallGroup& g, const SmallGroup& delta, const SmallGroup& prior, const double* x,
                             const DevLoss* losses, bool with_J, double* cost_part, double* part_delta, double* part_prior, const ReduceRide* red) {
  ReduceRide rr;
  if (with_J && red) rr = *red;
  const int n_imu_blocks = (delta.n + prior.n + 1) / 2, grid = n_imu_blocks + (g.n + 127) / 128 + (rr.n_entries > 0 ? rr.n_slots + 1 : 0);
  const bool ext = g.type == BSGPU_F_RELPOSE_EXT;
#define BSG_LAUNCH_RI(E, W) hipLaunchKernelGGL((relpose_imu_eval_kernel<E, W>), dim3(grid), dim3(128), 0, s, delta, prior, part_delta, part_prior, n_imu_blocks, g, x, losses, cost_part, rr)
  if (ext) { if (with_J) BSG_LAUNCH_RI(true, true); else BSG_LAUNCH_RI(true, false); }
  else { if (with_J) BSG_LAUNCH_RI(false, true); else BSG_LAUNCH_RI(false, false); }
#undef BSG_LAUNCH_RI
}

// absolute pose prior: blocks (p, q), r = A [p - b_p ; AngleAxis(b_q^-1 q)]
template <bool WITH_J>
__device__ __forceinline__ void abspose_kernel_body(const SmallGroup& g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part, const int bsg_bx) {
  const int f = bsg_bx * 128 + threadIdx.x;
  if (f >= g.n) return;
  const int* xo = g.xoff + (size_t)f * 2;
  const int* to = g.toff + (size_t)f * 2;
  const double* c = g.consts + (size_t)f * 43;
  const double* A = c + 7;
  double e[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) e[k] = x[xo[0] + k] - c[k];
  const double q[4] = {x[xo[1]], x[xo[1] + 1], x[xo[1] + 2], x[xo[1] + 3]};
  const double binv[4] = {c[3], -c[4], -c[5], -c[6]};
  double diff[4];
  quat_mul(binv, q, diff);
  quat_to_angle_axis(diff, e + 3);
  double r[6], s = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) a += A[6 * i + k] * e[k];
    r[i] = a; s += a * a;
  }
  double sc, cost;
  finish_small(g, f, losses, s, &sc, &cost);
  cost_part[f] = cost;
  if (!WITH_J) return;
#pragma unroll
  for (int i = 0; i < 6; ++i) g.r[(size_t)f * 6 + i] = r[i] * sc;
  double Jr[9];
  so3_jr_inv(e + 3, Jr);
  double* Jo = g.J + (size_t)f * 36;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Jo[i * 6 + j] = to[0] < 0 ? 0.0 : A[6 * i + j] * sc;
      const double a = A[6 * i + 3] * Jr[j] + A[6 * i + 4] * Jr[3 + j] + A[6 * i + 5] * Jr[6 + j];
      Jo[i * 6 + 3 + j] = to[1] < 0 ? 0.0 : a * sc;
    }
}
template <bool WITH_J>
__global__ __launch_bounds__(128) void abspose_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part) {
  abspose_kernel_body<WITH_J>(g, x, losses, cost_part, (int)blockIdx.x);
}

// r = A (x - b)  /  r = A ((x2 - x1) - d)
template <bool REL, bool WITH_J>
__device__ __forceinline__ void vec3_kernel_body(const SmallGroup& g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part, const int bsg_bx) {
  const int f = bsg_bx * 128 + threadIdx.x;
  if (f >= g.n) return;
  constexpr int NV = REL ? 2 : 1;
  const int* xo = g.xoff + (size_t)f * NV;
  const int* to = g.toff + (size_t)f * NV;
  const double* c = g.consts + (size_t)f * 12;
  const double* A = c + 3;
  double e[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) e[k] = REL ? (x[xo[1] + k] - x[xo[0] + k] - c[k]) : (x[xo[0] + k] - c[k]);
  double r[3], s = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { r[i] = A[3 * i] * e[0] + A[3 * i + 1] * e[1] + A[3 * i + 2] * e[2]; s += r[i] * r[i]; }
  double sc, cost;
  finish_small(g, f, losses, s, &sc, &cost);
  cost_part[f] = cost;
  if (!WITH_J) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) g.r[(size_t)f * 3 + i] = r[i] * sc;
  double* Jo = g.J + (size_t)f * 3 * 3 * NV;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (REL) {
        Jo[i * 6 + j] = to[0] < 0 ? 0.0 : -A[3 * i + j] * sc;
        Jo[i * 6 + 3 + j] = to[1] < 0 ? 0.0 : A[3 * i + j] * sc;
      } else {
        Jo[i * 3 + j] = to[0] < 0 ? 0.0 : A[3 * i + j] * sc;
      }
    }
}
template <bool REL, bool WITH_J>
__global__ __launch_bounds__(128) void vec3_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part) {
  vec3_kernel_body<REL, WITH_J>(g, x, losses, cost_part, (int)blockIdx.x);
}

// gravity alignment: r = A2x2 [R(q) g_b]_{xy}
template <bool WITH_J>
__device__ __forceinline__ void gravity_kernel_body(const SmallGroup& g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part, const int bsg_bx) {
  const int f = bsg_bx * 128 + threadIdx.x;
  if (f >= g.n) return;
  const int xo = g.xoff[f];
  const int to = g.toff[f];
  const double* c = g.consts + (size_t)f * 7;
  const double q[4] = {x[xo], x[xo + 1], x[xo + 2], x[xo + 3]};
  double R[9], gw[3];
  quat_to_rot_normalized(q, R);
  mat3_vec(R, c, gw);
  const double r0 = c[3] * gw[0] + c[4] * gw[1], r1 = c[5] * gw[0] + c[6] * gw[1];
  double sc, cost;
  finish_small(g, f, losses, r0 * r0 + r1 * r1, &sc, &cost);
  cost_part[f] = cost;
  if (!WITH_J) return;
  g.r[(size_t)f * 2] = r0 * sc; g.r[(size_t)f * 2 + 1] = r1 * sc;
  // d(R g)/dtheta = -R [g]x
  const double Gx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  double D[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) D[3 * i + j] = -(R[3 * i] * Gx[j] + R[3 * i + 1] * Gx[3 + j] + R[3 * i + 2] * Gx[6 + j]);
  double* Jo = g.J + (size_t)f * 6;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Jo[j] = to < 0 ? 0.0 : (c[3] * D[j] + c[4] * D[3 + j]) * sc;
    Jo[3 + j] = to < 0 ? 0.0 : (c[5] * D[j] + c[6] * D[3 + j]) * sc;
  }
}
template <bool WITH_J>
__global__ __launch_bounds__(128) void gravity_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part) {
  gravity_kernel_body<WITH_J>(g, x, losses, cost_part, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// A7 inverse-depth reprojection (bs_constraints/visual/inversedepth_reprojection_functor.h:57-125 and
// ..._functor_unary.h:36-72), one lane per factor.  With T_cam_baselink = (R_cb, t_cb), bearing m, inverse
// depth rho, anchor pose (R_a, p_a) and measurement pose (R_m, p_m):
//   a_b = R_cb^T (m - rho t_cb)            rho-scaled point in the anchor baselink frame
//   v   = R_m^T (R_a a_b + rho (p_a - p_m))          ... in the measurement baselink frame
//   c   = R_cb v + rho t_cb                           = [R|t]_(cm<-ca) (m; rho)
//   r   = w (z - (fx c0/c2 + cx, fy c1/c2 + cy))
// Tangent Jacobians (right perturbation R <- R Exp(d)):
//   dc/dtheta_a = -R_cb R_m^T R_a [a_b]x     dc/dp_a = rho R_cb R_m^T
//   dc/dtheta_m =  R_cb [v]x                 dc/dp_m = -rho R_cb R_m^T
//   dc/drho     =  R_cb R_m^T (p_a - p_m - R_a R_cb^T t_cb) + t_cb
// J layout: 2 x 15 (binary: theta_a, p_a, theta_m, p_m, [rho 0 0]) or 2 x 9 (unary: all zero — the unary
// functor uses T = I, its residual is constant in every block).
// ---------------------------------------------------------------------------------------------------
template <bool UNARY, bool WITH_J>
__device__ __forceinline__ void idp_kernel_body(const SmallGroup& g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part, const int bsg_bx) {
  const int f_raw = bsg_bx * 128 + threadIdx.x;
  const bool live = f_raw < g.n;
  const int f = live ? f_raw : g.n - 1;   // (idle lanes of the last workgroup redo the last factor and store nothing)
  constexpr int NV = UNARY ? 3 : 5;
  const int* xo = g.xoff + (size_t)f * NV;
  const int* to = g.toff + (size_t)f * NV;
  const double* k = g.consts + (size_t)f * 6;
  const DevCamera cam = g.cams[g.cam[f]];
  const double w = k[2], m[3] = {k[3], k[4], k[5]};
  double c[3], rho = 0.0;
  double Ra[9], Rm[9], ab[3], v[3], dpm[3];
  if (UNARY) {
    c[0] = m[0]; c[1] = m[1]; c[2] = m[2];
  } else {
    rho = x[xo[4]];
    const double qa[4] = {x[xo[0]], x[xo[0] + 1], x[xo[0] + 2], x[xo[0] + 3]};
    const double qm[4] = {x[xo[2]], x[xo[2] + 1], x[xo[2] + 2], x[xo[2] + 3]};
    quat_to_rot(qa, Ra);
    quat_to_rot(qm, Rm);
    const double mt[3] = {m[0] - rho * cam.t[0], m[1] - rho * cam.t[1], m[2] - rho * cam.t[2]};
    mat3t_vec(cam.R, mt, ab);
    double wv[3];
    mat3_vec(Ra, ab, wv);
#pragma unroll
    for (int i = 0; i < 3; ++i) { dpm[i] = x[xo[1] + i] - x[xo[3] + i]; wv[i] += rho * dpm[i]; }
    mat3t_vec(Rm, wv, v);
    mat3_vec(cam.R, v, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] += rho * cam.t[i];
  }
  const double iz = 1.0 / c[2];
  const double r0 = w * (k[0] - (cam.fx * c[0] * iz + cam.cx)), r1 = w * (k[1] - (cam.fy * c[1] * iz + cam.cy));
  double sc, cost;
  finish_small(g, f, losses, r0 * r0 + r1 * r1, &sc, &cost);
  block_cost_128(live ? cost : 0.0, cost_part, bsg_bx);
  if (!WITH_J || !live) return;
  g.r[(size_t)f * 2] = r0 * sc; g.r[(size_t)f * 2 + 1] = r1 * sc;
  double* Jo = g.J + (size_t)f * 2 * 3 * NV;
  if (UNARY) {
#pragma unroll
    for (int i = 0; i < 2 * 3 * NV; ++i) Jo[i] = 0.0;
    return;
  }
  // M = -w sc dpi/dc R_cb   (2 x 3): dr/d* = M dv/d*  (+ the t_cb term for rho)
  const double P[6] = {cam.fx * iz, 0.0, -cam.fx * c[0] * iz * iz, 0.0, cam.fy * iz, -cam.fy * c[1] * iz * iz};
  double M[6], MRmt[6], MRmtRa[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      M[3 * i + j] = -w * sc * (P[3 * i] * cam.R[j] + P[3 * i + 1] * cam.R[3 + j] + P[3 * i + 2] * cam.R[6 + j]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)   // M R_m^T
      MRmt[3 * i + j] = M[3 * i] * Rm[3 * j] + M[3 * i + 1] * Rm[3 * j + 1] + M[3 * i + 2] * Rm[3 * j + 2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)   // M R_m^T R_a
      MRmtRa[3 * i + j] = MRmt[3 * i] * Ra[j] + MRmt[3 * i + 1] * Ra[3 + j] + MRmt[3 * i + 2] * Ra[6 + j];
  // d(a_b)/drho = -R_cb^T t_cb
  double dab[3];
  mat3t_vec(cam.R, cam.t, dab);
  double Radab[3];
  mat3_vec(Ra, dab, Radab);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double* A = MRmtRa + 3 * i;   // row of M R_m^T R_a : times -[a_b]x
    const double* B = M + 3 * i;        // row of M           : times  [v]x
    // row * [u]x = (row x u) with sign: (row [u]x)_j = sum_k row_k eps(k, j, l) u_l... written out:
    const double ja[3] = {-(A[1] * ab[2] - A[2] * ab[1]), -(A[2] * ab[0] - A[0] * ab[2]), -(A[0] * ab[1] - A[1] * ab[0])};
    const double jm[3] = {B[1] * v[2] - B[2] * v[1], B[2] * v[0] - B[0] * v[2], B[0] * v[1] - B[1] * v[0]};
    double* Ji = Jo + 15 * i;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ji[j] = to[0] < 0 ? 0.0 : ja[j];
      Ji[3 + j] = to[1] < 0 ? 0.0 : rho * MRmt[3 * i + j];
      Ji[6 + j] = to[2] < 0 ? 0.0 : jm[j];
      Ji[9 + j] = to[3] < 0 ? 0.0 : -rho * MRmt[3 * i + j];
    }
    double jr = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) jr += MRmt[3 * i + j] * (dpm[j] - Radab[j]);
    jr += -w * sc * (P[3 * i] * cam.t[0] + P[3 * i + 1] * cam.t[1] + P[3 * i + 2] * cam.t[2]);
    Ji[12] = to[4] < 0 ? 0.0 : jr;
    Ji[13] = 0.0; Ji[14] = 0.0;
  }
}
template <bool UNARY, bool WITH_J>
__global__ __launch_bounds__(128) void idp_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part) {
  idp_kernel_body<UNARY, WITH_J>(g, x, losses, cost_part, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// Euclidean reprojection with a landmark block that is NOT eliminated (slots q, p, P; J 2 x 9 like the
// visual tables of k_reproj.hip, same closed form: euclidean_reprojection_function.h:66-172)
// ---------------------------------------------------------------------------------------------------
template <bool WITH_J>
__device__ __forceinline__ void reproj_dense_kernel_body(const SmallGroup& g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part, const int bsg_bx) {
  const int f = bsg_bx * 128 + threadIdx.x;
  if (f >= g.n) return;
  const int* xo = g.xoff + (size_t)f * 3;
  const int* to = g.toff + (size_t)f * 3;
  const double* k = g.consts + (size_t)f * 3;
  const DevCamera cam = g.cams[g.cam[f]];
  const double q[4] = {x[xo[0]], x[xo[0] + 1], x[xo[0] + 2], x[xo[0] + 3]};
  const double t[3] = {x[xo[1]], x[xo[1] + 1], x[xo[1] + 2]};
  const double P[3] = {x[xo[2]], x[xo[2] + 1], x[xo[2] + 2]};
  double R[9], a[3], b[3], Pc[3];
  quat_to_rot(q, R);
  mat3t_vec(R, P, a);
  mat3t_vec(R, t, b);
  const double Pb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  mat3_vec(cam.R, Pb, Pc);
  Pc[0] += cam.t[0]; Pc[1] += cam.t[1]; Pc[2] += cam.t[2];
  const double iz = 1.0 / Pc[2], w = k[2];
  const double r0 = w * (k[0] - (cam.fx * Pc[0] + cam.cx * Pc[2]) * iz), r1 = w * (k[1] - (cam.fy * Pc[1] + cam.cy * Pc[2]) * iz);
  double sc, cost;
  finish_small(g, f, losses, r0 * r0 + r1 * r1, &sc, &cost);
  cost_part[f] = cost;
  if (!WITH_J) return;
  g.r[(size_t)f * 2] = r0 * sc; g.r[(size_t)f * 2 + 1] = r1 * sc;
  const double jx0 = cam.fx * iz, jx2 = -cam.fx * Pc[0] * iz * iz, jy1 = cam.fy * iz, jy2 = -cam.fy * Pc[1] * iz * iz;
  const double ws = w * sc;
  double M[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[j] = ws * (jx0 * cam.R[j] + jx2 * cam.R[6 + j]);
    M[3 + j] = ws * (jy1 * cam.R[3 + j] + jy2 * cam.R[6 + j]);
  }
  double* Jo = g.J + (size_t)f * 18;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double m0 = M[3 * i], m1 = M[3 * i + 1], m2 = M[3 * i + 2];
    Jo[9 * i + 0] = to[0] < 0 ? 0.0 : -(m1 * Pb[2] - m2 * Pb[1]);
    Jo[9 * i + 1] = to[0] < 0 ? 0.0 : -(m2 * Pb[0] - m0 * Pb[2]);
    Jo[9 * i + 2] = to[0] < 0 ? 0.0 : -(m0 * Pb[1] - m1 * Pb[0]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double mr = m0 * R[3 * j] + m1 * R[3 * j + 1] + m2 * R[3 * j + 2];
      Jo[9 * i + 3 + j] = to[1] < 0 ? 0.0 : mr;
      Jo[9 * i + 6 + j] = to[2] < 0 ? 0.0 : -mr;
    }
  }
}
template <bool WITH_J>
__global__ __launch_bounds__(128) void reproj_dense_kernel(SmallGroup g, const double* __restrict__ x, const DevLoss* __restrict__ losses,
    double* __restrict__ cost_part) {
  reproj_dense_kernel_body<WITH_J>(g, x, losses, cost_part, (int)blockIdx.x);
}

// entries of a group's cost array: one per 128-factor block for the types whose kernels reduce it (block_cost_128), else one per factor
int small_cost_parts(const SmallGroup& g) {
  const bool by_block = g.type == BSGPU_F_RELPOSE_EXT || g.type == BSGPU_F_RELPOSE || g.type == BSGPU_F_IDP_REPROJ || g.type == BSGPU_F_IDP_REPROJ_UNARY;
  return by_block ? (g.n + 127) / 128 : g.n;
}
void launch_small_eval(hipStream_t s, const SmallGroup& g, const double* x, const DevLoss* losses, bool with_J,
                       double* cost_part) {
  if (g.n == 0) return;
  const int g128 = (g.n + 127) / 128;
#define BSG_LAUNCH(K, grid, block) \
  do { if (with_J) hipLaunchKernelGGL((K<true>), dim3(grid), dim3(block), 0, s, g, x, losses, cost_part); \
       else hipLaunchKernelGGL((K<false>), dim3(grid), dim3(block), 0, s, g, x, losses, cost_part); } while (0)
#define BSG_LAUNCH2(K, B, grid, block) \
  do { if (with_J) hipLaunchKernelGGL((K<B, true>), dim3(grid), dim3(block), 0, s, g, x, losses, cost_part); \
       else hipLaunchKernelGGL((K<B, false>), dim3(grid), dim3(block), 0, s, g, x, losses, cost_part); } while (0)
  switch (g.type) {
    case BSGPU_F_IMU_DELTA: BSG_LAUNCH(imu_delta_kernel, g.n, 64); break;
    case BSGPU_F_IMU_PRIOR: BSG_LAUNCH(imu_prior_kernel, g.n, 64); break;
    case BSGPU_F_RELPOSE_EXT: BSG_LAUNCH2(relpose_kernel, true, g128, 128); break;
    case BSGPU_F_RELPOSE: BSG_LAUNCH2(relpose_kernel, false, g128, 128); break;
    case BSGPU_F_ABSPOSE: BSG_LAUNCH(abspose_kernel, g128, 128); break;
    case BSGPU_F_ABS_VEC3: BSG_LAUNCH2(vec3_kernel, false, g128, 128); break;
    case BSGPU_F_REL_VEC3: BSG_LAUNCH2(vec3_kernel, true, g128, 128); break;
    case BSGPU_F_GRAVITY: BSG_LAUNCH(gravity_kernel, g128, 128); break;
    case BSGPU_F_IDP_REPROJ: BSG_LAUNCH2(idp_kernel, false, g128, 128); break;
    case BSGPU_F_IDP_REPROJ_UNARY: BSG_LAUNCH2(idp_kernel, true, g128, 128); break;
    case BSGPU_F_NUM_TYPES /* internal: reprojection with a non-eliminated landmark */: BSG_LAUNCH(reproj_dense_kernel, g128, 128); break;
    default: break;
  }
#undef BSG_LAUNCH
#undef BSG_LAUNCH2
}

// ---------------------------------------------------------------------------------------------------
// assembly of a pose-only group into the dense reduced system: one workgroup (four waves) per factor, J staged in LDS,
// lanes stride over the (column a, column b) pairs; FP64 atomics into S, the rhs row, grad and hdiag.
// ---------------------------------------------------------------------------------------------------
// up to kSetMax groups per launch (a window has two or three pose-only factor types, some with a single factor: one
// launch each would cost more in dispatch than in work)
__device__ __forceinline__ void small_assemble_kernel_body(const int bsg_bx, const SmallGroupSet& set, double* __restrict__ S, int ld, int rhs_row,
                                                            double* __restrict__ grad, double* __restrict__ hdiag,
                                                            const int* __restrict__ perm) {
  __shared__ double sJ[15 * 30];
  __shared__ double sr[15];
  __shared__ int st[60];
  small_assemble_unit(set, bsg_bx, threadIdx.x, 256, sJ, sr, st, S, ld, rhs_row, grad, hdiag, perm);
}
__global__ __launch_bounds__(256) void small_assemble_kernel(SmallGroupSet set, double* __restrict__ S, int ld, int rhs_row,
                                                            double* __restrict__ grad, double* __restrict__ hdiag,
                                                            const int* __restrict__ perm) {
  small_assemble_kernel_body((int)blockIdx.x, set, S, ld, rhs_row, grad, hdiag, perm);
}

// Assembly by SEGMENTS, for the factor types whose factors pile onto the same 3x3 blocks of J^T J (bsgpu_finalize.cpp decides and
// builds the lists): a segment is up to 64 (factor, slot a, slot b) contributions to one block (ra, rb); sixteen lanes stride over
// them, a butterfly sums the sixteen partial blocks, and the block is added to the reduced system once.  A diagonal segment
// (ra == rb) also yields the block's part of the right-hand side, the gradient and diag(J^T J).  C3 (20 000 relative-pose factors
// over 100 keyframes and one extrinsics variable): 3.6 M FP64 atomics, 720 k of them onto the same 36 addresses, become 0.2 M.
BSG_DEV double sum16(double v) {   // over an aligned group of sixteen lanes
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// One workgroup per GROUP of factors that share every slot's variable (the ~21 lidar constraints between two key frames).  The group's
// record (AsmGroup) names its factors, its slots' tangent offsets and the type's tables: a lane's first load is a factor index, its second
// that factor's rows, all of a lane's loads in flight at once (a loop that fetched index, then rows, per pass was ten dependent trips: 8 of
// the launch's 21 us on C3).  The rows go through LDS and [J r]^T [J r] — J^T J with J^T r as its last row — is formed on the matrix cores:
// the K = count x m rows are the contraction, a wave per 16 x 16 output tile (ONE tile when te + 1 <= 16 columns — relative-pose factors
// whose extrinsics are constant — else the three lower tiles of a 2 x 2 grid).  (One thread per entry with two LDS reads per FMA was bound
// by the compute unit's LDS bandwidth: another 9 us.)  The sums are added to the reduced system once per group — 20 000 relative-pose
// factors of C3 over ~950 keyframe pairs: one set of atomics per pair instead of one per 64 contributions per 3x3 block.
constexpr int kGroupRowMax = 6 * 18;      // doubles of J per factor (m <= 6 rows of tw <= 18)
constexpr int kGroupLoads = (kAsmGroupMax * (kGroupRowMax / 2) + 255) / 256;   // 16-byte loads of a thread
BSG_DEV void small_assemble_group(const AsmGroup* __restrict__ Gp, double* sJ /* kAsmGroupMax x 108 */, double* sr /* kAsmGroupMax x 6 */,
                                  double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag,
                                  const int* __restrict__ perm) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef double d4_t __attribute__((ext_vector_type(4)));
  const int count = Gp->count, m = Gp->m, tw = 3 * Gp->nv, te = Gp->te, per = m * tw, half = per / 2;   // (per is even: checked at finalize)
  if (te == 0) return;
  const double* __restrict__ gJ = Gp->J;
  const double* __restrict__ gr = Gp->r;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nrows = count * m, nr16 = (nrows + 15) & ~15, nh = count * half;
  // ---- every load of the rows asked for before any is waited for
  int fq[kGroupLoads], oq[kGroupLoads];
#pragma unroll
  for (int it = 0; it < kGroupLoads; ++it) {
    const int i = min(tid + 256 * it, nh - 1), q = i / half;
    oq[it] = i - q * half;
    fq[it] = Gp->fac[q];
  }
  const int ir = min(tid, nrows - 1), qr = ir / m;
  const int fr = Gp->fac[qr];
  // where this lane's sums go (D[(lane >> 4) + 4 reg][lane & 15] of its wave's tile); row te of the square is J^T r
  const int n_tiles = te + 1 <= 16 ? 1 : 3;
  const int ti = wave >= 1 ? 1 : 0, tj = wave == 2 ? 1 : 0;      // waves 0, 1, 2 -> tiles (0,0), (1,0), (1,1)
  const bool mine = wave < n_tiles;
  const int cj = 16 * tj + (lane & 15);
  int Rv[4], Cv = -1;
  if (mine) {
    if (cj < te) { const int t = Gp->toff[cj / 3]; Cv = t < 0 ? -1 : t + cj % 3; }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ci = 16 * ti + (lane >> 4) + 4 * reg;
      Rv[reg] = -1;
      if (ci < te) { const int t = Gp->toff[ci / 3]; Rv[reg] = t < 0 ? -1 : t + ci % 3; }
    }
  }
  d2_t vq[kGroupLoads];
#pragma unroll
  for (int it = 0; it < kGroupLoads; ++it) vq[it] = reinterpret_cast<const d2_t*>(gJ + (size_t)fq[it] * per)[oq[it]];
  const double vr = gr[(size_t)fr * m + (ir - qr * m)];
  int pRv[4], pCv = 0;
  if (mine) {
    pCv = Cv >= 0 ? perm[Cv] : 0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) pRv[reg] = Rv[reg] >= 0 ? perm[Rv[reg]] : 0;
  }
  // ---- rows packed: row q m + k at sJ[(q m + k) tw]; the rows up to the next multiple of 16 are zeros
#pragma unroll
  for (int it = 0; it < kGroupLoads; ++it) {
    const int i = tid + 256 * it;
    if (i < nh) reinterpret_cast<d2_t*>(sJ)[i] = vq[it];
  }
  for (int i = nrows * tw + tid; i < nr16 * tw; i += 256) sJ[i] = 0.0;
  if (tid < nr16) sr[tid] = tid < nrows ? vr : 0.0;
  __syncthreads();
  // ---- the contraction shared out over the four waves (one per SIMD: with a wave per tile the first SIMD of a compute unit did the
  // products of every group on it), 4 rows per product: wave w takes the row quads w, w + 4, ...  Column block c of [J r] is the A operand
  // of row tile c and the B operand of column tile c alike.  Operands are loaded unconditionally — from a column that exists — and masked,
  // so that the loads of a pass are in flight ahead of its products; two accumulators per tile in turn (consecutive products into one
  // accumulator are ~175 cycles apart against 64 of issue).
  d4_t acc[3][2];
#pragma unroll
  for (int tl = 0; tl < 3; ++tl) { acc[tl][0] = d4_t{0.0, 0.0, 0.0, 0.0}; acc[tl][1] = d4_t{0.0, 0.0, 0.0, 0.0}; }
  const int c0 = lane & 15, c1 = 16 + c0, lr = lane >> 4;
  const int l0 = lr * tw + min(c0, tw - 1), l1 = lr * tw + min(c1, tw - 1);
  const bool c0J = c0 < te, c0r = c0 == te, c1J = c1 < te, c1r = c1 == te;
  for (int k0 = 8 * wave; k0 < nr16; k0 += 32) {   // (nr16 is a multiple of 16: two quads per pass, or none)
    double x0[2], x1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = k0 + 4 * u;
      const double j0 = sJ[row * tw + l0], rr = sr[row + lr];
      x0[u] = c0J ? j0 : (c0r ? rr : 0.0);
      x1[u] = 0.0;
      if (n_tiles > 1) { const double j1 = sJ[row * tw + l1]; x1[u] = c1J ? j1 : (c1r ? rr : 0.0); }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      acc[0][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], acc[0][u], 0, 0, 0);
      if (n_tiles > 1) {
        acc[1][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x0[u], acc[1][u], 0, 0, 0);
        acc[2][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x1[u], acc[2][u], 0, 0, 0);
      }
    }
  }
  __syncthreads();   // (the rows are done with: the waves' partial tiles take their place)
  // (of tiles (1,0) and (1,1) only rows 16 .. te <= 18 are wanted: register 0 of their results)
  d4_t* sP = reinterpret_cast<d4_t*>(sJ);             // tile (0,0): [wave][lane]
  double* sQ = sJ + 4 * 64 * 4;                       // tiles (1,0), (1,1): [wave][tile - 1][lane]
  sP[wave * 64 + lane] = acc[0][0] + acc[0][1];
  if (n_tiles > 1) {
    sQ[(wave * 2 + 0) * 64 + lane] = acc[1][0][0] + acc[1][1][0];
    sQ[(wave * 2 + 1) * 64 + lane] = acc[2][0][0] + acc[2][1][0];
  }
  __syncthreads();
  if (!mine) return;
  d4_t sum;
  if (wave == 0) sum = (sP[lane] + sP[64 + lane]) + (sP[128 + lane] + sP[192 + lane]);
  else {
    const int q = wave - 1;
    sum = d4_t{(sQ[(0 + q) * 64 + lane] + sQ[(2 + q) * 64 + lane]) + (sQ[(4 + q) * 64 + lane] + sQ[(6 + q) * 64 + lane]), 0.0, 0.0, 0.0};
  }
  if (Cv < 0) return;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int ci = 16 * ti + (lane >> 4) + 4 * reg;
    const double v = sum[reg];
    if (ci > te || v == 0.0) continue;
    if (ci == te) {   // J^T r
      atomicAdd(&S[(size_t)rhs_row * ld + pCv], v);
      atomicAdd(&grad[Cv], v);
      continue;
    }
    if (cj > ci || Rv[reg] < 0) continue;   // (the entries on and below the diagonal; the mirror image goes with them)
    atomicAdd(&S[(size_t)pRv[reg] * ld + pCv], v);
    if (Rv[reg] != Cv) atomicAdd(&S[(size_t)pCv * ld + pRv[reg]], v);
    else atomicAdd(&hdiag[Cv], v);
  }
}
__device__ __forceinline__ void small_assemble_seg_kernel_body(const int bsg_bx, const SmallGroup* __restrict__ groups, int n_seg,
                                                                const int* __restrict__ seg_start, const int* __restrict__ seg_ra,
                                                                const int* __restrict__ seg_rb, const int2* __restrict__ contrib,
                                                                double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                                double* __restrict__ hdiag, const int* __restrict__ perm, const SmallGroupSet& fw,
                                                                int n_fw_units, int n_grp, const AsmGroup* __restrict__ grp,
                                                                int first_grp_block) {
  // (one area for the kinds of workgroup of this launch: 29 KB, five workgroups to a compute unit — with an area per kind it was 33 KB and
  //  four, and C3's 1 080 workgroups took two rounds on 1 024 places)
  __shared__ __attribute__((aligned(32))) double sArea[kAsmGroupMax * kGroupRowMax + kAsmGroupMax * 6];
  if (bsg_bx >= first_grp_block) {   // a group of same-slot factors per workgroup
    small_assemble_group(grp + (bsg_bx - first_grp_block), sArea, sArea + kAsmGroupMax * kGroupRowMax, S, ld, rhs_row, grad, hdiag, perm);
    return;
  }
  if (bsg_bx < n_fw_units) {
    // the groups assembled one workgroup per factor (the IMU factors of a lidar-inertial window), as the first workgroups of this launch
    // instead of a launch of their own (as in pairs_kernel)
    small_assemble_unit(fw, bsg_bx, threadIdx.x, 256, sArea, sArea + 15 * 30, reinterpret_cast<int*>(sArea + 15 * 30 + 16), S, ld, rhs_row, grad, hdiag, perm);
    return;
  }
  // sixteen lanes per segment (a segment of C3 has ~8 contributions, an IMU factor's blocks one or two: a whole wave per segment idles)
  const int seg = (bsg_bx - n_fw_units) * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  if (seg >= n_seg) return;
  const int beg = seg_start[seg], end = seg_start[seg + 1];
  const int ra = seg_ra[seg], rb = seg_rb[seg];
  double acc[9], gs[3], hs[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { gs[i] = 0.0; hs[i] = 0.0; }
  for (int e = beg + lane; e < end; e += 16) {
    const int2 cb = contrib[e];
    const int t = cb.x >> 24, f = cb.x & ((1 << 24) - 1), sa = cb.y >> 8, sb = cb.y & 255;
    const SmallGroup& g = groups[t];
    const int m = g.m, tw = 3 * g.nv;
    const double* J = g.J + (size_t)f * m * tw;
    const double* r = g.r + (size_t)f * m;
    const bool dg = sa == sb;
    // (the last slot may be narrower than three columns: its padding columns do not enter)
    const int wa = sa == g.nv - 1 ? g.w_last : 3, wb = sb == g.nv - 1 ? g.w_last : 3;
    for (int k = 0; k < m; ++k) {
      const double a0 = J[k * tw + 3 * sa], a1 = wa > 1 ? J[k * tw + 3 * sa + 1] : 0.0, a2 = wa > 2 ? J[k * tw + 3 * sa + 2] : 0.0;
      const double b0 = J[k * tw + 3 * sb], b1 = wb > 1 ? J[k * tw + 3 * sb + 1] : 0.0, b2 = wb > 2 ? J[k * tw + 3 * sb + 2] : 0.0;
      acc[0] += a0 * b0; acc[1] += a0 * b1; acc[2] += a0 * b2;
      acc[3] += a1 * b0; acc[4] += a1 * b1; acc[5] += a1 * b2;
      acc[6] += a2 * b0; acc[7] += a2 * b1; acc[8] += a2 * b2;
      if (dg) {
        const double rk = r[k];
        gs[0] += a0 * rk; gs[1] += a1 * rk; gs[2] += a2 * rk;
        hs[0] += a0 * a0; hs[1] += a1 * a1; hs[2] += a2 * a2;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = sum16(acc[i]);
  const bool diag_seg = ra == rb;
  if (diag_seg) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { gs[i] = sum16(gs[i]); hs[i] = sum16(hs[i]); }
  }
  // (the sums of a narrow slot's padding columns are exact zeros and are not written)
  if (lane < 9) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) v = (lane == i) ? acc[i] : v;
    const int rr = ra + lane / 3, cc = rb + lane % 3;
    if (v != 0.0) {
      const size_t pr = (size_t)perm[rr], pc = (size_t)perm[cc];
      atomicAdd(&S[pr * ld + pc], v);
      if (!diag_seg) atomicAdd(&S[pc * ld + pr], v);   // (only the blocks on and below the diagonal have segments: the mirror image goes with them)
    }
  } else if (diag_seg && lane < 12) {
    const int i = lane - 9;
    double g0 = 0.0, h0 = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { g0 = (i == q) ? gs[q] : g0; h0 = (i == q) ? hs[q] : h0; }
    if (h0 != 0.0) {
      const int rr = ra + i;
      atomicAdd(&S[(size_t)rhs_row * ld + perm[rr]], g0);
      atomicAdd(&grad[rr], g0);
      atomicAdd(&hdiag[rr], h0);
    }
  }
}
__global__ __launch_bounds__(256) void small_assemble_seg_kernel(const SmallGroup* __restrict__ groups, int n_seg,
                                                                const int* __restrict__ seg_start, const int* __restrict__ seg_ra,
                                                                const int* __restrict__ seg_rb, const int2* __restrict__ contrib,
                                                                double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                                double* __restrict__ hdiag, const int* __restrict__ perm, SmallGroupSet fw,
                                                                int n_fw_units, int n_grp, const AsmGroup* __restrict__ grp,
                                                                int first_grp_block) {
  small_assemble_seg_kernel_body((int)blockIdx.x, groups, n_seg, seg_start, seg_ra, seg_rb, contrib, S, ld, rhs_row, grad, hdiag, perm, fw, n_fw_units, n_grp, grp, first_grp_block);
}
// ... with the end-of-step reduction of the step before as its first workgroups (a pose-only window's assembly issued ahead of the host's
// decision: k_reproj.hip landmark_reduce_kernel says why)
__global__ __launch_bounds__(256) void small_assemble_seg_reduce_kernel(ReduceRide red, const SmallGroup* __restrict__ groups, int n_seg,
                                                                       const int* __restrict__ seg_start, const int* __restrict__ seg_ra,
                                                                       const int* __restrict__ seg_rb, const int2* __restrict__ contrib,
                                                                       double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                                       double* __restrict__ hdiag, const int* __restrict__ perm, SmallGroupSet fw,
                                                                       int n_fw_units, int n_grp, const AsmGroup* __restrict__ grp,
                                                                       int first_grp_block) {
  const int n_units = red.n_slots + 1;
  if ((int)blockIdx.x < n_units) {
    __shared__ double sred[16];
    final_reduce_unit<256>((int)blockIdx.x, (int)threadIdx.x, red, n_units, sred);
    return;
  }
  small_assemble_seg_kernel_body((int)blockIdx.x - n_units, groups, n_seg, seg_start, seg_ra, seg_rb, contrib, S, ld, rhs_row, grad, hdiag, perm, fw, n_fw_units, n_grp, grp, first_grp_block);
}
// ... with the J^T J / J^T r of the window's dense prior as the launch's last workgroups (marg_body.h: 16 x 16 output tiles, then the gradient's
// row of workgroups) instead of marg_assemble_kernel behind it (7.4 us): both add into S, the gradient and the diagonal with atomics
__global__ __launch_bounds__(256) void small_assemble_seg_marg_kernel(const SmallGroup* __restrict__ groups, int n_seg,
                                                                     const int* __restrict__ seg_start, const int* __restrict__ seg_ra,
                                                                     const int* __restrict__ seg_rb, const int2* __restrict__ contrib,
                                                                     double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad,
                                                                     double* __restrict__ hdiag, const int* __restrict__ perm, SmallGroupSet fw,
                                                                     int n_fw_units, int n_grp, const AsmGroup* __restrict__ grp,
                                                                     int first_grp_block, MargDev m, int first_marg_block, int mg) {
  if ((int)blockIdx.x >= first_marg_block) {
    const int b = (int)blockIdx.x - first_marg_block;
    marg_assemble_kernel_body(b % mg, b / mg, mg + 1, m, S, ld, perm, rhs_row, grad, hdiag);
    return;
  }
  small_assemble_seg_kernel_body((int)blockIdx.x, groups, n_seg, seg_start, seg_ra, seg_rb, contrib, S, ld, rhs_row, grad, hdiag, perm, fw, n_fw_units, n_grp, grp, first_grp_block);
}
// marg: the window's (one) dense prior, carried by this launch; returns whether it was (false: no launch of this kind — the caller launches the prior's own)
bool launch_small_assemble_seg(hipStream_t s, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_ra, const int* seg_rb,
                               const int2* contrib, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm,
                               const SmallGroupSet* fw, int n_fw_units, int n_grp, const AsmGroup* grp, const MargDev* marg, const ReduceRide* red, bool* red_carried) {
  if (red_carried) *red_carried = false;
  if (n_seg <= 0 && n_grp <= 0) return false;
  SmallGroupSet none;
  none.n = 0; none.first[0] = 0;
  const int extra = fw ? n_fw_units : 0;
  const int seg_blocks = (std::max(0, n_seg) + 15) / 16, first_grp_block = extra + seg_blocks;
  const int own = first_grp_block + std::max(0, n_grp);
  if (marg && marg->rows > 0 && marg->cols > 0) {
    const int mg = (marg->cols + 15) / 16;
    hipLaunchKernelGGL(small_assemble_seg_marg_kernel, dim3(own + mg * (mg + 1)), dim3(256), 0, s, groups_dev, std::max(0, n_seg), seg_start, seg_ra, seg_rb,
                       contrib, S, ld, rhs_row, grad, hdiag, perm, fw ? *fw : none, extra, n_grp, grp, first_grp_block, *marg, own, mg);
    return true;
  }
  if (red && red->n_entries > 0) {
    hipLaunchKernelGGL(small_assemble_seg_reduce_kernel, dim3(red->n_slots + 1 + own), dim3(256), 0, s, *red, groups_dev, std::max(0, n_seg), seg_start, seg_ra, seg_rb,
                       contrib, S, ld, rhs_row, grad, hdiag, perm, fw ? *fw : none, extra, n_grp, grp, first_grp_block);
    if (red_carried) *red_carried = true;
    return false;
  }
  hipLaunchKernelGGL(small_assemble_seg_kernel, dim3(own), dim3(256), 0, s, groups_dev, std::max(0, n_seg), seg_start, seg_ra, seg_rb,
                     contrib, S, ld, rhs_row, grad, hdiag, perm, fw ? *fw : none, extra, n_grp, grp, first_grp_block);
  return false;
}

// the first (up to kSetMax) non-empty groups as ONE set of one-factor units, for a caller that runs them inside another launch
// (pairs_kernel); returns the number of units, *n_taken = how many entries of `groups` it consumed
int small_assemble_first_set(const SmallGroup* groups, int n_groups, SmallGroupSet* set, int* n_taken) {
  set->n = 0;
  int blocks = 0, i = 0;
  for (; i < n_groups && set->n < kSetMax; ++i) {
    if (!groups[i].n) continue;
    set->g[set->n] = groups[i]; set->first[set->n] = blocks; set->part[set->n] = nullptr;
    blocks += groups[i].n; ++set->n;
  }
  set->first[set->n] = blocks;
  *n_taken = i;
  return blocks;
}
void launch_small_assemble_set(hipStream_t s, const SmallGroup* groups, int n_groups, double* S, int ld, int rhs_row, double* grad,
                               double* hdiag, const int* perm) {
  SmallGroupSet set;
  set.n = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (!set.n) return;
    set.first[set.n] = blocks;
    hipLaunchKernelGGL(small_assemble_kernel, dim3(blocks), dim3(256), 0, s, set, S, ld, rhs_row, grad, hdiag, perm);
    set.n = 0; blocks = 0;
  };
  for (int i = 0; i < n_groups; ++i) {
    if (!groups[i].n) continue;
    set.g[set.n] = groups[i]; set.first[set.n] = blocks; set.part[set.n] = nullptr;
    blocks += groups[i].n; ++set.n;
    if (set.n == kSetMax) flush();
  }
  flush();
}

// model cost change term of a pose-only group, one lane per residual row:
//   part[f*m + k] = -(J_k d) (r_k + J_k d / 2)
__device__ __forceinline__ void small_mcc_kernel_body(const int bsg_bx, const SmallGroupSet& set, const double* __restrict__ delta, const UpdateRide& up, int first_update_block,
                                                        const ZeroStep& zs, int first_zero_block) {
  __shared__ double s2[2];
  if (first_zero_block >= 0 && bsg_bx >= first_zero_block) {
    // the NEXT step's clearing (the tiles of the reduced system the assembly writes, the pose gradient, diag(J^T J)): nothing reads them
    // any more in this step — the factorisation is done — so the next assembly finds them clean and needs no launch of its own for it
    // (a window with Euclidean landmarks clears in its landmark launch instead)
    const int z = bsg_bx - first_zero_block;
    if (z < zs.n_tiles) {
      const int nt = zs.ld >> 6, ti = zs.tiles[z] / nt, tj = zs.tiles[z] - ti * nt;
      double2* base = reinterpret_cast<double2*>(zs.S + (size_t)ti * 64 * zs.ld + (size_t)tj * 64);
      const int r0 = threadIdx.x >> 5, c2 = threadIdx.x & 31;
#pragma unroll
      for (int p = 0; p < 16; ++p) base[(size_t)(r0 + 4 * p) * (zs.ld >> 1) + c2] = make_double2(0.0, 0.0);
    } else {
      const int i = (z - zs.n_tiles) * 128 + (int)threadIdx.x;
      if (i < zs.na) zs.a[i] = 0.0;
      if (i < zs.nb) zs.b[i] = 0.0;
    }
    return;
  }
  if (up.n_blocks > 0 && bsg_bx >= first_update_block) {
    // a window without Euclidean landmarks: the candidate x (+) delta of every block as extra workgroups of this launch (both only need
    // the step), 128 blocks each — as backsub_mcc_kernel carries it where there are landmarks
    const int unit = bsg_bx - first_update_block, b = unit * 128 + (int)threadIdx.x;
    double d2 = 0.0, x2 = 0.0;
    if (b < up.n_blocks) update_block(up.blocks ? up.blocks[b] : b, up.xoff, up.toff, up.size, up.manifold, up.x, delta, up.x_cand, d2, x2);
    const double a = wave_sum(d2), c = wave_sum(x2);
    __shared__ double s4[4];
    if ((threadIdx.x & 63) == 0) { s4[threadIdx.x >> 6] = a; s4[2 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { up.part[2 * unit] = s4[0] + s4[1]; up.part[2 * unit + 1] = s4[2] + s4[3]; }
    return;
  }
  small_mcc_unit(set, bsg_bx, threadIdx.x, delta, s2);
}
__global__ __launch_bounds__(128) void small_mcc_kernel(SmallGroupSet set, const double* __restrict__ delta, UpdateRide up, int first_update_block,
                                                        ZeroStep zs, int first_zero_block) {
  small_mcc_kernel_body((int)blockIdx.x, set, delta, up, first_update_block, zs, first_zero_block);
}

// the first (up to kSetMax) non-empty groups as ONE set, for a caller that runs their units inside another launch (backsub_mcc_kernel);
// returns the number of 128-row units, *n_taken = how many entries of `groups` it consumed
int small_mcc_first_set(const SmallGroup* groups, double* const* parts, int n_groups, SmallGroupSet* set, int* n_taken) {
  set->n = 0;
  int blocks = 0, i = 0;
  for (; i < n_groups && set->n < kSetMax; ++i) {
    if (!groups[i].n) continue;
    set->g[set->n] = groups[i]; set->first[set->n] = blocks; set->part[set->n] = parts[i];
    blocks += (groups[i].n * groups[i].m + 127) / 128; ++set->n;
  }
  set->first[set->n] = blocks;
  *n_taken = i;
  return blocks;
}
void launch_update_ride_only(hipStream_t s, const double* delta, const UpdateRide& upd) {   // (no pose-only launch to ride in)
  if (upd.n_blocks <= 0) return;
  SmallGroupSet none;
  none.n = 0; none.first[0] = 0;
  hipLaunchKernelGGL(small_mcc_kernel, dim3((upd.n_blocks + 127) / 128), dim3(128), 0, s, none, delta, upd, 0, ZeroStep(), -1);
}
bool launch_small_mcc_set(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* delta, const UpdateRide* upd,
                          const ZeroStep* zero) {
  SmallGroupSet set;
  set.n = 0;
  int blocks = 0;
  bool carried = false;   // the update rides in the first launch
  auto flush = [&]() {
    if (!set.n) return;
    set.first[set.n] = blocks;
    const int upd_units = (upd && !carried && upd->n_blocks > 0) ? (upd->n_blocks + 127) / 128 : 0;
    // (the next step's clearing rides with the update: zero != null only together with upd)
    const int zero_units = (upd_units && zero) ? zero->n_tiles + (std::max(zero->na, zero->nb) + 127) / 128 : 0;
    hipLaunchKernelGGL(small_mcc_kernel, dim3(blocks + upd_units + zero_units), dim3(128), 0, s, set, delta, upd_units ? *upd : UpdateRide(), blocks,
                       zero_units ? *zero : ZeroStep(), zero_units ? blocks + upd_units : -1);
    if (upd_units) carried = true;
    set.n = 0; blocks = 0;
  };
  for (int i = 0; i < n_groups; ++i) {
    if (!groups[i].n) continue;
    set.g[set.n] = groups[i]; set.first[set.n] = blocks; set.part[set.n] = parts[i];
    blocks += (groups[i].n * groups[i].m + 127) / 128; ++set.n;
    if (set.n == kSetMax) flush();
  }
  flush();
  return carried;
}

// ---------------------------------------------------------------------------------------------------
// The pose-only family over several windows in one launch (bsgpu_batch.cpp: lidar-inertial windows, dense-path pose graphs, the
// pose-only factors of any window — bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115 loops over exactly such graphs).
// blockIdx.y picks the window of list `bsg_list`; entry w of a table is what window w's lone launch passes (a zero grid: the window has
// no such launch).  The bodies are the lone kernels' bodies: a window's numbers are those of its lone solve.
// ---------------------------------------------------------------------------------------------------
// the pose-only groups of a window that no fused evaluation carries (launch_small_eval's kernels, one after the other, as ONE launch):
// 128 factors per workgroup — an IMU factor: a wave each, two per workgroup
constexpr int kEvalSetMax = 8;
struct small_eval_set_Args {
  int bsg_grid;
  int n;
  SmallGroup g[kEvalSetMax];
  double* part[kEvalSetMax];
  int first[kEvalSetMax + 1];
  const double* x;
  const DevLoss* losses;
};
template <bool WITH_J>
__device__ __forceinline__ void small_eval_set_dispatch(const small_eval_set_Args& a, const int bsg_bx) {
  int gi = 0;
  while (gi + 1 < a.n && bsg_bx >= a.first[gi + 1]) ++gi;
  const SmallGroup& g = a.g[gi];
  const int bx = bsg_bx - a.first[gi];
  double* part = a.part[gi];
  __shared__ __attribute__((aligned(16))) double sJ[kSmallStage<WITH_J>];
  switch (g.type) {
    case BSGPU_F_IMU_DELTA: { const int f = __builtin_amdgcn_readfirstlane(2 * bx + ((int)threadIdx.x >> 6)); if (f < g.n) imu_delta_body<WITH_J, 2>(g, f, a.x, a.losses, part, threadIdx.x & 63, sJ); break; }
    case BSGPU_F_IMU_PRIOR: { const int f = __builtin_amdgcn_readfirstlane(2 * bx + ((int)threadIdx.x >> 6)); if (f < g.n) imu_prior_body<WITH_J>(g, f, a.x, a.losses, part, threadIdx.x & 63); break; }
    case BSGPU_F_RELPOSE_EXT: relpose_body<true, WITH_J>(g, a.x, a.losses, part, bx, sJ); break;
    case BSGPU_F_RELPOSE: relpose_body<false, WITH_J>(g, a.x, a.losses, part, bx, sJ); break;
    case BSGPU_F_ABSPOSE: abspose_kernel_body<WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_ABS_VEC3: vec3_kernel_body<false, WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_REL_VEC3: vec3_kernel_body<true, WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_GRAVITY: gravity_kernel_body<WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_IDP_REPROJ: idp_kernel_body<false, WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_IDP_REPROJ_UNARY: idp_kernel_body<true, WITH_J>(g, a.x, a.losses, part, bx); break;
    case BSGPU_F_NUM_TYPES: reproj_dense_kernel_body<WITH_J>(g, a.x, a.losses, part, bx); break;
    default: break;
  }
}
template <bool WITH_J>
__global__ __launch_bounds__(128) void small_eval_set_kernel_batch(const small_eval_set_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const small_eval_set_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  small_eval_set_dispatch<WITH_J>(a, (int)blockIdx.x);
}
// ... and for ONE window: its pose-only groups that no fused evaluation carries in one launch instead of one each (a pose graph's relative-pose
// factors and the prior on its first pose: the prior's launch was 7.9 us of latency for one factor, on the path of every evaluation of C4)
template <bool WITH_J>
__global__ __launch_bounds__(128) void small_eval_set_kernel(small_eval_set_Args a) {
  small_eval_set_dispatch<WITH_J>(a, (int)blockIdx.x);
}
static bool fill_small_eval_set(small_eval_set_Args& a, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses) {
  a.n = 0; a.x = x; a.losses = losses;
  int blocks = 0;
  for (int i = 0; i < n_groups; ++i) {
    if (!groups[i].n) continue;
    if (a.n == kEvalSetMax) return false;
    const bool imu = groups[i].type == BSGPU_F_IMU_DELTA || groups[i].type == BSGPU_F_IMU_PRIOR;
    a.g[a.n] = groups[i]; a.part[a.n] = parts[i]; a.first[a.n] = blocks;
    blocks += imu ? (groups[i].n + 1) / 2 : (groups[i].n + 127) / 128;
    ++a.n;
  }
  for (int i = a.n; i <= kEvalSetMax; ++i) a.first[i] = blocks;
  a.bsg_grid = blocks;
  return true;
}
// false: more groups than one launch takes, or a type the launch does not carry (the caller launches them one by one)
bool launch_small_eval_set(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses, bool with_J) {
  small_eval_set_Args a;
  if (!fill_small_eval_set(a, groups, parts, n_groups, x, losses)) return false;
  if (a.bsg_grid <= 0) return true;
  if (with_J) hipLaunchKernelGGL(small_eval_set_kernel<true>, dim3(a.bsg_grid), dim3(128), 0, s, a);
  else hipLaunchKernelGGL(small_eval_set_kernel<false>, dim3(a.bsg_grid), dim3(128), 0, s, a);
  return true;
}
// ... with the window's dense prior (true marginalisation: fixed_lag_smoother.cpp:269-272) as the launch's last workgroups, two rows each:
// marg_eval_kernel was a launch of its own behind this one on the path of every evaluation (4.6 us for a 111 x 159 prior)
template <bool WITH_J>
__global__ __launch_bounds__(128) void small_eval_set_marg_kernel(small_eval_set_Args a, MargDev m, double* __restrict__ marg_part) {
  if ((int)blockIdx.x < a.bsg_grid) { small_eval_set_dispatch<WITH_J>(a, (int)blockIdx.x); return; }
  marg_eval_kernel_body<WITH_J, true, 128>(2 * ((int)blockIdx.x - a.bsg_grid), m, a.x, marg_part);
}
// false: nothing was launched (more groups than one launch takes, a prior too wide for the LDS copy)
bool launch_small_eval_set_marg(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses, bool with_J,
                                const MargDev& m, double* marg_part) {
  small_eval_set_Args a;
  if (!marg_fits_lds(m) || m.rows <= 0 || !fill_small_eval_set(a, groups, parts, n_groups, x, losses)) return false;
  const int grid = a.bsg_grid + (m.rows + 1) / 2;
  if (with_J) hipLaunchKernelGGL(small_eval_set_marg_kernel<true>, dim3(grid), dim3(128), 0, s, a, m, marg_part);
  else hipLaunchKernelGGL(small_eval_set_marg_kernel<false>, dim3(grid), dim3(128), 0, s, a, m, marg_part);
  return true;
}
// groups[i], parts[i]: the window's groups this launch evaluates (n_groups <= kEvalSetMax; 0: a zero grid)
bool batchargs_small_eval_set(BatchArgTable& t, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses) {
  small_eval_set_Args a;
  if (!fill_small_eval_set(a, groups, parts, n_groups, x, losses)) return false;
  t.push(a);
  return true;
}
void launch_small_eval_set_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J) {
  if (n <= 0 || t.max_grid <= 0) return;
  const auto* A = static_cast<const small_eval_set_Args*>(t.dev);
  if (with_J) hipLaunchKernelGGL(small_eval_set_kernel_batch<true>, dim3(t.max_grid, n), dim3(128), 0, s, A, dyn, list);
  else hipLaunchKernelGGL(small_eval_set_kernel_batch<false>, dim3(t.max_grid, n), dim3(128), 0, s, A, dyn, list);
}

struct small_assemble_kernel_Args {
  int bsg_grid;
  SmallGroupSet set;
  double* S; int ld; int rhs_row; double* grad; double* hdiag; const int* perm;
};
__global__ __launch_bounds__(256) void small_assemble_kernel_batch(const small_assemble_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const small_assemble_kernel_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  small_assemble_kernel_body((int)blockIdx.x, a.set, a.S, a.ld, a.rhs_row, a.grad, a.hdiag, a.perm);
}
// the groups assembled one workgroup per factor that ride in no other launch (at most kSetMax of them: false otherwise)
bool batchargs_small_assemble_set(BatchArgTable& t, const SmallGroup* groups, int n_groups, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm) {
  small_assemble_kernel_Args a;
  a.set.n = 0; a.S = S; a.ld = ld; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag; a.perm = perm;
  int blocks = 0;
  for (int i = 0; i < n_groups; ++i) {
    if (!groups[i].n) continue;
    if (a.set.n == kSetMax) return false;
    a.set.g[a.set.n] = groups[i]; a.set.first[a.set.n] = blocks; a.set.part[a.set.n] = nullptr;
    blocks += groups[i].n; ++a.set.n;
  }
  a.set.first[a.set.n] = blocks;
  a.bsg_grid = blocks;
  t.push(a);
  return true;
}
void launch_small_assemble_set_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(small_assemble_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const small_assemble_kernel_Args*>(t.dev), dyn, list);
}

struct small_assemble_seg_kernel_Args {
  int bsg_grid;
  const SmallGroup* groups; int n_seg; const int* seg_start; const int* seg_ra; const int* seg_rb; const int2* contrib;
  double* S; int ld; int rhs_row; double* grad; double* hdiag; const int* perm;
  SmallGroupSet fw; int n_fw_units; int n_grp; const AsmGroup* grp; int first_grp_block;
};
__global__ __launch_bounds__(256) void small_assemble_seg_kernel_batch(const small_assemble_seg_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const small_assemble_seg_kernel_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  small_assemble_seg_kernel_body((int)blockIdx.x, a.groups, a.n_seg, a.seg_start, a.seg_ra, a.seg_rb, a.contrib, a.S, a.ld, a.rhs_row, a.grad, a.hdiag, a.perm, a.fw, a.n_fw_units,
                                 a.n_grp, a.grp, a.first_grp_block);
}
void batchargs_small_assemble_seg(BatchArgTable& t, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_ra, const int* seg_rb, const int2* contrib,
                                  double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* fw, int n_fw_units, int n_grp,
                                  const AsmGroup* grp) {
  small_assemble_seg_kernel_Args a;
  SmallGroupSet none;
  none.n = 0; none.first[0] = 0;
  const int extra = fw ? n_fw_units : 0;
  const int seg_blocks = (std::max(0, n_seg) + 15) / 16;
  a.first_grp_block = extra + seg_blocks;
  a.bsg_grid = (n_seg <= 0 && n_grp <= 0) ? 0 : a.first_grp_block + std::max(0, n_grp);
  a.groups = groups_dev; a.n_seg = std::max(0, n_seg); a.seg_start = seg_start; a.seg_ra = seg_ra; a.seg_rb = seg_rb; a.contrib = contrib;
  a.S = S; a.ld = ld; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag; a.perm = perm; a.fw = fw ? *fw : none; a.n_fw_units = extra; a.n_grp = n_grp; a.grp = grp;
  t.push(a);
}
void launch_small_assemble_seg_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(small_assemble_seg_kernel_batch, dim3(t.max_grid, n), dim3(256), 0, s, static_cast<const small_assemble_seg_kernel_Args*>(t.dev), dyn, list);
}

struct small_mcc_kernel_Args {
  int bsg_grid;
  SmallGroupSet set; const double* delta; UpdateRide up; int first_update_block; ZeroStep zs; int first_zero_block;
};
__global__ __launch_bounds__(128) void small_mcc_kernel_batch(const small_mcc_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const small_mcc_kernel_Args& a = bsg_A[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  small_mcc_kernel_body((int)blockIdx.x, a.set, a.delta, a.up, a.first_update_block, a.zs, a.first_zero_block);
}
// the pose-only groups whose model-cost terms ride in no other launch (at most kSetMax: false otherwise), with the candidate of every block
// (upd) and the next step's clearing (zero; only together with upd) as the launch's last workgroups — launch_small_mcc_set's single launch
bool batchargs_small_mcc(BatchArgTable& t, const SmallGroup* groups, double* const* parts, int n_groups, const double* delta, const UpdateRide* upd, const ZeroStep* zero) {
  small_mcc_kernel_Args a;
  a.set.n = 0; a.delta = delta;
  int blocks = 0;
  for (int i = 0; i < n_groups; ++i) {
    if (!groups[i].n) continue;
    if (a.set.n == kSetMax) return false;
    a.set.g[a.set.n] = groups[i]; a.set.first[a.set.n] = blocks; a.set.part[a.set.n] = parts[i];
    blocks += (groups[i].n * groups[i].m + 127) / 128; ++a.set.n;
  }
  a.set.first[a.set.n] = blocks;
  const int upd_units = (upd && upd->n_blocks > 0) ? (upd->n_blocks + 127) / 128 : 0;
  const int zero_units = (upd_units && zero) ? zero->n_tiles + (std::max(zero->na, zero->nb) + 127) / 128 : 0;
  a.up = upd_units ? *upd : UpdateRide(); a.first_update_block = blocks;
  a.zs = zero_units ? *zero : ZeroStep(); a.first_zero_block = zero_units ? blocks + upd_units : -1;
  a.bsg_grid = blocks + upd_units + zero_units;
  t.push(a);
  return true;
}
void launch_small_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL(small_mcc_kernel_batch, dim3(t.max_grid, n), dim3(128), 0, s, static_cast<const small_mcc_kernel_Args*>(t.dev), dyn, list);
}

}  // namespace bsg
