// =================================================================================================
// TEST INFRASTRUCTURE — CPU ORACLE.  NOT PART OF THE PRODUCT.
//
// Double-precision CPU restatement of the reference's fixed-lag-smoother solve path
// (bs_optimizers/src/fixed_lag_smoother.cpp:281 -> fuse HashGraph::optimize -> ceres::Solve with the
// cost functors of bs_constraints).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library, and only as the checker / reported baseline.
//
// PARITY STATUS: the reference cannot be built in this environment (Ceres, fuse, Eigen, libbeam, ROS
// are absent: SURVEY.md §8c), so there is no reference binary behind this oracle.  It is pinned
// against (tests/test_oracle_*.py): the reference's own known-answer tests
// (absolute_imu_state_3d_stamped_constraint_test.cpp:22-52,150-165; imu_preintegration_tests.cpp:292-477;
// reprojection_test.cpp:21-73), its property tests (jacobian_helper_tests.cpp, euclidean_reprojection_test.cpp:183-196),
// central finite differences on every functor, and scipy.optimize.least_squares on small graphs.
// Per-iteration LM trajectories and large-graph results are NOT pinned by anything the reference
// ships: "parity unpinned" for those.
//
// Exposes the same C entry points as include/bsgpu.h with the prefix bso_ instead of bsgpu_.
// =================================================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/bsgpu.h"
#include "functors.h"

namespace bso {

struct TypeInfo {
  int nidx;      // int32 per factor in block_idx (variable slots + optional camera slot)
  int nvar;      // variable slots
  int nconst;
  int m;         // residual rows
  int amb[10];   // ambient size per variable slot
};
static const TypeInfo kTypes[BSGPU_F_NUM_TYPES] = {
    /* REPROJ        */ {4, 3, 3, 2, {4, 3, 3}},
    /* REPROJ_ONLINE */ {6, 5, 3, 2, {4, 3, 3, 4, 3}},
    /* IMU_DELTA     */ {10, 10, 287, 15, {4, 3, 3, 3, 3, 4, 3, 3, 3, 3}},
    /* IMU_PRIOR     */ {5, 5, 241, 15, {4, 3, 3, 3, 3}},
    /* RELPOSE_EXT   */ {6, 6, 43, 6, {3, 4, 3, 4, 3, 4}},
    /* RELPOSE       */ {4, 4, 43, 6, {3, 4, 3, 4}},
    /* ABSPOSE       */ {2, 2, 43, 6, {3, 4}},
    /* ABS_VEC3      */ {1, 1, 12, 3, {3}},
    /* REL_VEC3      */ {2, 2, 12, 3, {3, 3}},
    /* GRAVITY       */ {1, 1, 7, 2, {4}},
    /* IDP_REPROJ    */ {6, 5, 6, 2, {4, 3, 4, 3, 1}},
    /* IDP_UNARY     */ {4, 3, 6, 2, {4, 3, 1}},
};
// tangent width of a slot.  J rows are laid out with slot sl starting at column 3*sl: a slot narrower than 3
// (the inverse-depth scalar) is only ever the LAST slot of its type, so that offset stays valid.
static inline int tsz(const TypeInfo& ti, int sl) { return ti.amb[sl] == 4 ? 3 : ti.amb[sl]; }
static inline bool has_camera(int type) { return type <= 1 || type == BSGPU_F_IDP_REPROJ || type == BSGPU_F_IDP_REPROJ_UNARY; }

struct Group {
  int type = 0;
  int n = 0;
  std::vector<int32_t> idx;
  std::vector<double> consts;
  std::vector<int32_t> loss_kind;
  std::vector<double> loss_a;
  // evaluation outputs (robustified)
  int tw = 0;                     // tangent width = sum of slot tangent sizes (3 per slot here)
  std::vector<double> r;          // n x m
  std::vector<double> J;          // n x m x tw  (tangent, columns per slot, 3 each)
  std::vector<uint8_t> active;    // 0 => all blocks constant (cost goes to fixed_cost)
  int row0 = 0;                   // first residual row of this group
};

// [EXT] fuse_constraints::MarginalConstraint (what fuse_constraints::marginalizeVariables puts into the graph,
// bs_optimizers/src/fixed_lag_smoother.cpp:270-271), restated from fuse's published MarginalCostFunction:
//   r = b + sum_i A_i (x_i [-] xbar_i),   [-] = LocalParameterization::Minus(xbar_i, x_i)
//   (quaternion: QuaternionToAngleAxis(xbar^-1 (x) x), bs_constraints/src/jacobians.cpp:37-50)
//   ambient Jacobian A_i * MinusJacobian(x_i) (jacobians.cpp:160-174; fuse evaluates ComputeMinusJacobian at the
//   current parameter value), so the tangent Jacobian MinusJacobian(x) PlusJacobian(x) A_i-weighted is A_i |x|^2
struct Marginal {
  std::vector<int32_t> blocks;
  int rows = 0, cols = 0;            // cols = sum of tangent sizes
  std::vector<double> A, b, xbar;    // A: rows x cols row-major
  std::vector<int> col_t;            // per column: tangent index, -1 for a constant block
  std::vector<double> r, J;          // evaluation outputs (J: rows x cols, tangent)
  int row0 = 0;
  bool active = true;
};

struct Iter {
  bsgpu_iteration it;
};

struct Ctx {
  std::string err;
  int nb = 0;
  std::vector<double> x, x0;
  std::vector<int32_t> off;
  std::vector<uint8_t> size, manifold, is_const, is_const_in;   // is_const_in: as given; is_const: + blocks no factor touches
  std::vector<int> tsize, toff;
  std::vector<uint8_t> is_lm;
  std::vector<int> pose_blocks, lm_blocks;  // in tangent order
  int n_tan = 0, n_pose = 0, n_lm = 0;
  std::vector<Camera> cams;
  Group groups[BSGPU_F_NUM_TYPES];
  std::vector<Marginal> marginals;
  struct MargResult { std::vector<int32_t> kept; int rows = 0, cols = 0; std::vector<double> A, b, xbar; bool valid = false; } marg_result;
  bool finalized = false;
  int num_res = 0;
  int reproj_mode = 0;  // 0 closed form, 1 reference FD quaternion Jacobian, 2 autodiff
  int num_threads = 0;
  std::vector<bsgpu_iteration> iters;
  // landmark structure
  std::vector<int> lm_fac_start, lm_fac;   // CSR landmark -> (type<<28 | factor)
  std::vector<int> cam_id_of_fac[2];       // per reproj group: owner id
  int n_owner = 0;
  std::vector<int> owner_start, owner_fac;  // CSR owner -> (type<<28|factor)
  bool owner_parallel_ok = true;
  // bso_sync_factors_indirect: the table of the previous call per type (kept across bso_clear), to check the caller's change lists
  struct SyncPrev { bool valid = false; std::vector<int32_t> idx, loss_kind; std::vector<double> consts, loss_a; };
  std::vector<SyncPrev> sync_prev = std::vector<SyncPrev>(BSGPU_F_NUM_TYPES);
};

static inline void plus_jacobian(const double* q, double P[12]) {
  // bs_constraints/src/jacobians.cpp:144-158 (== fuse Orientation3DLocalParameterization::ComputeJacobian)
  const double x0 = q[0] / 2, x1 = q[1] / 2, x2 = q[2] / 2, x3 = q[3] / 2;
  P[0] = -x1; P[1] = -x2; P[2] = -x3;
  P[3] = x0;  P[4] = -x3; P[5] = x2;
  P[6] = x3;  P[7] = x0;  P[8] = -x1;
  P[9] = -x2; P[10] = x1; P[11] = x0;
}

static inline void manifold_plus(int kind, int size, const double* x, const double* d, double* out) {
  if (kind == BSGPU_MANIFOLD_QUAT_RIGHT) {
    // bs_constraints/src/jacobians.cpp:24-35: x (x) AngleAxisToQuaternion(delta)
    double qd[4];
    AngleAxisToQuaternion(d, qd);
    QuaternionProduct(x, qd, out);
  } else {
    for (int i = 0; i < size; ++i) out[i] = x[i] + d[i];
  }
}

// ------------------------------------------------------------------------------------------------
// per-type evaluation
// ------------------------------------------------------------------------------------------------
template <typename T> struct Slots { const T* p[10]; };

template <typename T>
static void eval_functor(const Ctx& c, int type, const int32_t* idx, const double* k, const Slots<T>& s, T* r) {
  switch (type) {
    case BSGPU_F_REPROJ: {
      const Camera& cam = c.cams[idx[3]];
      ReprojResidual(cam, k, k[2], s.p[0], s.p[1], s.p[2], r);
    } break;
    case BSGPU_F_REPROJ_ONLINE_CALIB: {
      const Camera& cam = c.cams[idx[5]];
      ReprojOnlineCalibResidual(cam, k, k[2], s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r);
    } break;
    case BSGPU_F_IMU_DELTA:
      ImuDeltaResidual(k, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.p[5], s.p[6], s.p[7], s.p[8], s.p[9], r);
      break;
    case BSGPU_F_IMU_PRIOR: ImuPriorResidual(k, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r); break;
    case BSGPU_F_RELPOSE_EXT: DeltaPoseExtResidual(k, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.p[5], r); break;
    case BSGPU_F_RELPOSE: DeltaPoseResidual(k, s.p[0], s.p[1], s.p[2], s.p[3], r); break;
    case BSGPU_F_ABSPOSE: PriorPoseResidual(k, s.p[0], s.p[1], r); break;
    case BSGPU_F_ABS_VEC3: AbsVec3Residual(k, s.p[0], r); break;
    case BSGPU_F_REL_VEC3: RelVec3Residual(k, s.p[0], s.p[1], r); break;
    case BSGPU_F_GRAVITY: GravityResidual(k, s.p[0], r); break;
    case BSGPU_F_IDP_REPROJ: InverseDepthReprojResidual(c.cams[idx[5]], k, s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r); break;
    case BSGPU_F_IDP_REPROJ_UNARY: InverseDepthReprojUnaryResidual(c.cams[idx[3]], k, s.p[0], s.p[1], s.p[2], r); break;
    default: break;
  }
}

template <int N>
static void autodiff(const Ctx& c, int type, const int32_t* idx, const double* k, const double* x, double* r,
                     double* Jamb /* m x N */) {
  const TypeInfo& ti = kTypes[type];
  Jet<N> store[N];
  Slots<Jet<N>> s;
  int col = 0;
  for (int sl = 0; sl < ti.nvar; ++sl) {
    const double* xp = x + c.off[idx[sl]];
    s.p[sl] = store + col;
    for (int i = 0; i < ti.amb[sl]; ++i, ++col) store[col] = Jet<N>(xp[i], col);
  }
  Jet<N> rj[15];
  eval_functor<Jet<N>>(c, type, idx, k, s, rj);
  for (int i = 0; i < ti.m; ++i) {
    r[i] = rj[i].a;
    for (int j = 0; j < N; ++j) Jamb[i * N + j] = rj[i].v[j];
  }
}

static void eval_value(const Ctx& c, int type, const int32_t* idx, const double* k, const double* x, double* r) {
  const TypeInfo& ti = kTypes[type];
  Slots<double> s;
  for (int sl = 0; sl < ti.nvar; ++sl) s.p[sl] = x + c.off[idx[sl]];
  eval_functor<double>(c, type, idx, k, s, r);
}

// closed-form tangent Jacobian of the reprojection factor:
//   dr/dtheta = -A Jpi R_cb [P_b]x ; dr/dt = +A Jpi R_cb R^T ; dr/dP = -A Jpi R_cb R^T
// plus the reference's own forward-difference variant for the quaternion block
// (euclidean_reprojection_function.h:124-143) when mode == 1.
static void reproj_analytic(const Ctx& c, int mode, const int32_t* idx, const double* k, const double* x,
                            double* r, double* Jt /* 2 x 9 tangent */) {
  const Camera& cam = c.cams[idx[3]];
  const double* q = x + c.off[idx[0]];
  const double* t = x + c.off[idx[1]];
  const double* P = x + c.off[idx[2]];
  const double w = k[2];
  double R[9];
  EigenQuatToRot(q, R);
  double a[3], b[3], Pb[3], Pc[3];
  Mat3TVec(R, P, a);
  Mat3TVec(R, t, b);
  for (int i = 0; i < 3; ++i) Pb[i] = a[i] - b[i];
  for (int i = 0; i < 3; ++i)
    Pc[i] = cam.R_cb[3 * i] * Pb[0] + cam.R_cb[3 * i + 1] * Pb[1] + cam.R_cb[3 * i + 2] * Pb[2] + cam.t_cb[i];
  const double hx = cam.fx * Pc[0] + cam.cx * Pc[2];
  const double hy = cam.fy * Pc[1] + cam.cy * Pc[2];
  r[0] = w * (k[0] - hx / Pc[2]);
  r[1] = w * (k[1] - hy / Pc[2]);
  // DImageProjectionDPoint (jacobians.cpp:202-214)
  const double z = Pc[2], z2 = z * z;
  const double Jpi[6] = {cam.fx / z, 0.0, -cam.fx * Pc[0] / z2, 0.0, cam.fy / z, -cam.fy * Pc[1] / z2};
  // M = Jpi * R_cb  (2x3)
  double M[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      M[3 * i + j] = Jpi[3 * i] * cam.R_cb[j] + Jpi[3 * i + 1] * cam.R_cb[3 + j] + Jpi[3 * i + 2] * cam.R_cb[6 + j];
  // MRt = M * R^T (2x3)
  double MRt[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) MRt[3 * i + j] = M[3 * i] * R[3 * j] + M[3 * i + 1] * R[3 * j + 1] + M[3 * i + 2] * R[3 * j + 2];
  if (mode == 0) {
    // -w * M * [Pb]x
    const double S[9] = {0, -Pb[2], Pb[1], Pb[2], 0, -Pb[0], -Pb[1], Pb[0], 0};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        Jt[9 * i + j] = -w * (M[3 * i] * S[j] + M[3 * i + 1] * S[3 + j] + M[3 * i + 2] * S[6 + j]);
  } else {
    // reference: forward difference on the 4 quaternion coefficients with re-normalisation,
    // EPSILON = 1e-8 (euclidean_reprojection_function.h:13-24,124-143), then x PlusJacobian
    auto point_transformation = [&](const double qq[4], double out[3]) {
      const double n = std::sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
      const double qn[4] = {qq[0] / n, qq[1] / n, qq[2] / n, qq[3] / n};
      double Rn[9], u[3], v[3];
      EigenQuatToRot(qn, Rn);
      Mat3TVec(Rn, P, u);
      Mat3TVec(Rn, t, v);
      for (int i = 0; i < 3; ++i) out[i] = u[i] - v[i];
    };
    const double eps = 1e-8;
    double res[3];
    point_transformation(q, res);
    double dPb_dq[12];  // 3x4
    for (int i = 0; i < 4; ++i) {
      double qp[4] = {q[0], q[1], q[2], q[3]};
      qp[i] += eps;
      double rp[3];
      point_transformation(qp, rp);
      for (int rr = 0; rr < 3; ++rr) dPb_dq[4 * rr + i] = (rp[rr] - res[rr]) / eps;
    }
    double Jq[8];  // 2x4 = -w * M * dPb_dq
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        Jq[4 * i + j] = -w * (M[3 * i] * dPb_dq[j] + M[3 * i + 1] * dPb_dq[4 + j] + M[3 * i + 2] * dPb_dq[8 + j]);
    double Pj[12];
    plus_jacobian(q, Pj);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int kk = 0; kk < 4; ++kk) s += Jq[4 * i + kk] * Pj[3 * kk + j];
        Jt[9 * i + j] = s;
      }
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      Jt[9 * i + 3 + j] = w * MRt[3 * i + j];    // d/dt  (function.h:146-157: -A * dE * R_cb * (-R^T))
      Jt[9 * i + 6 + j] = -w * MRt[3 * i + j];   // d/dP  (function.h:159-169)
    }
}

static void amb_to_tan(const Ctx& c, int type, const int32_t* idx, const double* x, int N, const double* Jamb,
                       double* Jt, int tw) {
  const TypeInfo& ti = kTypes[type];
  int ca = 0, ct = 0;
  for (int sl = 0; sl < ti.nvar; ++sl) {
    const int b = idx[sl];
    if (c.manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) {
      double Pj[12];
      plus_jacobian(x + c.off[b], Pj);
      for (int i = 0; i < ti.m; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int kk = 0; kk < 4; ++kk) s += Jamb[i * N + ca + kk] * Pj[3 * kk + j];
          Jt[i * tw + ct + j] = s;
        }
      ca += 4; ct += 3;
    } else {
      for (int i = 0; i < ti.m; ++i)
        for (int j = 0; j < ti.amb[sl]; ++j) Jt[i * tw + ct + j] = Jamb[i * N + ca + j];
      ca += ti.amb[sl]; ct += ti.amb[sl];
    }
  }
}

static void eval_factor(const Ctx& c, int type, const int32_t* idx, const double* k, const double* x, double* r,
                        double* Jt, int tw) {
  const TypeInfo& ti = kTypes[type];
  if (!Jt) { eval_value(c, type, idx, k, x, r); return; }
  if (type == BSGPU_F_REPROJ && c.reproj_mode != 2) { reproj_analytic(c, c.reproj_mode, idx, k, x, r, Jt); return; }
  double Jamb[15 * 32];
  int N = 0;
  for (int sl = 0; sl < ti.nvar; ++sl) N += ti.amb[sl];
  switch (N) {
    case 3: autodiff<3>(c, type, idx, k, x, r, Jamb); break;
    case 4: autodiff<4>(c, type, idx, k, x, r, Jamb); break;
    case 6: autodiff<6>(c, type, idx, k, x, r, Jamb); break;
    case 7: autodiff<7>(c, type, idx, k, x, r, Jamb); break;
    case 10: autodiff<10>(c, type, idx, k, x, r, Jamb); break;
    case 8: autodiff<8>(c, type, idx, k, x, r, Jamb); break;
    case 14: autodiff<14>(c, type, idx, k, x, r, Jamb); break;
    case 15: autodiff<15>(c, type, idx, k, x, r, Jamb); break;
    case 16: autodiff<16>(c, type, idx, k, x, r, Jamb); break;
    case 17: autodiff<17>(c, type, idx, k, x, r, Jamb); break;
    case 21: autodiff<21>(c, type, idx, k, x, r, Jamb); break;
    case 32: autodiff<32>(c, type, idx, k, x, r, Jamb); break;
    default: break;
  }
  amb_to_tan(c, type, idx, x, N, Jamb, Jt, tw);
}

// ceres::LossFunction::Evaluate -> rho[0..2]
static inline void loss_eval(int kind, double a, double s, double rho[3]) {
  if (kind == BSGPU_LOSS_CAUCHY) {
    const double b = a * a, cc = 1.0 / b;
    const double sum = 1.0 + s * cc, inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -cc * (inv * inv);
  } else if (kind == BSGPU_LOSS_HUBER) {
    const double b = a * a;
    if (s > b) {
      const double rr = std::sqrt(s);
      rho[0] = 2.0 * a * rr - b;
      rho[1] = std::max(std::numeric_limits<double>::min(), a / rr);
      rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

// Evaluate all factors at x.  Returns cost of active factors; fixed cost separately.
// ceres ResidualBlock::Evaluate: cost = 0.5 rho(|r|^2); Corrector with rho'' <= 0 (Cauchy/Huber):
// r <- sqrt(rho') r, J <- sqrt(rho') J.
static double evaluate(Ctx& c, const double* x, bool want_J, double* fixed_cost) {
  double cost = 0.0, fixed = 0.0;
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    Group& g = c.groups[t];
    if (!g.n) continue;
    const TypeInfo& ti = kTypes[t];
    const int m = ti.m, tw = g.tw;
    double gc = 0.0, gf = 0.0;
#pragma omp parallel for reduction(+ : gc, gf) schedule(static) if (g.n > 512)
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      const double* k = &g.consts[(size_t)f * ti.nconst];
      // a cost-only evaluation (the candidate point) must not disturb residuals_/jacobian_ of the
      // current point: Ceres evaluates the candidate into scratch (TrustRegionMinimizer::
      // ComputeCandidatePointAndEvaluateCost passes NULL for residuals/jacobian)
      double r_scratch[15];
      double* r = want_J ? &g.r[(size_t)f * m] : r_scratch;
      double* J = want_J ? &g.J[(size_t)f * m * tw] : nullptr;
      eval_factor(c, t, idx, k, x, r, J, tw);
      double s = 0;
      for (int i = 0; i < m; ++i) s += r[i] * r[i];
      double rho[3];
      loss_eval(g.loss_kind[f], g.loss_a[f], s, rho);
      if (g.active[f]) gc += 0.5 * rho[0]; else gf += 0.5 * rho[0];
      const double sc = std::sqrt(rho[1]);
      if (sc != 1.0) {
        if (J) for (int i = 0; i < m * tw; ++i) J[i] *= sc;
        for (int i = 0; i < m; ++i) r[i] *= sc;
      }
      if (J) {  // zero the columns of constant blocks
        int ct = 0;
        for (int sl = 0; sl < ti.nvar; ++sl) {
          const int b = idx[sl];
          const int ts = c.tsize[b];
          if (c.is_const[b]) for (int i = 0; i < m; ++i) for (int j = 0; j < ts; ++j) J[i * tw + ct + j] = 0.0;
          ct += ts;
        }
      }
    }
    cost += gc; fixed += gf;
  }
  for (Marginal& mg : c.marginals) {
    std::vector<double> delta(mg.cols), r_scratch;
    if (want_J) mg.J.assign((size_t)mg.rows * mg.cols, 0.0);
    int ct = 0, ca = 0;
    for (size_t i = 0; i < mg.blocks.size(); ++i) {
      const int b = mg.blocks[i];
      const double* xb = &mg.xbar[ca];
      const double* xx = x + c.off[b];
      if (c.manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) {
        double inv[4], d[4];
        EigenQuatConj(xb, inv);           // jacobians.cpp:3-8,44-45 QuaternionInverse = conjugate
        QuaternionProduct(inv, xx, d);
        QuaternionToAngleAxis(d, &delta[ct]);
        if (want_J) {
          // ComputeMinusJacobian(parameters[i]): "the Jacobian of Minus(x1, x2) w.r.t. x2 evaluated at x1 = x2 = x"
          const double M[12] = {-2 * xx[1], 2 * xx[0], 2 * xx[3], -2 * xx[2], -2 * xx[2], -2 * xx[3], 2 * xx[0], 2 * xx[1],
                                -2 * xx[3], 2 * xx[2], -2 * xx[1], 2 * xx[0]};   // jacobians.cpp:160-174
          double P[12];
          plus_jacobian(xx, P);
          double MP[9];
          for (int a = 0; a < 3; ++a) for (int e = 0; e < 3; ++e) { double sacc = 0; for (int k = 0; k < 4; ++k) sacc += M[4 * a + k] * P[3 * k + e]; MP[3 * a + e] = sacc; }
          for (int rr = 0; rr < mg.rows; ++rr)
            for (int e = 0; e < 3; ++e) {
              double sacc = 0;
              for (int a = 0; a < 3; ++a) sacc += mg.A[(size_t)rr * mg.cols + ct + a] * MP[3 * a + e];
              mg.J[(size_t)rr * mg.cols + ct + e] = sacc;
            }
        }
        ct += 3; ca += 4;
      } else {
        const int sz = c.size[b];
        for (int k = 0; k < sz; ++k) delta[ct + k] = xx[k] - xb[k];
        if (want_J) for (int rr = 0; rr < mg.rows; ++rr) for (int k = 0; k < sz; ++k) mg.J[(size_t)rr * mg.cols + ct + k] = mg.A[(size_t)rr * mg.cols + ct + k];
        ct += sz; ca += sz;
      }
    }
    if (want_J) mg.r.assign(mg.rows, 0.0); else r_scratch.assign(mg.rows, 0.0);
    double* r = want_J ? mg.r.data() : r_scratch.data();
    double sacc = 0;
    for (int rr = 0; rr < mg.rows; ++rr) {
      double v = mg.b[rr];
      for (int k = 0; k < mg.cols; ++k) v += mg.A[(size_t)rr * mg.cols + k] * delta[k];
      r[rr] = v; sacc += v * v;
    }
    if (mg.active) cost += 0.5 * sacc; else fixed += 0.5 * sacc;
    if (want_J) for (int k = 0; k < mg.cols; ++k) if (mg.col_t[k] < 0) for (int rr = 0; rr < mg.rows; ++rr) mg.J[(size_t)rr * mg.cols + k] = 0.0;
  }
  if (fixed_cost) *fixed_cost = fixed;
  return cost;
}

// ------------------------------------------------------------------------------------------------
// structure
// ------------------------------------------------------------------------------------------------
static int finalize(Ctx& c) {
  if (c.finalized) return BSGPU_OK;
  // validate + landmark detection
  std::vector<int> lm_use(c.nb, 0), other_use(c.nb, 0);
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sl = 0; sl < ti.nvar; ++sl) {
        const int b = idx[sl];
        if (b < 0 || b >= c.nb) { c.err = "factor references block out of range"; return BSGPU_ERR_INVALID; }
        if (c.size[b] != ti.amb[sl]) { c.err = "block size does not match factor slot"; return BSGPU_ERR_INVALID; }
        if ((t == BSGPU_F_REPROJ || t == BSGPU_F_REPROJ_ONLINE_CALIB) && sl == 2) lm_use[b]++; else other_use[b]++;
      }
      if (has_camera(t)) {
        const int cam = idx[ti.nvar];
        if (cam < 0 || cam >= (int)c.cams.size()) { c.err = "camera index out of range"; return BSGPU_ERR_INVALID; }
      }
    }
  }
  for (const Marginal& mg : c.marginals)
    for (int b : mg.blocks) {
      if (b < 0 || b >= c.nb) { c.err = "marginal factor references block out of range"; return BSGPU_ERR_INVALID; }
      other_use[b]++;
    }
  // a parameter block no residual block touches is not part of the problem ([EXT] Ceres drops unused parameter blocks from the
  // reduced program; fuse's graph keeps e.g. landmarks whose last observation left the window): treated like a constant block
  if (c.is_const_in.size() != (size_t)c.nb) c.is_const_in = c.is_const;
  for (int b = 0; b < c.nb; ++b) c.is_const[b] = (c.is_const_in[b] || lm_use[b] + other_use[b] == 0) ? 1 : 0;
  c.tsize.assign(c.nb, 0); c.toff.assign(c.nb, -1); c.is_lm.assign(c.nb, 0);
  c.pose_blocks.clear(); c.lm_blocks.clear();
  for (int b = 0; b < c.nb; ++b) {
    c.tsize[b] = (c.manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : c.size[b];
    if (c.manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT && c.size[b] != 4) { c.err = "quaternion block must have size 4"; return BSGPU_ERR_INVALID; }
    if (c.is_const[b]) continue;
    if (lm_use[b] > 0 && other_use[b] == 0 && c.size[b] == 3 && c.manifold[b] == BSGPU_MANIFOLD_EUCLIDEAN) c.is_lm[b] = 1;
  }
  int to = 0;
  for (int b = 0; b < c.nb; ++b) if (!c.is_const[b] && !c.is_lm[b]) { c.toff[b] = to; to += c.tsize[b]; c.pose_blocks.push_back(b); }
  c.n_pose = to;
  for (int b = 0; b < c.nb; ++b) if (!c.is_const[b] && c.is_lm[b]) { c.toff[b] = to; to += 3; c.lm_blocks.push_back(b); }
  c.n_tan = to; c.n_lm = (int)c.lm_blocks.size();
  // groups
  int row = 0;
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    g.tw = 0;
    for (int sl = 0; sl < ti.nvar; ++sl) g.tw += (ti.amb[sl] == 4 ? 3 : ti.amb[sl]);
    g.row0 = row; row += g.n * ti.m;
    g.r.assign((size_t)g.n * ti.m, 0.0);
    g.J.assign((size_t)g.n * ti.m * g.tw, 0.0);
    g.active.assign(g.n, 0);
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sl = 0; sl < ti.nvar; ++sl) {
        if (c.manifold[idx[sl]] == BSGPU_MANIFOLD_EUCLIDEAN && ti.amb[sl] == 4) { c.err = "4-d slot must be a quaternion-manifold block"; return BSGPU_ERR_INVALID; }
        if (!c.is_const[idx[sl]]) g.active[f] = 1;
      }
    }
  }
  for (Marginal& mg : c.marginals) {
    int cols = 0, amb = 0;
    mg.col_t.clear(); mg.active = false;
    for (int b : mg.blocks) {
      for (int k = 0; k < c.tsize[b]; ++k) mg.col_t.push_back(c.is_const[b] ? -1 : c.toff[b] + k);
      cols += c.tsize[b]; amb += c.size[b];
      if (!c.is_const[b]) mg.active = true;
    }
    if (cols != mg.cols || amb != (int)mg.xbar.size()) { c.err = "marginal factor: A / xbar sizes do not match its blocks"; return BSGPU_ERR_INVALID; }
    mg.row0 = row; row += mg.rows;
    mg.r.assign(mg.rows, 0.0); mg.J.assign((size_t)mg.rows * mg.cols, 0.0);
  }
  c.num_res = row;
  // landmark -> factors CSR, owner (q,p pair) -> factors CSR
  std::vector<int> lm_index(c.nb, -1);
  for (int i = 0; i < c.n_lm; ++i) lm_index[c.lm_blocks[i]] = i;
  c.lm_fac_start.assign(c.n_lm + 1, 0);
  for (int t = 0; t < 2; ++t) {
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) { int l = lm_index[g.idx[(size_t)f * ti.nidx + 2]]; if (l >= 0) c.lm_fac_start[l + 1]++; }
  }
  for (int i = 0; i < c.n_lm; ++i) c.lm_fac_start[i + 1] += c.lm_fac_start[i];
  c.lm_fac.assign(c.lm_fac_start[c.n_lm], 0);
  {
    std::vector<int> fill(c.lm_fac_start.begin(), c.lm_fac_start.end() - 1);
    for (int t = 0; t < 2; ++t) {
      Group& g = c.groups[t];
      const TypeInfo& ti = kTypes[t];
      for (int f = 0; f < g.n; ++f) { int l = lm_index[g.idx[(size_t)f * ti.nidx + 2]]; if (l >= 0) c.lm_fac[fill[l]++] = (t << 28) | f; }
    }
  }
  // owners: unique (q,p) pairs among landmark factors
  {
    std::vector<std::pair<std::pair<int, int>, int>> keys;
    for (int t = 0; t < 2; ++t) {
      Group& g = c.groups[t];
      const TypeInfo& ti = kTypes[t];
      for (int f = 0; f < g.n; ++f) keys.push_back({{g.idx[(size_t)f * ti.nidx], g.idx[(size_t)f * ti.nidx + 1]}, (t << 28) | f});
    }
    std::sort(keys.begin(), keys.end());
    c.owner_start.clear(); c.owner_fac.clear();
    std::vector<int> blk_owner(c.nb, -1);
    c.owner_parallel_ok = true;
    int o = -1;
    for (size_t i = 0; i < keys.size(); ++i) {
      if (i == 0 || keys[i].first != keys[i - 1].first) {
        ++o; c.owner_start.push_back((int)c.owner_fac.size());
        for (int b : {keys[i].first.first, keys[i].first.second}) {
          if (blk_owner[b] != -1 && blk_owner[b] != o) c.owner_parallel_ok = false;
          blk_owner[b] = o;
        }
      }
      c.owner_fac.push_back(keys[i].second);
    }
    c.n_owner = o + 1;
    c.owner_start.push_back((int)c.owner_fac.size());
    // online-calib factors with non-constant extrinsics would write rows outside their owner
    Group& g1 = c.groups[1];
    for (int f = 0; f < g1.n; ++f)
      if (!c.is_const[g1.idx[(size_t)f * 6 + 3]] || !c.is_const[g1.idx[(size_t)f * 6 + 4]]) c.owner_parallel_ok = false;
    // a landmark block that is not eliminated (shared with another kind of factor) is a pose-side row written by
    // several owners
    for (int t = 0; t < 2; ++t) {
      Group& g = c.groups[t];
      const TypeInfo& ti = kTypes[t];
      for (int f = 0; f < g.n; ++f) {
        const int b = g.idx[(size_t)f * ti.nidx + 2];
        if (!c.is_const[b] && !c.is_lm[b]) c.owner_parallel_ok = false;
      }
    }
  }
  c.finalized = true;
  return BSGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// dense blocked Cholesky (lower, in place, row-major n x n) and solve
// ------------------------------------------------------------------------------------------------
static bool cholesky_lower(double* A, int n) {
  const int NB = 64;
  bool ok = true;
  for (int k = 0; k < n && ok; k += NB) {
    const int kb = std::min(NB, n - k);
    // factor diagonal block
    for (int j = k; j < k + kb; ++j) {
      double d = A[(size_t)j * n + j];
      for (int p = k; p < j; ++p) d -= A[(size_t)j * n + p] * A[(size_t)j * n + p];
      if (!(d > 0.0) || !std::isfinite(d)) { ok = false; break; }
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k + kb; ++i) {
        double s = A[(size_t)i * n + j];
        for (int p = k; p < j; ++p) s -= A[(size_t)i * n + p] * A[(size_t)j * n + p];
        A[(size_t)i * n + j] = s / d;
      }
    }
    if (!ok) break;
    // panel: rows below, solve X L11^T = A21
#pragma omp parallel for schedule(static)
    for (int i = k + kb; i < n; ++i) {
      double* Ai = A + (size_t)i * n;
      for (int j = k; j < k + kb; ++j) {
        double s = Ai[j];
        const double* Aj = A + (size_t)j * n;
        for (int p = k; p < j; ++p) s -= Ai[p] * Aj[p];
        Ai[j] = s / Aj[j];
      }
    }
    // trailing update (lower only): A22 -= L21 L21^T
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = k + kb; i < n; ++i) {
      double* Ai = A + (size_t)i * n;
      const double* Li = Ai + k;
      for (int j = k + kb; j <= i; ++j) {
        const double* Lj = A + (size_t)j * n + k;
        double s = 0;
        for (int p = 0; p < kb; ++p) s += Li[p] * Lj[p];
        Ai[j] -= s;
      }
    }
  }
  return ok;
}
static void cholesky_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double* Li = L + (size_t)i * n;
    for (int p = 0; p < i; ++p) s -= Li[p] * b[p];
    b[i] = s / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int p = i + 1; p < n; ++p) s -= L[(size_t)p * n + i] * b[p];
    b[i] = s / L[(size_t)i * n + i];
  }
}

static inline bool inv3_sym(const double* H, double* Hi) {
  const double a = H[0], b = H[1], cc = H[2], d = H[4], e = H[5], f = H[8];
  const double c00 = d * f - e * e, c01 = cc * e - b * f, c02 = b * e - cc * d;
  const double det = a * c00 + b * c01 + cc * c02;
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
  Hi[3] = Hi[1]; Hi[4] = (a * f - cc * cc) * id; Hi[5] = (b * cc - a * e) * id;
  Hi[6] = Hi[2]; Hi[7] = Hi[5]; Hi[8] = (a * d - b * b) * id;
  return true;
}

// tangent column offsets of a factor's slots
static inline void factor_cols(const Ctx& c, int type, const int32_t* idx, int* cols /* per slot toff or -1 */) {
  const TypeInfo& ti = kTypes[type];
  for (int sl = 0; sl < ti.nvar; ++sl) cols[sl] = c.toff[idx[sl]];
}

struct LinSys {
  std::vector<double> S;       // n_pose x n_pose
  std::vector<double> bp;      // n_pose  (J^T r, pose part) then reduced rhs
  std::vector<double> Hll;     // n_lm x 9
  std::vector<double> bl;      // n_lm x 3
  std::vector<double> Hll_inv; // n_lm x 9
};

// Solve (J^T J + D^2) y = J^T r with J, r the (scaled, robustified) values stored in the groups.
// D2 = squared LM diagonal (n_tan).  Exact: landmark Schur complement + dense Cholesky.
static bool solve_normal(Ctx& c, const double* D2, double* y, LinSys& ls) {
  const int np = c.n_pose, nl = c.n_lm;
  ls.S.assign((size_t)np * np, 0.0);
  ls.bp.assign(np, 0.0);
  ls.Hll.assign((size_t)nl * 9, 0.0);
  ls.bl.assign((size_t)nl * 3, 0.0);
  ls.Hll_inv.assign((size_t)nl * 9, 0.0);
  double* S = ls.S.data();
  // ---- pose-only factor groups (serial: few / small) and landmark factors' pose part by owner
  for (int t = 2; t < BSGPU_F_NUM_TYPES; ++t) {
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    const int m = ti.m, tw = g.tw;
    for (int f = 0; f < g.n; ++f) {
      if (!g.active[f]) continue;
      int cols[10];
      factor_cols(c, t, &g.idx[(size_t)f * ti.nidx], cols);
      const double* J = &g.J[(size_t)f * m * tw];
      const double* r = &g.r[(size_t)f * m];
      for (int sa = 0; sa < ti.nvar; ++sa) {
        if (cols[sa] < 0) continue;
        for (int ia = 0; ia < tsz(ti, sa); ++ia) {
          const int ca = sa * 3 + ia;
          double gsum = 0;
          for (int k = 0; k < m; ++k) gsum += J[k * tw + ca] * r[k];
          ls.bp[cols[sa] + ia] += gsum;
          for (int sb = 0; sb < ti.nvar; ++sb) {
            if (cols[sb] < 0) continue;
            for (int ib = 0; ib < tsz(ti, sb); ++ib) {
              const int cb = sb * 3 + ib;
              double s = 0;
              for (int k = 0; k < m; ++k) s += J[k * tw + ca] * J[k * tw + cb];
              S[(size_t)(cols[sa] + ia) * np + cols[sb] + ib] += s;
            }
          }
        }
      }
    }
  }
  for (const Marginal& mg : c.marginals) {
    if (!mg.active) continue;
    for (int a = 0; a < mg.cols; ++a) {
      const int ta = mg.col_t[a];
      if (ta < 0) continue;
      double gsum = 0;
      for (int k = 0; k < mg.rows; ++k) gsum += mg.J[(size_t)k * mg.cols + a] * mg.r[k];
      ls.bp[ta] += gsum;
      for (int b = 0; b < mg.cols; ++b) {
        const int tb = mg.col_t[b];
        if (tb < 0) continue;
        double sacc = 0;
        for (int k = 0; k < mg.rows; ++k) sacc += mg.J[(size_t)k * mg.cols + a] * mg.J[(size_t)k * mg.cols + b];
        S[(size_t)ta * np + tb] += sacc;
      }
    }
  }
  auto lmfac = [&](int code, int& t, int& f) { t = code >> 28; f = code & ((1 << 28) - 1); };
  // landmark blocks: Hll, bl
#pragma omp parallel for schedule(static) if (nl > 256)
  for (int l = 0; l < nl; ++l) {
    double H[9] = {0}, b[3] = {0};
    for (int e = c.lm_fac_start[l]; e < c.lm_fac_start[l + 1]; ++e) {
      int t, f; lmfac(c.lm_fac[e], t, f);
      Group& g = c.groups[t];
      const int tw = g.tw;
      const double* J = &g.J[(size_t)f * 2 * tw];
      const double* r = &g.r[(size_t)f * 2];
      for (int i = 0; i < 3; ++i) {
        b[i] += J[6 + i] * r[0] + J[tw + 6 + i] * r[1];
        for (int j = 0; j < 3; ++j) H[3 * i + j] += J[6 + i] * J[6 + j] + J[tw + 6 + i] * J[tw + 6 + j];
      }
    }
    const int to = c.n_pose + 3 * l;
    for (int i = 0; i < 3; ++i) H[4 * i] += D2[to + i];
    std::memcpy(&ls.Hll[(size_t)l * 9], H, sizeof(H));
    std::memcpy(&ls.bl[(size_t)l * 3], b, sizeof(b));
    inv3_sym(H, &ls.Hll_inv[(size_t)l * 9]);
  }
  std::vector<int> lm_index(c.nb, -1);
  for (int i = 0; i < nl; ++i) lm_index[c.lm_blocks[i]] = i;
  // pose part of landmark factors + Schur complement, row-owned by (q,p) pair
  auto pose_cols_of = [&](int t, int f, int* cols, int* jc) {
    // cols: tangent offsets of pose-side slots, jc: column offset inside J for each
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
    int n = 0, ct = 0;
    for (int sl = 0; sl < ti.nvar; ++sl) {
      // slot 2 (the landmark) counts as a pose-side slot when its block is not eliminated
      if (sl != 2 || (c.toff[idx[2]] >= 0 && c.toff[idx[2]] < c.n_pose)) { cols[n] = c.toff[idx[sl]]; jc[n] = ct; ++n; }
      ct += 3;
    }
    return n;
  };
#pragma omp parallel for schedule(dynamic, 1) if (c.owner_parallel_ok && c.n_owner > 1)
  for (int o = 0; o < c.n_owner; ++o) {
    for (int e = c.owner_start[o]; e < c.owner_start[o + 1]; ++e) {
      int t, f; lmfac(c.owner_fac[e], t, f);
      Group& g = c.groups[t];
      if (!g.active[f]) continue;
      const TypeInfo& ti = kTypes[t];
      const int tw = g.tw;
      const double* J = &g.J[(size_t)f * 2 * tw];
      const double* r = &g.r[(size_t)f * 2];
      int cols[5], jc[5];
      const int ns = pose_cols_of(t, f, cols, jc);
      // J_p^T J_p and J_p^T r
      for (int sa = 0; sa < ns; ++sa) {
        if (cols[sa] < 0) continue;
        for (int ia = 0; ia < 3; ++ia) {
          const int ca = jc[sa] + ia;
          ls.bp[cols[sa] + ia] += J[ca] * r[0] + J[tw + ca] * r[1];
          for (int sb = 0; sb < ns; ++sb) {
            if (cols[sb] < 0) continue;
            for (int ib = 0; ib < 3; ++ib) {
              const int cb = jc[sb] + ib;
              S[(size_t)(cols[sa] + ia) * np + cols[sb] + ib] += J[ca] * J[cb] + J[tw + ca] * J[tw + cb];
            }
          }
        }
      }
      const int l = lm_index[g.idx[(size_t)f * ti.nidx + 2]];
      if (l < 0) continue;  // constant landmark: no elimination
      // W = Hpl Hll^-1, Hpl(3ns x 3) = J_p^T J_l
      const double* Hi = &ls.Hll_inv[(size_t)l * 9];
      double W[15][3];
      for (int sa = 0; sa < ns; ++sa)
        for (int ia = 0; ia < 3; ++ia) {
          const int ca = jc[sa] + ia;
          double h[3];
          for (int j = 0; j < 3; ++j) h[j] = J[ca] * J[6 + j] + J[tw + ca] * J[tw + 6 + j];
          for (int j = 0; j < 3; ++j) W[sa * 3 + ia][j] = h[0] * Hi[j] + h[1] * Hi[3 + j] + h[2] * Hi[6 + j];
        }
      // rhs: bp -= W bl
      const double* bl = &ls.bl[(size_t)l * 3];
      for (int sa = 0; sa < ns; ++sa) {
        if (cols[sa] < 0) continue;
        for (int ia = 0; ia < 3; ++ia)
          ls.bp[cols[sa] + ia] -= W[sa * 3 + ia][0] * bl[0] + W[sa * 3 + ia][1] * bl[1] + W[sa * 3 + ia][2] * bl[2];
      }
      // S[rows(f), cols(f')] -= W Hpl_f'^T
      for (int e2 = c.lm_fac_start[l]; e2 < c.lm_fac_start[l + 1]; ++e2) {
        int t2, f2; lmfac(c.lm_fac[e2], t2, f2);
        Group& g2 = c.groups[t2];
        const int tw2 = g2.tw;
        const double* J2 = &g2.J[(size_t)f2 * 2 * tw2];
        int cols2[5], jc2[5];
        const int ns2 = pose_cols_of(t2, f2, cols2, jc2);
        for (int sb = 0; sb < ns2; ++sb) {
          if (cols2[sb] < 0) continue;
          for (int ib = 0; ib < 3; ++ib) {
            const int cb = jc2[sb] + ib;
            double h2[3];
            for (int j = 0; j < 3; ++j) h2[j] = J2[cb] * J2[6 + j] + J2[tw2 + cb] * J2[tw2 + 6 + j];
            for (int sa = 0; sa < ns; ++sa) {
              if (cols[sa] < 0) continue;
              for (int ia = 0; ia < 3; ++ia)
                S[(size_t)(cols[sa] + ia) * np + cols2[sb] + ib] -=
                    W[sa * 3 + ia][0] * h2[0] + W[sa * 3 + ia][1] * h2[1] + W[sa * 3 + ia][2] * h2[2];
            }
          }
        }
      }
    }
  }
  for (int i = 0; i < np; ++i) S[(size_t)i * np + i] += D2[i];
  if (np > 0) {
    if (!cholesky_lower(S, np)) return false;
    for (int i = 0; i < np; ++i) y[i] = ls.bp[i];
    cholesky_solve(S, np, y);
  }
  // back-substitution: y_l = Hll^-1 (bl - Hpl^T y_p)
#pragma omp parallel for schedule(static) if (nl > 256)
  for (int l = 0; l < nl; ++l) {
    double rhs[3] = {ls.bl[(size_t)l * 3], ls.bl[(size_t)l * 3 + 1], ls.bl[(size_t)l * 3 + 2]};
    for (int e = c.lm_fac_start[l]; e < c.lm_fac_start[l + 1]; ++e) {
      int t, f; lmfac(c.lm_fac[e], t, f);
      Group& g = c.groups[t];
      const int tw = g.tw;
      const double* J = &g.J[(size_t)f * 2 * tw];
      int cols[5], jc[5];
      const int ns = pose_cols_of(t, f, cols, jc);
      // (J_p y_p) then J_l^T (.)
      double jy[2] = {0, 0};
      for (int sa = 0; sa < ns; ++sa) {
        if (cols[sa] < 0) continue;
        for (int ia = 0; ia < 3; ++ia) {
          jy[0] += J[jc[sa] + ia] * y[cols[sa] + ia];
          jy[1] += J[tw + jc[sa] + ia] * y[cols[sa] + ia];
        }
      }
      for (int j = 0; j < 3; ++j) rhs[j] -= J[6 + j] * jy[0] + J[tw + 6 + j] * jy[1];
    }
    const double* Hi = &ls.Hll_inv[(size_t)l * 9];
    for (int j = 0; j < 3; ++j) y[np + 3 * l + j] = Hi[3 * j] * rhs[0] + Hi[3 * j + 1] * rhs[1] + Hi[3 * j + 2] * rhs[2];
  }
  for (int i = 0; i < c.n_tan; ++i) if (!std::isfinite(y[i])) return false;
  return true;
}

// The same system for a pose-only graph too large for the dense factorisation (BASELINE config 4: 30 000 tangent dimensions), solved
// to `tol` (relative residual) by conjugate gradients on the operator v -> J^T (J v) + D^2 v applied factor by factor, preconditioned with
// the inverse diagonal.  Test infrastructure for the block-sparse PCG path of the HIP library (tests/test_gpu_fullsize.py): the step of
// the reference's exact SPARSE_NORMAL_CHOLESKY (submap_pose_graph_optimization.cpp:144-146) to the accuracy of the tolerance.
static bool solve_normal_cg(Ctx& c, const double* D2, double* y, double tol, int max_it, int* iters_out) {
  const int n = c.n_tan;
  if (c.n_lm > 0 || !c.marginals.empty()) return false;
  std::vector<double> b(n, 0.0), diag(n, 0.0), r(n), z(n), p(n), q(n);
  struct Fac { int t, f; };
  std::vector<Fac> facs;
  for (int t = 2; t < BSGPU_F_NUM_TYPES; ++t) for (int f = 0; f < c.groups[t].n; ++f) if (c.groups[t].active[f]) facs.push_back({t, f});
  for (const Fac& fc : facs) {
    Group& g = c.groups[fc.t];
    const TypeInfo& ti = kTypes[fc.t];
    const int m = ti.m, tw = g.tw;
    int cols[10];
    factor_cols(c, fc.t, &g.idx[(size_t)fc.f * ti.nidx], cols);
    const double* J = &g.J[(size_t)fc.f * m * tw];
    const double* rr = &g.r[(size_t)fc.f * m];
    for (int sa = 0; sa < ti.nvar; ++sa) {
      if (cols[sa] < 0) continue;
      for (int ia = 0; ia < tsz(ti, sa); ++ia) {
        double gs = 0, hs = 0;
        for (int k = 0; k < m; ++k) { const double j = J[k * tw + sa * 3 + ia]; gs += j * rr[k]; hs += j * j; }
        b[cols[sa] + ia] += gs; diag[cols[sa] + ia] += hs;
      }
    }
  }
  for (int i = 0; i < n; ++i) diag[i] += D2[i];
  auto apply = [&](const std::vector<double>& v, std::vector<double>& out) {
    for (int i = 0; i < n; ++i) out[i] = D2[i] * v[i];
    for (const Fac& fc : facs) {
      Group& g = c.groups[fc.t];
      const TypeInfo& ti = kTypes[fc.t];
      const int m = ti.m, tw = g.tw;
      int cols[10];
      factor_cols(c, fc.t, &g.idx[(size_t)fc.f * ti.nidx], cols);
      const double* J = &g.J[(size_t)fc.f * m * tw];
      double jv[16];
      for (int k = 0; k < m; ++k) {
        double a = 0;
        for (int sa = 0; sa < ti.nvar; ++sa) if (cols[sa] >= 0) for (int ia = 0; ia < tsz(ti, sa); ++ia) a += J[k * tw + sa * 3 + ia] * v[cols[sa] + ia];
        jv[k] = a;
      }
      for (int sa = 0; sa < ti.nvar; ++sa) if (cols[sa] >= 0) for (int ia = 0; ia < tsz(ti, sa); ++ia) {
        double a = 0;
        for (int k = 0; k < m; ++k) a += J[k * tw + sa * 3 + ia] * jv[k];
        out[cols[sa] + ia] += a;
      }
    }
  };
  std::fill(y, y + n, 0.0);
  r = b;
  double bb = 0;
  for (int i = 0; i < n; ++i) bb += b[i] * b[i];
  if (bb == 0.0) { if (iters_out) *iters_out = 0; return true; }
  double rz = 0;
  for (int i = 0; i < n; ++i) { z[i] = diag[i] > 0 ? r[i] / diag[i] : r[i]; rz += r[i] * z[i]; }
  p = z;
  int it = 0;
  for (; it < max_it; ++it) {
    apply(p, q);
    double pq = 0;
    for (int i = 0; i < n; ++i) pq += p[i] * q[i];
    if (!(pq > 0)) break;
    const double alpha = rz / pq;
    double rr2 = 0;
    for (int i = 0; i < n; ++i) { y[i] += alpha * p[i]; r[i] -= alpha * q[i]; rr2 += r[i] * r[i]; }
    if (rr2 <= tol * tol * bb) { ++it; break; }
    double rz_new = 0;
    for (int i = 0; i < n; ++i) { z[i] = diag[i] > 0 ? r[i] / diag[i] : r[i]; rz_new += r[i] * z[i]; }
    const double beta = rz_new / rz;
    rz = rz_new;
    for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
  }
  if (iters_out) *iters_out = it;
  for (int i = 0; i < n; ++i) if (!std::isfinite(y[i])) return false;
  return true;
}

// helpers over all factors: column squared norms, gradient, J*v
template <typename Fn> static void for_each_factor(Ctx& c, Fn fn) {
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    Group& g = c.groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      if (!g.active[f]) continue;
      int cols[10];
      factor_cols(c, t, &g.idx[(size_t)f * ti.nidx], cols);
      fn(g, ti, f, cols);
    }
  }
}

static void gradient_of(Ctx& c, double* grad) {
  std::fill(grad, grad + c.n_tan, 0.0);
  for_each_factor(c, [&](Group& g, const TypeInfo& ti, int f, const int* cols) {
    const int m = ti.m, tw = g.tw;
    const double* J = &g.J[(size_t)f * m * tw];
    const double* r = &g.r[(size_t)f * m];
    for (int sl = 0; sl < ti.nvar; ++sl) {
      if (cols[sl] < 0) continue;
      for (int i = 0; i < tsz(ti, sl); ++i) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += J[k * tw + 3 * sl + i] * r[k];
        grad[cols[sl] + i] += s;
      }
    }
  });
  for (const Marginal& mg : c.marginals)
    if (mg.active) for (int a = 0; a < mg.cols; ++a) if (mg.col_t[a] >= 0) {
      double s = 0;
      for (int k = 0; k < mg.rows; ++k) s += mg.J[(size_t)k * mg.cols + a] * mg.r[k];
      grad[mg.col_t[a]] += s;
    }
}
static void colnorm2_of(Ctx& c, double* n2) {
  std::fill(n2, n2 + c.n_tan, 0.0);
  for_each_factor(c, [&](Group& g, const TypeInfo& ti, int f, const int* cols) {
    const int m = ti.m, tw = g.tw;
    const double* J = &g.J[(size_t)f * m * tw];
    for (int sl = 0; sl < ti.nvar; ++sl) {
      if (cols[sl] < 0) continue;
      for (int i = 0; i < tsz(ti, sl); ++i) {
        double s = 0;
        for (int k = 0; k < m; ++k) s += J[k * tw + 3 * sl + i] * J[k * tw + 3 * sl + i];
        n2[cols[sl] + i] += s;
      }
    }
  });
  for (const Marginal& mg : c.marginals)
    if (mg.active) for (int a = 0; a < mg.cols; ++a) if (mg.col_t[a] >= 0) {
      double s = 0;
      for (int k = 0; k < mg.rows; ++k) s += mg.J[(size_t)k * mg.cols + a] * mg.J[(size_t)k * mg.cols + a];
      n2[mg.col_t[a]] += s;
    }
}
static void scale_columns(Ctx& c, const double* sc) {
  for_each_factor(c, [&](Group& g, const TypeInfo& ti, int f, const int* cols) {
    const int m = ti.m, tw = g.tw;
    double* J = &g.J[(size_t)f * m * tw];
    for (int sl = 0; sl < ti.nvar; ++sl) {
      if (cols[sl] < 0) continue;
      for (int i = 0; i < tsz(ti, sl); ++i)
        for (int k = 0; k < m; ++k) J[k * tw + 3 * sl + i] *= sc[cols[sl] + i];
    }
  });
  for (Marginal& mg : c.marginals)
    if (mg.active) for (int a = 0; a < mg.cols; ++a) if (mg.col_t[a] >= 0)
      for (int k = 0; k < mg.rows; ++k) mg.J[(size_t)k * mg.cols + a] *= sc[mg.col_t[a]];
}
// returns -(J v).(r + J v / 2)
static double model_cost_change_of(Ctx& c, const double* v) {
  double acc = 0;
  for_each_factor(c, [&](Group& g, const TypeInfo& ti, int f, const int* cols) {
    const int m = ti.m, tw = g.tw;
    const double* J = &g.J[(size_t)f * m * tw];
    const double* r = &g.r[(size_t)f * m];
    for (int k = 0; k < m; ++k) {
      double jv = 0;
      for (int sl = 0; sl < ti.nvar; ++sl) {
        if (cols[sl] < 0) continue;
        for (int i = 0; i < tsz(ti, sl); ++i) jv += J[k * tw + 3 * sl + i] * v[cols[sl] + i];
      }
      acc -= jv * (r[k] + jv / 2.0);
    }
  });
  for (const Marginal& mg : c.marginals)
    if (mg.active) for (int k = 0; k < mg.rows; ++k) {
      double jv = 0;
      for (int a = 0; a < mg.cols; ++a) if (mg.col_t[a] >= 0) jv += mg.J[(size_t)k * mg.cols + a] * v[mg.col_t[a]];
      acc -= jv * (mg.r[k] + jv / 2.0);
    }
  return acc;
}

static void plus_all(const Ctx& c, const double* x, const double* delta, double* out) {
  std::memcpy(out, x, sizeof(double) * c.x.size());
  for (int b = 0; b < c.nb; ++b) {
    if (c.toff[b] < 0) continue;
    manifold_plus(c.manifold[b], c.size[b], x + c.off[b], delta + c.toff[b], out + c.off[b]);
  }
}
static double active_norm(const Ctx& c, const double* x) {
  double s = 0;
  for (int b = 0; b < c.nb; ++b) if (c.toff[b] >= 0) for (int i = 0; i < c.size[b]; ++i) s += x[c.off[b] + i] * x[c.off[b] + i];
  return std::sqrt(s);
}
static void active_diff_norms(const Ctx& c, const double* x, const double* y, double* l2, double* linf) {
  double s = 0, mx = 0;
  for (int b = 0; b < c.nb; ++b)
    if (c.toff[b] >= 0)
      for (int i = 0; i < c.size[b]; ++i) { const double d = x[c.off[b] + i] - y[c.off[b] + i]; s += d * d; mx = std::max(mx, std::fabs(d)); }
  *l2 = std::sqrt(s); *linf = mx;
}

// ------------------------------------------------------------------------------------------------
// [EXT] ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy restated
// (called through fixed_lag_smoother.cpp:281)
// ------------------------------------------------------------------------------------------------
static int solve(Ctx& c, const bsgpu_options& o, bsgpu_summary& sum) {
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clk::now() - t_start).count(); };
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  std::memset(&sum, 0, sizeof(sum));
  c.iters.clear();
  sum.num_parameters_tangent = c.n_tan;
  sum.num_residuals = c.num_res;
  sum.linear_solver_used = BSGPU_LINEAR_SCHUR_CHOLESKY;
  const int n = c.n_tan;
  const bool cg_ok = o.linear_solver_type == BSGPU_LINEAR_PCG && c.n_lm == 0 && c.marginals.empty();
  if ((size_t)c.n_pose > 20000 && !cg_ok) { c.err = "oracle: reduced system too large for the dense exact path"; return BSGPU_ERR_UNSUPPORTED; }
  std::vector<double> x = c.x, cand(c.x.size()), grad(n), neg(n), scale(n, 1.0), diag(n), D2(n), step(n), delta(n), tmp(c.x.size());
  LinSys ls;
  double fixed = 0;
  double x_cost = evaluate(c, x.data(), true, &fixed);
  gradient_of(c, grad.data());
  if (!std::isfinite(x_cost)) { sum.termination_type = BSGPU_FAILURE; std::snprintf(sum.message, sizeof(sum.message), "initial cost not finite"); return BSGPU_OK; }
  if (o.jacobi_scaling) {
    colnorm2_of(c, scale.data());
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(scale[i]));
    scale_columns(c, scale.data());
  }
  auto gradient_norms = [&](double* gmax, double* gnorm) {
    for (int i = 0; i < n; ++i) neg[i] = -grad[i];
    plus_all(c, x.data(), neg.data(), tmp.data());
    active_diff_norms(c, x.data(), tmp.data(), gnorm, gmax);
  };
  double x_norm = active_norm(c, x.data());
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int num_consecutive_invalid = 0;
  bsgpu_iteration it;
  std::memset(&it, 0, sizeof(it));
  it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost + fixed;
  gradient_norms(&it.gradient_max_norm, &it.gradient_norm);
  sum.initial_cost = x_cost + fixed;
  sum.fixed_cost = fixed;
  double minimum_cost = x_cost;
  std::vector<double> best = x;
  sum.termination_type = BSGPU_NO_CONVERGENCE;
  const char* msg = "";
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) {
      if (it.iteration > 0) sum.num_successful_steps++;
      if (x_cost <= minimum_cost) { minimum_cost = x_cost; best = x; }
    } else sum.num_unsuccessful_steps++;
    it.trust_region_radius = radius;
    c.iters.push_back(it);
    if (o.max_solver_time_in_seconds > 0 && elapsed() >= o.max_solver_time_in_seconds) { sum.termination_type = BSGPU_NO_CONVERGENCE; msg = "Maximum solver time reached."; break; }
    if (it.iteration >= o.max_num_iterations) { sum.termination_type = BSGPU_NO_CONVERGENCE; msg = "Maximum number of iterations reached."; break; }
    if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
    if (radius <= o.min_trust_region_radius) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
    bsgpu_iteration prev = it;
    std::memset(&it, 0, sizeof(it));
    it.iteration = prev.iteration + 1;
    it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
    // ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
    if (!reuse_diagonal) {
      colnorm2_of(c, diag.data());
      for (int i = 0; i < n; ++i) diag[i] = std::min(std::max(diag[i], o.min_lm_diagonal), o.max_lm_diagonal);
    }
    for (int i = 0; i < n; ++i) D2[i] = diag[i] / radius;  // lm_diagonal^2
    // (BSGPU_LINEAR_PCG on a pose-only graph: the exact step by conjugate gradients to pcg_tolerance — what a 30 000-dimensional
    // pose graph can be checked against; everything else: landmark Schur complement + dense Cholesky)
    const bool by_cg = o.linear_solver_type == BSGPU_LINEAR_PCG && c.n_lm == 0 && c.marginals.empty();
    int cg_it = 0;
    bool lin_ok = by_cg ? solve_normal_cg(c, D2.data(), step.data(), o.pcg_tolerance > 0 ? o.pcg_tolerance : 1e-12,
                                          o.pcg_max_iterations > 0 ? o.pcg_max_iterations : 20000, &cg_it)
                        : solve_normal(c, D2.data(), step.data(), ls);
    if (by_cg) { sum.num_inner_iterations += cg_it; sum.linear_solver_used = BSGPU_LINEAR_PCG; }
    sum.num_linear_solves++;
    reuse_diagonal = true;
    double model_cost_change = 0;
    if (lin_ok) {
      for (int i = 0; i < n; ++i) step[i] = -step[i];
      model_cost_change = model_cost_change_of(c, step.data());
      it.step_is_valid = model_cost_change > 0.0;
    }
    it.model_cost_change = model_cost_change;
    if (!it.step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
        sum.termination_type = BSGPU_FAILURE; msg = "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps.";
        break;
      }
      // [EXT] LevenbergMarquardtStrategy::StepIsInvalid(): radius_ *= 0.5, decrease_factor_ untouched (only StepRejected grows it);
      // the diagonal is recomputed — from the same Jacobian, i.e. to the same values
      radius *= 0.5; reuse_diagonal = true;
      it.cost = x_cost + fixed; it.step_is_successful = 0;
      continue;
    }
    num_consecutive_invalid = 0;
    for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
    plus_all(c, x.data(), delta.data(), cand.data());
    double cand_cost = evaluate(c, cand.data(), false, nullptr);
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    // ParameterToleranceReached
    double dl2, dinf;
    active_diff_norms(c, x.data(), cand.data(), &dl2, &dinf);
    it.step_norm = dl2;
    if (dl2 <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
      sum.termination_type = BSGPU_CONVERGENCE; msg = "Parameter tolerance reached."; break;
    }
    // FunctionToleranceReached
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= o.function_tolerance * x_cost) {
      sum.termination_type = BSGPU_CONVERGENCE; msg = "Function tolerance reached."; break;
    }
    it.relative_decrease = (x_cost - cand_cost) / model_cost_change;
    if (it.relative_decrease > o.min_relative_decrease) {
      // HandleSuccessfulStep
      x = cand; x_norm = active_norm(c, x.data());
      x_cost = evaluate(c, x.data(), true, nullptr);
      gradient_of(c, grad.data());
      if (o.jacobi_scaling) scale_columns(c, scale.data());
      gradient_norms(&it.gradient_max_norm, &it.gradient_norm);
      it.cost = x_cost + fixed; it.step_is_successful = 1;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(o.max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      // HandleUnsuccessfulStep
      it.step_is_successful = 0;
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      it.cost = cand_cost + fixed;
    }
  }
  c.x = best;
  sum.final_cost = minimum_cost + fixed;
  sum.num_iterations = (int)c.iters.size() - 1;
  sum.is_solution_usable = (sum.termination_type == BSGPU_CONVERGENCE || sum.termination_type == BSGPU_NO_CONVERGENCE) ? 1 : 0;
  sum.total_time_in_seconds = elapsed();
  std::snprintf(sum.message, sizeof(sum.message), "%s", msg);
  return BSGPU_OK;
}

}  // namespace bso

// =================================================================================================
// C entry points (same signatures as include/bsgpu.h, prefix bso_)
// =================================================================================================
using bso::Ctx;
extern "C" {

int bso_nidx(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? bso::kTypes[t].nidx : -1; }
int bso_nconst(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? bso::kTypes[t].nconst : -1; }
int bso_nres(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? bso::kTypes[t].m : -1; }

void bso_options_default(bsgpu_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50; o->linear_solver_type = BSGPU_LINEAR_AUTO; o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5; o->max_solver_time_in_seconds = 1e9;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->pcg_max_iterations = 500; o->pcg_tolerance = 1e-10;
}
void bso_options_vio(bsgpu_options* o) {  // beam_slam_launch/config/vio.yaml:7-17
  bso_options_default(o);
  o->max_num_iterations = 10; o->max_solver_time_in_seconds = 0.05;
  o->gradient_tolerance = 1.5e-7; o->parameter_tolerance = 1.5e-7; o->function_tolerance = 1.5e-7;
}

Ctx* bso_create(int) { return new Ctx(); }
void bso_destroy(Ctx* c) { delete c; }
const char* bso_last_error(const Ctx* c) { return c->err.c_str(); }
int bso_abi_version(void) { return BSGPU_ABI_VERSION; }
// reproj_mode is a setting of the checker, not problem data: it survives clear() (a load() starts with clear(); a test that set the
// mode first used to compare the closed form with itself)
int bso_clear(Ctx* c) { auto keep = std::move(c->sync_prev); const int mode = c->reproj_mode; *c = Ctx(); c->sync_prev = std::move(keep); c->reproj_mode = mode; return BSGPU_OK; }

int bso_set_blocks(Ctx* c, int32_t n, const double* values, const int32_t* offset, const uint8_t* size,
                   const uint8_t* manifold, const uint8_t* is_const) {
  c->nb = n;
  c->off.assign(offset, offset + n); c->size.assign(size, size + n);
  c->manifold.assign(manifold, manifold + n); c->is_const.assign(is_const, is_const + n); c->is_const_in = c->is_const;
  size_t tot = 0;
  for (int i = 0; i < n; ++i) tot = std::max(tot, (size_t)offset[i] + size[i]);
  c->x.assign(values, values + tot); c->x0 = c->x;
  c->finalized = false;
  return BSGPU_OK;
}
int bso_set_values(Ctx* c, const double* v, int64_t n) {
  if ((size_t)n != c->x.size()) { c->err = "set_values: size mismatch"; return BSGPU_ERR_INVALID; }
  c->x.assign(v, v + n); c->x0 = c->x; return BSGPU_OK;
}
int bso_set_cameras(Ctx* c, int32_t n, const bsgpu_camera* cams) {
  c->cams.resize(n);
  for (int i = 0; i < n; ++i) {
    c->cams[i].fx = cams[i].fx; c->cams[i].fy = cams[i].fy; c->cams[i].cx = cams[i].cx; c->cams[i].cy = cams[i].cy;
    std::memcpy(c->cams[i].R_cb, cams[i].R_cam_baselink, sizeof(double) * 9);
    std::memcpy(c->cams[i].t_cb, cams[i].t_cam_baselink, sizeof(double) * 3);
  }
  return BSGPU_OK;
}
int bso_add_factors(Ctx* c, int32_t type, int32_t n, const int32_t* idx, const double* consts,
                    const int32_t* loss_kind, const double* loss_a) {
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) { c->err = "unknown factor type"; return BSGPU_ERR_INVALID; }
  const bso::TypeInfo& ti = bso::kTypes[type];
  bso::Group& g = c->groups[type];
  g.type = type;
  g.idx.insert(g.idx.end(), idx, idx + (size_t)n * ti.nidx);
  g.consts.insert(g.consts.end(), consts, consts + (size_t)n * ti.nconst);
  for (int i = 0; i < n; ++i) { g.loss_kind.push_back(loss_kind ? loss_kind[i] : 0); g.loss_a.push_back(loss_a ? loss_a[i] : 1.0); }
  g.n += n;
  c->finalized = false;
  return BSGPU_OK;
}
int bso_add_factors_indirect(Ctx* c, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots, const int32_t* slot_to_block,
                             const double* consts, const int32_t* loss_kind, const double* loss_a) {
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) { c->err = "unknown factor type"; return BSGPU_ERR_INVALID; }
  const bso::TypeInfo& ti = bso::kTypes[type];
  std::vector<int32_t> idx(slot_idx, slot_idx + (size_t)n * ti.nidx);
  for (int f = 0; f < n; ++f)
    for (int k = 0; k < ti.nvar; ++k) {
      const int32_t s = idx[(size_t)f * ti.nidx + k];
      if (s < 0 || s >= n_slots || slot_to_block[s] < 0) { c->err = "add_factors_indirect: slot out of range or not mapped to a block"; return BSGPU_ERR_INVALID; }
      idx[(size_t)f * ti.nidx + k] = slot_to_block[s];
    }
  return bso_add_factors(c, type, n, idx.data(), consts, loss_kind, loss_a);
}
// bsgpu_sync_factors_indirect (include/bsgpu.h): the oracle has nothing to patch — it takes the whole table — but it CHECKS the
// caller's promise: every row not in the change list must equal the row of the previous call's table.
int bso_sync_factors_indirect(Ctx* c, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots, const int32_t* slot_to_block,
                              const double* consts, const int32_t* loss_kind, const double* loss_a, int32_t n_changed,
                              const int32_t* changed_rows) {
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) { c->err = "unknown factor type"; return BSGPU_ERR_INVALID; }
  if (c->groups[type].n != 0) { c->err = "sync_factors_indirect: the type already has factors in this description"; return BSGPU_ERR_INVALID; }
  const bso::TypeInfo& ti = bso::kTypes[type];
  Ctx::SyncPrev& pv = c->sync_prev[type];
  const size_t ni = ti.nidx, nc = ti.nconst;
  if (n_changed >= 0 && pv.valid) {
    std::vector<uint8_t> listed(n, 0);
    for (int i = 0; i < n_changed; ++i) {
      if (changed_rows[i] < 0 || changed_rows[i] >= n) { c->err = "sync_factors_indirect: changed row out of range"; return BSGPU_ERR_INVALID; }
      listed[changed_rows[i]] = 1;
    }
    const size_t old_n = pv.loss_kind.size();
    for (size_t r = 0; r < (size_t)n; ++r) {
      if (listed[r]) continue;
      bool same = r < old_n;
      for (size_t k = 0; k < ni && same; ++k) same = pv.idx[r * ni + k] == slot_idx[r * ni + k];
      for (size_t k = 0; k < nc && same; ++k) same = pv.consts[r * nc + k] == consts[r * nc + k];
      same = same && pv.loss_kind[r] == (loss_kind ? loss_kind[r] : 0) && pv.loss_a[r] == (loss_a ? loss_a[r] : 1.0);
      if (!same) { c->err = "sync_factors_indirect: the changed list does not account for every difference to the previous table"; return BSGPU_ERR_INVALID; }
    }
  }
  pv.idx.assign(slot_idx, slot_idx + (size_t)n * ni); pv.consts.assign(consts, consts + (size_t)n * nc);
  pv.loss_kind.assign(n, 0); pv.loss_a.assign(n, 1.0);
  for (int r = 0; r < n; ++r) { if (loss_kind) pv.loss_kind[r] = loss_kind[r]; if (loss_a) pv.loss_a[r] = loss_a[r]; }
  pv.valid = true;
  return bso_add_factors_indirect(c, type, n, slot_idx, n_slots, slot_to_block, consts, loss_kind, loss_a);
}
int bso_add_marginal(Ctx* c, int32_t n_blocks, const int32_t* blocks, int32_t n_rows, const double* A, const double* b,
                     const double* xbar) {
  if (n_blocks <= 0 || n_rows <= 0 || !blocks || !A || !b || !xbar) { c->err = "add_marginal: bad arguments"; return BSGPU_ERR_INVALID; }
  bso::Marginal mg;
  mg.blocks.assign(blocks, blocks + n_blocks);
  int cols = 0, amb = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const int bl = blocks[i];
    if (bl < 0 || bl >= c->nb) { c->err = "add_marginal: block out of range (set_blocks first)"; return BSGPU_ERR_INVALID; }
    cols += (c->manifold[bl] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : c->size[bl];
    amb += c->size[bl];
  }
  mg.rows = n_rows; mg.cols = cols;
  mg.A.assign(A, A + (size_t)n_rows * cols); mg.b.assign(b, b + n_rows); mg.xbar.assign(xbar, xbar + amb);
  c->marginals.push_back(std::move(mg));
  c->finalized = false;
  return BSGPU_OK;
}
int bso_finalize(Ctx* c) { return bso::finalize(*c); }
int bso_solve(Ctx* c, const bsgpu_options* o, bsgpu_summary* s) { return bso::solve(*c, *o, *s); }
int bso_get_blocks(Ctx* c, double* v, int64_t n) {
  if ((size_t)n != c->x.size()) { c->err = "get_blocks: size mismatch"; return BSGPU_ERR_INVALID; }
  std::memcpy(v, c->x.data(), sizeof(double) * n); return BSGPU_OK;
}
int bso_reset_values(Ctx* c) { c->x = c->x0; return BSGPU_OK; }
int bso_num_iterations_recorded(const Ctx* c) { return (int)c->iters.size(); }
int bso_get_iteration(const Ctx* c, int32_t i, bsgpu_iteration* out) {
  if (i < 0 || i >= (int)c->iters.size()) return BSGPU_ERR_INVALID;
  *out = c->iters[i]; return BSGPU_OK;
}
int bso_num_residuals(const Ctx* c) { return c->num_res; }
int bso_num_parameters_tangent(const Ctx* c) { return c->n_tan; }
int bso_tangent_offset(const Ctx* c, int32_t b) { return (b >= 0 && b < c->nb && c->finalized) ? c->toff[b] : -1; }

int bso_evaluate(Ctx* c, double* cost, double* residuals, double* gradient, double* jacobian) {
  int rc = bso::finalize(*c);
  if (rc != BSGPU_OK) return rc;
  double fixed = 0;
  const double cst = bso::evaluate(*c, c->x.data(), true, &fixed);
  if (cost) *cost = cst + fixed;
  if (residuals)
    for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
      bso::Group& g = c->groups[t];
      if (g.n) std::memcpy(residuals + g.row0, g.r.data(), sizeof(double) * g.r.size());
    }
  if (residuals) for (const bso::Marginal& mg : c->marginals) std::memcpy(residuals + mg.row0, mg.r.data(), sizeof(double) * mg.rows);
  if (gradient) bso::gradient_of(*c, gradient);
  if (jacobian) {
    if ((size_t)c->num_res * c->n_tan > ((size_t)64 << 20)) { c->err = "dense jacobian too large"; return BSGPU_ERR_UNSUPPORTED; }
    std::fill(jacobian, jacobian + (size_t)c->num_res * c->n_tan, 0.0);
    for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
      bso::Group& g = c->groups[t];
      const bso::TypeInfo& ti = bso::kTypes[t];
      for (int f = 0; f < g.n; ++f) {
        int cols[10];
        bso::factor_cols(*c, t, &g.idx[(size_t)f * ti.nidx], cols);
        for (int k = 0; k < ti.m; ++k)
          for (int sl = 0; sl < ti.nvar; ++sl) {
            if (cols[sl] < 0) continue;
            for (int i = 0; i < bso::tsz(ti, sl); ++i)
              jacobian[(size_t)(g.row0 + f * ti.m + k) * c->n_tan + cols[sl] + i] = g.J[((size_t)f * ti.m + k) * g.tw + 3 * sl + i];
          }
      }
    }
    for (const bso::Marginal& mg : c->marginals)
      for (int k = 0; k < mg.rows; ++k)
        for (int a = 0; a < mg.cols; ++a)
          if (mg.col_t[a] >= 0) jacobian[(size_t)(mg.row0 + k) * c->n_tan + mg.col_t[a]] = mg.J[(size_t)k * mg.cols + a];
  }
  return BSGPU_OK;
}

// Marginal covariance block in tangent space: blocks of (J^T J)^-1 of the robustified Jacobian at the
// current values ([EXT] ceres::Covariance as used by Graph::getCovariance,
// bs_constraints/tests/absolute_imu_state_3d_stamped_constraint_test.cpp:200-297).  Dense, small problems.
int bso_covariance(Ctx* c, int32_t ba, int32_t bb, double* out) {
  int rc = bso::finalize(*c);
  if (rc != BSGPU_OK) return rc;
  const int n = c->n_tan;
  if (n > 4000) { c->err = "oracle covariance: problem too large"; return BSGPU_ERR_UNSUPPORTED; }
  if (ba < 0 || bb < 0 || ba >= c->nb || bb >= c->nb || c->toff[ba] < 0 || c->toff[bb] < 0) { c->err = "covariance: bad block"; return BSGPU_ERR_INVALID; }
  std::vector<double> Jd((size_t)c->num_res * n);
  rc = bso_evaluate(c, nullptr, nullptr, nullptr, Jd.data());
  if (rc != BSGPU_OK) return rc;
  std::vector<double> H((size_t)n * n, 0.0);
  for (int r = 0; r < c->num_res; ++r)
    for (int i = 0; i < n; ++i) { const double a = Jd[(size_t)r * n + i]; if (a != 0.0) for (int j = 0; j < n; ++j) H[(size_t)i * n + j] += a * Jd[(size_t)r * n + j]; }
  if (!bso::cholesky_lower(H.data(), n)) { c->err = "covariance: J^T J is singular"; return BSGPU_ERR_NUMERIC; }
  const int ta = c->tsize[ba], tb = c->tsize[bb];
  std::vector<double> e(n);
  for (int j = 0; j < tb; ++j) {
    std::fill(e.begin(), e.end(), 0.0);
    e[c->toff[bb] + j] = 1.0;
    bso::cholesky_solve(H.data(), n, e.data());
    for (int i = 0; i < ta; ++i) out[i * tb + j] = e[c->toff[ba] + i];
  }
  return BSGPU_OK;
}

// [EXT] fuse_constraints::marginalizeVariables, restated densely (test infrastructure): the factors touching the
// marginalised blocks are linearised at the current values (robustified J, r), H = J^T J and g = J^T r are formed
// over [marginalised | kept], the marginalised part is eliminated with a dense Cholesky, and the Schur complement
// is factored with a COMPLETE-PIVOTING semi-definite Cholesky (a different algorithm from the device path, so
// only A^T A and A^T b are comparable, which is all a MarginalConstraint's cost depends on).
int bso_marginalize(Ctx* c, int32_t n_marg, const int32_t* marg_blocks, int32_t* n_kept, int32_t* n_rows, int32_t* n_cols) {
  int rc = bso::finalize(*c);
  if (rc != BSGPU_OK) return rc;
  c->marg_result = Ctx::MargResult();
  const int nb = c->nb;
  std::vector<uint8_t> is_marg(nb, 0), used(nb, 0);
  for (int i = 0; i < n_marg; ++i) {
    const int b = marg_blocks[i];
    if (b < 0 || b >= nb || c->is_const[b]) { c->err = "marginalize: bad block"; return BSGPU_ERR_INVALID; }
    is_marg[b] = 1;
  }
  Ctx sub;
  sub.nb = nb; sub.x = c->x; sub.x0 = c->x; sub.off = c->off; sub.size = c->size; sub.manifold = c->manifold; sub.cams = c->cams;
  sub.reproj_mode = c->reproj_mode;
  int n_connected = 0;
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    const bso::Group& g = c->groups[t];
    const bso::TypeInfo& ti = bso::kTypes[t];
    bso::Group& sg = sub.groups[t];
    sg.type = t;
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      bool touch = false;
      for (int sl = 0; sl < ti.nvar; ++sl) touch = touch || is_marg[idx[sl]];
      if (!touch) continue;
      for (int sl = 0; sl < ti.nvar; ++sl) used[idx[sl]] = 1;
      sg.idx.insert(sg.idx.end(), idx, idx + ti.nidx);
      sg.consts.insert(sg.consts.end(), &g.consts[(size_t)f * ti.nconst], &g.consts[(size_t)f * ti.nconst] + ti.nconst);
      sg.loss_kind.push_back(g.loss_kind[f]); sg.loss_a.push_back(g.loss_a[f]);
      sg.n++; ++n_connected;
    }
  }
  for (const bso::Marginal& mg : c->marginals) {
    bool touch = false;
    for (int b : mg.blocks) touch = touch || is_marg[b];
    if (!touch) continue;
    for (int b : mg.blocks) used[b] = 1;
    bso::Marginal cp; cp.blocks = mg.blocks; cp.rows = mg.rows; cp.cols = mg.cols; cp.A = mg.A; cp.b = mg.b; cp.xbar = mg.xbar;
    sub.marginals.push_back(cp);
    ++n_connected;
  }
  if (!n_connected) { c->err = "marginalize: no factor touches the blocks to marginalise"; return BSGPU_ERR_INVALID; }
  sub.is_const.assign(nb, 1);
  std::vector<int32_t> kept;
  for (int b = 0; b < nb; ++b) {
    sub.is_const[b] = (c->is_const[b] || !used[b]) ? 1 : 0;
    if (used[b] && !c->is_const[b] && !is_marg[b]) kept.push_back(b);
  }
  rc = bso::finalize(sub);
  if (rc != BSGPU_OK) { c->err = "marginalize (sub-problem): " + sub.err; return rc; }
  const int nt = sub.n_tan, nr = sub.num_res;
  if ((size_t)nt > 6000) { c->err = "oracle marginalize: problem too large"; return BSGPU_ERR_UNSUPPORTED; }
  std::vector<double> J((size_t)nr * nt), r(nr);
  rc = bso_evaluate(&sub, nullptr, r.data(), nullptr, J.data());
  if (rc != BSGPU_OK) { c->err = sub.err; return rc; }
  // order
  std::vector<int> order;
  for (int b = 0; b < nb; ++b) if (is_marg[b] && sub.toff[b] >= 0) for (int k = 0; k < sub.tsize[b]; ++k) order.push_back(sub.toff[b] + k);
  const int m = (int)order.size();
  for (int b : kept) for (int k = 0; k < sub.tsize[b]; ++k) order.push_back(sub.toff[b] + k);
  const int n = (int)order.size(), kd = n - m;
  if (n != nt) { c->err = "internal: marginalisation order does not cover the sub-problem"; return BSGPU_ERR_UNSUPPORTED; }
  std::vector<double> H((size_t)n * n, 0.0), g(n, 0.0);
  for (int row = 0; row < nr; ++row) {
    const double* Jr = &J[(size_t)row * nt];
    for (int a = 0; a < n; ++a) {
      const double ja = Jr[order[a]];
      if (ja == 0.0) continue;
      g[a] += ja * r[row];
      for (int b2 = 0; b2 < n; ++b2) H[(size_t)a * n + b2] += ja * Jr[order[b2]];
    }
  }
  // eliminate the first m variables (plain right-looking Cholesky steps; must be positive definite)
  for (int j = 0; j < m; ++j) {
    const double d = H[(size_t)j * n + j];
    if (!(d > 0.0) || !std::isfinite(d)) { c->err = "marginalize: the blocks to marginalise are not fully constrained by the factors that touch them"; return BSGPU_ERR_NUMERIC; }
    const double l = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) H[(size_t)i * n + j] /= l;
    g[j] /= l;
    for (int i = j + 1; i < n; ++i) {
      const double lij = H[(size_t)i * n + j];
      g[i] -= lij * g[j];
      for (int k = j + 1; k <= i; ++k) H[(size_t)i * n + k] -= lij * H[(size_t)k * n + j];
    }
  }
  // Schur complement (lower triangle valid) -> full symmetric kd x kd
  std::vector<double> Sx((size_t)kd * kd), gs(kd);
  double dmax = 0.0;
  for (int i = 0; i < kd; ++i) {
    gs[i] = g[m + i];
    for (int k = 0; k <= i; ++k) Sx[(size_t)i * kd + k] = Sx[(size_t)k * kd + i] = H[(size_t)(m + i) * n + m + k];
    dmax = std::max(dmax, Sx[(size_t)i * kd + i]);
  }
  // complete-pivoting semi-definite Cholesky: P S P^T = L L^T (rank rk)
  std::vector<int> piv(kd);
  for (int i = 0; i < kd; ++i) piv[i] = i;
  std::vector<double> L((size_t)kd * kd, 0.0);   // row = step, column = original variable:  A = L (rk x kd) with A^T A = S
  std::vector<double> diag(kd);
  for (int i = 0; i < kd; ++i) diag[i] = Sx[(size_t)i * kd + i];
  std::vector<double> bvec;
  std::vector<double> work = Sx;    // updated in place (full symmetric)
  std::vector<double> gw = gs;
  int rk = 0;
  std::vector<uint8_t> done(kd, 0);
  for (int step = 0; step < kd; ++step) {
    int p = -1; double best = 0.0;
    for (int i = 0; i < kd; ++i) if (!done[i] && work[(size_t)i * kd + i] > best) { best = work[(size_t)i * kd + i]; p = i; }
    if (p < 0 || best <= 1e-11 * dmax) break;
    const double l = std::sqrt(best);
    // row `rk` of A: a_p = l, a_i = work[i][p] / l for the not-yet-eliminated i
    double* Ar = &L[(size_t)rk * kd];
    Ar[p] = l;
    for (int i = 0; i < kd; ++i) if (!done[i] && i != p) Ar[i] = work[(size_t)i * kd + p] / l;
    const double bp = gw[p] / l;
    bvec.push_back(bp);
    done[p] = 1;
    for (int i = 0; i < kd; ++i) {
      if (done[i]) continue;
      gw[i] -= Ar[i] * bp;
      for (int k = 0; k < kd; ++k) if (!done[k]) work[(size_t)i * kd + k] -= Ar[i] * Ar[k];
    }
    ++rk;
  }
  Ctx::MargResult& R = c->marg_result;
  R.kept = kept; R.rows = rk; R.cols = kd;
  R.A.assign(L.begin(), L.begin() + (size_t)rk * kd);
  R.b = bvec;
  for (int b : kept) R.xbar.insert(R.xbar.end(), &c->x[c->off[b]], &c->x[c->off[b]] + c->size[b]);
  R.valid = true;
  *n_kept = (int32_t)kept.size(); *n_rows = rk; *n_cols = kd;
  return BSGPU_OK;
}
int bso_get_marginal(const Ctx* c, int32_t* kept_blocks, double* A, double* b, double* xbar) {
  if (!c->marg_result.valid) return BSGPU_ERR_INVALID;
  const auto& R = c->marg_result;
  if (kept_blocks) std::memcpy(kept_blocks, R.kept.data(), sizeof(int32_t) * R.kept.size());
  if (A) std::memcpy(A, R.A.data(), sizeof(double) * R.A.size());
  if (b) std::memcpy(b, R.b.data(), sizeof(double) * R.b.size());
  if (xbar) std::memcpy(xbar, R.xbar.data(), sizeof(double) * R.xbar.size());
  return BSGPU_OK;
}

// oracle-only knobs
void bso_set_reproj_mode(Ctx* c, int mode) { c->reproj_mode = mode; }
void bso_set_num_threads(Ctx*, int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int bso_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
// manifold helpers exposed for the property tests (jacobian_helper_tests.cpp)
void bso_quat_plus(const double* q, const double* d, double* out) { bso::manifold_plus(BSGPU_MANIFOLD_QUAT_RIGHT, 4, q, d, out); }
void bso_plus_jacobian(const double* q, double* P12) { bso::plus_jacobian(q, P12); }
void bso_quat_to_angle_axis(const double* q, double* aa) { bso::QuaternionToAngleAxis(q, aa); }
void bso_angle_axis_to_quat(const double* aa, double* q) { bso::AngleAxisToQuaternion(aa, q); }

}  // extern "C"
