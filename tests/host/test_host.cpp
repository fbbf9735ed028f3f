// Host-side (C++) tests of the fuse/bs_optimizers mirror in beam_slam_amd/host/, written like the
// reference's own gtests.  Built twice by tests/test_host_cpp.py:
//   * against libbsgpu.so              -> run on the GPU box (-m gpu): the product path
//   * with -include oracle_backend.h   -> against the CPU oracle, to exercise the HOST LOGIC
//     (flattening order, pack(), transactions, lag window, pseudo-marginalisation) where no GPU exists
#include <cstdio>
#include <algorithm>
#include <iostream>
#include <map>
#include <random>
#include <thread>
#include <set>

#define BS_GRAPH_COW_BUCKETS 8   // small copy-on-write granules: every test graph spans several chunks / fills every bucket
#define BS_GRAPH_COW_CHUNK 16
#include "../../beam_slam_amd/host/fixed_lag_smoother.h"

using namespace bs_math;
static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)
#define CHECK_NEAR(a, b, tol) do { const double a_ = (a), b_ = (b); if (!(std::fabs(a_ - b_) <= (tol))) { std::printf("  CHECK_NEAR FAILED %s:%d: %s=%.12g vs %s=%.12g (tol %g)\n", __FILE__, __LINE__, #a, a_, #b, b_, (double)(tol)); ++g_fail; } } while (0)

static Mat<3, 3> I3() { return Mat<3, 3>::Identity(); }
static Mat<6, 6> I6() { return Mat<6, 6>::Identity(); }

// bs_models/tests/imu_preintegration_tests.cpp:292-477
static void test_simple_2_state_fg() {
  std::printf("Simple2StateFG\n");
  bs_common::ImuState IS1(fuse_core::Time(1.0), {0.952, 0.038, -0.189, 0.239}, {1.5, -3.0, 1.0}, {1.5, -3.0, 1.0}, {4e-5, 5e-5, 6e-5}, {1e-5, 2e-5, 3e-5});
  bs_common::ImuState IS2(fuse_core::Time(2.0), {0.944, -0.128, 0.145, -0.269}, {-1.5, 3.0, -1.0}, {-1.5, 3.0, -1.0}, {4e-5, 5e-5, 6e-5}, {1e-5, 2e-5, 3e-5});
  bs_optimizers::GpuGraph graph;
  for (const auto* s : {&IS1, &IS2}) {
    graph.addVariable(s->Orientation().clone()); graph.addVariable(s->Position().clone()); graph.addVariable(s->Velocity().clone());
    graph.addVariable(s->GyroBias().clone()); graph.addVariable(s->AccelBias().clone());
  }
  graph.addConstraint(std::make_shared<fuse_constraints::AbsolutePose3DStampedConstraint>("test", IS1.Position(), IS1.Orientation(), bs_constraints::Vector7d{0, 0, 0, 1, 0, 0, 0}, I6()));
  graph.addConstraint(bs_constraints::AbsoluteVelocityLinear3DStampedConstraint("test", IS1.Velocity(), {0, 0, 0}, I3()));
  graph.addConstraint(bs_constraints::AbsoluteGyroBias3DStampedConstraint("test", IS1.GyroBias(), {0, 0, 0}, I3()));
  graph.addConstraint(bs_constraints::AbsoluteAccelBias3DStampedConstraint("test", IS1.AccelBias(), {0, 0, 0}, I3()));
  graph.addConstraint(std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("test", IS1.Position(), IS1.Orientation(), IS2.Position(), IS2.Orientation(), bs_constraints::Vector7d{1, 0, 0, 1, 0, 0, 0}, I6()));
  graph.addConstraint(bs_constraints::RelativeVelocityLinear3DStampedConstraint("test", IS1.Velocity(), IS2.Velocity(), {1, 0, 0}, I3()));
  graph.addConstraint(bs_constraints::RelativeGyroBias3DStampedConstraint("test", IS1.GyroBias(), IS2.GyroBias(), {0.001, 0, 0}, I3()));
  graph.addConstraint(bs_constraints::RelativeAccelBias3DStampedConstraint("test", IS1.AccelBias(), IS2.AccelBias(), {0.001, 0, 0}, I3()));
  auto summary = graph.optimize();
  CHECK(summary.IsSolutionUsable());
  CHECK(IS1.Update(graph)); CHECK(IS2.Update(graph));
  CHECK(IS1.Updates() == 1);
  const double e1[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double e2[16] = {1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0.001, 0, 0, 0.001, 0, 0};
  const auto s1 = IS1.GetStateVector(), s2 = IS2.GetStateVector();
  for (int i = 0; i < 16; ++i) { CHECK_NEAR(s1[i], e1[i], i < 4 ? 1e-3 : 1e-5); CHECK_NEAR(s2[i], e2[i], i < 4 ? 1e-3 : 1e-5); }
}

// bs_constraints/tests/absolute_imu_state_3d_stamped_constraint_test.cpp:22-165
static void test_absolute_imu_state() {
  std::printf("AbsoluteImuState3DStampedConstraint\n");
  bs_common::ImuState st(fuse_core::Time(1.0), {0.952, 0.038, -0.189, 0.239}, {1.5, -3.0, 10.0}, {1.5, -3.0, 10.0}, {0.15, -0.30, 1.0}, {0.15, -0.30, 1.0});
  std::array<double, 16> mean{1.0, 0.0, 0.0, 0.0, 1.0, 2.0, 3.0, 1.0, 2.0, 3.0, 0.1, 0.2, 0.3, 0.1, 0.2, 0.3};
  Mat<15, 15> cov;
  for (int i = 0; i < 15; ++i) cov(i, i) = i + 1.0;
  for (int j = 1; j < 15; ++j) cov(0, j) = cov(j, 0) = 0.1 * j;
  for (int i = 1; i < 15; ++i) for (int j = i + 1; j < 15; ++j) cov(i, j) = cov(j, i) = 1.5 - 0.1 * (j - i - 1);
  auto c = std::make_shared<bs_constraints::AbsoluteImuState3DStampedConstraint>("test", st, mean, cov);
  // :74-83 sqrt information: U^T U == cov^-1
  Mat<15, 15> info;
  CHECK(invertSpd(cov, info));
  const Mat<15, 15> UtU = transpose(c->sqrtInformation()) * c->sqrtInformation();
  for (int i = 0; i < 225; ++i) CHECK_NEAR(UtU.a[i], info.a[i], 1e-9);
  bs_optimizers::GpuGraph graph;
  graph.addVariable(st.Orientation().clone()); graph.addVariable(st.Position().clone()); graph.addVariable(st.Velocity().clone());
  graph.addVariable(st.GyroBias().clone()); graph.addVariable(st.AccelBias().clone());
  graph.addConstraint(c);
  auto summary = graph.optimize();
  CHECK(summary.IsSolutionUsable());
  CHECK(st.Update(graph));
  const auto s = st.GetStateVector();
  for (int i = 0; i < 16; ++i) CHECK_NEAR(s[i], mean[i], i < 4 ? 1e-3 : 1e-5);
  // :167-297 covariance of the optimised state in tangent space == the prior covariance (1e-5)
  const fuse_core::UUID ids[5] = {st.Orientation().uuid(), st.Position().uuid(), st.Velocity().uuid(), st.GyroBias().uuid(), st.AccelBias().uuid()};
  std::vector<std::pair<fuse_core::UUID, fuse_core::UUID>> requests;
  for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j) requests.push_back({ids[i], ids[j]});
  std::vector<std::vector<double>> blocks;
  graph.getCovariance(requests, blocks);
  CHECK(blocks.size() == requests.size());
  size_t k = 0;
  for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j, ++k)
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) CHECK_NEAR(blocks[k][3 * a + b], cov(3 * i + a, 3 * j + b), 1e-5);
}

// deterministic variable index + constraint payloads
static void test_block_order_and_pack() {
  std::printf("BlockOrderAndPack\n");
  bs_optimizers::GpuGraph graph;
  auto ext_q = bs_variables::Orientation3D::make_shared("cam", "base");
  auto ext_p = bs_variables::Position3D::make_shared("cam", "base");
  graph.addVariable(ext_q); graph.addVariable(ext_p);
  graph.addVariable(bs_variables::Point3DLandmark::make_shared(17));
  graph.addVariable(bs_variables::Point3DLandmark::make_shared(3));
  for (double t : {2.0, 1.0}) {
    bs_common::ImuState s{fuse_core::Time(t)};
    graph.addVariable(s.AccelBias().clone()); graph.addVariable(s.Velocity().clone()); graph.addVariable(s.Orientation().clone());
    graph.addVariable(s.GyroBias().clone()); graph.addVariable(s.Position().clone());
  }
  const auto v = graph.orderedVariables();
  const char* expect[14] = {"fuse_variables::Orientation3DStamped", "fuse_variables::Position3DStamped", "fuse_variables::VelocityLinear3DStamped",
                            "bs_variables::GyroscopeBias3DStamped", "bs_variables::AccelerationBias3DStamped",
                            "fuse_variables::Orientation3DStamped", "fuse_variables::Position3DStamped", "fuse_variables::VelocityLinear3DStamped",
                            "bs_variables::GyroscopeBias3DStamped", "bs_variables::AccelerationBias3DStamped",
                            "bs_variables::Point3DLandmark", "bs_variables::Point3DLandmark", "", ""};
  CHECK(v.size() == 14);
  for (int i = 0; i < 12; ++i) CHECK(v[i]->type() == expect[i]);
  CHECK(v[0]->stamp() == fuse_core::Time(1.0)); CHECK(v[5]->stamp() == fuse_core::Time(2.0));
  CHECK(v[10]->landmarkId() == 3); CHECK(v[11]->landmarkId() == 17);
  CHECK(v[12]->holdConstant() && v[13]->holdConstant());
  // same ids -> same uuids (fuse_core::uuid::generate semantics)
  CHECK(bs_variables::Point3DLandmark(3).uuid() == v[10]->uuid());
  CHECK(fuse_variables::Position3DStamped(fuse_core::Time(1.0)).uuid() == v[1]->uuid());
  // pack(): block order = constraint's variables() order, camera de-duplication, Cauchy loss
  fuse_core::FactorTables t;
  std::map<fuse_core::UUID, int32_t> bi;
  for (size_t i = 0; i < v.size(); ++i) bi[v[i]->uuid()] = (int32_t)i;
  const fuse_core::BlockOf block_of([&](const fuse_core::UUID& u) { return bi.at(u); });
  Mat<4, 4> T = Mat<4, 4>::Identity(); T(0, 3) = 0.1;
  Mat<3, 3> K = Mat<3, 3>::Identity(); K(0, 0) = 458.654; K(1, 1) = 457.296; K(0, 2) = 367.215; K(1, 2) = 248.375;
  bs_common::ImuState s1{fuse_core::Time(1.0)};
  for (int k = 0; k < 2; ++k) {
    bs_constraints::EuclideanReprojectionConstraint rc("vo", s1.Orientation(), s1.Position(), bs_variables::Point3DLandmark(k ? 17 : 3), T, K, {100.0 + k, 50.0}, 1.0);
    rc.loss(std::make_shared<fuse_loss::CauchyLoss>(5.0));
    rc.pack(block_of, t);
  }
  CHECK(t.count(BSGPU_F_REPROJ) == 2); CHECK(t.cameras.size() == 1);
  const int32_t exp_idx[8] = {0, 1, 10, 0, 0, 1, 11, 0};
  for (int i = 0; i < 8; ++i) CHECK(t.idx[BSGPU_F_REPROJ][i] == exp_idx[i]);
  CHECK(t.loss_kind[BSGPU_F_REPROJ][0] == BSGPU_LOSS_CAUCHY); CHECK_NEAR(t.loss_a[BSGPU_F_REPROJ][1], 5.0, 0);
  CHECK_NEAR(t.cameras[0].fx, 458.654, 0); CHECK_NEAR(t.cameras[0].t_cam_baselink[0], 0.1, 0);
}

// inverse-depth landmarks (use_idp) through the host mirror — bearing validation
// (bs_variables/src/inverse_depth_landmark.cpp:21-26), anchor (unary) + later (binary) observations, recovery of the
// generating inverse depths with the two anchor poses held by priors
static void test_inverse_depth_window() {
  std::printf("InverseDepthWindow\n");
  bool threw = false;
  try { bs_variables::InverseDepthLandmark bad(1, Vec3{0.1, 0.2, 1.0}, fuse_core::Time(0.0)); } catch (const std::runtime_error&) { threw = true; }
  CHECK(threw);
  std::mt19937 rng(11);
  std::normal_distribution<double> N(0.0, 1.0);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  Mat<4, 4> T = Mat<4, 4>::Identity(); T(0, 3) = 0.05; T(1, 3) = -0.02;
  Mat<3, 3> K = Mat<3, 3>::Identity(); K(0, 0) = 458.654; K(1, 1) = 457.296; K(0, 2) = 367.215; K(1, 2) = 248.375;
  const int n_kf = 4, n_lm = 25;
  bs_optimizers::GpuGraph graph;
  std::vector<fuse_variables::Orientation3DStamped::SharedPtr> qs;
  std::vector<fuse_variables::Position3DStamped::SharedPtr> ps;
  std::vector<Vec3> p_true;
  for (int k = 0; k < n_kf; ++k) {
    auto q = fuse_variables::Orientation3DStamped::make_shared(fuse_core::Time(1.0 + 0.1 * k));
    auto p = fuse_variables::Position3DStamped::make_shared(fuse_core::Time(1.0 + 0.1 * k));
    p_true.push_back(Vec3{0.4 * k, 0.1 * k * k, 0.05 * k});
    q->data()[0] = 1.0;   // identity attitude (truth), positions start perturbed
    for (int i = 0; i < 3; ++i) p->data()[i] = p_true[k][i] + (k >= 2 ? 0.03 * N(rng) : 0.0);
    graph.addVariable(q); graph.addVariable(p);
    qs.push_back(q); ps.push_back(p);
  }
  // gauge + scale: tight priors on the first two poses
  Mat<6, 6> cov6 = Mat<6, 6>::Identity(); for (int i = 0; i < 6; ++i) cov6(i, i) = 1e-8;
  for (int k = 0; k < 2; ++k)
    graph.addConstraint(std::make_shared<fuse_constraints::AbsolutePose3DStampedConstraint>(
        "prior", *ps[k], *qs[k], bs_constraints::Vector7d{p_true[k][0], p_true[k][1], p_true[k][2], 1, 0, 0, 0}, cov6));
  std::vector<bs_variables::InverseDepthLandmark::SharedPtr> lms;
  std::vector<double> rho_true;
  for (int l = 0; l < n_lm; ++l) {
    // point in the anchor (kf 0) camera; camera = baselink shifted by T_cam_baselink's translation, attitude identity
    const double z = 4.0 + 6.0 * U(rng), x = (U(rng) - 0.5) * 0.8 * z, y = (U(rng) - 0.5) * 0.5 * z;
    const double n = std::sqrt(x * x + y * y + z * z);
    auto lm = std::make_shared<bs_variables::InverseDepthLandmark>((uint64_t)(100 + l), Vec3{x / n, y / n, z / n}, fuse_core::Time(1.0));
    rho_true.push_back(1.0 / n);
    lm->inverse_depth() = (1.0 / n) * (1.0 + 0.1 * N(rng));
    graph.addVariable(lm);
    lms.push_back(lm);
    for (int k = 0; k < n_kf; ++k) {
      // P_cam_k = P_cam_0 + (p_0 - p_k)  (identity attitudes, same extrinsic)
      const double px = x + p_true[0][0] - p_true[k][0], py = y + p_true[0][1] - p_true[k][1], pz = z + p_true[0][2] - p_true[k][2];
      const std::array<double, 2> uv = {K(0, 0) * px / pz + K(0, 2), K(1, 1) * py / pz + K(1, 2)};
      if (k == 0) {
        auto c = std::make_shared<bs_constraints::InverseDepthReprojectionConstraintUnary>("vo", *qs[0], *ps[0], *lm, T, K, uv, 1.0);
        graph.addConstraint(c);
      } else {
        auto c = std::make_shared<bs_constraints::InverseDepthReprojectionConstraint>("vo", *qs[0], *ps[0], *qs[k], *ps[k], *lm, T, K, uv, 1.0);
        c->loss(std::make_shared<fuse_loss::CauchyLoss>(5.0));
        graph.addConstraint(c);
      }
    }
  }
  auto summary = graph.optimize();
  CHECK(summary.IsSolutionUsable());
  CHECK(summary.final_cost < 1e-6 * std::max(1.0, summary.initial_cost));
  for (int l = 0; l < n_lm; ++l) {
    const auto& v = graph.getVariable(lms[l]->uuid());
    CHECK_NEAR(v.data()[0], rho_true[l], 1e-5);
  }
  for (int k = 2; k < n_kf; ++k) {
    const auto& v = graph.getVariable(ps[k]->uuid());
    for (int i = 0; i < 3; ++i) CHECK_NEAR(v.data()[i], p_true[k][i], 1e-4);
  }
  std::printf("  cost %.4e -> %.4e in %d iterations\n", summary.initial_cost, summary.final_cost, (int)summary.iterations.size() - 1);
}

// [EXT] fuse_constraints::marginalizeVariables through the host mirror: on a linear-Gaussian chain marginalisation is
// exact, so the remaining variables must reach the optimum of the full graph; a second marginalisation absorbs the
// first MarginalConstraint (what a sliding window does every cycle).
static void test_true_marginalization_linear_chain() {
  std::printf("MarginalizeVariablesLinearChain\n");
  std::mt19937 rng(3);
  std::normal_distribution<double> N(0.0, 1.0);
  const int n = 7;
  std::vector<fuse_variables::VelocityLinear3DStamped::SharedPtr> v;
  std::vector<fuse_core::Constraint::SharedPtr> cs;
  std::vector<Vec3> truth(n);
  for (int i = 0; i < n; ++i) {
    truth[i] = Vec3{0.5 * i + 0.2 * N(rng), -0.3 * i + 0.2 * N(rng), 0.1 * i * i};
    auto x = fuse_variables::VelocityLinear3DStamped::make_shared(fuse_core::Time(1.0 + i));
    for (int k = 0; k < 3; ++k) x->data()[k] = truth[i][k] + 0.3 * N(rng);
    v.push_back(x);
  }
  auto diag = [](double a, double b, double c) { Mat<3, 3> m; m(0, 0) = a; m(1, 1) = b; m(2, 2) = c; return m; };
  cs.push_back(bs_constraints::AbsoluteVelocityLinear3DStampedConstraint("t", *v[0], Vec3{truth[0][0] + 0.05 * N(rng), truth[0][1], truth[0][2]}, diag(0.05, 0.04, 0.06)));
  cs.push_back(bs_constraints::AbsoluteVelocityLinear3DStampedConstraint("t", *v[4], Vec3{truth[4][0], truth[4][1] + 0.05 * N(rng), truth[4][2]}, diag(0.03, 0.05, 0.05)));
  for (int i = 0; i + 1 < n; ++i)
    cs.push_back(bs_constraints::RelativeVelocityLinear3DStampedConstraint("t", *v[i], *v[i + 1], Vec3{truth[i + 1][0] - truth[i][0] + 0.05 * N(rng), truth[i + 1][1] - truth[i][1] + 0.05 * N(rng), truth[i + 1][2] - truth[i][2]}, diag(0.02, 0.05, 0.03)));
  for (int i = 0; i + 2 < n; ++i)
    cs.push_back(bs_constraints::RelativeVelocityLinear3DStampedConstraint("t", *v[i], *v[i + 2], Vec3{truth[i + 2][0] - truth[i][0], truth[i + 2][1] - truth[i][1] + 0.1 * N(rng), truth[i + 2][2] - truth[i][2] + 0.1 * N(rng)}, diag(0.1, 0.08, 0.12)));
  ceres_compat::SolverOptions tight;
  tight.function_tolerance = 1e-16; tight.gradient_tolerance = 1e-14; tight.parameter_tolerance = 1e-14; tight.max_num_iterations = 100;
  bs_optimizers::GpuGraph full, graph;
  for (auto* g : {&full, &graph}) { for (auto& x : v) g->addVariable(x->clone()); for (auto& c : cs) g->addConstraint(c->clone()); }
  CHECK(full.optimize(tight).IsSolutionUsable());
  // marginalise the two oldest variables BEFORE optimising (linear problem: exact wherever it is linearised)
  auto tr = fuse_constraints::marginalizeVariables("test", {v[0]->uuid(), v[1]->uuid()}, graph);
  CHECK(tr.removedVariables().size() == 2);
  CHECK(tr.removedConstraints().size() == 5);     // prior on 0, edges 0-1, 1-2, 0-2, 1-3
  CHECK(tr.addedConstraints().size() == 1);
  if (tr.addedConstraints().size() == 1) {
    const auto* mc = dynamic_cast<const fuse_constraints::MarginalConstraint*>(tr.addedConstraints()[0].get());
    CHECK(mc != nullptr);
    if (mc) { CHECK(mc->variables().size() == 2); CHECK(mc->cols() == 6); CHECK(mc->rows() == 6); CHECK(mc->variables()[0] == v[2]->uuid()); CHECK(mc->variables()[1] == v[3]->uuid()); }
  }
  graph.update(tr);
  CHECK(graph.numVariables() == (size_t)n - 2);
  CHECK(graph.optimize(tight).IsSolutionUsable());
  for (int i = 2; i < n; ++i) for (int k = 0; k < 3; ++k) CHECK_NEAR(graph.getVariable(v[i]->uuid()).data()[k], full.getVariable(v[i]->uuid()).data()[k], 1e-8);
  // slide once more: the previous MarginalConstraint is consumed and replaced
  auto tr2 = fuse_constraints::marginalizeVariables("test", {v[2]->uuid()}, graph);
  CHECK(tr2.addedConstraints().size() == 1);
  graph.update(tr2);
  int n_marginal = 0;
  for (const auto* c : graph.getConstraints()) if (c->type() == "fuse_constraints::MarginalConstraint") ++n_marginal;
  CHECK(n_marginal == 1);
  CHECK(graph.optimize(tight).IsSolutionUsable());
  for (int i = 3; i < n; ++i) for (int k = 0; k < 3; ++k) CHECK_NEAR(graph.getVariable(v[i]->uuid()).data()[k], full.getVariable(v[i]->uuid()).data()[k], 1e-8);
  // a variable nothing constrains is just removed
  auto lonely = fuse_variables::VelocityLinear3DStamped::make_shared(fuse_core::Time(99.0));
  graph.addVariable(lonely);
  auto tr3 = fuse_constraints::marginalizeVariables("test", {lonely->uuid()}, graph);
  CHECK(tr3.addedConstraints().empty()); CHECK(tr3.removedVariables().size() == 1);
}

// a synthetic visual-inertial stream through the fixed-lag smoother: lag window, pseudo-marginalisation
// (fixed_lag_smoother.cpp:244-268) or, with pseudo_marginalization == false, true marginalisation (:269-272)
static void test_fixed_lag_smoother_window(bool pseudo_marginalization) {
  std::printf("FixedLagSmootherWindow (%s)\n", pseudo_marginalization ? "pseudo-marginalisation" : "true marginalisation");
  std::mt19937 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  const int n_kf = 14, per = 20;
  const double dt_kf = 0.1, dt_imu = dt_kf / per;
  // straight-ish motion with gentle turning; body z up, camera looks along body z offset (identity extrinsic for simplicity)
  auto pos = [](double t) { return Vec3{1.0 * t, 0.3 * std::sin(t), 0.05 * t}; };
  auto vel = [](double t) { return Vec3{1.0, 0.3 * std::cos(t), 0.05}; };
  auto acc = [](double t) { return Vec3{0.0, -0.3 * std::sin(t), 0.0}; };
  bs_optimizers::FixedLagSmootherParams params;
  params.lag_duration = 0.55;  // ~6 keyframes
  params.pseudo_marginalization = pseudo_marginalization;
  params.solver_options = ceres_compat::SolverOptions();
  params.solver_options.max_num_iterations = 20;
  bs_optimizers::FixedLagSmoother smoother(bs_optimizers::GpuGraph::make_unique(), params);
  {   // setDiagnostics (fixed_lag_smoother.cpp:676-740) before anything has arrived: "Started" false, an empty queue, no summary fields
    const auto d0 = smoother.diagnostics();
    CHECK(d0.find("Started") && *d0.find("Started") == "False");
    CHECK(d0.find("Pending Transactions") && *d0.find("Pending Transactions") == "0");
    CHECK(d0.find("Final Cost") == nullptr);
  }
  // notify(transaction, graph->clone()) (fixed_lag_smoother.cpp:308): a publisher keeps every snapshot it was handed
  struct Seen { std::shared_ptr<const bs_optimizers::GpuGraph> graph; size_t n_var, n_con, n_added; };
  std::vector<Seen> seen;
  smoother.setNotifyCallback([&](std::shared_ptr<const fuse_core::Transaction> tr, std::shared_ptr<const bs_optimizers::GpuGraph> g) {
    seen.push_back(Seen{g, g->numVariables(), g->numConstraints(), tr->addedConstraints().size()});
  });
  Mat<4, 4> T_cam_baselink = Mat<4, 4>::Identity();
  Mat<3, 3> K = Mat<3, 3>::Identity(); K(0, 0) = 458.654; K(1, 1) = 457.296; K(0, 2) = 367.215; K(1, 2) = 248.375;
  // landmarks ahead on a wall z = 8 (camera frame == body frame here: +z forward means world +z; fine for a synthetic test)
  const int n_lm = 60;
  std::vector<Vec3> P(n_lm);
  for (int j = 0; j < n_lm; ++j) P[j] = {0.2 * j - 2.0 + 0.5 * N(rng), 1.5 * N(rng), 8.0 + N(rng)};
  std::set<uint64_t> known_landmarks;
  bs_common::ImuState prev;
  int usable = 0;
  size_t max_vars = 0;
  for (int k = 0; k < n_kf; ++k) {
    const double t = k * dt_kf;
    auto tr = std::make_shared<fuse_core::Transaction>();
    tr->stamp(fuse_core::Time(t));
    const Vec3 p = pos(t), v = vel(t);
    bs_common::ImuState st(fuse_core::Time(t), quatFromAngleAxis({0.01 * N(rng), 0.01 * N(rng), 0.01 * N(rng)}),
                           {p[0] + 0.02 * N(rng), p[1] + 0.02 * N(rng), p[2] + 0.02 * N(rng)}, {v[0] + 0.02 * N(rng), v[1] + 0.02 * N(rng), v[2] + 0.02 * N(rng)});
    tr->addVariable(st.Orientation().clone()); tr->addVariable(st.Position().clone()); tr->addVariable(st.Velocity().clone());
    tr->addVariable(st.GyroBias().clone()); tr->addVariable(st.AccelBias().clone());
    tr->addInvolvedStamp(fuse_core::Time(t));
    if (k == 0) {  // prior on the first state (imu_preintegration.cpp:267-277)
      Mat<15, 15> cov = 1e-3 * Mat<15, 15>::Identity();
      tr->addConstraint(std::make_shared<bs_constraints::AbsoluteImuState3DStampedConstraint>("imu", st, st.GetStateVector(), cov));
    } else {
      auto pre = std::make_shared<bs_common::PreIntegrator>();
      pre->cov_w = 5.7e-4 * I3(); pre->cov_a = 9.4e-4 * I3(); pre->cov_bg = 3.7e-6 * I3(); pre->cov_ba = 2.4e-6 * I3();
      for (int i = 0; i <= per; ++i) {
        const double ti = (k - 1) * dt_kf + i * dt_imu;
        bs_common::IMUData d; d.t = fuse_core::Time(ti);
        const Vec3 a = acc(ti);
        d.w = {0, 0, 0}; d.a = {a[0], a[1], a[2] + 9.80665};   // identity attitude: f = a - g
        pre->data[d.t] = d;
      }
      CHECK(pre->Integrate(fuse_core::Time(t), {0, 0, 0}, {0, 0, 0}, true, true, true));
      CHECK_NEAR(pre->delta.t, dt_kf, 1e-9);
      tr->addConstraint(std::make_shared<bs_constraints::RelativeImuState3DStampedConstraint>("imu", prev, st, pre, 1.0));
      tr->addInvolvedStamp(prev.Stamp());
    }
    for (int j = 0; j < n_lm; ++j) {
      const Vec3 pc{P[j][0] - p[0], P[j][1] - p[1], P[j][2] - p[2]};
      if (pc[2] < 1.0) continue;
      const double u = K(0, 0) * pc[0] / pc[2] + K(0, 2), vv = K(1, 1) * pc[1] / pc[2] + K(1, 2);
      if (u < 0 || u > 752 || vv < 0 || vv > 480) continue;
      auto lm = bs_variables::Point3DLandmark::make_shared(j);
      if (!known_landmarks.count(j)) { lm->x() = P[j][0] + 0.1 * N(rng); lm->y() = P[j][1] + 0.1 * N(rng); lm->z() = P[j][2] + 0.1 * N(rng); tr->addVariable(lm); known_landmarks.insert(j); }
      auto c = std::make_shared<bs_constraints::EuclideanReprojectionConstraint>("vo", st.Orientation(), st.Position(), *lm, T_cam_baselink, K,
                                                                              std::array<double, 2>{std::round(u + N(rng)), std::round(vv + N(rng))}, 1.0);
      c->loss(std::make_shared<fuse_loss::CauchyLoss>(5.0));
      tr->addConstraint(c);
    }
    smoother.transactionCallback("synthetic", tr);
    const auto res = smoother.optimizeOnce();
    CHECK(res == bs_optimizers::FixedLagSmoother::CycleResult::Optimized);
    if (smoother.summary().IsSolutionUsable()) ++usable;
    CHECK(smoother.summary().final_cost <= smoother.summary().initial_cost * (1 + 1e-12));
    max_vars = std::max(max_vars, smoother.graph().numVariables());
    prev = st;
    prev.Update(smoother.graph());
  }
  CHECK(usable == n_kf);
  CHECK((int)seen.size() == n_kf);
  for (const auto& sn : seen) {   // later cycles slid the window and re-solved: the old snapshots did not move
    CHECK(sn.graph->numVariables() == sn.n_var && sn.graph->numConstraints() == sn.n_con && sn.n_added > 0);
    CHECK(sn.graph->getConstraints().size() == sn.n_con);
  }
  CHECK(seen.front().graph->numVariables() < seen.back().graph->numVariables());
  CHECK(smoother.optimizeOnce() == bs_optimizers::FixedLagSmoother::CycleResult::NothingToDo);
  {   // the diagnostics after the last cycle: the fields and the level / message the reference derives from the termination type
    const auto d = smoother.diagnostics();
    CHECK(d.find("Started") && *d.find("Started") == "True");
    CHECK(d.find("Pending Transactions") && *d.find("Pending Transactions") == "0");
    CHECK(d.find("Optimization Termination Type") != nullptr && d.find("Optimization Total Time [s]") != nullptr);
    CHECK(d.find("Optimization Iterations") && std::stoul(*d.find("Optimization Iterations")) == smoother.summary().iterations.size());
    CHECK(d.find("Initial Cost") != nullptr && d.find("Final Cost") != nullptr);
    CHECK_NEAR(std::stod(*d.find("Final Cost")), smoother.summary().final_cost, 1e-5 * (1.0 + smoother.summary().final_cost));
    const bool conv = smoother.summary().termination_type == ceres_compat::CONVERGENCE;
    CHECK(d.level == (conv ? 0 : 1));
    CHECK(d.message == (conv ? "Optimization converged" : "Optimization didn't converge"));
  }
  // the window is bounded: stamped states older than the lag are gone, and a MARGINALIZATION prior exists
  int n_pos = 0, n_marg = 0;
  fuse_core::Time oldest(1e9);
  for (const auto* v : smoother.graph().getVariables()) if (v->type() == "fuse_variables::Position3DStamped") { ++n_pos; if (v->stamp() < oldest) oldest = v->stamp(); }
  int n_marginal_constraints = 0;
  for (const auto* c : smoother.graph().getConstraints()) {
    if (c->source() == "MARGINALIZATION") ++n_marg;
    if (c->type() == "fuse_constraints::MarginalConstraint") ++n_marginal_constraints;
  }
  CHECK(n_pos <= 8); CHECK(n_pos >= 5);
  CHECK(oldest >= smoother.lagExpiration());
  if (pseudo_marginalization) { CHECK(n_marg >= 1); CHECK(n_marginal_constraints == 0); }
  else { CHECK(n_marg == 0); CHECK(n_marginal_constraints == 1); }   // every slide absorbs the previous prior
  const auto first = smoother.GetWindowStartState();
  CHECK(first.Stamp() == oldest);
  // estimates stay near the truth (position within 0.2 m, see the noise above)
  for (const auto* v : smoother.graph().getVariables())
    if (v->type() == "fuse_variables::Position3DStamped") {
      const Vec3 pt = pos(v->stamp().toSec());
      for (int i = 0; i < 3; ++i) CHECK_NEAR(v->data()[i], pt[i], 0.2);
    }
  std::printf("  cycles %d, window positions %d, max variables %zu, final cost %.4f\n", smoother.numCycles(), n_pos, max_vars, smoother.summary().final_cost);
}

// Graph::clone() (fixed_lag_smoother.cpp:308): the copy shares the constraint side with the original behind
// copy-on-write handles (gpu_graph.h) — it must still behave as a deep copy: transactions applied to either side
// afterwards are invisible to the other, both sides stay solvable, and a chain of clones keeps that up.
static void test_clone_is_an_independent_snapshot() {
  std::printf("CloneIsAnIndependentSnapshot\n");
  std::mt19937 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  bs_optimizers::GpuGraph graph;
  std::vector<bs_common::ImuState> st;
  Mat<6, 6> c6 = 1e-2 * I6();
  auto add_state = [&](bs_optimizers::GpuGraph& g, int k) {
    while ((int)st.size() <= k) st.emplace_back(fuse_core::Time(0.1 * st.size()), std::array<double, 4>{1, 0, 0, 0},
                                                std::array<double, 3>{0.1 * st.size() + 0.02 * N(rng), 0.02 * N(rng), 0.02 * N(rng)}, std::array<double, 3>{1, 0, 0});
    g.addVariable(st[k].Orientation().clone()); g.addVariable(st[k].Position().clone());
  };
  std::vector<fuse_core::UUID> odom;
  auto add_odom = [&](bs_optimizers::GpuGraph& g, int k) {
    auto c = std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("odom", st[k].Position(), st[k].Orientation(), st[k + 1].Position(),
                                                                               st[k + 1].Orientation(), bs_constraints::Vector7d{0.1, 0, 0, 1, 0, 0, 0}, c6);
    if ((int)odom.size() <= k) odom.resize(k + 1);
    odom[k] = c->uuid();
    g.addConstraint(c);
  };
  const int n = 120;   // 7 chunks of constraint slots, every index bucket populated
  for (int k = 0; k < n; ++k) add_state(graph, k);
  graph.addConstraint(std::make_shared<fuse_constraints::AbsolutePose3DStampedConstraint>("prior", st[0].Position(), st[0].Orientation(),
                                                                                        bs_constraints::Vector7d{0, 0, 0, 1, 0, 0, 0}, 1e-4 * I6()));
  for (int k = 0; k + 1 < n; ++k) add_odom(graph, k);
  auto snap = graph.clone();
  CHECK(snap->numVariables() == graph.numVariables() && snap->numConstraints() == graph.numConstraints());
  // slide the original: drop the 40 oldest odometry factors and states 0..39 keep only a new prior on state 40; extend by 30 states
  {
    fuse_core::Transaction tr;
    for (const auto* c : graph.getConnectedConstraints(st[0].Position().uuid())) tr.removeConstraint(c->uuid());
    for (int k = 1; k < 40; ++k) tr.removeConstraint(odom[k]);
    for (int k = 0; k < 40; ++k) { tr.removeVariable(st[k].Position().uuid()); tr.removeVariable(st[k].Orientation().uuid()); }
    graph.update(tr);
    graph.addConstraint(std::make_shared<fuse_constraints::AbsolutePose3DStampedConstraint>("prior", st[40].Position(), st[40].Orientation(),
                                                                                          bs_constraints::Vector7d{4.0, 0, 0, 1, 0, 0, 0}, 1e-4 * I6()));
    for (int k = n; k < n + 30; ++k) { add_state(graph, k); add_odom(graph, k - 1); }
  }
  CHECK(graph.numVariables() == 2u * (n - 40 + 30)); CHECK(graph.numConstraints() == (size_t)(n - 1 - 40 + 30 + 1));
  // the snapshot is untouched
  CHECK(snap->numVariables() == 2u * n); CHECK(snap->numConstraints() == (size_t)n);
  CHECK(snap->variableExists(st[0].Position().uuid()) && !graph.variableExists(st[0].Position().uuid()));
  CHECK(snap->constraintExists(odom[5]) && !graph.constraintExists(odom[5]));
  CHECK(!snap->variableExists(st[n + 3].Position().uuid()) && graph.variableExists(st[n + 3].Position().uuid()));
  CHECK(snap->getConnectedConstraints(st[0].Position().uuid()).size() == 2);           // prior + first odometry factor
  CHECK(graph.getConnectedConstraints(st[40].Position().uuid()).size() == 2);          // new prior + odometry 40 -> 41
  CHECK(snap->getConnectedConstraints(st[40].Position().uuid()).size() == 2);          // odometry 39 -> 40 and 40 -> 41
  CHECK(snap->getConstraints().size() == (size_t)n && graph.getConstraints().size() == graph.numConstraints());
  // both sides solve their own problem: chain anchored at x = 0 (snapshot) resp. x = 4.0 at state 40 (original)
  auto s_snap = snap->optimize();
  auto s_orig = graph.optimize();
  CHECK(s_snap.IsSolutionUsable() && s_orig.IsSolutionUsable());
  CHECK_NEAR(snap->getVariable(st[100].Position().uuid()).data()[0], 0.1 * 100, 1e-3);
  CHECK_NEAR(graph.getVariable(st[100].Position().uuid()).data()[0], 4.0 + 0.1 * 60, 1e-3);
  CHECK_NEAR(graph.getVariable(st[n + 29].Position().uuid()).data()[0], 4.0 + 0.1 * (n + 29 - 40), 1e-3);
  CHECK_NEAR(snap->getVariable(st[100].Position().uuid()).data()[0] - graph.getVariable(st[100].Position().uuid()).data()[0], 0.0, 1e-3);
  // a clone of the clone, mutated, leaves its parent alone; removing a variable that is still used keeps throwing
  auto snap2 = snap->clone();
  snap2->removeConstraint(odom[n - 2]);
  CHECK(snap->constraintExists(odom[n - 2]) && !snap2->constraintExists(odom[n - 2]));
  bool threw = false;
  try { snap2->removeVariable(st[7].Position().uuid()); } catch (const std::logic_error&) { threw = true; }
  CHECK(threw);
  snap.reset();                      // the parent goes first: the child keeps what it shares alive
  CHECK(snap2->numConstraints() == (size_t)n - 1);
  CHECK(snap2->getConnectedConstraints(st[n - 1].Position().uuid()).empty());
  // the lone last state is now unconstrained: the back-end refuses nothing (it is simply not moved), the rest solves
  snap2->removeVariable(st[n - 1].Position().uuid()); snap2->removeVariable(st[n - 1].Orientation().uuid());
  CHECK(snap2->optimize().IsSolutionUsable());
}

// Randomised bookkeeping test of GpuGraph (no counterpart in the reference, whose HashGraph is [EXT] fuse): a family of graphs —
// an original and clones of clones — takes random transactions; each is checked against a plain std::map model of what it
// should hold, and its packed factor tables (kept incrementally behind copy-on-write handles) against a graph rebuilt from
// scratch out of the model: same initial cost, same solve.
namespace {
struct ModelGraph {
  std::map<fuse_core::UUID, fuse_core::Variable::SharedPtr> vars;          // prototypes (values are read from the graph under test)
  std::map<fuse_core::UUID, fuse_core::Constraint::SharedPtr> cons;
  size_t usesOf(const fuse_core::UUID& v) const {
    size_t n = 0;
    for (const auto& c : cons) n += std::count(c.second->variables().begin(), c.second->variables().end(), v) ? 1 : 0;
    return n;
  }
};
}  // namespace
static void check_against_model(bs_optimizers::GpuGraph& g, const ModelGraph& m, const std::vector<fuse_core::UUID>& graveyard, bool solve) {
  CHECK(g.numVariables() == m.vars.size());
  CHECK(g.numConstraints() == m.cons.size());
  CHECK(g.getVariables().size() == m.vars.size());
  CHECK(g.orderedVariables().size() == m.vars.size());
  {
    std::set<fuse_core::UUID> have;
    for (const auto* c : g.getConstraints()) have.insert(c->uuid());
    CHECK(have.size() == m.cons.size());
    for (const auto& c : m.cons) CHECK(have.count(c.first) == 1 && g.constraintExists(c.first));
  }
  for (const auto& v : m.vars) {
    CHECK(g.variableExists(v.first));
    if (!g.variableExists(v.first)) continue;
    std::set<fuse_core::UUID> conn;
    for (const auto* c : g.getConnectedConstraints(v.first)) conn.insert(c->uuid());
    CHECK(conn.size() == m.usesOf(v.first));
    for (const auto& u : conn) {
      auto it = m.cons.find(u);
      CHECK(it != m.cons.end() && std::count(it->second->variables().begin(), it->second->variables().end(), v.first) > 0);
    }
  }
  for (const auto& u : graveyard) {
    if (!m.vars.count(u)) CHECK(!g.variableExists(u));
    if (!m.cons.count(u)) CHECK(!g.constraintExists(u));
  }
  {
    const auto ord = g.orderedVariables();   // keyframe-major, (q, p, v, ...) inside a keyframe
    for (size_t i = 1; i < ord.size(); ++i)
      CHECK(ord[i - 1]->stamp() < ord[i]->stamp() || (!(ord[i]->stamp() < ord[i - 1]->stamp()) && ord[i - 1]->stateSlot() <= ord[i]->stateSlot()));
  }
  if (!solve || m.cons.empty()) return;
  bs_optimizers::GpuGraph fresh;
  for (const auto& v : m.vars) {
    auto c = v.second->clone();
    std::memcpy(c->data(), g.getVariable(v.first).data(), c->size() * sizeof(double));
    fresh.addVariable(c);
  }
  for (auto it = m.cons.rbegin(); it != m.cons.rend(); ++it) fresh.addConstraint(it->second);   // some other order than the graph saw
  ceres_compat::SolverOptions o;
  o.max_num_iterations = 4;
  const auto a = g.optimize(o), b = fresh.optimize(o);
  CHECK(a.IsSolutionUsable() && b.IsSolutionUsable());
  CHECK_NEAR(a.initial_cost, b.initial_cost, 1e-12 * std::max(1.0, b.initial_cost));
  CHECK_NEAR(a.final_cost, b.final_cost, 1e-7 * std::max(1.0, b.initial_cost));
  for (const auto& v : m.vars)
    for (size_t k = 0; k < v.second->size(); ++k) CHECK_NEAR(g.getVariable(v.first).data()[k], fresh.getVariable(v.first).data()[k], 1e-6);
}
static void test_random_transactions_against_model() {
  std::printf("RandomTransactionsAgainstModel\n");
  std::mt19937 rng(20250620);
  std::normal_distribution<double> N(0.0, 1.0);
  auto U = [&](size_t n) { return (size_t)(rng() % n); };
  struct Member { bs_optimizers::GpuGraph::UniquePtr g; ModelGraph m; };
  std::vector<Member> family;
  family.push_back(Member{bs_optimizers::GpuGraph::UniquePtr(new bs_optimizers::GpuGraph()), ModelGraph()});
  std::vector<fuse_core::UUID> graveyard;
  int next_stamp = 0;
  const Mat<3, 3> c3 = 0.25 * I3();
  const Mat<6, 6> c6 = 0.25 * I6();
  auto pick_of_type = [&](const ModelGraph& m, const std::string& type, fuse_core::Variable::SharedPtr& out) {
    std::vector<fuse_core::Variable::SharedPtr> c;
    for (const auto& v : m.vars) if (v.second->type() == type) c.push_back(v.second);
    if (c.empty()) return false;
    out = c[U(c.size())];
    return true;
  };
  auto random_constraint = [&](const ModelGraph& m) -> fuse_core::Constraint::SharedPtr {
    using fuse_variables::VelocityLinear3DStamped; using fuse_variables::Position3DStamped; using fuse_variables::Orientation3DStamped;
    fuse_core::Variable::SharedPtr a, b;
    const Vec3 d{N(rng), N(rng), N(rng)};
    switch (U(4)) {
      case 0:
        if (!pick_of_type(m, "fuse_variables::VelocityLinear3DStamped", a)) return nullptr;
        return bs_constraints::AbsoluteVelocityLinear3DStampedConstraint("rnd" + std::to_string(rng()), static_cast<VelocityLinear3DStamped&>(*a), d, c3);
      case 1:
        if (!pick_of_type(m, "fuse_variables::VelocityLinear3DStamped", a) || !pick_of_type(m, "fuse_variables::VelocityLinear3DStamped", b) || a == b) return nullptr;
        return bs_constraints::RelativeVelocityLinear3DStampedConstraint("rnd" + std::to_string(rng()), static_cast<VelocityLinear3DStamped&>(*a),
                                                                         static_cast<VelocityLinear3DStamped&>(*b), d, c3);
      case 2: {
        if (!pick_of_type(m, "fuse_variables::Position3DStamped", a)) return nullptr;
        const auto q = m.vars.find(fuse_core::uuid::generate("fuse_variables::Orientation3DStamped", a->stamp()));
        if (q == m.vars.end()) return nullptr;
        return std::make_shared<fuse_constraints::AbsolutePose3DStampedConstraint>("rnd" + std::to_string(rng()), static_cast<Position3DStamped&>(*a),
                   static_cast<Orientation3DStamped&>(*q->second), bs_constraints::Vector7d{d[0], d[1], d[2], 1, 0, 0, 0}, c6);
      }
      default: {
        if (!pick_of_type(m, "fuse_variables::Position3DStamped", a) || !pick_of_type(m, "fuse_variables::Position3DStamped", b) || a == b) return nullptr;
        const auto qa = m.vars.find(fuse_core::uuid::generate("fuse_variables::Orientation3DStamped", a->stamp()));
        const auto qb = m.vars.find(fuse_core::uuid::generate("fuse_variables::Orientation3DStamped", b->stamp()));
        if (qa == m.vars.end() || qb == m.vars.end()) return nullptr;
        return std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("rnd" + std::to_string(rng()), static_cast<Position3DStamped&>(*a),
                   static_cast<Orientation3DStamped&>(*qa->second), static_cast<Position3DStamped&>(*b), static_cast<Orientation3DStamped&>(*qb->second),
                   bs_constraints::Vector7d{d[0], d[1], d[2], 1, 0, 0, 0}, c6);
      }
    }
  };
  const int rounds = 500;
  size_t max_vars = 0, max_cons = 0, n_clones = 0, n_drops = 0;
  for (int round = 0; round < rounds; ++round) {
    Member& mb = family[U(family.size())];
    // one random transaction: removals of constraints, of variables nothing uses any more, then additions
    fuse_core::Transaction tr;
    ModelGraph after = mb.m;
    const size_t n_rc = after.cons.empty() ? 0 : U(std::min<size_t>(after.cons.size(), after.cons.size() > 150 ? 40 : 5) + 1);
    for (size_t k = 0; k < n_rc && !after.cons.empty(); ++k) {
      auto it = after.cons.begin(); std::advance(it, U(after.cons.size()));
      tr.removeConstraint(it->first); graveyard.push_back(it->first); after.cons.erase(it);
    }
    {
      std::vector<fuse_core::UUID> unused;
      for (const auto& v : after.vars) if (after.usesOf(v.first) == 0) unused.push_back(v.first);
      for (const auto& u : unused) if (U(3) == 0) { tr.removeVariable(u); graveyard.push_back(u); after.vars.erase(u); }
    }
    const size_t n_av = U(7);
    for (size_t k = 0; k < n_av; ++k) {
      const fuse_core::Time stamp(0.01 * (U(2) && next_stamp > 4 ? (int)U(next_stamp) : next_stamp++));   // also between older stamps, also re-adds
      std::vector<fuse_core::Variable::SharedPtr> vs;
      if (U(2)) vs.push_back(fuse_variables::VelocityLinear3DStamped::make_shared(stamp));
      else { vs.push_back(fuse_variables::Orientation3DStamped::make_shared(stamp)); vs.push_back(fuse_variables::Position3DStamped::make_shared(stamp)); }
      for (auto& v : vs) {
        if (after.vars.count(v->uuid())) continue;        // (a re-add would overwrite the value: keep the model simple)
        if (v->size() == 3) for (int i = 0; i < 3; ++i) v->data()[i] = N(rng);
        tr.addVariable(v); after.vars[v->uuid()] = v;
      }
    }
    const size_t n_ac = U(16);
    for (size_t k = 0; k < n_ac; ++k) {
      auto c = random_constraint(after);
      if (!c) continue;
      tr.addConstraint(c); after.cons[c->uuid()] = c;
    }
    mb.g->update(tr);
    mb.m = std::move(after);
    max_vars = std::max(max_vars, mb.m.vars.size()); max_cons = std::max(max_cons, mb.m.cons.size());
    check_against_model(*mb.g, mb.m, graveyard, round % 9 == 0);
    // family events: clone (of any member), drop a member (parents before children, too), verify a bystander
    if (U(4) == 0 && family.size() < 5) {
      Member& src = family[U(family.size())];
      Member cl{src.g->clone(), src.m};
      family.push_back(std::move(cl));
      ++n_clones;
      check_against_model(*family.back().g, family.back().m, graveyard, false);
    } else if (U(7) == 0 && family.size() > 1) {
      family.erase(family.begin() + U(family.size()));
      ++n_drops;
    }
    Member& other = family[U(family.size())];
    check_against_model(*other.g, other.m, graveyard, round % 13 == 0);
    if (graveyard.size() > 400) graveyard.erase(graveyard.begin(), graveyard.begin() + 200);
    if (g_fail) { std::printf("  (first failure in round %d, family of %zu)\n", round, family.size()); return; }
  }
  std::printf("  %d transactions, %zu clones, %zu graphs dropped, largest graph %zu variables / %zu constraints\n", rounds, n_clones, n_drops, max_vars, max_cons);
  // removing a variable that is in use throws and leaves the graph as it was
  for (auto& mb : family) {
    if (mb.m.cons.empty()) continue;
    const auto& c = *mb.m.cons.begin()->second;
    bool threw = false;
    try { mb.g->removeVariable(c.variables()[0]); } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
    check_against_model(*mb.g, mb.m, graveyard, true);
  }
}


// A snapshot is read on ANOTHER thread (the publishers') while the graph it was taken from goes on with the next transaction
// (fixed_lag_smoother.cpp:308: notify(transaction, graph->clone()) and on to the next cycle).  The two share the packed tables through
// an undo log (gpu_graph.h TableUndo): the snapshot's first look at its tables rebuilds its version under the table's mutex while the
// owner is overwriting rows — checked against the model of what the snapshot must hold, 150 times.
static void test_snapshot_read_while_the_graph_moves_on() {
  std::printf("SnapshotReadWhileTheGraphMovesOn\n");
  std::mt19937 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  auto U = [&](size_t n) { return (size_t)(rng() % n); };
  using fuse_variables::VelocityLinear3DStamped;
  bs_optimizers::GpuGraph g;
  ModelGraph m;
  const Mat<3, 3> c3 = 0.25 * I3();
  std::vector<fuse_core::Variable::SharedPtr> vs;
  for (int k = 0; k < 120; ++k) {
    auto v = VelocityLinear3DStamped::make_shared(fuse_core::Time(0.01 * k));
    for (int i = 0; i < 3; ++i) v->data()[i] = N(rng);
    vs.push_back(v); g.addVariable(v->clone()); m.vars[v->uuid()] = v;
  }
  int serial = 0;
  auto new_constraint = [&]() -> fuse_core::Constraint::SharedPtr {
    const size_t a = U(vs.size()), b = U(vs.size());
    const Vec3 d{N(rng), N(rng), N(rng)};
    if (a == b) return bs_constraints::AbsoluteVelocityLinear3DStampedConstraint("t" + std::to_string(serial++), static_cast<VelocityLinear3DStamped&>(*vs[a]), d, c3);
    return bs_constraints::RelativeVelocityLinear3DStampedConstraint("t" + std::to_string(serial++), static_cast<VelocityLinear3DStamped&>(*vs[a]),
                                                                     static_cast<VelocityLinear3DStamped&>(*vs[b]), d, c3);
  };
  for (int k = 0; k < 900; ++k) { auto c = new_constraint(); g.addConstraint(c); m.cons[c->uuid()] = c; }
  const std::vector<fuse_core::UUID> none;
  for (int round = 0; round < 150 && !g_fail; ++round) {
    auto snap = g.clone();
    const ModelGraph snap_model = m;
    fuse_core::Transaction tr;
    for (int k = 0; k < 40 && !m.cons.empty(); ++k) {
      auto it = m.cons.begin(); std::advance(it, U(m.cons.size()));
      tr.removeConstraint(it->first); m.cons.erase(it);
    }
    for (int k = 0; k < 45; ++k) { auto c = new_constraint(); tr.addConstraint(c); m.cons[c->uuid()] = c; }
    std::thread reader([&]() { check_against_model(*snap, snap_model, none, round % 25 == 0); });
    g.update(tr);
    reader.join();
    if (round % 10 == 0) check_against_model(g, m, none, true);
    if (round % 3 == 0) { auto snap2 = snap->clone(); check_against_model(*snap2, snap_model, none, false); }   // a clone of a snapshot that has its own tables by now
  }
  check_against_model(g, m, none, true);
}

// One snapshot, SEVERAL publisher threads (fixed_lag_smoother.cpp:308 hands the clone to every publisher as a const graph): the first
// constraintExists / getConnectedConstraints on it builds the uuid index, the connectivity and the snapshot's own tables lazily behind
// const accessors — four readers start on a fresh snapshot at once while the owner moves on; every reader must see the model.
static void test_one_snapshot_many_readers() {
  std::printf("OneSnapshotManyReaders\n");
  std::mt19937 rng(11);
  std::normal_distribution<double> N(0.0, 1.0);
  auto U = [&](size_t n) { return (size_t)(rng() % n); };
  using fuse_variables::VelocityLinear3DStamped;
  bs_optimizers::GpuGraph g;
  const Mat<3, 3> c3 = 0.25 * I3();
  std::vector<fuse_core::Variable::SharedPtr> vs;
  for (int k = 0; k < 200; ++k) {
    auto v = VelocityLinear3DStamped::make_shared(fuse_core::Time(0.01 * k));
    for (int i = 0; i < 3; ++i) v->data()[i] = N(rng);
    vs.push_back(v); g.addVariable(v->clone());
  }
  int serial = 0;
  std::vector<fuse_core::Constraint::SharedPtr> cons;
  std::vector<std::pair<size_t, size_t>> ends;
  auto new_constraint = [&]() {
    size_t a = U(vs.size()), b = U(vs.size());
    if (a == b) b = (a + 1) % vs.size();
    const Vec3 d{N(rng), N(rng), N(rng)};
    auto c = bs_constraints::RelativeVelocityLinear3DStampedConstraint("m" + std::to_string(serial++), static_cast<VelocityLinear3DStamped&>(*vs[a]),
                                                                       static_cast<VelocityLinear3DStamped&>(*vs[b]), d, c3);
    ends.emplace_back(a, b);
    return c;
  };
  for (int k = 0; k < 3000; ++k) { auto c = new_constraint(); g.addConstraint(c); cons.push_back(c); }
  for (int round = 0; round < 40 && !g_fail; ++round) {
    std::shared_ptr<const bs_optimizers::GpuGraph> snap(g.clone().release());
    const size_t n_cons = cons.size();
    std::vector<size_t> degree(vs.size(), 0);
    for (size_t i = 0; i < n_cons; ++i) { ++degree[ends[i].first]; ++degree[ends[i].second]; }
    std::atomic<int> bad{0};
    std::vector<std::thread> readers;
    for (int t = 0; t < 4; ++t)
      readers.emplace_back([&, t]() {
        for (size_t i = t; i < n_cons; i += 3) if (!snap->constraintExists(cons[i]->uuid())) ++bad;
        for (size_t v = t; v < vs.size(); v += 2) if (snap->getConnectedConstraints(vs[v]->uuid()).size() != degree[v]) ++bad;
        if (snap->numConstraints() != n_cons) ++bad;
      });
    // the owner moves on underneath: new constraints the snapshot must not see
    fuse_core::Transaction tr;
    std::vector<fuse_core::Constraint::SharedPtr> fresh;
    for (int k = 0; k < 60; ++k) { auto c = new_constraint(); tr.addConstraint(c); fresh.push_back(c); }
    g.update(tr);
    for (auto& th : readers) th.join();
    for (auto& c : fresh) if (snap->constraintExists(c->uuid())) ++bad;
    CHECK(bad.load() == 0);
    for (auto& c : fresh) cons.push_back(c);
  }
}

// the pending-transaction queue rules of fixed_lag_smoother.cpp:335-477 (processQueue) and :548-627 (transactionCallback): ignition,
// purge of pre-ignition transactions, transactions older than the lag window, motion-model failure -> retry until transaction_timeout
static void test_process_queue_rules() {
  std::printf("ProcessQueueRules\n");
  using FLS = bs_optimizers::FixedLagSmoother;
  auto make_tr = [](double t, double x) {   // one stamped position with an absolute prior: a transaction that can be optimised on its own
    auto tr = std::make_shared<fuse_core::Transaction>();
    tr->stamp(fuse_core::Time(t)); tr->addInvolvedStamp(fuse_core::Time(t));
    auto p = fuse_variables::Position3DStamped::make_shared(fuse_core::Time(t));
    p->x() = x; p->y() = 0; p->z() = 0;
    tr->addVariable(p);
    Mat<3, 3> cov = 1e-2 * Mat<3, 3>::Identity();
    tr->addConstraint(std::make_shared<fuse_constraints::AbsoluteVec3Constraint>("fuse_constraints::AbsolutePosition3DStampedConstraint", "src", *p,
                                                                                 Vec3{x + 0.5, 0, 0}, cov));
    return tr;
  };
  bs_optimizers::FixedLagSmootherParams params;
  params.lag_duration = 1.0; params.transaction_timeout = 0.25; params.pseudo_marginalization = false;
  {  // (1) with an ignition sensor registered nothing happens before its first transaction; older transactions of others are purged
    FLS s(bs_optimizers::GpuGraph::make_unique(), params);
    s.registerSensorModel("slam_init", true); s.registerSensorModel("vo", false);
    s.transactionCallback("vo", make_tr(0.10, 1.0));
    s.transactionCallback("vo", make_tr(0.20, 2.0));
    CHECK(!s.started()); CHECK(s.pendingTransactions() == 2);
    CHECK(s.optimizeOnce() == FLS::CycleResult::NothingToDo);               // optimizerTimerCallback: not started
    s.transactionCallback("vo", make_tr(0.60, 3.0));                         // purge_time = 0.60 - 0.25: the 0.10 and 0.20 ones go
    CHECK(s.pendingTransactions() == 1);
    s.transactionCallback("slam_init", make_tr(1.00, 4.0));                  // ignition: start time 1.00, the 0.60 one is older -> purged
    CHECK(s.started()); CHECK(s.pendingTransactions() == 1);
    s.transactionCallback("vo", make_tr(0.90, 9.0));                         // before the start time: ignored (:552-561)
    CHECK(s.pendingTransactions() == 1);
    s.transactionCallback("vo", make_tr(1.10, 5.0));
    s.transactionCallback("vo", make_tr(1.20, 6.0));
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);                  // the ignition transaction alone (:344-356)
    CHECK(s.graph().numVariables() == 1); CHECK(s.pendingTransactions() == 2);
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);                  // then the rest together
    CHECK(s.graph().numVariables() == 3); CHECK(s.pendingTransactions() == 0);
  }
  {  // (2) an ignition transaction whose motion models fail is dropped; without another one the optimizer un-starts (:380-404)
    FLS s(bs_optimizers::GpuGraph::make_unique(), params);
    s.registerSensorModel("slam_init", true); s.registerSensorModel("vo", false);
    bool fail_ignition = true;
    s.setMotionModelCallback([&](const std::string& sensor, fuse_core::Transaction&) { return !(fail_ignition && sensor == "slam_init"); });
    s.transactionCallback("slam_init", make_tr(1.00, 1.0));
    s.transactionCallback("vo", make_tr(1.10, 2.0));
    CHECK(s.started());
    CHECK(s.optimizeOnce() == FLS::CycleResult::NothingToDo);
    CHECK(!s.started()); CHECK(s.numQueueErrors() == 1); CHECK(s.pendingTransactions() == 1);
    // ... with a second ignition transaction queued, everything older than IT is purged and it is tried next cycle (:405-414)
    s.transactionCallback("slam_init", make_tr(1.20, 3.0));                  // re-ignites; the 1.10 "vo" one is older -> purged at once
    CHECK(s.started()); CHECK(s.pendingTransactions() == 1);
    s.transactionCallback("slam_init", make_tr(1.30, 4.0));
    CHECK(s.optimizeOnce() == FLS::CycleResult::NothingToDo);                // 1.20 fails again; 1.30 is the next ignition transaction
    CHECK(s.started()); CHECK(s.pendingTransactions() == 1);
    fail_ignition = false;
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);
    CHECK(s.graph().numVariables() == 1); CHECK(s.pendingTransactions() == 0);
  }
  {  // (3) no ignition sensor: autostart; a motion-model failure blocks the LATER transactions of that sensor only and is retried
     //     until transaction_timeout; transactions older than the lag window are dropped (:424-476)
    FLS s(bs_optimizers::GpuGraph::make_unique(), params);
    s.registerSensorModel("vo", false); s.registerSensorModel("lo", false);
    double imu_covers_until = 2.05;   // the "IMU buffer": a motion model can be generated up to this stamp
    s.setMotionModelCallback([&](const std::string& sensor, fuse_core::Transaction& t) {
      if (sensor == "vo" && std::fabs(t.maxStamp().toSec() - 2.10) < 1e-9) return false;   // falls into a gap of the IMU stream: never possible
      return t.maxStamp().toSec() <= imu_covers_until;
    });
    s.transactionCallback("vo", make_tr(2.00, 1.0));
    s.transactionCallback("vo", make_tr(2.10, 2.0));                         // no motion model possible
    s.transactionCallback("vo", make_tr(2.04, 3.0));                         // covered, but queued behind nothing: processed (older than 2.10)
    s.transactionCallback("lo", make_tr(2.02, 4.0));
    CHECK(s.started());
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);
    CHECK(s.graph().numVariables() == 3); CHECK(s.pendingTransactions() == 1);   // 2.10 waits
    s.transactionCallback("vo", make_tr(2.20, 5.0));                         // behind the blocked 2.10 of the same sensor: must wait too
    s.transactionCallback("lo", make_tr(2.21, 6.0));
    imu_covers_until = 2.05;
    const auto r = s.optimizeOnce();
    CHECK(r == FLS::CycleResult::NothingToDo);                               // vo blocked at 2.10; lo's 2.21 not covered either (and not timed out)
    CHECK(s.pendingTransactions() == 3); CHECK(s.numTimedOutTransactions() == 0);
    s.transactionCallback("lo", make_tr(2.40, 7.0));                         // current time 2.40: 2.10 + 0.25 < 2.40 -> the 2.10 one has timed out
    imu_covers_until = 2.30;
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);
    CHECK(s.numTimedOutTransactions() == 1);
    CHECK(s.graph().numVariables() == 5);                                    // + 2.20, 2.21
    CHECK(s.pendingTransactions() == 1);                                     // 2.40 still waits for the IMU
    // lag window: the graph's newest stamp is 2.21, lag 1.0 -> expiration 1.21; a late transaction about t = 1.0 is dropped
    s.transactionCallback("vo", make_tr(1.00, 8.0));
    imu_covers_until = 10.0;
    CHECK(s.optimizeOnce() == FLS::CycleResult::Optimized);
    CHECK(s.numExpiredTransactions() == 1);
    CHECK(s.pendingTransactions() == 0);
  }
}

int main() {
  test_block_order_and_pack();
  test_simple_2_state_fg();
  test_absolute_imu_state();
  test_inverse_depth_window();
  test_true_marginalization_linear_chain();
  test_clone_is_an_independent_snapshot();
  test_snapshot_read_while_the_graph_moves_on();
  test_one_snapshot_many_readers();
  test_random_transactions_against_model();
  test_process_queue_rules();
  test_fixed_lag_smoother_window(true);
  test_fixed_lag_smoother_window(false);
  if (g_fail) { std::printf("FAILED: %d checks\n", g_fail); return 1; }
  std::printf("ALL HOST TESTS PASSED\n");
  return 0;
}
