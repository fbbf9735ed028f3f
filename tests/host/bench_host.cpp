// Times one optimisation cycle of the host mirror at the size of BASELINE config C2 (200 keyframes, 50 000 landmarks,
// ~400 000 reprojection constraints): GpuGraph::flatten (ordering + pack + C-ABI hand-over) vs the device solve.
// Run on the GPU box:  g++ -O2 -std=c++17 tests/host/bench_host.cpp -Lbeam_slam_amd/csrc -lbsgpu ... (scripts/bench_host.sh)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <map>
#include <random>

#include "../../beam_slam_amd/host/fixed_lag_smoother.h"

using namespace bs_math;
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char** argv) {
  const int n_kf = argc > 1 ? atoi(argv[1]) : 200, n_lm = argc > 2 ? atoi(argv[2]) : 50000;
  const bool host_only = argc > 3 && std::string(argv[3]).rfind("host-only", 0) == 0;
  const bool no_snapshot = argc > 3 && std::string(argv[3]) == "host-only-nosnap";   // clone / update timings without a solve (any back-end)
  // "lio": a lidar-inertial window (lio.yaml:2: tens of key frames, no landmarks) — n_lm is then the number of scan-registration relative-pose
  // factors a key frame takes part in (against the key frames before it); cycles: argv[4] (the small windows are measured over more of them)
  const bool lio = (argc > 3 && std::string(argv[3]) == "lio") || (argc > 4 && std::string(argv[4]) == "lio");
  const int reps = n_kf <= 50 ? 41 : 6;
  // what a slide by one key frame brings in, scaled with the window (C2: 250 new landmarks seen from the last 4 key frames + 1000 observations of recent ones)
  const int new_lm = lio ? 0 : std::max(1, n_lm / n_kf), re_obs = 4 * new_lm;
  std::mt19937 rng(1);
  std::normal_distribution<double> N(0.0, 1.0);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  Mat<4, 4> T = Mat<4, 4>::Identity();
  Mat<3, 3> K = Mat<3, 3>::Identity(); K(0, 0) = 458.654; K(1, 1) = 457.296; K(0, 2) = 367.215; K(1, 2) = 248.375;
  bs_optimizers::GpuGraph graph;
  const auto t0 = clk::now();
  std::vector<bs_common::ImuState> st;
  for (int k = 0; k < n_kf; ++k) {
    bs_common::ImuState s(fuse_core::Time(0.1 * k), {1, 0, 0, 0}, {0.1 * k + 0.01 * N(rng), 0.01 * N(rng), 0.01 * N(rng)}, {1.0, 0, 0});
    graph.addVariable(s.Orientation().clone()); graph.addVariable(s.Position().clone()); graph.addVariable(s.Velocity().clone());
    graph.addVariable(s.GyroBias().clone()); graph.addVariable(s.AccelBias().clone());
    st.push_back(s);
  }
  Mat<15, 15> cov = 1e-4 * Mat<15, 15>::Identity();
  graph.addConstraint(std::make_shared<bs_constraints::AbsoluteImuState3DStampedConstraint>("prior", st[0], st[0].GetStateVector(), cov));
  Mat<6, 6> c6 = 1e-2 * Mat<6, 6>::Identity();
  for (int k = 0; k + 1 < n_kf; ++k)   // stand-in for the IMU chain: relative poses keep the window connected
    graph.addConstraint(std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("odom", st[k].Position(), st[k].Orientation(), st[k + 1].Position(),
                                                                                          st[k + 1].Orientation(), bs_constraints::Vector7d{0.1, 0, 0, 1, 0, 0, 0}, c6));
  size_t n_obs = 0;
  std::map<fuse_core::UUID, int> lm_obs;   // observations a landmark still has: it leaves the graph with its last one (as visual_odometry.cpp prunes its map)
  if (lio)
    for (int k = 1; k < n_kf; ++k)
      for (int d = 1; d <= n_lm && k - d >= 0; ++d)
        graph.addConstraint(std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("scan", st[k - d].Position(), st[k - d].Orientation(), st[k].Position(),
                                                                                              st[k].Orientation(), bs_constraints::Vector7d{0.1 * d, 0, 0, 1, 0, 0, 0}, c6));
  std::vector<int> lm_last_kf(lio ? 0 : n_lm, 0);
  for (int j = 0; j < (lio ? 0 : n_lm); ++j) {
    const int len = 4 + (int)(U(rng) * 9), k0 = (int)(U(rng) * (n_kf - len));
    const double z = 4.0 + 8.0 * U(rng), x0 = 0.1 * k0 + (U(rng) - 0.3) * 0.8 * z, y = (U(rng) - 0.5) * 0.6 * z;
    auto lm = bs_variables::Point3DLandmark::make_shared(j);
    lm->x() = x0 + 0.05 * N(rng); lm->y() = y + 0.05 * N(rng); lm->z() = z + 0.05 * N(rng);
    graph.addVariable(lm);
    lm_last_kf[j] = k0 + len - 1;
    for (int k = k0; k < k0 + len; ++k) {
      const double px = x0 - 0.1 * k, u = K(0, 0) * px / z + K(0, 2), v = K(1, 1) * y / z + K(1, 2);
      auto c = std::make_shared<bs_constraints::EuclideanReprojectionConstraint>("vo", st[k].Orientation(), st[k].Position(), *lm, T, K,
                                                                              std::array<double, 2>{u + N(rng), v + N(rng)}, 1.0);
      c->loss(std::make_shared<fuse_loss::CauchyLoss>(5.0));
      graph.addConstraint(c);
      ++lm_obs[lm->uuid()];
      ++n_obs;
    }
  }
  const auto t1 = clk::now();
  std::printf("graph: %zu variables, %zu constraints (%zu reprojection) built in %.0f ms\n", graph.numVariables(), graph.numConstraints(), n_obs, ms(t0, t1));
  auto opts = ceres_compat::SolverOptions::Vio();
  opts.max_solver_time_in_seconds = 1e9;
  bs_optimizers::GpuGraph::UniquePtr snapshot;   // what the publishers hold: the previous cycle's clone stays alive across the next update
  std::vector<uint64_t> recent_lm;   // landmarks the newest keyframes see: the ones a new keyframe can still observe
  for (size_t j = 0; j < lm_last_kf.size(); ++j) if (lm_last_kf[j] >= n_kf - 4) recent_lm.push_back(j);
  uint64_t next_lm = lio ? 0 : n_lm;
  int oldest = 0;
  std::vector<double> t_opt, t_backend, t_clone, t_release, t_update;
  for (int rep = 0; rep < reps; ++rep) {
    const auto a = clk::now();
    ceres_compat::SolverSummary s;
    if (!host_only) s = graph.optimize(opts);
    const auto b = clk::now();
    const auto& bs = graph.lastBackendSummary();
    if (!host_only && (reps <= 6 || rep % 10 == 0)) std::printf("cycle %d: optimize() %.1f ms total | back-end (finalize + solve) %.1f ms (%d it) | host flatten + hand-over %.1f ms | cost %.4e -> %.4e\n", rep,
                ms(a, b), 1e3 * bs.total_time_in_seconds, bs.num_iterations, ms(a, b) - 1e3 * bs.total_time_in_seconds, s.initial_cost, s.final_cost);
    const auto c0 = clk::now();
    auto fresh = graph.clone();   // what fixed_lag_smoother.cpp:308 does every cycle for the publishers
    const auto c0b = clk::now();
    snapshot = std::move(fresh);  // (the publishers let go of the previous one)
    if (no_snapshot) snapshot.reset();
    const auto c1 = clk::now();
    // the window slides by one keyframe: the oldest state and everything attached to it leave, a new keyframe with
    // 250 new landmarks (seen from the last 4 keyframes) and 1000 observations of recent landmarks enters
    fuse_core::Transaction tr;
    for (const auto* c : graph.getConnectedConstraints(st[oldest].Orientation().uuid())) {
      tr.removeConstraint(c->uuid());
      if (c->variables().size() == 3) { auto it = lm_obs.find(c->variables()[2]); if (it != lm_obs.end() && --it->second == 0) { tr.removeVariable(it->first); lm_obs.erase(it); } }
    }
    for (const auto* c : graph.getConnectedConstraints(st[oldest].Velocity().uuid())) tr.removeConstraint(c->uuid());
    tr.removeVariable(st[oldest].Orientation().uuid()); tr.removeVariable(st[oldest].Position().uuid()); tr.removeVariable(st[oldest].Velocity().uuid());
    tr.removeVariable(st[oldest].GyroBias().uuid()); tr.removeVariable(st[oldest].AccelBias().uuid());
    ++oldest;
    const int k = (int)st.size();
    st.emplace_back(fuse_core::Time(0.1 * k), std::array<double, 4>{1, 0, 0, 0}, std::array<double, 3>{0.1 * k, 0, 0}, std::array<double, 3>{1.0, 0, 0});
    tr.addVariable(st[k].Orientation().clone()); tr.addVariable(st[k].Position().clone()); tr.addVariable(st[k].Velocity().clone());
    tr.addVariable(st[k].GyroBias().clone()); tr.addVariable(st[k].AccelBias().clone());
    tr.addConstraint(std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("odom", st[k - 1].Position(), st[k - 1].Orientation(), st[k].Position(),
                                                                                       st[k].Orientation(), bs_constraints::Vector7d{0.1, 0, 0, 1, 0, 0, 0}, c6));
    auto observe = [&](const bs_variables::Point3DLandmark& lm, double X, double Y, double Z, int kf) {
      const double px = X - 0.1 * kf, u = K(0, 0) * px / Z + K(0, 2), v = K(1, 1) * Y / Z + K(1, 2);
      auto c = std::make_shared<bs_constraints::EuclideanReprojectionConstraint>("vo", st[kf].Orientation(), st[kf].Position(), lm, T, K,
                                                                              std::array<double, 2>{u + N(rng), v + N(rng)}, 1.0);
      c->loss(std::make_shared<fuse_loss::CauchyLoss>(5.0));
      tr.addConstraint(c);
      ++lm_obs[lm.uuid()];
    };
    if (lio)
      for (int d = 1; d <= n_lm && k - d >= oldest; ++d)
        tr.addConstraint(std::make_shared<fuse_constraints::RelativePose3DStampedConstraint>("scan", st[k - d].Position(), st[k - d].Orientation(), st[k].Position(),
                                                                                           st[k].Orientation(), bs_constraints::Vector7d{0.1 * d, 0, 0, 1, 0, 0, 0}, c6));
    for (int j = 0; j < new_lm; ++j) {
      const double z = 4.0 + 8.0 * U(rng), x0 = 0.1 * k + (U(rng) - 0.3) * 0.8 * z, y = (U(rng) - 0.5) * 0.6 * z;
      auto lm = bs_variables::Point3DLandmark::make_shared(next_lm++);
      lm->x() = x0; lm->y() = y; lm->z() = z;
      tr.addVariable(lm);
      for (int kf = k - 3; kf <= k; ++kf) observe(*lm, x0, y, z, kf);
      recent_lm.push_back(lm->id());
    }
    for (int j = 0; j < re_obs && !recent_lm.empty(); ++j) {
      const uint64_t id = recent_lm[recent_lm.size() - 1 - (size_t)(U(rng) * std::min<size_t>(10 * (size_t)new_lm, recent_lm.size()))];
      const auto& lm = static_cast<const bs_variables::Point3DLandmark&>(graph.variableExists(bs_variables::Point3DLandmark(id).uuid())
                                                                         ? graph.getVariable(bs_variables::Point3DLandmark(id).uuid())
                                                                         : *tr.addedVariables()[5 + (id - (next_lm - new_lm))]);
      observe(lm, lm.data()[0], lm.data()[1], lm.data()[2], k);
    }
    const auto c2 = clk::now();
    graph.update(tr);
    const auto c3 = clk::now();
    if (rep > 0) { t_opt.push_back(ms(a, b)); t_backend.push_back(1e3 * bs.total_time_in_seconds); t_clone.push_back(ms(c0, c0b)); t_release.push_back(ms(c0b, c1)); t_update.push_back(ms(c2, c3)); }
    if (reps <= 6 || rep % 10 == 0)
    std::printf("         Graph::clone() %.1f ms + %.1f ms releasing the previous snapshot | transaction built in %.1f ms (-%zu +%zu constraints) | Graph::update() %.1f ms with the snapshot alive -> %zu constraints\n",
                ms(c0, c0b), ms(c0b, c1), ms(c1, c2), tr.removedConstraints().size(), tr.addedConstraints().size(), ms(c2, c3), graph.numConstraints());
  }
  // one line for bench.py (other_configs.host_cycle): medians over the cycles after the first (which pays the context's allocations)
  auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double o = med(t_opt), be = med(t_backend), cl = med(t_clone), re = med(t_release), up = med(t_update);
  char what[160];
  if (lio) std::snprintf(what, sizeof what, "lidar-inertial window, %d key frames, %d scan-registration factors per key frame, the window slides by one key frame per cycle", n_kf, n_lm);
  else std::snprintf(what, sizeof what, "%d key frames x %d landmarks, the window slides by one key frame per cycle (+%d landmarks, +%d constraints)", n_kf, n_lm, new_lm, 1 + 4 * new_lm + re_obs);
  std::printf("HOST_CYCLE_JSON {\"workload\": \"%s, previous snapshot alive\", "
              "\"host_cycle_ms\": %.3f, \"optimize_ms\": %.3f, \"of_which_backend_finalize_and_solve_ms\": %.3f, \"clone_ms\": %.3f, \"release_previous_snapshot_ms\": %.3f, "
              "\"update_ms\": %.3f, \"cycles\": %zu, \"reference\": \"fixed_lag_smoother.cpp:220,274,281,308\"}\n",
              what, o + cl + re + up, o, be, cl, re, up, t_opt.size());
  return 0;
}
