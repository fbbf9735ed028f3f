// The trust-region decision the device takes (beam_slam_amd/csrc/lm_decide.h lm_decide) against the one copy of the loop the host runs
// (lm_state.h LmState::advance; [EXT] ceres TrustRegionMinimizer / LevenbergMarquardtStrategy, beam_slam_launch/config/vio.yaml:7-17): on
// random step scalars — accepted, rejected, invalid, every tolerance — the device says "accepted" exactly when LmState requests STEP_ACCEPT
// with a full step, and names the SAME radius bit for bit (so the assembly issued ahead is adopted, bsgpu_solve.cpp enqueue_step); and the
// cube both of them use (lm_cube: rounded once from two exact products) is within an ulp of libm's pow(t, 3) — Ceres' expression — everywhere
// and equal to it in all but a fraction of a percent of arguments (glibc's pow is not always correctly rounded; the long-double product is).
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I beam_slam_amd/csrc tests/plan/test_lm_decide.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "lm_state.h"
#include "lm_decide.h"

using namespace bsg;

static uint64_t bits(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }

int main() {
  std::mt19937_64 rng(20250930);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  auto logu = [&](double lo, double hi) { return std::exp(std::log(lo) + U(rng) * (std::log(hi) - std::log(lo))); };
  // 1. the cube
  long cube_bad = 0, cube_diff = 0, cube_vs_ld = 0;
  for (int i = 0; i < 4000000; ++i) {
    const double t = i % 3 == 0 ? 2.0 * U(rng) - 1.0 : (i % 3 == 1 ? 1.0 - logu(1e-16, 1.0) : -1.0 + logu(1e-16, 2.0));
    const double c = lm_cube(t), pw = std::pow(t, 3);
    if (bits(c) != bits(pw)) { ++cube_diff; if (std::fabs(c - pw) > std::fabs(std::nextafter(pw, 2 * pw) - pw)) ++cube_bad; }
    const long double tl = t;   // (64-bit mantissa: t^3 to 2^-62, then one rounding — a double rounding only in ~2^-9 of the near-ties)
    if (bits(c) != bits((double)(tl * tl * tl))) ++cube_vs_ld;
  }
  std::printf("cube: %ld of 4000000 differ from pow(t, 3) (all within an ulp: %s), %ld from the long-double product\n", cube_diff, cube_bad ? "NO" : "yes", cube_vs_ld);
  if (cube_diff > 8000 || cube_vs_ld > 8000) cube_bad += 1;
  // 2. the decision
  long n_accept = 0, n_other = 0, bad = 0;
  const int N = 2000000;
  for (int i = 0; i < N; ++i) {
    bsgpu_options o;
    std::memset(&o, 0, sizeof(o));
    o.max_num_iterations = 50; o.initial_trust_region_radius = 1e4; o.max_trust_region_radius = i % 7 == 0 ? logu(1e3, 1e8) : 1e16; o.min_trust_region_radius = 1e-32;
    o.min_relative_decrease = 1e-3; o.function_tolerance = i % 5 == 0 ? 1e-3 : 1.5e-7; o.gradient_tolerance = i % 11 == 0 ? 1e-2 : 1.5e-7; o.parameter_tolerance = i % 13 == 0 ? 1e-3 : 1.5e-7;
    o.max_num_consecutive_invalid_steps = 5; o.max_solver_time_in_seconds = 0.0;
    bsgpu_summary sum;
    std::vector<bsgpu_iteration> iters;
    LmState lm;
    lm.start(&o, &sum, &iters, 10, 10, 0);
    // a first step, then the step under test (kind: what the driver was asked to compute — an accepted point or a retry after a rejection)
    double h[SC_NUM] = {0};
    const double x_cost0 = logu(1e-2, 1e8);
    h[SC_COST_X] = x_cost0; h[SC_GRAD_MAX] = logu(1e-3, 1e6); h[SC_GRAD_NORM2] = 1.0;
    auto draw_step = [&](double x_cost) {
      const int kind = (int)(U(rng) * 10);
      const double mcc = kind == 0 ? -logu(1e-9, 1.0) : x_cost * logu(1e-9, 0.9);
      const double rho = kind == 1 ? -U(rng) : (kind == 2 ? 1e-3 * U(rng) * 2 : (kind == 3 ? 1.0 + 1e-3 * (U(rng) - 0.5) : U(rng) * 1.3));
      h[SC_MCC] = kind == 4 ? std::nan("") : mcc;
      h[SC_COST_CAND] = kind == 5 ? INFINITY : x_cost - rho * mcc;
      h[SC_STEP_NORM2] = logu(1e-20, 1e2); h[SC_X_NORM2] = logu(1e-2, 1e6);
      h[SC_CHOL_FAIL] = kind == 6 ? 1.0 : 0.0;
    };
    draw_step(x_cost0);
    lm.begin(h, 0.0, false);
    for (int step = 0; step < 3 && !lm.done; ++step) {
      // the state BEFORE the decision of the step the driver computes now (what solve() hands to enqueue_step)
      LmDecide d;
      d.on = 1; d.radius = lm.radius; d.check_grad = lm.kind != STEP_REJECT ? 1 : 0;
      d.min_relative_decrease = o.min_relative_decrease; d.max_radius = o.max_trust_region_radius; d.function_tolerance = o.function_tolerance;
      d.parameter_tolerance = o.parameter_tolerance; d.gradient_tolerance = o.gradient_tolerance;
      const bool stale = U(rng) < 0.5;   // the cost at this step's point: held by the host (cand_cost) or SC_COST_X of this reduction — the same sum
      const double x_cost = lm.kind == STEP_REJECT ? lm.x_cost : lm.cand_cost;
      d.x_from_scal = (lm.kind != STEP_REJECT && !stale) ? 1 : 0;
      d.x_cost = x_cost;
      h[SC_COST_X] = x_cost;
      h[SC_GRAD_MAX] = U(rng) < 0.05 ? 1e-9 : logu(1e-3, 1e6);
      draw_step(x_cost);
      LmScal v;
      v.mcc = h[SC_MCC]; v.sn2 = h[SC_STEP_NORM2]; v.xn2 = h[SC_X_NORM2]; v.cand = h[SC_COST_CAND]; v.cost_x = h[SC_COST_X]; v.gmax = h[SC_GRAD_MAX]; v.chol_fail = h[SC_CHOL_FAIL];
      double r_dev = 0.0;
      const int go = lm_decide(d, v, &r_dev);
      const int it_before = lm.it.iteration;
      lm.advance(h, stale && lm.kind != STEP_REJECT, false);
      const bool host_accept = !lm.done && lm.kind == STEP_ACCEPT && !lm.grad_only && lm.it.iteration == it_before + 1;
      if (go) { ++n_accept; if (!host_accept || bits(r_dev) != bits(lm.radius)) { if (++bad <= 10) std::printf("MISMATCH accept: host_accept %d radius host %.17g device %.17g\n", (int)host_accept, lm.radius, r_dev); } }
      else { ++n_other; if (host_accept) { if (++bad <= 10) std::printf("MISMATCH: host accepts (radius %.17g), device does not\n", lm.radius); } }
    }
  }
  std::printf("decisions: %ld accepted, %ld other, %ld mismatches\n", n_accept, n_other, bad);
  std::printf(bad == 0 && cube_bad == 0 ? "all ok\n" : "FAILED\n");
  return bad == 0 && cube_bad == 0 ? 0 : 1;
}
