// The trust-region state machine of one window: [EXT] ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy restated
// (the reference configures it in beam_slam_launch/config/vio.yaml:7-17 and calls it through graph_->optimize(),
// bs_optimizers/src/fixed_lag_smoother.cpp:281), cut at the points where the driver enqueues a step on the device and waits for its
// scalars.  ONE copy: bsgpu_solve.cpp's solve() drives it for a lone window (a batch of one), bsgpu_batch.cpp's solve_batched() drives
// one per window of a batch — a fix to the loop is made here, once.
//
// Protocol: start() -> the driver computes STEP_FIRST -> begin(h) -> while (!done): the driver computes the requested step (kind, radius,
// grad_only) -> advance(h).  `h` = the step's scalars as the end-of-step reduction mirrors them (SC_*).  retry_timeout: a wait inside a
// single-launch kernel timed out (the GPU is shared) — the driver recomputes the SAME step another way (REQ: kind = STEP_REJECT at the
// same radius) and calls advance() again, or gives the window up (batch: it is solved again alone).
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/bsgpu.h"
#include "bsgpu_internal.h"
#include "lm_decide.h"

namespace bsg {

enum StepKind { STEP_FIRST = 0, STEP_ACCEPT = 1, STEP_REJECT = 2 };

struct LmState {
  const bsgpu_options* o = nullptr;
  bsgpu_summary* sum = nullptr;
  std::vector<bsgpu_iteration>* iters = nullptr;
  double radius = 0.0, decrease_factor = 2.0, x_cost = 0.0, fixed = 0.0, cand_cost = 0.0;
  bsgpu_iteration it{};
  int num_consecutive_invalid = 0;
  const char* msg = "";
  bool done = false, retry_timeout = false, absorb = false;
  int kind = STEP_FIRST;      // the step the driver is asked to compute next
  bool grad_only = false;     // ... without a linear solve: the iteration budget is used up, only the accepted point's cost and gradient are wanted
  std::chrono::steady_clock::time_point t_start;

  double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); }
  void request(int k, bool g) { kind = k; grad_only = g; absorb = true; }

  void start(const bsgpu_options* opt, bsgpu_summary* s, std::vector<bsgpu_iteration>* records, int n_tan, int n_res, int linear_solver_used) {
    o = opt; sum = s; iters = records;
    std::memset(sum, 0, sizeof(*sum));
    iters->clear();
    sum->num_parameters_tangent = n_tan; sum->num_residuals = n_res;
    sum->linear_solver_used = linear_solver_used;
    radius = o->initial_trust_region_radius; decrease_factor = 2.0;
    kind = STEP_FIRST; grad_only = false; absorb = false; done = false; retry_timeout = false;
    msg = "";
    t_start = std::chrono::steady_clock::now();
  }
  // the summary's fields the loop owns (times and inner-iteration counts: the driver)
  void finish() {
    done = true;
    sum->num_iterations = (int)iters->size() - 1;
    sum->is_solution_usable = (sum->termination_type == BSGPU_CONVERGENCE || sum->termination_type == BSGPU_NO_CONVERGENCE) ? 1 : 0;
    std::snprintf(sum->message, sizeof(sum->message), "%s", msg);
  }
  // after the scalars of STEP_FIRST.  fixed_cost: the cost of residual blocks whose parameter blocks are all constant (Ceres: fixed_cost)
  void begin(const double* h, double fixed_cost, bool can_retry) {
    fixed = fixed_cost;
    x_cost = h[SC_COST_X];
    std::memset(&it, 0, sizeof(it));
    it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost + fixed;
    it.gradient_max_norm = h[SC_GRAD_MAX]; it.gradient_norm = std::sqrt(h[SC_GRAD_NORM2]);
    sum->initial_cost = x_cost + fixed; sum->fixed_cost = fixed;
    sum->termination_type = BSGPU_NO_CONVERGENCE;
    num_consecutive_invalid = 0;
    absorb = false;
    if (!std::isfinite(x_cost)) { sum->termination_type = BSGPU_FAILURE; msg = "Initial cost is not finite."; sum->final_cost = sum->initial_cost; finish(); return; }
    advance(h, false, can_retry);
  }
  // cost_x_stale: the step's reduction left SC_COST_X alone (it rode in the evaluation launched ahead of the decision, which rewrites those
  // partials): the cost at the accepted point is the candidate's cost the host already holds — the same sum.
  // can_retry: SC_CHOL_FAIL == 2 (a single-launch kernel's wait timed out) is answered with retry_timeout instead of an invalid step.
  void advance(const double* h, bool cost_x_stale, bool can_retry) {
    if (absorb && it.step_is_successful) {
      x_cost = cost_x_stale ? cand_cost : h[SC_COST_X];
      it.cost = x_cost + fixed;
      it.gradient_max_norm = h[SC_GRAD_MAX];
      it.gradient_norm = std::sqrt(h[SC_GRAD_NORM2]);
    }
    absorb = false;
    retry_timeout = false;
    for (;;) {
      if (it.step_is_successful) { if (it.iteration > 0) sum->num_successful_steps++; } else sum->num_unsuccessful_steps++;
      it.trust_region_radius = radius;
      iters->push_back(it);
      if (o->max_solver_time_in_seconds > 0 && elapsed() >= o->max_solver_time_in_seconds) { msg = "Maximum solver time reached."; break; }
      if (it.iteration >= o->max_num_iterations) { msg = "Maximum number of iterations reached."; break; }
      if (it.step_is_successful && it.gradient_max_norm <= o->gradient_tolerance) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
      if (radius <= o->min_trust_region_radius) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
      if (can_retry && h[SC_CHOL_FAIL] == 2.0) {
        // not a numerical failure: the record of this iteration is taken back (it is pushed again when the driver comes back with the
        // recomputed step), the step is wanted again at the same point and radius
        iters->pop_back();
        if (it.step_is_successful) { if (it.iteration > 0) sum->num_successful_steps--; } else sum->num_unsuccessful_steps--;
        kind = STEP_REJECT; grad_only = false; retry_timeout = true;
        return;
      }
      const bsgpu_iteration prev = it;
      std::memset(&it, 0, sizeof(it));
      it.iteration = prev.iteration + 1;
      it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      sum->num_linear_solves++;
      const double mcc = h[SC_MCC];
      const bool lin_ok = !(h[SC_CHOL_FAIL] > 0.0) && std::isfinite(mcc) && std::isfinite(h[SC_STEP_NORM2]);
      it.model_cost_change = lin_ok ? mcc : 0.0;
      it.step_is_valid = lin_ok && mcc > 0.0;
      if (!it.step_is_valid) {
        if (++num_consecutive_invalid >= o->max_num_consecutive_invalid_steps) {
          sum->termination_type = BSGPU_FAILURE;
          msg = "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps.";
          break;
        }
        radius *= 0.5;   // [EXT] LevenbergMarquardtStrategy::StepIsInvalid(): the radius is halved, decrease_factor_ is untouched
        it.cost = x_cost + fixed; it.step_is_successful = 0;
        if (it.iteration >= o->max_num_iterations) continue;   // the loop ends at its top: a step from here would never be looked at
        request(STEP_REJECT, false);
        return;
      }
      num_consecutive_invalid = 0;
      cand_cost = h[SC_COST_CAND];
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      it.step_norm = std::sqrt(h[SC_STEP_NORM2]);
      const double x_norm = std::sqrt(h[SC_X_NORM2]);
      if (it.step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Parameter tolerance reached."; break; }
      it.cost_change = x_cost - cand_cost;
      if (std::fabs(it.cost_change) <= o->function_tolerance * x_cost) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Function tolerance reached."; break; }
      it.relative_decrease = (x_cost - cand_cost) / mcc;
      const bool last_iteration = it.iteration >= o->max_num_iterations;
      if (it.relative_decrease > o->min_relative_decrease) {
        radius = radius / std::max(1.0 / 3.0, 1.0 - lm_cube(2.0 * it.relative_decrease - 1.0));   // ([EXT] pow(2 rho - 1, 3): lm_decide.h lm_cube, shared with the device's decision)
        radius = std::min(o->max_trust_region_radius, radius);
        decrease_factor = 2.0;
        it.step_is_successful = 1;
        // the next step is computed right away so that one synchronisation per iteration suffices; when this was the last iteration the
        // budget allows, only the accepted point's cost and gradient are (a full step would be thrown away)
        request(STEP_ACCEPT, last_iteration);
        return;
      }
      it.step_is_successful = 0;
      radius = radius / decrease_factor; decrease_factor *= 2.0;
      it.cost = cand_cost + fixed;
      if (last_iteration) continue;
      request(STEP_REJECT, false);
      return;
    }
    sum->final_cost = x_cost + fixed;
    finish();
  }
};

}  // namespace bsg
