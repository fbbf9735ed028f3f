// The trust-region decision as the device takes it (bsgpu_device.h final_reduce_done -> bsgpu_solve.cpp enqueue_step): LmState::advance (lm_state.h;
// [EXT] ceres TrustRegionMinimizer + LevenbergMarquardtStrategy::StepAccepted, configured by beam_slam_launch/config/vio.yaml:7-17) restated for the
// common case.  Host- and device-compilable: tests/plan/test_lm_decide.cpp runs it on the CPU against LmState::advance itself, bit for bit.
#pragma once
#include <cmath>

#include "bsgpu_internal.h"

#if defined(__HIPCC__)
#define BSG_LMD_FN __host__ __device__ __forceinline__
#else
#define BSG_LMD_FN inline
#endif

namespace bsg {

struct LmScal { double mcc, sn2, xn2, cand, cost_x, gmax, chol_fail, word /* (the decision's word on its way from thread 0 to the storing wave) */; };

// t^3 rounded ONCE (t^2 and its product with t as exact sums of two doubles): what LmState::advance and lm_decide both use for Ceres'
// pow(2 rho - 1, 3) — one function, so the host's radius and the device's are the same bits by construction.  glibc's pow(t, 3) itself is
// within an ulp but not always correctly rounded: it differs from this in 0.045 % of arguments (tests/plan/test_lm_decide.cpp).
BSG_LMD_FN double lm_cube(double t) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double t2 = t * t, e2 = __builtin_fma(t, t, -t2);
  const double p = t2 * t, ep = __builtin_fma(t2, t, -p) + e2 * t;
  return p + ep;
}

// 1 = accepted, with the radius of the next step in *radius_out.  Anything else — an invalid step, a tolerance reached, a rejected step — is 0 and
// left to the host.  The host's arithmetic: no contraction (its x86 code has none), the cube by lm_cube.
BSG_LMD_FN int lm_decide(const LmDecide& d, const LmScal& v, double* radius_out) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  *radius_out = d.radius;
  if (d.check_grad && v.gmax <= d.gradient_tolerance) return 0;
  const double mcc = v.mcc, sn2 = v.sn2;
  const bool lin_ok = !(v.chol_fail > 0.0) && std::isfinite(mcc) && std::isfinite(sn2);
  if (!(lin_ok && mcc > 0.0)) return 0;
  double cand = v.cand;
  if (!std::isfinite(cand)) cand = 1.7976931348623157e308;
  const double x_cost = d.x_from_scal ? v.cost_x : d.x_cost;
  const double step_norm = sqrt(sn2), x_norm = sqrt(v.xn2);
  if (step_norm <= d.parameter_tolerance * (x_norm + d.parameter_tolerance)) return 0;
  const double cost_change = x_cost - cand;
  if (fabs(cost_change) <= d.function_tolerance * x_cost) return 0;
  const double rd = cost_change / mcc;
  if (!(rd > d.min_relative_decrease)) return 0;
  double r = d.radius / fmax(1.0 / 3.0, 1.0 - lm_cube(2.0 * rd - 1.0));
  r = fmin(d.max_radius, r);
  *radius_out = r;
  return 1;
}

}  // namespace bsg
