// TEST INFRASTRUCTURE — part of the CPU oracle (see oracle/README.md).  Never linked into,
// imported by or executed from the product path (beam_slam_amd/).
//
// Forward-mode dual numbers, restating what ceres::Jet<double, N> gives
// ceres::AutoDiffCostFunction (the mechanism every bs_constraints costFunction() except the
// analytic reprojection factor goes through, e.g.
// bs_constraints/src/inertial/relative_imu_state_3d_stamped_constraint.cpp:48-53).
#pragma once
#include <cmath>

namespace bso {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h; const double gi = 1.0 / g.a; const double q = f.a * gi; h.a = q;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - q * g.v[i]) * gi; return h; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> h = -f; h.a += s; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }
template <int N> inline Jet<N>& operator*=(Jet<N>& f, const Jet<N>& g) { f = f * g; return f; }
template <int N> inline Jet<N>& operator/=(Jet<N>& f, const Jet<N>& g) { f = f / g; return f; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double g) { return f.a < g; }
template <int N> inline bool operator>(const Jet<N>& f, double g) { return f.a > g; }

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
  Jet<N> h; h.a = std::sqrt(f.a); const double d = 1.0 / (2.0 * h.a);
  for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <int N> inline Jet<N> sin(const Jet<N>& f) {
  Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> cos(const Jet<N>& f) {
  Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a);
  for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  // d atan2(g, f) = (f dg - g df) / (f^2 + g^2)
  Jet<N> h; h.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) h.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]); return h; }

inline double sqrt(double x) { return std::sqrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }

}  // namespace bso
