// Host-side mirrors (ROS/Eigen-free) of the variable types, ImuState and PreIntegrator the solve path
// of the reference is written against:
//   fuse_variables::{Orientation3DStamped, Position3DStamped, VelocityLinear3DStamped}            [EXT fuse]
//   bs_variables::{GyroscopeBias3DStamped, AccelerationBias3DStamped, Point3DLandmark,
//                  Orientation3D, Position3D}        (bs_variables/include/bs_variables/*.h)
//   bs_common::ImuState                               (bs_common/include/bs_common/imu_state.h)
//   bs_common::PreIntegrator                          (bs_common/src/bs_common/preintegrator.cpp:26-144)
// The pre-integrator runs on the host in the reference too (once per factor creation,
// bs_models/src/lib/imu/imu_preintegration.cpp:245-318); it produces the constants of the IMU factor.
#pragma once
#include <new>
#include <algorithm>
#include <cmath>

#include "fuse_core_compat.h"

namespace bs_math {  // minimal fixed-size dense helpers (Eigen is not available)

template <int R, int C> struct Mat {
  double a[R * C];
  Mat() { for (int i = 0; i < R * C; ++i) a[i] = 0.0; }
  double& operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  static Mat Identity() { Mat m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0; return m; }
  double norm() const { double s = 0; for (int i = 0; i < R * C; ++i) s += a[i] * a[i]; return std::sqrt(s); }
  bool allFinite() const { for (int i = 0; i < R * C; ++i) if (!std::isfinite(a[i])) return false; return true; }
};
template <int R, int K, int C> Mat<R, C> operator*(const Mat<R, K>& x, const Mat<K, C>& y) {
  Mat<R, C> o;
  for (int i = 0; i < R; ++i) for (int k = 0; k < K; ++k) { const double v = x(i, k); if (v != 0.0) for (int j = 0; j < C; ++j) o(i, j) += v * y(k, j); }
  return o;
}
template <int R, int C> Mat<R, C> operator+(const Mat<R, C>& x, const Mat<R, C>& y) { Mat<R, C> o; for (int i = 0; i < R * C; ++i) o.a[i] = x.a[i] + y.a[i]; return o; }
template <int R, int C> Mat<R, C> operator-(const Mat<R, C>& x, const Mat<R, C>& y) { Mat<R, C> o; for (int i = 0; i < R * C; ++i) o.a[i] = x.a[i] - y.a[i]; return o; }
template <int R, int C> Mat<R, C> operator*(double s, const Mat<R, C>& x) { Mat<R, C> o; for (int i = 0; i < R * C; ++i) o.a[i] = s * x.a[i]; return o; }
template <int R, int C> Mat<C, R> transpose(const Mat<R, C>& x) { Mat<C, R> o; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) o(j, i) = x(i, j); return o; }
template <int R, int C, int R2, int C2> void setBlock(Mat<R, C>& m, int r0, int c0, const Mat<R2, C2>& b) { for (int i = 0; i < R2; ++i) for (int j = 0; j < C2; ++j) m(r0 + i, c0 + j) = b(i, j); }
template <int R2, int C2, int R, int C> Mat<R2, C2> block(const Mat<R, C>& m, int r0, int c0) { Mat<R2, C2> b; for (int i = 0; i < R2; ++i) for (int j = 0; j < C2; ++j) b(i, j) = m(r0 + i, c0 + j); return b; }
using Mat3 = Mat<3, 3>;
using Vec3 = std::array<double, 3>;
using Quat = std::array<double, 4>;  // (w, x, y, z) everywhere, like the reference (orientation_3d.h:32)

// lower Cholesky; returns false if not positive definite
template <int N> bool choleskyLower(const Mat<N, N>& A, Mat<N, N>& L) {
  L = Mat<N, N>();
  for (int j = 0; j < N; ++j) {
    double d = A(j, j);
    for (int p = 0; p < j; ++p) d -= L(j, p) * L(j, p);
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    L(j, j) = std::sqrt(d);
    for (int i = j + 1; i < N; ++i) { double s = A(i, j); for (int p = 0; p < j; ++p) s -= L(i, p) * L(j, p); L(i, j) = s / L(j, j); }
  }
  return true;
}
template <int N> bool invertSpd(const Mat<N, N>& A, Mat<N, N>& Ainv) {
  Mat<N, N> L;
  if (!choleskyLower(A, L)) return false;
  Mat<N, N> Li;  // L^-1
  for (int c = 0; c < N; ++c)
    for (int i = 0; i < N; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L(i, k) * Li(k, c); Li(i, c) = s / L(i, i); }
  Ainv = transpose(Li) * Li;
  return true;
}
// cov.inverse().llt().matrixU(): upper U with U^T U = cov^-1
// (absolute_imu_state_3d_stamped_constraint.cpp:22, relative_pose_3d_stamped_with_extrinsics_constraint.cpp:29,
//  preintegrator.cpp:135-138)
template <int N> bool sqrtInformationUpper(const Mat<N, N>& cov, Mat<N, N>& U) {
  Mat<N, N> info, L;
  if (!invertSpd(cov, info) || !choleskyLower(info, L)) return false;
  U = transpose(L);
  return true;
}

inline Mat3 skew(const Vec3& v) { Mat3 m; m(0, 1) = -v[2]; m(0, 2) = v[1]; m(1, 0) = v[2]; m(1, 2) = -v[0]; m(2, 0) = -v[1]; m(2, 1) = v[0]; return m; }
inline Mat3 quatToRot(const Quat& q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  Mat3 R;
  R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - w * z); R(0, 2) = 2 * (x * z + w * y);
  R(1, 0) = 2 * (x * y + w * z); R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - w * x);
  R(2, 0) = 2 * (x * z - w * y); R(2, 1) = 2 * (y * z + w * x); R(2, 2) = 1 - 2 * (x * x + y * y);
  return R;
}
inline Quat quatMul(const Quat& a, const Quat& b) {
  return {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
          a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
}
inline Quat quatNormalized(const Quat& q) { const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); return {q[0] / n, q[1] / n, q[2] / n, q[3] / n}; }
inline Quat quatFromAngleAxis(const Vec3& w) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (th < 1e-12) return quatNormalized({1.0, 0.5 * w[0], 0.5 * w[1], 0.5 * w[2]});
  const double k = std::sin(th / 2) / th;
  return {std::cos(th / 2), k * w[0], k * w[1], k * w[2]};
}
inline Vec3 matVec(const Mat3& M, const Vec3& v) { return {M(0, 0) * v[0] + M(0, 1) * v[1] + M(0, 2) * v[2], M(1, 0) * v[0] + M(1, 1) * v[1] + M(1, 2) * v[2], M(2, 0) * v[0] + M(2, 1) * v[1] + M(2, 2) * v[2]}; }
inline Mat3 so3Exp(const Vec3& w) {  // [EXT] beam::LieAlgebraToR
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const Mat3 K = skew(w);
  if (th < 1e-10) return Mat3::Identity() + K + 0.5 * (K * K);
  return Mat3::Identity() + (std::sin(th) / th) * K + ((1 - std::cos(th)) / (th * th)) * (K * K);
}
inline Mat3 so3RightJacobian(const Vec3& w) {  // [EXT] beam::RightJacobianOfSO3
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const Mat3 K = skew(w);
  if (th < 1e-8) return Mat3::Identity() - 0.5 * K + (1.0 / 6.0) * (K * K);
  return Mat3::Identity() - ((1 - std::cos(th)) / (th * th)) * K + ((th - std::sin(th)) / (th * th * th)) * (K * K);
}

}  // namespace bs_math

// ---------------------------------------------------------------------------------------------------
// variables
// ---------------------------------------------------------------------------------------------------
// (only for classes whose members are plain values — uuid, stamp, ids, a fixed array of doubles: destroyIsNoop() says so)
#define BS_CLONE_IN_PLACE(CLASS)                                          \
  size_t cloneSize() const override { return sizeof(CLASS); }            \
  fuse_core::Variable* cloneAt(void* mem) const override { return new (mem) CLASS(*this); } \
  bool destroyIsNoop() const override { return true; }
namespace fuse_variables {
template <int N> class FixedSizeVariable : public fuse_core::Variable {
 public:
  explicit FixedSizeVariable(const fuse_core::UUID& u) : Variable(u) { data_.fill(0.0); }
  size_t size() const override { return N; }
  const double* data() const override { return data_.data(); }
  double* data() override { return data_.data(); }
 protected:
  std::array<double, N> data_;
};
template <int N> class StampedVariable : public FixedSizeVariable<N> {
 public:
  StampedVariable(const std::string& type, const fuse_core::Time& stamp, const fuse_core::UUID& device, int slot)
      : FixedSizeVariable<N>(fuse_core::uuid::generate(type, stamp, device)), stamp_(stamp), device_(device), slot_(slot) {}
  bool isStamped() const override { return true; }
  fuse_core::Time stamp() const override { return stamp_; }
  const fuse_core::UUID& deviceId() const { return device_; }
  int stateSlot() const override { return slot_; }
 private:
  fuse_core::Time stamp_;
  fuse_core::UUID device_;
  int slot_;
};
#define BS_STAMPED_VARIABLE(NS_CLASS, CLASS, N, SLOT, TYPESTR, EXTRA)                                        \
  class CLASS : public StampedVariable<N> {                                                                   \
   public:                                                                                                     \
    using SharedPtr = std::shared_ptr<CLASS>;                                                                  \
    explicit CLASS(const fuse_core::Time& stamp, const fuse_core::UUID& device = fuse_core::UUID())           \
        : StampedVariable<N>(TYPESTR, stamp, device, SLOT) { init(); }                                         \
    static SharedPtr make_shared(const fuse_core::Time& stamp, const fuse_core::UUID& device = fuse_core::UUID()) { return std::make_shared<CLASS>(stamp, device); } \
    std::string type() const override { return TYPESTR; }                                                      \
    fuse_core::Variable::SharedPtr clone() const override { return std::make_shared<CLASS>(*this); }           \
    BS_CLONE_IN_PLACE(CLASS)                                                                                   \
    EXTRA                                                                                                      \
  };
BS_STAMPED_VARIABLE(fuse_variables, Orientation3DStamped, 4, 0, "fuse_variables::Orientation3DStamped",
                    void init() { data_[0] = 1.0; }
                    int manifold() const override { return BSGPU_MANIFOLD_QUAT_RIGHT; }   // Orientation3DLocalParameterization
                    size_t localSize() const override { return 3; }
                    double& w() { return data_[0]; } double& x() { return data_[1]; } double& y() { return data_[2]; } double& z() { return data_[3]; })
BS_STAMPED_VARIABLE(fuse_variables, Position3DStamped, 3, 1, "fuse_variables::Position3DStamped",
                    void init() {} double& x() { return data_[0]; } double& y() { return data_[1]; } double& z() { return data_[2]; })
BS_STAMPED_VARIABLE(fuse_variables, VelocityLinear3DStamped, 3, 2, "fuse_variables::VelocityLinear3DStamped",
                    void init() {} double& x() { return data_[0]; } double& y() { return data_[1]; } double& z() { return data_[2]; })
}  // namespace fuse_variables

namespace bs_variables {
using fuse_variables::StampedVariable;
// bs_variables/include/bs_variables/gyro_bias_3d_stamped.h, accel_bias_3d_stamped.h:21
BS_STAMPED_VARIABLE(bs_variables, GyroscopeBias3DStamped, 3, 3, "bs_variables::GyroscopeBias3DStamped",
                    void init() {} double& x() { return data_[0]; } double& y() { return data_[1]; } double& z() { return data_[2]; })
BS_STAMPED_VARIABLE(bs_variables, AccelerationBias3DStamped, 3, 4, "bs_variables::AccelerationBias3DStamped",
                    void init() {} double& x() { return data_[0]; } double& y() { return data_[1]; } double& z() { return data_[2]; })

// bs_variables/include/bs_variables/point_3d_landmark.h:58 — uuid from (type, landmark id) (point_3d_landmark.cpp:49)
class Point3DLandmark : public fuse_variables::FixedSizeVariable<3> {
 public:
  using SharedPtr = std::shared_ptr<Point3DLandmark>;
  explicit Point3DLandmark(uint64_t id) : FixedSizeVariable<3>(fuse_core::uuid::generate("bs_variables::Point3DLandmark", id)), id_(id) {}
  static SharedPtr make_shared(uint64_t id) { return std::make_shared<Point3DLandmark>(id); }
  std::string type() const override { return "bs_variables::Point3DLandmark"; }
  fuse_core::Variable::SharedPtr clone() const override { return std::make_shared<Point3DLandmark>(*this); }
  BS_CLONE_IN_PLACE(Point3DLandmark)
  bool isLandmark() const override { return true; }
  uint64_t landmarkId() const override { return id_; }
  uint64_t id() const { return id_; }
  double& x() { return data_[0]; } double& y() { return data_[1]; } double& z() { return data_[2]; }
 private:
  uint64_t id_;
};
// bs_variables/include/bs_variables/inverse_depth_landmark.h:22-60, src/inverse_depth_landmark.cpp:14-27:
// one scalar (the inverse depth) + a constant unit bearing in the anchor camera; uuid from the landmark id
class InverseDepthLandmark : public fuse_variables::FixedSizeVariable<1> {
 public:
  using SharedPtr = std::shared_ptr<InverseDepthLandmark>;
  InverseDepthLandmark(uint64_t id, const bs_math::Vec3& bearing, const fuse_core::Time& anchor_stamp)
      : FixedSizeVariable<1>(fuse_core::uuid::generate("bs_variables::InverseDepthLandmark", id)), id_(id), bearing_(bearing),
        anchor_stamp_(anchor_stamp) {
    const double norm = std::sqrt(bearing[0] * bearing[0] + bearing[1] * bearing[1] + bearing[2] * bearing[2]);
    if (norm < 1.0 - 1e-10 || norm > 1.0 + 1e-10) throw std::runtime_error("Invalid bearing vector, norm must equal 1.0.");
  }
  static SharedPtr make_shared(uint64_t id, const bs_math::Vec3& b, const fuse_core::Time& t) { return std::make_shared<InverseDepthLandmark>(id, b, t); }
  std::string type() const override { return "bs_variables::InverseDepthLandmark"; }
  fuse_core::Variable::SharedPtr clone() const override { return std::make_shared<InverseDepthLandmark>(*this); }
  BS_CLONE_IN_PLACE(InverseDepthLandmark)
  bool isLandmark() const override { return true; }
  uint64_t landmarkId() const override { return id_; }
  uint64_t id() const { return id_; }
  double& inverse_depth() { return data_[0]; }
  const double& inverse_depth() const { return data_[0]; }
  bs_math::Vec3 camera_t_point() const { const double d = 1.0 / data_[0]; return bs_math::Vec3{d * bearing_[0], d * bearing_[1], d * bearing_[2]}; }
  const bs_math::Vec3& bearing() const { return bearing_; }
  const fuse_core::Time& anchorStamp() const { return anchor_stamp_; }
 private:
  uint64_t id_;
  bs_math::Vec3 bearing_;
  fuse_core::Time anchor_stamp_;
};
// unstamped extrinsic blocks: holdConstant() == true (bs_variables/src/orientation_3d.cpp:39-41, position_3d.cpp:34);
// uuid from child+parent frame (orientation_3d.cpp:17-18)
class Orientation3D : public fuse_variables::FixedSizeVariable<4> {
 public:
  using SharedPtr = std::shared_ptr<Orientation3D>;
  Orientation3D(const std::string& child, const std::string& parent)
      : FixedSizeVariable<4>(fuse_core::uuid::generate("bs_variables::Orientation3D" + child + parent)) { data_[0] = 1.0; }
  static SharedPtr make_shared(const std::string& c, const std::string& p) { return std::make_shared<Orientation3D>(c, p); }
  std::string type() const override { return "bs_variables::Orientation3D"; }
  fuse_core::Variable::SharedPtr clone() const override { return std::make_shared<Orientation3D>(*this); }
  BS_CLONE_IN_PLACE(Orientation3D)
  int manifold() const override { return BSGPU_MANIFOLD_QUAT_RIGHT; }
  size_t localSize() const override { return 3; }
  bool holdConstant() const override { return true; }
};
class Position3D : public fuse_variables::FixedSizeVariable<3> {
 public:
  using SharedPtr = std::shared_ptr<Position3D>;
  Position3D(const std::string& child, const std::string& parent)
      : FixedSizeVariable<3>(fuse_core::uuid::generate("bs_variables::Position3D" + child + parent)) {}
  static SharedPtr make_shared(const std::string& c, const std::string& p) { return std::make_shared<Position3D>(c, p); }
  std::string type() const override { return "bs_variables::Position3D"; }
  fuse_core::Variable::SharedPtr clone() const override { return std::make_shared<Position3D>(*this); }
  BS_CLONE_IN_PLACE(Position3D)
  bool holdConstant() const override { return true; }
};
}  // namespace bs_variables

// ---------------------------------------------------------------------------------------------------
// bs_common
// ---------------------------------------------------------------------------------------------------
namespace bs_common {
using bs_math::Mat; using bs_math::Mat3; using bs_math::Quat; using bs_math::Vec3;

static const Vec3 GRAVITY_WORLD{0.0, 0.0, -9.80665};  // bs_common/include/bs_common/utils.h:20-24

// bs_common/include/bs_common/imu_state.h: a bundle of the five stamped variables of one IMU state
class ImuState {
 public:
  ImuState() : ImuState(fuse_core::Time()) {}
  explicit ImuState(const fuse_core::Time& t)
      : stamp_(t), orientation_(t), position_(t), velocity_(t), gyrobias_(t), accelbias_(t) {}
  ImuState(const fuse_core::Time& t, const Quat& q, const Vec3& p, const Vec3& v, const Vec3& bg = {0, 0, 0}, const Vec3& ba = {0, 0, 0})
      : ImuState(t) { SetOrientation(q); SetPosition(p); SetVelocity(v); SetGyroBias(bg); SetAccelBias(ba); }
  fuse_core::Time Stamp() const { return stamp_; }
  const fuse_variables::Orientation3DStamped& Orientation() const { return orientation_; }
  const fuse_variables::Position3DStamped& Position() const { return position_; }
  const fuse_variables::VelocityLinear3DStamped& Velocity() const { return velocity_; }
  const bs_variables::GyroscopeBias3DStamped& GyroBias() const { return gyrobias_; }
  const bs_variables::AccelerationBias3DStamped& AccelBias() const { return accelbias_; }
  Quat OrientationQuat() const { const double* d = orientation_.data(); return {d[0], d[1], d[2], d[3]}; }
  Mat3 OrientationMat() const { return bs_math::quatToRot(OrientationQuat()); }
  Vec3 PositionVec() const { const double* d = position_.data(); return {d[0], d[1], d[2]}; }
  Vec3 VelocityVec() const { const double* d = velocity_.data(); return {d[0], d[1], d[2]}; }
  Vec3 GyroBiasVec() const { const double* d = gyrobias_.data(); return {d[0], d[1], d[2]}; }
  Vec3 AccelBiasVec() const { const double* d = accelbias_.data(); return {d[0], d[1], d[2]}; }
  void SetOrientation(const Quat& q) { for (int i = 0; i < 4; ++i) orientation_.data()[i] = q[i]; }
  void SetPosition(const Vec3& p) { for (int i = 0; i < 3; ++i) position_.data()[i] = p[i]; }
  void SetVelocity(const Vec3& v) { for (int i = 0; i < 3; ++i) velocity_.data()[i] = v[i]; }
  void SetGyroBias(const Vec3& b) { for (int i = 0; i < 3; ++i) gyrobias_.data()[i] = b[i]; }
  void SetAccelBias(const Vec3& b) { for (int i = 0; i < 3; ++i) accelbias_.data()[i] = b[i]; }
  // GetStateVector(): (q, p, v, bg, ba), 16 values (bs_common/src/bs_common/imu_state.cpp:348-354)
  std::array<double, 16> GetStateVector() const {
    std::array<double, 16> s{};
    const Quat q = OrientationQuat(); const Vec3 p = PositionVec(), v = VelocityVec(), bg = GyroBiasVec(), ba = AccelBiasVec();
    for (int i = 0; i < 4; ++i) s[i] = q[i];
    for (int i = 0; i < 3; ++i) { s[4 + i] = p[i]; s[7 + i] = v[i]; s[10 + i] = bg[i]; s[13 + i] = ba[i]; }
    return s;
  }
  // Update(graph): pull the current estimates of the five variables back (imu_state.cpp Update)
  template <typename GraphT> bool Update(const GraphT& graph) {
    const fuse_core::UUID ids[5] = {orientation_.uuid(), position_.uuid(), velocity_.uuid(), gyrobias_.uuid(), accelbias_.uuid()};
    for (const auto& u : ids) if (!graph.variableExists(u)) return false;
    std::memcpy(orientation_.data(), graph.getVariable(ids[0]).data(), 4 * sizeof(double));
    std::memcpy(position_.data(), graph.getVariable(ids[1]).data(), 3 * sizeof(double));
    std::memcpy(velocity_.data(), graph.getVariable(ids[2]).data(), 3 * sizeof(double));
    std::memcpy(gyrobias_.data(), graph.getVariable(ids[3]).data(), 3 * sizeof(double));
    std::memcpy(accelbias_.data(), graph.getVariable(ids[4]).data(), 3 * sizeof(double));
    ++updates_;
    return true;
  }
  int Updates() const { return updates_; }
 private:
  fuse_core::Time stamp_;
  fuse_variables::Orientation3DStamped orientation_;
  fuse_variables::Position3DStamped position_;
  fuse_variables::VelocityLinear3DStamped velocity_;
  bs_variables::GyroscopeBias3DStamped gyrobias_;
  bs_variables::AccelerationBias3DStamped accelbias_;
  int updates_ = 0;
};

enum ErrorStateLocation { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12, ES_SIZE = 15 };  // preintegrator.h:10-17
struct IMUData { fuse_core::Time t; Vec3 w{0, 0, 0}; Vec3 a{0, 0, 0}; };
struct Delta { double t = 0; Quat q{1, 0, 0, 0}; Vec3 p{0, 0, 0}; Vec3 v{0, 0, 0}; Mat<15, 15> cov; Mat<15, 15> sqrt_inv_cov; };
struct Jacobian { Mat3 dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba; };

// bs_common/src/bs_common/preintegrator.cpp
class PreIntegrator {
 public:
  double cov_tol{1e-5}, bias_cov_tol{1e-9};                  // preintegrator.h:129-130
  Mat3 cov_w, cov_a, cov_bg, cov_ba;                          // continuous-time covariances
  Delta delta;
  Jacobian jacobian;
  std::map<fuse_core::Time, IMUData> data;
  double invalid_inv_cov_weight_{1e-4};

  void Reset() { delta = Delta(); jacobian = Jacobian(); }                                        // :7-20
  void Clear(const fuse_core::Time& t) { while (!data.empty() && data.begin()->first < t) data.erase(data.begin()); }  // :22-24

  void Increment(double dtd, const IMUData& d, const Vec3& bg, const Vec3& ba, bool compute_jacobian, bool compute_covariance) {  // :26-89
    using namespace bs_math;
    const Vec3 w{d.w[0] - bg[0], d.w[1] - bg[1], d.w[2] - bg[2]}, a{d.a[0] - ba[0], d.a[1] - ba[1], d.a[2] - ba[2]};
    const Vec3 wdt{w[0] * dtd, w[1] * dtd, w[2] * dtd}, whalf{0.5 * wdt[0], 0.5 * wdt[1], 0.5 * wdt[2]};
    const Mat3 R_full = so3Exp(wdt), Rdq = quatToRot(delta.q), Sa = skew(a), Jr = so3RightJacobian(wdt);
    if (compute_covariance) {
      Mat<9, 9> A = Mat<9, 9>::Identity();
      setBlock(A, ES_Q, ES_Q, transpose(R_full));
      setBlock(A, ES_V, ES_Q, (-dtd) * (Rdq * Sa));
      setBlock(A, ES_P, ES_Q, (-0.5 * dtd * dtd) * (Rdq * Sa));
      setBlock(A, ES_P, ES_V, dtd * Mat3::Identity());
      Mat<9, 6> B;
      setBlock(B, ES_Q, 0, dtd * Jr);
      setBlock(B, ES_V, 3, dtd * Rdq);
      setBlock(B, ES_P, 3, (0.5 * dtd * dtd) * Rdq);
      const double inv_dtd = 1.0 / std::max(dtd, 1.0e-7);
      Mat<6, 6> Q;
      setBlock(Q, 0, 0, inv_dtd * cov_w);
      setBlock(Q, 3, 3, inv_dtd * cov_a);
      const Mat<9, 9> c9 = block<9, 9>(delta.cov, 0, 0);
      setBlock(delta.cov, 0, 0, A * c9 * transpose(A) + B * Q * transpose(B));
      setBlock(delta.cov, ES_BG, ES_BG, block<3, 3>(delta.cov, ES_BG, ES_BG) + dtd * cov_bg);
      setBlock(delta.cov, ES_BA, ES_BA, block<3, 3>(delta.cov, ES_BA, ES_BA) + dtd * cov_ba);
    }
    if (compute_jacobian) {  // order matters (:69-80)
      jacobian.dp_dbg = jacobian.dp_dbg + dtd * jacobian.dv_dbg - (0.5 * dtd * dtd) * (Rdq * Sa * jacobian.dq_dbg);
      jacobian.dp_dba = jacobian.dp_dba + dtd * jacobian.dv_dba - (0.5 * dtd * dtd) * Rdq;
      jacobian.dv_dbg = jacobian.dv_dbg - dtd * (Rdq * Sa * jacobian.dq_dbg);
      jacobian.dv_dba = jacobian.dv_dba - dtd * Rdq;
      jacobian.dq_dbg = transpose(R_full) * jacobian.dq_dbg - dtd * Jr;
    }
    const Quat q_mid = quatMul(delta.q, quatFromAngleAxis(whalf));
    const Vec3 a_mid = matVec(quatToRot(q_mid), a);
    delta.t += dtd;
    for (int i = 0; i < 3; ++i) { delta.p[i] += dtd * delta.v[i] + 0.5 * dtd * dtd * a_mid[i]; }
    for (int i = 0; i < 3; ++i) delta.v[i] += dtd * a_mid[i];
    delta.q = quatNormalized(quatMul(delta.q, quatFromAngleAxis(wdt)));
  }

  bool Integrate(const fuse_core::Time& t, const Vec3& bg, const Vec3& ba, bool compute_jacobian, bool compute_covariance,
                 bool compute_information) {  // :91-115
    if (data.empty()) return false;
    Reset();
    for (auto it = data.begin(); std::next(it) != data.end(); ++it) {
      const auto nx = std::next(it);
      if (nx->first > t) break;
      Increment(nx->first - it->first, it->second, bg, ba, compute_jacobian, compute_covariance);
    }
    const double dt = t - data.rbegin()->first;
    if (dt > 0) Increment(dt, data.rbegin()->second, bg, ba, compute_jacobian, compute_covariance);
    if (compute_information) ComputeSqrtInvCov();
    return true;
  }

  void ComputeSqrtInvCov() {  // :117-143
    using namespace bs_math;
    if (block<9, 9>(delta.cov, 0, 0).norm() < cov_tol) setBlock(delta.cov, 0, 0, cov_tol * Mat<9, 9>::Identity());
    if (block<6, 6>(delta.cov, ES_BG, ES_BG).norm() < bias_cov_tol) setBlock(delta.cov, ES_BG, ES_BG, bias_cov_tol * Mat<6, 6>::Identity());
    if (!sqrtInformationUpper(delta.cov, delta.sqrt_inv_cov) || !delta.sqrt_inv_cov.allFinite())
      delta.sqrt_inv_cov = invalid_inv_cov_weight_ * Mat<15, 15>::Identity();
  }
};

}  // namespace bs_common
