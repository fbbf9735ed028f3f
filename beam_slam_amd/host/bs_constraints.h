// Host-side mirrors of the constraint classes on the solve path: same class names, type() strings,
// constructor arguments and variable (= parameter-block) order as the reference; instead of
// costFunction() -> ceres::CostFunction* each class has pack(), which appends its payload to the flat
// tables of include/bsgpu.h.  No arithmetic of the solve happens here.
#pragma once
#include "bs_common.h"

namespace bs_constraints {
using bs_math::Mat; using bs_math::Quat; using bs_math::Vec3;
using BlockOf = fuse_core::BlockOf;

inline void appendBlocks(const fuse_core::Constraint& c, const BlockOf& block_of, std::vector<int32_t>& idx) {
  for (size_t i = 0; i < c.variables().size(); ++i) idx.push_back(block_of(c, i));
}
template <int N> void appendMat(std::vector<double>& v, const Mat<N, N>& m) { v.insert(v.end(), m.a, m.a + N * N); }

inline bsgpu_camera makeCamera(const Mat<3, 3>& K, const Mat<4, 4>& T_cam_baselink) {
  bsgpu_camera c;
  c.fx = K(0, 0); c.fy = K(1, 1); c.cx = K(0, 2); c.cy = K(1, 2);   // K must be skew-free (jacobians.cpp:205-212)
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c.R_cam_baselink[3 * i + j] = T_cam_baselink(i, j); c.t_cam_baselink[i] = T_cam_baselink(i, 3); }
  return c;
}

// bs_constraints/src/visual/euclidean_reprojection_constraint.cpp:15-30
class EuclideanReprojectionConstraint : public fuse_core::Constraint {
 public:
  EuclideanReprojectionConstraint(const std::string& source, const fuse_variables::Orientation3DStamped& R_WORLD_BASELINK,
                                  const fuse_variables::Position3DStamped& t_WORLD_BASELINK,
                                  const bs_variables::Point3DLandmark& P_WORLD, const Mat<4, 4>& T_cam_baselink,
                                  const Mat<3, 3>& intrinsic_matrix, const std::array<double, 2>& measurement,
                                  double reprojection_information_weight = 1.0)
      : Constraint(source, {R_WORLD_BASELINK.uuid(), t_WORLD_BASELINK.uuid(), P_WORLD.uuid()}),
        T_cam_baselink_(T_cam_baselink), intrinsic_matrix_(intrinsic_matrix), pixel_(measurement),
        weight_(reprojection_information_weight) {}
  std::string type() const override { return "bs_constraints::EuclideanReprojectionConstraint"; }
  const std::array<double, 2>& pixel() const { return pixel_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_REPROJ]);
    t.idx[BSGPU_F_REPROJ].push_back(t.cameraId(makeCamera(intrinsic_matrix_, T_cam_baselink_)));
    auto& c = t.consts[BSGPU_F_REPROJ];
    c.push_back(pixel_[0]); c.push_back(pixel_[1]); c.push_back(weight_);
    t.pushLoss(BSGPU_F_REPROJ, loss());
  }
  SharedPtr clone() const override { return std::make_shared<EuclideanReprojectionConstraint>(*this); }
 protected:
  Mat<4, 4> T_cam_baselink_;
  Mat<3, 3> intrinsic_matrix_;
  std::array<double, 2> pixel_;
  double weight_;
};

// bs_constraints/src/visual/inversedepth_reprojection_constraint.cpp:14-49 (AutoDiff<2,4,3,4,3,1>):
// anchor pose, measurement pose, inverse-depth landmark; sqrt information = weight * I2
class InverseDepthReprojectionConstraint : public fuse_core::Constraint {
 public:
  InverseDepthReprojectionConstraint(const std::string& source, const fuse_variables::Orientation3DStamped& o_WORLD_BASELINKa,
                                     const fuse_variables::Position3DStamped& p_WORLD_BASELINKa,
                                     const fuse_variables::Orientation3DStamped& o_WORLD_BASELINKm,
                                     const fuse_variables::Position3DStamped& p_WORLD_BASELINKm,
                                     const bs_variables::InverseDepthLandmark& idp, const Mat<4, 4>& T_cam_baselink,
                                     const Mat<3, 3>& intrinsic_matrix, const std::array<double, 2>& measurement,
                                     double reprojection_information_weight = 1.0)
      : Constraint(source, {o_WORLD_BASELINKa.uuid(), p_WORLD_BASELINKa.uuid(), o_WORLD_BASELINKm.uuid(),
                            p_WORLD_BASELINKm.uuid(), idp.uuid()}),
        T_cam_baselink_(T_cam_baselink), intrinsic_matrix_(intrinsic_matrix), pixel_(measurement), bearing_(idp.bearing()),
        weight_(reprojection_information_weight) {}
  std::string type() const override { return "bs_constraints::InverseDepthReprojectionConstraint"; }
  const std::array<double, 2>& pixel() const { return pixel_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_IDP_REPROJ]);
    t.idx[BSGPU_F_IDP_REPROJ].push_back(t.cameraId(makeCamera(intrinsic_matrix_, T_cam_baselink_)));
    auto& c = t.consts[BSGPU_F_IDP_REPROJ];
    c.push_back(pixel_[0]); c.push_back(pixel_[1]); c.push_back(weight_);
    for (int i = 0; i < 3; ++i) c.push_back(bearing_[i]);
    t.pushLoss(BSGPU_F_IDP_REPROJ, loss());
  }
  SharedPtr clone() const override { return std::make_shared<InverseDepthReprojectionConstraint>(*this); }
 protected:
  Mat<4, 4> T_cam_baselink_;
  Mat<3, 3> intrinsic_matrix_;
  std::array<double, 2> pixel_;
  Vec3 bearing_;
  double weight_;
};

// bs_constraints/src/visual/inversedepth_reprojection_constraint_unary.cpp:15-53 (AutoDiff<2,4,3,1>):
// the observation made from the anchor keyframe itself
class InverseDepthReprojectionConstraintUnary : public fuse_core::Constraint {
 public:
  InverseDepthReprojectionConstraintUnary(const std::string& source, const fuse_variables::Orientation3DStamped& o_WORLD_BASELINKa,
                                          const fuse_variables::Position3DStamped& p_WORLD_BASELINKa,
                                          const bs_variables::InverseDepthLandmark& idp, const Mat<4, 4>& T_cam_baselink,
                                          const Mat<3, 3>& intrinsic_matrix, const std::array<double, 2>& measurement,
                                          double reprojection_information_weight = 1.0)
      : Constraint(source, {o_WORLD_BASELINKa.uuid(), p_WORLD_BASELINKa.uuid(), idp.uuid()}), T_cam_baselink_(T_cam_baselink),
        intrinsic_matrix_(intrinsic_matrix), pixel_(measurement), bearing_(idp.bearing()), weight_(reprojection_information_weight) {}
  std::string type() const override { return "bs_constraints::InverseDepthReprojectionConstraintUnary"; }
  const std::array<double, 2>& pixel() const { return pixel_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_IDP_REPROJ_UNARY]);
    t.idx[BSGPU_F_IDP_REPROJ_UNARY].push_back(t.cameraId(makeCamera(intrinsic_matrix_, T_cam_baselink_)));
    auto& c = t.consts[BSGPU_F_IDP_REPROJ_UNARY];
    c.push_back(pixel_[0]); c.push_back(pixel_[1]); c.push_back(weight_);
    for (int i = 0; i < 3; ++i) c.push_back(bearing_[i]);
    t.pushLoss(BSGPU_F_IDP_REPROJ_UNARY, loss());
  }
  SharedPtr clone() const override { return std::make_shared<InverseDepthReprojectionConstraintUnary>(*this); }
 protected:
  Mat<4, 4> T_cam_baselink_;
  Mat<3, 3> intrinsic_matrix_;
  std::array<double, 2> pixel_;
  Vec3 bearing_;
  double weight_;
};

// bs_constraints/src/visual/euclidean_reprojection_constraint_online_calib.cpp (functor_online_calib.h:16-83)
class EuclideanReprojectionConstraintOnlineCalib : public fuse_core::Constraint {
 public:
  EuclideanReprojectionConstraintOnlineCalib(const std::string& source, const fuse_variables::Orientation3DStamped& R_WORLD_BASELINK,
                                             const fuse_variables::Position3DStamped& t_WORLD_BASELINK,
                                             const bs_variables::Point3DLandmark& P_WORLD,
                                             const bs_variables::Orientation3D& R_BASELINK_CAM,
                                             const bs_variables::Position3D& t_BASELINK_CAM, const Mat<3, 3>& intrinsic_matrix,
                                             const std::array<double, 2>& measurement, double weight = 1.0)
      : Constraint(source, {R_WORLD_BASELINK.uuid(), t_WORLD_BASELINK.uuid(), P_WORLD.uuid(), R_BASELINK_CAM.uuid(), t_BASELINK_CAM.uuid()}),
        intrinsic_matrix_(intrinsic_matrix), pixel_(measurement), weight_(weight) {}
  std::string type() const override { return "bs_constraints::EuclideanReprojectionConstraintOnlineCalib"; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_REPROJ_ONLINE_CALIB]);
    Mat<4, 4> I = Mat<4, 4>::Identity();
    t.idx[BSGPU_F_REPROJ_ONLINE_CALIB].push_back(t.cameraId(makeCamera(intrinsic_matrix_, I)));  // K only
    auto& c = t.consts[BSGPU_F_REPROJ_ONLINE_CALIB];
    c.push_back(pixel_[0]); c.push_back(pixel_[1]); c.push_back(weight_);
    t.pushLoss(BSGPU_F_REPROJ_ONLINE_CALIB, loss());
  }
  SharedPtr clone() const override { return std::make_shared<EuclideanReprojectionConstraintOnlineCalib>(*this); }
 protected:
  Mat<3, 3> intrinsic_matrix_;
  std::array<double, 2> pixel_;
  double weight_;
};

// bs_constraints/src/inertial/relative_imu_state_3d_stamped_constraint.cpp:13-27,48-53
class RelativeImuState3DStampedConstraint : public fuse_core::Constraint {
 public:
  RelativeImuState3DStampedConstraint(const std::string& source, const bs_common::ImuState& imu_state_i,
                                      const bs_common::ImuState& imu_state_j,
                                      const std::shared_ptr<bs_common::PreIntegrator>& pre_integrator, double info_weight = 1.0)
      : Constraint(source, {imu_state_i.Orientation().uuid(), imu_state_i.Position().uuid(), imu_state_i.Velocity().uuid(),
                            imu_state_i.GyroBias().uuid(), imu_state_i.AccelBias().uuid(), imu_state_j.Orientation().uuid(),
                            imu_state_j.Position().uuid(), imu_state_j.Velocity().uuid(), imu_state_j.GyroBias().uuid(),
                            imu_state_j.AccelBias().uuid()}),
        imu_state_i_(imu_state_i), imu_state_j_(imu_state_j), delta_(pre_integrator->delta), jacobian_(pre_integrator->jacobian),
        info_weight_(info_weight) {}
  std::string type() const override { return "bs_constraints::RelativeImuState3DStampedConstraint"; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_IMU_DELTA]);
    auto& c = t.consts[BSGPU_F_IMU_DELTA];
    c.push_back(delta_.t);
    c.insert(c.end(), delta_.q.begin(), delta_.q.end());
    c.insert(c.end(), delta_.p.begin(), delta_.p.end());
    c.insert(c.end(), delta_.v.begin(), delta_.v.end());
    appendMat(c, jacobian_.dq_dbg); appendMat(c, jacobian_.dp_dbg); appendMat(c, jacobian_.dp_dba);
    appendMat(c, jacobian_.dv_dbg); appendMat(c, jacobian_.dv_dba);
    const Vec3 bg = imu_state_i_.GyroBiasVec(), ba = imu_state_i_.AccelBiasVec();   // linearisation biases (functor.h:86-87)
    c.insert(c.end(), bg.begin(), bg.end()); c.insert(c.end(), ba.begin(), ba.end());
    appendMat(c, info_weight_ * delta_.sqrt_inv_cov);                                // A_ (functor.h:57)
    t.pushLoss(BSGPU_F_IMU_DELTA, loss());
  }
  SharedPtr clone() const override { return std::make_shared<RelativeImuState3DStampedConstraint>(*this); }
 protected:
  bs_common::ImuState imu_state_i_, imu_state_j_;
  bs_common::Delta delta_;
  bs_common::Jacobian jacobian_;
  double info_weight_;
};

// bs_constraints/src/inertial/absolute_imu_state_3d_stamped_constraint.cpp:13-22,43-47
class AbsoluteImuState3DStampedConstraint : public fuse_core::Constraint {
 public:
  AbsoluteImuState3DStampedConstraint(const std::string& source, const bs_common::ImuState& s, const std::array<double, 16>& mean,
                                      const Mat<15, 15>& covariance)
      : Constraint(source, {s.Orientation().uuid(), s.Position().uuid(), s.Velocity().uuid(), s.GyroBias().uuid(), s.AccelBias().uuid()}),
        mean_(mean) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return "bs_constraints::AbsoluteImuState3DStampedConstraint"; }
  const std::array<double, 16>& mean() const { return mean_; }
  const Mat<15, 15>& sqrtInformation() const { return sqrt_information_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_IMU_PRIOR]);
    auto& c = t.consts[BSGPU_F_IMU_PRIOR];
    c.insert(c.end(), mean_.begin(), mean_.end());
    appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_IMU_PRIOR, loss());
  }
  SharedPtr clone() const override { return std::make_shared<AbsoluteImuState3DStampedConstraint>(*this); }
 protected:
  std::array<double, 16> mean_;
  Mat<15, 15> sqrt_information_;
};

using Vector7d = std::array<double, 7>;  // (x, y, z, qw, qx, qy, qz)

// bs_constraints/src/relative_pose/relative_pose_3d_stamped_with_extrinsics_constraint.cpp:13-29
class RelativePose3DStampedWithExtrinsicsConstraint : public fuse_core::Constraint {
 public:
  RelativePose3DStampedWithExtrinsicsConstraint(const std::string& source, const fuse_variables::Position3DStamped& position1,
                                                const fuse_variables::Orientation3DStamped& orientation1,
                                                const fuse_variables::Position3DStamped& position2,
                                                const fuse_variables::Orientation3DStamped& orientation2,
                                                const bs_variables::Position3D& position_extrinsics,
                                                const bs_variables::Orientation3D& orientation_extrinsics,
                                                const Vector7d& d_Sensor1_Sensor2, const Mat<6, 6>& covariance)
      : Constraint(source, {position1.uuid(), orientation1.uuid(), position2.uuid(), orientation2.uuid(), position_extrinsics.uuid(),
                            orientation_extrinsics.uuid()}),
        delta_(d_Sensor1_Sensor2) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return "bs_constraints::RelativePose3DStampedWithExtrinsicsConstraint"; }
  const Vector7d& delta() const { return delta_; }
  const Mat<6, 6>& sqrtInformation() const { return sqrt_information_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_RELPOSE_EXT]);
    auto& c = t.consts[BSGPU_F_RELPOSE_EXT];
    c.insert(c.end(), delta_.begin(), delta_.end());
    appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_RELPOSE_EXT, loss());
  }
  SharedPtr clone() const override { return std::make_shared<RelativePose3DStampedWithExtrinsicsConstraint>(*this); }
 protected:
  Vector7d delta_;
  Mat<6, 6> sqrt_information_;
};

// bs_constraints/src/global/gravity_alignment_stamped_constraint.cpp:12-37
class GravityAlignmentStampedConstraint : public fuse_core::Constraint {
 public:
  GravityAlignmentStampedConstraint(const std::string& source, const fuse_core::UUID& orientation_uuid,
                                    const Vec3& gravity_in_baselink, const Mat<2, 2>& covariance)
      : Constraint(source, {orientation_uuid}), g_(gravity_in_baselink) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return "bs_constraints::GravityAlignmentStampedConstraint"; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    appendBlocks(*this, block_of, t.idx[BSGPU_F_GRAVITY]);
    auto& c = t.consts[BSGPU_F_GRAVITY];
    c.insert(c.end(), g_.begin(), g_.end());
    appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_GRAVITY, loss());
  }
  SharedPtr clone() const override { return std::make_shared<GravityAlignmentStampedConstraint>(*this); }
 protected:
  Vec3 g_;
  Mat<2, 2> sqrt_information_;
};

}  // namespace bs_constraints

// ---------------------------------------------------------------------------------------------------
// [EXT] fuse_constraints used by the reference on this path (pose_3d_stamped_transaction.cpp:38-96,
// bs_common/src/bs_common/utils.cpp:128-132, absolute_constraint.h:10-25, relative_constraints.h:12-19)
// ---------------------------------------------------------------------------------------------------
namespace fuse_constraints {
// [EXT] fuse_constraints::MarginalConstraint — the dense linear prior fuse_constraints::marginalizeVariables leaves
// behind (bs_optimizers/src/fixed_lag_smoother.cpp:270-271):  cost = 1/2 | b + sum_i A_i (x_i [-] x_bar_i) |^2.
// A is stored as one rows x cols row-major matrix, columns in variables() order (tangent sizes).
class MarginalConstraint : public fuse_core::Constraint {
 public:
  MarginalConstraint(const std::string& source, std::vector<fuse_core::UUID> variables, int rows, int cols, std::vector<double> A,
                     std::vector<double> b, std::vector<double> x_bar)
      : Constraint(source, std::move(variables)), rows_(rows), cols_(cols), A_(std::move(A)), b_(std::move(b)), x_bar_(std::move(x_bar)) {
    if ((size_t)rows_ * cols_ != A_.size() || (size_t)rows_ != b_.size()) throw std::invalid_argument("MarginalConstraint: A / b sizes disagree");
  }
  std::string type() const override { return "fuse_constraints::MarginalConstraint"; }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  const std::vector<double>& A() const { return A_; }
  const std::vector<double>& b() const { return b_; }
  const std::vector<double>& x_bar() const { return x_bar_; }
  void pack(const bs_constraints::BlockOf& block_of, fuse_core::FactorTables& t) const override {
    fuse_core::FactorTables::MarginalEntry e;
    for (size_t i = 0; i < variables().size(); ++i) e.blocks.push_back(block_of(*this, i));
    e.rows = rows_; e.A = A_; e.b = b_; e.xbar = x_bar_;
    t.marginals.push_back(std::move(e));
  }
  SharedPtr clone() const override { return std::make_shared<MarginalConstraint>(*this); }
 private:
  int rows_, cols_;
  std::vector<double> A_, b_, x_bar_;
};

using bs_constraints::BlockOf; using bs_constraints::Vector7d; using bs_math::Mat; using bs_math::Vec3;

class RelativePose3DStampedConstraint : public fuse_core::Constraint {
 public:
  RelativePose3DStampedConstraint(const std::string& source, const fuse_variables::Position3DStamped& position1,
                                  const fuse_variables::Orientation3DStamped& orientation1,
                                  const fuse_variables::Position3DStamped& position2,
                                  const fuse_variables::Orientation3DStamped& orientation2, const Vector7d& delta,
                                  const Mat<6, 6>& covariance)
      : Constraint(source, {position1.uuid(), orientation1.uuid(), position2.uuid(), orientation2.uuid()}), delta_(delta) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return "fuse_constraints::RelativePose3DStampedConstraint"; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    bs_constraints::appendBlocks(*this, block_of, t.idx[BSGPU_F_RELPOSE]);
    auto& c = t.consts[BSGPU_F_RELPOSE];
    c.insert(c.end(), delta_.begin(), delta_.end());
    bs_constraints::appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_RELPOSE, loss());
  }
  SharedPtr clone() const override { return std::make_shared<RelativePose3DStampedConstraint>(*this); }
 protected:
  Vector7d delta_;
  Mat<6, 6> sqrt_information_;
};

class AbsolutePose3DStampedConstraint : public fuse_core::Constraint {
 public:
  AbsolutePose3DStampedConstraint(const std::string& source, const fuse_variables::Position3DStamped& position,
                                  const fuse_variables::Orientation3DStamped& orientation, const Vector7d& mean,
                                  const Mat<6, 6>& covariance)
      : Constraint(source, {position.uuid(), orientation.uuid()}), mean_(mean) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return "fuse_constraints::AbsolutePose3DStampedConstraint"; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    bs_constraints::appendBlocks(*this, block_of, t.idx[BSGPU_F_ABSPOSE]);
    auto& c = t.consts[BSGPU_F_ABSPOSE];
    c.insert(c.end(), mean_.begin(), mean_.end());
    bs_constraints::appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_ABSPOSE, loss());
  }
  SharedPtr clone() const override { return std::make_shared<AbsolutePose3DStampedConstraint>(*this); }
 protected:
  Vector7d mean_;
  Mat<6, 6> sqrt_information_;
};

// AbsoluteConstraint<V> / RelativeConstraint<V> for the 3-vector variables; the type() strings are the
// ones the reference forces (absolute_constraint_impl.h:13-41, relative_constraint_impl.h:11-27)
class AbsoluteVec3Constraint : public fuse_core::Constraint {
 public:
  AbsoluteVec3Constraint(const std::string& type_name, const std::string& source, const fuse_core::Variable& v, const Vec3& mean,
                         const Mat<3, 3>& covariance)
      : Constraint(source, {v.uuid()}), type_(type_name), mean_(mean) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return type_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    bs_constraints::appendBlocks(*this, block_of, t.idx[BSGPU_F_ABS_VEC3]);
    auto& c = t.consts[BSGPU_F_ABS_VEC3];
    c.insert(c.end(), mean_.begin(), mean_.end());
    bs_constraints::appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_ABS_VEC3, loss());
  }
  SharedPtr clone() const override { return std::make_shared<AbsoluteVec3Constraint>(*this); }
 protected:
  std::string type_;
  Vec3 mean_;
  Mat<3, 3> sqrt_information_;
};
class RelativeVec3Constraint : public fuse_core::Constraint {
 public:
  RelativeVec3Constraint(const std::string& type_name, const std::string& source, const fuse_core::Variable& v1,
                         const fuse_core::Variable& v2, const Vec3& delta, const Mat<3, 3>& covariance)
      : Constraint(source, {v1.uuid(), v2.uuid()}), type_(type_name), delta_(delta) {
    if (!bs_math::sqrtInformationUpper(covariance, sqrt_information_)) throw std::invalid_argument("covariance is not positive definite");
  }
  std::string type() const override { return type_; }
  void pack(const BlockOf& block_of, fuse_core::FactorTables& t) const override {
    bs_constraints::appendBlocks(*this, block_of, t.idx[BSGPU_F_REL_VEC3]);
    auto& c = t.consts[BSGPU_F_REL_VEC3];
    c.insert(c.end(), delta_.begin(), delta_.end());
    bs_constraints::appendMat(c, sqrt_information_);
    t.pushLoss(BSGPU_F_REL_VEC3, loss());
  }
  SharedPtr clone() const override { return std::make_shared<RelativeVec3Constraint>(*this); }
 protected:
  std::string type_;
  Vec3 delta_;
  Mat<3, 3> sqrt_information_;
};
}  // namespace fuse_constraints

namespace bs_constraints {
// the reference's aliases (global/absolute_constraint.h:10-25, relative_pose/relative_constraints.h:12-19)
inline fuse_core::Constraint::SharedPtr AbsoluteVelocityLinear3DStampedConstraint(const std::string& src, const fuse_variables::VelocityLinear3DStamped& v, const Vec3& mean, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::AbsoluteVec3Constraint>("fuse_constraints::AbsoluteVelocityLinear3DStampedConstraint", src, v, mean, cov); }
inline fuse_core::Constraint::SharedPtr AbsoluteGyroBias3DStampedConstraint(const std::string& src, const bs_variables::GyroscopeBias3DStamped& v, const Vec3& mean, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::AbsoluteVec3Constraint>("fuse_constraints::AbsoluteGyroBias3DStampedConstraint", src, v, mean, cov); }
inline fuse_core::Constraint::SharedPtr AbsoluteAccelBias3DStampedConstraint(const std::string& src, const bs_variables::AccelerationBias3DStamped& v, const Vec3& mean, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::AbsoluteVec3Constraint>("fuse_constraints::AbsoluteAccelBias3DStampedConstraint", src, v, mean, cov); }
inline fuse_core::Constraint::SharedPtr RelativeVelocityLinear3DStampedConstraint(const std::string& src, const fuse_variables::VelocityLinear3DStamped& a, const fuse_variables::VelocityLinear3DStamped& b, const Vec3& d, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::RelativeVec3Constraint>("fuse_constraints::RelativeVelocityLinear3DStampedConstraint", src, a, b, d, cov); }
inline fuse_core::Constraint::SharedPtr RelativeGyroBias3DStampedConstraint(const std::string& src, const bs_variables::GyroscopeBias3DStamped& a, const bs_variables::GyroscopeBias3DStamped& b, const Vec3& d, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::RelativeVec3Constraint>("fuse_constraints::RelativeGyroBias3DStampedConstraint", src, a, b, d, cov); }
inline fuse_core::Constraint::SharedPtr RelativeAccelBias3DStampedConstraint(const std::string& src, const bs_variables::AccelerationBias3DStamped& a, const bs_variables::AccelerationBias3DStamped& b, const Vec3& d, const Mat<3, 3>& cov) {
  return std::make_shared<fuse_constraints::RelativeVec3Constraint>("fuse_constraints::RelativeAccelBias3DStampedConstraint", src, a, b, d, cov); }
}  // namespace bs_constraints
