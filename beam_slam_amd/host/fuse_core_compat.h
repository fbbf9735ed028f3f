// ROS-free, dependency-free look-alikes of the fuse_core types the reference's solve path is written
// against (fuse is an absent git submodule: /root/reference/.gitmodules:4-6, dependencies/fuse empty).
// Same names, member functions and semantics as far as the path uses them:
//   fuse_core::Variable     data(), size(), localSize(), holdConstant(), uuid(), type()
//   fuse_core::Constraint   type(), uuid(), source(), variables(), loss(), print()  + pack()
//   fuse_core::Loss         CauchyLoss / HuberLoss / (null = trivial)
//   fuse_core::Transaction  stamp(), addVariable/addConstraint/removeVariable/removeConstraint, merge(), involved stamps
//   fuse_core::Graph        implemented by bs_optimizers::GpuGraph (gpu_graph.h)
// This is host-side plumbing; all arithmetic of the solve happens behind include/bsgpu.h.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <ostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bsgpu.h"

namespace fuse_core {

// ---- time ----------------------------------------------------------------------------------------
struct Time {  // ros::Time stand-in (integer nanoseconds; total order, exact arithmetic)
  int64_t ns = 0;
  Time() = default;
  explicit Time(double sec) : ns((int64_t)(sec * 1e9 + (sec >= 0 ? 0.5 : -0.5))) {}
  static Time fromNSec(int64_t n) { Time t; t.ns = n; return t; }
  double toSec() const { return ns * 1e-9; }
  bool isZero() const { return ns == 0; }
  bool operator<(const Time& o) const { return ns < o.ns; }
  bool operator>(const Time& o) const { return ns > o.ns; }
  bool operator<=(const Time& o) const { return ns <= o.ns; }
  bool operator>=(const Time& o) const { return ns >= o.ns; }
  bool operator==(const Time& o) const { return ns == o.ns; }
  bool operator!=(const Time& o) const { return ns != o.ns; }
};
inline Time operator+(const Time& t, double dsec) { return Time::fromNSec(t.ns + Time(dsec).ns); }
inline double operator-(const Time& a, const Time& b) { return (a.ns - b.ns) * 1e-9; }

// ---- UUID ----------------------------------------------------------------------------------------
// fuse_core::uuid::generate(type, stamp, device_id) is a name-based UUID.  Only identity within one
// graph matters to the solve path, so a deterministic 128-bit FNV-1a name hash stands in for it.
struct UUID {
  uint64_t hi = 0, lo = 0;
  bool operator<(const UUID& o) const { return hi != o.hi ? hi < o.hi : lo < o.lo; }
  bool operator==(const UUID& o) const { return hi == o.hi && lo == o.lo; }
  bool operator!=(const UUID& o) const { return !(*this == o); }
};
inline std::ostream& operator<<(std::ostream& s, const UUID& u) {
  char b[40];
  std::snprintf(b, sizeof(b), "%016llx%016llx", (unsigned long long)u.hi, (unsigned long long)u.lo);
  return s << b;
}
namespace uuid {
inline UUID generate(const std::string& name) {
  UUID u;
  uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
  for (unsigned char ch : name) { a = (a ^ ch) * 0x100000001b3ull; b = (b ^ (ch + 0x9e)) * 0x100000001b3ull; b ^= a >> 29; }
  u.hi = a; u.lo = b;
  return u;
}
inline UUID generate(const std::string& type, const Time& stamp, const UUID& device = UUID()) {
  std::ostringstream s;
  s << type << "@" << stamp.ns << "#" << device;
  return generate(s.str());
}
inline UUID generate(const std::string& type, uint64_t id) { return generate(type + "/" + std::to_string(id)); }
inline UUID generate() { static uint64_t counter = 0; return generate("anonymous", ++counter); }
}  // namespace uuid

// ---- Loss ----------------------------------------------------------------------------------------
class Loss {
 public:
  using SharedPtr = std::shared_ptr<Loss>;
  virtual ~Loss() = default;
  virtual std::string type() const = 0;
  virtual int kind() const = 0;   // BSGPU_LOSS_*
  virtual double a() const = 0;
  virtual void print(std::ostream& s) const { s << type() << "(a=" << a() << ")"; }
};
}  // namespace fuse_core
namespace fuse_loss {
class CauchyLoss : public fuse_core::Loss {  // -> ceres::CauchyLoss(a)
 public:
  explicit CauchyLoss(double a = 1.0) : a_(a) {}  // default-constructed at pose_3d_stamped_transaction.cpp:18
  std::string type() const override { return "fuse_loss::CauchyLoss"; }
  int kind() const override { return BSGPU_LOSS_CAUCHY; }
  double a() const override { return a_; }
 private:
  double a_;
};
class HuberLoss : public fuse_core::Loss {
 public:
  explicit HuberLoss(double a = 1.0) : a_(a) {}
  std::string type() const override { return "fuse_loss::HuberLoss"; }
  int kind() const override { return BSGPU_LOSS_HUBER; }
  double a() const override { return a_; }
 private:
  double a_;
};
}  // namespace fuse_loss

namespace fuse_core {

// ---- Variable --------------------------------------------------------------------------------------
class Variable {
 public:
  using SharedPtr = std::shared_ptr<Variable>;
  explicit Variable(const UUID& uuid) : uuid_(uuid) {}
  virtual ~Variable() = default;
  const UUID& uuid() const { return uuid_; }
  virtual std::string type() const = 0;
  virtual size_t size() const = 0;
  virtual size_t localSize() const { return size(); }
  virtual const double* data() const = 0;
  virtual double* data() = 0;
  virtual int manifold() const { return BSGPU_MANIFOLD_EUCLIDEAN; }  // localParameterization() stand-in
  virtual bool holdConstant() const { return false; }
  virtual SharedPtr clone() const = 0;
  // in-place copy for a graph that clones all its variables into ONE allocation (GpuGraph::clone): the bytes needed, and the copy
  // constructed at `mem` (suitably aligned, at least cloneSize() bytes).  0 / nullptr: the type only offers clone().
  virtual size_t cloneSize() const { return 0; }
  virtual Variable* cloneAt(void* mem) const { (void)mem; return nullptr; }
  // a copy made by cloneAt() holds nothing that a destructor must release (its members are plain values): the slab such copies live in may
  // drop them without 51 000 virtual destructor calls
  virtual bool destroyIsNoop() const { return false; }
  // ordering keys for the deterministic block index
  virtual bool isStamped() const { return false; }
  virtual Time stamp() const { return Time(); }
  virtual int stateSlot() const { return 99; }       // q,p,v,bg,ba = 0..4 inside one keyframe
  virtual bool isLandmark() const { return false; }
  virtual uint64_t landmarkId() const { return 0; }
  virtual void print(std::ostream& s) const {
    s << type() << " uuid " << uuid() << " [";
    for (size_t i = 0; i < size(); ++i) s << (i ? ", " : "") << data()[i];
    s << "]";
  }
 private:
  UUID uuid_;
};

// ---- Constraint ------------------------------------------------------------------------------------
struct FactorTables;  // bs_constraints/gpu_pack (below)
class Constraint;

// Resolves the i-th variable of a constraint to its index in the flat block table: either through block indices the
// graph resolved beforehand (the per-cycle path: no UUID lookups) or through a uuid -> index function.
class BlockOf {
 public:
  explicit BlockOf(const int32_t* resolved) : resolved_(resolved) {}
  template <class F, class = typename std::enable_if<std::is_convertible<decltype(std::declval<F&>()(std::declval<const UUID&>())), int32_t>::value>::type>
  BlockOf(F fn) : fn_(std::move(fn)) {}
  inline int32_t operator()(const Constraint& c, size_t i) const;
 private:
  const int32_t* resolved_ = nullptr;
  std::function<int32_t(const UUID&)> fn_;
};

class Constraint {
 public:
  using SharedPtr = std::shared_ptr<Constraint>;
  Constraint(const std::string& source, std::vector<UUID> variables)
      : source_(source), uuid_(uuid::generate()), variables_(std::move(variables)) {}
  virtual ~Constraint() = default;
  virtual std::string type() const = 0;
  const UUID& uuid() const { return uuid_; }
  const std::string& source() const { return source_; }
  const std::vector<UUID>& variables() const { return variables_; }  // = Ceres parameter-block order
  const Loss::SharedPtr& loss() const { return loss_; }
  void loss(Loss::SharedPtr l) { loss_ = std::move(l); }
  virtual void print(std::ostream& s) const { s << type() << " source " << source_ << " uuid " << uuid_; }
  // The reference hands Ceres a heap-allocated CostFunction per call (costFunction()); the GPU path
  // needs the packed payload instead.
  virtual void pack(const BlockOf& block_of, FactorTables& out) const = 0;
  virtual SharedPtr clone() const = 0;
 protected:
  std::string source_;
  UUID uuid_;
  std::vector<UUID> variables_;
  Loss::SharedPtr loss_;
};

inline int32_t BlockOf::operator()(const Constraint& c, size_t i) const { return resolved_ ? resolved_[i] : fn_(c.variables()[i]); }

// flat per-type tables a Graph hands to bsgpu_add_factors
struct FactorTables {
  std::vector<int32_t> idx[BSGPU_F_NUM_TYPES];
  std::vector<double> consts[BSGPU_F_NUM_TYPES];
  std::vector<int32_t> loss_kind[BSGPU_F_NUM_TYPES];
  std::vector<double> loss_a[BSGPU_F_NUM_TYPES];
  struct MarginalEntry { std::vector<int32_t> blocks; int rows = 0; std::vector<double> A, b, xbar; };
  std::vector<MarginalEntry> marginals;   // dense linear priors (bsgpu_add_marginal)
  std::vector<bsgpu_camera> cameras;
  int32_t cameraId(const bsgpu_camera& c) {
    for (size_t i = 0; i < cameras.size(); ++i)
      if (std::memcmp(&cameras[i], &c, sizeof(c)) == 0) return (int32_t)i;
    cameras.push_back(c);
    return (int32_t)cameras.size() - 1;
  }
  void pushLoss(int type, const Loss::SharedPtr& l) {
    loss_kind[type].push_back(l ? l->kind() : BSGPU_LOSS_TRIVIAL);
    loss_a[type].push_back(l ? l->a() : 1.0);
  }
  int count(int type) const { return (int)loss_kind[type].size(); }
};

// ---- Transaction -----------------------------------------------------------------------------------
class Transaction {
 public:
  using SharedPtr = std::shared_ptr<Transaction>;
  const Time& stamp() const { return stamp_; }
  void stamp(const Time& t) { stamp_ = t; }
  void addInvolvedStamp(const Time& t) { involved_.insert(t); }
  const std::set<Time>& involvedStamps() const { return involved_; }
  Time minStamp() const { return involved_.empty() ? stamp_ : std::min(stamp_, *involved_.begin()); }
  Time maxStamp() const { return involved_.empty() ? stamp_ : std::max(stamp_, *involved_.rbegin()); }
  void addVariable(Variable::SharedPtr v, bool overwrite = false) {
    for (auto& e : added_variables_) if (e->uuid() == v->uuid()) { if (overwrite) e = v; return; }
    added_variables_.push_back(std::move(v));
  }
  void addConstraint(Constraint::SharedPtr c) { added_constraints_.push_back(std::move(c)); }
  void removeVariable(const UUID& u) { removed_variables_.push_back(u); }
  void removeConstraint(const UUID& u) { removed_constraints_.push_back(u); }
  const std::vector<Variable::SharedPtr>& addedVariables() const { return added_variables_; }
  const std::vector<Constraint::SharedPtr>& addedConstraints() const { return added_constraints_; }
  const std::vector<UUID>& removedVariables() const { return removed_variables_; }
  const std::vector<UUID>& removedConstraints() const { return removed_constraints_; }
  bool empty() const { return added_variables_.empty() && added_constraints_.empty() && removed_variables_.empty() && removed_constraints_.empty(); }
  void merge(const Transaction& o, bool overwrite = false) {
    stamp_ = std::max(stamp_, o.stamp_);
    involved_.insert(o.involved_.begin(), o.involved_.end());
    for (const auto& v : o.added_variables_) addVariable(v, overwrite);
    for (const auto& c : o.added_constraints_) addConstraint(c);
    removed_variables_.insert(removed_variables_.end(), o.removed_variables_.begin(), o.removed_variables_.end());
    removed_constraints_.insert(removed_constraints_.end(), o.removed_constraints_.begin(), o.removed_constraints_.end());
  }
 private:
  Time stamp_;
  std::set<Time> involved_;
  std::vector<Variable::SharedPtr> added_variables_;
  std::vector<Constraint::SharedPtr> added_constraints_;
  std::vector<UUID> removed_variables_, removed_constraints_;
};

}  // namespace fuse_core
