// bs_optimizers::GpuGraph — the fuse_core::Graph the reference's optimizer would be handed instead of
// fuse_graphs::HashGraph (bs_optimizers/src/fixed_lag_smoother_node.cpp:42-45).  Graph bookkeeping on
// the host (UUID maps, like HashGraph [EXT]); optimize() flattens to the IR of include/bsgpu.h with the
// deterministic block order of SURVEY.md §8a A17 and runs the solve on the GPU through the C-ABI.
//
// Back-end selection: the product builds against libbsgpu (prefix bsgpu_).  The CPU-side unit tests
// of this host logic compile the same header with -DBS_BACKEND_PREFIX=bso_ against the test oracle,
// because no GPU exists where they run; that configuration is test infrastructure only.
#pragma once
#include <chrono>

#include "bs_constraints.h"

#ifndef BS_BACKEND_PREFIX
#define BS_BACKEND_PREFIX bsgpu_
#endif
#define BS_CAT2(a, b) a##b
#define BS_CAT(a, b) BS_CAT2(a, b)
#define BS_API(name) BS_CAT(BS_BACKEND_PREFIX, name)

extern "C" {  // declarations for the oracle-prefixed build (identical signatures; bsgpu_* come from bsgpu.h)
bsgpu_ctx* BS_API(create)(int);
void BS_API(destroy)(bsgpu_ctx*);
const char* BS_API(last_error)(const bsgpu_ctx*);
int BS_API(clear)(bsgpu_ctx*);
int BS_API(set_blocks)(bsgpu_ctx*, int32_t, const double*, const int32_t*, const uint8_t*, const uint8_t*, const uint8_t*);
int BS_API(set_cameras)(bsgpu_ctx*, int32_t, const bsgpu_camera*);
int BS_API(add_factors)(bsgpu_ctx*, int32_t, int32_t, const int32_t*, const double*, const int32_t*, const double*);
int BS_API(solve)(bsgpu_ctx*, const bsgpu_options*, bsgpu_summary*);
int BS_API(get_blocks)(bsgpu_ctx*, double*, int64_t);
void BS_API(options_default)(bsgpu_options*);
int BS_API(nidx)(int);
int BS_API(covariance)(bsgpu_ctx*, int32_t, int32_t, double*);
int BS_API(add_marginal)(bsgpu_ctx*, int32_t, const int32_t*, int32_t, const double*, const double*, const double*);
int BS_API(marginalize)(bsgpu_ctx*, int32_t, const int32_t*, int32_t*, int32_t*, int32_t*);
int BS_API(get_marginal)(const bsgpu_ctx*, int32_t*, double*, double*, double*);
}

namespace ceres_compat {  // the ceres::Solver fields the reference sets (vio.yaml:7-17) and reads (fixed_lag_smoother.cpp:286,705-716)
enum TerminationType { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };
struct SolverOptions {
  int max_num_iterations = 50;
  double max_solver_time_in_seconds = 1e9;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  int num_threads = 1;                 // accepted and ignored: the device decides its own parallelism
  static SolverOptions Vio() {         // beam_slam_launch/config/vio.yaml:7-17
    SolverOptions o; o.max_num_iterations = 10; o.max_solver_time_in_seconds = 0.05;
    o.function_tolerance = o.gradient_tolerance = o.parameter_tolerance = 1.5e-7; o.num_threads = 6; return o;
  }
};
struct IterationSummary { int iteration = 0; double cost = 0; bool step_is_successful = false; };
struct SolverSummary {
  TerminationType termination_type = FAILURE;
  double initial_cost = 0, final_cost = 0, total_time_in_seconds = 0;
  std::vector<IterationSummary> iterations;
  std::string message;
  bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE; }
  std::string FullReport() const {
    std::ostringstream s;
    s << "bsgpu solve: " << (termination_type == CONVERGENCE ? "CONVERGENCE" : termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE")
      << " (" << message << ") iterations " << (iterations.empty() ? 0 : iterations.size() - 1) << " cost " << initial_cost << " -> " << final_cost
      << " time " << total_time_in_seconds << " s";
    return s.str();
  }
};
}  // namespace ceres_compat

namespace bs_optimizers {

class GpuGraph {
 public:
  using UniquePtr = std::unique_ptr<GpuGraph>;
  static UniquePtr make_unique(int device = 0) { return UniquePtr(new GpuGraph(device)); }
  explicit GpuGraph(int device = 0) : device_(device), ctx_(BS_API(create)(device)) {
    if (!ctx_) throw std::runtime_error("GpuGraph: no usable back-end (libbsgpu has no CPU fallback)");
  }
  ~GpuGraph() { if (ctx_) BS_API(destroy)(ctx_); }
  GpuGraph(const GpuGraph&) = delete;
  GpuGraph& operator=(const GpuGraph&) = delete;

  // ---- fuse_core::Graph surface used by the reference ---------------------------------
  void clear() { variables_.clear(); constraints_.clear(); by_variable_.clear(); on_hold_.clear(); }
  bool variableExists(const fuse_core::UUID& u) const { return variables_.count(u) != 0; }
  bool constraintExists(const fuse_core::UUID& u) const { return constraints_.count(u) != 0; }
  const fuse_core::Variable& getVariable(const fuse_core::UUID& u) const {
    auto it = variables_.find(u);
    if (it == variables_.end()) throw std::out_of_range("variable not in graph");
    return *it->second;
  }
  fuse_core::Variable& getVariable(const fuse_core::UUID& u) { return const_cast<fuse_core::Variable&>(static_cast<const GpuGraph*>(this)->getVariable(u)); }
  std::vector<const fuse_core::Variable*> getVariables() const { std::vector<const fuse_core::Variable*> v; for (auto& kv : variables_) v.push_back(kv.second.get()); return v; }
  std::vector<const fuse_core::Constraint*> getConstraints() const { std::vector<const fuse_core::Constraint*> v; for (auto& kv : constraints_) v.push_back(kv.second.get()); return v; }
  std::vector<const fuse_core::Constraint*> getConnectedConstraints(const fuse_core::UUID& var) const {
    if (!variableExists(var)) throw std::logic_error("getConnectedConstraints: variable not in graph");
    std::vector<const fuse_core::Constraint*> out;
    auto it = by_variable_.find(var);
    if (it != by_variable_.end()) for (const auto& cu : it->second) out.push_back(constraints_.at(cu).get());
    return out;
  }
  bool addVariable(fuse_core::Variable::SharedPtr v) {
    auto it = variables_.find(v->uuid());
    if (it != variables_.end()) { std::memcpy(it->second->data(), v->data(), v->size() * sizeof(double)); return false; }  // HashGraph: overwrite value
    variables_[v->uuid()] = std::move(v);
    return true;
  }
  bool removeVariable(const fuse_core::UUID& u) {
    auto it = variables_.find(u);
    if (it == variables_.end()) return false;
    auto cit = by_variable_.find(u);
    if (cit != by_variable_.end() && !cit->second.empty()) throw std::logic_error("removeVariable: variable still used by a constraint");
    by_variable_.erase(u); on_hold_.erase(u); variables_.erase(it);
    return true;
  }
  bool addConstraint(fuse_core::Constraint::SharedPtr c) {
    if (constraints_.count(c->uuid())) return false;
    for (const auto& u : c->variables()) if (!variableExists(u)) throw std::logic_error("addConstraint: constraint " + c->type() + " uses a variable that is not in the graph");
    for (const auto& u : c->variables()) by_variable_[u].insert(c->uuid());
    constraints_[c->uuid()] = std::move(c);
    return true;
  }
  bool removeConstraint(const fuse_core::UUID& u) {
    auto it = constraints_.find(u);
    if (it == constraints_.end()) return false;
    for (const auto& v : it->second->variables()) by_variable_[v].erase(u);
    constraints_.erase(it);
    return true;
  }
  void holdVariable(const fuse_core::UUID& u, bool hold = true) { if (hold) on_hold_.insert(u); else on_hold_.erase(u); }
  // Graph::update(transaction): removals first, then additions ([EXT] fuse_core::Graph::update)
  void update(const fuse_core::Transaction& t) {
    for (const auto& u : t.removedConstraints()) removeConstraint(u);
    for (const auto& u : t.removedVariables()) removeVariable(u);
    for (const auto& v : t.addedVariables()) addVariable(v->clone());
    for (const auto& c : t.addedConstraints()) addConstraint(c->clone());
  }
  // Graph::clone(): deep copy of variables and constraints (fixed_lag_smoother.cpp:308 does this every cycle)
  UniquePtr clone() const {
    UniquePtr g(new GpuGraph(device_));
    for (auto& kv : variables_) g->variables_[kv.first] = kv.second->clone();
    for (auto& kv : constraints_) g->constraints_[kv.first] = kv.second->clone();
    g->by_variable_ = by_variable_; g->on_hold_ = on_hold_;
    return g;
  }
  size_t numVariables() const { return variables_.size(); }
  size_t numConstraints() const { return constraints_.size(); }
  void print(std::ostream& s) const {
    s << "GpuGraph\n  variables:\n";
    for (auto& kv : variables_) { s << "   - "; kv.second->print(s); s << "\n"; }
    s << "  constraints:\n";
    for (auto& kv : constraints_) { s << "   - "; kv.second->print(s); s << "\n"; }
  }

  // ---- the hot call -----------------------------------------------------------------------------------
  // Deterministic block order: keyframes ascending by stamp, (q,p,v,bg,ba) inside one
  // keyframe (ImuState::GetStateVector, imu_state.cpp:348-354); then landmarks ascending by id
  // (graph_access.cpp:200-216); unstamped extrinsic blocks last.
  std::vector<const fuse_core::Variable*> orderedVariables() const {
    std::vector<const fuse_core::Variable*> v = getVariables();
    auto cls = [](const fuse_core::Variable* a) { return a->isStamped() ? 0 : a->isLandmark() ? 1 : 2; };
    std::stable_sort(v.begin(), v.end(), [&](const fuse_core::Variable* a, const fuse_core::Variable* b) {
      const int ca = cls(a), cb = cls(b);
      if (ca != cb) return ca < cb;
      if (ca == 0) { if (a->stamp() != b->stamp()) return a->stamp() < b->stamp(); if (a->stateSlot() != b->stateSlot()) return a->stateSlot() < b->stateSlot(); }
      if (ca == 1 && a->landmarkId() != b->landmarkId()) return a->landmarkId() < b->landmarkId();
      return a->uuid() < b->uuid();
    });
    return v;
  }

  // Flat IR of the current graph (variables at their current values) loaded into the back-end context
  struct Flat {
    std::vector<const fuse_core::Variable*> vars;
    std::vector<double> values;
    std::vector<int32_t> offset;
    std::vector<uint8_t> size, manifold, is_const;
    std::map<fuse_core::UUID, int32_t> block_index;
  };
  bool flatten(Flat& f) {
    f.vars = orderedVariables();
    for (const auto* v : f.vars) {
      f.block_index[v->uuid()] = (int32_t)f.offset.size();
      f.offset.push_back((int32_t)f.values.size());
      f.size.push_back((uint8_t)v->size());
      f.manifold.push_back((uint8_t)v->manifold());
      f.is_const.push_back((v->holdConstant() || on_hold_.count(v->uuid())) ? 1 : 0);
      f.values.insert(f.values.end(), v->data(), v->data() + v->size());
    }
    if (f.offset.empty()) return false;
    fuse_core::FactorTables t;
    auto block_of = [&](const fuse_core::UUID& u) { return f.block_index.at(u); };
    for (auto& kv : constraints_) kv.second->pack(block_of, t);
    check(BS_API(clear)(ctx_));
    check(BS_API(set_blocks)(ctx_, (int32_t)f.offset.size(), f.values.data(), f.offset.data(), f.size.data(), f.manifold.data(), f.is_const.data()));
    if (!t.cameras.empty()) check(BS_API(set_cameras)(ctx_, (int32_t)t.cameras.size(), t.cameras.data()));
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty)
      if (t.count(ty)) check(BS_API(add_factors)(ctx_, ty, t.count(ty), t.idx[ty].data(), t.consts[ty].data(), t.loss_kind[ty].data(), t.loss_a[ty].data()));
    for (const auto& m : t.marginals)
      check(BS_API(add_marginal)(ctx_, (int32_t)m.blocks.size(), m.blocks.data(), m.rows, m.A.data(), m.b.data(), m.xbar.data()));
    return true;
  }

  ceres_compat::SolverSummary optimize(const ceres_compat::SolverOptions& o = ceres_compat::SolverOptions()) {
    const auto t0 = std::chrono::steady_clock::now();
    Flat f;
    ceres_compat::SolverSummary s;
    if (!flatten(f)) { s.termination_type = ceres_compat::CONVERGENCE; s.message = "empty graph"; return s; }
    auto& vars = f.vars; auto& values = f.values; auto& offset = f.offset; auto& size = f.size;
    bsgpu_options bo;
    BS_API(options_default)(&bo);
    bo.max_num_iterations = o.max_num_iterations; bo.max_solver_time_in_seconds = o.max_solver_time_in_seconds;
    bo.function_tolerance = o.function_tolerance; bo.gradient_tolerance = o.gradient_tolerance; bo.parameter_tolerance = o.parameter_tolerance;
    bsgpu_summary bs;
    check(BS_API(solve)(ctx_, &bo, &bs));
    check(BS_API(get_blocks)(ctx_, values.data(), (int64_t)values.size()));
    for (size_t i = 0; i < vars.size(); ++i)   // Variable::data() updated in place, like Ceres does through the raw pointers
      std::memcpy(variables_.at(vars[i]->uuid())->data(), values.data() + offset[i], size[i] * sizeof(double));
    s.termination_type = bs.termination_type == BSGPU_CONVERGENCE ? ceres_compat::CONVERGENCE
                       : bs.termination_type == BSGPU_NO_CONVERGENCE ? ceres_compat::NO_CONVERGENCE : ceres_compat::FAILURE;
    s.initial_cost = bs.initial_cost; s.final_cost = bs.final_cost;
    s.iterations.resize(bs.num_iterations + 1);
    s.message = bs.message;
    s.total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    last_summary_ = bs;
    return s;
  }
  // Graph::optimizeFor(duration, options): same with max_solver_time_in_seconds = duration
  ceres_compat::SolverSummary optimizeFor(double max_seconds, ceres_compat::SolverOptions o = ceres_compat::SolverOptions()) {
    o.max_solver_time_in_seconds = max_seconds;
    return optimize(o);
  }
  const bsgpu_summary& lastBackendSummary() const { return last_summary_; }

  // [EXT] fuse_constraints::marginalizeVariables(source, marginalized_variables, graph) (fixed_lag_smoother.cpp:270-271):
  // the transaction that removes `to_marginalize` and every constraint touching them and adds ONE MarginalConstraint
  // on the remaining variables those constraints touch, linearised at the variables' current values.  The Schur
  // complement runs on the device (bsgpu_marginalize); variables no constraint touches are simply removed.
  fuse_core::Transaction marginalizeVariables(const std::string& source, const std::vector<fuse_core::UUID>& to_marginalize) {
    fuse_core::Transaction tr;
    std::vector<fuse_core::UUID> constrained;
    std::set<fuse_core::UUID> removed_constraints;
    for (const auto& u : to_marginalize) {
      if (!variableExists(u)) throw std::out_of_range("marginalizeVariables: variable not in graph");
      const auto it = by_variable_.find(u);
      if (it != by_variable_.end() && !it->second.empty()) {
        constrained.push_back(u);
        for (const auto& cu : it->second) if (removed_constraints.insert(cu).second) tr.removeConstraint(cu);
      }
      tr.removeVariable(u);
    }
    if (constrained.empty()) return tr;
    Flat f;
    if (!flatten(f)) return tr;
    std::vector<int32_t> marg;
    for (const auto& u : constrained) {
      const int32_t b = f.block_index.at(u);
      if (f.is_const[b]) throw std::logic_error("marginalizeVariables: variable is held constant");
      marg.push_back(b);
    }
    int32_t n_kept = 0, n_rows = 0, n_cols = 0;
    check(BS_API(marginalize)(ctx_, (int32_t)marg.size(), marg.data(), &n_kept, &n_rows, &n_cols));
    if (n_rows == 0) return tr;
    std::vector<int32_t> kept(n_kept);
    check(BS_API(get_marginal)(ctx_, kept.data(), nullptr, nullptr, nullptr));
    size_t amb = 0;
    for (int32_t b : kept) amb += f.size[b];
    std::vector<double> A((size_t)n_rows * n_cols), bvec(n_rows), xbar(amb);
    check(BS_API(get_marginal)(ctx_, kept.data(), A.data(), bvec.data(), xbar.data()));
    std::vector<fuse_core::UUID> kept_ids;
    for (int32_t b : kept) kept_ids.push_back(f.vars[b]->uuid());
    tr.addConstraint(std::make_shared<fuse_constraints::MarginalConstraint>(source, std::move(kept_ids), n_rows, n_cols, std::move(A), std::move(bvec), std::move(xbar)));
    return tr;
  }

  // fuse_core::Graph::getCovariance(covariance_requests, covariance_matrices) in tangent space (the form
  // bs_publishers/src/odometry_3d_publisher.cpp:82 consumes): one row-major localSize(a) x localSize(b) matrix per
  // requested pair, marginal covariance at the variables' current values.  Pose-side variables only: landmarks are
  // eliminated by the solver (throws, like fuse does for an uncomputable request).
  void getCovariance(const std::vector<std::pair<fuse_core::UUID, fuse_core::UUID>>& covariance_requests,
                     std::vector<std::vector<double>>& covariance_matrices) {
    Flat f;
    covariance_matrices.clear();
    if (covariance_requests.empty()) return;
    if (!flatten(f)) throw std::runtime_error("getCovariance: empty graph");
    for (const auto& rq : covariance_requests) {
      const auto a = f.block_index.find(rq.first), b = f.block_index.find(rq.second);
      if (a == f.block_index.end() || b == f.block_index.end()) throw std::out_of_range("getCovariance: variable not in graph");
      const size_t ta = variables_.at(rq.first)->localSize(), tb = variables_.at(rq.second)->localSize();
      std::vector<double> m(ta * tb);
      check(BS_API(covariance)(ctx_, a->second, b->second, m.data()));
      covariance_matrices.push_back(std::move(m));
    }
  }

 private:
  void check(int rc) { if (rc != BSGPU_OK) throw std::runtime_error(std::string("bsgpu: ") + BS_API(last_error)(ctx_)); }
  int device_;
  bsgpu_ctx* ctx_;
  std::map<fuse_core::UUID, fuse_core::Variable::SharedPtr> variables_;
  std::map<fuse_core::UUID, fuse_core::Constraint::SharedPtr> constraints_;
  std::map<fuse_core::UUID, std::set<fuse_core::UUID>> by_variable_;
  std::set<fuse_core::UUID> on_hold_;
  bsgpu_summary last_summary_{};
};

}  // namespace bs_optimizers

namespace fuse_constraints {
// the reference's call shape: fuse_constraints::marginalizeVariables(ros::this_node::getName(), vars, *graph_)
inline fuse_core::Transaction marginalizeVariables(const std::string& source, const std::vector<fuse_core::UUID>& marginalized_variables,
                                                   bs_optimizers::GpuGraph& graph) {
  return graph.marginalizeVariables(source, marginalized_variables);
}
}  // namespace fuse_constraints
