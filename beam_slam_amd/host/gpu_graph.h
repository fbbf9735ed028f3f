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
#include <cstdio>
#include <cstdlib>

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
  void clear() {
    variables_.clear(); constraints_.clear(); by_variable_.clear(); connectivity_valid_ = true; on_hold_.clear(); ordered_.clear();
    for (auto& t : tables_) t = TypeTable();
    marginal_rows_.clear();
  }
  bool variableExists(const fuse_core::UUID& u) const { return variables_.count(u) != 0; }
  bool constraintExists(const fuse_core::UUID& u) const { return constraints_.count(u) != 0; }
  const fuse_core::Variable& getVariable(const fuse_core::UUID& u) const {
    auto it = variables_.find(u);
    if (it == variables_.end()) throw std::out_of_range("variable not in graph");
    return *it->second;
  }
  fuse_core::Variable& getVariable(const fuse_core::UUID& u) { return const_cast<fuse_core::Variable&>(static_cast<const GpuGraph*>(this)->getVariable(u)); }
  std::vector<const fuse_core::Variable*> getVariables() const { std::vector<const fuse_core::Variable*> v; for (auto& kv : variables_) v.push_back(kv.second.get()); return v; }
  std::vector<const fuse_core::Constraint*> getConstraints() const { std::vector<const fuse_core::Constraint*> v; for (auto& kv : constraints_) v.push_back(kv.second.c.get()); return v; }
  std::vector<const fuse_core::Constraint*> getConnectedConstraints(const fuse_core::UUID& var) const {
    if (!variableExists(var)) throw std::logic_error("getConnectedConstraints: variable not in graph");
    std::vector<const fuse_core::Constraint*> out;
    ensureConnectivity();
    auto it = by_variable_.find(var);
    if (it != by_variable_.end()) for (const auto& cu : it->second) out.push_back(constraints_.at(cu).c.get());
    return out;
  }
  bool addVariable(fuse_core::Variable::SharedPtr v) {
    auto it = variables_.find(v->uuid());
    if (it != variables_.end()) { std::memcpy(it->second->data(), v->data(), v->size() * sizeof(double)); return false; }  // HashGraph: overwrite value
    const fuse_core::Variable* raw = v.get();
    variables_[v->uuid()] = std::move(v);
    ordered_.insert(std::upper_bound(ordered_.begin(), ordered_.end(), raw, orderBefore), raw);
    return true;
  }
  bool removeVariable(const fuse_core::UUID& u) {
    auto it = variables_.find(u);
    if (it == variables_.end()) return false;
    ensureConnectivity();
    auto cit = by_variable_.find(u);
    if (cit != by_variable_.end() && !cit->second.empty()) throw std::logic_error("removeVariable: variable still used by a constraint");
    {
      const fuse_core::Variable* raw = it->second.get();
      auto pos = std::lower_bound(ordered_.begin(), ordered_.end(), raw, orderBefore);
      while (pos != ordered_.end() && *pos != raw) ++pos;   // (equal keys cannot occur: the uuid breaks every tie)
      if (pos != ordered_.end()) ordered_.erase(pos);
    }
    by_variable_.erase(u); on_hold_.erase(u); variables_.erase(it);
    return true;
  }
  bool addConstraint(fuse_core::Constraint::SharedPtr c) {
    if (constraints_.count(c->uuid())) return false;
    CEntry e;
    for (const auto& u : c->variables()) {   // resolve the variables once: flatten() then needs no UUID lookup per slot
      auto it = variables_.find(u);
      if (it == variables_.end()) throw std::logic_error("addConstraint: constraint " + c->type() + " uses a variable that is not in the graph");
      e.vars.push_back(it->second.get());
    }
    ensureConnectivity();
    for (const auto& u : c->variables()) by_variable_[u].insert(c->uuid());
    const fuse_core::UUID id = c->uuid();
    e.c = std::move(c);
    appendRow(e, id);
    constraints_.emplace(id, std::move(e));
    return true;
  }
  bool removeConstraint(const fuse_core::UUID& u) {
    auto it = constraints_.find(u);
    if (it == constraints_.end()) return false;
    ensureConnectivity();
    for (const auto& v : it->second.c->variables()) by_variable_[v].erase(u);
    removeRow(it->second);
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
  // Graph::clone() (fixed_lag_smoother.cpp:308 does this every cycle for the publishers): variables are deep-copied,
  // the immutable constraints are shared, the packed tables are copied with their variable pointers remapped
  UniquePtr clone() const {
    UniquePtr g(new GpuGraph(device_));
    std::vector<fuse_core::Variable*> copies;
    copies.reserve(variables_.size());
    auto hint = g->variables_.end();
    for (auto& kv : variables_) {   // same key order: hinted insertion; the flat index carries old -> new
      kv.second->flatIndex((int32_t)copies.size());
      hint = g->variables_.emplace_hint(hint, kv.first, kv.second->clone());
      copies.push_back(hint->second.get());
    }
    auto chint = g->constraints_.end();
    for (auto& kv : constraints_) {
      CEntry e;
      e.c = kv.second.c;   // constraints are immutable once in a graph: the copy shares them (variables are deep-copied)
      e.type = kv.second.type; e.row = kv.second.row;
      for (const auto* v : kv.second.vars) e.vars.push_back(copies[v->flatIndex()]);
      chint = g->constraints_.emplace_hint(chint, kv.first, std::move(e));
    }
    g->cameras_ = cameras_;
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      g->tables_[ty] = tables_[ty];
      for (auto& v : g->tables_[ty].vars) v = copies[v->flatIndex()];
    }
    g->marginal_rows_ = marginal_rows_;
    g->ordered_.reserve(ordered_.size());
    for (const auto* v : ordered_) g->ordered_.push_back(copies[v->flatIndex()]);   // same order in the copy
    g->connectivity_valid_ = false;   // variable -> constraints index of the copy: rebuilt on first use (publishers rarely need it)
    g->on_hold_ = on_hold_;
    return g;
  }
  size_t numVariables() const { return variables_.size(); }
  size_t numConstraints() const { return constraints_.size(); }
  void print(std::ostream& s) const {
    s << "GpuGraph\n  variables:\n";
    for (auto& kv : variables_) { s << "   - "; kv.second->print(s); s << "\n"; }
    s << "  constraints:\n";
    for (auto& kv : constraints_) { s << "   - "; kv.second.c->print(s); s << "\n"; }
  }

  // ---- the hot call -----------------------------------------------------------------------------------
  // Deterministic block order: keyframes ascending by stamp, (q,p,v,bg,ba) inside one
  // keyframe (ImuState::GetStateVector, imu_state.cpp:348-354); then landmarks ascending by id
  // (graph_access.cpp:200-216); unstamped extrinsic blocks last.
  // The order is kept incrementally: a variable is inserted at its sorted position when it enters the graph and erased
  // when it leaves (a sliding window adds and drops a few hundred variables per cycle out of tens of thousands).
  const std::vector<const fuse_core::Variable*>& orderedVariables() const { return ordered_; }

  // Flat IR of the current graph (variables at their current values) loaded into the back-end context
  struct Flat {
    std::vector<const fuse_core::Variable*> vars;
    std::vector<double> values;
    std::vector<int32_t> offset;
    std::vector<uint8_t> size, manifold, is_const;
  };
  int32_t blockIndexOf(const fuse_core::UUID& u) const { return variables_.at(u)->flatIndex(); }   // valid right after flatten()
  bool flatten(Flat& f) {
    const bool timing = std::getenv("BS_HOST_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[GpuGraph::flatten] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    f.vars = orderedVariables();
    lap("block order");
    for (const auto* v : f.vars) {
      v->flatIndex((int32_t)f.offset.size());
      f.offset.push_back((int32_t)f.values.size());
      f.size.push_back((uint8_t)v->size());
      f.manifold.push_back((uint8_t)v->manifold());
      f.is_const.push_back((v->holdConstant() || on_hold_.count(v->uuid())) ? 1 : 0);
      f.values.insert(f.values.end(), v->data(), v->data() + v->size());
    }
    if (f.offset.empty()) return false;
    lap("block table");
    // block indices of every packed row: the only per-cycle work on the factor side (constants, losses and camera ids
    // were packed once, when the constraint entered the graph)
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      TypeTable& tb = tables_[ty];
      if (!tb.rows) continue;
      const int nidx = tb.nvar + (tb.has_cam ? 1 : 0);
      tb.idx.resize((size_t)tb.rows * nidx);
      for (size_t r = 0; r < tb.rows; ++r) {
        for (int sl = 0; sl < tb.nvar; ++sl) tb.idx[r * nidx + sl] = tb.vars[r * tb.nvar + sl]->flatIndex();
        if (tb.has_cam) tb.idx[r * nidx + tb.nvar] = tb.cam[r];
      }
    }
    lap("block indices of factors");
    check(BS_API(clear)(ctx_));
    check(BS_API(set_blocks)(ctx_, (int32_t)f.offset.size(), f.values.data(), f.offset.data(), f.size.data(), f.manifold.data(), f.is_const.data()));
    if (!cameras_.empty()) check(BS_API(set_cameras)(ctx_, (int32_t)cameras_.size(), cameras_.data()));
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      const TypeTable& tb = tables_[ty];
      if (tb.rows) check(BS_API(add_factors)(ctx_, ty, (int32_t)tb.rows, tb.idx.data(), tb.consts.data(), tb.loss_kind.data(), tb.loss_a.data()));
    }
    for (const auto& kv : marginal_rows_) {
      const auto& m = kv.second;
      std::vector<int32_t> blocks;
      for (const auto* v : m.vars) blocks.push_back(v->flatIndex());
      check(BS_API(add_marginal)(ctx_, (int32_t)blocks.size(), blocks.data(), m.e.rows, m.e.A.data(), m.e.b.data(), m.e.xbar.data()));
    }
    lap("hand-over (C-ABI copies)");
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
      std::memcpy(const_cast<fuse_core::Variable*>(vars[i])->data(), values.data() + offset[i], size[i] * sizeof(double));
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
      ensureConnectivity();
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
      const int32_t b = blockIndexOf(u);
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
      if (!variableExists(rq.first) || !variableExists(rq.second)) throw std::out_of_range("getCovariance: variable not in graph");
      const int32_t a = blockIndexOf(rq.first), b = blockIndexOf(rq.second);
      const size_t ta = variables_.at(rq.first)->localSize(), tb = variables_.at(rq.second)->localSize();
      std::vector<double> m(ta * tb);
      check(BS_API(covariance)(ctx_, a, b, m.data()));
      covariance_matrices.push_back(std::move(m));
    }
  }

 private:
  void check(int rc) { if (rc != BSGPU_OK) throw std::runtime_error(std::string("bsgpu: ") + BS_API(last_error)(ctx_)); }
  int device_;
  bsgpu_ctx* ctx_;
  std::map<fuse_core::UUID, fuse_core::Variable::SharedPtr> variables_;
  // resolved variables of one constraint: inline up to 10 (the IMU factor), heap only for wide marginal priors
  struct VarList {
    const fuse_core::Variable* inl[10];
    std::vector<const fuse_core::Variable*> big;
    uint32_t n = 0;
    void push_back(const fuse_core::Variable* v) {
      if (n < 10) inl[n] = v;
      else { if (n == 10) big.assign(inl, inl + 10); big.push_back(v); }
      ++n;
    }
    size_t size() const { return n; }
    const fuse_core::Variable* const* data() const { return n <= 10 ? inl : big.data(); }
    const fuse_core::Variable* operator[](size_t i) const { return data()[i]; }
    const fuse_core::Variable* const* begin() const { return data(); }
    const fuse_core::Variable* const* end() const { return data() + n; }
  };
  struct CEntry { fuse_core::Constraint::SharedPtr c; VarList vars; int type = -1; size_t row = 0; };
  // Packed per-type factor tables, persisted across cycles: a constraint is packed ONCE, when it
  // enters the graph; removal swaps the last row into the hole.  Row order therefore follows the transaction history.
  struct TypeTable {
    size_t rows = 0;
    int nvar = 0;
    bool has_cam = false;
    std::vector<const fuse_core::Variable*> vars;   // rows x nvar
    std::vector<int32_t> cam;                       // rows (camera-table id) when has_cam
    std::vector<double> consts;
    std::vector<int32_t> loss_kind;
    std::vector<double> loss_a;
    std::vector<fuse_core::UUID> owner;             // rows: the constraint of each row
    std::vector<int32_t> idx;                       // scratch: rows x nidx, rebuilt by flatten()
  };
  struct MarginalRow { fuse_core::FactorTables::MarginalEntry e; std::vector<const fuse_core::Variable*> vars; };
  void appendRow(CEntry& e, const fuse_core::UUID& id) {
    fuse_core::FactorTables t1;
    int32_t slots[64];
    std::vector<int32_t> slots_big;
    int32_t* sl = slots;
    if (e.vars.size() > 64) { slots_big.resize(e.vars.size()); sl = slots_big.data(); }
    for (size_t i = 0; i < e.vars.size(); ++i) sl[i] = (int32_t)i;
    e.c->pack(fuse_core::BlockOf(sl), t1);
    if (!t1.marginals.empty()) {
      MarginalRow m; m.e = std::move(t1.marginals[0]); m.vars.assign(e.vars.begin(), e.vars.end());
      marginal_rows_[id] = std::move(m);
      e.type = -2;
      return;
    }
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      if (!t1.count(ty)) continue;
      TypeTable& tb = tables_[ty];
      const int nvar = (int)e.vars.size(), nidx = (int)t1.idx[ty].size();
      if (!tb.rows && tb.vars.empty()) { tb.nvar = nvar; tb.has_cam = nidx > nvar; }
      tb.vars.insert(tb.vars.end(), e.vars.begin(), e.vars.end());
      if (tb.has_cam) {
        const bsgpu_camera& c1 = t1.cameras.at(t1.idx[ty][nvar]);
        int32_t cid = -1;
        for (size_t i = 0; i < cameras_.size(); ++i) if (std::memcmp(&cameras_[i], &c1, sizeof(c1)) == 0) { cid = (int32_t)i; break; }
        if (cid < 0) { cameras_.push_back(c1); cid = (int32_t)cameras_.size() - 1; }
        tb.cam.push_back(cid);
      }
      tb.consts.insert(tb.consts.end(), t1.consts[ty].begin(), t1.consts[ty].end());
      tb.loss_kind.push_back(t1.loss_kind[ty][0]); tb.loss_a.push_back(t1.loss_a[ty][0]);
      tb.owner.push_back(id);
      e.type = ty; e.row = tb.rows++;
      return;
    }
  }
  void removeRow(const CEntry& e) {
    if (e.type == -2) { marginal_rows_.erase(e.c->uuid()); return; }
    if (e.type < 0) return;
    TypeTable& tb = tables_[e.type];
    const size_t last = tb.rows - 1, r = e.row, nc = tb.consts.size() / tb.rows;
    if (r != last) {
      for (int sl = 0; sl < tb.nvar; ++sl) tb.vars[r * tb.nvar + sl] = tb.vars[last * tb.nvar + sl];
      if (tb.has_cam) tb.cam[r] = tb.cam[last];
      for (size_t k = 0; k < nc; ++k) tb.consts[r * nc + k] = tb.consts[last * nc + k];
      tb.loss_kind[r] = tb.loss_kind[last]; tb.loss_a[r] = tb.loss_a[last];
      tb.owner[r] = tb.owner[last];
      constraints_.at(tb.owner[r]).row = r;
    }
    tb.vars.resize(last * tb.nvar);
    if (tb.has_cam) tb.cam.pop_back();
    tb.consts.resize(last * nc);
    tb.loss_kind.pop_back(); tb.loss_a.pop_back(); tb.owner.pop_back();
    tb.rows = last;
  }
  TypeTable tables_[BSGPU_F_NUM_TYPES];
  std::vector<bsgpu_camera> cameras_;
  std::map<fuse_core::UUID, MarginalRow> marginal_rows_;
  std::map<fuse_core::UUID, CEntry> constraints_;
  // (class: stamped / landmark / other, stamp | landmark id, state slot, uuid) — SURVEY.md §8a A17
  static bool orderBefore(const fuse_core::Variable* x, const fuse_core::Variable* y) {
    const int cx = x->isStamped() ? 0 : x->isLandmark() ? 1 : 2, cy = y->isStamped() ? 0 : y->isLandmark() ? 1 : 2;
    if (cx != cy) return cx < cy;
    if (cx == 0) {
      if (x->stamp() != y->stamp()) return x->stamp() < y->stamp();
      if (x->stateSlot() != y->stateSlot()) return x->stateSlot() < y->stateSlot();
    } else if (cx == 1 && x->landmarkId() != y->landmarkId()) return x->landmarkId() < y->landmarkId();
    return x->uuid() < y->uuid();
  }
  std::vector<const fuse_core::Variable*> ordered_;   // deterministic block order, maintained on insertion / removal
  void ensureConnectivity() const {
    if (connectivity_valid_) return;
    by_variable_.clear();
    for (const auto& kv : constraints_) for (const auto& u : kv.second.c->variables()) by_variable_[u].insert(kv.first);
    connectivity_valid_ = true;
  }
  mutable std::map<fuse_core::UUID, std::set<fuse_core::UUID>> by_variable_;
  mutable bool connectivity_valid_ = true;
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
