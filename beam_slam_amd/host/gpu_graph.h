// bs_optimizers::GpuGraph — the fuse_core::Graph the reference's optimizer would be handed instead of
// fuse_graphs::HashGraph (bs_optimizers/src/fixed_lag_smoother_node.cpp:42-45).  Graph bookkeeping on
// the host (UUID maps, like HashGraph [EXT]); optimize() flattens to the IR of include/bsgpu.h with the
// deterministic block order of SURVEY.md §8a A17 and runs the solve on the GPU through the C-ABI.
//
// The back-end is libbsgpu through the C-ABI of include/bsgpu.h, nothing else.  (The CPU-only unit tests of this host logic
// re-point the bsgpu_* names at the test oracle from OUTSIDE this header: tests/host/oracle_backend.h, force-included by
// tests/test_host_cpp.py — the product tree knows nothing about the oracle.)
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "bs_constraints.h"

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

// granularity of the copy-on-write sharing between a graph and its clones (small values only make sense in tests)
#ifndef BS_GRAPH_COW_BUCKETS
#define BS_GRAPH_COW_BUCKETS 16384
#endif
#ifndef BS_GRAPH_COW_CHUNK
#define BS_GRAPH_COW_CHUNK 1024
#endif

namespace bs_optimizers {

namespace detail {
// uuid -> slot index whose copy is O(buckets): every bucket is a small sorted vector behind a shared_ptr, a graph copy
// shares the buckets and whoever mutates one first copies that bucket only (copy-on-write).
class CowIndex {
 public:
  struct Item { fuse_core::UUID key; int32_t val; };
  int32_t find(const fuse_core::UUID& k) const {
    const auto& b = buckets_[bucketOf(k)];
    if (!b) return -1;
    auto it = std::lower_bound(b->begin(), b->end(), k, [](const Item& a, const fuse_core::UUID& x) { return a.key < x; });
    return (it != b->end() && it->key == k) ? it->val : -1;
  }
  void insert(const fuse_core::UUID& k, int32_t v) {
    auto& b = mut(bucketOf(k));
    b.insert(std::lower_bound(b.begin(), b.end(), k, [](const Item& a, const fuse_core::UUID& x) { return a.key < x; }), Item{k, v});
    ++size_;
  }
  // one traversal for "is it there? if not, put it in": false (nothing changed — no bucket copied) if the key is present
  bool insertIfAbsent(const fuse_core::UUID& k, int32_t v) {
    const size_t bi = bucketOf(k);
    const auto& b0 = buckets_[bi];
    size_t pos = 0;
    if (b0) {
      auto it = std::lower_bound(b0->begin(), b0->end(), k, [](const Item& a, const fuse_core::UUID& x) { return a.key < x; });
      if (it != b0->end() && it->key == k) return false;
      pos = (size_t)(it - b0->begin());
    }
    auto& b = mut(bi);
    b.insert(b.begin() + pos, Item{k, v});
    ++size_;
    return true;
  }
  void prefetch(const fuse_core::UUID& k) const { const auto& b = buckets_[bucketOf(k)]; if (b) __builtin_prefetch(b->data(), 0, 1); }
  void erase(const fuse_core::UUID& k) {
    auto& b = mut(bucketOf(k));
    auto it = std::lower_bound(b.begin(), b.end(), k, [](const Item& a, const fuse_core::UUID& x) { return a.key < x; });
    if (it != b.end() && it->key == k) { b.erase(it); --size_; }
  }
  size_t size() const { return size_; }
  void clear() { for (auto& b : buckets_) b.reset(); size_ = 0; }
 private:
  static constexpr size_t kBuckets = BS_GRAPH_COW_BUCKETS;
  static size_t bucketOf(const fuse_core::UUID& k) { return (size_t)((k.hi ^ (k.lo * 0x9e3779b97f4a7c15ull)) >> 17) % kBuckets; }
  std::vector<Item>& mut(size_t i) {
    auto& b = buckets_[i];
    if (!b) b = std::make_shared<std::vector<Item>>();
    else if (b.use_count() > 1) b = std::make_shared<std::vector<Item>>(*b);
    return *b;
  }
  std::array<std::shared_ptr<std::vector<Item>>, kBuckets> buckets_;
  size_t size_ = 0;
};
// uuid -> slot for the CONSTRAINTS: one flat open-addressing table, owned by one graph and never shared.  A clone() does not copy it — a
// snapshot handed to the publishers reads variables and walks constraints, it does not look constraints up by uuid — and builds its own
// from the constraint slots the first time it needs one.  (Behind copy-on-write buckets like the variables' index, every constraint added
// or removed with a snapshot alive copied a bucket: two allocations and ~600 bytes, 3 500 times per cycle at C2 — a third of Graph::update().)
class FlatIndex {
 public:
  int32_t find(const fuse_core::UUID& k) const {
    if (t_.empty()) return -1;
    for (size_t i = h(k) & mask_;; i = (i + 1) & mask_) {
      if (t_[i].val < 0) return -1;
      if (t_[i].key == k) return t_[i].val;
    }
  }
  bool insertIfAbsent(const fuse_core::UUID& k, int32_t v) {
    if ((n_ + 1) * 10 > t_.size() * 6) grow();
    for (size_t i = h(k) & mask_;; i = (i + 1) & mask_) {
      if (t_[i].val < 0) { t_[i].key = k; t_[i].val = v; ++n_; return true; }
      if (t_[i].key == k) return false;
    }
  }
  void erase(const fuse_core::UUID& k) {
    if (t_.empty()) return;
    size_t i = h(k) & mask_;
    for (;; i = (i + 1) & mask_) { if (t_[i].val < 0) return; if (t_[i].key == k) break; }
    // backward-shift deletion: no tombstones, look-ups stay short however long the window slides
    for (size_t j = (i + 1) & mask_;; j = (j + 1) & mask_) {
      if (t_[j].val < 0) break;
      const size_t home = h(t_[j].key) & mask_;
      if (((j - home) & mask_) >= ((j - i) & mask_)) { t_[i] = t_[j]; i = j; }
    }
    t_[i].val = -1;
    --n_;
  }
  void prefetch(const fuse_core::UUID& k) const { if (!t_.empty()) __builtin_prefetch(&t_[h(k) & mask_], 0, 1); }
  size_t size() const { return n_; }
  void clear() { t_.clear(); n_ = 0; mask_ = 0; }
  void reserve(size_t n) { size_t c = 1024; while (c * 6 < n * 10) c <<= 1; if (c > t_.size()) rehash(c); }
 private:
  struct Slot { fuse_core::UUID key; int32_t val = -1; };
  static size_t h(const fuse_core::UUID& k) { return (size_t)((k.hi ^ (k.lo * 0x9e3779b97f4a7c15ull)) >> 13); }
  void grow() { rehash(t_.empty() ? 1024 : t_.size() * 2); }
  void rehash(size_t cap) {
    std::vector<Slot> old;
    old.swap(t_);
    t_.assign(cap, Slot());
    mask_ = cap - 1; n_ = 0;
    for (const Slot& s : old) if (s.val >= 0) insertIfAbsent(s.key, s.val);
  }
  std::vector<Slot> t_;
  size_t n_ = 0, mask_ = 0;
};
// slot -> T in chunks behind shared_ptrs, copy-on-write per chunk (same idea as CowIndex)
template <class T>
class CowChunks {
 public:
  static constexpr size_t kChunk = BS_GRAPH_COW_CHUNK;
  size_t size() const { return size_; }
  const T& operator[](size_t i) const { return (*chunks_[i / kChunk])[i % kChunk]; }
  T& mut(size_t i) {
    auto& c = chunks_[i / kChunk];
    if (c.use_count() > 1) c = std::make_shared<std::array<T, kChunk>>(*c);
    return (*c)[i % kChunk];
  }
  void push_back(T v) {
    if (size_ % kChunk == 0 && size_ / kChunk == chunks_.size()) chunks_.push_back(std::make_shared<std::array<T, kChunk>>());
    mut(size_) = std::move(v);
    ++size_;
  }
  void clear() { chunks_.clear(); size_ = 0; }
 private:
  std::vector<std::shared_ptr<std::array<T, kChunk>>> chunks_;
  size_t size_ = 0;
};
}  // namespace detail

class GpuGraph {
 public:
  using UniquePtr = std::unique_ptr<GpuGraph>;
  static UniquePtr make_unique(int device = 0) { return UniquePtr(new GpuGraph(device)); }
  explicit GpuGraph(int device = 0) : device_(device), ctx_(bsgpu_create(device)) {
    if (!ctx_) throw std::runtime_error("GpuGraph: no usable back-end (libbsgpu has no CPU fallback)");
  }
 private:
  struct DeferContext {};
  GpuGraph(int device, DeferContext) : device_(device), ctx_(nullptr) {}   // clone(): a snapshot that is only read never opens a device context
 public:
  ~GpuGraph() {
    if (ctx_) bsgpu_destroy(ctx_);
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty)
      if (undo_[ty]) { std::lock_guard<std::mutex> lk(tables_[ty]->mu); undo_[ty]->detached = true; }
    // a snapshot taken from this graph may outlive it: what this graph owns goes to the generation the snapshots hold on to
    if (gen_ && gen_.use_count() > 1) for (auto& c : cown_) if (c) gen_->graveyard.push_back(std::move(c));
  }
  GpuGraph(const GpuGraph&) = delete;
  GpuGraph& operator=(const GpuGraph&) = delete;

  // ---- fuse_core::Graph surface used by the reference ---------------------------------
  void clear() {
    vslots_.clear(); slabs_.clear(); vfree_.clear(); vindex_.clear(); vmeta_.clear(); vdata_.clear(); ordered_.clear(); on_hold_.clear();
    for (auto& c : cown_) if (c) retire(std::move(c));
    cown_.clear();
    cptr_.clear(); ctype_.clear(); crow_.clear(); cfree_.clear(); cindex_.clear(); cindex_valid_ = true; n_constraints_ = 0;
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      if (undo_[ty]) { std::lock_guard<std::mutex> lk(tables_[ty]->mu); undo_[ty]->detached = true; }
      undo_[ty].reset(); tables_[ty].reset();
      dirty_[ty].clear(); synced_[ty] = false;
    }
    tables_own_ = true;
    cameras_.clear(); marginal_rows_.clear();
    conn_.clear(); connectivity_valid_ = true;
  }
  bool variableExists(const fuse_core::UUID& u) const { return vindex_.find(u) >= 0; }
  bool constraintExists(const fuse_core::UUID& u) const { ensureConstraintIndex(); return cindex_.find(u) >= 0; }
  const fuse_core::Variable& getVariable(const fuse_core::UUID& u) const {
    const int32_t s = vindex_.find(u);
    if (s < 0) throw std::out_of_range("variable not in graph");
    return *vslots_[s];
  }
  fuse_core::Variable& getVariable(const fuse_core::UUID& u) { return const_cast<fuse_core::Variable&>(static_cast<const GpuGraph*>(this)->getVariable(u)); }
  std::vector<const fuse_core::Variable*> getVariables() const {
    std::vector<const fuse_core::Variable*> v;
    v.reserve(vindex_.size());
    for (const auto& p : vslots_) if (p) v.push_back(p.get());
    return v;
  }
  std::vector<const fuse_core::Constraint*> getConstraints() const {
    std::vector<const fuse_core::Constraint*> v;
    v.reserve(n_constraints_);
    for (size_t i = 0; i < cptr_.size(); ++i) if (ctype_[i] != kFree) v.push_back(cptr_[i]);
    return v;
  }
  std::vector<const fuse_core::Constraint*> getConnectedConstraints(const fuse_core::UUID& var) const {
    const int32_t s = vindex_.find(var);
    if (s < 0) throw std::logic_error("getConnectedConstraints: variable not in graph");
    ensureConnectivity();
    std::vector<const fuse_core::Constraint*> out;
    for (int32_t cs : conn_[s]) out.push_back(cptr_[cs]);
    return out;
  }
  bool addVariable(fuse_core::Variable::SharedPtr v) {
    const int32_t have = vindex_.find(v->uuid());
    if (have >= 0) { std::memcpy(vslots_[have]->data(), v->data(), v->size() * sizeof(double)); return false; }  // HashGraph: overwrite value
    int32_t s;
    if (!vfree_.empty()) { s = vfree_.back(); vfree_.pop_back(); }
    else { s = (int32_t)vslots_.size(); vslots_.emplace_back(); vmeta_.emplace_back(); vdata_.emplace_back(); if (connectivity_valid_) conn_.emplace_back(); }
    vmeta_[s] = VMeta{(uint8_t)v->size(), (uint8_t)v->manifold(), (uint8_t)(v->holdConstant() ? 1 : 0), (uint16_t)std::min<size_t>(v->cloneSize(), 65535)};
    vdata_[s] = v->data();
    vindex_.insert(v->uuid(), s);
    vslots_[s] = std::move(v);
    ordered_.insert(std::upper_bound(ordered_.begin(), ordered_.end(), s, [this](int32_t a, int32_t b) { return orderBefore(vslots_[a].get(), vslots_[b].get()); }), s);
    return true;
  }
  bool removeVariable(const fuse_core::UUID& u) {
    const int32_t s = vindex_.find(u);
    if (s < 0) return false;
    ensureConnectivity();
    if (!conn_[s].empty()) throw std::logic_error("removeVariable: variable still used by a constraint");
    {
      auto pos = std::lower_bound(ordered_.begin(), ordered_.end(), s, [this](int32_t a, int32_t b) { return orderBefore(vslots_[a].get(), vslots_[b].get()); });
      while (pos != ordered_.end() && *pos != s) ++pos;   // (equal keys cannot occur: the uuid breaks every tie)
      if (pos != ordered_.end()) ordered_.erase(pos);
    }
    on_hold_.erase(u); vindex_.erase(u);
    vslots_[s].reset();
    vdata_[s] = nullptr;
    vfree_.push_back(s);
    return true;
  }
  bool addConstraint(fuse_core::Constraint::SharedPtr c) {
    // (the slot is chosen first so that the uuid index is walked ONCE — "present?" and the insertion are one look-up; a new constraint's
    //  bucket is cold, and with a snapshot alive the insertion copies it)
    ensureConstraintIndex();
    const bool reuse = !cfree_.empty();
    const int32_t cs_next = reuse ? cfree_.back() : (int32_t)cptr_.size();
    if (!cindex_.insertIfAbsent(c->uuid(), cs_next)) return false;
    ++n_constraints_;
    std::vector<int32_t>& vars = vars_scratch_;
    vars.clear();
    vars.reserve(c->variables().size());
    for (const auto& u : c->variables()) {   // resolve the variables once: flatten() then needs no UUID lookup per slot
      const int32_t s = vindex_.find(u);
      if (s < 0) { cindex_.erase(c->uuid()); --n_constraints_; throw std::logic_error("addConstraint: constraint " + c->type() + " uses a variable that is not in the graph"); }
      vars.push_back(s);
    }
    const int32_t cs = cs_next;
    if (reuse) cfree_.pop_back();
    else { cptr_.push_back(nullptr); ctype_.push_back(kFree); crow_.push_back(0); }
    if (cown_.size() < cptr_.size()) cown_.resize(cptr_.size());
    ensureConnectivity();
    // (a constraint is new to every list; only a variable it names twice must not get it twice — checked within `vars`, not by
    //  scanning the variable's list: a keyframe pose carries thousands of constraints)
    for (size_t i = 0; i < vars.size(); ++i)
      if (std::find(vars.begin(), vars.begin() + i, vars[i]) == vars.begin() + i) conn_[vars[i]].push_back(cs);
    appendRow(*c, cs, vars);
    cptr_.mut(cs) = c.get();
    cown_[cs] = std::move(c);
    return true;
  }
  bool removeConstraint(const fuse_core::UUID& u) {
    ensureConstraintIndex();
    const int32_t cs = cindex_.find(u);
    if (cs < 0) return false;
    ensureConnectivity();
    forEachVariableOf(cs, [&](int32_t s) { auto& l = conn_[s]; auto it = std::find(l.begin(), l.end(), cs); if (it != l.end()) { *it = l.back(); l.pop_back(); } });
    removeRow(cs);
    cindex_.erase(u);
    --n_constraints_;
    cptr_.mut(cs) = nullptr;
    if ((size_t)cs < cown_.size() && cown_[cs]) retire(std::move(cown_[cs]));
    ctype_[cs] = kFree;
    cfree_.push_back(cs);
    return true;
  }
  void holdVariable(const fuse_core::UUID& u, bool hold = true) { if (hold) on_hold_.insert(u); else on_hold_.erase(u); }
  // Graph::update(transaction): removals first, then additions ([EXT] fuse_core::Graph::update)
  void update(const fuse_core::Transaction& t) {
    const bool timing = std::getenv("BS_HOST_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[GpuGraph::update] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    // Removals as a batch: a sliding window drops its oldest key frame's ~1 500 constraints at once, and a key frame's pose blocks carry
    // thousands — one std::find per constraint in those lists was most of Graph::update().  The constraints are marked, and every variable
    // that loses some filters its list ONCE.
    {
      std::vector<int32_t> gone, touched;
      gone.reserve(t.removedConstraints().size());
      ensureConstraintIndex();
      for (const auto& u : t.removedConstraints()) { const int32_t cs = cindex_.find(u); if (cs >= 0) gone.push_back(cs); }
      if (!gone.empty()) {
        ensureConnectivity();
        cmark_.assign(cptr_.size(), 0);
        if (vmark_.size() < vslots_.size()) vmark_.resize(vslots_.size(), 0);
        for (int32_t cs : gone) {
          if (cmark_[cs]) continue;   // (named twice in the transaction)
          cmark_[cs] = 1;
          forEachVariableOf(cs, [&](int32_t s) { if (!vmark_[s]) { vmark_[s] = 1; touched.push_back(s); } });
        }
        for (int32_t s : touched) {
          auto& l = conn_[s];
          l.erase(std::remove_if(l.begin(), l.end(), [&](int32_t cs) { return cmark_[cs] != 0; }), l.end());
          vmark_[s] = 0;
        }
        for (int32_t cs : gone) {
          if (!cmark_[cs]) continue;
          cmark_[cs] = 0;
          removeRow(cs);
          cindex_.erase(cptr_[cs]->uuid());
          --n_constraints_;
          cptr_.mut(cs) = nullptr;
          if ((size_t)cs < cown_.size() && cown_[cs]) retire(std::move(cown_[cs]));
          ctype_[cs] = kFree;
          cfree_.push_back(cs);
        }
      }
    }
    lap("remove constraints");
    for (const auto& u : t.removedVariables()) removeVariable(u);
    lap("remove variables");
    for (const auto& v : t.addedVariables()) addVariable(v->clone());
    lap("add variables");
    {
      // (the uuid-index buckets of the constraints to come are requested ahead: each is a cold line behind two pointers)
      const auto& add = t.addedConstraints();
      constexpr size_t kAhead = 8;
      for (size_t i = 0; i < add.size(); ++i) {
        if (i + kAhead < add.size()) cindex_.prefetch(add[i + kAhead]->uuid());
        addConstraint(add[i]->clone());
      }
    }
    lap("add constraints");
  }
  // Graph::clone() (fixed_lag_smoother.cpp:308 does this every cycle for the publishers).  Variables are deep-copied (the
  // solve writes them).  Everything on the constraint side is immutable between transactions and is SHARED with the copy
  // behind copy-on-write handles — the constraints themselves, the uuid indices, the packed per-type tables — so the
  // copy costs O(variables), and the first transaction applied afterwards pays one flat copy of what it touches.  This
  // is possible because the packed tables name variables by graph-local slot, and a copy keeps the slot numbering.
  UniquePtr clone() const {
    const bool timing = std::getenv("BS_HOST_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[GpuGraph::clone] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    UniquePtr g(new GpuGraph(device_, DeferContext{}));
    g->vslots_.resize(vslots_.size());
    g->vdata_.assign(vslots_.size(), nullptr);
    {
      // The copies are constructed side by side in ONE allocation that they own together (aliasing shared_ptrs): 51 000 make_shared
      // calls — and as many frees when the snapshot is dropped — become one of each.  Types without cloneAt() are cloned one by one.
      // Offsets from the slot table alone (the sizes were noted when the variables entered); the source objects are separate heap
      // allocations and the copy pass is bound by the misses on them.
      const size_t n = vslots_.size();
      std::vector<uint32_t> off(n + 1, 0);
      size_t live = 0;
      for (size_t i = 0; i < n; ++i) {
        const size_t sz = (vslots_[i] && vmeta_[i].clone_size && vmeta_[i].clone_size < 65535) ? vmeta_[i].clone_size : 0;
        off[i + 1] = off[i] + (uint32_t)((sz + 15) & ~(size_t)15);
        live += vslots_[i] ? 1 : 0;
      }
      auto slab = std::make_shared<VariableSlab>(off[n]);
      slab->objects.assign(n, nullptr);
      unsigned char* const mem = slab->mem.get();
      GpuGraph* const gp = g.get();
      auto copy_range = [&, gp](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
          if (i + 16 < hi) __builtin_prefetch(vslots_[i + 16].get(), 0, 1);
          if (!vslots_[i]) continue;
          fuse_core::Variable* o = off[i + 1] > off[i] ? vslots_[i]->cloneAt(mem + off[i]) : nullptr;
          if (o) { slab->objects[i] = o; if (!o->destroyIsNoop()) slab->needs_destroy = true; }
          else gp->vslots_[i] = vslots_[i]->clone();
        }
      };
      (void)live;
      copy_range(0, n);   // (four threads were tried: 1.7 ms against 1.3 — the misses overlap well enough under one thread's prefetches)
      // The copy's handles to the slab's objects own NOTHING (aliasing shared_ptrs with an empty owner: no control block, no atomic
      // count — 51 000 increments here and as many decrements when the snapshot is dropped were a third of clone() + release); the slab
      // itself is kept by the graph (slabs_), so the objects live exactly as long as the graph that hands out references to them.
      for (size_t i = 0; i < n; ++i) {
        if (slab->objects[i]) g->vslots_[i] = fuse_core::Variable::SharedPtr(std::shared_ptr<void>(), slab->objects[i]);
        if (g->vslots_[i]) g->vdata_[i] = g->vslots_[i]->data();
      }
      g->slabs_.push_back(std::move(slab));
    }
    g->vfree_ = vfree_; g->vindex_ = vindex_; g->vmeta_ = vmeta_; g->ordered_ = ordered_; g->on_hold_ = on_hold_;
    lap("variables");
    // the constraints are shared as RAW pointers (a chunk is copied with memcpy when either side writes to it — behind shared_ptrs
    // every such copy was 1 024 atomic increments, 400 000 per cycle at C2, and as many decrements when the snapshot was dropped);
    // what keeps them alive for the snapshot is the GENERATION it holds (see Generation below)
    g->cptr_ = cptr_; g->ctype_ = ctype_; g->crow_ = crow_; g->cfree_ = cfree_;
    g->cindex_valid_ = false; g->n_constraints_ = n_constraints_;   // (uuid -> constraint slot of the copy: built on first use, like its connectivity)
    g->keep_ = keep_;                 // (what this graph itself inherited)
    g->keep_.push_back(gen_);         // everything this graph owns now or retires from now on
    {
      auto next = std::make_shared<Generation>();
      gen_->next = next;
      gen_ = std::move(next);
    }
    materialiseTables();              // (a snapshot that is cloned needs its own tables first)
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      if (!tables_[ty] || !tables_[ty]->rows) continue;
      auto u = std::make_shared<TableUndo>();
      u->rows = tables_[ty]->rows;
      u->saved.assign(u->rows, 0);
      {
        std::lock_guard<std::mutex> lk(tables_[ty]->mu);
        tables_[ty]->pruneWatchers();
        tables_[ty]->watchers.push_back(u);
      }
      g->tables_[ty] = tables_[ty];
      g->undo_[ty] = std::move(u);
      g->tables_own_.store(false, std::memory_order_relaxed);   // (published with the clone itself)
    }
    g->cameras_ = cameras_;
    g->marginal_rows_ = marginal_rows_;
    g->connectivity_valid_ = false;   // variable -> constraints index of the copy: rebuilt on first use (publishers rarely need it)
    lap("constraints (shared)");
    return g;
  }
  size_t numVariables() const { return vindex_.size(); }
  size_t numConstraints() const { return n_constraints_; }
  void print(std::ostream& s) const {
    s << "GpuGraph\n  variables:\n";
    for (const auto* v : getVariables()) { s << "   - "; v->print(s); s << "\n"; }
    s << "  constraints:\n";
    for (const auto* c : getConstraints()) { s << "   - "; c->print(s); s << "\n"; }
  }

  // ---- the hot call -----------------------------------------------------------------------------------
  // Deterministic block order: keyframes ascending by stamp, (q,p,v,bg,ba) inside one
  // keyframe (ImuState::GetStateVector, imu_state.cpp:348-354); then landmarks ascending by id
  // (graph_access.cpp:200-216); unstamped extrinsic blocks last.
  // The order is kept incrementally: a variable is inserted at its sorted position when it enters the graph and erased
  // when it leaves (a sliding window adds and drops a few hundred variables per cycle out of tens of thousands).
  std::vector<const fuse_core::Variable*> orderedVariables() const {
    std::vector<const fuse_core::Variable*> v;
    v.reserve(ordered_.size());
    for (int32_t s : ordered_) v.push_back(vslots_[s].get());
    return v;
  }

  // Flat IR of the current graph (variables at their current values) loaded into the back-end context
  struct Flat {
    std::vector<double> values;
    std::vector<int32_t> offset;
    std::vector<uint8_t> size, manifold, is_const;
    std::vector<int32_t> slot_to_block;   // graph-local variable slot -> block index (-1: free slot)
  };
  // block index of a variable in the table of the last flatten() (valid until the next graph mutation)
  int32_t blockIndexOf(const Flat& f, const fuse_core::UUID& u) const {
    const int32_t s = vindex_.find(u);
    if (s < 0) throw std::out_of_range("variable not in graph");
    return f.slot_to_block[s];
  }
  bool flatten(Flat& f) {
    const bool timing = std::getenv("BS_HOST_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[GpuGraph::flatten] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    const size_t nb = ordered_.size();
    if (!nb) return false;
    f.offset.resize(nb); f.size.resize(nb); f.manifold.resize(nb); f.is_const.resize(nb);
    f.slot_to_block.assign(vslots_.size(), -1);
    size_t nval = 0;
    for (size_t i = 0; i < nb; ++i) { f.offset[i] = (int32_t)nval; nval += vmeta_[ordered_[i]].size; }
    f.values.resize(nval);
    // The values live inside 51 000 separately allocated Variable objects: the loop is a chain of cache misses unless the
    // objects are requested ahead.  Nothing is written into them (the block index of a variable is f.slot_to_block).
    const int32_t* ord = ordered_.data();
    constexpr size_t kAhead = 24;
    for (size_t i = 0; i < nb; ++i) {
      if (i + kAhead < nb) __builtin_prefetch(vdata_[ord[i + kAhead]], 0, 0);
      const int32_t s = ord[i];
      const VMeta m = vmeta_[s];
      f.slot_to_block[s] = (int32_t)i;
      f.size[i] = m.size; f.manifold[i] = m.manifold; f.is_const[i] = m.hold_constant;
      const double* src = vdata_[s];
      double* dst = &f.values[f.offset[i]];
      for (int k = 0; k < m.size; ++k) dst[k] = src[k];
    }
    for (const auto& u : on_hold_) { const int32_t s = vindex_.find(u); if (s >= 0) f.is_const[f.slot_to_block[s]] = 1; }
    lap("block table");
    // The packed rows name their variables by slot and were written once, when the constraint entered the graph.  The back-end
    // keeps its own slot-named copy of the big tables across cycles (bsgpu_sync_factors_indirect) and is told which rows this
    // graph has written since the last hand-over; slot -> block is translated on its side.
    check(bsgpu_clear(ctx()));
    check(bsgpu_set_blocks(ctx(), (int32_t)nb, f.values.data(), f.offset.data(), f.size.data(), f.manifold.data(), f.is_const.data()));
    if (!cameras_.empty()) check(bsgpu_set_cameras(ctx(), (int32_t)cameras_.size(), cameras_.data()));
    materialiseTables();
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      const TypeTable* tb = tables_[ty].get();
      if (!tb || !tb->rows) continue;   // (nothing to say: the change list keeps growing until the next call for the type)
      std::vector<int32_t>& d = dirty_[ty];
      d.erase(std::remove_if(d.begin(), d.end(), [&](int32_t r) { return (size_t)r >= tb->rows; }), d.end());   // rows that have left since
      check(bsgpu_sync_factors_indirect(ctx(), ty, (int32_t)tb->rows, tb->idx.data(), (int32_t)f.slot_to_block.size(), f.slot_to_block.data(),
                                          tb->consts.data(), tb->loss_kind.data(), tb->loss_a.data(), synced_[ty] ? (int32_t)d.size() : -1, d.data()));
      synced_[ty] = true;
      d.clear();
    }
    for (const auto& kv : marginal_rows_) {
      const auto& m = kv.second;
      std::vector<int32_t> blocks;
      for (int32_t s : m.vars) blocks.push_back(f.slot_to_block[s]);
      check(bsgpu_add_marginal(ctx(), (int32_t)blocks.size(), blocks.data(), m.e.rows, m.e.A.data(), m.e.b.data(), m.e.xbar.data()));
    }
    lap("hand-over (C-ABI copies)");
    return true;
  }

  ceres_compat::SolverSummary optimize(const ceres_compat::SolverOptions& o = ceres_compat::SolverOptions()) {
    const auto t0 = std::chrono::steady_clock::now();
    Flat& f = flat_;   // (kept between cycles: no 2 MB of fresh pages per call)
    ceres_compat::SolverSummary s;
    if (!flatten(f)) { s.termination_type = ceres_compat::CONVERGENCE; s.message = "empty graph"; return s; }
    auto& values = f.values; auto& offset = f.offset; auto& size = f.size;
    bsgpu_options bo;
    bsgpu_options_default(&bo);
    bo.max_num_iterations = o.max_num_iterations; bo.max_solver_time_in_seconds = o.max_solver_time_in_seconds;
    bo.function_tolerance = o.function_tolerance; bo.gradient_tolerance = o.gradient_tolerance; bo.parameter_tolerance = o.parameter_tolerance;
    bsgpu_summary bs;
    std::memset(&bs, 0, sizeof(bs));
    const int rc = bsgpu_solve(ctx(), &bo, &bs);
    if (rc != BSGPU_OK) {
      // a back-end error is what Ceres reports as a FAILURE summary (the caller — FixedLagSmoother::optimizationLoop, fixed_lag_smoother.cpp:281-292 —
      // looks at the summary, it does not catch); the variables keep their values, as with !IsSolutionUsable()
      s.termination_type = ceres_compat::FAILURE;
      s.message = std::string("back-end error: ") + bsgpu_last_error(ctx());
      s.total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      last_summary_ = bs;
      return s;
    }
    if (bs.is_solution_usable) {   // [EXT] ceres::Solver: parameter blocks are written back only if IsSolutionUsable()
      check(bsgpu_get_blocks(ctx(), values.data(), (int64_t)values.size()));
      const int32_t* ord = ordered_.data();
      const size_t nb = ordered_.size();
      constexpr size_t kAhead = 24;
      for (size_t i = 0; i < nb; ++i) {   // Variable::data() updated in place, like Ceres does through the raw pointers
        if (i + kAhead < nb) __builtin_prefetch(vdata_[ord[i + kAhead]], 1, 0);
        if (f.is_const[i]) continue;       // (a constant block comes back as it went in)
        double* dst = vdata_[ord[i]];
        const double* src = values.data() + offset[i];
        for (int k = 0; k < size[i]; ++k) dst[k] = src[k];
      }
    }
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
    std::set<int32_t> removed_constraints;
    ensureConnectivity();
    for (const auto& u : to_marginalize) {
      const int32_t s = vindex_.find(u);
      if (s < 0) throw std::out_of_range("marginalizeVariables: variable not in graph");
      if (!conn_[s].empty()) {
        constrained.push_back(u);
        for (int32_t cs : conn_[s]) if (removed_constraints.insert(cs).second) tr.removeConstraint(cptr_[cs]->uuid());
      }
      tr.removeVariable(u);
    }
    if (constrained.empty()) return tr;
    Flat& f = flat_;
    if (!flatten(f)) return tr;
    std::vector<int32_t> marg;
    for (const auto& u : constrained) {
      const int32_t b = blockIndexOf(f, u);
      if (f.is_const[b]) throw std::logic_error("marginalizeVariables: variable is held constant");
      marg.push_back(b);
    }
    int32_t n_kept = 0, n_rows = 0, n_cols = 0;
    check(bsgpu_marginalize(ctx(), (int32_t)marg.size(), marg.data(), &n_kept, &n_rows, &n_cols));
    if (n_rows == 0) return tr;
    std::vector<int32_t> kept(n_kept);
    check(bsgpu_get_marginal(ctx(), kept.data(), nullptr, nullptr, nullptr));
    size_t amb = 0;
    for (int32_t b : kept) amb += f.size[b];
    std::vector<double> A((size_t)n_rows * n_cols), bvec(n_rows), xbar(amb);
    check(bsgpu_get_marginal(ctx(), kept.data(), A.data(), bvec.data(), xbar.data()));
    std::vector<fuse_core::UUID> kept_ids;
    for (int32_t b : kept) kept_ids.push_back(vslots_[ordered_[b]]->uuid());
    tr.addConstraint(std::make_shared<fuse_constraints::MarginalConstraint>(source, std::move(kept_ids), n_rows, n_cols, std::move(A), std::move(bvec), std::move(xbar)));
    return tr;
  }

  // fuse_core::Graph::getCovariance(covariance_requests, covariance_matrices) in tangent space (the form
  // bs_publishers/src/odometry_3d_publisher.cpp:82 consumes): one row-major localSize(a) x localSize(b) matrix per
  // requested pair, marginal covariance at the variables' current values.  Pose-side variables only: landmarks are
  // eliminated by the solver (throws, like fuse does for an uncomputable request).
  void getCovariance(const std::vector<std::pair<fuse_core::UUID, fuse_core::UUID>>& covariance_requests,
                     std::vector<std::vector<double>>& covariance_matrices) {
    Flat& f = flat_;
    covariance_matrices.clear();
    if (covariance_requests.empty()) return;
    if (!flatten(f)) throw std::runtime_error("getCovariance: empty graph");
    for (const auto& rq : covariance_requests) {
      if (!variableExists(rq.first) || !variableExists(rq.second)) throw std::out_of_range("getCovariance: variable not in graph");
      const int32_t a = blockIndexOf(f, rq.first), b = blockIndexOf(f, rq.second);
      const size_t ta = getVariable(rq.first).localSize(), tb = getVariable(rq.second).localSize();
      std::vector<double> m(ta * tb);
      check(bsgpu_covariance(ctx(), a, b, m.data()));
      covariance_matrices.push_back(std::move(m));
    }
  }

 private:
  void check(int rc) { if (rc != BSGPU_OK) throw std::runtime_error(std::string("bsgpu: ") + bsgpu_last_error(ctx_)); }
  int device_;
  bsgpu_ctx* ctx_;
  bsgpu_ctx* ctx() {
    if (!ctx_ && !(ctx_ = bsgpu_create(device_))) throw std::runtime_error("GpuGraph: no usable back-end (libbsgpu has no CPU fallback)");
    return ctx_;
  }

  // ---- variables: graph-local slots (stable while the variable is in the graph, kept by clone()) ---------------------
  struct VariableSlab {   // the variables of a clone(): one block, released (destructors first) with the last variable that lives in it
    explicit VariableSlab(size_t bytes) : mem(static_cast<unsigned char*>(::operator new(bytes ? bytes : 1, std::align_val_t(16)))) {}
    ~VariableSlab() { if (needs_destroy) for (fuse_core::Variable* o : objects) if (o) o->~Variable(); }
    bool needs_destroy = false;   // some object's destructor does something (Variable::destroyIsNoop)
    struct Free { void operator()(unsigned char* p) const { ::operator delete(p, std::align_val_t(16)); } };
    std::unique_ptr<unsigned char, Free> mem;
    std::vector<fuse_core::Variable*> objects;
    VariableSlab(const VariableSlab&) = delete;
    VariableSlab& operator=(const VariableSlab&) = delete;
  };
  struct VMeta { uint8_t size, manifold, hold_constant; uint16_t clone_size; };   // clone_size: Variable::cloneSize() (0: clone() only)
  std::vector<fuse_core::Variable::SharedPtr> vslots_;   // slot -> variable (null: free slot); a clone()'s initial ones are non-owning handles into slabs_
  std::vector<std::shared_ptr<VariableSlab>> slabs_;     // what a clone()'s variables live in (declared after vslots_: destroyed before... the handles own nothing)
  std::vector<VMeta> vmeta_;                              // slot -> what flatten() needs without touching the object
  std::vector<double*> vdata_;                            // slot -> Variable::data() (stable while the variable is in the graph)
  std::vector<int32_t> vfree_;
  detail::CowIndex vindex_;                               // uuid -> slot
  std::vector<int32_t> ordered_;                          // slots in the deterministic block order, maintained on insertion / removal
  std::set<fuse_core::UUID> on_hold_;
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

  // ---- constraints: slots + packed per-type tables, all shareable with a clone ---------------------------------------
  // Packed per-type factor tables, persisted across cycles: a constraint is packed ONCE, when it
  // enters the graph; removal swaps the last row into the hole.  Row order therefore follows the transaction history.
  // idx holds, per row, the variable SLOTS (then the camera-table id for the camera types) in the column layout of
  // bsgpu_add_factors, so the table is handed to the back-end as it is.
  // A snapshot (clone) does NOT get a copy of a table and the graph that goes on writing it does not copy it either (the 22 MB
  // reprojection table of C2 was copied once per cycle, by whichever side wrote first): both name the SAME object and the snapshot
  // registers an UNDO LOG with it.  Before the owner overwrites or truncates a row that existed when the snapshot was taken it saves the
  // row's content there (first write wins); a snapshot that ever needs its tables — it is optimised, mutated, cloned or asked for its
  // connectivity — rebuilds its version from the shared arrays + its log (materialiseTables) and owns a private table from then on.
  // Writes by the owner, and the rebuild, take the table's mutex while a log is registered (the snapshot lives on another thread).
  struct TableUndo {
    size_t rows = 0;                   // rows of the table when the snapshot was taken
    std::vector<uint8_t> saved;        // per such row: already in the log
    std::vector<uint32_t> row;         // the log: row numbers and their contents at that time
    std::vector<int32_t> idx, loss_kind, owner;
    std::vector<double> consts, loss_a;
    bool detached = false;             // the snapshot has its own table now, or is gone
  };
  struct TypeTable {
    size_t rows = 0;
    int nvar = 0, nidx = 0;
    std::vector<int32_t> idx;          // rows x nidx
    std::vector<double> consts;
    std::vector<int32_t> loss_kind;
    std::vector<double> loss_a;
    std::vector<int32_t> owner;        // rows: constraint slot of each row
    std::mutex mu;
    std::vector<std::shared_ptr<TableUndo>> watchers;
    // (owner side, mu held) row q is about to be overwritten or dropped
    void save(size_t q) {
      const size_t nc = rows ? consts.size() / rows : 0, ni = (size_t)nidx;
      for (auto& w : watchers) {
        if (w->detached || q >= w->rows || w->saved[q]) continue;
        w->saved[q] = 1; w->row.push_back((uint32_t)q);
        w->idx.insert(w->idx.end(), idx.begin() + q * ni, idx.begin() + (q + 1) * ni);
        w->consts.insert(w->consts.end(), consts.begin() + q * nc, consts.begin() + (q + 1) * nc);
        w->loss_kind.push_back(loss_kind[q]); w->loss_a.push_back(loss_a[q]); w->owner.push_back(owner[q]);
      }
    }
    void pruneWatchers() { watchers.erase(std::remove_if(watchers.begin(), watchers.end(), [](const std::shared_ptr<TableUndo>& w) { return w->detached; }), watchers.end()); }
  };
  // tables given back by the last owner that dropped them, capacity kept (at most a handful; thread-safe: snapshots die on other threads)
  class TablePool {
   public:
    static TablePool& instance() { static TablePool p; return p; }
    std::shared_ptr<TypeTable> acquire(int ty) {   // (per factor type: the 22 MB reprojection table must not go to the prior's)
      TypeTable* t = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu_);
        auto& f = free_[ty];
        if (!f.empty()) { t = f.back().release(); f.pop_back(); }
      }
      if (!t) t = new TypeTable();
      return std::shared_ptr<TypeTable>(t, [ty](TypeTable* p) { TablePool::instance().give_back(ty, p); });
    }
   private:
    void give_back(int ty, TypeTable* p) {
      std::unique_ptr<TypeTable> u(p);
      u->rows = 0; u->idx.clear(); u->consts.clear(); u->loss_kind.clear(); u->loss_a.clear(); u->owner.clear(); u->watchers.clear();
      std::lock_guard<std::mutex> lk(mu_);
      if (free_[ty].size() < 2) free_[ty].push_back(std::move(u));
    }
    std::mutex mu_;
    std::vector<std::unique_ptr<TypeTable>> free_[BSGPU_F_NUM_TYPES];
  };
  static constexpr int32_t kFree = -3, kMarginal = -2, kUnpacked = -1;
  struct MarginalRow { fuse_core::FactorTables::MarginalEntry e; std::vector<int32_t> vars; };
  detail::CowChunks<const fuse_core::Constraint*> cptr_;        // constraint slot -> constraint (raw: see cown_ / Generation)
  // Ownership.  cown_[slot] owns the constraints THIS graph added; a snapshot (clone) owns nothing it inherited and instead holds the
  // source's generation at the time of the copy.  A constraint the source removes later is retired into the source's CURRENT
  // generation, and a generation keeps its successor alive — so everything a snapshot can still point to lives as long as the
  // snapshot does, a constraint retired with no snapshot alive is released at once, and a source that dies first leaves what it
  // owns to its last generation.
  struct Generation {
    std::vector<fuse_core::Constraint::SharedPtr> graveyard;
    std::shared_ptr<Generation> next;
    ~Generation() {   // (unlink iteratively: a long-lived snapshot can sit at the head of thousands of generations)
      std::shared_ptr<Generation> n = std::move(next);
      while (n && n.use_count() == 1) { std::shared_ptr<Generation> nn = std::move(n->next); n = std::move(nn); }
    }
  };
  std::vector<fuse_core::Constraint::SharedPtr> cown_;
  mutable std::shared_ptr<Generation> gen_ = std::make_shared<Generation>();
  std::vector<std::shared_ptr<Generation>> keep_;
  void retire(fuse_core::Constraint::SharedPtr c) { if (gen_.use_count() > 1) gen_->graveyard.push_back(std::move(c)); }
  std::vector<int32_t> ctype_;                                  // constraint slot -> factor type | kFree | kMarginal | kUnpacked
  std::vector<uint32_t> crow_;                                  // constraint slot -> row in its type's table
  std::vector<int32_t> cfree_;
  mutable detail::FlatIndex cindex_;                            // uuid -> constraint slot (a clone's: built on first use)
  mutable std::atomic<bool> cindex_valid_{true};
  // A snapshot goes to several publisher threads as a const graph: the lazy builds behind const accessors (uuid index, connectivity,
  // own tables) are each done once under this mutex; the flags are read with acquire so that a reader that sees `true` sees the data.
  mutable std::recursive_mutex lazy_mu_;   // (recursive: the connectivity build reads the tables, which may materialise them)
  mutable std::atomic<bool> tables_own_{true};   // false while some undo_[ty] is set (a clone that still reads another graph's tables)
  size_t n_constraints_ = 0;
  void ensureConstraintIndex() const {
    if (cindex_valid_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::recursive_mutex> lk(lazy_mu_);
    if (cindex_valid_.load(std::memory_order_relaxed)) return;
    cindex_.clear();
    cindex_.reserve(n_constraints_);
    for (size_t cs = 0; cs < ctype_.size(); ++cs) if (ctype_[cs] != kFree) cindex_.insertIfAbsent(cptr_[cs]->uuid(), (int32_t)cs);
    cindex_valid_.store(true, std::memory_order_release);
  }
  mutable std::shared_ptr<TypeTable> tables_[BSGPU_F_NUM_TYPES];        // owned — or, while undo_[ty] is set, another graph's table to be read through the log
  mutable std::shared_ptr<TableUndo> undo_[BSGPU_F_NUM_TYPES];          // (a snapshot that has not needed its tables yet)
  std::vector<bsgpu_camera> cameras_;
  std::map<int32_t, MarginalRow> marginal_rows_;                // by constraint slot
  // rows of each table written since this graph's context last took the table (bsgpu_sync_factors_indirect's change list); a
  // clone starts un-synced: its context, if it ever opens one, reads the tables whole
  std::vector<int32_t> dirty_[BSGPU_F_NUM_TYPES];
  bool synced_[BSGPU_F_NUM_TYPES] = {};
  void markDirty(int ty, uint32_t row, size_t rows) {
    if (!synced_[ty]) return;
    dirty_[ty].push_back((int32_t)row);
    if (dirty_[ty].size() > rows / 2 + 4096) { synced_[ty] = false; dirty_[ty].clear(); }   // (cheaper to re-read the table)
  }
  fuse_core::FactorTables pack_scratch_;
  std::vector<int32_t> slot_scratch_, vars_scratch_;
  std::vector<uint8_t> cmark_, vmark_;   // update(): constraints being removed / variables whose lists lose some of them
  // this graph's own version of every table it still reads through an undo log (const: a lazy copy, not a change of the graph)
  void materialiseTables() const {
    if (tables_own_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::recursive_mutex> lazy(lazy_mu_);
    if (tables_own_.load(std::memory_order_relaxed)) return;
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      if (!undo_[ty]) continue;
      std::shared_ptr<TypeTable> src = tables_[ty];
      std::shared_ptr<TypeTable> n = TablePool::instance().acquire(ty);
      {
        std::lock_guard<std::mutex> lk(src->mu);
        const TableUndo& u = *undo_[ty];
        const size_t rows = u.rows, ni = (size_t)src->nidx, nc = src->rows ? src->consts.size() / src->rows : (u.row.empty() ? 0 : u.consts.size() / u.row.size());
        n->rows = rows; n->nvar = src->nvar; n->nidx = src->nidx;
        const size_t shared = std::min(rows, src->rows);   // (rows beyond the table's present size are all in the log)
        n->idx.reserve(rows * ni + rows * ni / 8 + 64); n->consts.reserve(rows * nc + rows * nc / 8 + 64);
        n->idx.assign(src->idx.begin(), src->idx.begin() + shared * ni); n->idx.resize(rows * ni);
        n->consts.assign(src->consts.begin(), src->consts.begin() + shared * nc); n->consts.resize(rows * nc);
        n->loss_kind.assign(src->loss_kind.begin(), src->loss_kind.begin() + shared); n->loss_kind.resize(rows);
        n->loss_a.assign(src->loss_a.begin(), src->loss_a.begin() + shared); n->loss_a.resize(rows);
        n->owner.assign(src->owner.begin(), src->owner.begin() + shared); n->owner.resize(rows);
        for (size_t i = 0; i < u.row.size(); ++i) {
          const size_t q = u.row[i];
          std::copy(u.idx.begin() + i * ni, u.idx.begin() + (i + 1) * ni, n->idx.begin() + q * ni);
          std::copy(u.consts.begin() + i * nc, u.consts.begin() + (i + 1) * nc, n->consts.begin() + q * nc);
          n->loss_kind[q] = u.loss_kind[i]; n->loss_a[q] = u.loss_a[i]; n->owner[q] = u.owner[i];
        }
        undo_[ty]->detached = true;
      }
      tables_[ty] = std::move(n);
      undo_[ty].reset();
    }
    tables_own_.store(true, std::memory_order_release);
  }
  TypeTable& tableMut(int ty) {
    materialiseTables();
    auto& t = tables_[ty];
    if (!t) t = TablePool::instance().acquire(ty);
    return *t;
  }
  template <class F>
  void forEachVariableOf(int32_t cs, F fn) const {
    const int ty = ctype_[cs];
    if (ty >= 0) materialiseTables();
    if (ty >= 0) { const TypeTable& tb = *tables_[ty]; for (int k = 0; k < tb.nvar; ++k) fn(tb.idx[(size_t)crow_[cs] * tb.nidx + k]); }
    else if (ty == kMarginal) for (int32_t s : marginal_rows_.at(cs).vars) fn(s);
    else if (ty == kUnpacked) for (const auto& u : cptr_[cs]->variables()) { const int32_t s = vindex_.find(u); if (s >= 0) fn(s); }
  }
  void appendRow(const fuse_core::Constraint& c, int32_t cs, const std::vector<int32_t>& vars) {
    fuse_core::FactorTables& t1 = pack_scratch_;   // (reused: a transaction packs thousands of constraints)
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) { t1.idx[ty].clear(); t1.consts[ty].clear(); t1.loss_kind[ty].clear(); t1.loss_a[ty].clear(); }
    t1.marginals.clear(); t1.cameras.clear();
    std::vector<int32_t>& sl = slot_scratch_;
    sl.resize(vars.size());
    for (size_t i = 0; i < sl.size(); ++i) sl[i] = (int32_t)i;
    c.pack(fuse_core::BlockOf(sl.data()), t1);
    ctype_[cs] = kUnpacked;
    if (!t1.marginals.empty()) {
      MarginalRow m; m.e = std::move(t1.marginals[0]); m.vars = vars;
      marginal_rows_[cs] = std::move(m);
      ctype_[cs] = kMarginal;
      return;
    }
    for (int ty = 0; ty < BSGPU_F_NUM_TYPES; ++ty) {
      if (!t1.count(ty)) continue;
      TypeTable& tb = tableMut(ty);
      std::unique_lock<std::mutex> lk(tb.mu, std::defer_lock);
      if (!tb.watchers.empty()) lk.lock();   // (an append may move the arrays a snapshot is reading; rows it adds are in no log)
      const int nvar = (int)vars.size(), nidx = (int)t1.idx[ty].size();
      if (!tb.rows && tb.idx.empty()) { tb.nvar = nvar; tb.nidx = nidx; }
      tb.idx.insert(tb.idx.end(), vars.begin(), vars.end());
      if (nidx > nvar) {   // camera types: the row's last column is the id in the graph-wide camera table
        const bsgpu_camera& c1 = t1.cameras.at(t1.idx[ty][nvar]);
        int32_t cid = -1;
        for (size_t i = 0; i < cameras_.size(); ++i) if (std::memcmp(&cameras_[i], &c1, sizeof(c1)) == 0) { cid = (int32_t)i; break; }
        if (cid < 0) { cameras_.push_back(c1); cid = (int32_t)cameras_.size() - 1; }
        tb.idx.push_back(cid);
      }
      tb.consts.insert(tb.consts.end(), t1.consts[ty].begin(), t1.consts[ty].end());
      tb.loss_kind.push_back(t1.loss_kind[ty][0]); tb.loss_a.push_back(t1.loss_a[ty][0]);
      tb.owner.push_back(cs);
      ctype_[cs] = ty; crow_[cs] = (uint32_t)tb.rows++;
      markDirty(ty, crow_[cs], tb.rows);
      return;
    }
  }
  void removeRow(int32_t cs) {
    const int ty = ctype_[cs];
    if (ty == kMarginal) { marginal_rows_.erase(cs); return; }
    if (ty < 0) return;
    TypeTable& tb = tableMut(ty);
    const size_t last = tb.rows - 1, r = crow_[cs], nc = tb.consts.size() / tb.rows, ni = (size_t)tb.nidx;
    std::unique_lock<std::mutex> lk(tb.mu, std::defer_lock);
    if (!tb.watchers.empty()) {
      lk.lock();
      tb.pruneWatchers();
      tb.save(r); tb.save(last);       // what the snapshots still see in these two rows
    }
    if (r != last) {
      for (size_t k = 0; k < ni; ++k) tb.idx[r * ni + k] = tb.idx[last * ni + k];
      for (size_t k = 0; k < nc; ++k) tb.consts[r * nc + k] = tb.consts[last * nc + k];
      tb.loss_kind[r] = tb.loss_kind[last]; tb.loss_a[r] = tb.loss_a[last];
      tb.owner[r] = tb.owner[last];
      crow_[tb.owner[r]] = (uint32_t)r;
      markDirty(ty, (uint32_t)r, tb.rows);
    }
    tb.idx.resize(last * ni);
    tb.consts.resize(last * nc);
    tb.loss_kind.pop_back(); tb.loss_a.pop_back(); tb.owner.pop_back();
    tb.rows = last;
  }
  // variable slot -> constraint slots: maintained incrementally once built; a clone rebuilds it on first use from the tables
  void ensureConnectivity() const {
    if (connectivity_valid_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::recursive_mutex> lk(lazy_mu_);
    if (connectivity_valid_.load(std::memory_order_relaxed)) return;
    conn_.assign(vslots_.size(), {});
    for (size_t cs = 0; cs < ctype_.size(); ++cs)
      if (ctype_[cs] != kFree)
        forEachVariableOf((int32_t)cs, [&](int32_t s) { auto& l = conn_[s]; if (std::find(l.begin(), l.end(), (int32_t)cs) == l.end()) l.push_back((int32_t)cs); });
    connectivity_valid_.store(true, std::memory_order_release);
  }
  // (tables_ / undo_ are mutable: materialiseTables() is a lazy copy behind const accessors)
  mutable std::vector<std::vector<int32_t>> conn_;
  mutable std::atomic<bool> connectivity_valid_{true};
  bsgpu_summary last_summary_{};
  Flat flat_;
};

}  // namespace bs_optimizers

namespace fuse_constraints {
// the reference's call shape: fuse_constraints::marginalizeVariables(ros::this_node::getName(), vars, *graph_)
inline fuse_core::Transaction marginalizeVariables(const std::string& source, const std::vector<fuse_core::UUID>& marginalized_variables,
                                                   bs_optimizers::GpuGraph& graph) {
  return graph.marginalizeVariables(source, marginalized_variables);
}
}  // namespace fuse_constraints
