// bs_optimizers::FixedLagSmoother — ROS-free core of the reference's optimizer
// (bs_optimizers/src/fixed_lag_smoother.cpp): the pending-transaction queue (:548-627, sorted by stamp),
// one optimisation cycle (:185-309: merge queue, drop constraints that touch already-marginalised
// variables :199-216, graph update, lag expiration :141-149, pseudo-marginalisation :244-268 with the
// "MARGINALIZATION" prior on the first in-window state :742-797, optimize :281, usable-solution check
// :286-295), and the [EXT] fuse VariableStampIndex it relies on (:153-159).
// The queue rules of transactionCallback (:548-627) and processQueue (:335-477) are mirrored: ignition sensors start the
// optimizer and get their first transaction processed on its own, transactions older than the start / the lag window are
// purged, a transaction whose motion models cannot be generated yet is retried until `transaction_timeout` and blocks the
// later transactions of its sensor meanwhile.  Sensor-model and motion-model PLUGINS themselves are out of scope:
// a sensor is a name + its ignition flag (registerSensorModel), the motion models are one callback (setMotionModelCallback,
// [EXT] fuse_optimizers::Optimizer::applyMotionModels).  ROS timers / threads / services: the caller drives optimizeOnce().
#pragma once
#include <atomic>
#include <deque>
#include <mutex>

#include "gpu_graph.h"

namespace bs_optimizers {

// [EXT] fuse_optimizers::VariableStampIndex: stamped variables expire with their own stamp; unstamped
// ones (landmarks, extrinsics) once every stamped variable they are connected to has expired.
class VariableStampIndex {
 public:
  void addNewTransaction(const fuse_core::Transaction& t) {
    for (const auto& v : t.addedVariables()) {
      auto& e = vars_[v->uuid()];
      e.stamped = v->isStamped();
      if (e.stamped) e.stamp = v->stamp();
    }
    for (const auto& c : t.addedConstraints()) {
      constraints_[c->uuid()] = c->variables();
      for (const auto& u : c->variables()) vars_[u].constraints.insert(c->uuid());
    }
    applyRemovals(t);
  }
  void addMarginalTransaction(const fuse_core::Transaction& t) {
    for (const auto& c : t.addedConstraints()) {
      constraints_[c->uuid()] = c->variables();
      for (const auto& u : c->variables()) vars_[u].constraints.insert(c->uuid());
    }
    applyRemovals(t);
  }
  fuse_core::Time currentStamp() const {
    fuse_core::Time m;
    bool any = false;
    for (const auto& kv : vars_) if (kv.second.stamped && (!any || kv.second.stamp > m)) { m = kv.second.stamp; any = true; }
    return m;
  }
  // variables whose (effective) stamp is older than `stamp`
  std::vector<fuse_core::UUID> query(const fuse_core::Time& stamp) const {
    std::vector<fuse_core::UUID> out;
    for (const auto& kv : vars_) {
      if (kv.second.stamped) { if (kv.second.stamp < stamp) out.push_back(kv.first); continue; }
      bool any = false, all_old = true;
      for (const auto& cu : kv.second.constraints)
        for (const auto& vu : constraints_.at(cu)) {
          auto it = vars_.find(vu);
          if (it == vars_.end() || !it->second.stamped) continue;
          any = true;
          if (!(it->second.stamp < stamp)) all_old = false;
        }
      if (any && all_old) out.push_back(kv.first);
    }
    return out;
  }
  size_t size() const { return vars_.size(); }
 private:
  struct Entry { bool stamped = false; fuse_core::Time stamp; std::set<fuse_core::UUID> constraints; };
  void applyRemovals(const fuse_core::Transaction& t) {
    for (const auto& cu : t.removedConstraints()) {
      auto it = constraints_.find(cu);
      if (it == constraints_.end()) continue;
      for (const auto& vu : it->second) { auto v = vars_.find(vu); if (v != vars_.end()) v->second.constraints.erase(cu); }
      constraints_.erase(it);
    }
    for (const auto& vu : t.removedVariables()) vars_.erase(vu);
  }
  std::map<fuse_core::UUID, Entry> vars_;
  std::map<fuse_core::UUID, std::vector<fuse_core::UUID>> constraints_;
};

struct FixedLagSmootherParams {     // [EXT] fuse_optimizers::FixedLagSmootherParams + pseudo_marginalization (:92-93)
  double lag_duration = 7.0;        // vio.yaml:3
  double optimization_period = 0.07;  // vio.yaml:2 (the caller's timer)
  bool pseudo_marginalization = true;  // vio.yaml:4 — on in every shipped config
  double transaction_timeout = 0.1;    // [EXT] fuse_optimizers::FixedLagSmootherParams default: how long a transaction whose motion
                                       // models cannot be generated is kept for another try (fixed_lag_smoother.cpp:454-470)
  ceres_compat::SolverOptions solver_options = ceres_compat::SolverOptions::Vio();
};

class FixedLagSmoother {
 public:
  FixedLagSmoother(GpuGraph::UniquePtr graph, FixedLagSmootherParams params = FixedLagSmootherParams())
      : graph_(std::move(graph)), params_(params) {}

  // [EXT] fuse_optimizers::Optimizer loads the sensor models from the parameter server; here: a name and its `ignition` flag.
  // With no ignition sensor registered the optimizer starts at once (autostart(), :125-137) — also the behaviour when nothing is
  // registered at all.
  void registerSensorModel(const std::string& name, bool ignition) { sensor_models_[name] = ignition; }
  // [EXT] Optimizer::applyMotionModels(sensor, transaction): generates the motion-model constraints between the stamps the
  // transaction involves and merges them into it; false = not possible yet (e.g. the IMU buffer does not cover the stamp)
  using MotionModelCallback = std::function<bool(const std::string&, fuse_core::Transaction&)>;
  void setMotionModelCallback(MotionModelCallback cb) { motion_models_ = std::move(cb); }
  bool started() const { return started_; }

  // fixed_lag_smoother.cpp:548-627.  The queue is kept oldest-first (the reference keeps it newest-first and works from the back).
  void transactionCallback(const std::string& sensor_name, fuse_core::Transaction::SharedPtr transaction) {
    autostart();
    const fuse_core::Time max_time = transaction->maxStamp();
    if (started_) {                                                          // :552-561 before the start time: ignored
      std::lock_guard<std::mutex> tl(start_time_mutex_);
      if (max_time < start_time_) return;
    }
    std::lock_guard<std::mutex> lock(pending_transactions_mutex_);
    auto pos = std::upper_bound(pending_.begin(), pending_.end(), transaction->stamp(),
                                [](const fuse_core::Time& s, const Pending& p) { return s < p.transaction->stamp(); });
    pos = pending_.insert(pos, Pending{sensor_name, std::move(transaction)});
    if (started_) return;
    if (isIgnition(sensor_name)) {                                           // :584-611
      const fuse_core::Time min_time = pos->transaction->minStamp();
      { std::lock_guard<std::mutex> tl(start_time_mutex_); start_time_ = min_time; }
      started_ = true; ignited_ = true;
      pending_.erase(std::remove_if(pending_.begin(), pending_.end(),
                                    [&](const Pending& p) {
                                      return p.sensor_name != sensor_name &&
                                             (p.transaction->minStamp() < min_time || !(max_time < p.transaction->maxStamp()));
                                    }),
                     pending_.end());
    } else {                                                                 // :612-626 bounded queue while waiting for an ignition sensor
      const fuse_core::Time last_pending_time = pending_.back().transaction->stamp();
      const fuse_core::Time purge_time = last_pending_time + (-params_.transaction_timeout);
      if (fuse_core::Time() + params_.transaction_timeout < last_pending_time)
        while (!pending_.empty() && pending_.front().transaction->maxStamp() < purge_time) pending_.pop_front();
    }
  }
  size_t pendingTransactions() const { std::lock_guard<std::mutex> lock(pending_transactions_mutex_); return pending_.size(); }

  enum class CycleResult { NothingToDo, Optimized, UnusableSolution, GraphUpdateFailed };

  // one pass of the body of optimizationLoop() (:185-309)
  CycleResult optimizeOnce() {
    std::lock_guard<std::mutex> lock(optimization_mutex_);
    autostart();
    if (!started_) return CycleResult::NothingToDo;                          // :312-316 no ignition transaction yet
    fuse_core::Transaction new_transaction;
    processQueue(new_transaction, lag_expiration_);                          // :194
    if (new_transaction.empty()) return CycleResult::NothingToDo;            // :197
    // :199-216 drop added constraints that touch variables the previous cycle marginalised.  Only CONSTRAINTS are filtered (the
    // reference calls new_transaction->removeConstraint on them); a variable the transaction adds again stays added.
    fuse_core::Transaction filtered;
    filtered.stamp(new_transaction.stamp());
    for (const auto& s : new_transaction.involvedStamps()) filtered.addInvolvedStamp(s);
    for (const auto& v : new_transaction.addedVariables()) filtered.addVariable(v);
    for (const auto& c : new_transaction.addedConstraints()) {
      bool faulty = false;
      for (const auto& vu : c->variables())
        for (const auto& m : marginal_transaction_.removedVariables()) if (vu == m) faulty = true;
      if (faulty) { ++num_dropped_constraints_; continue; }
      filtered.addConstraint(c);
    }
    for (const auto& u : new_transaction.removedConstraints()) filtered.removeConstraint(u);
    for (const auto& u : new_transaction.removedVariables()) filtered.removeVariable(u);
    try {
      graph_->update(filtered);            // :219-236
    } catch (const std::exception& ex) {
      last_error_ = ex.what();
      return CycleResult::GraphUpdateFailed;
    }
    // marginalisation (:238-276)
    timestamp_tracking_.addNewTransaction(filtered);
    lag_expiration_ = computeLagExpirationTime();
    const auto vars_to_marginalize = timestamp_tracking_.query(lag_expiration_);
    marginal_transaction_ = fuse_core::Transaction();
    if (params_.pseudo_marginalization) {
      if (vars_to_marginalize.size() > 1) {   // sic: "> 1" (:246)
        for (const auto& uuid : vars_to_marginalize) {
          try {
            for (const auto* c : graph_->getConnectedConstraints(uuid)) marginal_transaction_.removeConstraint(c->uuid());
            marginal_transaction_.removeVariable(uuid);
          } catch (const std::exception&) {}  // swallowed, like the reference (:257)
        }
        const bs_common::ImuState first = GetWindowStartState();
        Mat15 cov = 0.00001 * Mat15::Identity();   // :266
        marginal_transaction_.addConstraint(std::make_shared<bs_constraints::AbsoluteImuState3DStampedConstraint>(
            "MARGINALIZATION", first, first.GetStateVector(), cov));
      }
    } else if (!vars_to_marginalize.empty()) {
      // :269-272 true marginalisation: Schur complement of the expired variables on the device -> one MarginalConstraint
      marginal_transaction_ = fuse_constraints::marginalizeVariables("fixed_lag_smoother", vars_to_marginalize, *graph_);
    }
    // removals of constraints must precede removals of variables and may list a constraint twice
    graph_->update(dedup(marginal_transaction_));
    timestamp_tracking_.addMarginalTransaction(marginal_transaction_);
    // :281 — the hot call
    summary_ = graph_->optimize(params_.solver_options);
    have_summary_ = true;
    ++num_cycles_;
    if (!summary_.IsSolutionUsable()) return CycleResult::UnusableSolution;   // :286-295
    // :305-308 notify(new_transaction, graph_->clone()): the publishers / sensor models get the transaction and a snapshot
    // of the optimised graph ([EXT] fuse_optimizers::Optimizer::notify).  The snapshot shares the constraint side with
    // graph_ copy-on-write (gpu_graph.h), so handing one out every cycle costs O(variables).
    if (notify_) notify_(std::make_shared<const fuse_core::Transaction>(std::move(filtered)), std::shared_ptr<const GpuGraph>(graph_->clone()));
    return CycleResult::Optimized;
  }
  using NotifyCallback = std::function<void(std::shared_ptr<const fuse_core::Transaction>, std::shared_ptr<const GpuGraph>)>;
  void setNotifyCallback(NotifyCallback cb) { notify_ = std::move(cb); }

  const ceres_compat::SolverSummary& summary() const { return summary_; }
  const GpuGraph& graph() const { return *graph_; }
  GpuGraph& graph() { return *graph_; }
  fuse_core::Time lagExpiration() const { return lag_expiration_; }
  // setDiagnostics (:676-740) without ROS: the same fields under the same names, the level and message of
  // terminationTypeToDiagnosticStatus (:647-669).  Level: 0 OK, 1 WARN, 2 ERROR (diagnostic_msgs::DiagnosticStatus).
  struct Diagnostics {
    int level = 0;
    std::string message;
    std::vector<std::pair<std::string, std::string>> values;
    const std::string* find(const std::string& key) const { for (const auto& kv : values) if (kv.first == key) return &kv.second; return nullptr; }
  };
  Diagnostics diagnostics() const {
    Diagnostics d;
    const bool started = started_;
    d.values.emplace_back("Started", started ? "True" : "False");
    d.values.emplace_back("Pending Transactions", std::to_string(pendingTransactions()));
    if (!started) return d;
    ceres_compat::SolverSummary summary;
    bool have = false;
    {
      std::unique_lock<std::mutex> lock(optimization_mutex_, std::try_to_lock);   // (:697-705: a running optimisation is reported, not waited for)
      if (lock) { summary = summary_; have = have_summary_; }
      else d.message = "Optimization running";
    }
    if (have) {   // (the reference tests total_time_in_seconds >= 0: -1 in a default-constructed ceres summary)
      const char* tt = summary.termination_type == ceres_compat::CONVERGENCE ? "CONVERGENCE" : summary.termination_type == ceres_compat::NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE";
      d.values.emplace_back("Optimization Termination Type", tt);
      d.values.emplace_back("Optimization Total Time [s]", std::to_string(summary.total_time_in_seconds));
      d.values.emplace_back("Optimization Iterations", std::to_string(summary.iterations.size()));
      d.values.emplace_back("Initial Cost", std::to_string(summary.initial_cost));
      d.values.emplace_back("Final Cost", std::to_string(summary.final_cost));
      int level = 2; const char* msg = "Optimization failed";
      if (summary.termination_type == ceres_compat::CONVERGENCE) { level = 0; msg = "Optimization converged"; }
      else if (summary.termination_type == ceres_compat::NO_CONVERGENCE) { level = 1; msg = "Optimization didn't converge"; }
      if (level > d.level || d.message.empty()) { d.level = std::max(d.level, level); d.message = msg; }   // mergeSummary: the worse level wins
    }
    return d;
  }
  int numCycles() const { return num_cycles_; }
  int numDroppedConstraints() const { return num_dropped_constraints_; }
  const std::string& lastError() const { return last_error_; }
  // :479-546 reset service: clear graph and queue
  void reset() {
    std::lock_guard<std::mutex> lock(optimization_mutex_);
    std::lock_guard<std::mutex> qlock(pending_transactions_mutex_);
    pending_.clear(); graph_->clear(); timestamp_tracking_ = VariableStampIndex(); marginal_transaction_ = fuse_core::Transaction();
    started_ = false; ignited_ = false; lag_expiration_ = fuse_core::Time();
    { std::lock_guard<std::mutex> tl(start_time_mutex_); start_time_ = fuse_core::Time(); }
  }

  // fixed_lag_smoother.cpp:335-477
  void processQueue(fuse_core::Transaction& transaction, const fuse_core::Time& lag_expiration) {
    std::lock_guard<std::mutex> lock(pending_transactions_mutex_);
    if (pending_.empty()) return;
    if (ignited_) {
      // the transaction that started things up is the oldest one; it is processed on its own so that the motion models see an
      // optimised first state before the other queued transactions are attached to it (:344-356)
      ignited_ = false;
      Pending& element = pending_.front();
      if (!isIgnition(element.sensor_name)) {
        ++num_queue_errors_;                                                 // :364-371 logged; processed with the others below
      } else {
        if (applyMotionModels(element.sensor_name, *element.transaction)) {
          transaction.merge(*element.transaction, true);
          pending_.pop_front();
        } else {
          // :380-414 an ignition transaction that cannot be processed is dropped with everything older than the NEXT ignition
          // transaction; without one the optimizer goes back to "not started"
          ++num_queue_errors_;
          pending_.pop_front();
          auto next = std::find_if(pending_.begin(), pending_.end(), [this](const Pending& p) { return isIgnition(p.sensor_name); });
          if (next == pending_.end()) started_ = false;
          else { pending_.erase(pending_.begin(), next); ignited_ = true; }
        }
        return;
      }
    }
    const fuse_core::Time current_time = pending_.back().transaction->stamp();   // the most recent stamp (:424)
    std::vector<std::string> sensor_blacklist;
    for (auto it = pending_.begin(); it != pending_.end();) {                // oldest first, like the reference's reverse walk
      const fuse_core::Time min_stamp = it->transaction->minStamp();
      if (min_stamp < lag_expiration) {
        ++num_expired_transactions_;                                         // :431-444 older than the lag window
        it = pending_.erase(it);
      } else if (std::find(sensor_blacklist.begin(), sensor_blacklist.end(), it->sensor_name) != sensor_blacklist.end()) {
        ++it;                                                                // :445-448
      } else if (applyMotionModels(it->sensor_name, *it->transaction)) {
        transaction.merge(*it->transaction, true);                           // :449-453
        it = pending_.erase(it);
      } else if (it->transaction->maxStamp() + params_.transaction_timeout < current_time) {
        ++num_timed_out_transactions_;                                       // :454-470
        it = pending_.erase(it);
      } else {
        sensor_blacklist.push_back(it->sensor_name);                         // :471-474 try again next cycle, keep the sensor's order
        ++it;
      }
    }
  }
  int numExpiredTransactions() const { return num_expired_transactions_; }
  int numTimedOutTransactions() const { return num_timed_out_transactions_; }
  int numQueueErrors() const { return num_queue_errors_; }

  // :742-797: the first in-window state = smallest Position3DStamped stamp newer than the lag expiration
  bs_common::ImuState GetWindowStartState() const {
    bool found = false;
    fuse_core::Time first;
    for (const auto* v : graph_->getVariables()) {
      if (v->type() != "fuse_variables::Position3DStamped") continue;
      if (v->stamp() > lag_expiration_ && (!found || v->stamp() < first)) { first = v->stamp(); found = true; }
    }
    if (!found) throw std::out_of_range("Invalid start state of new graph, no in-window position variable.");
    bs_common::ImuState s(first);
    if (!s.Update(*graph_)) throw std::out_of_range("Invalid start state of new graph, not all imu state variables exist.");
    return s;
  }

 private:
  using Mat15 = bs_math::Mat<15, 15>;
  struct Pending { std::string sensor_name; fuse_core::Transaction::SharedPtr transaction; };
  fuse_core::Time computeLagExpirationTime() const {   // :141-149
    const fuse_core::Time now = timestamp_tracking_.currentStamp();
    std::lock_guard<std::mutex> tl(start_time_mutex_);
    return (start_time_ + params_.lag_duration < now) ? now + (-params_.lag_duration) : start_time_;
  }
  static fuse_core::Transaction dedup(const fuse_core::Transaction& t) {
    fuse_core::Transaction o;
    std::set<fuse_core::UUID> seen;
    for (const auto& u : t.removedConstraints()) if (seen.insert(u).second) o.removeConstraint(u);
    for (const auto& u : t.removedVariables()) o.removeVariable(u);
    for (const auto& c : t.addedConstraints()) o.addConstraint(c);
    return o;
  }
  bool isIgnition(const std::string& sensor) const { auto it = sensor_models_.find(sensor); return it != sensor_models_.end() && it->second; }
  bool applyMotionModels(const std::string& sensor, fuse_core::Transaction& t) const { return motion_models_ ? motion_models_(sensor, t) : true; }
  void autostart() {   // :125-137: no ignition sensor configured -> start immediately, start time 0
    if (started_) return;
    for (const auto& kv : sensor_models_) if (kv.second) return;
    { std::lock_guard<std::mutex> tl(start_time_mutex_); start_time_ = fuse_core::Time(); }
    started_ = true;
  }
  GpuGraph::UniquePtr graph_;
  NotifyCallback notify_;
  MotionModelCallback motion_models_;
  std::map<std::string, bool> sensor_models_;
  bool ignited_ = false;
  int num_expired_transactions_ = 0, num_timed_out_transactions_ = 0, num_queue_errors_ = 0;
  FixedLagSmootherParams params_;
  mutable std::mutex pending_transactions_mutex_;
  mutable std::mutex optimization_mutex_;
  std::deque<Pending> pending_;
  VariableStampIndex timestamp_tracking_;
  fuse_core::Transaction marginal_transaction_;
  fuse_core::Time lag_expiration_, start_time_;   // start_time_: under start_time_mutex_ (the reference's start_time_mutex_)
  mutable std::mutex start_time_mutex_;
  std::atomic<bool> started_{false};   // (set by sensor threads in transactionCallback, read by the optimisation thread: the reference's std::atomic<bool>)
  ceres_compat::SolverSummary summary_;
  bool have_summary_ = false;   // (an optimisation has run: the diagnostics report its summary)
  int num_cycles_ = 0, num_dropped_constraints_ = 0;
  std::string last_error_;
};

}  // namespace bs_optimizers
