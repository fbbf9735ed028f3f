// bsgpu_solve_batch: several windows advanced by ONE set of launches per LM iteration.
//
// Reference: the unit of independence is the window — the local fixed-lag smoother, the global mapper and the submap refinements run
// side by side (beam_slam_launch/launch/vio.launch:19-30), and submap refinement is a serial loop of solves over fresh graphs
// (bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115), at a few tens of key frames each (vio.yaml:3,56).  A window of that
// size is 16 workgroups per kernel on a 256-CU chip and its LM iteration is a chain of ~10 dependent launches of a few microseconds of
// work each: alone it uses a few per cent of the device, and one host thread + stream per window (the first form of this entry point)
// does not overlap them better than the hardware queues do.
//
// Here every kernel of the step has a `_batch` form (k_*.hip): blockIdx.y picks a window from a list, and the kernel's arguments for
// that window come from a table in device memory — entry w is exactly what window w's lone launch passes, so every window's tables,
// partial sums and reductions are laid out as in its lone solve and its iterations come out the same.  What changes per iteration
// (which windows still iterate, whose step was accepted or rejected, the radii) is one small block uploaded once per iteration
// (BatchDyn).  The LM decisions stay on the host, per window, restated from bsgpu_solve.cpp's loop ([EXT] ceres TrustRegionMinimizer);
// the host waits for the stamps of all windows' end-of-step reductions, decides, and enqueues the next set.  Windows that converge drop
// out of the lists.  Windows the batched kernels do not cover (pose-only graphs, PCG, inverse-depth landmarks, dense priors, hipGraph
// replay) are solved by the thread-per-window form, in the same call.
#include <thread>

#include "bsgpu_ctx.h"

namespace bsg {
namespace {

enum { K_FIRST = 0, K_ACCEPT = 1, K_REJECT = 2 };

// [EXT] ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy as bsgpu_solve.cpp solve() restates them, cut at the points
// where solve() enqueues a step and waits for its scalars: advance() is called with the scalars of the requested step in c->h_scal and
// either requests the next step (kind, radius, gradient_only) or finishes the summary.
struct LmWindow {
  bsgpu_ctx* c = nullptr;
  const bsgpu_options* o = nullptr;
  bsgpu_summary* sum = nullptr;
  double radius = 0.0, decrease_factor = 2.0, x_cost = 0.0;
  bsgpu_iteration it{};
  int num_consecutive_invalid = 0;
  const char* msg = "";
  bool done = false, timed_out = false, absorb = false;
  int kind = K_FIRST;
  bool grad_only = false;
  std::chrono::steady_clock::time_point t_start;

  void request(int k, bool g) { kind = k; grad_only = g; absorb = true; }
  void start() {
    std::memset(sum, 0, sizeof(*sum));
    c->iters.clear();
    sum->num_parameters_tangent = c->n_tan; sum->num_residuals = c->n_res;
    sum->linear_solver_used = BSGPU_LINEAR_SCHUR_CHOLESKY;
    radius = o->initial_trust_region_radius; decrease_factor = 2.0;
    kind = K_FIRST; grad_only = false; absorb = false; done = false; timed_out = false;
    t_start = std::chrono::steady_clock::now();
  }
  void finish() {
    done = true;
    sum->num_iterations = (int)c->iters.size() - 1;
    sum->num_inner_iterations = 0;
    sum->is_solution_usable = (sum->termination_type == BSGPU_CONVERGENCE || sum->termination_type == BSGPU_NO_CONVERGENCE) ? 1 : 0;
    std::snprintf(sum->message, sizeof(sum->message), "%s", msg);
  }
  // after the scalars of STEP_FIRST
  void begin() {
    x_cost = c->h_scal[SC_COST_X];
    std::memset(&it, 0, sizeof(it));
    it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost;
    it.gradient_max_norm = c->h_scal[SC_GRAD_MAX]; it.gradient_norm = std::sqrt(c->h_scal[SC_GRAD_NORM2]);
    sum->initial_cost = x_cost; sum->fixed_cost = 0.0;
    sum->termination_type = BSGPU_NO_CONVERGENCE;
    num_consecutive_invalid = 0;
    absorb = false;
    if (!std::isfinite(x_cost)) { sum->termination_type = BSGPU_FAILURE; msg = "Initial cost is not finite."; sum->final_cost = sum->initial_cost; finish(); return; }
    advance();
  }
  void advance() {
    const double* h = c->h_scal;
    if (absorb && it.step_is_successful) {
      x_cost = h[SC_COST_X];
      it.cost = x_cost;
      it.gradient_max_norm = h[SC_GRAD_MAX];
      it.gradient_norm = std::sqrt(h[SC_GRAD_NORM2]);
    }
    absorb = false;
    for (;;) {
      if (it.step_is_successful) { if (it.iteration > 0) sum->num_successful_steps++; } else sum->num_unsuccessful_steps++;
      it.trust_region_radius = radius;
      c->iters.push_back(it);
      if (o->max_solver_time_in_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= o->max_solver_time_in_seconds) { msg = "Maximum solver time reached."; break; }
      if (it.iteration >= o->max_num_iterations) { msg = "Maximum number of iterations reached."; break; }
      if (it.step_is_successful && it.gradient_max_norm <= o->gradient_tolerance) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
      if (radius <= o->min_trust_region_radius) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
      if (h[SC_CHOL_FAIL] == 2.0) { timed_out = true; done = true; return; }   // a single-launch kernel's wait timed out (shared GPU): the caller re-solves this window alone
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
        radius *= 0.5;
        it.cost = x_cost; it.step_is_successful = 0;
        if (it.iteration >= o->max_num_iterations) continue;
        request(K_REJECT, false);
        return;
      }
      num_consecutive_invalid = 0;
      double cand_cost = h[SC_COST_CAND];
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      it.step_norm = std::sqrt(h[SC_STEP_NORM2]);
      const double x_norm = std::sqrt(h[SC_X_NORM2]);
      if (it.step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Parameter tolerance reached."; break; }
      it.cost_change = x_cost - cand_cost;
      if (std::fabs(it.cost_change) <= o->function_tolerance * x_cost) { sum->termination_type = BSGPU_CONVERGENCE; msg = "Function tolerance reached."; break; }
      it.relative_decrease = (x_cost - cand_cost) / mcc;
      const bool last_iteration = it.iteration >= o->max_num_iterations;
      if (it.relative_decrease > o->min_relative_decrease) {
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
        radius = std::min(o->max_trust_region_radius, radius);
        decrease_factor = 2.0;
        it.step_is_successful = 1;
        request(K_ACCEPT, last_iteration);
        return;
      }
      it.step_is_successful = 0;
      radius = radius / decrease_factor; decrease_factor *= 2.0;
      it.cost = cand_cost;
      if (last_iteration) continue;
      request(K_REJECT, false);
      return;
    }
    sum->final_cost = x_cost;
    finish();
  }
};

// Windows per call up to which the LM diagonal / gradient norms ride in the factorisation's launch (bsgpu_solve.cpp does that for a lone
// window); above it they take their own batched launch and the task lists without those tasks.  0: never — a batch is bound by the number
// of its (LDS-heavy, one per CU) factorisation workgroups, and the separate launch's 256-thread workgroups are the cheaper form: 8 windows
// of C2 5 350 against 5 150 LM it/s, 32 windows of 20 KF x 500 99 100 against 91 700.
constexpr int kBatchDiagInCholMax = 0;
// the argument tables of one set of windows, built once per (contexts, finalize generations, options that enter the tables)
struct BatchPlan {
  std::vector<bsgpu_ctx*> ctxs;
  std::vector<uint64_t> gens;
  std::vector<const double*> xptr;   // each window's current-point buffer when the tables were built (a lone solve swaps it with the candidate's)
  std::vector<double*> backup;       // the windows' values when the call began (a window whose single-launch kernels time out is solved again, alone)
  int jacobi = -1;
  double lm_lo = 0.0, lm_hi = 0.0;
  int device = -1;
  hipStream_t stream = nullptr;
  BatchDyn* h_dyn = nullptr;   // pinned
  BatchDyn* d_dyn = nullptr;
  std::vector<void*> dev_allocs;
  BatchArgTable t_eval_x, t_eval_cand, t_eval_spec, t_lm, t_lm_tail, t_pairs, t_gn, t_chol, t_bs[4], t_backsub, t_reduce, t_accept, t_backup;
  std::vector<int> bs_form;
  std::vector<char> diag_in_chol;    // the window's LM diagonal / gradient norms ride in its factorisation's launch
  void release() {
    for (void* p : dev_allocs) (void)hipFree(p);
    dev_allocs.clear();
    for (BatchArgTable* t : tables()) *t = BatchArgTable();
    ctxs.clear(); gens.clear(); bs_form.clear(); diag_in_chol.clear(); xptr.clear(); backup.clear();
  }
  std::vector<BatchArgTable*> tables() {
    return {&t_eval_x, &t_eval_cand, &t_eval_spec, &t_lm, &t_lm_tail, &t_pairs, &t_gn, &t_chol, &t_bs[0], &t_bs[1], &t_bs[2], &t_bs[3], &t_backsub, &t_reduce, &t_accept, &t_backup};
  }
};

// what the batched kernels cover: a window with Euclidean landmarks eliminated on the landmark side, optionally IMU factors (evaluated in
// the reprojection launch, assembled in the pair launch), the fused factorisation, nothing else
}  // namespace
bool batch_covers(bsgpu_ctx* c, const bsgpu_options& o) {
  if (!c->finalized) return false;
  if (!(o.linear_solver_type == BSGPU_LINEAR_AUTO || o.linear_solver_type == BSGPU_LINEAR_SCHUR_CHOLESKY)) return false;
  if (!c->dense_ok || c->use_graphs || !c->d_S || !c->h_scal_dev || !c->d_reduce_counter || c->n_reduce <= 0) return false;
  if (c->vis.n <= 0 || c->vis.n_lm <= 0 || c->vis.n_seg <= 0 || c->n_pose <= 0) return false;
  if (c->any_inactive || !c->marg.empty() || c->n_idp_lm > 0 || c->idp.n_lm > 0) return false;
  for (int t = 2; t < kNumInternal; ++t) if (c->small[t].n > 0 && t != BSGPU_F_IMU_DELTA && t != BSGPU_F_IMU_PRIOR) return false;
  if (c->n_sa_seg + c->n_asm_grp > 0) return false;
  if (c->n_upd_blocks <= 0 || c->upd_in_mcc) return false;
  if (!c->d_ftasks || !c->d_fsync || !c->d_tile_tot || !c->d_Winv) return false;
  if (c->plan.ftasks.size() > 65535) return false;   // (the window's tasks are the y dimension of the batched factorisation's grid)
  if (getenv("BSGPU_EVAL_MERGE") || getenv("BSGPU_SCALARS_EVENT") || getenv("BSGPU_UPDATE_SEPARATE")) return false;
  SmallGroupSet set;
  int taken = 0;
  (void)small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken);
  for (int i = taken; i < kNumInternal - 2; ++i) if (c->small_factorwise[2 + i].n > 0) return false;
  (void)small_mcc_first_set(c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, &set, &taken);
  for (int i = taken; i < kNumInternal - 2; ++i) if (c->small[2 + i].n > 0) return false;
  return true;
}
namespace {

int64_t g_stat_windows = 0, g_stat_rounds = 0;
BatchPlan g_plan;   // (one cached plan: a caller that alternates between sets of windows rebuilds)
std::mutex g_plan_mutex;

bool plan_matches(const BatchPlan& P, bsgpu_ctx* const* ctxs, int n, const bsgpu_options& o) {
  if ((int)P.ctxs.size() != n || P.jacobi != o.jacobi_scaling || P.lm_lo != o.min_lm_diagonal || P.lm_hi != o.max_lm_diagonal) return false;
  for (int i = 0; i < n; ++i) if (P.ctxs[i] != ctxs[i] || P.gens[i] != ctxs[i]->finalize_gen || P.xptr[i] != ctxs[i]->d_x) return false;
  return true;
}

// false: some window's plan is not covered after all (its back-substitution takes a form the batch does not launch)
bool build_plan(BatchPlan& P, bsgpu_ctx* const* ctxs, int n, const bsgpu_options& o) {
  P.release();
  if (P.device != ctxs[0]->device || !P.stream) {
    if (P.stream) (void)hipStreamDestroy(P.stream);
    P.stream = nullptr;
    P.device = ctxs[0]->device;
    if (hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); P.stream = nullptr; return false; }
  }
  if (!P.h_dyn && hipHostMalloc((void**)&P.h_dyn, sizeof(BatchDyn)) != hipSuccess) { (void)hipGetLastError(); P.h_dyn = nullptr; return false; }
  if (!P.d_dyn && hipMalloc((void**)&P.d_dyn, sizeof(BatchDyn)) != hipSuccess) { (void)hipGetLastError(); P.d_dyn = nullptr; return false; }
  P.jacobi = o.jacobi_scaling; P.lm_lo = o.min_lm_diagonal; P.lm_hi = o.max_lm_diagonal;
  P.bs_form.assign(n, -1);
  P.diag_in_chol.assign(n, 0);
  for (int w = 0; w < n; ++w) {
    bsgpu_ctx* c = ctxs[w];
    P.ctxs.push_back(c); P.gens.push_back(c->finalize_gen); P.xptr.push_back(c->d_x);
    {
      double* bk = nullptr;
      if (hipMalloc((void**)&bk, sizeof(double) * std::max<size_t>(1, c->h_x.size())) != hipSuccess) { (void)hipGetLastError(); return false; }
      P.dev_allocs.push_back(bk); P.backup.push_back(bk);
      batchargs_copy(P.t_backup, c->d_x, bk, (int64_t)c->h_x.size());
    }
    const SmallGroup& dl = c->small[BSGPU_F_IMU_DELTA];
    const SmallGroup& pr = c->small[BSGPU_F_IMU_PRIOR];
    // evaluation: residuals + Jacobians at x, cost only at the candidate, residuals + Jacobians at the candidate (ahead of the decision)
    batchargs_visual_imu_eval(P.t_eval_x, c->vis, dl, pr, c->d_x, c->d_cams, c->d_losses, c->vis.cost_part, c->d_small_part[BSGPU_F_IMU_DELTA], c->d_small_part[BSGPU_F_IMU_PRIOR]);
    batchargs_visual_imu_eval(P.t_eval_cand, c->vis, dl, pr, c->d_xcand, c->d_cams, c->d_losses, c->vis.cost_part_cand, c->d_small_part_cand[BSGPU_F_IMU_DELTA],
                              c->d_small_part_cand[BSGPU_F_IMU_PRIOR]);
    batchargs_visual_imu_eval(P.t_eval_spec, c->vis, dl, pr, c->d_xcand, c->d_cams, c->d_losses, c->vis.cost_part, c->d_small_part[BSGPU_F_IMU_DELTA], c->d_small_part[BSGPU_F_IMU_PRIOR]);
    // assembly
    ZeroStep zs;
    zs.S = c->d_S; zs.ld = c->npad; zs.tiles = c->d_touched; zs.n_tiles = c->n_touched;
    zs.a = c->d_grad; zs.na = c->n_pose; zs.b = c->d_hdiag; zs.nb = c->n_pose;
    zs.c = c->d_scal + SC_GRAD_MAX; zs.nc = 3;
    zs.radius_slot = c->d_scal + SC_RADIUS; zs.radius = 0.0;
    batchargs_landmark(P.t_lm, P.t_lm_tail, c->vis, c->n_pose, o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad, zs);
    SmallGroupSet set;
    int taken = 0;
    const int units = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken);
    batchargs_pairs(P.t_pairs, c->vis, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, units > 0 ? &set : nullptr, units);
    batchargs_grad_norms_pose_diag(P.t_gn, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_gpart, c->n_pose, c->d_S, c->npad,
                                   c->d_hdiag, o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_inat);
    // linear solve
    const DenseDev D{c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
                     c->d_bs_desc, c->d_chain_begin, c->d_chain_end, c->d_tile_sync, c->d_ftasks, c->d_fsync,
                     c->d_bs_desc_chain, c->d_rows_flat_chain, c->d_bs_upd, c->d_bs_upd_rows,
                     c->d_bs_chain_group, c->d_bs_grp_nchains, c->d_bs_grp_nitems, c->d_bs_items4, c->d_bs_tile_updated, c->d_bs_sync, c->d_scal, c->d_Winv, c->d_bs_order, c->d_tile_tot, 1};
    // (the LM diagonal and the gradient norms of a full step ride in the factorisation's launch when the window's plan has the tasks for them;
    //  radius and the step's flags are patched in per round, BatchDyn)
    // (kBatchDiagInCholMax: for few windows only)
    P.diag_in_chol[w] = n <= kBatchDiagInCholMax && c->plan.diag_tasks && c->plan.rider_tasks * 256 >= c->nb;
    LmDiag lmd;
    GradNormRide gnr;
    if (P.diag_in_chol[w]) {
      lmd.hdiag = c->d_hdiag; lmd.scale = c->d_scale; lmd.dcl = c->d_dcl; lmd.inat = c->d_inat; lmd.jacobi = o.jacobi_scaling;
      lmd.lm_lo = o.min_lm_diagonal; lmd.lm_hi = o.max_lm_diagonal;
      gnr.nb = c->nb; gnr.xoff = c->d_blk_xoff; gnr.toff = c->d_blk_toff; gnr.size = c->d_blk_size; gnr.manifold = c->d_blk_manifold; gnr.x = c->d_x; gnr.grad = c->d_grad;
      gnr.gpart = c->d_gpart;
    }
    const bool plain = !P.diag_in_chol[w] && c->d_ftasks_plain && c->d_tile_tot_plain;
    batchargs_chol_fused(P.t_chol, c->d_S, D.Lp, c->plan.npad, plain ? c->d_ftasks_plain : D.ftasks, plain ? c->n_ftasks_plain : (int)c->plan.ftasks.size(),
                         plain ? c->d_tile_tot_plain : D.tile_tot, D.nreal, D.Vinv, c->d_scal, D.fsync, D.Winv, D.rhs_rows, lmd, gnr);
    P.bs_form[w] = batchargs_backsolve(P.t_bs, c->plan, D, c->d_y, c->d_inat, c->n_pose, c->d_ytan, c->d_delta);
    if (P.bs_form[w] < 0) return false;
    // landmark back-substitution + model cost change + candidate
    int taken2 = 0;
    const int units2 = small_mcc_first_set(c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, &set, &taken2);
    UpdateRide up;
    up.n_blocks = c->n_upd_blocks; up.blocks = c->d_upd_blocks; up.xoff = c->d_blk_xoff; up.toff = c->d_blk_toff; up.size = c->d_blk_size;
    up.manifold = c->d_blk_manifold; up.lm_xoff = c->d_lm_xoff; up.x = c->d_x; up.x_cand = c->d_xcand; up.part = c->d_part_upd;
    batchargs_backsub_mcc(P.t_backsub, c->vis, c->n_pose, c->d_ytan, c->d_delta, c->vis.mcc_part, units2 > 0 ? &set : nullptr, units2, &up);
    batchargs_final_reduce(P.t_reduce, c->d_reduce, c->n_reduce, SC_GRAD_NORM2 + 1, c->d_scal, c->h_scal_dev, c->d_reduce_counter);
    batchargs_copy(P.t_accept, c->d_xcand, c->d_x, (int64_t)c->h_x.size());
  }
  for (BatchArgTable* t : P.tables()) {
    if (t->host.empty()) continue;
    void* d = nullptr;
    if (hipMalloc(&d, t->host.size()) != hipSuccess) { (void)hipGetLastError(); return false; }
    P.dev_allocs.push_back(d);
    if (hipMemcpy(d, t->host.data(), t->host.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return false; }
    t->dev = d;
  }
  return true;
}

// one set of launches for the steps the windows of `L` requested; `act`: indices into L of the windows still iterating
void enqueue_round(BatchPlan& P, std::vector<LmWindow>& L, const std::vector<int>& act) {
  BatchDyn& d = *P.h_dyn;
  for (int q = 0; q < BL_NUM; ++q) d.n[q] = 0;
  for (int w : act) {
    LmWindow& lw = L[w];
    bsgpu_ctx* c = lw.c;
    d.idx[BL_ALL][d.n[BL_ALL]++] = w;
    if (!lw.grad_only) {
      d.idx[BL_FULL][d.n[BL_FULL]++] = w;
      const int f = P.bs_form[w];   // 0, 1: single-launch form (shallow, deep); 2, 3: one chain
      const int list = (f < 2 ? BL_BS_FUSED : BL_BS_CHAIN) + (f & 1);
      d.idx[list][d.n[list]++] = w;
    }
    if (lw.grad_only || !P.diag_in_chol[w]) d.idx[BL_DIAG][d.n[BL_DIAG]++] = w;
    if (lw.kind == K_ACCEPT) d.idx[BL_ACC][d.n[BL_ACC]++] = w; else d.idx[BL_REJ][d.n[BL_REJ]++] = w;   // (first steps evaluate at x like rejected ones)
    d.radius[w] = lw.radius;
    d.first[w] = lw.kind == K_FIRST ? 1 : 0;
    d.new_J[w] = lw.kind != K_REJECT ? 1 : 0;
    d.grad_only[w] = lw.grad_only ? 1 : 0;
    c->reduce_seq += 1.0;
    d.seq[w] = c->reduce_seq;
  }
  hipStream_t s = P.stream;
  (void)hipMemcpyAsync(P.d_dyn, P.h_dyn, sizeof(BatchDyn), hipMemcpyHostToDevice, s);
  const BatchDyn* dd = P.d_dyn;
  launch_copy_batch(s, P.t_accept, dd, BL_ACC, d.n[BL_ACC]);                              // x <- x_cand
  launch_visual_imu_eval_batch(s, P.t_eval_x, dd, BL_REJ, d.n[BL_REJ], true);              // Jacobians at x (accepted windows have them: evaluated ahead)
  launch_landmark_batch(s, P.t_lm, P.t_lm_tail, dd, BL_ALL, d.n[BL_ALL]);
  launch_pairs_batch(s, P.t_pairs, dd, BL_ALL, d.n[BL_ALL]);
  launch_grad_norms_pose_diag_batch(s, P.t_gn, dd, BL_DIAG, d.n[BL_DIAG]);   // (the others: in the factorisation's launch)
  if (d.n[BL_FULL] > 0) {
    launch_chol_fused_batch(s, P.t_chol, dd, BL_FULL, d.n[BL_FULL]);
    const int nf[4] = {d.n[BL_BS_FUSED], d.n[BL_BS_FUSED + 1], d.n[BL_BS_CHAIN], d.n[BL_BS_CHAIN + 1]};
    launch_backsolve_batch(s, P.t_bs, dd, nf);
    launch_backsub_mcc_batch(s, P.t_backsub, dd, BL_FULL, d.n[BL_FULL]);
    launch_visual_imu_eval_batch(s, P.t_eval_cand, dd, BL_FULL, d.n[BL_FULL], false);
  }
  launch_final_reduce_batch(s, P.t_reduce, dd, BL_ALL, d.n[BL_ALL]);
  if (d.n[BL_FULL] > 0) launch_visual_imu_eval_batch(s, P.t_eval_spec, dd, BL_FULL, d.n[BL_FULL], true);   // ahead of the decisions, under the host round trip
}

int wait_round(BatchPlan& P, std::vector<LmWindow>& L, const std::vector<int>& act) {
  if (hipGetLastError() != hipSuccess) return BSGPU_ERR_DEVICE;
  const auto t0 = std::chrono::steady_clock::now();
  for (int w : act) {
    bsgpu_ctx* c = L[w].c;
    const volatile double* stamp = &c->h_scal[SC_SEQ];
    long spins = 0;
    while (__atomic_load_n(reinterpret_cast<const volatile uint64_t*>(stamp), __ATOMIC_ACQUIRE) != *reinterpret_cast<const uint64_t*>(&c->reduce_seq)) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) {
        if (hipStreamSynchronize(P.stream) != hipSuccess) return BSGPU_ERR_DEVICE;
        if (*stamp != c->reduce_seq) return BSGPU_ERR_DEVICE;
        break;
      }
    }
  }
  return BSGPU_OK;
}

}  // namespace

// solves windows idx[0..m) of the call (all covered by the batched kernels, all on one device); rc[i] per window.  Returns false when
// the batch could not be set up (the caller then solves them one thread per window).
bool solve_batched(bsgpu_ctx* const* ctxs, const int* idx, int m, const bsgpu_options* o, int options_stride, bsgpu_summary* s, int* rc) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  std::vector<bsgpu_ctx*> cs(m);
  for (int i = 0; i < m; ++i) cs[i] = ctxs[idx[i]];
  const bsgpu_options& o0 = o[options_stride ? idx[0] : 0];
  if (hipSetDevice(cs[0]->device) != hipSuccess) return false;
  BatchPlan& P = g_plan;
  if (!plan_matches(P, cs.data(), m, o0) && !build_plan(P, cs.data(), m, o0)) { P.release(); return false; }
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  std::vector<LmWindow> L(m);
  std::vector<int> act;
  for (int i = 0; i < m; ++i) {
    L[i].c = cs[i]; L[i].o = &o[options_stride ? idx[i] : 0]; L[i].sum = &s[idx[i]];
    // (what the window's own stream still holds — bsgpu_reset_values is asynchronous — comes first)
    if (hipStreamQuery(cs[i]->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(cs[i]->stream); }
    L[i].start();
    cs[i]->use_pcg = false; cs[i]->use_spcg = false; cs[i]->spec_J = false;
    act.push_back(i);
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
  (void)hipEventRecord(ev0, P.stream);
  {   // the values the call starts from (BL_ALL of a block that only this launch reads)
    BatchDyn& d = *P.h_dyn;
    d.n[BL_ALL] = m;
    for (int i = 0; i < m; ++i) d.idx[BL_ALL][i] = i;
    (void)hipMemcpyAsync(P.d_dyn, P.h_dyn, sizeof(BatchDyn), hipMemcpyHostToDevice, P.stream);
    launch_copy_batch(P.stream, P.t_backup, P.d_dyn, BL_ALL, m);
    (void)hipStreamSynchronize(P.stream);   // (h_dyn is rewritten for the first round)
  }
  bool first = true;
  int err = BSGPU_OK;
  g_stat_windows += m;
  while (!act.empty()) {
    ++g_stat_rounds;
    enqueue_round(P, L, act);
    err = wait_round(P, L, act);
    if (err != BSGPU_OK) break;
    std::vector<int> next;
    for (int w : act) {
      if (first) L[w].begin(); else L[w].advance();
      if (!L[w].done) next.push_back(w);
    }
    first = false;
    act.swap(next);
  }
  (void)hipEventRecord(ev1, P.stream);
  (void)hipEventSynchronize(ev1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  const double total = std::chrono::duration<double>(clk::now() - t_start).count();
  for (int i = 0; i < m; ++i) {
    bsgpu_ctx* c = cs[i];
    c->scal_mirrored = false; c->ev_reduce_pending = false; c->seq_pending = false; c->spec_J = false; c->pre_cleared = false;
    if (err != BSGPU_OK) { rc[idx[i]] = fail(c, err, "solve_batch: device error in the batched step"); continue; }
    L[i].sum->device_time_in_seconds = ms * 1e-3;
    L[i].sum->total_time_in_seconds = total;
    rc[idx[i]] = BSGPU_OK;
    if (L[i].timed_out) {
      // a wait inside a single-launch kernel timed out (the device is shared): this window again, alone, on the launch-per-step path
      c->d_ftasks = nullptr;
      (void)hipMemcpy(c->d_x, P.backup[i], sizeof(double) * c->h_x.size(), hipMemcpyDeviceToDevice);
      rc[idx[i]] = solve(c, *L[i].o, *L[i].sum);
    }
  }
  return true;
}

void batch_stats(int64_t* windows, int64_t* rounds) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  if (windows) *windows = g_stat_windows;
  if (rounds) *rounds = g_stat_rounds;
}

}  // namespace bsg
