// bsgpu_solve_batch: several windows advanced by ONE set of launches per LM iteration.
//
// Reference: the unit of independence is the window — the local fixed-lag smoother, the global mapper and the submap refinements run
// side by side (beam_slam_launch/launch/vio.launch:19-30), submap refinement is a serial loop of solves over fresh graphs of
// scan-registration relative-pose factors (bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115,
// scan_to_map_registration.cpp:74-78), the pose-graph optimisation another (submap_pose_graph_optimization.cpp:22-150), the lidar
// odometry a window of tens of key frames at 25 Hz (lio.yaml:2), the VIO one at a few tens of key frames (vio.yaml:3,56).  A window of
// that size is a handful of workgroups per kernel on a 256-CU chip and its LM iteration is a chain of ~10 dependent launches of a few
// microseconds of work each: alone it uses a few per cent of the device, and one host thread + stream per window does not overlap them
// better than the hardware queues do.
//
// Here every kernel of the step has a `_batch` form (k_*.hip): blockIdx.y picks a window from a list, and the kernel's arguments for
// that window come from a table in device memory — entry w is exactly what window w's lone launch passes (a zero grid where the window
// has no such launch: windows of DIFFERENT kinds share the call), so every window's tables, partial sums and reductions are laid out as
// in its lone solve and its iterations come out the same.  What changes per iteration (which windows still iterate, whose step was
// accepted or rejected, the radii) is one small block uploaded once per iteration (BatchDyn).  The LM decisions stay on the host, per
// window: lm_state.h, the same state machine bsgpu_solve.cpp drives for a lone window.  Windows that converge drop out of the lists.
//
// Covered: visual(-inertial) windows with Euclidean landmarks eliminated on the landmark side; lidar-inertial windows (relative-pose
// factors with or without extrinsics + IMU factors); pose graphs on the dense path; pose-only factors of any kind riding along (absolute
// poses, vector priors, gravity, reprojection factors with a kept landmark); inverse-depth landmarks eliminated on the landmark side (k_idp.hip); a window's dense marginal prior (the state after a slide
// with true marginalisation, fixed_lag_smoother.cpp:269-272); windows with constant blocks (the fixed cost is taken once, before the
// first round).  Not covered (a thread per window, in the same call): block-sparse PCG, more than one dense
// prior, hipGraph replay, windows on another device.
#include <list>
#include <memory>
#include <thread>

#include "bsgpu_ctx.h"
#include "lm_state.h"

namespace bsg {
namespace {

// the environment switches that change which launches a lone step is made of (the batch mirrors the default set only): read once
bool env_step_variants() {
  static const bool v = getenv("BSGPU_EVAL_MERGE") || getenv("BSGPU_SCALARS_EVENT") || getenv("BSGPU_UPDATE_SEPARATE") || getenv("BSGPU_CLEAR_AT_START");
  return v;
}

// which launches a window's step is made of — bsgpu_solve.cpp eval_all() / assemble() / linear_solve_and_candidate(), decided once
struct WinShape {
  bool vis = false;        // landmark + pair + landmark back-substitution launches (the step's clearing rides in the landmark launch)
  bool imu_pair = false;   // IMU factors present
  int rel_t = -1;          // the relative-pose group whose evaluation launch carries the IMU factors (no visual launch), or -1
  int n_set = 0;           // pose-only groups evaluated by the generic launch
  int set_t[kNumInternal];
  bool zero_in_mcc = false;   // the next step's clearing rides in the pose-only model-cost launch
};
bool shape_of(const bsgpu_ctx* c, WinShape& w) {
  w = WinShape();
  w.vis = c->vis.n > 0;
  w.imu_pair = c->small[BSGPU_F_IMU_DELTA].n + c->small[BSGPU_F_IMU_PRIOR].n > 0;
  if (!w.vis)
    for (int t : {(int)BSGPU_F_RELPOSE_EXT, (int)BSGPU_F_RELPOSE}) if (w.rel_t < 0 && c->small[t].n > 0) w.rel_t = t;
  const bool imu_carried = w.imu_pair && (w.vis || w.rel_t >= 0);
  for (int t = 2; t < kNumInternal; ++t) {
    if (!c->small[t].n || t == w.rel_t) continue;
    if (imu_carried && (t == BSGPU_F_IMU_DELTA || t == BSGPU_F_IMU_PRIOR)) continue;
    w.set_t[w.n_set++] = t;
  }
  w.zero_in_mcc = c->upd_in_mcc && c->h_scal_dev != nullptr;
  return w.n_set <= 8;
}

}  // namespace

// what the batched kernels cover (the table of contents above)
bool batch_covers(bsgpu_ctx* c, const bsgpu_options& o) {
  if (!c->finalized) return false;
  if (!(o.linear_solver_type == BSGPU_LINEAR_AUTO || o.linear_solver_type == BSGPU_LINEAR_SCHUR_CHOLESKY)) return false;
  if (!c->dense_ok || c->use_graphs || !c->d_S || !c->h_scal_dev || !c->d_reduce_counter || c->n_reduce <= 0 || c->n_pose <= 0) return false;
  {   // at most ONE dense prior with free blocks (a window after a slide with true marginalisation), narrow enough for the one-launch evaluation
    int n_act = 0;
    for (const auto& mc : c->marg) if (mc.active) { ++n_act; if (mc.dev.cols > 1024 || mc.dev.nblk > 1024) return false; }
    if (n_act > 1) return false;
  }
  if (c->vis.n > 0) {
    if (c->vis.n_lm <= 0 || (c->vis.n_seg <= 0 && c->vis.n_band_units <= 0) || c->n_upd_blocks <= 0 || c->upd_in_mcc) return false;
  } else if (!c->upd_in_mcc) return false;
  if (!c->d_ftasks || !c->d_fsync || !c->d_tile_tot || !c->d_Winv) return false;
  if (c->plan.ftasks.size() > 65535) return false;   // (the window's tasks are the y dimension of the batched factorisation's grid)
  if (env_step_variants()) return false;
  WinShape w;
  if (!shape_of(c, w)) return false;
  // the pose-only groups that ride in no other launch must fit ONE set each (assembly one workgroup per factor; model-cost terms)
  SmallGroupSet set;
  int taken = 0, units = 0, taken2 = 0, units2 = 0;
  if (c->vis.n_seg > 0 || c->vis.n_band_units > 0) units = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken);
  if (units == 0) taken = 0;
  if (units == 0 && c->n_sa_seg + c->n_asm_grp > 0) units2 = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken2);
  if (units2 == 0) taken2 = 0;
  int left = 0;
  for (int i = taken + taken2; i < kNumInternal - 2; ++i) if (c->small_factorwise[2 + i].n > 0) ++left;
  if (left > kSetMax) return false;
  taken = 0;
  if (c->vis.n > 0) { (void)small_mcc_first_set(c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, &set, &taken); }
  left = 0;
  for (int i = taken; i < kNumInternal - 2; ++i) if (c->small[2 + i].n > 0) ++left;
  if (left > kSetMax) return false;
  return true;
}

namespace {

int64_t g_stat_windows = 0, g_stat_rounds = 0;
std::mutex g_stat_mutex;

// the argument tables of one set of windows, built once per (contexts, finalize generations, options that enter the tables)
struct BatchPlan {
  std::vector<bsgpu_ctx*> ctxs;
  std::vector<uint64_t> gens;        // process-unique finalize stamps (a context re-created at the same address never matches)
  std::vector<const double*> xptr;   // each window's current-point buffer when the tables were built (a lone solve swaps it with the candidate's)
  std::vector<double*> backup;       // the windows' values when the call began (a window whose single-launch kernels time out is solved again, alone)
  std::vector<WinShape> shape;
  int jacobi = -1;
  double lm_lo = 0.0, lm_hi = 0.0;
  int device = -1;
  hipStream_t stream = nullptr;
  BatchDyn* h_dyn = nullptr;   // pinned
  BatchDyn* d_dyn = nullptr;
  std::vector<void*> dev_allocs;
  // [3]: at x (rejected / first steps) | cost only at the candidate | residuals + Jacobians at the candidate, ahead of the decision
  BatchArgTable t_eval_vis[3], t_eval_rel[3], t_eval_set[3], t_eval_marg[3], t_marg_asm, t_marg_mcc, t_idp_lm, t_idp_view, t_idp_pairs, t_idp_backsub;
  BatchArgTable t_lm, t_lm_tail, t_zero, t_pairs, t_pairs_band, t_asm_set, t_asm_seg, t_gn, t_chol, t_bs[4], t_backsub, t_small_mcc, t_reduce, t_accept, t_backup;
  std::vector<int> bs_form;
  size_t max_tasks = 0;
  // The candidate evaluated ONCE, with Jacobians, into the candidate's cost partials — no cost-only pass, the round's reduction behind that
  // evaluation instead of in front of it: what a round of LARGE windows spends on the cost-only pass (8 x 10 us of C2's eight) is more than the
  // host's round trip costs once the evaluation ahead no longer covers it; small windows keep the two passes (enqueue_round).
  bool one_pass = false;
  std::vector<BatchArgTable*> tables() {
    std::vector<BatchArgTable*> v = {&t_lm, &t_lm_tail, &t_zero, &t_pairs, &t_pairs_band, &t_asm_set, &t_asm_seg, &t_gn, &t_chol, &t_bs[0], &t_bs[1], &t_bs[2], &t_bs[3],
                                     &t_backsub, &t_small_mcc, &t_reduce, &t_accept, &t_backup};
    for (int i = 0; i < 3; ++i) { v.push_back(&t_eval_vis[i]); v.push_back(&t_eval_rel[i]); v.push_back(&t_eval_set[i]); v.push_back(&t_eval_marg[i]); }
    v.push_back(&t_marg_asm); v.push_back(&t_marg_mcc); v.push_back(&t_idp_lm); v.push_back(&t_idp_view); v.push_back(&t_idp_pairs); v.push_back(&t_idp_backsub);
    return v;
  }
  bool names(const bsgpu_ctx* c) const { for (const bsgpu_ctx* x : ctxs) if (x == c) return true; return false; }
  ~BatchPlan() {
    if (device >= 0) (void)hipSetDevice(device);
    for (void* p : dev_allocs) (void)hipFree(p);
    if (d_dyn) (void)hipFree(d_dyn);
    if (h_dyn) (void)hipHostFree(h_dyn);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// One small cache of plans PER DEVICE, each behind its own mutex: threads driving different devices from one process do not serialise
// or evict each other's tables; a caller alternating between a few sets of windows on one device (chunks of a large call) finds them.
constexpr size_t kPlansPerDevice = 4;
struct DevicePlans {
  std::mutex m;
  std::list<std::unique_ptr<BatchPlan>> plans;   // most recently used first
};
std::mutex g_reg_mutex;
std::map<int, std::unique_ptr<DevicePlans>> g_devices;

DevicePlans& device_plans(int device) {
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  auto& p = g_devices[device];
  if (!p) p.reset(new DevicePlans());
  return *p;
}

bool plan_matches(const BatchPlan& P, bsgpu_ctx* const* ctxs, int n, const bsgpu_options& o) {
  if ((int)P.ctxs.size() != n || P.jacobi != o.jacobi_scaling || P.lm_lo != o.min_lm_diagonal || P.lm_hi != o.max_lm_diagonal) return false;
  for (int i = 0; i < n; ++i) if (P.ctxs[i] != ctxs[i] || P.gens[i] != ctxs[i]->finalize_gen || P.xptr[i] != ctxs[i]->d_x) return false;
  return true;
}

// false: some window's plan is not covered after all (its back-substitution takes a form the batch does not launch), or no memory
// The camera-pair segments of a window are cut for ONE window on the device (pair_chunk: 64 entries per wave on a window of the
// reference's size — a wave's length is what a lone window waits for).  Thirty-two such windows side by side are bound by the launch's
// work instead: the same entries in segments of up to kPairChunk (every fourth boundary of a pair's fine list), built once per finalize
// on first use.
static bool coarse_pair_segments(bsgpu_ctx* c) {
  Visual& v = c->vis;
  if (v.n_seg <= 0 || v.n_seg_c > 0) return true;
  if (pair_chunk((size_t)v.n_ent) >= kPairChunk) { v.n_seg_c = v.n_seg; v.seg_ci_c = v.seg_ci; v.seg_cj_c = v.seg_cj; v.seg_start_c = v.seg_start; return true; }
  std::vector<int> st((size_t)v.n_seg + 1), ci(v.n_seg), cj(v.n_seg);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  if (hipMemcpy(st.data(), v.seg_start, sizeof(int) * st.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(ci.data(), v.seg_ci, sizeof(int) * ci.size(), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(cj.data(), v.seg_cj, sizeof(int) * cj.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
  std::vector<int> st_c, ci_c, cj_c;
  for (int sgi = 0; sgi < v.n_seg; ++sgi) {
    const bool same_pair = !ci_c.empty() && ci_c.back() == ci[sgi] && cj_c.back() == cj[sgi];
    if (!same_pair || st[sgi] - st_c.back() >= kPairChunk) { st_c.push_back(st[sgi]); ci_c.push_back(ci[sgi]); cj_c.push_back(cj[sgi]); }
  }
  st_c.push_back(st[v.n_seg]);
  v.seg_start_c = c->upload(st_c); v.seg_ci_c = c->upload(ci_c); v.seg_cj_c = c->upload(cj_c);
  if (!v.seg_start_c || !v.seg_ci_c || !v.seg_cj_c) { v.seg_start_c = v.seg_ci_c = v.seg_cj_c = nullptr; return false; }
  v.n_seg_c = (int)ci_c.size();
  return true;
}
bool build_plan(BatchPlan& P, bsgpu_ctx* const* ctxs, int n, const bsgpu_options& o) {
  P.device = ctxs[0]->device;
  if (hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); P.stream = nullptr; return false; }
  if (hipHostMalloc((void**)&P.h_dyn, sizeof(BatchDyn), hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); P.h_dyn = nullptr; return false; }
  if (hipMalloc((void**)&P.d_dyn, sizeof(BatchDyn)) != hipSuccess) { (void)hipGetLastError(); P.d_dyn = nullptr; return false; }
  P.jacobi = o.jacobi_scaling; P.lm_lo = o.min_lm_diagonal; P.lm_hi = o.max_lm_diagonal;
  P.bs_form.assign(n, -1);
  P.shape.resize(n);
  size_t tasks_per_round = 0, vis_factors = 0;
  for (int w = 0; w < n; ++w) tasks_per_round += ctxs[w]->plan.ftasks.size();
  static const char* bulk_env = getenv("BSGPU_BATCH_BULK");   // (0: never, 1: always)
  const bool bulk_lists = bulk_env ? atoi(bulk_env) != 0 : tasks_per_round >= 2048;
  for (int w = 0; w < n; ++w) {
    bsgpu_ctx* c = ctxs[w];
    P.ctxs.push_back(c); P.gens.push_back(c->finalize_gen); P.xptr.push_back(c->d_x);
    P.max_tasks = std::max(P.max_tasks, c->plan.ftasks.size());
    vis_factors += (size_t)c->vis.n;
    WinShape& sh = P.shape[w];
    if (!shape_of(c, sh)) return false;
    {
      double* bk = nullptr;
      if (hipMalloc((void**)&bk, sizeof(double) * std::max<size_t>(1, c->h_x.size())) != hipSuccess) { (void)hipGetLastError(); return false; }
      P.dev_allocs.push_back(bk); P.backup.push_back(bk);
      batchargs_copy(P.t_backup, c->d_x, bk, (int64_t)c->h_x.size());
    }
    const SmallGroup none;
    const bsgpu_ctx::MargCtx* mg = nullptr;   // the window's dense prior (batch_covers: at most one with free blocks)
    for (const auto& mc : c->marg) if (mc.active) mg = &mc;
    const SmallGroup& dl = c->small[BSGPU_F_IMU_DELTA];
    const SmallGroup& pr = c->small[BSGPU_F_IMU_PRIOR];
    // ---- evaluation (eval_all): the point and the cost-partial arrays of the three passes
    const double* xs[3] = {c->d_x, c->d_xcand, c->d_xcand};
    for (int v = 0; v < 3; ++v) {
      const bool cand = v == 1;
      double* const* part = cand ? c->d_small_part_cand : c->d_small_part;
      double* vis_part = cand ? c->vis.cost_part_cand : c->vis.cost_part;
      // (no visual factors: empty groups and n = 0 give the entry a zero grid)
      batchargs_visual_imu_eval(P.t_eval_vis[v], c->vis, sh.vis ? dl : none, sh.vis ? pr : none, xs[v], c->d_cams, c->d_losses, vis_part, part[BSGPU_F_IMU_DELTA], part[BSGPU_F_IMU_PRIOR]);
      batchargs_relpose_imu_eval(P.t_eval_rel[v], sh.rel_t >= 0 ? &c->small[sh.rel_t] : nullptr, (sh.rel_t >= 0 && sh.imu_pair) ? dl : none, (sh.rel_t >= 0 && sh.imu_pair) ? pr : none,
                                 xs[v], c->d_losses, sh.rel_t >= 0 ? part[sh.rel_t] : nullptr, part[BSGPU_F_IMU_DELTA], part[BSGPU_F_IMU_PRIOR]);
      SmallGroup gs[kNumInternal];
      double* ps[kNumInternal];
      for (int i = 0; i < sh.n_set; ++i) { gs[i] = c->small[sh.set_t[i]]; ps[i] = part[sh.set_t[i]]; }
      if (!batchargs_small_eval_set(P.t_eval_set[v], gs, ps, sh.n_set, xs[v], c->d_losses)) return false;
      if (!batchargs_marg_eval(P.t_eval_marg[v], mg ? &mg->dev : nullptr, xs[v], mg ? (cand ? mg->part_cand : mg->part) : nullptr)) return false;
    }
    // ---- assembly (assemble())
    ZeroStep zs;
    zs.S = c->d_S; zs.ld = c->npad; zs.tiles = c->d_touched; zs.n_tiles = c->n_touched;
    zs.a = c->d_grad; zs.na = c->n_pose; zs.b = c->d_hdiag; zs.nb = c->n_pose;
    zs.c = c->d_scal + SC_GRAD_MAX; zs.nc = 3;
    zs.radius_slot = c->d_scal + SC_RADIUS; zs.radius = 0.0;
    batchargs_landmark(P.t_lm, P.t_lm_tail, c->vis, c->n_pose, o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad, zs);
    batchargs_zero_tiles_multi(P.t_zero, sh.vis ? nullptr : &zs);
    // (inverse-depth landmarks: their scalar elimination + the view pairs, k_idp.hip — after the clearing)
    batchargs_idp_landmark(P.t_idp_lm, P.t_idp_view, c->idp, c->small[BSGPU_F_IDP_REPROJ], o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad);
    batchargs_idp_pairs(P.t_idp_pairs, c->idp, c->small[BSGPU_F_IDP_REPROJ], c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos);
    batchargs_idp_backsub(P.t_idp_backsub, c->idp, c->d_ytan, c->d_delta);
    SmallGroupSet set, set2;
    int taken = 0, units = 0, taken2 = 0, units2 = 0;
    const bool band = c->vis.n_band_units > 0;
    if (c->vis.n_seg > 0 || band) units = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken);
    if (units == 0) taken = 0;
    batchargs_pairs_band(P.t_pairs_band, c->vis, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, (units > 0 && band) ? &set : nullptr, band ? units : 0);
    (void)coarse_pair_segments(c);   // (on failure the fine list stays: correct, only slower side by side)
    Visual vis_b = c->vis;   // (the pair launch of a batch walks the coarse segments)
    if (c->vis.n_seg_c > 0) { vis_b.n_seg = c->vis.n_seg_c; vis_b.seg_ci = c->vis.seg_ci_c; vis_b.seg_cj = c->vis.seg_cj_c; vis_b.seg_start = c->vis.seg_start_c; }
    batchargs_pairs(P.t_pairs, vis_b, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, (units > 0 && !band) ? &set : nullptr, band ? 0 : units);
    if (units == 0 && c->n_sa_seg + c->n_asm_grp > 0) units2 = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set2, &taken2);
    if (units2 == 0) taken2 = 0;
    if (!batchargs_small_assemble_set(P.t_asm_set, c->small_factorwise + 2 + taken + taken2, kNumInternal - 2 - taken - taken2, c->d_S, c->npad, c->plan.rhs_row, c->d_grad,
                                      c->d_hdiag, c->d_dpos)) return false;
    batchargs_small_assemble_seg(P.t_asm_seg, c->d_small_groups, c->n_sa_seg, c->d_sa_seg_start, c->d_sa_seg_ra, c->d_sa_seg_rb, c->d_sa_contrib, c->d_S, c->npad, c->plan.rhs_row,
                                 c->d_grad, c->d_hdiag, c->d_dpos, units2 > 0 ? &set2 : nullptr, units2, c->n_asm_grp, c->d_asm_grp);
    batchargs_marg_assemble(P.t_marg_asm, mg ? &mg->dev : nullptr, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos);
    batchargs_marg_mcc(P.t_marg_mcc, mg ? &mg->dev : nullptr, c->d_delta, mg ? mg->part_mcc : nullptr);
    batchargs_grad_norms_pose_diag(P.t_gn, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_gpart, c->n_pose, c->d_S, c->npad,
                                   c->d_hdiag, o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_inat);
    // ---- linear solve: the task list WITHOUT the LM-diagonal / rider tasks (a batch is bound by the number of its LDS-heavy, one-per-CU
    // factorisation workgroups; the separate launch's 256-thread workgroups are the cheaper form: 8 windows of C2 5 350 against 5 150 LM it/s,
    // 32 windows of 20 KF x 500 99 100 against 91 700 — round 4)
    const DenseDev D{c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
                     c->d_bs_desc, c->d_chain_begin, c->d_chain_end, c->d_tile_sync, c->d_ftasks, c->d_fsync,
                     c->d_bs_desc_chain, c->d_rows_flat_chain, c->d_bs_upd, c->d_bs_upd_rows,
                     c->d_bs_chain_group, c->d_bs_grp_nchains, c->d_bs_grp_nitems, c->d_bs_items4, c->d_bs_tile_updated, c->d_bs_sync, c->d_scal, c->d_Winv, c->d_bs_order, c->d_tile_tot, 1};
    // (a call whose factorisations are bound by the number of their workgroups — each holds a CU — takes the lists without the K-chunks of
    //  §3.2f, which shorten ONE window's critical path at the price of four to five workgroups per dealt-out update: 16 pose graphs of 200
    //  poses + 6-10 %, 8 C2 windows + 2 %, 32 windows of 20 KF x 500 no difference — so: from 2 048 tasks per round on)
    const bool bulk = bulk_lists && c->d_ftasks_bulk && c->d_tile_tot_bulk;
    const bool plain = c->d_ftasks_plain && c->d_tile_tot_plain;
    const FusedTask* tl = bulk ? c->d_ftasks_bulk : plain ? c->d_ftasks_plain : D.ftasks;
    int ntl = bulk ? c->n_ftasks_bulk : plain ? c->n_ftasks_plain : (int)c->plan.ftasks.size();
    const int* tt = bulk ? c->d_tile_tot_bulk : plain ? c->d_tile_tot_plain : D.tile_tot;
    // ... and, where that list is the one the row segments were made from, the list with the segments: an update task is ~7 us of a compute unit for
    // 1.3 - 2 us of products, a segment's further updates ~2 (dense_plan.h build_row_segments; launches without turns only; BSGPU_CHOL_ROWS=0: never)
    static const bool rows_off = (getenv("BSGPU_CHOL_ROWS") && atoi(getenv("BSGPU_CHOL_ROWS")) == 0) || (getenv("BSGPU_CHOL_NOTURN") && atoi(getenv("BSGPU_CHOL_NOTURN")) == 0);
    const int src_now = bulk ? 2 : plain ? 1 : 0;
    if (bulk_lists && !rows_off && c->d_ftasks_rows && c->plan.frows_src == src_now) { tl = c->d_ftasks_rows; ntl = c->n_ftasks_rows; }
    batchargs_chol_fused(P.t_chol, c->d_S, D.Lp, c->plan.npad, tl, ntl, tt, D.nreal, D.Vinv, c->d_scal, D.fsync, D.Winv, D.rhs_rows, LmDiag(), GradNormRide(), /*diag_tasks_in_list=*/!bulk && !plain && c->plan.diag_tasks);
    P.bs_form[w] = batchargs_backsolve(P.t_bs, c->plan, D, c->d_y, c->d_inat, c->n_pose, c->d_ytan, c->d_delta);
    if (P.bs_form[w] < 0) return false;
    // ---- landmark back-substitution + model cost change + candidate (linear_solve_and_candidate())
    int taken3 = 0, units3 = 0;
    if (backsub_mcc_groups(c->vis) > 0) units3 = small_mcc_first_set(c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, &set, &taken3);
    if (units3 == 0) taken3 = 0;
    UpdateRide up;
    if (c->n_upd_blocks > 0) {
      up.n_blocks = c->n_upd_blocks; up.blocks = c->d_upd_blocks; up.xoff = c->d_blk_xoff; up.toff = c->d_blk_toff; up.size = c->d_blk_size;
      up.manifold = c->d_blk_manifold; up.lm_xoff = c->d_lm_xoff; up.x = c->d_x; up.x_cand = c->d_xcand; up.part = c->d_part_upd;
    }
    batchargs_backsub_mcc(P.t_backsub, c->vis, c->n_pose, c->d_ytan, c->d_delta, c->vis.mcc_part, units3 > 0 ? &set : nullptr, units3, c->n_upd_blocks > 0 ? &up : nullptr);
    UpdateRide all;
    if (c->upd_in_mcc) {   // (no Euclidean landmarks: every block's candidate rides in the pose-only groups' launch)
      all.n_blocks = c->nb; all.blocks = nullptr; all.xoff = c->d_blk_xoff; all.toff = c->d_blk_toff; all.size = c->d_blk_size;
      all.manifold = c->d_blk_manifold; all.x = c->d_x; all.x_cand = c->d_xcand; all.part = c->d_part_upd;
    }
    ZeroStep zn;
    if (sh.zero_in_mcc) {
      zn.S = c->d_S; zn.ld = c->npad; zn.tiles = c->d_touched; zn.n_tiles = c->n_touched;
      zn.a = c->d_grad; zn.na = c->n_pose; zn.b = c->d_hdiag; zn.nb = c->n_pose;
    }
    if (!batchargs_small_mcc(P.t_small_mcc, c->small + 2 + taken3, c->d_small_part_mcc + 2 + taken3, kNumInternal - 2 - taken3, c->d_delta, c->upd_in_mcc ? &all : nullptr,
                             sh.zero_in_mcc ? &zn : nullptr)) return false;
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
  {
    // (BSGPU_BATCH_ONE_PASS=0|1 forces it; by itself: from a million reprojection factors per round on — eight C2 windows 3.2 M, thirty-two windows of
    //  the reference's size 0.13 M)
    static const int env = getenv("BSGPU_BATCH_ONE_PASS") ? atoi(getenv("BSGPU_BATCH_ONE_PASS")) : -1;
    P.one_pass = env >= 0 ? env != 0 : vis_factors >= 1000000;
  }
  return true;
}

// one window of the batch: its trust-region state and what the rounds have to remember about its device state
struct BatchWin {
  bsgpu_ctx* c = nullptr;
  LmState lm;
  bool cleared = false;     // the reduced system, gradient and diagonal were cleared at the end of the window's previous step
  bool timed_out = false;
};

void evals(BatchPlan& P, int v, const BatchDyn* dd, int list, int n, bool with_J) {
  launch_visual_imu_eval_batch(P.stream, P.t_eval_vis[v], dd, list, n, with_J);
  launch_relpose_imu_eval_batch(P.stream, P.t_eval_rel[v], dd, list, n, with_J);
  launch_small_eval_set_batch(P.stream, P.t_eval_set[v], dd, list, n, with_J);
  launch_marg_eval_batch(P.stream, P.t_eval_marg[v], dd, list, n, with_J);
}

// one set of launches for the steps the windows of `L` requested; `act`: indices into L of the windows still iterating
void enqueue_round(BatchPlan& P, std::vector<BatchWin>& L, const std::vector<int>& act) {
  BatchDyn& d = *P.h_dyn;
  for (int q = 0; q < BL_NUM; ++q) d.n[q] = 0;
  for (int w : act) {
    BatchWin& bw = L[w];
    const LmState& lw = bw.lm;
    bsgpu_ctx* c = bw.c;
    d.idx[BL_ALL][d.n[BL_ALL]++] = w;
    d.idx[BL_DIAG][d.n[BL_DIAG]++] = w;
    if (!lw.grad_only) {
      d.idx[BL_FULL][d.n[BL_FULL]++] = w;
      const int f = P.bs_form[w];   // 0, 1: single-launch form (shallow, deep); 2, 3: one chain
      const int list = (f < 2 ? BL_BS_FUSED : BL_BS_CHAIN) + (f & 1);
      d.idx[list][d.n[list]++] = w;
    }
    if (lw.kind == STEP_ACCEPT) d.idx[BL_ACC][d.n[BL_ACC]++] = w; else d.idx[BL_REJ][d.n[BL_REJ]++] = w;   // (first steps evaluate at x like rejected ones)
    if (!P.shape[w].vis && !bw.cleared) d.idx[BL_CLEAR][d.n[BL_CLEAR]++] = w;
    bw.cleared = !lw.grad_only && P.shape[w].zero_in_mcc;   // (what this round's model-cost launch leaves behind)
    d.radius[w] = lw.radius;
    d.first[w] = lw.kind == STEP_FIRST ? 1 : 0;
    d.new_J[w] = lw.kind != STEP_REJECT ? 1 : 0;
    d.grad_only[w] = lw.grad_only ? 1 : 0;
    c->reduce_seq += 1.0;
    d.seq[w] = c->reduce_seq;
  }
  hipStream_t s = P.stream;
  (void)hipMemcpyAsync(P.d_dyn, P.h_dyn, sizeof(BatchDyn), hipMemcpyHostToDevice, s);
  const BatchDyn* dd = P.d_dyn;
  launch_copy_batch(s, P.t_accept, dd, BL_ACC, d.n[BL_ACC]);                              // x <- x_cand
  evals(P, 0, dd, BL_REJ, d.n[BL_REJ], true);                                              // Jacobians at x (accepted windows have them: evaluated ahead)
  launch_zero_tiles_multi_batch(s, P.t_zero, dd, BL_CLEAR, d.n[BL_CLEAR]);
  launch_landmark_batch(s, P.t_lm, P.t_lm_tail, dd, BL_ALL, d.n[BL_ALL]);
  launch_idp_landmark_batch(s, P.t_idp_lm, P.t_idp_view, dd, BL_ALL, d.n[BL_ALL]);
  launch_idp_pairs_batch(s, P.t_idp_pairs, dd, BL_ALL, d.n[BL_ALL]);
  launch_pairs_band_batch(s, P.t_pairs_band, dd, BL_ALL, d.n[BL_ALL]);
  launch_pairs_batch(s, P.t_pairs, dd, BL_ALL, d.n[BL_ALL]);
  launch_small_assemble_set_batch(s, P.t_asm_set, dd, BL_ALL, d.n[BL_ALL]);
  launch_small_assemble_seg_batch(s, P.t_asm_seg, dd, BL_ALL, d.n[BL_ALL]);
  launch_marg_assemble_batch(s, P.t_marg_asm, dd, BL_ALL, d.n[BL_ALL]);
  launch_grad_norms_pose_diag_batch(s, P.t_gn, dd, BL_DIAG, d.n[BL_DIAG]);
  if (d.n[BL_FULL] > 0) {
    launch_chol_fused_batch(s, P.t_chol, dd, BL_FULL, d.n[BL_FULL]);
    const int nf[4] = {d.n[BL_BS_FUSED], d.n[BL_BS_FUSED + 1], d.n[BL_BS_CHAIN], d.n[BL_BS_CHAIN + 1]};
    launch_backsolve_batch(s, P.t_bs, dd, nf);
    launch_idp_backsub_batch(s, P.t_idp_backsub, dd, BL_FULL, d.n[BL_FULL]);   // (before the pose-only groups' model-cost terms, which read the step of rho)
    launch_backsub_mcc_batch(s, P.t_backsub, dd, BL_FULL, d.n[BL_FULL]);
    launch_small_mcc_batch(s, P.t_small_mcc, dd, BL_FULL, d.n[BL_FULL]);
    launch_marg_mcc_batch(s, P.t_marg_mcc, dd, BL_FULL, d.n[BL_FULL]);
    // (one pass: the candidate's residuals AND Jacobians now, its costs into the candidate's partial arrays — the table of the cost-only pass)
    evals(P, 1, dd, BL_FULL, d.n[BL_FULL], P.one_pass);
  }
  launch_final_reduce_batch(s, P.t_reduce, dd, BL_ALL, d.n[BL_ALL]);
  if (d.n[BL_FULL] > 0 && !P.one_pass) evals(P, 2, dd, BL_FULL, d.n[BL_FULL], true);   // ahead of the decisions, under the host round trip
}

int wait_round(BatchPlan& P, std::vector<BatchWin>& L, const std::vector<int>& act) {
  if (hipGetLastError() != hipSuccess) return BSGPU_ERR_DEVICE;
  const auto t0 = std::chrono::steady_clock::now();
  const double budget_s = 4.0 + 2e-6 * (double)P.max_tasks * (double)act.size();   // (fetch_scalars' rule, for all the windows' plans side by side)
  for (int w : act) {
    bsgpu_ctx* c = L[w].c;
    const volatile double* stamp = &c->h_scal[SC_SEQ];
    long spins = 0;
    while (__atomic_load_n(reinterpret_cast<const volatile uint64_t*>(stamp), __ATOMIC_ACQUIRE) != *reinterpret_cast<const uint64_t*>(&c->reduce_seq)) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget_s) {
        if (hipStreamSynchronize(P.stream) != hipSuccess) return BSGPU_ERR_DEVICE;
        if (*stamp != c->reduce_seq) return BSGPU_ERR_DEVICE;
        break;
      }
    }
  }
  return BSGPU_OK;
}

}  // namespace

void batch_forget(const bsgpu_ctx* c) {
  std::vector<DevicePlans*> all;
  {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (auto& kv : g_devices) all.push_back(kv.second.get());
  }
  for (DevicePlans* dp : all) {
    std::lock_guard<std::mutex> lock(dp->m);
    for (auto it = dp->plans.begin(); it != dp->plans.end();) { if ((*it)->names(c)) it = dp->plans.erase(it); else ++it; }
  }
}

// solves windows idx[0..m) of the call (all covered by the batched kernels, all on one device); rc[i] per window.  Returns false when
// the batch could not be set up or a round failed on the device (the windows are back at the values the call began with and the caller
// solves them one by one).
bool solve_batched(bsgpu_ctx* const* ctxs, const int* idx, int m, const bsgpu_options* o, int options_stride, bsgpu_summary* s, int* rc) {
  std::vector<bsgpu_ctx*> cs(m);
  for (int i = 0; i < m; ++i) cs[i] = ctxs[idx[i]];
  const bsgpu_options& o0 = o[options_stride ? idx[0] : 0];
  if (hipSetDevice(cs[0]->device) != hipSuccess) return false;
  DevicePlans& dp = device_plans(cs[0]->device);
  std::lock_guard<std::mutex> lock(dp.m);
  BatchPlan* Pp = nullptr;
  for (auto it = dp.plans.begin(); it != dp.plans.end(); ++it)
    if (plan_matches(**it, cs.data(), m, o0)) { dp.plans.splice(dp.plans.begin(), dp.plans, it); Pp = dp.plans.front().get(); break; }
  if (!Pp) {
    std::unique_ptr<BatchPlan> np(new BatchPlan());
    if (!build_plan(*np, cs.data(), m, o0)) return false;
    // (tables that name one of these contexts at an older generation are dead weight: dropped with the least recently used ones)
    for (auto it = dp.plans.begin(); it != dp.plans.end();) {
      bool stale = false;
      for (bsgpu_ctx* c : cs) stale = stale || (*it)->names(c);
      if (stale) it = dp.plans.erase(it); else ++it;
    }
    dp.plans.push_front(std::move(np));
    while (dp.plans.size() > kPlansPerDevice) dp.plans.pop_back();
    Pp = dp.plans.front().get();
  }
  BatchPlan& P = *Pp;
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  std::vector<BatchWin> L(m);
  std::vector<int> act;
  for (int i = 0; i < m; ++i) {
    bsgpu_ctx* c = cs[i];
    L[i].c = c;
    // (what the window's own stream still holds — bsgpu_reset_values is asynchronous — comes first)
    if (hipStreamQuery(c->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->stream); }
    L[i].lm.start(&o[options_stride ? idx[i] : 0], &s[idx[i]], &c->iters, c->n_tan, c->n_res, BSGPU_LINEAR_SCHUR_CHOLESKY);
    c->use_pcg = false; c->use_spcg = false; c->spec_J = false; c->cost_x_stale = false; c->pre_cleared = false;
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
  for (int i = 0; i < m; ++i) enqueue_fixed_cost(cs[i], P.stream);   // (windows with constant blocks: once, ahead of the first round)
  bool first = true;
  int err = BSGPU_OK;
  int64_t rounds = 0;
  while (!act.empty()) {
    ++rounds;
    enqueue_round(P, L, act);
    err = wait_round(P, L, act);
    if (err != BSGPU_OK) break;
    std::vector<int> next;
    for (int w : act) {
      bsgpu_ctx* c = L[w].c;
      LmState& lm = L[w].lm;
      // (one pass: after an accepted step the current point's own partial arrays hold an older point's costs — its cost is the candidate's cost the host
      //  holds, the same sum: LmState::advance cost_x_stale, as in a lone solve's one-pass steps)
      if (first) lm.begin(c->h_scal, c->any_inactive ? c->h_scal[SC_FIXED_COST] : 0.0, true);
      else lm.advance(c->h_scal, P.one_pass && lm.kind == STEP_ACCEPT, true);
      if (lm.retry_timeout) { L[w].timed_out = true; continue; }   // a single-launch kernel's wait timed out (shared GPU): this window again, alone, below
      if (!lm.done) next.push_back(w);
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
  {
    std::lock_guard<std::mutex> sl(g_stat_mutex);
    g_stat_windows += m; g_stat_rounds += rounds;
  }
  for (int i = 0; i < m; ++i) {
    bsgpu_ctx* c = cs[i];
    c->scal_mirrored = false; c->ev_reduce_pending = false; c->seq_pending = false; c->spec_J = false; c->pre_cleared = false; c->cost_x_stale = false;
  }
  if (err != BSGPU_OK) {
    // a device error or a round that never reported: every window back to where the call found it; the caller solves them one by one
    // (a dead device then reports through each window's own solve)
    (void)hipStreamSynchronize(P.stream);
    (void)hipGetLastError();
    for (int i = 0; i < m; ++i) (void)hipMemcpy(cs[i]->d_x, P.backup[i], sizeof(double) * cs[i]->h_x.size(), hipMemcpyDeviceToDevice);
    return false;
  }
  for (int i = 0; i < m; ++i) {
    bsgpu_ctx* c = cs[i];
    bsgpu_summary& sum = s[idx[i]];
    rc[idx[i]] = BSGPU_OK;
    if (L[i].timed_out) {
      // this window again, alone, on the launch-per-step path, from the values the call began with
      c->d_ftasks = nullptr;
      (void)hipMemcpy(c->d_x, P.backup[i], sizeof(double) * c->h_x.size(), hipMemcpyDeviceToDevice);
      rc[idx[i]] = solve(c, o[options_stride ? idx[i] : 0], sum);
      continue;
    }
    sum.device_time_in_seconds = ms * 1e-3;
    sum.total_time_in_seconds = total;
  }
  return true;
}

void batch_stats(int64_t* windows, int64_t* rounds) {
  std::lock_guard<std::mutex> lock(g_stat_mutex);
  if (windows) *windows = g_stat_windows;
  if (rounds) *rounds = g_stat_rounds;
}

}  // namespace bsg
