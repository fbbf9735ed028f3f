// solve(): [EXT] ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / SPARSE_NORMAL_CHOLESKY as the reference configures it
// (beam_slam_launch/config/vio.yaml:7-17) — a restatement of Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy that drives
// the device kernels: one host<->device round trip per LM iteration, hidden under the evaluation at the candidate.
#include <atomic>

#include <thread>

#include "bsgpu_ctx.h"
#include "lm_state.h"

namespace bsg {

void assemble_pcg(bsgpu_ctx* c, const bsgpu_options& o, double radius, bool new_J, bool first) {
  hipStream_t s = c->stream;
  launch_zero4(s, c->d_val, (int64_t)c->nblk * 9, c->d_rhs, c->n_pose, c->d_grad, c->n_pose, c->d_hdiag, c->n_pose);
  for (int t = 2; t < kNumInternal; ++t) launch_bsr_assemble(s, c->small[t], c->d_slots[t], c->d_val);
  launch_bsr_assemble_seg(s, c->d_small_groups, c->n_bsr_seg, c->d_bsr_seg_start, c->d_bsr_seg_slot, c->d_bsr_seg_row, c->d_bsr_contrib, c->d_val, c->d_rhs,
                          c->d_grad, c->d_hdiag);
  launch_bsr_finish_diag(s, c->nbr, c->d_diag_slot, c->d_pair_slot, c->d_val, c->d_hdiag, radius, first ? 1 : 0, new_J ? 1 : 0, o.jacobi_scaling,
                         o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_Minv);
  if (new_J) {
    launch_grad_norms(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_gpart);
  }
}

// (H + Lambda) y = g by block-Jacobi PCG; the stop test lives on the device, the host looks at it every 20 iterations
static std::atomic<int> g_pcg_persistent_in_flight{0};
void pcg_solve(bsgpu_ctx* c, const bsgpu_options& o) {
  hipStream_t s = c->stream;
  if (c->pcg_persist.G > 0) {
    // one resident launch for the whole solve (k_pcg.hip).  Its workgroups wait for each other, so two of them must not share the
    // device: a second solver thread of this process takes the launch-per-iteration path meanwhile, and a time-out inside the kernel
    // (another process's) makes this context do so for good.
    const double tol2p = o.pcg_tolerance * o.pcg_tolerance;
    const int max_itp = o.pcg_max_iterations > 0 ? o.pcg_max_iterations : 2000;
    const int ns = pcg_num_scalars();
    int expected = 0;
    if (g_pcg_persistent_in_flight.compare_exchange_strong(expected, 1)) {
      launch_pcg_coarse(s, c->pcg_persist, c->nbr, c->d_row_ptr, c->d_col, c->d_val, c->d_x);
      const bool launched = launch_pcg_persistent(s, c->pcg_persist, c->nbr, c->d_row_ptr, c->d_val, c->d_Minv, c->d_rhs, c->d_px, c->pcg_persist.zg, c->d_psc, tol2p, max_itp);
      // Its verdict (done, iterations) is read with the step's other scalars, not here: a synchronisation at this point left the device idle
      // for the host's round trip and the launches of the rest of the step, once per LM iteration (C4: ~25 of 690 us).  A launch that did
      // not finish (its workgroups were not all scheduled: a shared device) makes pcg_check() ask for the step again, launch per iteration.
      if (!c->h_pcg_lazy && hipHostMalloc((void**)&c->h_pcg_lazy, sizeof(double) * 8) != hipSuccess) { c->h_pcg_lazy = nullptr; (void)hipGetLastError(); }
      if (launched && c->h_pcg_lazy && hipMemcpyAsync(c->h_pcg_lazy, c->d_psc, sizeof(double) * ns, hipMemcpyDeviceToHost, s) == hipSuccess) {
        c->pcg_check_pending = true;   // (the guard stays taken until the verdict is read)
        return;
      }
      double sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      bool ok = launched && hipMemcpyAsync(sc, c->d_psc, sizeof(double) * ns, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
      g_pcg_persistent_in_flight.store(0);
      if (ok && sc[pcg_done_slot()] > 0.0) { c->pcg_iters_total += (int)sc[pcg_iters_slot()]; return; }
      fprintf(stderr, "[bsgpu] the resident PCG launch was given up (launched %d, done %g after %g iterations): launch-per-iteration path from here on\n", (int)launched, sc[pcg_done_slot()], sc[pcg_iters_slot()]);
      (void)hipGetLastError();
      c->pcg_persist.G = 0;
    }
  }
  launch_pcg_init(s, c->nbr, c->d_rhs, c->d_Minv, c->d_px, c->d_pr, c->d_pz, c->d_pp, c->d_pp1, c->d_ppart2, c->d_psc);
  const double tol2 = o.pcg_tolerance * o.pcg_tolerance;
  const int max_it = o.pcg_max_iterations > 0 ? o.pcg_max_iterations : 2000;
  const int ns = pcg_num_scalars();
  if (!c->h_pcg && hipHostMalloc((void**)&c->h_pcg, sizeof(double) * 2 * ns) != hipSuccess) { c->h_pcg = nullptr; (void)hipGetLastError(); }
  for (hipEvent_t& e : c->pcg_ev)
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; (void)hipGetLastError(); }
  const bool pipelined = c->h_pcg && c->pcg_ev[0] && c->pcg_ev[1];
  // The stop flag lives on the device and is sticky; the host looks at it once per chunk of iterations.  The read-back of
  // chunk n is waited for only after chunk n+1 has been enqueued, so the stream never drains while the host decides
  // (a blocking check per chunk left the device idle ~28 us each time); the iterations enqueued past convergence see the
  // flag and do nothing.
  const int kChunk = 12;
  double last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto enqueue_chunk = [&](int it0, int n, int slot) {
    for (int k = 0; k < n; ++k)
      launch_pcg_iteration(s, it0 + k, c->nbr, c->d_row_ptr, c->d_col, c->d_val, c->d_Minv, c->d_px, c->d_pr, c->d_pz, c->d_pp, c->d_pp1,
                           c->d_pq, c->d_ppart, c->d_ppart2, c->d_psc, tol2);
    if (pipelined) {
      (void)hipMemcpyAsync(c->h_pcg + slot * ns, c->d_psc, sizeof(double) * ns, hipMemcpyDeviceToHost, s);
      (void)hipEventRecord(c->pcg_ev[slot], s);
    }
  };
  if (!pipelined) {
    for (int it = 0; it < max_it;) {
      const int chunk = std::min(20, max_it - it);
      enqueue_chunk(it, chunk, 0);
      it += chunk;
      (void)hipMemcpyAsync(last, c->d_psc, sizeof(double) * ns, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      if (last[pcg_done_slot()] != 0.0) break;
    }
  } else {
    int it = std::min(kChunk, max_it), slot = 0;
    enqueue_chunk(0, it, slot);
    for (;;) {
      int next_n = std::min(kChunk, max_it - it);
      if (next_n > 0) enqueue_chunk(it, next_n, slot ^ 1);
      (void)hipEventSynchronize(c->pcg_ev[slot]);
      std::memcpy(last, c->h_pcg + slot * ns, sizeof(double) * ns);
      if (last[pcg_done_slot()] != 0.0 || next_n <= 0) {
        if (next_n > 0) {   // the chunk in flight: let it drain so that its read-back does not land in a later solve's slot
          (void)hipEventSynchronize(c->pcg_ev[slot ^ 1]);
          std::memcpy(last, c->h_pcg + (slot ^ 1) * ns, sizeof(double) * ns);
        }
        break;
      }
      it += next_n;
      slot ^= 1;
    }
  }
  c->pcg_iters_total += (int)last[pcg_iters_slot()];
}

// after the step's scalars are on the host (the stream has passed the read-back queued behind the resident launch): its verdict.
// false: the launch did not finish — the step's linear solve is void; the context takes the launch-per-iteration path from here on.
bool pcg_check(bsgpu_ctx* c) {
  if (!c->pcg_check_pending) return true;
  c->pcg_check_pending = false;
  g_pcg_persistent_in_flight.store(0);
  double done = c->h_pcg_lazy[pcg_done_slot()];
  const double iters = c->h_pcg_lazy[pcg_iters_slot()];
  // (tests: BSGPU_PCG_GIVE_UP=n declares the n-th verdict of the process a failure — the path a shared device takes)
  if (const char* e = getenv("BSGPU_PCG_GIVE_UP")) { if (++c->pcg_verdicts_seen == atoi(e)) done = 0.0; }   // (counted per context)
  if (done > 0.0) { c->pcg_iters_total += (int)iters; return true; }
  fprintf(stderr, "[bsgpu] the resident PCG launch was given up (done %g after %g iterations): launch-per-iteration path from here on\n", done, iters);
  c->pcg_persist.G = 0;
  return false;
}

// S y = rhs on the assembled reduced camera system by block-Jacobi PCG (ITERATIVE_SCHUR + SCHUR_JACOBI): same pipelined stop test
void spcg_solve(bsgpu_ctx* c, const bsgpu_options& o) {
  hipStream_t s = c->stream;
  const int T = c->plan.T, ld = c->npad;
  const double* b = c->d_S + (size_t)c->plan.rhs_row * ld;
  launch_spcg_prepare(s, T, c->d_S, ld, c->d_sMinv);
  launch_spcg_init(s, T, b, c->d_sMinv, c->d_sx, c->d_sr, c->d_sz, c->d_sp0, c->d_sp1, c->d_spart, c->d_ssc);
  const double tol2 = o.pcg_tolerance * o.pcg_tolerance;
  const int max_it = o.pcg_max_iterations > 0 ? o.pcg_max_iterations : 2000;
  const int ns = pcg_num_scalars();
  if (!c->h_pcg && hipHostMalloc((void**)&c->h_pcg, sizeof(double) * 2 * ns) != hipSuccess) { c->h_pcg = nullptr; (void)hipGetLastError(); }
  for (hipEvent_t& e : c->pcg_ev)
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; (void)hipGetLastError(); }
  const bool pipelined = c->h_pcg && c->pcg_ev[0] && c->pcg_ev[1];
  const int kChunk = 12;
  double last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto enqueue_chunk = [&](int it0, int n, int slot) {
    for (int k = 0; k < n; ++k)
      launch_spcg_iteration(s, it0 + k, T, c->d_S, ld, c->n_schunks, c->d_schunk_row, c->d_schunk_ptr, c->d_srow_chunk_ptr, c->d_tcol, c->d_sMinv,
                            c->d_sx, c->d_sr, c->d_sz, c->d_sp0, c->d_sp1, c->d_sq, c->d_spart_pq, c->d_spart, c->d_ssc, tol2);
    if (pipelined) {
      (void)hipMemcpyAsync(c->h_pcg + slot * ns, c->d_ssc, sizeof(double) * ns, hipMemcpyDeviceToHost, s);
      (void)hipEventRecord(c->pcg_ev[slot], s);
    }
  };
  if (!pipelined) {
    for (int it = 0; it < max_it;) {
      const int chunk = std::min(20, max_it - it);
      enqueue_chunk(it, chunk, 0);
      it += chunk;
      (void)hipMemcpyAsync(last, c->d_ssc, sizeof(double) * ns, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      if (last[pcg_done_slot()] != 0.0) break;
    }
  } else {
    int it = std::min(kChunk, max_it), slot = 0;
    enqueue_chunk(0, it, slot);
    for (;;) {
      int next_n = std::min(kChunk, max_it - it);
      if (next_n > 0) enqueue_chunk(it, next_n, slot ^ 1);
      (void)hipEventSynchronize(c->pcg_ev[slot]);
      std::memcpy(last, c->h_pcg + slot * ns, sizeof(double) * ns);
      if (last[pcg_done_slot()] != 0.0 || next_n <= 0) {
        if (next_n > 0) {
          (void)hipEventSynchronize(c->pcg_ev[slot ^ 1]);
          std::memcpy(last, c->h_pcg + (slot ^ 1) * ns, sizeof(double) * ns);
        }
        break;
      }
      it += next_n;
      slot ^= 1;
    }
  }
  c->pcg_iters_total += (int)last[pcg_iters_slot()];
  launch_spcg_finish(s, T, c->d_sx, c->d_inat, c->n_pose, c->d_ytan, c->d_delta);
}

// A window after a slide with true marginalisation carries ONE dense prior (fixed_lag_smoother.cpp:269-272): its evaluation, its J^T J and its
// model-cost terms ride in the launches of the window's other pose-only factors instead of three launches of their own behind them
// (BSGPU_MARG_RIDE=0: the launches of their own; several priors: the first one rides)
static int marg_rider(const bsgpu_ctx* c) {
  static const bool off = getenv("BSGPU_MARG_RIDE") && atoi(getenv("BSGPU_MARG_RIDE")) == 0;
  if (off || c->use_graphs) return -1;
  for (size_t i = 0; i < c->marg.size(); ++i)
    if (c->marg[i].active && c->marg[i].dev.rows > 0) return (int)i;
  return -1;
}

// residuals (+ Jacobians) of every factor group; per-group cost partials go to the arrays the
// end-of-step reduction sums (current point: slot SC_COST_X, candidate: SC_COST_CAND)
// red: the end-of-step reduction of the step just computed rides in the visual-inertial evaluation launch (the caller has checked that
// this window takes that launch: reduce_rides())
void eval_all(bsgpu_ctx* c, const double* x, bool with_J, int slot, const ReduceRide* red) {
  hipStream_t s = c->stream;
  const bool cand = slot == SC_COST_CAND;
  // (either IMU type may be absent: a window whose oldest state has been marginalised / held constant has no IMU prior any more)
  const bool imu_pair = c->small[BSGPU_F_IMU_DELTA].n + c->small[BSGPU_F_IMU_PRIOR].n > 0;
  // (a visual-inertial window: the IMU factors ride in the reprojection launch — the cost-only pass always, the pass with Jacobians
  // unless BSGPU_EVAL_MERGE=0: there the IMU body's registers cost the reprojection kernel a wave of occupancy per SIMD)
  static const int merge_mode = getenv("BSGPU_EVAL_MERGE") ? atoi(getenv("BSGPU_EVAL_MERGE")) : 2;   // 0: never, 1: cost-only passes, 2: both
  const bool merged = imu_pair && c->vis.n > 0 && (with_J ? merge_mode >= 2 : merge_mode >= 1);
  if (merged)
    launch_visual_imu_eval(s, c->vis, c->small[BSGPU_F_IMU_DELTA], c->small[BSGPU_F_IMU_PRIOR], x, c->d_cams, c->d_losses, with_J,
                           cand ? c->vis.cost_part_cand : c->vis.cost_part,
                           cand ? c->d_small_part_cand[BSGPU_F_IMU_DELTA] : c->d_small_part[BSGPU_F_IMU_DELTA],
                           cand ? c->d_small_part_cand[BSGPU_F_IMU_PRIOR] : c->d_small_part[BSGPU_F_IMU_PRIOR], red);
  else if (c->vis.n) launch_reproj_eval(s, c->vis, x, c->d_cams, c->d_losses, with_J, cand ? c->vis.cost_part_cand : c->vis.cost_part);
  if (with_J) phase_mark(c, BSGPU_PHASE_EVAL_REPROJ);
  // (a lidar-inertial window: they ride in the relative-pose evaluation instead)
  int rel_t = -1;
  if (imu_pair && !merged && (with_J ? merge_mode >= 2 : merge_mode >= 1))
    for (int t : {(int)BSGPU_F_RELPOSE_EXT, (int)BSGPU_F_RELPOSE}) if (rel_t < 0 && c->small[t].n > 0) rel_t = t;
  if (rel_t >= 0)
    launch_relpose_imu_eval(s, c->small[rel_t], c->small[BSGPU_F_IMU_DELTA], c->small[BSGPU_F_IMU_PRIOR], x, c->d_losses, with_J,
                            cand ? c->d_small_part_cand[rel_t] : c->d_small_part[rel_t],
                            cand ? c->d_small_part_cand[BSGPU_F_IMU_DELTA] : c->d_small_part[BSGPU_F_IMU_DELTA],
                            cand ? c->d_small_part_cand[BSGPU_F_IMU_PRIOR] : c->d_small_part[BSGPU_F_IMU_PRIOR], red);
  if (imu_pair && !merged && rel_t < 0)
    launch_imu_eval(s, c->small[BSGPU_F_IMU_DELTA], c->small[BSGPU_F_IMU_PRIOR], x, c->d_losses, with_J,
                    cand ? c->d_small_part_cand[BSGPU_F_IMU_DELTA] : c->d_small_part[BSGPU_F_IMU_DELTA],
                    cand ? c->d_small_part_cand[BSGPU_F_IMU_PRIOR] : c->d_small_part[BSGPU_F_IMU_PRIOR]);
  int marg_done = -1;
  {
    // the groups no fused launch carries: ONE launch for all of them when there are several (a pose graph's constraints + the prior on its
    // first pose), else the group's own kernel
    SmallGroup gs[kNumInternal];
    double* ps[kNumInternal];
    int ng = 0;
    for (int t = 2; t < kNumInternal; ++t) {
      if (imu_pair && (t == BSGPU_F_IMU_DELTA || t == BSGPU_F_IMU_PRIOR)) continue;
      if (t == rel_t || !c->small[t].n) continue;
      gs[ng] = c->small[t]; ps[ng] = cand ? c->d_small_part_cand[t] : c->d_small_part[t]; ++ng;
    }
    static const bool separate = getenv("BSGPU_EVAL_SEPARATE") != nullptr;
    const int mr = separate ? -1 : marg_rider(c);
    if (mr >= 0 && launch_small_eval_set_marg(s, gs, ps, ng, x, c->d_losses, with_J, c->marg[mr].dev, cand ? c->marg[mr].part_cand : c->marg[mr].part))
      marg_done = mr;
    else if (ng < 2 || separate || !launch_small_eval_set(s, gs, ps, ng, x, c->d_losses, with_J))
      for (int i = 0; i < ng; ++i) launch_small_eval(s, gs[i], x, c->d_losses, with_J, ps[i]);
  }
  for (size_t i = 0; i < c->marg.size(); ++i) {
    const auto& mc = c->marg[i];
    if (mc.active && (int)i != marg_done) launch_marg_eval(s, mc.dev, x, with_J, cand ? mc.part_cand : mc.part);
  }
  if (with_J) phase_mark(c, BSGPU_PHASE_EVAL_OTHER);
}
// the reduction can ride in the evaluation launched ahead of the decision (k_small.hip visual_imu_eval_reduce_kernel): a visual-inertial
// window on eager launches whose host polls the mirror's stamp, outside bsgpu_profile_step (which times the reduction in its own phase)
// (what any riding reduction needs: eager launches, a host that polls the mirror's stamp, outside bsgpu_profile_step)
bool reduce_can_ride(const bsgpu_ctx* c) {
  static const bool by_event = getenv("BSGPU_SCALARS_EVENT") != nullptr, off = getenv("BSGPU_REDUCE_LAUNCH") != nullptr;
  return !off && !by_event && !c->use_graphs && !c->use_pcg && !c->prof_events && c->h_scal_dev != nullptr && c->d_reduce_counter != nullptr && c->n_reduce > 0;
}
bool reduce_rides(const bsgpu_ctx* c) {
  static const bool by_event = getenv("BSGPU_SCALARS_EVENT") != nullptr, off = getenv("BSGPU_REDUCE_LAUNCH") != nullptr;
  static const int merge_mode = getenv("BSGPU_EVAL_MERGE") ? atoi(getenv("BSGPU_EVAL_MERGE")) : 2;
  const bool imu_pair = c->small[BSGPU_F_IMU_DELTA].n + c->small[BSGPU_F_IMU_PRIOR].n > 0;
  // (the launch that carries it: the visual-inertial evaluation, or a lidar-inertial window's relative-pose + IMU evaluation — eval_all)
  const bool carrier = imu_pair && merge_mode >= 2 && (c->vis.n > 0 || c->small[BSGPU_F_RELPOSE_EXT].n > 0 || c->small[BSGPU_F_RELPOSE].n > 0);
  return !off && !by_event && !c->use_graphs && !c->use_pcg && !c->prof_events && carrier && c->h_scal_dev != nullptr && c->d_reduce_counter != nullptr &&
         c->n_reduce > 0;
}
void final_reduce(bsgpu_ctx* c) {
  // (BSGPU_SCALARS_EVENT=1: the host waits for an event recorded behind the reduction instead of polling the mirror's stamp)
  static const bool by_event = getenv("BSGPU_SCALARS_EVENT") != nullptr;
  const bool stamp = !c->use_graphs && !by_event && c->h_scal_dev != nullptr && c->d_reduce_counter != nullptr && c->n_reduce > 0;
  if (stamp) c->reduce_seq += 1.0;
  launch_final_reduce(c->stream, c->d_reduce, c->n_reduce, SC_GRAD_NORM2 + 1, c->d_scal, c->h_scal_dev, stamp ? c->d_reduce_counter : nullptr, c->reduce_seq);
  c->scal_mirrored = c->h_scal_dev != nullptr && c->n_reduce > 0;
  c->seq_pending = stamp;
  if (stamp) { c->ev_reduce_pending = false; return; }
  // (not under graph capture / replay: an event recorded while capturing is a graph node, not something the host can wait on)
  if (c->use_graphs) { c->ev_reduce_pending = false; return; }
  if (!c->ev_reduce && hipEventCreateWithFlags(&c->ev_reduce, hipEventDisableTiming) != hipSuccess) { c->ev_reduce = nullptr; (void)hipGetLastError(); }
  c->ev_reduce_pending = c->scal_mirrored && c->ev_reduce && hipEventRecord(c->ev_reduce, c->stream) == hipSuccess;
}

// gradient_only: the caller wants the gradient (and its norms) of the current point and will not factorise — the
// camera-pair blocks of the reduced system, four fifths of the pair kernel's work, are skipped
void assemble(bsgpu_ctx* c, const bsgpu_options& o, double radius, bool new_J, bool first, bool gradient_only, bool factor_follows, const ReduceRide* red) {
  c->reduce_carried = false;
  if (c->use_pcg) { assemble_pcg(c, o, radius, new_J, first); return; }
  hipStream_t s = c->stream;
  // one launch clears the reduced system, gradient, diagonal and the scalars of this step (GRAD_MAX, GRAD_NORM2, CHOL_FAIL)
  // ... and carries the radius of this step (not under graph replay, whose kernel arguments are frozen)
  ZeroStep zs;
  zs.S = c->d_S; zs.ld = c->npad; zs.tiles = c->d_touched; zs.n_tiles = c->n_touched;
  zs.a = c->d_grad; zs.na = c->n_pose; zs.b = c->d_hdiag; zs.nb = c->n_pose;
  zs.c = new_J ? c->d_scal + SC_GRAD_MAX : c->d_scal + SC_CHOL_FAIL; zs.nc = new_J ? 3 : 1;
  zs.radius_slot = c->use_graphs ? nullptr : c->d_scal + SC_RADIUS; zs.radius = radius;
  // (an assembly ahead whose radius the reduction riding in its landmark launch decides — LmDecide: `radius` is not the step's, and the launches
  //  behind the landmark launch return at once unless that decision was "accepted")
  const bool dev_decides = red != nullptr && red->lmd.on && red->dec != nullptr && c->vis.n_lm > 0;
  GoWord go;
  if (dev_decides) { zs.radius_slot = nullptr; go.p = red->dec; }
  // (with landmarks and eager launches the clearing rides in the landmark launch: independent work, one launch less on the path)
  const bool merged = c->vis.n_lm > 0 && !c->use_graphs;
  // (a window without Euclidean landmarks clears at the END of a step, in the launch that carries the candidate update: the next
  //  assembly then finds everything clean — linear_solve_and_candidate)
  const bool cleared = c->pre_cleared && !merged && !c->use_graphs;
  c->pre_cleared = false;
  if (!merged && !cleared) launch_zero_tiles_multi(s, zs.S, zs.ld, zs.tiles, zs.n_tiles, zs.a, zs.na, zs.b, zs.nb, zs.c, zs.nc, zs.radius_slot, zs.radius);
  c->scal_mirrored = false;
  // (a reduction handed in rides in the first launch that takes it: the landmark launch, or below the pose-only factors' segment launch)
  const bool red_in_lm = red != nullptr && c->vis.n_lm > 0;
  // (... whose clearing then leaves the step's scalars alone: the reduction's last unit mirrors the factorisation's flag and clears it itself,
  //  and the gradient norms' slots are the reduction's to write)
  if (red_in_lm) zs.nc = 0;
  launch_landmark(s, c->vis, c->n_pose, merged ? nullptr : c->d_scal + SC_RADIUS, first ? 1 : 0, new_J ? 1 : 0, o.jacobi_scaling, o.min_lm_diagonal,
                  o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad, merged ? &zs : nullptr, radius, red_in_lm ? red : nullptr);
  if (red_in_lm) c->reduce_carried = true;
  // (inverse-depth landmarks: their scalar elimination, k_idp.hip — after the clearing above, which rides in the landmark launch)
  launch_idp_landmark(s, c->idp, c->small[BSGPU_F_IDP_REPROJ], c->use_graphs ? c->d_scal + SC_RADIUS : nullptr, radius, first ? 1 : 0, new_J ? 1 : 0,
                      o.jacobi_scaling, o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad);
  launch_idp_pairs(s, c->idp, c->small[BSGPU_F_IDP_REPROJ], c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, gradient_only);
  phase_mark(c, BSGPU_PHASE_LANDMARK);
  int marg_done = -1;
  {
    // (the factor-wise assembled pose-only groups ride in the pair launch when there is one; further groups, or all of them, go by themselves)
    SmallGroupSet set;
    int taken = 0, units = 0;
    const bool band = c->vis.n_band_units > 0;   // (band landmarks: k_band.hip; what does not qualify keeps its pair entries)
    if (c->vis.n_seg > 0 || band) units = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set, &taken);
    if (units == 0) taken = 0;
    // (the band units add to the lower triangle only where the fused tiled factorisation is what reads the system next: enqueue_step's assemblies — the
    //  covariance / marginalisation entry points, the PCG on the reduced system and the launch-per-step fallback get both triangles.  BSGPU_BAND_LOWER=0: always both)
    static const bool lower_off = getenv("BSGPU_BAND_LOWER") && atoi(getenv("BSGPU_BAND_LOWER")) == 0;
    const bool lower_only = !lower_off && factor_follows && !gradient_only && !c->use_spcg && !c->use_pcg && c->dense_ok && c->d_ftasks && c->d_fsync && c->d_tile_tot && c->d_Winv;
    if (band) launch_pairs_band(s, c->vis, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, gradient_only, units > 0 ? &set : nullptr, units, lower_only, go);
    launch_pairs(s, c->vis, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, gradient_only, (units > 0 && !band) ? &set : nullptr, band ? 0 : units, go);
    phase_mark(c, BSGPU_PHASE_PAIRS);
    // (... or, without a pair launch, in the launch of the segment-wise assembled groups)
    SmallGroupSet set2;
    int taken2 = 0, units2 = 0;
    if (units == 0 && c->n_sa_seg + c->n_asm_grp > 0) units2 = small_assemble_first_set(c->small_factorwise + 2, kNumInternal - 2, &set2, &taken2);
    if (units2 == 0) taken2 = 0;
    launch_small_assemble_set(s, c->small_factorwise + 2 + taken + taken2, kNumInternal - 2 - taken - taken2, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag,
                              c->d_dpos);
    const int mr = marg_rider(c);
    bool seg_carried = false;
    if (launch_small_assemble_seg(s, c->d_small_groups, c->n_sa_seg, c->d_sa_seg_start, c->d_sa_seg_ra, c->d_sa_seg_rb, c->d_sa_contrib, c->d_S, c->npad,
                                  c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos, units2 > 0 ? &set2 : nullptr, units2, c->n_asm_grp, c->d_asm_grp,
                                  mr >= 0 ? &c->marg[mr].dev : nullptr, (red && !c->reduce_carried && mr < 0) ? red : nullptr, &seg_carried))
      marg_done = mr;
    if (seg_carried) c->reduce_carried = true;
  }
  for (size_t i = 0; i < c->marg.size(); ++i) {
    const auto& mc = c->marg[i];
    if (mc.active && (int)i != marg_done) launch_marg_assemble(s, mc.dev, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_dpos);
  }
  // With the single-launch factorisation next, neither needs a launch of its own on the dependent path: its plan has one task per tile that
  // adds the LM diagonal as the tile's first update, and the gradient norms — which only the end-of-step reduction reads — are units of
  // work of the same launch (dense_plan.h kFusedDiagAdd / kFusedRider; BSGPU_POSE_DIAG_LAUNCH=1 plans without them).
  c->diag_in_chol = factor_follows && !gradient_only && !c->use_graphs && !c->use_spcg && c->dense_ok && c->n_pose > 0 && c->plan.diag_tasks &&
                    c->plan.rider_tasks * 256 >= c->nb && c->d_ftasks && c->d_fsync && c->d_tile_tot && c->d_Winv;
  c->lm_diag = LmDiag(); c->gn_ride = GradNormRide();
  if (c->diag_in_chol) {
    LmDiag& d = c->lm_diag;
    d.hdiag = c->d_hdiag; d.scale = c->d_scale; d.dcl = c->d_dcl; d.inat = c->d_inat;
    d.inv_radius = 1.0 / radius; d.lm_lo = o.min_lm_diagonal; d.lm_hi = o.max_lm_diagonal;
    d.compute_scale = first ? 1 : 0; d.compute_dcl = new_J ? 1 : 0; d.jacobi = o.jacobi_scaling;
    if (new_J) {
      GradNormRide& g = c->gn_ride;
      g.nb = c->nb; g.xoff = c->d_blk_xoff; g.toff = c->d_blk_toff; g.size = c->d_blk_size; g.manifold = c->d_blk_manifold; g.x = c->d_x; g.grad = c->d_grad;
      g.gpart = c->d_gpart;
    }
  } else if (new_J)   // the LM diagonal and the gradient norms both follow the assembly and do not depend on each other: one launch
    launch_grad_norms_pose_diag(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_gpart,
                                c->n_pose, c->d_S, c->npad, c->d_hdiag, cleared ? nullptr : c->d_scal + SC_RADIUS, first ? 1 : 0, 1, o.jacobi_scaling,
                                o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_inat, radius);
  else
    launch_pose_diag(s, c->n_pose, c->d_S, c->npad, c->d_hdiag, cleared ? nullptr : c->d_scal + SC_RADIUS, 0, 0, o.jacobi_scaling,
                     o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_inat, radius);
  phase_mark(c, BSGPU_PHASE_ASSEMBLE_OTHER);
}

void dense_factor(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* scal) {
  const int ld = P.npad;
  if (D.ftasks && D.fsync && D.tile_tot && D.Winv) {   // the whole factorisation in one launch
    const bool plain = !D.diag.hdiag && D.gn.nb == 0 && D.ftasks_plain && D.tile_tot_plain;   // (nothing to carry: the list without those tasks)
    launch_chol_fused(s, S, D.Lp, ld, plain ? D.ftasks_plain : D.ftasks, plain ? D.n_ftasks_plain : (int)P.ftasks.size(), plain ? D.tile_tot_plain : D.tile_tot, D.nreal,
                      D.Vinv, scal, D.fsync, D.Winv, D.rhs_rows, D.diag, D.gn, /*diag_tasks_in_list=*/!plain && P.diag_tasks);
    return;
  }
  for (int st = 0; st < P.n_steps(); ++st) {
    // (tiles no look-ahead factors are factored inside the panel step itself: PanelDesc::self_potrf)
    launch_chol_panel_step(s, S, D.Lp, ld, D.panels + P.step_off[st], P.step_off[st + 1] - P.step_off[st], P.step_maxrows[st],
                           D.rows_flat, D.nreal, D.Vinv, scal, D.tile_sync, P.panels.data() + P.step_off[st], P.rows_flat.data());
  }
}
void dense_factor_solve(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* y, double* scal, const int* iperm, int n_pose,
                        double* y_tan, double* delta) {
  dense_factor(s, P, D, S, scal);
  dense_backsolve(s, P, D, S, y, iperm, n_pose, y_tan, delta);
}
void dense_backsolve(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* y, const int* iperm, int n_pose, double* y_tan,
                     double* delta) {
  const int ld = P.npad;
  // y' = the rhs row after forward substitution: row rhs_row of the shadow matrix (the rhs tile is an
  // off-diagonal row tile of every panel)
  const double* rhs_row = D.Lp + (size_t)P.rhs_row * ld;
  const bool single_root = P.bs_group_off.size() > 1 && P.bs_group_off[1] - P.bs_group_off[0] == 1;
  // one launch per group of chains: root separator, the separator levels below it, then every piece (dense_plan.h); level-synchronous
  // form: the chains walk their own row tiles only, and between two groups one wide launch applies the finished group to the rest
  const bool level_sync = P.bs_level_sync && D.bs_desc_chain && D.bs_upd;
  const int G = (int)P.bs_group_off.size() - 1;
  if (level_sync && D.bs_items4 && D.bs_sync && D.scal && D.Winv && D.ftasks && D.fsync) {   // (the fused factorisation writes the tile inverses)   // everything in one launch, if its workgroups can all be resident
    int max_len = 1, max_rows = 0;
    for (size_t i = 0; i < P.chain_begin.size(); ++i) max_len = std::max(max_len, P.chain_end[i] - P.chain_begin[i]);
    for (int g = 0; g < G; ++g) max_rows = std::max(max_rows, P.bs_group_maxrows[g]);
    if (launch_chol_backsolve_fused(s, D.Lp, D.Winv, ld, D.bs_desc_chain, D.chain_begin, D.chain_end, D.rows_flat_chain, (int)P.chain_begin.size(),
                                    D.bs_chain_group, D.bs_grp_nchains, D.bs_grp_nitems, G, D.bs_items4, (int)P.bs_upd.size() / 3 * (P.bs_upd_off.back() > 0 ? 1 : 0),
                                    D.bs_upd_rows, D.bs_tile_updated, y, P.npad, max_len, std::max(1, max_rows), rhs_row, iperm, n_pose, y_tan, delta,
                                    D.bs_sync, D.scal, D.bs_order))
      return;
  }
  const bool w_valid = D.ftasks && D.fsync && D.tile_tot && D.Winv;   // (dense_factor: the fused factorisation ran and left the tile inverses)
  if (!single_root) launch_copy(s, rhs_row, y, (int64_t)P.T * 64, 64);   // (a single root chain copies it itself on the way)
  for (int g = 0; g < G; ++g) {
    const int c0 = P.bs_group_off[g], c1 = P.bs_group_off[g + 1];
    int max_len = 1;
    for (int i = c0; i < c1; ++i) max_len = std::max(max_len, P.chain_end[i] - P.chain_begin[i]);
    launch_chol_backsolve_chains(s, S, D.Lp, D.Vinv, ld, level_sync ? D.bs_desc_chain : D.bs_desc, D.chain_begin + c0, D.chain_end + c0, c1 - c0,
                                 level_sync ? D.rows_flat_chain : D.rows_flat, y, P.npad, max_len, (g == 0 && single_root) ? rhs_row : nullptr, iperm,
                                 n_pose, y_tan, delta, level_sync ? P.bs_group_maxrows[g] : 0, w_valid ? D.Winv : nullptr);
    if (level_sync && g + 1 < G) {
      const int i0 = P.bs_upd_off[g], i1 = P.bs_upd_off[g + 1];
      launch_chol_backsolve_update(s, D.Lp, ld, D.bs_upd + 3 * (size_t)i0, i1 - i0, D.bs_upd_rows, y);
    }
  }
}

void linear_solve_and_candidate(bsgpu_ctx* c, const bsgpu_options& o, bool defer_reduce = false, bool skip_cand_cost = false) {
  hipStream_t s = c->stream;
  if (c->use_pcg) {
    pcg_solve(c, o);
    launch_negate_pose(s, c->n_pose, c->d_px, c->d_delta);
  } else if (c->use_spcg && c->n_pose > 0) {
    spcg_solve(c, o);
    phase_mark(c, BSGPU_PHASE_FACTOR);
    phase_mark(c, BSGPU_PHASE_BACKSOLVE);
  } else if (c->n_pose > 0) {
    DenseDev D{c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
               c->d_bs_desc, c->d_chain_begin, c->d_chain_end, c->d_tile_sync, c->d_ftasks, c->d_fsync,
               c->d_bs_desc_chain, c->d_rows_flat_chain, c->d_bs_upd, c->d_bs_upd_rows,
               c->d_bs_chain_group, c->d_bs_grp_nchains, c->d_bs_grp_nitems, c->d_bs_items4, c->d_bs_tile_updated, c->d_bs_sync, c->d_scal, c->d_Winv, c->d_bs_order, c->d_tile_tot, 1};
    if (c->diag_in_chol) { D.diag = c->lm_diag; D.gn = c->gn_ride; }
    D.ftasks_plain = c->d_ftasks_plain; D.tile_tot_plain = c->d_tile_tot_plain; D.n_ftasks_plain = c->n_ftasks_plain;
    dense_factor(s, c->plan, D, c->d_S, c->d_scal);
    phase_mark(c, BSGPU_PHASE_FACTOR);
    dense_backsolve(s, c->plan, D, c->d_S, c->d_y, c->d_inat, c->n_pose, c->d_ytan, c->d_delta);
    phase_mark(c, BSGPU_PHASE_BACKSOLVE);
  }
  // landmark back-substitution + the model-cost-change terms of the visual factors (partial arrays only, summed once at the end)
  launch_idp_backsub(s, c->idp, c->d_ytan, c->d_delta);   // (before the pose-only groups' model-cost terms, which read the step of rho)
  int marg_mcc_done = -1;
  {
    // (the model-cost terms of the first pose-only groups ride in the back-substitution launch; further groups, or all of them when
    // there is no visual launch, go by themselves)
    SmallGroupSet set;
    int taken = 0, units = 0;
    if (backsub_mcc_groups(c->vis) > 0) units = small_mcc_first_set(c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, &set, &taken);
    UpdateRide up;
    if (c->n_upd_blocks > 0) {
      up.n_blocks = c->n_upd_blocks; up.blocks = c->d_upd_blocks; up.xoff = c->d_blk_xoff; up.toff = c->d_blk_toff; up.size = c->d_blk_size;
      up.manifold = c->d_blk_manifold; up.lm_xoff = c->d_lm_xoff; up.x = c->d_x; up.x_cand = c->d_xcand; up.part = c->d_part_upd;
    }
    const int mr = marg_rider(c);
    if (launch_backsub_mcc(s, c->vis, c->n_pose, c->d_ytan, c->d_delta, c->vis.mcc_part, units > 0 ? &set : nullptr, units, c->n_upd_blocks > 0 ? &up : nullptr,
                           mr >= 0 ? &c->marg[mr].dev : nullptr, mr >= 0 ? c->marg[mr].part_mcc : nullptr))
      marg_mcc_done = mr;
    if (units == 0) taken = 0;
    UpdateRide all;
    if (c->upd_in_mcc) {   // (no Euclidean landmarks: every block's candidate rides in the pose-only groups' launch)
      all.n_blocks = c->nb; all.blocks = nullptr; all.xoff = c->d_blk_xoff; all.toff = c->d_blk_toff; all.size = c->d_blk_size;
      all.manifold = c->d_blk_manifold; all.x = c->d_x; all.x_cand = c->d_xcand; all.part = c->d_part_upd;
    }
    // ... and the next step's clearing with it (the factorisation of this step is done; nothing reads S, the gradient or the diagonal
    // before the next assembly) — when the end-of-step reduction mirrors the scalars (it then clears the factorisation's flag)
    static const bool clear_at_start = getenv("BSGPU_CLEAR_AT_START") != nullptr;
    ZeroStep zn;
    const bool ride_zero = c->upd_in_mcc && !c->use_pcg && !c->use_graphs && c->dense_ok && c->h_scal_dev != nullptr && !clear_at_start;
    if (ride_zero) {
      zn.S = c->d_S; zn.ld = c->npad; zn.tiles = c->d_touched; zn.n_tiles = c->n_touched;
      zn.a = c->d_grad; zn.na = c->n_pose; zn.b = c->d_hdiag; zn.nb = c->n_pose;
    }
    const bool carried = launch_small_mcc_set(s, c->small + 2 + taken, c->d_small_part_mcc + 2 + taken, kNumInternal - 2 - taken, c->d_delta,
                                              c->upd_in_mcc ? &all : nullptr, ride_zero ? &zn : nullptr);
    c->pre_cleared = ride_zero && carried;
    if (c->upd_in_mcc && !carried) launch_update_ride_only(s, c->d_delta, all);
  }
  for (size_t i = 0; i < c->marg.size(); ++i) {
    const auto& mc = c->marg[i];
    if (mc.active && (int)i != marg_mcc_done) launch_marg_mcc(s, mc.dev, c->d_delta, mc.part_mcc);
  }
  phase_mark(c, BSGPU_PHASE_BACKSUB);
  int n_part = 0;
  if (c->n_upd_blocks == 0 && !c->upd_in_mcc)   // (else the update rode in the landmark back-substitution / the pose-only groups' launch above)
    launch_update(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_delta, c->d_xcand,
                  c->d_part_upd, &n_part);
  if (!skip_cand_cost) eval_all(c, c->d_xcand, false, SC_COST_CAND);   // (skipped: the caller evaluates the candidate with Jacobians into the same partials)
  if (!defer_reduce) final_reduce(c);   // (deferred: it rides in the evaluation the caller launches next)
  phase_mark(c, BSGPU_PHASE_CANDIDATE);
}

// ---------------------------------------------------------------------------------------------------
// one LM step = [x <- x_cand] [evaluate J] assemble -> factor -> back-substitute -> candidate -> cost.
// The three variants are captured once per finalized problem as hipGraphs and replayed: the host-side
// launch cost (~4.5 us per kernel, > 100 kernels per step) otherwise bounds the iteration rate.
// ---------------------------------------------------------------------------------------------------

// gradient_only: the iteration budget is used up — the point just accepted still needs its cost and gradient norms for the
// iteration record, but no step will be taken from it: evaluation + assembly (which produces the gradient), no factorisation,
// no candidate
// radius_ahead > 0: the radius the NEXT step will most often be computed at (solve(): an accepted step whose relative decrease is above
// 0.937 divides the radius by max(1/3, ...) = 1/3 — every step of the reference-sized windows, seven of C2's ten): the next step's ASSEMBLY
// (landmark elimination with the step's clearing, camera pairs, pose-only factors: everything up to the factorisation) is queued behind
// the evaluation ahead instead of behind the host's decision (8 us per iteration in which a small window's device idled before its
// landmark launch: stamp over PCIe, LmState::advance, one launch — scripts/small_gaps.sh).  Nothing is decided on the device: the next step
// takes the assembly as its own when the decision is "accepted, at that radius" (bit for bit), and assembles again otherwise — the
// landmark launch clears what the assembly adds into and rewrites all else it wrote.
// lmd (with radius_ahead > 0): the decision is ALSO taken on the device, by the reduction that rides in the landmark launch of the assembly
// ahead (LmDecide, bsgpu_device.h lm_decide): that assembly runs at the radius the host will name — there is no guess to miss — or returns at
// once; radius_ahead itself is then only "an assembly ahead is wanted".  What a large window gains is the candidate's cost-only pass (C2: 9.7 us
// and a launch boundary per iteration), which the smaller windows' guessed assemblies had shed already.
void enqueue_step(bsgpu_ctx* c, const bsgpu_options& o, int kind, double radius, bool gradient_only = false, double radius_ahead = 0.0, const LmDecide* lmd = nullptr) {
  hipStream_t s = c->stream;
  c->cost_x_stale = false;   // (set below only for a step whose reduction rides in the evaluation ahead: a gradient-only step's full reduction gives SC_COST_X)
  // an assembly ahead that the decision did not confirm has rewritten the LM diagonal's clamped H_jj (and the gradient norms' inputs) from
  // the CANDIDATE's Jacobians: this step recomputes them from the current point's (the same bits as before)
  // (... or, decided on the device: the mirror of the reduction the host has just read says "accepted", at this radius bit for bit)
  const bool dev_confirmed = c->spec_dev && c->h_scal[SC_DEC_GO] == 1.0 && c->h_scal[SC_DEC_RADIUS] == radius;
  const bool assembled_ahead = kind == STEP_ACCEPT && !gradient_only && radius > 0.0 && (c->spec_lm_radius == radius || dev_confirmed);
  const bool ahead_unconfirmed = c->spec_dirty && !assembled_ahead;
  if (assembled_ahead && dev_confirmed) c->lm_diag.inv_radius = 1.0 / radius;   // (the assembly configured the LM diagonal's tasks with a placeholder)
  static const bool log_refusals = getenv("BSGPU_TIMING") != nullptr;
  if (log_refusals && c->spec_dev && !assembled_ahead)
    fprintf(stderr, "[bsgpu] device decision not adopted: host kind %d radius %.17g, device go %g radius %.17g\n", kind, radius, c->h_scal[SC_DEC_GO], c->h_scal[SC_DEC_RADIUS]);
  c->spec_lm_radius = 0.0; c->spec_dirty = false; c->spec_dev = false;
  if (kind == STEP_ACCEPT) {
    // the accepted candidate becomes the current point: a pointer swap (every launch takes x as an argument; the next update
    // rewrites all of the other buffer) — except under graph replay, whose kernel arguments are frozen
    if (c->use_graphs) launch_copy(s, c->d_xcand, c->d_x, (int64_t)c->h_x.size(), 0);
    else std::swap(c->d_x, c->d_xcand);
  }
  // Jacobians at the current point: new for a first / accepted step — unless they were evaluated ahead at the candidate that has
  // just been accepted (below) — and to be restored for a rejected one if that evaluation overwrote them
  const bool have_J = (kind == STEP_ACCEPT && c->spec_J) || (kind == STEP_REJECT && !c->spec_J);
  if (!have_J) { eval_all(c, c->d_x, true, SC_COST_X); c->xpart_stale = false; }
  else if (kind == STEP_ACCEPT && c->spec_cand_arrays) c->xpart_stale = true;   // (this point's costs were summed as a candidate's: its own partial arrays hold an older point's)
  c->spec_J = false; c->spec_cand_arrays = false;
  // (an assembly ahead configured the gradient norms that ride in the factorisation for the point that was current THEN: the candidate
  //  it was computed at is the current point now)
  if (assembled_ahead && c->gn_ride.nb > 0) c->gn_ride.x = c->d_x;
  if (!assembled_ahead) assemble(c, o, radius, kind != STEP_REJECT || ahead_unconfirmed, kind == STEP_FIRST, gradient_only, /*factor_follows=*/true);
  if (gradient_only) { c->cost_x_stale = c->xpart_stale; final_reduce(c); return; }
  // With an assembly ahead (radius_ahead) the candidate is evaluated ONCE: with Jacobians, into the candidate's cost partials — the
  // cost-only pass in front of it computed the same residuals — and the step's reduction rides in the assembly's first launch, which follows
  // that evaluation (4.7 us + a launch boundary less per iteration of a reference-sized window, 5.2 of C3's).
  static const bool one_pass_off = getenv("BSGPU_CAND_ONE_PASS") && atoi(getenv("BSGPU_CAND_ONE_PASS")) == 0;
  // (an assembly ahead is adopted only when the LM diagonal and the gradient norms ride in the factorisation — diag_in_chol's static
  //  preconditions — and it zeroes scalar slots the host must already have read through the mirror: without either it would be assembled
  //  and thrown away every iteration, or the host would read a zeroed gradient norm)
  const bool diag_can_ride = c->plan.diag_tasks && c->plan.rider_tasks * 256 >= c->nb && c->d_ftasks && c->d_fsync && c->d_tile_tot && c->d_Winv;
  const bool mirror_ok = c->h_scal_dev != nullptr && c->d_reduce_counter != nullptr && c->n_reduce > 0;
  bool ahead = radius_ahead > 0.0 && !c->use_graphs && !c->use_pcg && !c->use_spcg && c->dense_ok && c->n_pose > 0 && c->idp.n_lm == 0 &&
               diag_can_ride && mirror_ok;
  bool one_pass = ahead && !one_pass_off && reduce_can_ride(c) && (c->vis.n_lm > 0 || (c->n_sa_seg + c->n_asm_grp > 0 && marg_rider(c) < 0));
  // (the decision on the device rides in the landmark launch of a one-pass assembly: without one there is no assembly ahead at all — radius_ahead is no guess)
  const bool dev = lmd != nullptr && lmd->on;
  if (dev && !(one_pass && c->vis.n_lm > 0 && c->d_dec != nullptr)) ahead = one_pass = false;
  // (not on the first step: the reduction that rides cannot give the cost at x — the launch that carries it rewrites those partials — and
  //  only the first step's is read: after an accepted step the cost at x is the candidate's cost the host already holds)
  const bool ride = !gradient_only && kind != STEP_FIRST && !one_pass && reduce_rides(c);
  c->cost_x_stale = ride || c->xpart_stale;
  linear_solve_and_candidate(c, o, ride || one_pass, one_pass);
  // The host now waits for this step's scalars and decides; in the common case (accepted) the next thing the device needs is the
  // residuals and Jacobians at the candidate: evaluated ahead, underneath the host round trip (~26 us per iteration otherwise
  // idle).  A rejected step pays for it with a re-evaluation at the current point (above).
  if (!c->use_graphs) {
    hipEvent_t* const prof = c->prof_events;   // (bsgpu_profile_step times the step up to here: this evaluation belongs to the next one)
    c->prof_events = nullptr;
    if (one_pass) {
      ReduceRide r;
      c->reduce_seq += 1.0;
      r.entries = c->d_reduce; r.n_entries = c->n_reduce; r.n_slots = SC_GRAD_NORM2 + 1; r.scal = c->d_scal; r.host_scal = c->h_scal_dev; r.counter = c->d_reduce_counter;
      r.seq = c->reduce_seq; r.skip_slot = -1;
      // With the decision on the device the reduction is split: only the candidate's cost waits for the evaluation — every other unit rides in the
      // evaluation's launch (as under `ride`), and the landmark launch carries that one unit, which finds the others' scalars done and decides two
      // memory round trips into the launch (all eight units there: the decision came 18 us into a 15 us launch, behind the landmark waves' loads).
      static const bool split_off = getenv("BSGPU_LM_DEVICE_SPLIT") && atoi(getenv("BSGPU_LM_DEVICE_SPLIT")) == 0;
      const bool split = dev && !split_off && reduce_rides(c);
      if (split) {
        ReduceRide re = r;
        re.defer_slot = SC_COST_CAND; re.lmd.on = 1;   // (its mirror unit keeps the factorisation's flag for the decision: SC_CHOL_FAIL_SEEN)
        eval_all(c, c->d_xcand, true, SC_COST_CAND, &re);
        r.only_slot = SC_COST_CAND;
        // (its arrays in the launch's arguments when they fit the unit's one-trip form: at most four, sums, one of up to 4 096 values, the others of up to 256)
        static const bool args_off = getenv("BSGPU_LM_DEVICE_ARGS") && atoi(getenv("BSGPU_LM_DEVICE_ARGS")) == 0;
        int ne = 0, n_big = 0;
        bool fits = !args_off;
        for (const ReduceEntry& en : c->h_reduce) {
          if (en.slot != SC_COST_CAND) continue;
          if (ne == 4 || en.op != 0 || en.n > 4096) { fits = false; break; }
          if (en.n > 256) ++n_big;
          r.early[ne++] = en;
        }
        r.n_early = (fits && n_big <= 1) ? ne : 0;
      } else
        eval_all(c, c->d_xcand, true, SC_COST_CAND);
      c->spec_J = true; c->spec_cand_arrays = true;
      if (dev) {
        const int bank = (c->dec_count++) & 1;   // (consecutive deciding launches take the two banks in turn: each clears the other's)
        r.dec = c->d_dec + (size_t)bank * kDecSlots * kDecStride; r.dec_next = c->d_dec + (size_t)(bank ^ 1) * kDecSlots * kDecStride;
        r.lmd = *lmd;
        r.lmd.radius = radius;
        r.lmd.check_grad = kind != STEP_REJECT ? 1 : 0;                            // (a new point: LmState::advance tests its gradient first)
        r.lmd.x_from_scal = (kind != STEP_REJECT && !c->cost_x_stale) ? 1 : 0;      // (else lmd->x_cost: the cost the host holds for this step's point)
      }
      assemble(c, o, radius_ahead, /*new_J=*/true, /*first=*/false, /*gradient_only=*/false, /*factor_follows=*/true, &r);
      if (c->reduce_carried) { c->scal_mirrored = true; c->seq_pending = true; c->ev_reduce_pending = false; }
      else { c->reduce_seq -= 1.0; final_reduce(c); }   // (no launch of this window's assembly takes it: a launch of its own)
      c->spec_dirty = true;
      c->spec_lm_radius = (c->diag_in_chol && !dev) ? radius_ahead : 0.0;
      c->spec_dev = c->diag_in_chol && dev && lmd->on == 1 && c->reduce_carried;   // (on == 2, the timing probe whose landmark waves took a placeholder radius: never adopted)
      c->prof_events = prof;
      return;
    }
    if (ride) {
      ReduceRide r;
      c->reduce_seq += 1.0;
      r.entries = c->d_reduce; r.n_entries = c->n_reduce; r.n_slots = SC_GRAD_NORM2 + 1; r.scal = c->d_scal; r.host_scal = c->h_scal_dev; r.counter = c->d_reduce_counter;
      r.seq = c->reduce_seq; r.skip_slot = SC_COST_X;
      eval_all(c, c->d_xcand, true, SC_COST_X, &r);
      c->scal_mirrored = true; c->seq_pending = true; c->ev_reduce_pending = false;
    } else
      eval_all(c, c->d_xcand, true, SC_COST_X);
    c->prof_events = prof;
    c->spec_J = true;
    if (ahead) {
      // the next step's assembly as an accepted step at radius_ahead has it (new Jacobians: the ones just evaluated); what the host keeps
      // about THIS step's scalars is not the assembly's to reset
      const bool sm = c->scal_mirrored, sp = c->seq_pending, ep = c->ev_reduce_pending;
      assemble(c, o, radius_ahead, /*new_J=*/true, /*first=*/false, /*gradient_only=*/false, /*factor_follows=*/true);
      c->scal_mirrored = sm; c->seq_pending = sp; c->ev_reduce_pending = ep;
      c->spec_dirty = true;
      // (usable as it is only when the gradient norms ride in the factorisation: a launch of their own would have read the wrong point)
      c->spec_lm_radius = c->diag_in_chol ? radius_ahead : 0.0;
    }
  }
}

// Measurement (bsgpu_profile_step): `reps` full LM steps from the current point — exactly what solve() enqueues for an accepted
// step (Jacobians, assembly, factorisation, back-substitution, candidate, its cost) — with a HIP event at every phase boundary on
// the solver's stream; every step is taken as accepted, at a fixed radius.  ms_out / work_out: BSGPU_PHASE_NUM entries.
int profile_step(bsgpu_ctx* c, const bsgpu_options& o, int reps, double* ms_out, double* work_out) {
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  if (c->use_pcg || !c->dense_ok || c->use_graphs) return fail(c, BSGPU_ERR_UNSUPPORTED, "profile_step: dense Schur path, eager launches only");
  HIPCHK(c, hipSetDevice(c->device));
  c->use_pcg = false;
  const int ne = BSGPU_PHASE_NUM + 1;
  std::vector<hipEvent_t> ev((size_t)reps * ne);
  for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
  const double radius = o.initial_trust_region_radius;
  c->spec_J = false;
  enqueue_step(c, o, STEP_FIRST, radius);   // warm-up: scale / diagonal arrays initialised, a candidate exists
  for (int r = 0; r < reps; ++r) {
    c->spec_J = false;                      // the Jacobian evaluation is part of what is timed
    c->prof_events = ev.data() + (size_t)r * ne;
    (void)hipEventRecord(c->prof_events[0], c->stream);
    enqueue_step(c, o, STEP_ACCEPT, radius);
    c->prof_events = nullptr;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->scal_mirrored = false; c->ev_reduce_pending = false; c->spec_J = false;
  for (int p = 0; p < BSGPU_PHASE_NUM; ++p) ms_out[p] = 0.0;
  for (int r = 0; r < reps; ++r)
    for (int p = 0; p < BSGPU_PHASE_NUM; ++p) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[(size_t)r * ne + p], ev[(size_t)r * ne + p + 1]);
      ms_out[p] += (double)ms / reps;
    }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (work_out) {
    const double n = (double)c->vis.n, ne_ = (double)c->vis.n_elim, nl = (double)c->vis.n_lm, nent = (double)c->vis.n_ent;
    for (int p = 0; p < BSGPU_PHASE_NUM; ++p) work_out[p] = 0.0;
    work_out[BSGPU_PHASE_EVAL_REPROJ] = n * 200.0 + 8.0 * (double)c->h_x.size();      // bytes (bsgpu_reproj_jacobian_bytes)
    // landmark part of J + r in, CR out; per landmark Linv + z + gradient out, scale + LM diagonal in, its range; and the step's CLEARING, which
    // rides in this launch: the tiles of S the assembly adds into (32 KB each), the pose gradient and diag(H)
    // (Visual::no_cr: no C rows — 64 B per observation less written here and read by the back-substitution; the band kernel reads B, 48 B, instead, and Linv + z)
    const double cr = c->vis.no_cr ? 0.0 : 64.0;
    work_out[BSGPU_PHASE_LANDMARK] = ne_ * (48.0 + 16.0 + cr) + nl * (72.0 + 24.0 + 48.0 + 4.0) + 32768.0 * (double)c->n_touched + 16.0 * (double)c->n_pose;
    work_out[BSGPU_PHASE_PAIRS] = ne_ * (96.0 + (c->vis.no_cr ? 48.0 : 64.0) + (c->vis.n_band_units > 0 ? 16.0 : 0.0)) + nent * 8.0 + (c->vis.no_cr ? nl * 72.0 : 0.0);   // every pose part of J and CR row once (the band form: r too) + the entry list
    work_out[BSGPU_PHASE_FACTOR] = c->plan.fused_flops;                               // FP64 flops of the planned factorisation
    work_out[BSGPU_PHASE_BACKSUB] = n * (144.0 + 16.0) + ne_ * cr + nl * 72.0;        // J, r, CR once; Linv + z per landmark
    work_out[BSGPU_PHASE_CANDIDATE] = n * 40.0 + 3.0 * 8.0 * (double)c->h_x.size();   // candidate cost: 40 B / factor + the block update
  }
  // the timed steps moved the point: back to where the context was finalized
  HIPCHK(c, hipMemcpyAsync(c->d_x, c->d_x0, sizeof(double) * c->h_x.size(), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BSGPU_OK;
}

bool same_graph_options(const bsgpu_options& a, const bsgpu_options& b) {
  return a.jacobi_scaling == b.jacobi_scaling && a.min_lm_diagonal == b.min_lm_diagonal && a.max_lm_diagonal == b.max_lm_diagonal &&
         a.linear_solver_type == b.linear_solver_type;
}

void build_graphs(bsgpu_ctx* c, const bsgpu_options& o) {
  if (c->graphs_tried && same_graph_options(o, c->graph_opts)) return;
  c->destroy_graphs();
  c->graphs_tried = true;
  c->graph_opts = o;
  if (!c->use_graphs || c->use_pcg || c->use_spcg) return;   // the PCG path synchronises inside a step: stays eager
  hipGraphExec_t* execs[3] = {&c->g_first, &c->g_accept, &c->g_reject};
  for (int kind = 0; kind < 3; ++kind) {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return; }
    enqueue_step(c, o, kind, 1.0);
    if (hipStreamEndCapture(c->stream, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); c->destroy_graphs(); c->graphs_tried = true; return; }
    const hipError_t e = hipGraphInstantiate(execs[kind], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { (void)hipGetLastError(); c->destroy_graphs(); c->graphs_tried = true; return; }
  }
  c->graphs_ok = true;
  if (getenv("BSGPU_TIMING")) fprintf(stderr, "[bsgpu] LM step captured as hipGraphs\n");
}

void run_step(bsgpu_ctx* c, const bsgpu_options& o, int kind, double radius, bool gradient_only = false, double radius_ahead = 0.0, const LmDecide* lmd = nullptr) {
  if (c->use_graphs) {   // replayed kernels read the radius from device memory
    *c->h_radius = radius;
    (void)hipMemcpyAsync(c->d_scal + SC_RADIUS, c->h_radius, sizeof(double), hipMemcpyHostToDevice, c->stream);
  }
  if (c->graphs_ok && !gradient_only) {
    hipGraphExec_t g = kind == STEP_FIRST ? c->g_first : kind == STEP_ACCEPT ? c->g_accept : c->g_reject;
    if (hipGraphLaunch(g, c->stream) == hipSuccess) return;
    (void)hipGetLastError();
    c->graphs_ok = false;   // fall back to eager launches of the same kernels
  }
  enqueue_step(c, o, kind, radius, gradient_only, radius_ahead, lmd);
}

// sorted visual position -> source factor: built on the host, or downloaded on first use when the device flattened the window
int ensure_vis_src(bsgpu_ctx* c) {
  if ((int)c->vis_src.size() == c->vis.n || !c->d_vis_src) return BSGPU_OK;
  c->vis_src.resize(c->vis.n);
  HIPCHK(c, hipMemcpy(c->vis_src.data(), c->d_vis_src, sizeof(int) * (size_t)c->vis.n, hipMemcpyDeviceToHost));
  return BSGPU_OK;
}

int fetch_scalars(bsgpu_ctx* c) {
  HIPCHK(c, hipGetLastError());  // a kernel that failed to launch must not pass silently
  if (c->scal_mirrored && c->seq_pending) {
    // the step's scalars are complete in the pinned mirror once its stamp shows this reduction's number; kernels queued behind the
    // reduction keep running.  (A device fault never stamps: after two seconds the stream is asked.)
    const volatile double* stamp = &c->h_scal[SC_SEQ];
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    // (a short spin — a step is tens to hundreds of microseconds — then the core is yielded between looks: n windows of a
    // bsgpu_solve_batch would otherwise burn n cores; a large plan's step — seconds for a dense 30 000-dimensional factorisation — ends
    // in a stream synchronisation once the budget, which grows with the plan, is used up)
    const double budget_s = 2.0 + 2e-6 * (double)c->plan.ftasks.size();
    while (__atomic_load_n(reinterpret_cast<const volatile uint64_t*>(stamp), __ATOMIC_ACQUIRE) != *reinterpret_cast<const uint64_t*>(&c->reduce_seq)) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      asm volatile("yield" ::: "memory");
#endif
      ++spins;
      if (spins > 20000 && (spins & 0x3f) == 0) std::this_thread::yield();
      if ((spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budget_s) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (*stamp != c->reduce_seq) return fail(c, BSGPU_ERR_DEVICE, "the end-of-step reduction did not report (device fault?)");
        break;
      }
    }
  } else if (c->scal_mirrored && c->ev_reduce_pending) {
    HIPCHK(c, hipEventSynchronize(c->ev_reduce));   // the step's scalars are in the pinned mirror; kernels queued behind the reduction keep running
  } else {
    if (!c->scal_mirrored) HIPCHK(c, hipMemcpyAsync(c->h_scal, c->d_scal, sizeof(double) * SC_NUM, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->scal_mirrored = false; c->ev_reduce_pending = false; c->seq_pending = false;
  return BSGPU_OK;
}

// the cost of the residual blocks whose parameter blocks are all constant (Ceres: fixed_cost) into SC_FIXED_COST, once per solve, on stream s
// (the partial arrays it borrows are rewritten by the first step's evaluations, which follow on the same stream)
void enqueue_fixed_cost(bsgpu_ctx* c, hipStream_t s) {
  if (!c->any_inactive) return;
  launch_zero(s, c->d_scal + SC_FIXED_COST, 1);
  for (int t = 2; t < kNumInternal; ++t) {
    if (!c->small[t].n) continue;
    SmallGroup g = c->small[t];
    g.active = c->d_small_inactive[t];
    launch_small_eval(s, g, c->d_x, c->d_losses, false, c->d_small_part[t]);
    launch_sum(s, c->d_small_part[t], small_cost_parts(g), c->d_scal + SC_FIXED_COST, 1);
  }
  for (const auto& mc : c->marg) {
    if (mc.active) continue;
    launch_marg_eval(s, mc.dev, c->d_x, false, mc.part);
    launch_sum(s, mc.part, mc.dev.rows, c->d_scal + SC_FIXED_COST, 1);
  }
  if (c->vis_any_inactive) {   // reprojection factors whose three blocks are all constant
    launch_reproj_eval(s, c->vis, c->d_x, c->d_cams, c->d_losses, false, c->vis.cost_part_cand, true);
    launch_sum(s, c->vis.cost_part_cand, c->vis.n_cost_part, c->d_scal + SC_FIXED_COST, 1);
  }
}

// ---------------------------------------------------------------------------------------------------
// [EXT] ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy, restated
// ---------------------------------------------------------------------------------------------------
constexpr int kAssemblyAheadMaxResiduals = 400000;   // ... and residual rows in all (C3: 121 500; C2: 804 000)
constexpr int kAssemblyAheadMaxFactors = 150000;   // reprojection factors up to which the next step's assembly is issued ahead of the decision
int solve(bsgpu_ctx* c, const bsgpu_options& o, bsgpu_summary& sum) {
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clk::now() - t_start).count(); };
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  std::memset(&sum, 0, sizeof(sum));
  c->iters.clear();
  c->spec_lm_radius = 0.0; c->spec_dirty = false; c->spec_dev = false;
  sum.num_parameters_tangent = c->n_tan;
  sum.num_residuals = c->n_res;
  c->use_pcg = (o.linear_solver_type == BSGPU_LINEAR_PCG) || (o.linear_solver_type == BSGPU_LINEAR_AUTO && !c->dense_ok);
  if (!c->use_pcg && !c->dense_ok)
    return fail(c, BSGPU_ERR_UNSUPPORTED, "reduced system above the limit of the tiled exact path (12288 pose-only, 49152 with landmarks); use BSGPU_LINEAR_AUTO or BSGPU_LINEAR_PCG");
  if (c->use_pcg) { rc = build_bsr(c); if (rc != BSGPU_OK) return rc; }
  c->use_spcg = !c->use_pcg && o.linear_solver_type == BSGPU_LINEAR_SCHUR_PCG;
  if (c->use_spcg) { rc = build_spcg(c); if (rc != BSGPU_OK) return rc; }
  c->pcg_iters_total = 0;
  sum.linear_solver_used = c->use_pcg ? BSGPU_LINEAR_PCG : c->use_spcg ? BSGPU_LINEAR_SCHUR_PCG : BSGPU_LINEAR_SCHUR_CHOLESKY;
  hipStream_t s = c->stream;
  // (the solve's two timing events are the context's: a pair created per solve was never destroyed on the error returns below)
  if (!c->ev_solve0) HIPCHK(c, hipEventCreate(&c->ev_solve0));
  if (!c->ev_solve1) HIPCHK(c, hipEventCreate(&c->ev_solve1));
  const hipEvent_t ev0 = c->ev_solve0, ev1 = c->ev_solve1;
  HIPCHK(c, hipEventRecord(ev0, s));

  // iteration zero
  double fixed = 0.0;
  enqueue_fixed_cost(c, s);
  build_graphs(c, o);
  // the trust-region loop itself: lm_state.h (one copy, shared with bsgpu_solve_batch); this driver computes the steps it asks for
  LmState lm;
  const bsgpu_summary head = sum;   // (what was filled in above: start() clears the summary)
  lm.start(&o, &sum, &c->iters, head.num_parameters_tangent, head.num_residuals, head.linear_solver_used);
  lm.t_start = t_start;
  // the assembly of the step after this one goes out ahead of the decision (enqueue_step) where it is short — a wrong guess costs its
  // length: nothing on the reference's window sizes, C2's three guesses in ten that miss cost more than the seven that hit gain (measured)
  static const int ahead_env = getenv("BSGPU_LM_AHEAD") ? atoi(getenv("BSGPU_LM_AHEAD")) : -1;
  const bool lm_ahead = ahead_env >= 0 ? ahead_env != 0 : (c->vis.n <= kAssemblyAheadMaxFactors && c->n_res <= kAssemblyAheadMaxResiduals);   // (pose-only windows too: their assembly takes no radius, only the guess "accepted")
  // (the radius of LmState::advance for a relative decrease above 0.937, in its own arithmetic: r / (1/3) is not 3 r in every last bit)
  // ... and only while the guesses hold: after a step that did not end "accepted, at the guessed radius" (a pose graph's early steps, C2's
  // fifth to seventh) the next assembly waits for the decision again, until a step ends that way
  // Windows above those sizes: the assembly ahead at the radius the DEVICE decides (enqueue_step, LmDecide) — no guess that can miss, and the
  // candidate's cost-only pass goes.  BSGPU_LM_DEVICE=0: never, 1: on the smaller windows too (instead of their guesses).
  static const int dev_env = getenv("BSGPU_LM_DEVICE") ? atoi(getenv("BSGPU_LM_DEVICE")) : -1;
  const bool dev_possible = dev_env != 0 && ahead_env != 0 && c->vis.n_lm > 0 && c->d_dec != nullptr && !c->use_graphs && !c->use_pcg && !c->use_spcg;
  // (a smaller window whose last guess missed waits for the host's decision until a step ends as guessed again; the device deciding for it
  //  meanwhile measured the same — scripts/rejected_steps.py, 50 KF x 5 000 with four rejected steps in twenty: 6 700 LM it/s either way)
  const bool lm_dev = dev_possible && (dev_env > 0 || !lm_ahead);
  LmDecide lmd;
  lmd.on = dev_possible ? (getenv("BSGPU_LM_DEVICE_NOWAIT") ? 2 : 1) : 0;   // (2: a timing probe — the landmark waves do not wait, the assembly is never adopted)
  lmd.min_relative_decrease = o.min_relative_decrease; lmd.max_radius = o.max_trust_region_radius; lmd.function_tolerance = o.function_tolerance;
  lmd.parameter_tolerance = o.parameter_tolerance; lmd.gradient_tolerance = o.gradient_tolerance;
  bool guess_held = true;
  double rel_prev = 0.0, rel_last = 0.0;   // relative cost changes of the last two accepted steps
  double guessed = 0.0;   // what the step in flight was guessed to end at (0: nothing was guessed)
  auto radius_ahead = [&](double r) {
    guessed = std::min(o.max_trust_region_radius, r / std::max(1.0 / 3.0, 0.0));
    if (lm_dev) return r > o.min_trust_region_radius ? r : 0.0;   // (no guess: "an assembly ahead is wanted")
    return (lm_ahead && guess_held && !c->use_graphs && !c->use_pcg) ? guessed : 0.0;
  };
  run_step(c, o, STEP_FIRST, lm.radius, false, radius_ahead(lm.radius), lm_dev ? &lmd : nullptr);
  rc = fetch_scalars(c);
  if (rc != BSGPU_OK) { (void)pcg_check(c); return rc; }
  bool pcg_redo = !pcg_check(c);
  if (pcg_redo) c->h_scal[SC_CHOL_FAIL] = 2.0;   // (lm_state.h: the step is wanted again, at the same point and radius)
  fixed = c->any_inactive ? c->h_scal[SC_FIXED_COST] : 0.0;
  lm.begin(c->h_scal, fixed, c->d_ftasks != nullptr || pcg_redo);
  while (!lm.done) {
    if (lm.retry_timeout && !c->use_pcg) {
      // A wait inside one of the single-launch kernels (factorisation / back-substitution) timed out: the GPU is shared and their
      // workgroups were not scheduled in time — not a numerical failure.  This context takes the launch-per-step path from
      // here on, and the step is computed again at the same point and radius.
      c->d_ftasks = nullptr;
      if (getenv("BSGPU_TIMING")) fprintf(stderr, "[bsgpu] single-launch Cholesky timed out: launch-per-step path from here on\n");
      if (c->use_graphs) { c->destroy_graphs(); build_graphs(c, o); }   // (the captured sequences still hold the single-launch kernel)
    }
    guess_held = lm.kind == STEP_ACCEPT && lm.radius == guessed;
    // ... and not for a step that will most likely not be taken: the one after the iteration budget's last (a gradient-only step follows),
    // or after a step whose cost change — the last one times the ratio of the last two — will be under the function tolerance (a solve
    // that ends that way left an unused assembly in the queue, 16 us of a reference-sized window's 640: the geometric guess names five of
    // the bench windows' six last steps)
    if (lm.it.step_is_successful && lm.it.iteration > 0) { rel_prev = rel_last; rel_last = std::fabs(lm.it.cost_change) / std::max(1e-300, std::fabs(lm.x_cost)); }
    const bool likely_last = lm.it.iteration + 1 >= o.max_num_iterations ||
                             (rel_prev > 0.0 && rel_last * std::min(1.0, rel_last / rel_prev) < 4.0 * o.function_tolerance);
    lmd.x_cost = lm.kind == STEP_REJECT ? lm.x_cost : lm.cand_cost;   // (the cost at the point this step is computed at, where the host holds it: enqueue_step)
    run_step(c, o, lm.kind, lm.radius, lm.grad_only, (lm.grad_only || likely_last) ? 0.0 : radius_ahead(lm.radius), lm_dev ? &lmd : nullptr);
    rc = fetch_scalars(c);
    if (rc != BSGPU_OK) { (void)pcg_check(c); return rc; }
    pcg_redo = !pcg_check(c);
    if (pcg_redo) c->h_scal[SC_CHOL_FAIL] = 2.0;
    lm.advance(c->h_scal, c->cost_x_stale, c->d_ftasks != nullptr || pcg_redo);
  }
  HIPCHK(c, hipEventRecord(ev1, s));
  HIPCHK(c, hipEventSynchronize(ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  sum.device_time_in_seconds = ms * 1e-3;
  c->scal_mirrored = false; c->ev_reduce_pending = false; c->spec_J = false; c->spec_lm_radius = 0.0; c->spec_dirty = false; c->spec_dev = false;   // (the stream has drained: nothing of this solve is pending)
  sum.num_inner_iterations = c->pcg_iters_total;
  sum.total_time_in_seconds = elapsed();
  return BSGPU_OK;
}

}  // namespace bsg
