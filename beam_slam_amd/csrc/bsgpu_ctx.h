// Host-side state of one libbsgpu context and the internal entry points of the three host translation units:
//   bsgpu_finalize.cpp  flattening of the described problem to device tables ([EXT] fuse HashGraph::createProblem)
//   bsgpu_solve.cpp     the LM loop and its step ([EXT] ceres TrustRegionMinimizer + LevenbergMarquardtStrategy)
//   bsgpu_api.cpp       the C-ABI of include/bsgpu.h
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bsgpu.h"
#include "bsgpu_internal.h"
#include "dense_plan.h"


namespace bsg {


struct TypeInfo { int nidx, nvar, nconst, m; int amb[10]; };
// internal group: reprojection factors whose landmark block is NOT eliminated (it also appears in another kind of
// factor, e.g. a marginal prior): they are evaluated and assembled like the pose-only groups, slots (q, p, P)
constexpr int T_REPROJ_DENSE = BSGPU_F_NUM_TYPES;
constexpr int kNumInternal = BSGPU_F_NUM_TYPES + 1;
const TypeInfo kTypes[kNumInternal] = {
    {4, 3, 3, 2, {4, 3, 3}},
    {6, 5, 3, 2, {4, 3, 3, 4, 3}},
    {10, 10, 287, 15, {4, 3, 3, 3, 3, 4, 3, 3, 3, 3}},
    {5, 5, 241, 15, {4, 3, 3, 3, 3}},
    {6, 6, 43, 6, {3, 4, 3, 4, 3, 4}},
    {4, 4, 43, 6, {3, 4, 3, 4}},
    {2, 2, 43, 6, {3, 4}},
    {1, 1, 12, 3, {3}},
    {2, 2, 12, 3, {3, 3}},
    {1, 1, 7, 2, {4}},
    {6, 5, 6, 2, {4, 3, 4, 3, 1}},
    {4, 3, 6, 2, {4, 3, 1}},
    {4, 3, 3, 2, {4, 3, 3}},   // T_REPROJ_DENSE: idx q, p, P, (derived) camera; consts u, v, w
};
inline bool has_camera(int t) { return t <= 1 || t == BSGPU_F_IDP_REPROJ || t == BSGPU_F_IDP_REPROJ_UNARY || t == T_REPROJ_DENSE; }


struct HostMarginal {
  std::vector<int32_t> blocks;
  int rows = 0, cols = 0;
  std::vector<double> A, b, xbar;
};

// Largest reduced (pose-side) system the dense tiled Cholesky takes: the back-substitution keeps the whole solution vector in LDS
// (k_chol.hip: sy[npad] next to the 64x64 tiles, 160 KB per CU).  C2 needs 3 008; 12 288 = 819 keyframes of 15-d states.
constexpr size_t kDenseLimit = 12288;
// Windows WITH eliminated landmarks have no other exact path (the block-sparse PCG covers pose-only graphs), and their reduced system
// is block-banded: above kDenseLimit the same tiled factorisation runs with the back-substitution's solution vector in global
// memory instead of LDS (slower per panel, no size cliff).  The bound is what the dense tile storage costs: 2 x npad^2 doubles
// (S and its shadow) = 2 x 19 GB at 49 152 — HBM is 288 GB.
constexpr size_t kDenseLimitLandmarks = 49152;
// windows with at least this many reprojection factors are flattened on the device (k_flatten.hip)
constexpr int kDeviceFlattenMin = 20000;

struct HostGroup {
  int n = 0;
  std::vector<int32_t> idx;
  std::vector<double> consts;
  std::vector<int32_t> loss_kind;
  std::vector<double> loss_a;
};

// The reprojection table of a caller that keeps it across solves and reports what changed (bsgpu_sync_factors_indirect,
// SURVEY.md §8f rank 2): a host and a device copy NAMED BY CALLER SLOTS that outlive bsgpu_clear() and are patched row by row,
// with per-slot use counts so that finalize() validates and classifies the blocks without a pass over 400 k rows.  The
// block-named host copy of the group (groups[BSGPU_F_REPROJ].idx ...) is only materialised for the paths that need it
// (host flattening, marginalisation).
struct SlotMirror {
  bool valid = false;          // holds the table of the last sync call
  bool active = false;         // the current description's group is this table (bsgpu_clear resets)
  bool materialized = false;   // groups[BSGPU_F_REPROJ] holds the translated copy
  int n = 0, n_slots = 0;
  std::vector<int32_t> idx;    // n x 4: q slot, p slot, landmark slot, camera id
  std::vector<double> consts;  // n x 3
  std::vector<int32_t> loss_kind;
  std::vector<double> loss_a;
  std::vector<int32_t> use_q, use_p, use_l;   // per slot: rows naming it as orientation / position / landmark
  std::vector<int32_t> cam_use;               // per camera id
  struct LossUse { int kind; double a; int64_t rows; };
  std::vector<LossUse> loss_use;              // distinct losses of the rows
  std::vector<int32_t> s2b;                   // slot -> block of the current description
  // device copy (own allocations: they live across finalize() calls)
  int32_t* d_idx = nullptr; double* d_consts = nullptr; int32_t* d_lk = nullptr; double* d_la = nullptr;
  size_t d_cap = 0;
  int32_t* d_s2b = nullptr; size_t d_s2b_cap = 0;
  unsigned char* h_stage = nullptr; unsigned char* d_stage = nullptr; size_t stage_cap = 0;   // packed changed rows (pinned / device)
  bool dev_valid = false;
  void release_device() {
    if (d_idx) (void)hipFree(d_idx);
    if (d_consts) (void)hipFree(d_consts);
    if (d_lk) (void)hipFree(d_lk);
    if (d_la) (void)hipFree(d_la);
    if (d_s2b) (void)hipFree(d_s2b);
    if (d_stage) (void)hipFree(d_stage);
    if (h_stage) (void)hipHostFree(h_stage);
    d_idx = nullptr; d_consts = nullptr; d_lk = nullptr; d_la = nullptr; d_s2b = nullptr; d_stage = nullptr; h_stage = nullptr;
    d_cap = d_s2b_cap = stage_cap = 0; dev_valid = false;
  }
};


}  // namespace bsg

using namespace bsg;   // (internal header: included by the three host translation units only)

struct bsgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // ---- host copy of the problem
  int nb = 0;
  std::vector<double> h_x;
  std::vector<int32_t> off;
  std::vector<uint8_t> size, manifold, is_const, is_const_in;   // is_const_in: as given; is_const: + blocks no factor touches
  std::vector<bsgpu_camera> cams;
  HostGroup groups[kNumInternal];
  SlotMirror mirror0;            // BSGPU_F_REPROJ through bsgpu_sync_factors_indirect
  bool finalized = false;
  uint64_t finalize_gen = 0;     // counts finalize() runs: cached per-window argument tables (bsgpu_batch.cpp) are stale when it moves
  // ---- derived structure
  std::vector<int> tsize, toff;
  std::vector<uint8_t> is_lm;
  int n_pose = 0, n_lm = 0, n_tan = 0, npad = 0, n_res = 0;
  int row0[BSGPU_F_NUM_TYPES] = {0};
  bool vis_any_inactive = false; // some reprojection factor has q, p and landmark all constant
  std::vector<int> vis_src;      // sorted visual position -> (type<<28 | index in its host group)
  int* d_vis_src = nullptr;      // ... on the device when the window was flattened there (downloaded on demand)
  std::vector<HostMarginal> marginals;
  struct MargCtx { MargDev dev; int row0 = 0; bool active = true; double *part = nullptr, *part_cand = nullptr, *part_mcc = nullptr; };
  std::vector<MargCtx> marg;
  struct MargResult { std::vector<int32_t> kept; int rows = 0, cols = 0; std::vector<double> A, b, xbar; bool valid = false; } marg_result;
  std::vector<int> dense_src;    // T_REPROJ_DENSE factor -> (type<<28 | index in its host group)
  std::vector<uint8_t> no_elim;  // per block: never Schur-eliminate (set by the marginalisation sub-problem)
  bool any_inactive = false;
  // ---- device
  std::vector<std::pair<void*, size_t>> allocs;   // live device buffers (pointer, bytes)
  std::multimap<size_t, void*> pool;              // released buffers kept for the next finalize()
  size_t pool_bytes = 0;
  double *d_x = nullptr, *d_xcand = nullptr, *d_x0 = nullptr;
  int *d_blk_xoff = nullptr, *d_blk_toff = nullptr;
  unsigned char *d_blk_size = nullptr, *d_blk_manifold = nullptr;
  DevCamera* d_cams = nullptr;
  DevLoss* d_losses = nullptr;
  Visual vis;
  IdpElim idp;                   // inverse-depth landmarks eliminated on the landmark side (bsgpu_finalize.cpp, k_idp.hip)
  int n_idp_lm = 0;
  SmallGroup small[kNumInternal];
  std::vector<unsigned char> h_small_active[kNumInternal];
  unsigned char* d_small_inactive[kNumInternal] = {nullptr};
  double* d_small_part[kNumInternal] = {nullptr};       // per-factor cost at the current point
  double* d_small_part_cand[kNumInternal] = {nullptr};  // ... at the candidate
  double* d_small_part_mcc[kNumInternal] = {nullptr};   // per-row model-cost-change terms
  // pose-only factors assembled without per-factor atomics: every 3x3 block (row block, column block) of J^T J they touch is a
  // SEGMENT of (factor, slot a, slot b) contributions, summed by one wave in a fixed order (k_small.hip small_assemble_seg_kernel)
  SmallGroup* d_small_groups = nullptr;   // the groups, as the kernel indexes them by internal type
  SmallGroup small_factorwise[kNumInternal];   // the groups assembled one workgroup per factor instead (n = 0 for the others)
  int n_sa_seg = 0;
  int *d_sa_seg_start = nullptr, *d_sa_seg_ra = nullptr, *d_sa_seg_rb = nullptr;
  int2* d_sa_contrib = nullptr;
  int n_asm_grp = 0;             // groups of factors with identical slots (AsmGroup), assembled a wave per group
  AsmGroup* d_asm_grp = nullptr;
  ReduceEntry* d_reduce = nullptr;
  int n_reduce = 0;
  double* d_part_upd = nullptr;
  double* d_gpart = nullptr;     // grad_norms_kernel: (max, sum of squares) per workgroup
  int n_gpart = 0;
  int n_part_upd = 0;
  // the update riding in the landmark back-substitution (windows with eliminated Euclidean landmarks): UpdateRide's tables
  int n_upd_blocks = 0;
  bool pre_cleared = false;      // the reduced system, gradient and diagonal were cleared at the end of the previous step (bsgpu_solve.cpp)
  bool upd_in_mcc = false;       // no Euclidean landmarks: the update of every block rides in the pose-only model-cost launch (128 blocks per unit)
  int* d_upd_blocks = nullptr;
  int* d_lm_xoff = nullptr;
  double *d_S = nullptr, *d_grad = nullptr, *d_hdiag = nullptr, *d_scale = nullptr, *d_dcl = nullptr;
  double *d_delta = nullptr, *d_y = nullptr, *d_scal = nullptr, *d_part = nullptr;
  double* h_scal = nullptr;  // pinned
  double* h_scal_dev = nullptr;  // the same buffer as the device sees it (final_reduce mirrors the step's scalars there)
  // the dissection cost model that won this context's last comparison (bsgpu_finalize.cpp: small systems are planned under several and keep the
  // shortest replay): a window that slides keeps its shape, so the next finalizes plan under the winner alone and compare again every 32nd time
  int dim_model = -1, dim_model_age = 0, dim_model_npose = -1;
  int plan_pref = 0;   // BSGPU_PLAN_LATENCY / BSGPU_PLAN_THROUGHPUT (bsgpu_set_plan_preference)
  bool spec_dev = false;         // ... for an accepted step at the radius the DEVICE decided (LmDecide): adopted when the host's decision names the same
  unsigned dec_count = 0;        // deciding launches so far (their bank of d_dec: the count's parity)
  double* d_dec = nullptr;       // two banks of kDecSlots x kDecStride doubles: the decision of a riding reduction, for the workgroups and launches of the assembly ahead
  double spec_lm_radius = 0.0;   // != 0: the next step's assembly is in the queue already, for an accepted step at this radius (bsgpu_solve.cpp enqueue_step)
  bool spec_cand_arrays = false; // the evaluation ahead wrote its cost partials into the candidate's arrays (it replaced the cost-only pass)
  bool reduce_carried = false;   // assemble(): the reduction it was handed rode in one of its launches
  bool xpart_stale = false;      // the cost partials of the CURRENT point are those of an older one (the point was a candidate evaluated into the candidate's arrays)
  bool spec_dirty = false;       // an assembly ahead has run since the last one a step took as its own
  bool scal_mirrored = false;    // the last enqueued work ended with a final_reduce that filled the mirror
  // tiled Cholesky plan (dense_plan.h) and its device tables
  DensePlan plan;
  std::vector<uint8_t> tile_adj;   // natural-tile adjacency of the reduced system
  int *d_dpos = nullptr, *d_inat = nullptr;   // DensePlan::dpos / inat on the device: tangent index -> position in S and back
  int *d_nreal = nullptr, *d_rows_flat = nullptr;
  PanelDesc* d_panels = nullptr;
  int *d_bs_desc_chain = nullptr, *d_rows_flat_chain = nullptr, *d_bs_upd = nullptr, *d_bs_upd_rows = nullptr;
  int *d_bs_chain_group = nullptr, *d_bs_grp_nchains = nullptr, *d_bs_grp_nitems = nullptr, *d_bs_items4 = nullptr, *d_bs_tile_updated = nullptr, *d_bs_sync = nullptr, *d_bs_order = nullptr;
  int *d_bs_desc = nullptr, *d_chain_begin = nullptr, *d_chain_end = nullptr, *d_tile_sync = nullptr, *d_touched = nullptr;
  int n_touched = 0;
  int* d_tile_tot = nullptr;
  FusedTask* d_ftasks = nullptr;   // fused single-launch factorisation: task list and its counters (k_chol.hip chol_fused_kernel)
  FusedTask* d_ftasks_rows = nullptr; int n_ftasks_rows = 0;   // ... with row segments (dense_plan.h ftasks_rows; the segments' pairs lie behind the list; tile counts: plan.frows_src's)
  FusedTask* d_ftasks_bulk = nullptr; int* d_tile_tot_bulk = nullptr; int n_ftasks_bulk = 0;      // ... and without the K-chunks (dense_plan.h ftasks_bulk)
  FusedTask* d_ftasks_plain = nullptr; int* d_tile_tot_plain = nullptr; int n_ftasks_plain = 0;   // ... the list without the diagonal / rider tasks (dense_plan.h)
  int* d_fsync = nullptr;
  double* d_Vinv = nullptr;
  double* d_Winv = nullptr;
  double* d_Lp = nullptr;     // shadow of S holding the off-diagonal L panels (k_chol.hip)
  double* d_ytan = nullptr;   // y in tangent order
  std::vector<bsgpu_iteration> iters;
  // captured LM-step sequences (hipGraph): iteration zero / after an accepted step / after a rejected step
  hipGraphExec_t g_first = nullptr, g_accept = nullptr, g_reject = nullptr;
  bool graphs_tried = false, graphs_ok = false, use_graphs = true;
  bsgpu_options graph_opts{};
  double* h_radius = nullptr;  // pinned
  double* h_pcg = nullptr;     // pinned: two read-backs of the PCG scalars in flight (pcg_solve)
  hipEvent_t pcg_ev[2] = {nullptr, nullptr};
  double* h_pcg_lazy = nullptr;   // pinned: the resident PCG launch's scalars, looked at with the step's other scalars (pcg_check)
  bool pcg_check_pending = false;
  int pcg_verdicts_seen = 0;      // (BSGPU_PCG_GIVE_UP test hook: verdicts read by THIS context)
  hipEvent_t ev_reduce = nullptr;   // recorded after the end-of-step reduction: what the host waits for (work may be queued behind it)
  hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;   // bsgpu_solve's timing events (created on first use)
  bool ev_reduce_pending = false;
  std::vector<ReduceEntry> h_reduce;   // the reduction table as uploaded (d_reduce): a deciding unit takes its arrays from here, in the launch's arguments
  int* d_reduce_counter = nullptr;  // final_reduce_kernel's ticket (the last workgroup stamps the host mirror with reduce_seq)
  double reduce_seq = 0.0;          // sequence number of the last end-of-step reduction enqueued
  bool seq_pending = false;         // ... and the host may poll for it instead of waiting for an event
  // block-sparse PCG path
  std::vector<uint8_t> leaf_tile;   // per natural tile of the reduced system: made of inverse-depth landmarks only (finalize)
  int n_leaf_tiles = 0;
  bool dense_ok = true, bsr_built = false, use_pcg = false;
  // PCG on the assembled reduced camera system (BSGPU_LINEAR_SCHUR_PCG): tile rows of S in CSR form, inverses of the diagonal tiles,
  // the CG vectors in solver order (built on first use, build_spcg())
  bool use_spcg = false, spcg_built = false;
  int n_schunks = 0;
  int *d_schunk_row = nullptr, *d_schunk_ptr = nullptr, *d_srow_chunk_ptr = nullptr, *d_tcol = nullptr;
  double *d_sMinv = nullptr, *d_sx = nullptr, *d_sr = nullptr, *d_sz = nullptr, *d_sp0 = nullptr, *d_sp1 = nullptr, *d_sq = nullptr,
         *d_spart_pq = nullptr, *d_spart = nullptr, *d_ssc = nullptr;
  // in-situ phase timing (bsgpu_profile_step): when non-null, the step records one of these events at every phase boundary
  hipEvent_t* prof_events = nullptr;
  // this step's LM diagonal and gradient norms are carried by the factorisation's launch (its diagonal / rider tasks, dense_plan.h) instead
  // of a launch of their own between the assembly and the factorisation — decided per step by assemble()
  bool diag_in_chol = false;
  bool cost_x_stale = false;   // the last step's reduction rode in the evaluation ahead and left SC_COST_X alone (that launch rewrites its partials)
  LmDiag lm_diag;
  GradNormRide gn_ride;
  bool spec_J = false;   // residuals + Jacobians currently hold the CANDIDATE's (evaluated ahead of the accept/reject decision)
  int nbr = 0, nblk = 0, pcg_iters_total = 0;
  PcgPersistDev pcg_persist;     // G = 0: the launch-per-iteration path only
  int *d_row_ptr = nullptr, *d_col = nullptr, *d_diag_slot = nullptr, *d_pair_slot = nullptr;
  int* d_slots[kNumInternal] = {nullptr};
  int n_bsr_seg = 0;               // shared blocks of the block-sparse system, by segments (k_pcg.hip bsr_assemble_seg_kernel)
  int *d_bsr_seg_start = nullptr, *d_bsr_seg_slot = nullptr, *d_bsr_seg_row = nullptr;
  int2* d_bsr_contrib = nullptr;
  double *d_val = nullptr, *d_Minv = nullptr, *d_rhs = nullptr, *d_px = nullptr, *d_pr = nullptr, *d_pz = nullptr, *d_pp = nullptr, *d_pp1 = nullptr,
         *d_pq = nullptr, *d_ppart = nullptr, *d_ppart2 = nullptr, *d_psc = nullptr;

  // device buffers are pooled across finalize() calls: a sliding window re-flattens every cycle with nearly the same
  // sizes, and hipMalloc / hipFree (which synchronise) would otherwise cost milliseconds per cycle
  template <typename T> T* alloc(size_t n) {
    void* p = nullptr;
    if (n == 0) n = 1;
    const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    auto it = pool.lower_bound(bytes);
    if (it != pool.end() && it->first <= bytes + bytes / 2 + 4096) { p = it->second; pool_bytes -= it->first; const size_t got = it->first; pool.erase(it); allocs.push_back({p, got}); return static_cast<T*>(p); }
    // a window that slides grows and shrinks by a fraction of a percent per cycle: headroom on the larger buffers, so that the
    // next cycle's slightly larger request finds this one in the pool instead of going to hipMalloc again (HBM is not scarce)
    size_t want = bytes > ((size_t)64 << 10) ? ((bytes + bytes / 8 + 255) & ~(size_t)255) : bytes;
    if (hipMalloc(&p, want) != hipSuccess) {
      release_pool();   // give cached buffers back and retry once, without the headroom
      want = bytes;
      if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    }
    allocs.push_back({p, want});
    return static_cast<T*>(p);
  }
  void release_pool() {
    for (auto& kv : pool) (void)hipFree(kv.second);
    pool.clear(); pool_bytes = 0;
  }
  // The ~40 small tables of a finalize() go through a pinned staging arena as asynchronous copies on the context's stream (every
  // consumer is a kernel on that stream): a blocking hipMemcpy each cost ~15 us of round trip.  The arena is rewound when the
  // stream has drained (free_device); what does not fit takes the blocking path.
  unsigned char* h_arena = nullptr;
  size_t arena_cap = 0, arena_off = 0;
  template <typename T> T* upload(const std::vector<T>& v) {
    T* p = alloc<T>(v.size());
    if (!p || v.empty()) return p;
    const size_t bytes = v.size() * sizeof(T), need = (bytes + 63) & ~(size_t)63;
    if (!h_arena && stream) {
      if (hipHostMalloc((void**)&h_arena, (size_t)16 << 20) == hipSuccess) arena_cap = (size_t)16 << 20; else { (void)hipGetLastError(); h_arena = nullptr; }
    }
    if (h_arena && stream && bytes <= ((size_t)4 << 20) && arena_off + need <= arena_cap) {
      std::memcpy(h_arena + arena_off, v.data(), bytes);
      if (hipMemcpyAsync(p, h_arena + arena_off, bytes, hipMemcpyHostToDevice, stream) == hipSuccess) { arena_off += need; return p; }
      (void)hipGetLastError();
    }
    if (stream) (void)hipStreamSynchronize(stream);   // (keeps the order with what is queued on the stream)
    (void)hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice);
    return p;
  }
  void free_device() {
    if (stream) (void)hipStreamSynchronize(stream);   // nothing may still be using the buffers that go back to the pool
    arena_off = 0;
    for (auto& a : allocs) { pool.emplace(a.second, a.first); pool_bytes += a.second; }
    allocs.clear();
    if (pool_bytes > ((size_t)8 << 30)) release_pool();
    vis = Visual();
    idp = IdpElim();
    for (auto& g : small) g = SmallGroup();
    d_x = d_xcand = d_x0 = nullptr;
    bsr_built = false; spcg_built = false;
    destroy_graphs();
  }
  void destroy_graphs() {
    for (hipGraphExec_t* g : {&g_first, &g_accept, &g_reject}) if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
    graphs_tried = graphs_ok = false;
  }
};


namespace bsg {

// phases of one LM step, in enqueue order (include/bsgpu.h BSGPU_PHASE_*)
inline void phase_mark(bsgpu_ctx* c, int phase) { if (c->prof_events) (void)hipEventRecord(c->prof_events[phase + 1], c->stream); }
inline int fail(bsgpu_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }
int api_exception(bsgpu_ctx* c) noexcept;   // bsgpu_api.cpp: where every entry point's function-try-block ends

#define HIPCHK(c, call)                                                                         \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, BSGPU_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// bsgpu_api.cpp
int materialize_mirror(bsgpu_ctx* c);   // block-named host copy of the mirrored reprojection group
// bsgpu_finalize.cpp
int finalize(bsgpu_ctx* c);
int build_bsr(bsgpu_ctx* c);
int build_spcg(bsgpu_ctx* c);
// bsgpu_solve.cpp
int solve(bsgpu_ctx* c, const bsgpu_options& o, bsgpu_summary& sum);
void enqueue_fixed_cost(bsgpu_ctx* c, hipStream_t s);
void eval_all(bsgpu_ctx* c, const double* x, bool with_J, int slot, const ReduceRide* red = nullptr);
// factor_follows: linear_solve_and_candidate() comes next — its factorisation launch may then carry the LM diagonal and the gradient norms
void assemble(bsgpu_ctx* c, const bsgpu_options& o, double radius, bool new_J, bool first, bool gradient_only = false, bool factor_follows = false,
              const ReduceRide* red = nullptr /* the step before's end-of-step reduction, to ride in the assembly's first launch (sets c->reduce_carried) */);
void final_reduce(bsgpu_ctx* c);
int fetch_scalars(bsgpu_ctx* c);
int ensure_vis_src(bsgpu_ctx* c);
// Cholesky of the (padded, rhs-augmented, solver-ordered) reduced system in S and the solve L^T y = y', following the plan's
// step schedule; y comes back in solver order (npad entries)
// (struct DenseDev: bsgpu_internal.h)
void dense_factor(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* scal);
// (iperm / y_tan / delta given: the back-substitution also writes the solution in tangent order and the step -y)
void dense_backsolve(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* y, const int* iperm = nullptr,
                     int n_pose = 0, double* y_tan = nullptr, double* delta = nullptr);
void dense_factor_solve(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* y, double* scal, const int* iperm = nullptr,
                        int n_pose = 0, double* y_tan = nullptr, double* delta = nullptr);
int profile_step(bsgpu_ctx* c, const bsgpu_options& o, int reps, double* ms_out, double* work_out);
// bsgpu_batch.cpp: several windows by one set of launches per LM iteration
bool batch_covers(bsgpu_ctx* c, const bsgpu_options& o);
void batch_stats(int64_t* windows, int64_t* rounds);
bool solve_batched(bsgpu_ctx* const* ctxs, const int* idx, int m, const bsgpu_options* o, int options_stride, bsgpu_summary* s, int* rc);
void batch_forget(const bsgpu_ctx* c);   // a context is going away: cached argument tables that name it are dropped

}  // namespace bsg
