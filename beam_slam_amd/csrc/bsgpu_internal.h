// Internal declarations shared by the host driver (bsgpu_finalize.cpp, bsgpu_solve.cpp, bsgpu_api.cpp) and the HIP kernels
// (bsgpu_kernels.hip) of libbsgpu.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <cstring>
#include <vector>

#include "dense_plan.h"

namespace bsg {

// ---- device-side tables -------------------------------------------------------------------------
struct DevCamera {  // 16 doubles
  double fx, fy, cx, cy;
  double R[9];  // R_cam_baselink, row-major
  double t[3];
};
struct DevLoss {
  int kind;
  int pad;
  double a;
};

// row pitch (doubles) of the pose part of the reprojection Jacobian.  (Padding the rows to one 128-byte line each, pitch 16, was
// tried for the pair kernel's gathers: no gain there, and the evaluation kernel's partial-line streaming stores went 18 -> 31 us.)
constexpr int kJAStride = 12;

// entries of one camera pair are cut into segments of at most this many (one single-wave workgroup of pairs_kernel each);
// a power of two.  Host and device flattening must agree (their tables are compared bit for bit).
constexpr int kPairChunk = 256;
// entries of one camera pair a wave of pairs_kernel sums (a power of two): a window of the reference's size has a few hundred camera pairs
// of up to a few hundred entries — 210 waves of up to four strides on 1 024 SIMDs; in chunks of 64 it is twice the waves of one stride each
// (20 KF x 500: pairs_kernel 10.1 -> 6.5 us, 9 420 -> 9 750 LM it/s; 30 KF x 2 000 + 1.3 %), while a window with more entries than waves to
// hide them behind pays for the additional atomics (50 KF x 5 000: - 1.3 % at 64, + 1 % at 128)
inline int pair_chunk(size_t n_ent) { return n_ent <= 100000 ? 64 : n_ent <= 1000000 ? 128 : kPairChunk; }
// a landmark whose observations lie within this many consecutive camera poses (every feature track of a sliding window), one per camera
// pose, is a BAND landmark: its whole contribution to the reduced system is formed by pairs_band_kernel (k_band.hip), it has no pair entries
constexpr int kBandCams = 13;
constexpr int kBandMinFactors = 150000;   // windows below this keep every pair as an entry (bsgpu_finalize.cpp)

// meta word of a reprojection factor: camera id | loss id | constant-block flags
constexpr int kMetaCamBits = 12, kMetaLossBits = 12;
constexpr int kLmRec = 10;   // doubles per landmark record (Visual::Linv / z)
constexpr int kFlagQConst = 1, kFlagPConst = 2, kFlagLConst = 4;
__host__ __device__ inline int meta_pack(int cam, int loss, int flags) {
  return cam | (loss << kMetaCamBits) | (flags << (kMetaCamBits + kMetaLossBits));
}

// scalar slots (device buffer of doubles, copied to the host once per LM iteration)
enum {
  SC_COST_X = 0,     // cost at the current point (active factors)
  SC_COST_CAND = 1,  // cost at the candidate point
  SC_MCC = 2,        // model cost change
  SC_STEP_NORM2 = 3, // |x_cand - x|^2 (ambient, non-constant blocks)
  SC_X_NORM2 = 4,    // |x|^2
  SC_GRAD_MAX = 5,   // |x - Plus(x,-g)|_inf   (5, 6: from the per-workgroup partials of grad_norms_kernel, by the end-of-step reduction)
  SC_GRAD_NORM2 = 6,
  SC_CHOL_FAIL = 7,  // > 0 when a pivot was not positive / finite
  SC_FIXED_COST = 8,
  SC_CHOL_FAIL_SEEN = 9,  // SC_CHOL_FAIL as the riding reduction mirrored it (which then clears the flag): what the decision on the device reads
  SC_RADIUS = 12,    // trust-region radius of the step being computed (written by the host before each step)
  SC_DEC_GO = 13,    // (host mirror) the decision taken on the device for the step this reduction closes: 1 = accepted, 0 = anything else (LmDecide)
  SC_DEC_RADIUS = 14,  // (host mirror) ... and the radius of the step after it
  SC_SEQ = 15,       // (host mirror only) sequence number of the end-of-step reduction that filled the mirror
  SC_NUM = 16
};

// pose-only ("small") factor group on the device: r (n x m), J (n x m x 3*nv) row-major
struct SmallGroup {
  int type = 0, n = 0, m = 0, nv = 0, nc = 0;
  int* xoff = nullptr;   // n x nv  offsets into x
  int* toff = nullptr;   // n x nv  tangent offsets (-1 = constant block)
  double* consts = nullptr;
  int* loss = nullptr;   // n  loss-table ids
  unsigned char* active = nullptr;  // n
  double* r = nullptr;
  double* J = nullptr;
  int w_last = 3;        // tangent width of the LAST slot (1 for the inverse-depth scalar); J keeps 3 columns per slot
  int* cam = nullptr;    // n  camera-table ids (types that project through a camera)
  const DevCamera* cams = nullptr;
};

// several pose-only groups handed to one launch
constexpr int kSetMax = 4;
struct SmallGroupSet {
  SmallGroup g[kSetMax];
  double* part[kSetMax];
  int first[kSetMax + 1];   // first workgroup of every group; first[n] = grid size
  int n;
};

// dense linear prior factor ([EXT] fuse_constraints::MarginalConstraint) on the device (k_marg.hip)
struct MargDev {
  int rows = 0, cols = 0, nblk = 0;
  const int* blk_xoff = nullptr;  // per block: offset into x
  const int* blk_size = nullptr;  //            ambient size
  const int* blk_quat = nullptr;  //            1 = quaternion manifold
  const int* blk_col = nullptr;   //            first tangent column inside the factor
  const int* blk_amb = nullptr;   //            offset into xbar
  const int* col_t = nullptr;     // per column: tangent index, -1 for a constant block
  const int* col_blk = nullptr;   //             block (position inside the factor)
  const double* A = nullptr;      // rows x cols row-major
  const double* b = nullptr;
  const double* xbar = nullptr;
  double* delta = nullptr;        // cols   x [-] xbar
  double* D = nullptr;            // nblk: |x|^2 of quaternion blocks
  double* r = nullptr;            // rows
  double* J = nullptr;            // rows x cols (tangent)
};

// everything the kernels need for the visual (landmark) part
struct Visual {
  int n = 0;         // reprojection factors, sorted by landmark (constant-landmark ones last)
  int n_elim = 0;    // factors whose landmark is eliminated
  int n_lm = 0;      // eliminated landmarks
  int4* fac = nullptr;        // (xoff q, xoff p, xoff P, meta)
  double2* pix = nullptr;
  double* w = nullptr;
  int* cam_pose = nullptr;    // factor -> camera-pose id (unique (q,p) block pair)
  int* lm_of = nullptr;       // factor -> landmark (tangent order) or -1
  int* lm_start = nullptr;    // n_lm + 1
  int n_cam_pose = 0;
  int* cp_tq = nullptr;       // camera pose -> tangent offset of q (or -1)
  int* cp_tp = nullptr;
  // pair segments
  int n_seg = 0, n_ent = 0;
  int* seg_ci = nullptr; int* seg_cj = nullptr; int* seg_start = nullptr;
  int* ent_fa = nullptr; int* ent_fb = nullptr;
  // ... the same entries in segments of up to kPairChunk, for launches that are bound by their work and not by a wave's length (a window
  // among many in bsgpu_solve_batch); built on first use (bsgpu_batch.cpp coarse_pair_segments), the fine list itself where it is that coarse
  int n_seg_c = 0;
  int* seg_ci_c = nullptr; int* seg_cj_c = nullptr; int* seg_start_c = nullptr;
  // band landmarks (band_plan.h): units = (first camera pose k0, a part of its landmarks by falling span); one record per landmark, in
  // unit order: (first factor row, slot mask | span << 16 | observations << 24, observation index of slot j in nibble j of z | w << 32, 15 = none)
  int n_band_units = 0, n_band_lm = 0;
  int* band_unit_start = nullptr; int* band_unit_cam = nullptr; int4* band_lm = nullptr;
  int* band_lm_id = nullptr;  // the landmark of every record of band_lm (the band kernel's look-up of Linv and z when no C rows are kept: no_cr)
  // NO C ROWS (round 6).  landmark_kernel's second pass wrote C = B Linv^T and rho = r - C z per observation (64 B) for the pair phase and the
  // back-substitution to read: 6.5 of its 22 us on C2 (measured with the pass taken out).  When EVERY landmark is a band landmark (no pair
  // entries, no factors of constant landmarks) the two consumers form C and rho themselves from the B rows (48 B, which they have or read instead)
  // and the landmark's Linv and z: lone solves pass CR = nullptr to the three launches; the batched launches keep the C rows.  BSGPU_NO_CR=0: never.
  bool no_cr = false;
  // outputs
  double2* r = nullptr;       // n
  double* J = nullptr;        // robustified Jacobian, split by consumer: pose part n x 12 ([A row 0 (theta, t: 6) | A row 1]) ...
  double* JB = nullptr;       // ... and landmark part n x 6 ([B row 0 | B row 1]) = J + 12 n: the landmark kernel streams 48 B per factor
                              // instead of dragging 144-byte rows through for a third of their bytes
  double* CR = nullptr;       // n x 8: C = B M (2x3), rho = r - C z (2)
  // per landmark ONE record of kLmRec doubles (80 B, 16-byte aligned): [Linv (6: lower-triangular inverse factor of Hll + lambda) | z = Linv g_l (3) | pad] — the
  // consumers that look a landmark up by its id (the band kernel without C rows) touch one or two lines for it instead of a partial line of each of two arrays
  double* Linv = nullptr;     // record l at Linv + kLmRec l
  double* z = nullptr;        // = Linv + 6: z of landmark l at z + kLmRec l
  double* cost_part = nullptr;       // per-workgroup cost partials at the current point
  double* cost_part_cand = nullptr;  // ... at the candidate point
  double* mcc_part = nullptr;        // per-workgroup model-cost-change partials
  int n_cost_part = 0;
};

// inverse-depth landmarks eliminated on the landmark side (k_idp.hip): the binary inverse-depth factors (SmallGroup of
// BSGPU_F_IDP_REPROJ: r, J 2 x 15 from idp_kernel) sorted by their scalar landmark; a VIEW is one (landmark, camera pose) pair — the
// anchor pose once per landmark, every measurement pose once — and carries u = sum A^T c over the factors of the landmark that see it
struct IdpElim {
  int n_lm = 0;               // eliminated inverse-depth landmarks; tangent offset of landmark l = to0 + l
  int to0 = 0;
  int n_fac = 0;              // binary factors with an eliminated landmark, sorted by it
  int n_view = 0;
  int* order = nullptr;       // sorted position -> factor of the group
  int* lm_start = nullptr;    // n_lm + 1, into order
  int* view_start = nullptr;  // n_lm + 1, into the views
  int2* fview = nullptr;      // sorted position -> (view of the anchor pose, view of the measurement pose)
  int* view_lm = nullptr;     // view -> landmark
  int* view_cp = nullptr;     // view -> camera pose
  int* view_code = nullptr;   // (direct) view -> (sorted position << 1 | side) of its only factor, or -(row of VD + 1)
  int direct = 0;             // 1: the factors' own pose-pose terms are assembled by idp_pairs_kernel (every binary factor is in `order`)
  int n_cam_pose = 0;
  int* cp_tq = nullptr;       // camera pose -> tangent offset of q / p (or -1)
  int* cp_tp = nullptr;
  int n_seg = 0, n_ent = 0;   // pair segments: entries (view a, view b) of one landmark, grouped by camera-pose pair (ci <= cj)
  int* seg_ci = nullptr; int* seg_cj = nullptr; int* seg_start = nullptr;
  int* ent_va = nullptr; int* ent_vb = nullptr; int* ent_code = nullptr;   // (code: idp_pairs_kernel)
  double* U = nullptr;        // n_view x 8: u (6), z of the landmark, 0
  double* VD = nullptr;       // (direct) one row of 48 per view of several factors: D = sum A^T A (36), sum A^T r (6), pad
  double* linv = nullptr;     // n_lm: 1 / sqrt(h + lambda)
  double* z = nullptr;        // n_lm: linv * g
  double* C = nullptr;        // n_fac x 2: c = w linv  (w = d r / d rho)
};
void launch_idp_landmark(hipStream_t s, const IdpElim& e, const SmallGroup& g, const double* radius_ptr, double radius_val, int compute_scale,
                         int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, double* grad);
void launch_idp_pairs(hipStream_t s, const IdpElim& e, const SmallGroup& g, double* S, int ld, int rhs_row, double* grad, double* hdiag,
                      const int* perm, bool grad_only);
void launch_idp_backsub(hipStream_t s, const IdpElim& e, const double* y_pose, double* delta);

// the candidate x (+) delta of the blocks the landmark back-substitution does not update itself, as extra workgroups of that launch
// (k_reproj.hip: backsub_mcc_kernel); the Euclidean landmark blocks are written by the lanes that compute their step
struct UpdateRide {
  int n_blocks = 0;                    // entries of `blocks` (0: the launch carries no update)
  const int* blocks = nullptr;         // block indices: everything but the eliminated Euclidean landmarks
  const int* xoff = nullptr; const int* toff = nullptr; const unsigned char* size = nullptr; const unsigned char* manifold = nullptr;
  const int* lm_xoff = nullptr;        // eliminated Euclidean landmark -> offset of its block in x
  const double* x = nullptr; double* x_cand = nullptr;
  double* part = nullptr;              // (|x_cand - x|^2, |x|^2) per update unit, then per landmark workgroup
};
// entry of the end-of-step reduction table (k_misc.hip: final_reduce_kernel)
struct ReduceEntry {
  const double* ptr;
  int n, stride, offset, slot;
  int op = 0;   // 0: sum, 1: maximum (of non-negative values)
};

// The decision of LmState::advance (lm_state.h) for the common case, taken on the device by the LAST unit of a riding reduction so that the
// assembly issued ahead of the host's decision (bsgpu_solve.cpp enqueue_step) runs at the radius the host WILL name instead of a guessed
// one: "accepted" + the next radius, or "anything else" (invalid step, a tolerance reached, rejected) — then the assembly's workgroups
// return at once.  The host still decides (same arithmetic, bit for bit) and adopts the assembly only if it names the same radius; what is
// passed in is the state BEFORE the decision, which the host knows when it enqueues the launch.
struct LmDecide {
  int on = 0;
  int check_grad = 0;     // the step was computed at a NEW point: its gradient tolerance test comes first (LmState::advance, top of the loop)
  int x_from_scal = 0;    // the cost at the current point is SC_COST_X of this reduction (else x_cost below: the cost the host holds)
  double radius = 0.0, x_cost = 0.0;
  double min_relative_decrease = 0.0, max_radius = 0.0, function_tolerance = 0.0, parameter_tolerance = 0.0, gradient_tolerance = 0.0;
};
// the decision: ONE word — the next radius, negative when the step was not accepted, zero while there is none — in 64 copies a cache line apart
// (the polling wave of workgroup w reads copy `w & 255`; 4 160 bytes apart: spread over the memory channels — the pollers' loads are device-coherent
// and go past the L2s), in two banks that consecutive deciding launches take in turn (a launch's decision clears the other)
constexpr int kDecSlots = 256, kDecStride = 520;
// a launch of an assembly ahead that returns at once unless the decision on the device was "accepted": p[0] > 0
struct GoWord { const double* p = nullptr; };

// the end-of-step reduction (k_misc.hip final_reduce_kernel) as units of work of another launch: n_slots + 1 units
struct ReduceRide {
  const ReduceEntry* entries = nullptr;
  int n_entries = 0, n_slots = 0;
  double* scal = nullptr; double* host_scal = nullptr; int* counter = nullptr;
  double seq = 0.0;
  int skip_slot = -1;   // a slot whose partial arrays the carrying launch itself rewrites: left alone (its unit only counts itself in)
  int defer_slot = -1;  // this launch leaves that slot's unit to a later launch (which carries it alone: only_slot) — the partial arrays it sums
                        // are written by THIS launch; every other unit runs here, and the deferred one is the reduction's last
  int only_slot = -1;
  // (only_slot's partial arrays, in the table's order, in the kernel's ARGUMENTS — the unit asks for all of them with its first loads instead of
  //  staging the table and walking it, an array per memory round trip, underneath the carrying launch's own traffic: at most one of more than
  //  `256` values (<= 4 096), the others of at most 256.  n_early = 0: the table)
  ReduceEntry early[4];
  int n_early = 0;
  double* dec = nullptr;   // kDecSlots x kDecStride doubles: where the last unit leaves the decision (lmd.on)
  double* dec_next = nullptr;   // ... and the bank it clears for the next deciding launch
  LmDecide lmd;
};

struct LaunchCtx {
  hipStream_t stream;
};

// ---- kernel launchers (bsgpu_kernels.hip) ----------------------------------------------------------
void launch_reproj_eval(hipStream_t s, const Visual& v, const double* x, const DevCamera* cams,
                        const DevLoss* losses, bool with_J, double* cost_part_out, bool count_inactive = false);
void launch_visual_imu_eval(hipStream_t s, const Visual& v, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevCamera* cams,
                            const DevLoss* losses, bool with_J, double* cost_part_vis, double* part_delta, double* part_prior,
                            const ReduceRide* red = nullptr /* with_J: the end-of-step reduction as the launch's first workgroups */);
void launch_relpose_imu_eval(hipStream_t s, const SmallGroup& g, const SmallGroup& delta, const SmallGroup& prior, const double* x,
                             const DevLoss* losses, bool with_J, double* cost_part, double* part_delta, double* part_prior, const ReduceRide* red = nullptr);
void launch_imu_eval(hipStream_t s, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevLoss* losses, bool with_J,
                     double* part_delta, double* part_prior);
int small_cost_parts(const SmallGroup& g);
void launch_small_eval(hipStream_t s, const SmallGroup& g, const double* x, const DevLoss* losses, bool with_J,
                       double* cost_part /* n doubles: per-factor cost */);
// several groups in ONE launch (false: too many groups or a type it does not carry — the caller launches them one by one)
bool launch_small_eval_set(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses, bool with_J);
// ... with the window's dense prior (k_marg.hip) evaluated by the same launch; false: nothing was launched
bool launch_small_eval_set_marg(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses, bool with_J,
                                const MargDev& m, double* marg_part);
// what an LM step clears before its assembly (zero_tiles_multi_kernel's arguments); rides in the landmark launch as extra workgroups
// Several windows advanced by ONE set of launches (bsgpu_batch.cpp; bs_models/src/lib/global_mapping/submap_refinement.cpp:35-115 is the
// reference's serial loop over submaps): every kernel of the LM step has a `_batch` form whose blockIdx.y picks a window out of one of
// these lists and whose arguments come from a per-window table in device memory.  The block below is what changes from one LM
// iteration to the next; it goes up once per iteration.
constexpr int kBatchMaxWin = 64;
enum { BL_ALL = 0, BL_FULL = 1, BL_ACC = 2, BL_REJ = 3, BL_BS_FUSED = 4 /* + deep */, BL_BS_CHAIN = 6 /* + deep */, BL_DIAG = 8 /* windows whose LM diagonal / gradient norms need their own launch this round */, BL_CLEAR = 9 /* windows without a landmark launch whose reduced system was not cleared at the end of their previous step */, BL_NUM = 10 };   // (BL_BS_*: the FULL windows by the form of their back-substitution)
struct BatchDyn {
  int n[BL_NUM];                       // windows in: every window still iterating | those that compute a full step (not just the gradient
  int idx[BL_NUM][kBatchMaxWin];       // of their last point) | those whose candidate was accepted (x <- x_cand) | those whose step was
                                       // rejected (Jacobians at x again)
  double radius[kBatchMaxWin];         // per window (indexed by window, not by list position)
  double seq[kBatchMaxWin];            // stamp of this iteration's end-of-step reduction
  int first[kBatchMaxWin], new_J[kBatchMaxWin], grad_only[kBatchMaxWin];
};
// device tables of the tiled Cholesky plan, as the factorisation / back-substitution entry points take them
// What the fused factorisation's launch carries besides the factorisation (dense_plan.h kFusedDiagAdd / kFusedRider tasks): the LM
// diagonal of the reduced system (pose_diag_kernel's work, per tile, as the FIRST update of every diagonal tile) and the gradient norms of
// the step (grad_norms_kernel's work, as workgroups that nothing waits for) — a launch less on the dependent path of every LM step.
struct LmDiag {
  const double* hdiag = nullptr;   // null: S carries the LM diagonal already (the diagonal tasks only count themselves in)
  double* scale = nullptr; double* dcl = nullptr;
  const int* inat = nullptr;       // position in S -> tangent index (or -1)
  double inv_radius = 0.0, lm_lo = 0.0, lm_hi = 0.0;
  int compute_scale = 0, compute_dcl = 0, jacobi = 0;
};
struct GradNormRide {
  int nb = 0;                      // 0: no norms this step
  const int* xoff = nullptr; const int* toff = nullptr; const unsigned char* size = nullptr; const unsigned char* manifold = nullptr;
  const double* x = nullptr; const double* grad = nullptr; double* gpart = nullptr;
};
struct DenseDev {
  const int *nreal, *rows_flat;
  const PanelDesc* panels;
  double *Lp, *Vinv;
  const int *bs_desc, *chain_begin, *chain_end;   // bs_desc: DensePlan::bs_desc on the device
  int* tile_sync;   // [expected arrivals | arrival counters] per tile (dense_plan.h)
  const FusedTask* ftasks = nullptr;   // fused single-launch factorisation (null: the launch-per-step path)
  int* fsync = nullptr;
  // level-synchronous back-substitution (DensePlan::bs_level_sync): chain-only panel records and the between-group update items
  const int *bs_desc_chain = nullptr, *rows_flat_chain = nullptr, *bs_upd = nullptr, *bs_upd_rows = nullptr;
  // ... and its single-launch form
  const int *bs_chain_group = nullptr, *bs_grp_nchains = nullptr, *bs_grp_nitems = nullptr, *bs_items4 = nullptr, *bs_tile_updated = nullptr;
  int* bs_sync = nullptr;
  double* scal = nullptr;
  double* Winv = nullptr;   // per tile: the full inverse of its factor (written by the fused factorisation, read by the single-launch back-substitution)
  const int* bs_order = nullptr;   // ticket -> role of the single-launch back-substitution (DensePlan::bs_order)
  const int* tile_tot = nullptr;   // fused factorisation: update tasks per tile (DensePlan::tile_tot)
  int rhs_rows = 0;                // rows of the rhs tile that are in use (the LM solve: 1); 0: every row may be
  LmDiag diag;                     // what the factorisation's launch carries along (plans with diagonal / rider tasks)
  GradNormRide gn;
  const FusedTask* ftasks_plain = nullptr; const int* tile_tot_plain = nullptr; int n_ftasks_plain = 0;   // the list for a launch that carries neither
};
// per-window argument table of one `_batch` kernel: entry w = the arguments window w's lone launch would pass (host image, then uploaded)
struct BatchArgTable {
  std::vector<unsigned char> host;
  size_t stride = 0, lds = 0;
  int max_grid = 0;
  void* dev = nullptr;
  template <class A> void push(const A& a) {
    stride = sizeof(A);
    const size_t o = host.size();
    host.resize(o + sizeof(A));
    std::memcpy(&host[o], &a, sizeof(A));
    if (a.bsg_grid > max_grid) max_grid = a.bsg_grid;
  }
};
struct ZeroStep {
  double* S = nullptr; int ld = 0; const int* tiles = nullptr; int n_tiles = 0;
  double* a = nullptr; int na = 0; double* b = nullptr; int nb = 0; double* c = nullptr; int nc = 0;
  double* radius_slot = nullptr; double radius = 0.0;
};
// `_batch` forms of the LM step's launches (k_*.hip): batchargs_* appends window w's entry (what its lone launch passes), launch_*_batch
// runs the windows of list `list` (BatchDyn::idx) in one launch
struct Visual; struct SmallGroup; struct SmallGroupSet; struct UpdateRide; struct ReduceEntry; struct DevCamera; struct DevLoss;
void batchargs_visual_imu_eval(BatchArgTable& t, const Visual& v, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevCamera* cams,
                               const DevLoss* losses, double* cost_part_vis, double* part_delta, double* part_prior);
void launch_visual_imu_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J);
void batchargs_landmark(BatchArgTable& t, BatchArgTable& t_tail, const Visual& v, int n_pose, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl,
                        double* grad, const ZeroStep& zero);
void launch_landmark_batch(hipStream_t s, const BatchArgTable& t, const BatchArgTable& t_tail, const BatchDyn* dyn, int list, int n);
void batchargs_pairs(BatchArgTable& t, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* small,
                     int n_small_units);
void launch_pairs_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_pairs_band(BatchArgTable& t, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* small,
                          int n_small_units);
void launch_pairs_band_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_backsub_mcc(BatchArgTable& t, const Visual& v, int n_pose, const double* y_pose, double* delta, double* mcc_part, const SmallGroupSet* small,
                           int n_small_units, const UpdateRide* upd);
void launch_backsub_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_grad_norms_pose_diag(BatchArgTable& t, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                                    const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart, int n_pose, double* S, int ld,
                                    const double* hdiag, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, int npad, const int* iperm);
void launch_grad_norms_pose_diag_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_final_reduce(BatchArgTable& t, const ReduceEntry* entries, int n_entries, int n_slots, double* scal, double* host_scal, int* counter);
void launch_final_reduce_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_copy(BatchArgTable& t, const double* src, double* dst, int64_t n);
void launch_copy_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_chol_fused(BatchArgTable& t, double* S, double* Lp, int ld, const FusedTask* tasks_dev, int n_tasks, const int* tile_tot_dev, const int* nreal_dev,
                          double* Vinv, double* scal, int* sync_dev, double* Winv, int rhs_rows, const LmDiag& diag, const GradNormRide& gn, bool diag_tasks_in_list);
void launch_chol_fused_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
// ... the pose-only family (k_small.hip, k_misc.hip): lidar-inertial windows, dense-path pose graphs, the pose-only factors of any window
struct AsmGroup;
void batchargs_relpose_imu_eval(BatchArgTable& t, const SmallGroup* g, const SmallGroup& delta, const SmallGroup& prior, const double* x, const DevLoss* losses,
                                double* cost_part, double* part_delta, double* part_prior);
void launch_relpose_imu_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J);
bool batchargs_small_eval_set(BatchArgTable& t, const SmallGroup* groups, double* const* parts, int n_groups, const double* x, const DevLoss* losses);
void launch_small_eval_set_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J);
bool batchargs_small_assemble_set(BatchArgTable& t, const SmallGroup* groups, int n_groups, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm);
void launch_small_assemble_set_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_small_assemble_seg(BatchArgTable& t, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_ra, const int* seg_rb, const int2* contrib,
                                  double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* fw, int n_fw_units, int n_grp,
                                  const AsmGroup* grp);
void launch_small_assemble_seg_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
bool batchargs_small_mcc(BatchArgTable& t, const SmallGroup* groups, double* const* parts, int n_groups, const double* delta, const UpdateRide* upd, const ZeroStep* zero);
void launch_small_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_zero_tiles_multi(BatchArgTable& t, const ZeroStep* zs);
void launch_zero_tiles_multi_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
// ... a window's dense prior (k_marg.hip)
struct MargDev;
bool batchargs_marg_eval(BatchArgTable& t, const MargDev* m, const double* x, double* cost_part);
void launch_marg_eval_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n, bool with_J);
void batchargs_marg_assemble(BatchArgTable& t, const MargDev* m, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm);
void launch_marg_assemble_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_marg_mcc(BatchArgTable& t, const MargDev* m, const double* delta_tan, double* part);
void launch_marg_mcc_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
// ... inverse-depth landmarks eliminated on the landmark side (k_idp.hip)
struct IdpElim;
void batchargs_idp_landmark(BatchArgTable& t, BatchArgTable& t_view, const IdpElim& e, const SmallGroup& g, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl, double* grad);
void launch_idp_landmark_batch(hipStream_t s, const BatchArgTable& t, const BatchArgTable& t_view, const BatchDyn* dyn, int list, int n);
void batchargs_idp_pairs(BatchArgTable& t, const IdpElim& e, const SmallGroup& g, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm);
void launch_idp_pairs_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
void batchargs_idp_backsub(BatchArgTable& t, const IdpElim& e, const double* y_pose, double* delta);
void launch_idp_backsub_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n);
int batchargs_backsolve(BatchArgTable* tabs, const DensePlan& P, const DenseDev& D, double* y, const int* iperm, int n_pose, double* y_tan, double* delta);
void launch_backsolve_batch(hipStream_t s, const BatchArgTable* tabs, const BatchDyn* dyn, const int* n_in_form);
void launch_landmark(hipStream_t s, const Visual& v, int n_pose, const double* radius_ptr, int compute_scale,
                     int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale, double* dcl,
                     double* grad, const ZeroStep* zero = nullptr, double radius_val = 0.0, const ReduceRide* red = nullptr /* the step before's end-of-step reduction as the launch's first workgroups */);
void launch_pairs(hipStream_t s, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm,
                  bool grad_only, const SmallGroupSet* small = nullptr, int n_small_units = 0, GoWord go = GoWord());
bool band_available();   // the current device gives the band kernels their dynamic LDS (k_band.hip)
void launch_pairs_band(hipStream_t s, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm,
                       bool grad_only, const SmallGroupSet* small = nullptr, int n_small_units = 0, bool lower_only = false /* only the entries the tiled factorisation reads: on and below the diagonal in solver order */,
                       GoWord go = GoWord());
int small_assemble_first_set(const SmallGroup* groups, int n_groups, SmallGroupSet* set, int* n_taken);
void launch_small_assemble_set(hipStream_t s, const SmallGroup* groups, int n_groups, double* S, int ld, int rhs_row, double* grad,
                               double* hdiag, const int* perm);
// factors of one pose-only type that name the SAME variables in every slot (the ~21 lidar constraints between two keyframes of a
// lidar-inertial window): their J^T J is summed on the matrix cores before it is added to the reduced system (k_small.hip: small_assemble_group).
// The record carries everything its workgroup needs — a lane's first load is its factor's index, its second the factor's rows (no
// trips through the type table and a factor list in between).
constexpr int kAsmGroupMax = 24;   // factors of a record (one pass through LDS: 144 rows of <= 18, 21 KB — seven workgroups to a compute unit)
struct AsmGroup {
  int type, count, m, nv;
  int toff[6];            // tangent offset of every slot (-1: constant, or not a slot)
  int te;                 // columns up to and including the last slot that is not constant: the sums beyond are zeros and are not formed
  int pad;
  const double* J;        // the type's Jacobian / residual tables
  const double* r;
  int fac[kAsmGroupMax];  // the factors (entries past count repeat the last one)
};
// (marg: the window's dense prior assembled by the same launch; returns whether it was carried)
bool launch_small_assemble_seg(hipStream_t s, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_ra, const int* seg_rb,
                               const int2* contrib, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm,
                               const SmallGroupSet* fw = nullptr, int n_fw_units = 0, int n_grp = 0, const AsmGroup* grp = nullptr,
                               const MargDev* marg = nullptr, const ReduceRide* red = nullptr /* the step before's reduction as the first workgroups (not with marg) */,
                               bool* red_carried = nullptr);
void launch_pose_diag(hipStream_t s, int n_pose, double* S, int ld, const double* hdiag, const double* radius_ptr,
                      int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale,
                      double* dcl, int npad, const int* iperm, double radius_val = 0.0 /* used when radius_ptr is null */);
void launch_grad_norms(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                       const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart /* 2 per workgroup: max, sum of squares */);
void launch_grad_norms_pose_diag(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                                 const unsigned char* blk_manifold, const double* x, const double* grad, double* gpart, int n_pose, double* S,
                                 int ld, const double* hdiag, const double* radius_ptr, int compute_scale, int compute_dcl, int jacobi,
                                 double lm_lo, double lm_hi, double* scale, double* dcl, int npad, const int* iperm, double radius_val = 0.0);
struct PanelDesc;
struct FusedTask;
void launch_chol_fused(hipStream_t s, double* S, double* Lp, int ld, const FusedTask* tasks_dev, int n_tasks, const int* tile_tot_dev, const int* nreal_dev,
                       double* Vinv, double* scal, int* sync_dev, double* Winv, int rhs_rows /* rows of the rhs tile in use; <= 0: all 64 */,
                       const LmDiag& diag = LmDiag(), const GradNormRide& gn = GradNormRide(),
                       bool diag_tasks_in_list = false /* the list has the LM-diagonal tasks of its diagonal tiles (DensePlan::ftasks, not ftasks_plain / _bulk) */);
void launch_chol_potrf_tiles(hipStream_t s, double* S, double* Lp, int ld, const int* tiles_dev, int n_tiles, const int* nreal_dev,
                             double* Vinv, double* scal);
void launch_chol_panel_step(hipStream_t s, double* S, double* Lp, int ld, const PanelDesc* descs_dev, int n_panels, int max_rows,
                            const int* rows_flat_dev, const int* nreal_dev, double* Vinv, double* scal, int* tile_sync_dev,
                            const PanelDesc* descs_host = nullptr, const int* rows_flat_host = nullptr);
void launch_chol_backsolve_chains(hipStream_t s, const double* S, const double* Lp, const double* Vinv, int ld,
                                  const int* bs_desc_dev, const int* chain_begin_dev,
                                  const int* chain_end_dev, int n_chains, const int* rows_flat_dev, double* y,
                                  int npad, int max_chain_len, const double* y_init = nullptr, const int* iperm_dev = nullptr, int n_pose = 0,
                                  double* y_tan = nullptr, double* delta = nullptr, int max_rows = 0, const double* Winv = nullptr);
void launch_chol_backsolve_update(hipStream_t s, const double* Lp, int ld, const int* items_dev, int n_items, const int* upd_rows_dev, double* y);
bool launch_chol_backsolve_fused(hipStream_t s, const double* Lp, const double* Winv, int ld, const int* bs_desc_dev, const int* chain_begin_dev,
                                 const int* chain_end_dev, const int* rows_flat_dev, int n_chains, const int* chain_group_dev,
                                 const int* grp_nchains_dev, const int* grp_nitems_dev, int G, const int* items_dev, int n_items,
                                 const int* upd_rows_dev, const int* tile_updated_dev, double* y, int npad, int max_chain_len, int max_rows,
                                 const double* y_init, const int* iperm_dev, int n_pose, double* y_tan, double* delta, int* sync_dev, double* scal,
                                 const int* order_dev);
size_t chol_backsolve_chain_lds(int npad, int max_chain_len);
void launch_marg_schur(hipStream_t s, const double* S, int ld, int rhs_row, const int* spos_dev, int n, int m, double rel_tol,
                       double* M, double* g, double* diag0, int* pivot_ok, double* status, double* A, double* b);
void launch_cov_units(hipStream_t s, double* S, int ld, int rhs_row, const int* cols_dev, int n);
void launch_cov_dots(hipStream_t s, const double* Lp, int ld, int rhs_row, int n_cols, int ta, int row_b0, int tb, double* out);
void launch_marg_eval(hipStream_t s, const MargDev& m, const double* x, bool with_J, double* cost_part /* rows */);
void launch_marg_assemble(hipStream_t s, const MargDev& m, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm);
void launch_marg_mcc(hipStream_t s, const MargDev& m, const double* delta_tan, double* part /* rows */);
void launch_reproj_errors(hipStream_t s, const Visual& v, const SmallGroup& dense, const double* x, const DevCamera* cams, double* out_vis,
                          double* out_dense);
void launch_preintegrate(hipStream_t s, int n_int, const int* sample_start, const double* ts, const double* wm, const double* am,
                         const double* t_end, const double* bg, const double* ba, const double* covs, double info_weight, double* out);
void launch_triangulate(hipStream_t s, int n_tracks, const int* track_start, const int2* pose_off, const double2* pix, const double* x,
                        const DevCamera& cam, bool truncate, double max_dist, double max_reproj, double* points, int* status);
// device-side flattening of the reprojection factors (k_flatten.hip): 0 = done, 1 = take the host path, < 0 = device error.
// `res` non-null: the raw table is already on the device, its block columns naming caller slots (SlotMirror, bsgpu_ctx.h)
struct FlattenResident { const int* idx; const double* consts; const int* loss_kind; const double* loss_a; const int* slot_map; };
struct FlattenSegsHost { std::vector<int> seg_ci, seg_cj, cp_tq, cp_tp; };   // host copies of the camera-pose pair segments (optional output)
void launch_patch_factor_rows(hipStream_t s, int n_ch, const int* rows, const int* idx4, const double* consts3, const int* lk, const double* la,
                              int* dst_idx, double* dst_consts, int* dst_lk, double* dst_la);
int flatten_visual_device(hipStream_t s, const std::function<void*(size_t)>& dalloc, int n, const int* h_idx, const double* h_consts,
                          const int* h_loss_kind, const double* h_loss_a, const std::vector<DevLoss>& losses, int nb, const int* d_blk_xoff,
                          const int* d_blk_toff, const unsigned char* d_blk_const, const int* d_blk_lm, int nl, int T, Visual& V,
                          int** d_vis_src, std::vector<unsigned char>& tile_adj, bool* any_all_const, const FlattenResident* res = nullptr,
                          FlattenSegsHost* segs_out = nullptr, bool band_enabled = true);
int chol_vinv_stride();
void chol_prepare();  // one-time function attributes (kept out of captured sequences)
// block-sparse PCG path (k_pcg.hip)
void launch_bsr_assemble(hipStream_t s, const SmallGroup& g, const int* slots, double* val);
void launch_bsr_assemble_seg(hipStream_t s, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_slot, const int* seg_row,
                             const int2* contrib, double* val, double* rhs, double* grad, double* hdiag);
void launch_bsr_finish_diag(hipStream_t s, int nbr, const int* diag_slot, const int* pair_slot, double* val, const double* hdiag, double radius,
                            int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale,
                            double* dcl, double* Minv);
int pcg_spmv_grid(int nbr);   // workgroups (= p.q partials) of one SpMV launch
// the whole PCG solve as one launch (k_pcg.hip pcg_persistent_kernel): per-workgroup row ranges and named-column lists
struct PcgPersistDev {
  int G = 0, max_cols = 0;
  int paired = 0;                // every named-column list consists of pairs (2m, 2m + 1): gathered 16 bytes at a time
  const int *wg_row = nullptr, *wg_colptr = nullptr, *wg_cols = nullptr, *lcol = nullptr;
  unsigned long long* slots = nullptr;
  int* abort_w = nullptr;
  double* zg = nullptr;          // the published z: two sets of 3 nbr entries
  size_t sync_bytes = 0;         // slots | z sets | abort word: one allocation, emptied (all bits set) by one fill per launch
  // two-level preconditioner (k_pcg.hip): the free poses (value offsets of p and q, tangent offsets of p and q), the centre the
  // rotations go through, W (3 nbr x 6), E = W^T A W (accumulated per LM step), E^-1
  int n_coarse = 0;
  const int4* co = nullptr;
  double centre[3] = {0.0, 0.0, 0.0};
  double *W = nullptr, *E = nullptr, *Einv = nullptr;
  unsigned* counter = nullptr;
};
size_t pcg_coarse_scratch_doubles(int nbr);
void launch_pcg_coarse(hipStream_t s, const PcgPersistDev& P, int nbr, const int* row_ptr, const int* col, const double* val, const double* x);
size_t pcg_persistent_lds(int max_cols);
size_t pcg_persistent_lds_limit();
int pcg_persistent_max_rows();
size_t pcg_persistent_slot_words(int G);
size_t pcg_persistent_z_words(int nbr);
bool launch_pcg_persistent(hipStream_t s, const PcgPersistDev& P, int nbr, const int* row_ptr, const double* val, const double* Minv, const double* b,
                           double* x, double* zg, double* sc, double tol2, int max_it);
void launch_pcg_init(hipStream_t s, int nbr, const double* b, const double* Minv, double* x, double* r, double* z, double* p0, double* p1,
                     double* part, double* sc);
void launch_pcg_iteration(hipStream_t s, int k, int nbr, const int* row_ptr, const int* col, const double* val, const double* Minv,
                          double* x, double* r, double* z, double* p0, double* p1, double* q, double* part_pq, double* part, double* sc,
                          double tol2);
int pcg_num_scalars();
int pcg_rows_grid(int nbr);
// PCG on the assembled reduced camera system (tiles of S), k_pcg.hip
void launch_spcg_prepare(hipStream_t s, int T, const double* S, int ld, double* Minv);
void launch_spcg_init(hipStream_t s, int T, const double* b, const double* Minv, double* x, double* r, double* z, double* p0, double* p1,
                      double* part, double* sc);
void launch_spcg_iteration(hipStream_t s, int k, int T, const double* S, int ld, int n_chunks, const int* chunk_row, const int* chunk_ptr,
                           const int* row_chunk_ptr, const int* tcol, const double* Minv, double* x, double* r, double* z, double* p0,
                           double* p1, double* qpart, double* part_pq, double* part, double* sc, double tol2);
int spcg_chunk_tiles();
void launch_spcg_finish(hipStream_t s, int T, const double* x, const int* iperm, int n_pose, double* y_tan, double* delta);
int pcg_done_slot();
int pcg_iters_slot();
int backsub_mcc_groups(const Visual& v);   // workgroups (= model-cost partials) of launch_backsub_mcc
// (marg / marg_part: the model-cost terms of the window's dense prior as further workgroups of the launch; returns whether they were carried)
bool launch_backsub_mcc(hipStream_t s, const Visual& v, int n_pose, const double* y_pose, double* delta, double* mcc_part,
                        const SmallGroupSet* small = nullptr, int n_small_units = 0, const UpdateRide* upd = nullptr, const MargDev* marg = nullptr,
                        double* marg_part = nullptr);
int small_mcc_first_set(const SmallGroup* groups, double* const* parts, int n_groups, SmallGroupSet* set, int* n_taken);
void launch_negate_pose(hipStream_t s, int n_pose, const double* y, double* delta);
// (returns whether `upd` — the candidate update of a window without Euclidean landmarks — rode in one of the launches)
void launch_update_ride_only(hipStream_t s, const double* delta, const UpdateRide& upd);
bool launch_small_mcc_set(hipStream_t s, const SmallGroup* groups, double* const* parts, int n_groups, const double* delta, const UpdateRide* upd = nullptr,
                          const ZeroStep* zero = nullptr /* the next step's clearing, carried with `upd` */);
void launch_update(hipStream_t s, int nb, const int* blk_xoff, const int* blk_toff, const unsigned char* blk_size,
                   const unsigned char* blk_manifold, const double* x, const double* delta, double* x_cand,
                   double* part /* 2 * nblocks_grid */, int* n_part);
void launch_sum(hipStream_t s, const double* part, int n, double* out, int accumulate);
void launch_zero(hipStream_t s, double* p, int64_t n);
void launch_zero4(hipStream_t s, double* p0, int64_t n0, double* p1, int64_t n1, double* p2, int64_t n2, double* p3, int64_t n3);
void launch_zero_tiles_multi(hipStream_t s, double* S, int ld, const int* tiles_dev, int n_tiles, double* a, int na, double* b, int nb, double* c,
                             int nc, double* radius_slot, double radius);
void launch_copy(hipStream_t s, const double* src, double* dst, int64_t n, int nzero_after);
void launch_final_reduce(hipStream_t s, const ReduceEntry* entries, int n_entries, int n_slots, double* scal, double* host_scal,
                         int* counter = nullptr, double seq = 0.0);

// measurement: the reprojection Jacobian kernel alone
void launch_reproj_jacobian_only(hipStream_t s, const Visual& v, const double* x, const DevCamera* cams,
                                 const DevLoss* losses);

}  // namespace bsg
