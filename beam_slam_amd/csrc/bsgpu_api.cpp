// Host driver of libbsgpu.so: the C-ABI of include/bsgpu.h on top of the HIP kernels.
//
// Replaces, for the reference, everything under `graph_->optimize(options)`
// (bs_optimizers/src/fixed_lag_smoother.cpp:281): [EXT] fuse HashGraph::createProblem (here: finalize(),
// flattening to device tables) and [EXT] ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT /
// SPARSE_NORMAL_CHOLESKY (here: solve(), a restatement of Ceres' TrustRegionMinimizer +
// LevenbergMarquardtStrategy driving device kernels; one host<->device synchronisation per LM iteration).
// There is no CPU fallback: without a HIP device bsgpu_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../../include/bsgpu.h"
#include "bsgpu_internal.h"
#include "dense_plan.h"

using namespace bsg;

namespace {

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

std::string g_create_error;

struct HostMarginal {
  std::vector<int32_t> blocks;
  int rows = 0, cols = 0;
  std::vector<double> A, b, xbar;
};

// Largest reduced (pose-side) system the dense tiled Cholesky takes: the back-substitution keeps the whole solution vector in LDS
// (k_chol.hip: sy[npad] next to the 64x64 tiles, 160 KB per CU).  C2 needs 3 008; 12 288 = 819 keyframes of 15-d states.
constexpr size_t kDenseLimit = 12288;

struct HostGroup {
  int n = 0;
  std::vector<int32_t> idx;
  std::vector<double> consts;
  std::vector<int32_t> loss_kind;
  std::vector<double> loss_a;
};

}  // namespace

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
  bool finalized = false;
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
  SmallGroup small[kNumInternal];
  std::vector<unsigned char> h_small_active[kNumInternal];
  unsigned char* d_small_inactive[kNumInternal] = {nullptr};
  double* d_small_part[kNumInternal] = {nullptr};       // per-factor cost at the current point
  double* d_small_part_cand[kNumInternal] = {nullptr};  // ... at the candidate
  double* d_small_part_mcc[kNumInternal] = {nullptr};   // per-row model-cost-change terms
  ReduceEntry* d_reduce = nullptr;
  int n_reduce = 0;
  double* d_part_upd = nullptr;
  int n_part_upd = 0;
  double *d_S = nullptr, *d_grad = nullptr, *d_hdiag = nullptr, *d_scale = nullptr, *d_dcl = nullptr;
  double *d_delta = nullptr, *d_y = nullptr, *d_scal = nullptr, *d_part = nullptr;
  double* h_scal = nullptr;  // pinned
  double* h_scal_dev = nullptr;  // the same buffer as the device sees it (final_reduce mirrors the step's scalars there)
  bool scal_mirrored = false;    // the last enqueued work ended with a final_reduce that filled the mirror
  // tiled Cholesky plan (dense_plan.h) and its device tables
  DensePlan plan;
  std::vector<uint8_t> tile_adj;   // natural-tile adjacency of the reduced system
  int *d_perm = nullptr, *d_iperm = nullptr, *d_nreal = nullptr, *d_rows_flat = nullptr;
  PanelDesc* d_panels = nullptr;
  int *d_panel_of_tile = nullptr, *d_chain_begin = nullptr, *d_chain_end = nullptr, *d_tile_sync = nullptr;
  double* d_Vinv = nullptr;
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
  hipEvent_t ev_reduce = nullptr;   // recorded after the end-of-step reduction: what the host waits for (work may be queued behind it)
  bool ev_reduce_pending = false;
  // block-sparse PCG path
  bool dense_ok = true, bsr_built = false, use_pcg = false;
  bool spec_J = false;   // residuals + Jacobians currently hold the CANDIDATE's (evaluated ahead of the accept/reject decision)
  int nbr = 0, nblk = 0, pcg_iters_total = 0;
  int *d_row_ptr = nullptr, *d_col = nullptr, *d_diag_slot = nullptr;
  int* d_slots[kNumInternal] = {nullptr};
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
  template <typename T> T* upload(const std::vector<T>& v) {
    T* p = alloc<T>(v.size());
    if (p && !v.empty()) (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
  }
  void free_device() {
    if (stream) (void)hipStreamSynchronize(stream);   // nothing may still be using the buffers that go back to the pool
    for (auto& a : allocs) { pool.emplace(a.second, a.first); pool_bytes += a.second; }
    allocs.clear();
    if (pool_bytes > ((size_t)8 << 30)) release_pool();
    vis = Visual();
    for (auto& g : small) g = SmallGroup();
    d_x = d_xcand = d_x0 = nullptr;
    bsr_built = false;
    destroy_graphs();
  }
  void destroy_graphs() {
    for (hipGraphExec_t* g : {&g_first, &g_accept, &g_reject}) if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
    graphs_tried = graphs_ok = false;
  }
};

namespace {

int fail(bsgpu_ctx* c, int code, const std::string& msg) { c->err = msg; return code; }
// No C++ exception crosses the C-ABI: every entry point is a function-try-block that ends here.
int api_exception(bsgpu_ctx* c) noexcept {
  int code = BSGPU_ERR_INVALID;
  const char* what = "unexpected exception";
  try { throw; }
  catch (const std::bad_alloc&) { code = BSGPU_ERR_DEVICE; what = "out of host memory"; }
  catch (const std::exception& e) { what = e.what(); try { if (c) c->err = std::string("internal error: ") + what; } catch (...) {} return code; }
  catch (...) {}
  try { if (c) c->err = what; } catch (...) {}
  return code;
}

#define HIPCHK(c, call)                                                                         \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, BSGPU_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

void eigen_quat_to_rot(const double* q, double* R) {
  const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0], txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// ---------------------------------------------------------------------------------------------------
// finalize: flatten to device tables.  Restates [EXT] fuse HashGraph::createProblem
// with a deterministic variable index: tangent columns in block order, pose-side
// blocks first, then the landmark blocks that the Schur complement eliminates.
// ---------------------------------------------------------------------------------------------------
int finalize(bsgpu_ctx* c) {
  if (c->finalized) return BSGPU_OK;
  const bool timing = getenv("BSGPU_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[bsgpu finalize] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  c->free_device();
  HIPCHK(c, hipSetDevice(c->device));
  lap("free previous");
  const int nb = c->nb;
  if (nb <= 0) return fail(c, BSGPU_ERR_INVALID, "no parameter blocks");
  // ---- validation + landmark detection (same rule as the oracle)
  std::vector<int> lm_use(nb, 0), other_use(nb, 0);
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sl = 0; sl < ti.nvar; ++sl) {
        const int b = idx[sl];
        if (b < 0 || b >= nb) return fail(c, BSGPU_ERR_INVALID, "factor references block out of range");
        if (c->size[b] != ti.amb[sl]) return fail(c, BSGPU_ERR_INVALID, "block size does not match factor slot");
        if (ti.amb[sl] == 4 && c->manifold[b] != BSGPU_MANIFOLD_QUAT_RIGHT)
          return fail(c, BSGPU_ERR_INVALID, "4-d slot must be a quaternion-manifold block");
        if (t <= 1 && sl == 2) lm_use[b]++; else other_use[b]++;
      }
      if (has_camera(t)) {
        const int cam = idx[ti.nvar];
        if (cam < 0 || cam >= (int)c->cams.size()) return fail(c, BSGPU_ERR_INVALID, "camera index out of range");
      }
    }
  }
  for (const HostMarginal& mg : c->marginals)
    for (int b : mg.blocks) {
      if (b < 0 || b >= nb) return fail(c, BSGPU_ERR_INVALID, "marginal factor references block out of range");
      other_use[b]++;
    }
  // a parameter block no residual block touches is not part of the problem ([EXT] Ceres drops unused parameter blocks from the
  // reduced program; fuse's graph keeps e.g. landmarks whose last observation left the window): treated like a constant block
  if (c->is_const_in.size() != (size_t)nb) c->is_const_in = c->is_const;
  for (int b = 0; b < nb; ++b) c->is_const[b] = (c->is_const_in[b] || lm_use[b] + other_use[b] == 0) ? 1 : 0;
  c->tsize.assign(nb, 0); c->toff.assign(nb, -1); c->is_lm.assign(nb, 0);
  for (int b = 0; b < nb; ++b) {
    if (c->size[b] > 4 || c->size[b] == 0) return fail(c, BSGPU_ERR_UNSUPPORTED, "block sizes 1..4 only");
    if (c->manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT && c->size[b] != 4) return fail(c, BSGPU_ERR_INVALID, "quaternion block must have size 4");
    c->tsize[b] = (c->manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : c->size[b];
    if (c->is_const[b]) continue;
    if (lm_use[b] > 0 && other_use[b] == 0 && c->size[b] == 3 && c->manifold[b] == BSGPU_MANIFOLD_EUCLIDEAN &&
        !(b < (int)c->no_elim.size() && c->no_elim[b])) c->is_lm[b] = 1;
  }
  int to = 0;
  for (int b = 0; b < nb; ++b) if (!c->is_const[b] && !c->is_lm[b]) { c->toff[b] = to; to += c->tsize[b]; }
  c->n_pose = to;
  std::vector<int> lm_index(nb, -1);
  int nl = 0;
  for (int b = 0; b < nb; ++b) if (!c->is_const[b] && c->is_lm[b]) { c->toff[b] = to; to += 3; lm_index[b] = nl++; }
  c->n_tan = to; c->n_lm = nl;
  c->npad = ((c->n_pose + 63) / 64 + 1) * 64;   // real tiles + one tile for the rhs row (dense_plan.h)
  c->dense_ok = (size_t)c->npad <= kDenseLimit;   // above: block-sparse PCG path only (pose-only problems)
  int row = 0;
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) { c->row0[t] = row; row += c->groups[t].n * kTypes[t].m; }
  for (const HostMarginal& mg : c->marginals) row += mg.rows;
  c->n_res = row;

  lap("validate + index");
  // ---- loss table
  std::vector<DevLoss> losses;
  std::map<std::pair<int, double>, int> loss_id;
  int last_kind = -1, last_id = -1;
  double last_a = 0.0;
  auto get_loss = [&](int kind, double a) {
    if (kind == BSGPU_LOSS_TRIVIAL) a = 1.0;
    if (kind == last_kind && a == last_a) return last_id;   // windows use a handful of distinct losses
    last_kind = kind; last_a = a;
    auto key = std::make_pair(kind, a);
    auto it = loss_id.find(key);
    if (it != loss_id.end()) return last_id = it->second;
    DevLoss L; L.kind = kind; L.pad = 0; L.a = a;
    losses.push_back(L);
    return last_id = loss_id[key] = (int)losses.size() - 1;
  };
  get_loss(BSGPU_LOSS_TRIVIAL, 1.0);

  // ---- camera table (online-calib factors fold their constant extrinsic blocks into derived cameras)
  std::vector<DevCamera> cams;
  for (const bsgpu_camera& hc : c->cams) {
    DevCamera d; d.fx = hc.fx; d.fy = hc.fy; d.cx = hc.cx; d.cy = hc.cy;
    std::memcpy(d.R, hc.R_cam_baselink, sizeof(d.R)); std::memcpy(d.t, hc.t_cam_baselink, sizeof(d.t));
    cams.push_back(d);
  }
  std::map<std::tuple<int, int, int>, int> derived_cam;

  // ---- visual factors: camera-pose ids, factors sorted by landmark, pair entries, tile adjacency.
  // Large plain windows are flattened on the device (k_flatten.hip); everything else — and any window the device
  // path declines (online calibration, landmark blocks shared with other factors, more than 8 distinct losses, an
  // orientation block paired with two position blocks) — takes the host path below.  BSGPU_FLATTEN=host|device forces one.
  c->any_inactive = false;
  c->vis_any_inactive = false;
  c->groups[T_REPROJ_DENSE] = HostGroup();
  c->dense_src.clear();
  c->vis_src.clear();
  c->d_vis_src = nullptr;
  auto host_visual = [&]() -> int {
  struct VF { int xq, xp, xl, bq, bp, meta_cam, loss, flags, lm, src; double u, v, w; };
  std::vector<VF> vf;
  for (int t = 0; t <= 1; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      VF e;
      e.bq = idx[0]; e.bp = idx[1];
      e.xq = c->off[idx[0]]; e.xp = c->off[idx[1]]; e.xl = c->off[idx[2]];
      int cam = idx[ti.nvar];
      if (t == 1) {
        const int bqe = idx[3], bpe = idx[4];
        if (!c->is_const[bqe] || !c->is_const[bpe])
          return fail(c, BSGPU_ERR_UNSUPPORTED,
                      "online-calibration reprojection factor with non-constant extrinsic blocks (the reference holds "
                      "them constant: bs_variables/src/orientation_3d.cpp:39-41)");
        auto key = std::make_tuple(bqe, bpe, cam);
        auto it = derived_cam.find(key);
        if (it == derived_cam.end()) {
          // T_CAM_BASELINK = InvertTransform(T_BASELINK_CAM)  (helpers.h:27-35, functor_online_calib.h:52-56)
          double Rbc[9];
          eigen_quat_to_rot(&c->h_x[c->off[bqe]], Rbc);
          const double* pbc = &c->h_x[c->off[bpe]];
          DevCamera d = cams[cam];
          for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d.R[3 * i + j] = Rbc[3 * j + i];
          for (int i = 0; i < 3; ++i) d.t[i] = -(d.R[3 * i] * pbc[0] + d.R[3 * i + 1] * pbc[1] + d.R[3 * i + 2] * pbc[2]);
          cams.push_back(d);
          it = derived_cam.emplace(key, (int)cams.size() - 1).first;
        }
        cam = it->second;
      }
      e.meta_cam = cam;
      e.loss = get_loss(g.loss_kind[f], g.loss_a[f]);
      e.flags = (c->is_const[idx[0]] ? kFlagQConst : 0) | (c->is_const[idx[1]] ? kFlagPConst : 0) |
                (c->is_const[idx[2]] ? kFlagLConst : 0);
      if (e.flags == 7) { c->any_inactive = true; c->vis_any_inactive = true; }
      e.lm = lm_index[idx[2]];
      e.src = (t << 28) | f;
      if (e.lm < 0 && !c->is_const[idx[2]]) {
        // the landmark block is not eliminated (it is shared with another kind of factor): pose-only style group
        HostGroup& dg = c->groups[T_REPROJ_DENSE];
        const int32_t di[4] = {idx[0], idx[1], idx[2], cam};
        dg.idx.insert(dg.idx.end(), di, di + 4);
        dg.consts.insert(dg.consts.end(), &g.consts[(size_t)f * 3], &g.consts[(size_t)f * 3] + 3);
        dg.loss_kind.push_back(g.loss_kind[f]); dg.loss_a.push_back(g.loss_a[f]);
        dg.n++;
        c->dense_src.push_back(e.src);
        continue;
      }
      e.u = g.consts[(size_t)f * 3]; e.v = g.consts[(size_t)f * 3 + 1]; e.w = g.consts[(size_t)f * 3 + 2];
      vf.push_back(e);
    }
  }
  lap("gather visual factors");
  if ((int)cams.size() >= (1 << kMetaCamBits) || (int)losses.size() >= (1 << kMetaLossBits))
    return fail(c, BSGPU_ERR_UNSUPPORTED, "too many distinct cameras / loss functions");
  const int nv = (int)vf.size();
  {  // stable counting sort by landmark (factors of constant landmarks, lm < 0, last)
    std::vector<int> start(nl + 2, 0);
    for (const VF& e : vf) start[(e.lm < 0 ? nl : e.lm) + 1]++;
    for (int l = 0; l <= nl; ++l) start[l + 1] += start[l];
    std::vector<VF> sorted(vf.size());
    for (const VF& e : vf) sorted[start[e.lm < 0 ? nl : e.lm]++] = e;
    vf.swap(sorted);
  }
  lap("sort by landmark");
  Visual& V = c->vis;
  V.n = nv; V.n_lm = nl;
  c->vis_src.resize(nv);
  {
    std::vector<int4> fac(nv);
    std::vector<double2> pix(nv);
    std::vector<double> w(nv);
    std::vector<int> cam_pose(nv), lm_of(nv), lm_start(nl + 1, 0);
    // camera poses = distinct (q block, p block) pairs, numbered in ascending (q, p) order
    std::vector<uint64_t> cp_keys;
    {
      std::vector<int> seen_p(nb, -1);   // fast path: a q block nearly always pairs with one p block
      for (const VF& e : vf) if (seen_p[e.bq] != e.bp) { seen_p[e.bq] = e.bp; cp_keys.push_back(((uint64_t)e.bq << 32) | (uint32_t)e.bp); }
      std::sort(cp_keys.begin(), cp_keys.end());
      cp_keys.erase(std::unique(cp_keys.begin(), cp_keys.end()), cp_keys.end());
    }
    const int k = (int)cp_keys.size();
    std::vector<int> cp_tq, cp_tp, cp_first(nb, -1);   // cp_first[bq] = first camera pose with that q block
    for (int i = 0; i < k; ++i) {
      const int bq = (int)(cp_keys[i] >> 32), bp = (int)(cp_keys[i] & 0xffffffffu);
      cp_tq.push_back(c->toff[bq]); cp_tp.push_back(c->toff[bp]);
      if (cp_first[bq] < 0) cp_first[bq] = i;
    }
    auto cp_of = [&](int bq, int bp) {
      int i = cp_first[bq];
      while ((int)(cp_keys[i] & 0xffffffffu) != bp) ++i;
      return i;
    };
    V.n_cam_pose = k;
    int n_elim = 0;
    for (int i = 0; i < nv; ++i) {
      const VF& e = vf[i];
      fac[i] = make_int4(e.xq, e.xp, e.xl, meta_pack(e.meta_cam, e.loss, e.flags));
      pix[i] = make_double2(e.u, e.v);
      w[i] = e.w;
      cam_pose[i] = cp_of(e.bq, e.bp);
      lm_of[i] = e.lm;
      c->vis_src[i] = e.src;
      if (e.lm >= 0) { lm_start[e.lm + 1]++; n_elim++; }
    }
    for (int l = 0; l < nl; ++l) lm_start[l + 1] += lm_start[l];
    V.n_elim = n_elim;
    lap("camera-pose ids");
    // pair entries (factor a, factor b) of every landmark, grouped by camera-pose pair (ca <= cb); inside a group the
    // order is landmark-major.  Two passes over the landmarks: count per pair key, then fill in place.
    const uint64_t ncp = (uint64_t)std::max(1, V.n_cam_pose);
    std::vector<int> seg_ci, seg_cj, seg_start, ent_fa, ent_fb;
    if (ncp * ncp <= (uint64_t)8 << 20) {
      std::vector<int> start(ncp * ncp + 1, 0);
      for (int l = 0; l < nl; ++l)
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a) {
          const uint64_t ra = (uint64_t)cam_pose[a] * ncp;
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b) if (cam_pose[a] <= cam_pose[b]) start[ra + cam_pose[b] + 1]++;
        }
      for (int f = n_elim; f < nv; ++f) start[(uint64_t)cam_pose[f] * ncp + cam_pose[f] + 1]++;
      for (size_t i = 0; i < ncp * ncp; ++i) start[i + 1] += start[i];
      const size_t n_ent = (size_t)start[ncp * ncp];
      ent_fa.resize(n_ent); ent_fb.resize(n_ent);
      lap("count pair entries");
      // segments (chunks of <= kPairChunk entries of one pair) straight from the counts
      for (uint64_t key = 0; key < ncp * ncp; ++key)
        for (int p0 = start[key]; p0 < start[key + 1]; p0 += kPairChunk) { seg_ci.push_back((int)(key / ncp)); seg_cj.push_back((int)(key % ncp)); seg_start.push_back(p0); }
      std::vector<int> pos(start.begin(), start.end() - 1);
      for (int l = 0; l < nl; ++l)
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a) {
          const uint64_t ra = (uint64_t)cam_pose[a] * ncp;
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
            if (cam_pose[a] <= cam_pose[b]) { const int p = pos[ra + cam_pose[b]]++; ent_fa[p] = a; ent_fb[p] = b; }
        }
      for (int f = n_elim; f < nv; ++f) { const int p = pos[(uint64_t)cam_pose[f] * ncp + cam_pose[f]]++; ent_fa[p] = f; ent_fb[p] = f; }
      lap("fill pair entries");
    } else {   // very many camera poses: comparison sort of explicit entries
      struct Ent { uint64_t key; int fa, fb; };
      std::vector<Ent> ents;
      ents.reserve((size_t)nv * 5);
      for (int l = 0; l < nl; ++l)
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a)
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
            if (cam_pose[a] <= cam_pose[b]) ents.push_back({(uint64_t)cam_pose[a] * ncp + cam_pose[b], a, b});
      for (int f = n_elim; f < nv; ++f) ents.push_back({(uint64_t)cam_pose[f] * ncp + cam_pose[f], f, f});
      std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.key < y.key; });
      ent_fa.resize(ents.size()); ent_fb.resize(ents.size());
      for (size_t i = 0; i < ents.size(); ++i) {
        if (i == 0 || ents[i].key != ents[i - 1].key || (int)i - seg_start.back() >= kPairChunk) {
          seg_ci.push_back((int)(ents[i].key / ncp)); seg_cj.push_back((int)(ents[i].key % ncp)); seg_start.push_back((int)i);
        }
        ent_fa[i] = ents[i].fa; ent_fb[i] = ents[i].fb;
      }
      lap("sort pair entries");
    }
    seg_start.push_back((int)ent_fa.size());
    V.n_seg = (int)seg_ci.size(); V.n_ent = (int)ent_fa.size();
    lap("segments");
    V.fac = c->upload(fac); V.pix = c->upload(pix); V.w = c->upload(w);
    V.cam_pose = c->upload(cam_pose); V.lm_of = c->upload(lm_of); V.lm_start = c->upload(lm_start);
    V.cp_tq = c->upload(cp_tq); V.cp_tp = c->upload(cp_tp);
    V.seg_ci = c->upload(seg_ci); V.seg_cj = c->upload(seg_cj); V.seg_start = c->upload(seg_start);
    V.ent_fa = c->upload(ent_fa); V.ent_fb = c->upload(ent_fb);
    // structural tile adjacency of the reduced system (natural 64-wide tiles) for the Cholesky plan
    const int T = (c->n_pose + 63) / 64;
    c->tile_adj.assign((size_t)T * T, 0);
    auto touch = [&](int ra, int rb) {  // tangent rows ra, rb (start of 3-blocks)
      if (ra < 0 || rb < 0) return;
      for (int a = ra; a < ra + 3; a += 2) for (int b = rb; b < rb + 3; b += 2) {
        c->tile_adj[(size_t)(a / 64) * T + b / 64] = 1; c->tile_adj[(size_t)(b / 64) * T + a / 64] = 1;
      }
    };
    for (int s = 0; s < V.n_seg; ++s) {
      const int i = seg_ci[s], j = seg_cj[s];
      const int ri[2] = {cp_tq[i], cp_tp[i]}, rj[2] = {cp_tq[j], cp_tp[j]};
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) touch(ri[a], rj[b]);
    }
  }
    return BSGPU_OK;
  };
  bool flattened_on_device = false;
  {
    const char* fe = getenv("BSGPU_FLATTEN");
    const bool force_dev = fe && !strcmp(fe, "device"), force_host = fe && !strcmp(fe, "host");
    const HostGroup& g0 = c->groups[BSGPU_F_REPROJ];
    if (!force_host && c->groups[BSGPU_F_REPROJ_ONLINE_CALIB].n == 0 && g0.n > 0 && (force_dev || g0.n >= 20000)) {
      // distinct losses of the reprojection factors (a window has one or two)
      bool ok = true;
      for (int f = 0; f < g0.n && ok; ++f) { get_loss(g0.loss_kind[f], g0.loss_a[f]); ok = losses.size() <= 8; }
      if (ok) {
        std::vector<int> bx(c->off.begin(), c->off.end());
        std::vector<unsigned char> bc(c->is_const.begin(), c->is_const.end());
        const int* d_bx = c->upload(bx); const int* d_bt = c->upload(c->toff);
        const unsigned char* d_bc = c->upload(bc); const int* d_bl = c->upload(lm_index);
        const int T = (c->n_pose + 63) / 64;
        bool all_const = false;
        auto dalloc = [&](size_t bytes) -> void* { return c->alloc<unsigned char>(bytes); };
        const int st = flatten_visual_device(c->stream, dalloc, g0.n, g0.idx.data(), g0.consts.data(), g0.loss_kind.data(), g0.loss_a.data(),
                                             losses, nb, d_bx, d_bt, d_bc, d_bl, nl, T, c->vis, &c->d_vis_src, c->tile_adj, &all_const);
        if (st < 0) return fail(c, BSGPU_ERR_DEVICE, "device error while flattening the reprojection factors");
        if (st == 0) {
          flattened_on_device = true;
          if (all_const) { c->any_inactive = true; c->vis_any_inactive = true; }
          if (c->vis.n_cam_pose >= (1 << 20)) return fail(c, BSGPU_ERR_UNSUPPORTED, "too many camera poses");
        } else {
          c->vis = Visual();
        }
      }
    }
  }
  if (flattened_on_device) lap("flatten on device");
  else { const int rc_host = host_visual(); if (rc_host != BSGPU_OK) return rc_host; }
  Visual& V = c->vis;
  {
    const int nv = V.n;
    V.r = c->alloc<double2>(nv); V.J = c->alloc<double>((size_t)nv * 18); V.CR = c->alloc<double>((size_t)nv * 8);
    V.Linv = c->alloc<double>((size_t)nl * 6); V.z = c->alloc<double>((size_t)nl * 3);
    V.n_cost_part = (nv + 255) / 256;
    V.cost_part = c->alloc<double>(V.n_cost_part);
    V.cost_part_cand = c->alloc<double>(V.n_cost_part);
    V.mcc_part = c->alloc<double>(V.n_cost_part);
    if (!V.J || !V.CR || !V.r) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (visual tables)");
  }
  lap("visual upload + alloc");
  // ---- pose-only groups
  size_t part_max = std::max<size_t>(V.n_cost_part, 2 * ((size_t)nb + 255) / 256 + 2);
  for (int t = 2; t < kNumInternal; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    SmallGroup& sg = c->small[t];
    sg = SmallGroup();
    sg.type = t; sg.n = g.n; sg.m = ti.m; sg.nv = ti.nvar; sg.nc = ti.nconst;
    sg.w_last = ti.amb[ti.nvar - 1] == 4 ? 3 : ti.amb[ti.nvar - 1];
    if (!g.n) continue;
    std::vector<int> xoff((size_t)g.n * ti.nvar), toff((size_t)g.n * ti.nvar), loss(g.n);
    std::vector<unsigned char> active(g.n, 0), inactive(g.n, 0);
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sl = 0; sl < ti.nvar; ++sl) {
        xoff[(size_t)f * ti.nvar + sl] = c->off[idx[sl]];
        toff[(size_t)f * ti.nvar + sl] = c->toff[idx[sl]];
        if (c->toff[idx[sl]] >= c->n_pose) return fail(c, BSGPU_ERR_UNSUPPORTED, "internal: landmark in a pose-only factor");
        if (!c->is_const[idx[sl]]) active[f] = 1;
      }
      inactive[f] = !active[f];
      if (!active[f]) c->any_inactive = true;
      loss[f] = get_loss(g.loss_kind[f], g.loss_a[f]);
      if (active[f]) {
        const int T = (c->n_pose + 63) / 64;
        for (int sa = 0; sa < ti.nvar; ++sa)
          for (int sb = 0; sb < ti.nvar; ++sb) {
            const int ra = c->toff[idx[sa]], rb = c->toff[idx[sb]];
            if (ra < 0 || rb < 0) continue;
            const int wa = c->tsize[idx[sa]], wb = c->tsize[idx[sb]];
            for (int a = ra; a < ra + wa; a += std::max(1, wa - 1)) for (int b = rb; b < rb + wb; b += std::max(1, wb - 1)) c->tile_adj[(size_t)(a / 64) * T + b / 64] = 1;
          }
      }
    }
    sg.xoff = c->upload(xoff); sg.toff = c->upload(toff); sg.consts = c->upload(g.consts); sg.loss = c->upload(loss);
    sg.active = c->upload(active);
    if (has_camera(t)) {
      std::vector<int> camv(g.n);
      for (int f = 0; f < g.n; ++f) camv[f] = g.idx[(size_t)f * ti.nidx + ti.nvar];
      sg.cam = c->upload(camv);
    }
    c->d_small_inactive[t] = c->upload(inactive);
    c->h_small_active[t] = active;
    sg.r = c->alloc<double>((size_t)g.n * ti.m);
    sg.J = c->alloc<double>((size_t)g.n * ti.m * 3 * ti.nvar);
    c->d_small_part[t] = c->alloc<double>((size_t)g.n * ti.m);
    c->d_small_part_cand[t] = c->alloc<double>(g.n);
    c->d_small_part_mcc[t] = c->alloc<double>((size_t)g.n * ti.m);
    part_max = std::max(part_max, (size_t)g.n * ti.m);
  }
  // ---- dense linear priors (marginal factors)
  c->marg.clear();
  {
    int mrow = 0;
    for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) mrow += c->groups[t].n * kTypes[t].m;
    const int T = (c->n_pose + 63) / 64;
    for (const HostMarginal& mg : c->marginals) {
      bsgpu_ctx::MargCtx mc;
      std::vector<int> bx, bs, bq, bc, ba, col_t, col_blk;
      int cols = 0, amb = 0;
      mc.active = false;
      for (size_t i = 0; i < mg.blocks.size(); ++i) {
        const int b = mg.blocks[i];
        bx.push_back(c->off[b]); bs.push_back(c->size[b]); bq.push_back(c->manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT ? 1 : 0);
        bc.push_back(cols); ba.push_back(amb);
        for (int k = 0; k < c->tsize[b]; ++k) { col_t.push_back(c->is_const[b] ? -1 : c->toff[b] + k); col_blk.push_back((int)i); }
        cols += c->tsize[b]; amb += c->size[b];
        if (!c->is_const[b]) mc.active = true;
      }
      if (cols != mg.cols || amb != (int)mg.xbar.size()) return fail(c, BSGPU_ERR_INVALID, "marginal factor: A / xbar sizes do not match its blocks");
      for (int t : col_t) if (t >= c->n_pose) return fail(c, BSGPU_ERR_UNSUPPORTED, "internal: eliminated block in a marginal factor");
      if (!mc.active) c->any_inactive = true;
      MargDev& d = mc.dev;
      d.rows = mg.rows; d.cols = cols; d.nblk = (int)mg.blocks.size();
      d.blk_xoff = c->upload(bx); d.blk_size = c->upload(bs); d.blk_quat = c->upload(bq); d.blk_col = c->upload(bc); d.blk_amb = c->upload(ba);
      d.col_t = c->upload(col_t); d.col_blk = c->upload(col_blk);
      d.A = c->upload(mg.A); d.b = c->upload(mg.b); d.xbar = c->upload(mg.xbar);
      d.delta = c->alloc<double>(cols); d.D = c->alloc<double>((size_t)d.nblk);
      d.r = c->alloc<double>(mg.rows); d.J = c->alloc<double>((size_t)mg.rows * cols);
      mc.part = c->alloc<double>(mg.rows); mc.part_cand = c->alloc<double>(mg.rows); mc.part_mcc = c->alloc<double>(mg.rows);
      if (!d.J || !mc.part_mcc) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (marginal factor)");
      mc.row0 = mrow; mrow += mg.rows;
      part_max = std::max(part_max, (size_t)mg.rows);
      if (mc.active)   // a dense prior couples every pair of its blocks
        for (int ta : col_t) for (int tb : col_t) if (ta >= 0 && tb >= 0) c->tile_adj[(size_t)(ta / 64) * T + tb / 64] = 1;
      c->marg.push_back(mc);
    }
  }
  if (losses.size() >= (1u << kMetaLossBits)) return fail(c, BSGPU_ERR_UNSUPPORTED, "too many distinct loss functions");
  c->d_cams = c->upload(cams);
  for (int t = 2; t < kNumInternal; ++t) c->small[t].cams = c->d_cams;
  c->d_losses = c->upload(losses);
  lap("pose-only groups + priors");
  // ---- blocks
  {
    std::vector<int> bx(c->off.begin(), c->off.end());
    c->d_blk_xoff = c->upload(bx);
    c->d_blk_toff = c->upload(c->toff);
    std::vector<unsigned char> sz(c->size.begin(), c->size.end()), mf(c->manifold.begin(), c->manifold.end());
    c->d_blk_size = c->upload(sz); c->d_blk_manifold = c->upload(mf);
    c->d_x = c->upload(c->h_x); c->d_x0 = c->upload(c->h_x);
    c->d_xcand = c->alloc<double>(c->h_x.size());
  }
  // ---- dense system + vectors
  if (c->dense_ok) {
    c->d_S = c->alloc<double>((size_t)c->npad * c->npad);
    if (!c->d_S) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (reduced system)");
  }
  c->d_grad = c->alloc<double>(c->n_tan); c->d_hdiag = c->alloc<double>(c->n_tan);
  c->d_scale = c->alloc<double>(c->n_tan); c->d_dcl = c->alloc<double>(c->n_tan);
  c->d_delta = c->alloc<double>(c->n_tan); c->d_y = c->alloc<double>(c->npad);
  c->d_scal = c->alloc<double>(SC_NUM);
  c->d_part = c->alloc<double>(part_max + 8);
  if (!c->h_scal) {
    HIPCHK(c, hipHostMalloc((void**)&c->h_scal, sizeof(double) * SC_NUM, hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&c->h_scal_dev, c->h_scal, 0) != hipSuccess) { (void)hipGetLastError(); c->h_scal_dev = nullptr; }
  }
  if (!c->h_radius) HIPCHK(c, hipHostMalloc((void**)&c->h_radius, sizeof(double)));
  chol_prepare();
  // hipGraph replay of the LM step is opt-in (BSGPU_GRAPH=1): on ROCm 7.2 the replay inserts a ~0.9 ms bubble
  // inside the long dependent kernel chain (profiles/README.md), which cancels what it saves on launches
  c->use_graphs = getenv("BSGPU_GRAPH") != nullptr;
  HIPCHK(c, hipMemset(c->d_scal, 0, sizeof(double) * SC_NUM));
  HIPCHK(c, hipMemset(c->d_delta, 0, sizeof(double) * std::max(1, c->n_tan)));
  lap("blocks + dense buffers");
  // ---- tiled Cholesky plan: nested-dissection tile order, symbolic factorisation, step schedule
  {
    const char* e = getenv("BSGPU_CHAINS");
    const int max_chains = e ? std::max(1, atoi(e)) : 16;
    const int T = (c->n_pose + 63) / 64;
    if (c->tile_adj.size() != (size_t)T * T) c->tile_adj.assign((size_t)T * T, 0);
    const char* e2 = getenv("BSGPU_MIN_PIECE");
    const char* e3 = getenv("BSGPU_SHARED");   // panels of one step may update the same tiles (atomics): on unless BSGPU_SHARED=0
    c->plan.build(c->n_pose, c->tile_adj, c->dense_ok ? max_chains : 1, e2 ? std::max(1, atoi(e2)) : 1, !(e3 && atoi(e3) == 0));
    c->npad = c->plan.npad;
    if (timing) fprintf(stderr, "[bsgpu finalize] Cholesky plan: %d tiles, %d pieces, %d panel steps, %d back-substitution launches\n", c->plan.T, c->plan.n_pieces,
                        c->plan.n_steps(), (int)c->plan.bs_group_off.size() - 1);
    std::vector<int> iperm(T + 1, -1);
    for (int t = 0; t < T; ++t) iperm[c->plan.perm[t]] = t;
    c->d_perm = c->upload(c->plan.perm); c->d_iperm = c->upload(iperm); c->d_nreal = c->upload(c->plan.nreal);
    c->d_rows_flat = c->upload(c->plan.rows_flat);
    c->d_panels = c->upload(c->plan.panels);
    c->d_panel_of_tile = c->upload(c->plan.panel_of_tile);
    c->d_tile_sync = c->upload(c->plan.tile_sync);
    c->d_chain_begin = c->upload(c->plan.chain_begin); c->d_chain_end = c->upload(c->plan.chain_end);
    c->d_Vinv = c->alloc<double>((size_t)std::max(1, T) * chol_vinv_stride());
    c->d_ytan = c->alloc<double>(std::max(1, c->n_pose));
    if (c->dense_ok) { c->d_Lp = c->alloc<double>((size_t)c->npad * c->npad); if (!c->d_Lp) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (L panels)"); }
  }
  {
    c->n_part_upd = (nb + 255) / 256;
    c->d_part_upd = c->alloc<double>(2 * (size_t)c->n_part_upd + 2);
    std::vector<ReduceEntry> tab;
    if (c->vis.n) {
      tab.push_back({c->vis.cost_part, c->vis.n_cost_part, 1, 0, SC_COST_X});
      tab.push_back({c->vis.cost_part_cand, c->vis.n_cost_part, 1, 0, SC_COST_CAND});
      tab.push_back({c->vis.mcc_part, c->vis.n_cost_part, 1, 0, SC_MCC});
    }
    for (int t = 2; t < kNumInternal; ++t) {
      if (!c->small[t].n) continue;
      tab.push_back({c->d_small_part[t], c->small[t].n, 1, 0, SC_COST_X});
      tab.push_back({c->d_small_part_cand[t], c->small[t].n, 1, 0, SC_COST_CAND});
      tab.push_back({c->d_small_part_mcc[t], (c->small[t].n * c->small[t].m + 127) / 128, 1, 0, SC_MCC});   // one partial per workgroup of small_mcc_kernel
    }
    for (const auto& mc : c->marg) {
      if (!mc.active) continue;
      tab.push_back({mc.part, mc.dev.rows, 1, 0, SC_COST_X});
      tab.push_back({mc.part_cand, mc.dev.rows, 1, 0, SC_COST_CAND});
      tab.push_back({mc.part_mcc, mc.dev.rows, 1, 0, SC_MCC});
    }
    tab.push_back({c->d_part_upd, c->n_part_upd, 2, 0, SC_STEP_NORM2});
    tab.push_back({c->d_part_upd, c->n_part_upd, 2, 1, SC_X_NORM2});
    c->n_reduce = (int)tab.size();
    c->d_reduce = c->upload(tab);
  }
  lap("plan + reduce table");
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipGetLastError());
  lap("device sync");
  c->finalized = true;
  return BSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// device steps of one LM iteration
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// block-sparse structure of the pose-only normal equations (3x3 blocks), built on first use
// ---------------------------------------------------------------------------------------------------
int build_bsr(bsgpu_ctx* c) {
  if (c->bsr_built) return BSGPU_OK;
  if (c->vis.n > 0)
    return fail(c, BSGPU_ERR_UNSUPPORTED, c->dense_ok ? "PCG path covers pose-only problems; landmark problems use the Schur + dense path"
                                                      : "window too large: the reduced camera system exceeds the 12288 dimensions of the dense Schur path (819 keyframes of 15-d states) and the PCG path covers pose-only problems");
  if (!c->marginals.empty()) return fail(c, BSGPU_ERR_UNSUPPORTED, "PCG path does not take dense marginal factors");
  for (int b = 0; b < c->nb; ++b)
    if (!c->is_const[b] && c->tsize[b] != 3) return fail(c, BSGPU_ERR_UNSUPPORTED, "PCG path needs 3-dimensional tangent blocks");
  const int nbr = c->n_pose / 3;
  std::vector<uint64_t> keys;
  for (int b = 0; b < nbr; ++b) keys.push_back(((uint64_t)b << 32) | (uint32_t)b);
  for (int t = 2; t < kNumInternal; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      if (!c->h_small_active[t][f]) continue;
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sa = 0; sa < ti.nvar; ++sa) for (int sb = 0; sb < ti.nvar; ++sb) {
        const int ra = c->toff[idx[sa]], rb = c->toff[idx[sb]];
        if (ra < 0 || rb < 0) continue;
        keys.push_back(((uint64_t)(ra / 3) << 32) | (uint32_t)(rb / 3));
      }
    }
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  const int nblk = (int)keys.size();
  std::vector<int> row_ptr(nbr + 1, 0), col(nblk), diag_slot(nbr, -1);
  for (int i = 0; i < nblk; ++i) {
    const int r = (int)(keys[i] >> 32), cc = (int)(keys[i] & 0xffffffffu);
    row_ptr[r + 1]++; col[i] = cc;
    if (r == cc) diag_slot[r] = i;
  }
  for (int r = 0; r < nbr; ++r) row_ptr[r + 1] += row_ptr[r];
  for (int t = 2; t < kNumInternal; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    if (!g.n) continue;
    std::vector<int> slots((size_t)g.n * ti.nvar * ti.nvar, -1);
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sa = 0; sa < ti.nvar; ++sa) for (int sb = 0; sb < ti.nvar; ++sb) {
        const int ra = c->toff[idx[sa]], rb = c->toff[idx[sb]];
        if (ra < 0 || rb < 0 || !c->h_small_active[t][f]) continue;
        const uint64_t key = ((uint64_t)(ra / 3) << 32) | (uint32_t)(rb / 3);
        slots[((size_t)f * ti.nvar + sa) * ti.nvar + sb] = (int)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
      }
    }
    c->d_slots[t] = c->upload(slots);
  }
  c->nbr = nbr; c->nblk = nblk;
  c->d_row_ptr = c->upload(row_ptr); c->d_col = c->upload(col); c->d_diag_slot = c->upload(diag_slot);
  c->d_val = c->alloc<double>((size_t)nblk * 9); c->d_Minv = c->alloc<double>((size_t)nbr * 9);
  c->d_rhs = c->alloc<double>(c->n_pose);
  c->d_px = c->alloc<double>(c->n_pose); c->d_pr = c->alloc<double>(c->n_pose); c->d_pz = c->alloc<double>(c->n_pose);
  c->d_pp = c->alloc<double>(c->n_pose); c->d_pp1 = c->alloc<double>(c->n_pose); c->d_pq = c->alloc<double>(c->n_pose);
  c->d_ppart = c->alloc<double>((size_t)pcg_spmv_grid(nbr) + 8); c->d_ppart2 = c->alloc<double>(4 * ((size_t)(nbr + 255) / 256) + 8);
  c->d_psc = c->alloc<double>(pcg_num_scalars());
  if (!c->d_val || !c->d_pq || !c->d_psc) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (block-sparse system)");
  c->bsr_built = true;
  return BSGPU_OK;
}

void assemble_pcg(bsgpu_ctx* c, const bsgpu_options& o, double radius, bool new_J, bool first) {
  hipStream_t s = c->stream;
  launch_zero(s, c->d_val, (int64_t)c->nblk * 9);
  launch_zero(s, c->d_rhs, c->n_pose);
  launch_zero(s, c->d_grad, c->n_pose);
  launch_zero(s, c->d_hdiag, c->n_pose);
  for (int t = 2; t < kNumInternal; ++t)
    launch_bsr_assemble(s, c->small[t], c->d_slots[t], c->d_val, c->d_rhs, c->d_grad, c->d_hdiag);
  launch_bsr_finish_diag(s, c->nbr, c->d_diag_slot, c->d_val, c->d_hdiag, radius, first ? 1 : 0, new_J ? 1 : 0, o.jacobi_scaling,
                         o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_Minv);
  if (new_J) {
    launch_grad_norms(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_scal);
  }
}

// (H + Lambda) y = g by block-Jacobi PCG; the stop test lives on the device, the host looks at it every 20 iterations
void pcg_solve(bsgpu_ctx* c, const bsgpu_options& o) {
  hipStream_t s = c->stream;
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

// residuals (+ Jacobians) of every factor group; per-group cost partials go to the arrays the
// end-of-step reduction sums (current point: slot SC_COST_X, candidate: SC_COST_CAND)
void eval_all(bsgpu_ctx* c, const double* x, bool with_J, int slot) {
  hipStream_t s = c->stream;
  const bool cand = slot == SC_COST_CAND;
  if (c->vis.n) launch_reproj_eval(s, c->vis, x, c->d_cams, c->d_losses, with_J, cand ? c->vis.cost_part_cand : c->vis.cost_part);
  const bool imu_pair = c->small[BSGPU_F_IMU_DELTA].n > 0 && c->small[BSGPU_F_IMU_PRIOR].n > 0;
  if (imu_pair)
    launch_imu_eval(s, c->small[BSGPU_F_IMU_DELTA], c->small[BSGPU_F_IMU_PRIOR], x, c->d_losses, with_J,
                    cand ? c->d_small_part_cand[BSGPU_F_IMU_DELTA] : c->d_small_part[BSGPU_F_IMU_DELTA],
                    cand ? c->d_small_part_cand[BSGPU_F_IMU_PRIOR] : c->d_small_part[BSGPU_F_IMU_PRIOR]);
  for (int t = 2; t < kNumInternal; ++t) {
    if (imu_pair && (t == BSGPU_F_IMU_DELTA || t == BSGPU_F_IMU_PRIOR)) continue;
    if (c->small[t].n) launch_small_eval(s, c->small[t], x, c->d_losses, with_J, cand ? c->d_small_part_cand[t] : c->d_small_part[t]);
  }
  for (const auto& mc : c->marg)
    if (mc.active) launch_marg_eval(s, mc.dev, x, with_J, cand ? mc.part_cand : mc.part);
}
void final_reduce(bsgpu_ctx* c) {
  launch_final_reduce(c->stream, c->d_reduce, c->n_reduce, SC_X_NORM2 + 1, c->d_scal, c->h_scal_dev);
  c->scal_mirrored = c->h_scal_dev != nullptr && c->n_reduce > 0;
  if (!c->ev_reduce && hipEventCreateWithFlags(&c->ev_reduce, hipEventDisableTiming) != hipSuccess) { c->ev_reduce = nullptr; (void)hipGetLastError(); }
  c->ev_reduce_pending = c->scal_mirrored && c->ev_reduce && hipEventRecord(c->ev_reduce, c->stream) == hipSuccess;
}

void assemble(bsgpu_ctx* c, const bsgpu_options& o, double radius, bool new_J, bool first) {
  if (c->use_pcg) { assemble_pcg(c, o, radius, new_J, first); return; }
  hipStream_t s = c->stream;
  // one launch clears the reduced system, gradient, diagonal and the scalars of this step (GRAD_MAX, GRAD_NORM2, CHOL_FAIL)
  // ... and carries the radius of this step (not under graph replay, whose kernel arguments are frozen)
  launch_zero_multi(s, c->d_S, (int64_t)c->npad * c->npad, c->d_grad, c->n_pose, c->d_hdiag, c->n_pose,
                    new_J ? c->d_scal + SC_GRAD_MAX : c->d_scal + SC_CHOL_FAIL, new_J ? 3 : 1,
                    c->use_graphs ? nullptr : c->d_scal + SC_RADIUS, radius);
  c->scal_mirrored = false;
  launch_landmark(s, c->vis, c->n_pose, c->d_scal + SC_RADIUS, first ? 1 : 0, new_J ? 1 : 0, o.jacobi_scaling, o.min_lm_diagonal,
                  o.max_lm_diagonal, c->d_scale, c->d_dcl, c->d_grad);
  launch_pairs(s, c->vis, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_perm);
  launch_small_assemble_set(s, c->small + 2, kNumInternal - 2, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_perm);
  for (const auto& mc : c->marg)
    if (mc.active) launch_marg_assemble(s, mc.dev, c->d_S, c->npad, c->plan.rhs_row, c->d_grad, c->d_hdiag, c->d_perm);
  if (new_J)   // the LM diagonal and the gradient norms both follow the assembly and do not depend on each other: one launch
    launch_grad_norms_pose_diag(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_grad, c->d_scal,
                                c->n_pose, c->d_S, c->npad, c->d_hdiag, c->d_scal + SC_RADIUS, first ? 1 : 0, 1, o.jacobi_scaling,
                                o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_iperm);
  else
    launch_pose_diag(s, c->n_pose, c->d_S, c->npad, c->d_hdiag, c->d_scal + SC_RADIUS, 0, 0, o.jacobi_scaling,
                     o.min_lm_diagonal, o.max_lm_diagonal, c->d_scale, c->d_dcl, c->npad, c->d_iperm);
}

// Cholesky of the (padded, rhs-augmented, solver-ordered) reduced system in S and the solve L^T y = y',
// following the plan's step schedule.  y comes back in solver order (npad entries).
struct DenseDev {
  const int *perm, *nreal, *rows_flat;
  const PanelDesc* panels;
  double *Lp, *Vinv;
  const int *panel_of_tile, *chain_begin, *chain_end;
  int* tile_sync;   // [expected arrivals | arrival counters] per tile (dense_plan.h)
};
void dense_factor(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* scal) {
  const int ld = P.npad;
  for (int st = 0; st < P.n_steps(); ++st) {
    // (tiles no look-ahead factors are factored inside the panel step itself: PanelDesc::self_potrf)
    launch_chol_panel_step(s, S, D.Lp, ld, D.panels + P.step_off[st], P.step_off[st + 1] - P.step_off[st], P.step_maxrows[st],
                           D.rows_flat, D.nreal, D.Vinv, scal, D.tile_sync, P.panels.data() + P.step_off[st], P.rows_flat.data());
  }
}
void dense_factor_solve(hipStream_t s, const DensePlan& P, const DenseDev& D, double* S, double* y, double* scal) {
  const int ld = P.npad;
  dense_factor(s, P, D, S, scal);
  // y' = the rhs row after forward substitution: row rhs_row of the shadow matrix (the rhs tile is an
  // off-diagonal row tile of every panel)
  const double* rhs_row = D.Lp + (size_t)P.rhs_row * ld;
  const bool single_root = P.bs_group_off.size() > 1 && P.bs_group_off[1] - P.bs_group_off[0] == 1;
  if (!single_root) launch_copy(s, rhs_row, y, (int64_t)P.T * 64, 64);   // (a single root chain copies it itself on the way)
  // one launch per group of chains: root separator, the separator levels below it, then every piece (dense_plan.h)
  for (size_t g = 0; g + 1 < P.bs_group_off.size(); ++g) {
    const int c0 = P.bs_group_off[g], c1 = P.bs_group_off[g + 1];
    int max_len = 1;
    for (int i = c0; i < c1; ++i) max_len = std::max(max_len, P.chain_end[i] - P.chain_begin[i]);
    launch_chol_backsolve_chains(s, S, D.Lp, D.Vinv, ld, D.panels, D.panel_of_tile, D.chain_begin + c0, D.chain_end + c0, c1 - c0,
                                 D.rows_flat, D.nreal, y, P.npad, max_len, (g == 0 && single_root) ? rhs_row : nullptr);
  }
}

void linear_solve_and_candidate(bsgpu_ctx* c, const bsgpu_options& o) {
  hipStream_t s = c->stream;
  if (c->use_pcg) {
    pcg_solve(c, o);
    launch_negate_pose(s, c->n_pose, c->d_px, c->d_delta);
  } else if (c->n_pose > 0) {
    const DenseDev D{c->d_perm, c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
                     c->d_panel_of_tile, c->d_chain_begin, c->d_chain_end, c->d_tile_sync};
    dense_factor_solve(s, c->plan, D, c->d_S, c->d_y, c->d_scal);
    launch_y_to_delta(s, c->n_pose, c->d_y, c->d_perm, c->d_ytan, c->d_delta);
  }
  launch_backsub_landmarks(s, c->vis, c->n_pose, c->d_ytan, c->d_delta);
  // model cost change terms, candidate point and its cost: partial arrays only, summed once at the end
  if (c->vis.n) launch_mcc(s, c->vis, c->n_pose, c->d_delta, c->vis.mcc_part);
  launch_small_mcc_set(s, c->small + 2, c->d_small_part_mcc + 2, kNumInternal - 2, c->d_delta);
  for (const auto& mc : c->marg)
    if (mc.active) launch_marg_mcc(s, mc.dev, c->d_delta, mc.part_mcc);
  int n_part = 0;
  launch_update(s, c->nb, c->d_blk_xoff, c->d_blk_toff, c->d_blk_size, c->d_blk_manifold, c->d_x, c->d_delta, c->d_xcand,
                c->d_part_upd, &n_part);
  eval_all(c, c->d_xcand, false, SC_COST_CAND);
  final_reduce(c);
}

// ---------------------------------------------------------------------------------------------------
// one LM step = [x <- x_cand] [evaluate J] assemble -> factor -> back-substitute -> candidate -> cost.
// The three variants are captured once per finalized problem as hipGraphs and replayed: the host-side
// launch cost (~4.5 us per kernel, > 100 kernels per step) otherwise bounds the iteration rate.
// ---------------------------------------------------------------------------------------------------
enum StepKind { STEP_FIRST = 0, STEP_ACCEPT = 1, STEP_REJECT = 2 };

// gradient_only: the iteration budget is used up — the point just accepted still needs its cost and gradient norms for the
// iteration record, but no step will be taken from it: evaluation + assembly (which produces the gradient), no factorisation,
// no candidate
void enqueue_step(bsgpu_ctx* c, const bsgpu_options& o, int kind, double radius, bool gradient_only = false) {
  hipStream_t s = c->stream;
  if (kind == STEP_ACCEPT) {
    // the accepted candidate becomes the current point: a pointer swap (every launch takes x as an argument; the next update
    // rewrites all of the other buffer) — except under graph replay, whose kernel arguments are frozen
    if (c->use_graphs) launch_copy(s, c->d_xcand, c->d_x, (int64_t)c->h_x.size(), 0);
    else std::swap(c->d_x, c->d_xcand);
  }
  // Jacobians at the current point: new for a first / accepted step — unless they were evaluated ahead at the candidate that has
  // just been accepted (below) — and to be restored for a rejected one if that evaluation overwrote them
  const bool have_J = (kind == STEP_ACCEPT && c->spec_J) || (kind == STEP_REJECT && !c->spec_J);
  if (!have_J) eval_all(c, c->d_x, true, SC_COST_X);
  c->spec_J = false;
  assemble(c, o, radius, kind != STEP_REJECT, kind == STEP_FIRST);
  if (gradient_only) { final_reduce(c); return; }
  linear_solve_and_candidate(c, o);
  // The host now waits for this step's scalars and decides; in the common case (accepted) the next thing the device needs is the
  // residuals and Jacobians at the candidate: evaluated ahead, underneath the host round trip (~26 us per iteration otherwise
  // idle).  A rejected step pays for it with a re-evaluation at the current point (above).
  if (!c->use_graphs) {
    eval_all(c, c->d_xcand, true, SC_COST_X);
    c->spec_J = true;
  }
}

bool same_graph_options(const bsgpu_options& a, const bsgpu_options& b) {
  return a.jacobi_scaling == b.jacobi_scaling && a.min_lm_diagonal == b.min_lm_diagonal && a.max_lm_diagonal == b.max_lm_diagonal;
}

void build_graphs(bsgpu_ctx* c, const bsgpu_options& o) {
  if (c->graphs_tried && same_graph_options(o, c->graph_opts)) return;
  c->destroy_graphs();
  c->graphs_tried = true;
  c->graph_opts = o;
  if (!c->use_graphs || c->use_pcg) return;   // the PCG path synchronises inside a step: stays eager
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

void run_step(bsgpu_ctx* c, const bsgpu_options& o, int kind, double radius, bool gradient_only = false) {
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
  enqueue_step(c, o, kind, radius, gradient_only);
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
  if (c->scal_mirrored && c->ev_reduce_pending) {
    HIPCHK(c, hipEventSynchronize(c->ev_reduce));   // the step's scalars are in the pinned mirror; kernels queued behind the reduction keep running
  } else {
    if (!c->scal_mirrored) HIPCHK(c, hipMemcpyAsync(c->h_scal, c->d_scal, sizeof(double) * SC_NUM, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->scal_mirrored = false; c->ev_reduce_pending = false;
  return BSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// [EXT] ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy, restated
// ---------------------------------------------------------------------------------------------------
int solve(bsgpu_ctx* c, const bsgpu_options& o, bsgpu_summary& sum) {
  using clk = std::chrono::steady_clock;
  const auto t_start = clk::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(clk::now() - t_start).count(); };
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  std::memset(&sum, 0, sizeof(sum));
  c->iters.clear();
  sum.num_parameters_tangent = c->n_tan;
  sum.num_residuals = c->n_res;
  c->use_pcg = (o.linear_solver_type == BSGPU_LINEAR_PCG) || (o.linear_solver_type == BSGPU_LINEAR_AUTO && !c->dense_ok);
  if (!c->use_pcg && !c->dense_ok)
    return fail(c, BSGPU_ERR_UNSUPPORTED, "reduced system larger than 12288: the dense exact path does not apply; use BSGPU_LINEAR_AUTO or BSGPU_LINEAR_PCG");
  if (c->use_pcg) { rc = build_bsr(c); if (rc != BSGPU_OK) return rc; }
  c->pcg_iters_total = 0;
  sum.linear_solver_used = c->use_pcg ? BSGPU_LINEAR_PCG : BSGPU_LINEAR_SCHUR_CHOLESKY;
  hipStream_t s = c->stream;
  hipEvent_t ev0, ev1;
  HIPCHK(c, hipEventCreate(&ev0)); HIPCHK(c, hipEventCreate(&ev1));
  HIPCHK(c, hipEventRecord(ev0, s));

  // iteration zero
  double fixed = 0.0;
  if (c->any_inactive) {
    // cost of residual blocks whose parameter blocks are all constant (Ceres: fixed_cost)
    launch_zero(s, c->d_scal + SC_FIXED_COST, 1);
    for (int t = 2; t < kNumInternal; ++t) {
      if (!c->small[t].n) continue;
      SmallGroup g = c->small[t];
      g.active = c->d_small_inactive[t];
      launch_small_eval(s, g, c->d_x, c->d_losses, false, c->d_small_part[t]);
      launch_sum(s, c->d_small_part[t], g.n, c->d_scal + SC_FIXED_COST, 1);
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
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  build_graphs(c, o);
  run_step(c, o, STEP_FIRST, radius);
  rc = fetch_scalars(c);
  if (rc != BSGPU_OK) return rc;
  fixed = c->any_inactive ? c->h_scal[SC_FIXED_COST] : 0.0;
  double x_cost = c->h_scal[SC_COST_X];
  bsgpu_iteration it;
  std::memset(&it, 0, sizeof(it));
  it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost + fixed;
  it.gradient_max_norm = c->h_scal[SC_GRAD_MAX]; it.gradient_norm = std::sqrt(c->h_scal[SC_GRAD_NORM2]);
  sum.initial_cost = x_cost + fixed; sum.fixed_cost = fixed;
  sum.termination_type = BSGPU_NO_CONVERGENCE;
  const char* msg = "";
  if (!std::isfinite(x_cost)) {
    sum.termination_type = BSGPU_FAILURE; msg = "Initial cost is not finite.";
    sum.final_cost = sum.initial_cost;
  } else {
    int num_consecutive_invalid = 0;
    // `pending` = a step (linear solve + candidate evaluation) has been computed for the current x/radius
    while (true) {
      if (it.step_is_successful) { if (it.iteration > 0) sum.num_successful_steps++; } else sum.num_unsuccessful_steps++;
      it.trust_region_radius = radius;
      c->iters.push_back(it);
      if (o.max_solver_time_in_seconds > 0 && elapsed() >= o.max_solver_time_in_seconds) { msg = "Maximum solver time reached."; break; }
      if (it.iteration >= o.max_num_iterations) { msg = "Maximum number of iterations reached."; break; }
      if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Gradient tolerance reached."; break; }
      if (radius <= o.min_trust_region_radius) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Minimum trust region radius reached."; break; }
      const bsgpu_iteration prev = it;
      std::memset(&it, 0, sizeof(it));
      it.iteration = prev.iteration + 1;
      it.gradient_max_norm = prev.gradient_max_norm; it.gradient_norm = prev.gradient_norm;
      sum.num_linear_solves++;
      // the step for (x, radius) is already on the host: h_scal
      const double mcc = c->h_scal[SC_MCC];
      const bool lin_ok = !(c->h_scal[SC_CHOL_FAIL] > 0.0) && std::isfinite(mcc) && std::isfinite(c->h_scal[SC_STEP_NORM2]);
      it.model_cost_change = lin_ok ? mcc : 0.0;
      it.step_is_valid = lin_ok && mcc > 0.0;
      if (!it.step_is_valid) {
        if (++num_consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
          sum.termination_type = BSGPU_FAILURE;
          msg = "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps.";
          break;
        }
        radius = radius / decrease_factor; decrease_factor *= 2.0;
        it.cost = x_cost + fixed; it.step_is_successful = 0;
        if (it.iteration >= o.max_num_iterations) continue;   // the loop ends at its top: a step from here would never be looked at
        run_step(c, o, STEP_REJECT, radius);
        rc = fetch_scalars(c);
        if (rc != BSGPU_OK) return rc;
        continue;
      }
      num_consecutive_invalid = 0;
      double cand_cost = c->h_scal[SC_COST_CAND];
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      it.step_norm = std::sqrt(c->h_scal[SC_STEP_NORM2]);
      const double x_norm = std::sqrt(c->h_scal[SC_X_NORM2]);
      if (it.step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Parameter tolerance reached."; break; }
      it.cost_change = x_cost - cand_cost;
      if (std::fabs(it.cost_change) <= o.function_tolerance * x_cost) { sum.termination_type = BSGPU_CONVERGENCE; msg = "Function tolerance reached."; break; }
      it.relative_decrease = (x_cost - cand_cost) / mcc;
      const bool last_iteration = it.iteration >= o.max_num_iterations;
      if (it.relative_decrease > o.min_relative_decrease) {
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
        radius = std::min(o.max_trust_region_radius, radius);
        decrease_factor = 2.0;
        it.step_is_successful = 1;
        // the next step is computed right away so that one synchronisation per iteration suffices; when this was the last
        // iteration the budget allows, only the accepted point's cost and gradient are (a full step would be thrown away)
        run_step(c, o, STEP_ACCEPT, radius, last_iteration);
      } else {
        it.step_is_successful = 0;
        radius = radius / decrease_factor; decrease_factor *= 2.0;
        it.cost = cand_cost + fixed;
        if (last_iteration) continue;
        run_step(c, o, STEP_REJECT, radius);
      }
      rc = fetch_scalars(c);
      if (rc != BSGPU_OK) return rc;
      if (it.step_is_successful) {
        x_cost = c->h_scal[SC_COST_X];
        it.cost = x_cost + fixed;
        it.gradient_max_norm = c->h_scal[SC_GRAD_MAX];
        it.gradient_norm = std::sqrt(c->h_scal[SC_GRAD_NORM2]);
      }
    }
    sum.final_cost = x_cost + fixed;
  }
  HIPCHK(c, hipEventRecord(ev1, s));
  HIPCHK(c, hipEventSynchronize(ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  sum.device_time_in_seconds = ms * 1e-3;
  c->scal_mirrored = false; c->ev_reduce_pending = false; c->spec_J = false;   // (the stream has drained: nothing of this solve is pending)
  sum.num_iterations = (int)c->iters.size() - 1;
  sum.num_inner_iterations = c->pcg_iters_total;
  sum.is_solution_usable = (sum.termination_type == BSGPU_CONVERGENCE || sum.termination_type == BSGPU_NO_CONVERGENCE) ? 1 : 0;
  sum.total_time_in_seconds = elapsed();
  std::snprintf(sum.message, sizeof(sum.message), "%s", msg);
  return BSGPU_OK;
}

}  // namespace

// ===================================================================================================
// C entry points
// ===================================================================================================
extern "C" {

int bsgpu_nidx(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? kTypes[t].nidx : -1; }
int bsgpu_nconst(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? kTypes[t].nconst : -1; }
int bsgpu_nres(int t) { return (t >= 0 && t < BSGPU_F_NUM_TYPES) ? kTypes[t].m : -1; }
int bsgpu_abi_version(void) { return BSGPU_ABI_VERSION; }

void bsgpu_options_default(bsgpu_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50; o->linear_solver_type = BSGPU_LINEAR_AUTO; o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5; o->max_solver_time_in_seconds = 1e9;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->pcg_max_iterations = 500; o->pcg_tolerance = 1e-10;
}
void bsgpu_options_vio(bsgpu_options* o) {  // beam_slam_launch/config/vio.yaml:7-17
  bsgpu_options_default(o);
  o->max_num_iterations = 10; o->max_solver_time_in_seconds = 0.05;
  o->gradient_tolerance = 1.5e-7; o->parameter_tolerance = 1.5e-7; o->function_tolerance = 1.5e-7;
}

const char* bsgpu_create_error(void) { return g_create_error.c_str(); }

bsgpu_ctx* bsgpu_create(int device) try {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                     "); libbsgpu has no CPU fallback";
    (void)hipGetLastError();
    return nullptr;
  }
  if (device < 0 || device >= n) { g_create_error = "device index out of range"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return nullptr; }
  bsgpu_ctx* c = new bsgpu_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
  return c;
} catch (...) { try { g_create_error = "out of host memory"; } catch (...) {} return nullptr; }
void bsgpu_destroy(bsgpu_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->free_device();
  c->release_pool();
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  if (c->h_radius) (void)hipHostFree(c->h_radius);
  if (c->h_pcg) (void)hipHostFree(c->h_pcg);
  for (hipEvent_t e : c->pcg_ev) if (e) (void)hipEventDestroy(e);
  if (c->ev_reduce) (void)hipEventDestroy(c->ev_reduce);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
const char* bsgpu_last_error(const bsgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bsgpu_clear(bsgpu_ctx* c) try {
  if (!c) return BSGPU_ERR_INVALID;
  c->nb = 0; c->h_x.clear(); c->off.clear(); c->size.clear(); c->manifold.clear(); c->is_const.clear(); c->is_const_in.clear();
  c->cams.clear();
  for (auto& g : c->groups) { g.n = 0; g.idx.clear(); g.consts.clear(); g.loss_kind.clear(); g.loss_a.clear(); }   // (capacity kept: a window is re-described every cycle)
  c->marginals.clear();
  c->no_elim.clear();
  c->finalized = false;
  c->iters.clear();
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_set_blocks(bsgpu_ctx* c, int32_t n, const double* values, const int32_t* offset, const uint8_t* size,
                     const uint8_t* manifold, const uint8_t* is_const) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (n <= 0 || !values || !offset || !size || !manifold || !is_const) return fail(c, BSGPU_ERR_INVALID, "set_blocks: null/empty argument");
  c->nb = n;
  c->off.assign(offset, offset + n); c->size.assign(size, size + n);
  c->manifold.assign(manifold, manifold + n); c->is_const.assign(is_const, is_const + n); c->is_const_in = c->is_const;
  size_t tot = 0;
  for (int i = 0; i < n; ++i) { if (offset[i] < 0) return fail(c, BSGPU_ERR_INVALID, "negative block offset"); tot = std::max(tot, (size_t)offset[i] + size[i]); }
  c->h_x.assign(values, values + tot);
  c->finalized = false;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_set_values(bsgpu_ctx* c, const double* v, int64_t n) try {
  if (!c) return BSGPU_ERR_INVALID;
  if ((size_t)n != c->h_x.size()) return fail(c, BSGPU_ERR_INVALID, "set_values: size mismatch");
  c->h_x.assign(v, v + n);
  if (c->finalized) {
    // derived cameras depend on constant extrinsic values: re-finalize if any online-calib factor exists
    if (c->groups[BSGPU_F_REPROJ_ONLINE_CALIB].n) { c->finalized = false; return BSGPU_OK; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(c->d_x, v, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_x0, v, sizeof(double) * n, hipMemcpyHostToDevice));
  }
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_set_cameras(bsgpu_ctx* c, int32_t n, const bsgpu_camera* cams) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (n < 0 || (n > 0 && !cams)) return fail(c, BSGPU_ERR_INVALID, "set_cameras: bad argument");
  c->cams.assign(cams, cams + n);
  c->finalized = false;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_add_factors(bsgpu_ctx* c, int32_t type, int32_t n, const int32_t* idx, const double* consts,
                      const int32_t* loss_kind, const double* loss_a) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) return fail(c, BSGPU_ERR_INVALID, "unknown factor type");
  if (n < 0 || (n > 0 && (!idx || !consts))) return fail(c, BSGPU_ERR_INVALID, "add_factors: bad argument");
  const TypeInfo& ti = kTypes[type];
  HostGroup& g = c->groups[type];
  g.idx.insert(g.idx.end(), idx, idx + (size_t)n * ti.nidx);
  g.consts.insert(g.consts.end(), consts, consts + (size_t)n * ti.nconst);
  for (int i = 0; i < n; ++i) {
    const int k = loss_kind ? loss_kind[i] : BSGPU_LOSS_TRIVIAL;
    if (k < 0 || k > BSGPU_LOSS_HUBER) return fail(c, BSGPU_ERR_INVALID, "unknown loss kind");
    g.loss_kind.push_back(k); g.loss_a.push_back(loss_a ? loss_a[i] : 1.0);
  }
  g.n += n;
  c->finalized = false;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_add_factors_indirect(bsgpu_ctx* c, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots, const int32_t* slot_to_block,
                               const double* consts, const int32_t* loss_kind, const double* loss_a) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) return fail(c, BSGPU_ERR_INVALID, "unknown factor type");
  if (n < 0 || n_slots < 0 || (n > 0 && (!slot_idx || !consts || !slot_to_block))) return fail(c, BSGPU_ERR_INVALID, "add_factors_indirect: bad argument");
  const TypeInfo& ti = kTypes[type];
  HostGroup& g = c->groups[type];
  for (int i = 0; i < n; ++i) {
    const int k = loss_kind ? loss_kind[i] : BSGPU_LOSS_TRIVIAL;
    if (k < 0 || k > BSGPU_LOSS_HUBER) return fail(c, BSGPU_ERR_INVALID, "unknown loss kind");
  }
  const size_t base = g.idx.size(), nidx = (size_t)ti.nidx, nvar = (size_t)ti.nvar;
  g.idx.resize(base + (size_t)n * nidx);
  int32_t* dst = g.idx.data() + base;
  bool bad = false;
  for (size_t f = 0; f < (size_t)n; ++f) {
    for (size_t k = 0; k < nvar; ++k) {
      const int32_t s = slot_idx[f * nidx + k];
      const int32_t b = ((uint32_t)s < (uint32_t)n_slots) ? slot_to_block[s] : -1;
      bad |= b < 0;
      dst[f * nidx + k] = b;
    }
    for (size_t k = nvar; k < nidx; ++k) dst[f * nidx + k] = slot_idx[f * nidx + k];
  }
  if (bad) { g.idx.resize(base); return fail(c, BSGPU_ERR_INVALID, "add_factors_indirect: slot out of range or not mapped to a block"); }
  g.consts.insert(g.consts.end(), consts, consts + (size_t)n * ti.nconst);
  if (loss_kind) g.loss_kind.insert(g.loss_kind.end(), loss_kind, loss_kind + n); else g.loss_kind.insert(g.loss_kind.end(), n, BSGPU_LOSS_TRIVIAL);
  if (loss_a) g.loss_a.insert(g.loss_a.end(), loss_a, loss_a + n); else g.loss_a.insert(g.loss_a.end(), n, 1.0);
  g.n += n;
  c->finalized = false;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_add_marginal(bsgpu_ctx* c, int32_t n_blocks, const int32_t* blocks, int32_t n_rows, const double* A, const double* b,
                       const double* xbar) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (n_blocks <= 0 || n_rows <= 0 || !blocks || !A || !b || !xbar) return fail(c, BSGPU_ERR_INVALID, "add_marginal: bad arguments");
  HostMarginal mg;
  mg.blocks.assign(blocks, blocks + n_blocks);
  int cols = 0, amb = 0;
  for (int i = 0; i < n_blocks; ++i) {
    const int bl = blocks[i];
    if (bl < 0 || bl >= c->nb) return fail(c, BSGPU_ERR_INVALID, "add_marginal: block out of range (set_blocks first)");
    cols += (c->manifold[bl] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : c->size[bl];
    amb += c->size[bl];
  }
  mg.rows = n_rows; mg.cols = cols;
  mg.A.assign(A, A + (size_t)n_rows * cols); mg.b.assign(b, b + n_rows); mg.xbar.assign(xbar, xbar + amb);
  c->marginals.push_back(std::move(mg));
  c->finalized = false;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_finalize(bsgpu_ctx* c) try { return c ? finalize(c) : BSGPU_ERR_INVALID; } catch (...) { return api_exception(c); }
int bsgpu_solve(bsgpu_ctx* c, const bsgpu_options* o, bsgpu_summary* s) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!o || !s) return fail(c, BSGPU_ERR_INVALID, "solve: null argument");
  return solve(c, *o, *s);
} catch (...) { return api_exception(c); }
int bsgpu_get_blocks(bsgpu_ctx* c, double* v, int64_t n) try {
  if (!c) return BSGPU_ERR_INVALID;
  if ((size_t)n != c->h_x.size()) return fail(c, BSGPU_ERR_INVALID, "get_blocks: size mismatch");
  if (!c->finalized) { std::memcpy(v, c->h_x.data(), sizeof(double) * n); return BSGPU_OK; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(v, c->d_x, sizeof(double) * n, hipMemcpyDeviceToHost));
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_reset_values(bsgpu_ctx* c) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!c->finalized) return BSGPU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->d_x, c->d_x0, sizeof(double) * c->h_x.size(), hipMemcpyDeviceToDevice, c->stream));
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_num_iterations_recorded(const bsgpu_ctx* c) { return c ? (int)c->iters.size() : 0; }
int bsgpu_get_iteration(const bsgpu_ctx* c, int32_t i, bsgpu_iteration* out) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (i < 0 || i >= (int)c->iters.size() || !out) return BSGPU_ERR_INVALID;
  *out = c->iters[i];
  return BSGPU_OK;
} catch (...) { return api_exception(const_cast<bsgpu_ctx*>(c)); }
int bsgpu_num_residuals(const bsgpu_ctx* c) { return c ? c->n_res : -1; }
int bsgpu_num_parameters_tangent(const bsgpu_ctx* c) { return c ? c->n_tan : -1; }
int bsgpu_tangent_offset(const bsgpu_ctx* c, int32_t b) { return (c && c->finalized && b >= 0 && b < c->nb) ? c->toff[b] : -1; }

int bsgpu_evaluate(bsgpu_ctx* c, double* cost, double* residuals, double* gradient, double* jacobian) try {
  if (!c) return BSGPU_ERR_INVALID;
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int n = c->n_tan, m = c->n_res;
  if (jacobian && (size_t)m * n > ((size_t)64 << 20)) return fail(c, BSGPU_ERR_UNSUPPORTED, "dense jacobian too large");
  hipStream_t s = c->stream;
  double fixed = 0.0;
  if (c->any_inactive) {
    launch_zero(s, c->d_scal + SC_FIXED_COST, 1);
    for (int t = 2; t < kNumInternal; ++t) {
      if (!c->small[t].n) continue;
      SmallGroup g = c->small[t];
      g.active = c->d_small_inactive[t];
      launch_small_eval(s, g, c->d_x, c->d_losses, false, c->d_small_part[t]);
      launch_sum(s, c->d_small_part[t], g.n, c->d_scal + SC_FIXED_COST, 1);
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
  eval_all(c, c->d_x, true, SC_COST_X);
  final_reduce(c);
  rc = fetch_scalars(c);
  if (rc != BSGPU_OK) return rc;
  if (c->any_inactive) fixed = c->h_scal[SC_FIXED_COST];
  if (cost) *cost = c->h_scal[SC_COST_X] + fixed;
  if (!residuals && !gradient && !jacobian) return BSGPU_OK;
  if (jacobian) std::fill(jacobian, jacobian + (size_t)m * n, 0.0);
  std::vector<double> grad(n, 0.0);
  // visual factors: un-permute to (type, insertion) order
  const Visual& V = c->vis;
  if ((rc = ensure_vis_src(c)) != BSGPU_OK) return rc;
  if (V.n) {
    std::vector<double> r((size_t)V.n * 2), J((size_t)V.n * 18);
    std::vector<int4> fac(V.n);
    std::vector<int> cam_pose(V.n), lm_of(V.n), cp_tq(V.n_cam_pose), cp_tp(V.n_cam_pose);
    HIPCHK(c, hipMemcpy(r.data(), V.r, sizeof(double) * 2 * V.n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(J.data(), V.J, sizeof(double) * 18 * V.n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cam_pose.data(), V.cam_pose, sizeof(int) * V.n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(lm_of.data(), V.lm_of, sizeof(int) * V.n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cp_tq.data(), V.cp_tq, sizeof(int) * V.n_cam_pose, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cp_tp.data(), V.cp_tp, sizeof(int) * V.n_cam_pose, hipMemcpyDeviceToHost));
    for (int i = 0; i < V.n; ++i) {
      const int t = c->vis_src[i] >> 28, f = c->vis_src[i] & ((1 << 28) - 1);
      const int row = c->row0[t] + 2 * f;
      const int cols[3] = {cp_tq[cam_pose[i]], cp_tp[cam_pose[i]], lm_of[i] >= 0 ? c->n_pose + 3 * lm_of[i] : -1};
      for (int k = 0; k < 2; ++k) {
        if (residuals) residuals[row + k] = r[2 * (size_t)i + k];
        for (int sl = 0; sl < 3; ++sl) {
          if (cols[sl] < 0) continue;
          for (int j = 0; j < 3; ++j) {
            const double v = J[(size_t)i * 18 + 9 * k + 3 * sl + j];
            grad[cols[sl] + j] += v * r[2 * (size_t)i + k];
            if (jacobian) jacobian[(size_t)(row + k) * n + cols[sl] + j] = v;
          }
        }
      }
    }
  }
  for (int t = 2; t < kNumInternal; ++t) {
    const SmallGroup& g = c->small[t];
    if (!g.n) continue;
    const int mm = g.m, tw = 3 * g.nv;
    std::vector<double> r((size_t)g.n * mm), J((size_t)g.n * mm * tw);
    std::vector<int> toff((size_t)g.n * g.nv);
    HIPCHK(c, hipMemcpy(r.data(), g.r, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(J.data(), g.J, sizeof(double) * J.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(toff.data(), g.toff, sizeof(int) * toff.size(), hipMemcpyDeviceToHost));
    for (int f = 0; f < g.n; ++f)
      for (int k = 0; k < mm; ++k) {
        const int row = (t == T_REPROJ_DENSE) ? c->row0[c->dense_src[f] >> 28] + 2 * (c->dense_src[f] & ((1 << 28) - 1)) + k
                                              : c->row0[t] + f * mm + k;
        if (residuals) residuals[row] = r[(size_t)f * mm + k];
        for (int sl = 0; sl < g.nv; ++sl) {
          const int tc = toff[(size_t)f * g.nv + sl];
          if (tc < 0) continue;
          for (int j = 0; j < (sl == g.nv - 1 ? g.w_last : 3); ++j) {
            const double v = J[((size_t)f * mm + k) * tw + 3 * sl + j];
            grad[tc + j] += v * r[(size_t)f * mm + k];
            if (jacobian) jacobian[(size_t)row * n + tc + j] = v;
          }
        }
      }
  }
  for (const auto& mc : c->marg) {
    const MargDev& d = mc.dev;
    std::vector<double> r(d.rows), J((size_t)d.rows * d.cols);
    std::vector<int> col_t(d.cols);
    if (!mc.active) {   // evaluated only in the fixed-cost pass: fill r / J now
      launch_marg_eval(s, d, c->d_x, true, mc.part);
      HIPCHK(c, hipStreamSynchronize(s));
    }
    HIPCHK(c, hipMemcpy(r.data(), d.r, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(J.data(), d.J, sizeof(double) * J.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(col_t.data(), d.col_t, sizeof(int) * col_t.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < d.rows; ++k) {
      if (residuals) residuals[mc.row0 + k] = r[k];
      for (int a = 0; a < d.cols; ++a) {
        if (col_t[a] < 0) continue;
        const double v = J[(size_t)k * d.cols + a];
        grad[col_t[a]] += v * r[k];
        if (jacobian) jacobian[(size_t)(mc.row0 + k) * n + col_t[a]] = v;
      }
    }
  }
  if (gradient) std::memcpy(gradient, grad.data(), sizeof(double) * n);
  return BSGPU_OK;
} catch (...) { return api_exception(c); }

// ---------------------------------------------------------------------------------------------------
// [EXT] fuse_constraints::marginalizeVariables on the device.  The factors that touch the marginalised blocks form a
// sub-problem (own context, same kernels): its landmark-only marginalised blocks leave through the landmark Schur
// complement, the undamped reduced system is gathered into [marginalised | kept] order and a single-workgroup
// positive-semi-definite Cholesky yields both the Schur complement onto the kept blocks and its factor.
// ---------------------------------------------------------------------------------------------------
int bsgpu_marginalize(bsgpu_ctx* c, int32_t n_marg, const int32_t* marg_blocks, int32_t* n_kept, int32_t* n_rows, int32_t* n_cols) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!marg_blocks || n_marg <= 0 || !n_kept || !n_rows || !n_cols) return fail(c, BSGPU_ERR_INVALID, "marginalize: bad arguments");
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  c->marg_result = bsgpu_ctx::MargResult();
  const int nb = c->nb;
  std::vector<uint8_t> is_marg(nb, 0), used(nb, 0);
  for (int i = 0; i < n_marg; ++i) {
    const int b = marg_blocks[i];
    if (b < 0 || b >= nb) return fail(c, BSGPU_ERR_INVALID, "marginalize: block out of range");
    if (c->is_const[b]) return fail(c, BSGPU_ERR_INVALID, "marginalize: constant block");
    is_marg[b] = 1;
  }
  // current values
  std::vector<double> xcur(c->h_x.size());
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(xcur.data(), c->d_x, sizeof(double) * xcur.size(), hipMemcpyDeviceToHost));
  // connected factors
  bsgpu_ctx* sub = bsgpu_create(c->device);
  if (!sub) return fail(c, BSGPU_ERR_DEVICE, "marginalize: cannot create the sub-problem context");
  struct Guard { bsgpu_ctx* p; ~Guard() { bsgpu_destroy(p); } } guard{sub};
  int n_connected = 0;
  std::vector<std::vector<int>> pick(BSGPU_F_NUM_TYPES);
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      bool touch = false;
      for (int sl = 0; sl < ti.nvar; ++sl) touch = touch || is_marg[idx[sl]];
      if (!touch) continue;
      pick[t].push_back(f);
      for (int sl = 0; sl < ti.nvar; ++sl) used[idx[sl]] = 1;
      ++n_connected;
    }
  }
  std::vector<int> pick_marg;
  for (size_t i = 0; i < c->marginals.size(); ++i) {
    bool touch = false;
    for (int b : c->marginals[i].blocks) touch = touch || is_marg[b];
    if (!touch) continue;
    pick_marg.push_back((int)i);
    for (int b : c->marginals[i].blocks) used[b] = 1;
    ++n_connected;
  }
  if (!n_connected) return fail(c, BSGPU_ERR_INVALID, "marginalize: no factor touches the blocks to marginalise");
  std::vector<uint8_t> sub_const(nb), no_elim(nb, 0);
  std::vector<int32_t> kept;
  for (int b = 0; b < nb; ++b) {
    sub_const[b] = (c->is_const[b] || !used[b]) ? 1 : 0;
    if (used[b] && !c->is_const[b] && !is_marg[b]) { kept.push_back(b); no_elim[b] = 1; }
  }
  auto subfail = [&](int code) { return fail(c, code, std::string("marginalize (sub-problem): ") + sub->err); };
  std::vector<int32_t> off(c->off.begin(), c->off.end());
  std::vector<uint8_t> size8(c->size.begin(), c->size.end()), man8(c->manifold.begin(), c->manifold.end());
  if ((rc = bsgpu_set_blocks(sub, nb, xcur.data(), off.data(), size8.data(), man8.data(), sub_const.data())) != BSGPU_OK) return subfail(rc);
  if (!c->cams.empty() && (rc = bsgpu_set_cameras(sub, (int32_t)c->cams.size(), c->cams.data())) != BSGPU_OK) return subfail(rc);
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    if (pick[t].empty()) continue;
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    std::vector<int32_t> idx, lk;
    std::vector<double> cs, la;
    for (int f : pick[t]) {
      idx.insert(idx.end(), &g.idx[(size_t)f * ti.nidx], &g.idx[(size_t)f * ti.nidx] + ti.nidx);
      cs.insert(cs.end(), &g.consts[(size_t)f * ti.nconst], &g.consts[(size_t)f * ti.nconst] + ti.nconst);
      lk.push_back(g.loss_kind[f]); la.push_back(g.loss_a[f]);
    }
    if ((rc = bsgpu_add_factors(sub, t, (int32_t)pick[t].size(), idx.data(), cs.data(), lk.data(), la.data())) != BSGPU_OK) return subfail(rc);
  }
  for (int i : pick_marg) {
    const HostMarginal& mg = c->marginals[i];
    if ((rc = bsgpu_add_marginal(sub, (int32_t)mg.blocks.size(), mg.blocks.data(), mg.rows, mg.A.data(), mg.b.data(), mg.xbar.data())) != BSGPU_OK) return subfail(rc);
  }
  sub->no_elim = no_elim;
  if ((rc = finalize(sub)) != BSGPU_OK) return subfail(rc);
  if (!sub->dense_ok) return fail(c, BSGPU_ERR_UNSUPPORTED, "marginalize: the connected sub-problem exceeds the dense limit");
  // order: marginalised pose-side dims first, kept dims after (both in block order)
  std::vector<int> spos;
  int m = 0;
  for (int b = 0; b < nb; ++b) if (is_marg[b] && !sub->is_lm[b] && sub->toff[b] >= 0) for (int k = 0; k < sub->tsize[b]; ++k, ++m) spos.push_back(sub->plan.spos(sub->toff[b] + k));
  int kdim = 0;
  for (int b : kept) for (int k = 0; k < sub->tsize[b]; ++k, ++kdim) spos.push_back(sub->plan.spos(sub->toff[b] + k));
  const int n = m + kdim;
  if (n != sub->n_pose) return fail(c, BSGPU_ERR_UNSUPPORTED, "internal: marginalisation order does not cover the reduced system");
  if (n > 8000) return fail(c, BSGPU_ERR_UNSUPPORTED, "marginalize: more than 8000 connected dimensions");
  if (kdim == 0) { *n_kept = 0; *n_rows = 0; *n_cols = 0; c->marg_result.valid = true; return BSGPU_OK; }
  // undamped normal equations of the sub-problem at the current values
  hipStream_t s = sub->stream;
  bsgpu_options o;
  bsgpu_options_default(&o);
  *sub->h_radius = 1e300;
  (void)hipMemcpyAsync(sub->d_scal + SC_RADIUS, sub->h_radius, sizeof(double), hipMemcpyHostToDevice, s);
  eval_all(sub, sub->d_x, true, SC_COST_X);
  sub->use_pcg = false;
  assemble(sub, o, 1e300, true, true);
  int* d_spos = sub->upload(spos);
  double* d_M = sub->alloc<double>((size_t)n * n);
  double* d_g = sub->alloc<double>(n);
  double* d_diag0 = sub->alloc<double>(n);
  int* d_ok = sub->alloc<int>(n);
  double* d_A = sub->alloc<double>((size_t)kdim * kdim);
  double* d_b = sub->alloc<double>(kdim);
  double* d_status = sub->alloc<double>(2);
  if (!d_M || !d_A || !d_status) return fail(c, BSGPU_ERR_DEVICE, "marginalize: out of device memory");
  launch_marg_schur(s, sub->d_S, sub->npad, sub->plan.rhs_row, d_spos, n, m, 1e-11, d_M, d_g, d_diag0, d_ok, d_status, d_A, d_b);
  double status[2] = {0, 0};
  (void)hipMemcpyAsync(status, d_status, sizeof(status), hipMemcpyDeviceToHost, s);
  if ((rc = fetch_scalars(sub)) != BSGPU_OK) return subfail(rc);
  if (sub->h_scal[SC_CHOL_FAIL] > 0.0 || status[1] > 0.0)
    return fail(c, BSGPU_ERR_NUMERIC, "marginalize: the blocks to marginalise are not fully constrained by the factors that touch them");
  bsgpu_ctx::MargResult& R = c->marg_result;
  R.kept = kept; R.rows = (int)status[0]; R.cols = kdim;
  R.A.resize((size_t)R.rows * kdim); R.b.resize(R.rows);
  if (R.rows) {
    HIPCHK(c, hipMemcpy(R.A.data(), d_A, sizeof(double) * R.A.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(R.b.data(), d_b, sizeof(double) * R.rows, hipMemcpyDeviceToHost));
  }
  for (int b : kept) R.xbar.insert(R.xbar.end(), &xcur[c->off[b]], &xcur[c->off[b]] + c->size[b]);
  R.valid = true;
  *n_kept = (int32_t)kept.size(); *n_rows = R.rows; *n_cols = kdim;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_get_marginal(const bsgpu_ctx* c, int32_t* kept_blocks, double* A, double* b, double* xbar) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!c->marg_result.valid) return BSGPU_ERR_INVALID;
  const auto& R = c->marg_result;
  if (kept_blocks) std::memcpy(kept_blocks, R.kept.data(), sizeof(int32_t) * R.kept.size());
  if (A) std::memcpy(A, R.A.data(), sizeof(double) * R.A.size());
  if (b) std::memcpy(b, R.b.data(), sizeof(double) * R.b.size());
  if (xbar) std::memcpy(xbar, R.xbar.data(), sizeof(double) * R.xbar.size());
  return BSGPU_OK;
} catch (...) { return api_exception(const_cast<bsgpu_ctx*>(c)); }

// Graph::getCovariance for pose-side blocks: Sigma_pp = (H_pp - H_pl H_ll^-1 H_lp)^-1 = S^-1 at the current values,
// without LM damping (what ceres::Covariance computes from the robustified J^T J, landmarks marginalised).
// One undamped assembly + one factorisation; the unit vectors of both blocks ride along as rows of the rhs tile.
int bsgpu_covariance(bsgpu_ctx* c, int32_t ba, int32_t bb, double* out) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!out) return fail(c, BSGPU_ERR_INVALID, "null argument");
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  if (ba < 0 || bb < 0 || ba >= c->nb || bb >= c->nb) return fail(c, BSGPU_ERR_INVALID, "covariance: block out of range");
  if (c->toff[ba] < 0 || c->toff[bb] < 0) return fail(c, BSGPU_ERR_INVALID, "covariance: constant block");
  if (c->is_lm[ba] || c->is_lm[bb])
    return fail(c, BSGPU_ERR_UNSUPPORTED, "covariance: landmark blocks are eliminated; only pose-side blocks can be queried");
  if (!c->dense_ok) return fail(c, BSGPU_ERR_UNSUPPORTED, "covariance: reduced system above the dense limit (block-sparse PCG path has no factor)");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  bsgpu_options o;
  bsgpu_options_default(&o);
  const bool was_pcg = c->use_pcg;
  c->use_pcg = false;
  *c->h_radius = 1e300;   // Lambda / radius -> 0: undamped normal equations
  (void)hipMemcpyAsync(c->d_scal + SC_RADIUS, c->h_radius, sizeof(double), hipMemcpyHostToDevice, s);
  eval_all(c, c->d_x, true, SC_COST_X);
  assemble(c, o, 1e300, true, true);
  c->use_pcg = was_pcg;
  const int ta = c->tsize[ba], tb = c->tsize[bb];
  std::vector<int> cols;
  for (int i = 0; i < ta; ++i) cols.push_back(c->plan.spos(c->toff[ba] + i));
  const int row_b0 = (ba == bb) ? 0 : ta;
  if (ba != bb) for (int j = 0; j < tb; ++j) cols.push_back(c->plan.spos(c->toff[bb] + j));
  int* d_cols = nullptr;
  double* d_out = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_cols, sizeof(int) * cols.size()));
  if (hipMalloc((void**)&d_out, sizeof(double) * ta * tb) != hipSuccess) { (void)hipFree(d_cols); return fail(c, BSGPU_ERR_DEVICE, "out of device memory"); }
  (void)hipMemcpyAsync(d_cols, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice, s);
  launch_cov_units(s, c->d_S, c->npad, c->plan.rhs_row, d_cols, (int)cols.size());
  const DenseDev D{c->d_perm, c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
                   c->d_panel_of_tile, c->d_chain_begin, c->d_chain_end, c->d_tile_sync};
  dense_factor(s, c->plan, D, c->d_S, c->d_scal);
  launch_cov_dots(s, c->d_Lp, c->npad, c->plan.rhs_row, c->plan.T * 64, ta, row_b0, tb, d_out);
  (void)hipMemcpyAsync(out, d_out, sizeof(double) * ta * tb, hipMemcpyDeviceToHost, s);
  rc = fetch_scalars(c);
  (void)hipFree(d_cols); (void)hipFree(d_out);
  if (rc != BSGPU_OK) return rc;
  if (c->h_scal[SC_CHOL_FAIL] > 0.0 || !std::isfinite(out[0]))
    return fail(c, BSGPU_ERR_NUMERIC, "covariance: J^T J is singular at the current values (gauge freedom or unobserved block)");
  return BSGPU_OK;
} catch (...) { return api_exception(c); }

int bsgpu_reprojection_errors(bsgpu_ctx* c, double* err) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!err) return fail(c, BSGPU_ERR_INVALID, "null argument");
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const Visual& V = c->vis;
  if ((rc = ensure_vis_src(c)) != BSGPU_OK) return rc;
  const SmallGroup& D = c->small[T_REPROJ_DENSE];
  const int n0 = c->groups[BSGPU_F_REPROJ].n;
  if (V.n + D.n == 0) return BSGPU_OK;
  double* d_out = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_out, sizeof(double) * (size_t)(V.n + D.n)));
  launch_reproj_errors(c->stream, V, D, c->d_x, c->d_cams, d_out, d_out + V.n);
  std::vector<double> h((size_t)V.n + D.n);
  const hipError_t e = hipMemcpyAsync(h.data(), d_out, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream);
  const hipError_t e2 = hipStreamSynchronize(c->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess || e2 != hipSuccess) return fail(c, BSGPU_ERR_DEVICE, "reprojection_errors: device error");
  auto slot = [&](int src) { const int t = src >> 28, f = src & ((1 << 28) - 1); return (t == BSGPU_F_REPROJ ? 0 : n0) + f; };
  for (int i = 0; i < V.n; ++i) err[slot(c->vis_src[i])] = h[i];
  for (int i = 0; i < D.n; ++i) err[slot(c->dense_src[i])] = h[(size_t)V.n + i];
  return BSGPU_OK;
} catch (...) { return api_exception(c); }

int bsgpu_preintegrate(int device, int32_t n, const int32_t* sample_start, const double* t, const double* w, const double* a,
                       const double* t_end, const double* bg, const double* ba, const double* cov_w, const double* cov_a,
                       const double* cov_bg, const double* cov_ba, double info_weight, double* consts_out) try {
  if (n <= 0 || !sample_start || !t || !w || !a || !t_end || !bg || !ba || !cov_w || !cov_a || !cov_bg || !cov_ba || !consts_out) return BSGPU_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return BSGPU_ERR_DEVICE;
  const int ns = sample_start[n];
  if (ns <= 0) return BSGPU_ERR_INVALID;
  std::vector<double> covs(36);
  std::memcpy(&covs[0], cov_w, 72); std::memcpy(&covs[9], cov_a, 72); std::memcpy(&covs[18], cov_bg, 72); std::memcpy(&covs[27], cov_ba, 72);
  std::vector<void*> bufs;
  auto up = [&](const void* src, size_t bytes) -> void* {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    bufs.push_back(d);
    if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  };
  int* d_ss = (int*)up(sample_start, sizeof(int) * (n + 1));
  double* d_t = (double*)up(t, sizeof(double) * ns);
  double* d_w = (double*)up(w, sizeof(double) * 3 * ns);
  double* d_a = (double*)up(a, sizeof(double) * 3 * ns);
  double* d_te = (double*)up(t_end, sizeof(double) * n);
  double* d_bg = (double*)up(bg, sizeof(double) * 3 * n);
  double* d_ba = (double*)up(ba, sizeof(double) * 3 * n);
  double* d_cov = (double*)up(covs.data(), sizeof(double) * 36);
  double* d_out = (double*)up(nullptr, sizeof(double) * 287 * (size_t)n);
  int rc = BSGPU_OK;
  if (!d_ss || !d_t || !d_w || !d_a || !d_te || !d_bg || !d_ba || !d_cov || !d_out) rc = BSGPU_ERR_DEVICE;
  if (rc == BSGPU_OK) {
    launch_preintegrate(nullptr, n, d_ss, d_t, d_w, d_a, d_te, d_bg, d_ba, d_cov, info_weight, d_out);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess ||
        hipMemcpy(consts_out, d_out, sizeof(double) * 287 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = BSGPU_ERR_DEVICE;
  }
  for (void* p : bufs) (void)hipFree(p);
  return rc;
} catch (...) { return api_exception(nullptr); }

int bsgpu_triangulate(bsgpu_ctx* c, int32_t n_tracks, const int32_t* track_start, const int32_t* q_block, const int32_t* p_block,
                      const double* pixels, int32_t camera, int32_t truncate_pixels, double max_dist, double max_reproj, double* points,
                      int32_t* status) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (n_tracks < 0 || !track_start || !points || !status) return fail(c, BSGPU_ERR_INVALID, "null argument");
  if (n_tracks == 0) return BSGPU_OK;
  const int n_obs = track_start[n_tracks];
  if (track_start[0] != 0 || n_obs < 0 || (n_obs > 0 && (!q_block || !p_block || !pixels)))
    return fail(c, BSGPU_ERR_INVALID, "triangulate: malformed track table");
  if (camera < 0 || camera >= (int)c->cams.size()) return fail(c, BSGPU_ERR_INVALID, "camera index out of range");
  std::vector<int32_t> pose_off((size_t)2 * n_obs);
  for (int i = 0; i < n_tracks; ++i)
    if (track_start[i + 1] < track_start[i]) return fail(c, BSGPU_ERR_INVALID, "triangulate: track_start must be non-decreasing");
  for (int o = 0; o < n_obs; ++o) {
    const int qb = q_block[o], pb = p_block[o];
    if (qb < 0 || qb >= c->nb || pb < 0 || pb >= c->nb) return fail(c, BSGPU_ERR_INVALID, "triangulate: block out of range");
    if (c->size[qb] != 4 || c->size[pb] != 3) return fail(c, BSGPU_ERR_INVALID, "triangulate: view blocks must be (orientation[4], position[3])");
    pose_off[2 * (size_t)o] = c->off[qb];
    pose_off[2 * (size_t)o + 1] = c->off[pb];
  }
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const bsgpu_camera& hc = c->cams[camera];
  DevCamera cam;
  cam.fx = hc.fx; cam.fy = hc.fy; cam.cx = hc.cx; cam.cy = hc.cy;
  std::memcpy(cam.R, hc.R_cam_baselink, sizeof(cam.R));
  std::memcpy(cam.t, hc.t_cam_baselink, sizeof(cam.t));
  int *d_start = nullptr, *d_status = nullptr;
  int2* d_off = nullptr;
  double2* d_pix = nullptr;
  double* d_pts = nullptr;
  auto release = [&]() { (void)hipFree(d_start); (void)hipFree(d_status); (void)hipFree(d_off); (void)hipFree(d_pix); (void)hipFree(d_pts); };
  hipError_t e = hipMalloc((void**)&d_start, sizeof(int) * ((size_t)n_tracks + 1));
  if (e == hipSuccess) e = hipMalloc((void**)&d_status, sizeof(int) * (size_t)n_tracks);
  if (e == hipSuccess) e = hipMalloc((void**)&d_pts, sizeof(double) * 3 * (size_t)n_tracks);
  if (e == hipSuccess && n_obs) e = hipMalloc((void**)&d_off, sizeof(int2) * (size_t)n_obs);
  if (e == hipSuccess && n_obs) e = hipMalloc((void**)&d_pix, sizeof(double2) * (size_t)n_obs);
  if (e == hipSuccess) e = hipMemcpyAsync(d_start, track_start, sizeof(int) * ((size_t)n_tracks + 1), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_obs) e = hipMemcpyAsync(d_off, pose_off.data(), sizeof(int2) * (size_t)n_obs, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_obs) e = hipMemcpyAsync(d_pix, pixels, sizeof(double2) * (size_t)n_obs, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    launch_triangulate(c->stream, n_tracks, d_start, d_off, d_pix, c->d_x, cam, truncate_pixels != 0, max_dist, max_reproj, d_pts, d_status);
    e = hipMemcpyAsync(points, d_pts, sizeof(double) * 3 * (size_t)n_tracks, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(status, d_status, sizeof(int) * (size_t)n_tracks, hipMemcpyDeviceToHost, c->stream);
  const hipError_t e2 = hipStreamSynchronize(c->stream);
  release();
  if (e != hipSuccess || e2 != hipSuccess) return fail(c, BSGPU_ERR_DEVICE, "triangulate: device error");
  return BSGPU_OK;
} catch (...) { return api_exception(c); }

double bsgpu_time_reproj_jacobian_ms(bsgpu_ctx* c, int32_t reps) {
  if (!c) return -1.0;
  if (finalize(c) != BSGPU_OK || c->vis.n == 0 || reps <= 0) return -1.0;
  if (hipSetDevice(c->device) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  launch_reproj_jacobian_only(c->stream, c->vis, c->d_x, c->d_cams, c->d_losses);  // warm
  (void)hipEventRecord(e0, c->stream);
  for (int i = 0; i < reps; ++i) launch_reproj_jacobian_only(c->stream, c->vis, c->d_x, c->d_cams, c->d_losses);
  (void)hipEventRecord(e1, c->stream);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms / reps;
}
int64_t bsgpu_reproj_jacobian_bytes(const bsgpu_ctx* c) {
  if (!c) return -1;
  // per factor: 16 B (3 offsets + meta) + 16 B pixel + 8 B weight in, 16 B residual + 144 B Jacobian out;
  // plus every parameter block once
  return (int64_t)c->vis.n * 200 + (int64_t)c->h_x.size() * 8;
}

// Stand-alone dense SPD solve A x = b through the same plan + kernels the reduced camera system uses
// (test + measurement hook for the MFMA path).  Host pointers in and out.  The tile structure (and with
// it the nested-dissection ordering) is derived from the non-zeros of A; max_chains <= 1 forces the
// natural order.
int bsgpu_dense_solve(int device, int32_t n, const double* A, const double* b, double* x, int32_t max_chains, double* ms_out) try {
  if (n <= 0 || !A || !b || !x) return BSGPU_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return BSGPU_ERR_DEVICE;
  chol_prepare();
  const int T = (n + 63) / 64;
  std::vector<uint8_t> adj((size_t)T * T, 0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (A[(size_t)i * n + j] != 0.0) adj[(size_t)(i / 64) * T + j / 64] = 1;
  DensePlan P;
  {
    const char* e2 = getenv("BSGPU_MIN_PIECE");
    const char* e3 = getenv("BSGPU_SHARED");
    P.build(n, adj, std::max(1, (int)max_chains), e2 ? std::max(1, atoi(e2)) : 1, !(e3 && atoi(e3) == 0));
  }
  const int npad = P.npad;
  std::vector<double> hS((size_t)npad * npad, 0.0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) hS[(size_t)P.spos(i) * npad + P.spos(j)] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) hS[(size_t)P.rhs_row * npad + P.spos(j)] = b[j];
  std::vector<uint8_t> real(npad, 0);
  for (int j = 0; j < n; ++j) real[P.spos(j)] = 1;
  for (int i = 0; i < npad; ++i) if (!real[i]) hS[(size_t)i * npad + i] = 1.0;
  double *dS = nullptr, *dLp = nullptr, *dV = nullptr, *dy = nullptr, *dscal = nullptr;
  int *dperm = nullptr, *dnreal = nullptr, *drows = nullptr;
  PanelDesc *dpan = nullptr, *dsep = nullptr;
  int *dpot2 = nullptr, *dcb = nullptr, *dce = nullptr, *dsync = nullptr;
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return BSGPU_ERR_DEVICE;
  auto up = [](const void* src, size_t bytes, void** dst) {
    if (hipMalloc(dst, bytes ? bytes : 8) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = hipMalloc(&dS, sizeof(double) * hS.size()) == hipSuccess && hipMalloc(&dLp, sizeof(double) * hS.size()) == hipSuccess &&
            hipMalloc(&dV, sizeof(double) * chol_vinv_stride() * std::max(1, T)) == hipSuccess && hipMalloc(&dy, sizeof(double) * npad) == hipSuccess &&
            hipMalloc(&dscal, sizeof(double) * SC_NUM) == hipSuccess;
  std::vector<int> rf = P.rows_flat; if (rf.empty()) rf.push_back(0);
  ok = ok && up(P.perm.data(), sizeof(int) * P.perm.size(), (void**)&dperm) && up(P.nreal.data(), sizeof(int) * P.nreal.size(), (void**)&dnreal) &&
       up(rf.data(), sizeof(int) * rf.size(), (void**)&drows) &&
       up(P.panels.data(), sizeof(PanelDesc) * P.panels.size(), (void**)&dpan) &&
       up(P.panel_of_tile.data(), sizeof(int) * P.panel_of_tile.size(), (void**)&dpot2) &&
       up(P.chain_begin.data(), sizeof(int) * P.chain_begin.size(), (void**)&dcb) && up(P.chain_end.data(), sizeof(int) * P.chain_end.size(), (void**)&dce) &&
       up(P.tile_sync.data(), sizeof(int) * P.tile_sync.size(), (void**)&dsync);
  int rc = BSGPU_OK;
  if (ok) {
    (void)hipMemcpy(dS, hS.data(), sizeof(double) * hS.size(), hipMemcpyHostToDevice);
    (void)hipMemset(dscal, 0, sizeof(double) * SC_NUM);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    const DenseDev D{dperm, dnreal, drows, dpan, dLp, dV, dpot2, dcb, dce, dsync};
    dense_factor_solve(s, P, D, dS, dy, dscal);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) rc = BSGPU_ERR_DEVICE;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms;
    double hscal[SC_NUM];
    std::vector<double> hy(npad);
    (void)hipMemcpy(hscal, dscal, sizeof(hscal), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hy.data(), dy, sizeof(double) * npad, hipMemcpyDeviceToHost);
    for (int j = 0; j < n; ++j) x[j] = hy[P.spos(j)];
    if (rc == BSGPU_OK && hscal[SC_CHOL_FAIL] > 0.0) rc = BSGPU_ERR_NUMERIC;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  } else rc = BSGPU_ERR_DEVICE;
  (void)hipFree(dS); (void)hipFree(dLp); (void)hipFree(dV); (void)hipFree(dy); (void)hipFree(dscal);
  (void)hipFree(dperm); (void)hipFree(dnreal); (void)hipFree(drows); (void)hipFree(dpan);
  (void)hipFree(dsep); (void)hipFree(dpot2); (void)hipFree(dcb); (void)hipFree(dce); (void)hipFree(dsync);
  (void)hipStreamDestroy(s);
  return rc;
} catch (...) { return api_exception(nullptr); }

// number of independent sub-chains / schedule steps of the current problem's Cholesky plan (diagnostics)
int bsgpu_plan_info(const bsgpu_ctx* c, int32_t* n_chains, int32_t* n_steps, int32_t* n_tiles) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!c->finalized) return BSGPU_ERR_INVALID;
  if (n_chains) *n_chains = c->plan.n_pieces;
  if (n_steps) *n_steps = c->plan.n_steps();
  if (n_tiles) *n_tiles = c->plan.T;
  return BSGPU_OK;
} catch (...) { return api_exception(const_cast<bsgpu_ctx*>(c)); }

}  // extern "C"
