// Host driver of libbsgpu.so: the C-ABI of include/bsgpu.h on top of the HIP kernels.
//
// Replaces, for the reference, everything under `graph_->optimize(options)`
// (bs_optimizers/src/fixed_lag_smoother.cpp:281): [EXT] fuse HashGraph::createProblem (finalize(), bsgpu_finalize.cpp:
// flattening to device tables) and [EXT] ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT /
// SPARSE_NORMAL_CHOLESKY (solve(), bsgpu_solve.cpp: a restatement of Ceres' TrustRegionMinimizer +
// LevenbergMarquardtStrategy driving device kernels).  This file: the entry points themselves.
// There is no CPU fallback: without a HIP device bsgpu_create() fails.
#include <thread>

#include "bsgpu_ctx.h"

using namespace bsg;

namespace bsg {
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
thread_local std::string g_create_error;   // why this thread's last bsgpu_create() returned NULL (bsgpu_create_error); contexts may be
                                           // created from several host threads at once (and by every bsgpu_marginalize)
// The description of a finalized problem is about to change (factors / marginals / cameras added): the next finalize() re-uploads
// the host copy of the values, so the point the device has reached (a solve's result) is brought back first — a solve, add_factors,
// solve / get_blocks sequence continues from the optimised point ("the best accepted point is the context's current value set").
int invalidate_keep_values(bsgpu_ctx* c) {
  if (c->finalized && c->d_x && !c->h_x.empty()) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->h_x.data(), c->d_x, sizeof(double) * c->h_x.size(), hipMemcpyDeviceToHost));
  }
  c->finalized = false;
  return BSGPU_OK;
}
}  // namespace bsg

// ===================================================================================================
// C entry points
// ===================================================================================================
// ---- bsgpu_sync_factors_indirect -------------------------------------------------------------------
namespace bsg {
// block-named host copy of the mirrored group, for the paths that walk the rows on the host
int materialize_mirror(bsgpu_ctx* c) {
  SlotMirror& m = c->mirror0;
  if (!m.active || m.materialized) return BSGPU_OK;
  HostGroup& g = c->groups[BSGPU_F_REPROJ];
  g.idx.resize((size_t)m.n * 4);
  for (size_t f = 0; f < (size_t)m.n; ++f) {
    for (int k = 0; k < 3; ++k) {
      const int32_t s = m.idx[4 * f + k];
      const int32_t b = ((uint32_t)s < (uint32_t)m.n_slots) ? m.s2b[s] : -1;
      if (b < 0) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: slot out of range or not mapped to a block");
      g.idx[4 * f + k] = b;
    }
    g.idx[4 * f + 3] = m.idx[4 * f + 3];
  }
  g.consts = m.consts; g.loss_kind = m.loss_kind; g.loss_a = m.loss_a;
  g.n = m.n;
  m.materialized = true;
  return BSGPU_OK;
}
namespace {
inline void mirror_count(SlotMirror& m, size_t r, int d) {
  const int32_t* row = &m.idx[4 * r];
  auto bump = [&](std::vector<int32_t>& v, int32_t s) { if ((size_t)s >= v.size()) v.resize((size_t)s + 1 + v.size() / 4, 0); v[s] += d; };
  bump(m.use_q, row[0]); bump(m.use_p, row[1]); bump(m.use_l, row[2]); bump(m.cam_use, row[3]);
  int kind = m.loss_kind[r];
  double a = kind == BSGPU_LOSS_TRIVIAL ? 1.0 : m.loss_a[r];
  for (auto& u : m.loss_use) if (u.kind == kind && u.a == a) { u.rows += d; return; }
  m.loss_use.push_back({kind, a, d});
}
}  // namespace
}  // namespace bsg

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
  o->pcg_max_iterations = 500; o->pcg_tolerance = 1e-10;  // (include/bsgpu.h: the reference-equivalent step; 1e-6 is an explicit choice of the caller)
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
  batch_forget(c);   // (argument tables of bsgpu_solve_batch that name this context hold its device pointers)
  (void)hipSetDevice(c->device);
  c->free_device();
  c->release_pool();
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  c->mirror0.release_device();
  if (c->h_arena) (void)hipHostFree(c->h_arena);
  if (c->h_radius) (void)hipHostFree(c->h_radius);
  if (c->h_pcg) (void)hipHostFree(c->h_pcg);
  if (c->h_pcg_lazy) (void)hipHostFree(c->h_pcg_lazy);
  for (hipEvent_t e : c->pcg_ev) if (e) (void)hipEventDestroy(e);
  if (c->ev_reduce) (void)hipEventDestroy(c->ev_reduce);
  if (c->ev_solve0) (void)hipEventDestroy(c->ev_solve0);
  if (c->ev_solve1) (void)hipEventDestroy(c->ev_solve1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
const char* bsgpu_last_error(const bsgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bsgpu_clear(bsgpu_ctx* c) try {
  if (!c) return BSGPU_ERR_INVALID;
  c->nb = 0; c->h_x.clear(); c->off.clear(); c->size.clear(); c->manifold.clear(); c->is_const.clear(); c->is_const_in.clear();
  c->cams.clear();
  for (auto& g : c->groups) { g.n = 0; g.idx.clear(); g.consts.clear(); g.loss_kind.clear(); g.loss_a.clear(); }   // (capacity kept: a window is re-described every cycle)
  c->mirror0.active = false; c->mirror0.materialized = false;   // (the mirror itself stays: the next sync call patches it)
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
  return invalidate_keep_values(c);
} catch (...) { return api_exception(c); }
int bsgpu_add_factors(bsgpu_ctx* c, int32_t type, int32_t n, const int32_t* idx, const double* consts,
                      const int32_t* loss_kind, const double* loss_a) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) return fail(c, BSGPU_ERR_INVALID, "unknown factor type");
  if (n < 0 || (n > 0 && (!idx || !consts))) return fail(c, BSGPU_ERR_INVALID, "add_factors: bad argument");
  const TypeInfo& ti = kTypes[type];
  if (type == BSGPU_F_REPROJ) { const int rc_m = materialize_mirror(c); if (rc_m != BSGPU_OK) return rc_m; }   // (rows appended to a synced table: the host copy first)
  HostGroup& g = c->groups[type];
  g.idx.insert(g.idx.end(), idx, idx + (size_t)n * ti.nidx);
  g.consts.insert(g.consts.end(), consts, consts + (size_t)n * ti.nconst);
  for (int i = 0; i < n; ++i) {
    const int k = loss_kind ? loss_kind[i] : BSGPU_LOSS_TRIVIAL;
    if (k < 0 || k > BSGPU_LOSS_HUBER) return fail(c, BSGPU_ERR_INVALID, "unknown loss kind");
    g.loss_kind.push_back(k); g.loss_a.push_back(loss_a ? loss_a[i] : 1.0);
  }
  g.n += n;
  return invalidate_keep_values(c);
} catch (...) { return api_exception(c); }
int bsgpu_add_factors_indirect(bsgpu_ctx* c, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots, const int32_t* slot_to_block,
                               const double* consts, const int32_t* loss_kind, const double* loss_a) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) return fail(c, BSGPU_ERR_INVALID, "unknown factor type");
  if (n < 0 || n_slots < 0 || (n > 0 && (!slot_idx || !consts || !slot_to_block))) return fail(c, BSGPU_ERR_INVALID, "add_factors_indirect: bad argument");
  const TypeInfo& ti = kTypes[type];
  if (type == BSGPU_F_REPROJ) { const int rc_m = materialize_mirror(c); if (rc_m != BSGPU_OK) return rc_m; }   // (rows appended to a synced table: the host copy first)
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
  return invalidate_keep_values(c);
} catch (...) { return api_exception(c); }

int bsgpu_sync_factors_indirect(bsgpu_ctx* c, int32_t type, int32_t n, const int32_t* slot_idx, int32_t n_slots, const int32_t* slot_to_block,
                                const double* consts, const int32_t* loss_kind, const double* loss_a, int32_t n_changed,
                                const int32_t* changed_rows) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (type < 0 || type >= BSGPU_F_NUM_TYPES) return fail(c, BSGPU_ERR_INVALID, "unknown factor type");
  if (n < 0 || n_slots < 0 || (n > 0 && (!slot_idx || !consts || !slot_to_block)) || (n_changed > 0 && !changed_rows))
    return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: bad argument");
  if (c->groups[type].n != 0) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: the type already has factors in this description (it takes the whole table)");
  // only the reprojection table is mirrored; the other types are a few thousand rows: copied in as before
  if (type != BSGPU_F_REPROJ) return bsgpu_add_factors_indirect(c, type, n, slot_idx, n_slots, slot_to_block, consts, loss_kind, loss_a);
  SlotMirror& m = c->mirror0;
  // the previous call's asynchronous copies read the mirror's host vectors (pageable memory), which are rewritten — and may be
  // re-allocated — below: they must have left (the stream is idle between two cycles, this costs a few microseconds)
  if (m.d_idx) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
  const bool force_full = getenv("BSGPU_SYNC_FULL") != nullptr, check = getenv("BSGPU_SYNC_CHECK") != nullptr;   // (debug switches, read per call: a test sets them mid-process)
  const bool full = n_changed < 0 || !m.valid || force_full;
  auto row_ok = [&](size_t r) {
    const int32_t* row = slot_idx + 4 * r;
    if (row[0] < 0 || row[1] < 0 || row[2] < 0 || row[3] < 0) return false;
    const int k = loss_kind ? loss_kind[r] : BSGPU_LOSS_TRIVIAL;
    return k >= 0 && k <= BSGPU_LOSS_HUBER;
  };
  auto copy_row = [&](size_t r) {
    std::memcpy(&m.idx[4 * r], slot_idx + 4 * r, sizeof(int32_t) * 4);
    std::memcpy(&m.consts[3 * r], consts + 3 * r, sizeof(double) * 3);
    m.loss_kind[r] = loss_kind ? loss_kind[r] : BSGPU_LOSS_TRIVIAL;
    m.loss_a[r] = loss_a ? loss_a[r] : 1.0;
  };
  m.valid = false;   // (until the patch is complete: an error below leaves no half-patched mirror behind)
  m.dev_valid = m.dev_valid && !full;
  if (full) {
    for (size_t r = 0; r < (size_t)n; ++r) if (!row_ok(r)) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: negative slot / camera id or unknown loss kind");
    m.idx.assign(slot_idx, slot_idx + (size_t)n * 4);
    m.consts.assign(consts, consts + (size_t)n * 3);
    if (loss_kind) m.loss_kind.assign(loss_kind, loss_kind + n); else m.loss_kind.assign(n, BSGPU_LOSS_TRIVIAL);
    if (loss_a) m.loss_a.assign(loss_a, loss_a + n); else m.loss_a.assign(n, 1.0);
    std::fill(m.use_q.begin(), m.use_q.end(), 0); std::fill(m.use_p.begin(), m.use_p.end(), 0); std::fill(m.use_l.begin(), m.use_l.end(), 0);
    std::fill(m.cam_use.begin(), m.cam_use.end(), 0);
    m.loss_use.clear();
    m.n = n;
    for (size_t r = 0; r < (size_t)n; ++r) mirror_count(m, r, +1);
  } else {
    const int old_n = m.n;
    for (int i = 0; i < n_changed; ++i) {
      const int32_t r = changed_rows[i];
      if (r < 0 || r >= n) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: changed row out of range");
      if (!row_ok((size_t)r)) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: negative slot / camera id or unknown loss kind");
    }
    for (int r = n; r < old_n; ++r) mirror_count(m, (size_t)r, -1);           // rows that left at the end of the table
    m.idx.resize((size_t)n * 4, -1); m.consts.resize((size_t)n * 3); m.loss_kind.resize(n); m.loss_a.resize(n);
    m.n = n;
    size_t fresh = 0;                                                          // every row beyond the old table must be listed
    for (int i = 0; i < n_changed; ++i) {
      const size_t r = (size_t)changed_rows[i];
      if (m.idx[4 * r] >= 0) mirror_count(m, r, -1); else ++fresh;
      copy_row(r);
      mirror_count(m, r, +1);
    }
    if (n > old_n && fresh != (size_t)(n - old_n)) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: a row appended since the last call is not in the changed list");
  }
  if (check) {
    bool same = std::memcmp(m.idx.data(), slot_idx, sizeof(int32_t) * 4 * (size_t)n) == 0 && std::memcmp(m.consts.data(), consts, sizeof(double) * 3 * (size_t)n) == 0;
    for (size_t r = 0; r < (size_t)n && same; ++r)
      same = m.loss_kind[r] == (loss_kind ? loss_kind[r] : BSGPU_LOSS_TRIVIAL) && m.loss_a[r] == (loss_a ? loss_a[r] : 1.0);
    if (!same) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: the changed list does not account for every difference to the previous table (BSGPU_SYNC_CHECK)");
  }
  m.n_slots = n_slots;
  m.s2b.assign(slot_to_block, slot_to_block + n_slots);
  m.valid = true; m.active = true; m.materialized = false;
  HostGroup& g = c->groups[BSGPU_F_REPROJ];
  g.n = n; g.idx.clear(); g.consts.clear(); g.loss_kind.clear(); g.loss_a.clear();
  const int rc = invalidate_keep_values(c);
  if (rc != BSGPU_OK) return rc;
  // ---- device copy (what finalize() flattens from when the window is large enough for the device path)
  {
    const char* fe = getenv("BSGPU_FLATTEN");   // (forced device flattening — the tests — keeps the resident copy of a small window too)
    if (n < kDeviceFlattenMin && !(fe && !strcmp(fe, "device") && n > 0)) { m.dev_valid = false; return BSGPU_OK; }
  }
  HIPCHK(c, hipSetDevice(c->device));
  if ((size_t)n > m.d_cap) {
    const size_t cap = (size_t)n + (size_t)n / 4 + 4096;
    int32_t* ni = nullptr; double* nc = nullptr; int32_t* nk = nullptr; double* na = nullptr;
    HIPCHK(c, hipMalloc((void**)&ni, sizeof(int32_t) * 4 * cap)); HIPCHK(c, hipMalloc((void**)&nc, sizeof(double) * 3 * cap));
    HIPCHK(c, hipMalloc((void**)&nk, sizeof(int32_t) * cap)); HIPCHK(c, hipMalloc((void**)&na, sizeof(double) * cap));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (m.d_idx) (void)hipFree(m.d_idx);
    if (m.d_consts) (void)hipFree(m.d_consts);
    if (m.d_lk) (void)hipFree(m.d_lk);
    if (m.d_la) (void)hipFree(m.d_la);
    m.d_idx = ni; m.d_consts = nc; m.d_lk = nk; m.d_la = na; m.d_cap = cap;
    m.dev_valid = false;   // (a grown table is re-sent whole: happens once per 25 % of growth)
  }
  if ((size_t)n_slots > m.d_s2b_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (m.d_s2b) (void)hipFree(m.d_s2b);
    m.d_s2b = nullptr; m.d_s2b_cap = 0;
    const size_t cap = (size_t)n_slots + (size_t)n_slots / 4 + 1024;
    HIPCHK(c, hipMalloc((void**)&m.d_s2b, sizeof(int32_t) * cap));
    m.d_s2b_cap = cap;
  }
  if (!m.dev_valid) {
    HIPCHK(c, hipMemcpyAsync(m.d_idx, m.idx.data(), sizeof(int32_t) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(m.d_consts, m.consts.data(), sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(m.d_lk, m.loss_kind.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(m.d_la, m.loss_a.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  } else if (n_changed > 0) {
    // packed changed rows: [idx x 4 | rows | loss kind] ints, then [consts x 3 | loss a] doubles
    const size_t nch = (size_t)n_changed;
    const size_t bytes_i = sizeof(int32_t) * nch * 6, off_d = (bytes_i + 15) & ~(size_t)15, bytes = off_d + sizeof(double) * nch * 4;
    if (bytes > m.stage_cap) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (m.h_stage) (void)hipHostFree(m.h_stage);
      if (m.d_stage) (void)hipFree(m.d_stage);
      m.h_stage = nullptr; m.d_stage = nullptr; m.stage_cap = 0;
      const size_t cap = bytes * 2 + 4096;
      HIPCHK(c, hipHostMalloc((void**)&m.h_stage, cap)); HIPCHK(c, hipMalloc((void**)&m.d_stage, cap));
      m.stage_cap = cap;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the staging buffer of the previous call has long been consumed; this makes it certain)
    int32_t* hi = reinterpret_cast<int32_t*>(m.h_stage);
    double* hd = reinterpret_cast<double*>(m.h_stage + off_d);
    for (size_t i = 0; i < nch; ++i) {
      const size_t r = (size_t)changed_rows[i];
      std::memcpy(hi + 4 * i, &m.idx[4 * r], sizeof(int32_t) * 4);
      hi[4 * nch + i] = (int32_t)r;
      hi[5 * nch + i] = m.loss_kind[r];
      std::memcpy(hd + 3 * i, &m.consts[3 * r], sizeof(double) * 3);
      hd[3 * nch + i] = m.loss_a[r];
    }
    HIPCHK(c, hipMemcpyAsync(m.d_stage, m.h_stage, bytes, hipMemcpyHostToDevice, c->stream));
    const int32_t* di = reinterpret_cast<const int32_t*>(m.d_stage);
    const double* dd = reinterpret_cast<const double*>(m.d_stage + off_d);
    launch_patch_factor_rows(c->stream, n_changed, di + 4 * nch, di, dd, di + 5 * nch, dd + 3 * nch, m.d_idx, m.d_consts, m.d_lk, m.d_la);
  }
  HIPCHK(c, hipMemcpyAsync(m.d_s2b, m.s2b.data(), sizeof(int32_t) * (size_t)n_slots, hipMemcpyHostToDevice, c->stream));
  m.dev_valid = true;
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
  return invalidate_keep_values(c);
} catch (...) { return api_exception(c); }
int bsgpu_finalize(bsgpu_ctx* c) try { return c ? finalize(c) : BSGPU_ERR_INVALID; } catch (...) { return api_exception(c); }
int bsgpu_solve(bsgpu_ctx* c, const bsgpu_options* o, bsgpu_summary* s) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!o || !s) return fail(c, BSGPU_ERR_INVALID, "solve: null argument");
  return solve(c, *o, *s);
} catch (...) { return api_exception(c); }
int bsgpu_solve_batch(bsgpu_ctx* const* ctxs, int32_t n, const bsgpu_options* o, int32_t options_stride, bsgpu_summary* s) {
  if (!ctxs || n <= 0 || !o || !s) return BSGPU_ERR_INVALID;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return BSGPU_ERR_INVALID;
    for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(ctxs[i], BSGPU_ERR_INVALID, "solve_batch: the same context twice");
  }
  std::vector<int> rc(n, BSGPU_OK);
  // The windows the batched kernels cover (bsgpu_batch.cpp: Euclidean-landmark windows on the fused factorisation) advance together,
  // one set of launches per LM iteration; every other window gets a thread of its own on its context's stream, as before.
  // BSGPU_BATCH_THREADS=1: the thread-per-window form for all of them.
  std::vector<int> batched, alone;
  static const bool threads_only = getenv("BSGPU_BATCH_THREADS") != nullptr;
  for (int i = 0; i < n; ++i) {
    bool covered = false;
    if (!threads_only) {
      const bsgpu_options& oi = o[options_stride ? i : 0];
      try {
        covered = finalize(ctxs[i]) == BSGPU_OK && batch_covers(ctxs[i], oi);
        if (covered && !batched.empty()) {   // (one device, and the options that are part of the argument tables equal)
          const bsgpu_options& o0 = o[options_stride ? batched[0] : 0];
          covered = ctxs[batched[0]]->device == ctxs[i]->device && o0.jacobi_scaling == oi.jacobi_scaling && o0.min_lm_diagonal == oi.min_lm_diagonal &&
                    o0.max_lm_diagonal == oi.max_lm_diagonal;
        }
      }
      catch (...) { covered = false; }
    }
    (covered ? batched : alone).push_back(i);
  }
  if (batched.size() < 2) { alone.insert(alone.end(), batched.begin(), batched.end()); batched.clear(); }
  auto one = [&](int i) { rc[i] = bsgpu_solve(ctxs[i], o + (options_stride ? i : 0), s + i); };   // (bsgpu_solve catches everything)
  std::vector<std::thread> th;
  size_t started = 0;
  const size_t keep = batched.empty() ? 1 : 0;   // (without a batch the calling thread takes the last lone window itself)
  try {
    for (; started + keep < alone.size(); ++started) th.emplace_back(one, alone[started]);
  } catch (...) {}   // no more threads: the rest in this one
  for (size_t b0 = 0; b0 < batched.size(); b0 += kBatchMaxWin) {
    const int m = (int)std::min<size_t>(kBatchMaxWin, batched.size() - b0);
    bool ok = false;
    try { ok = solve_batched(ctxs, batched.data() + b0, m, o, options_stride, s, rc.data()); } catch (...) { ok = false; }
    if (!ok) for (int q = 0; q < m; ++q) one(batched[b0 + q]);
  }
  for (; started < alone.size(); ++started) one(alone[started]);
  for (auto& t : th) t.join();
  for (int i = 0; i < n; ++i) if (rc[i] != BSGPU_OK) return rc[i];
  return BSGPU_OK;
}
int bsgpu_batch_stats(int64_t* windows_batched, int64_t* rounds) { batch_stats(windows_batched, rounds); return BSGPU_OK; }
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
    std::vector<double> r((size_t)V.n * 2), J((size_t)V.n * (kJAStride + 6));
    std::vector<int4> fac(V.n);
    std::vector<int> cam_pose(V.n), lm_of(V.n), cp_tq(V.n_cam_pose), cp_tp(V.n_cam_pose);
    HIPCHK(c, hipMemcpy(r.data(), V.r, sizeof(double) * 2 * V.n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(J.data(), V.J, sizeof(double) * (kJAStride + 6) * V.n, hipMemcpyDeviceToHost));
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
            // stored: pose parts [n][A row 0 | A row 1] first, then the landmark parts [n][B row 0 | B row 1]
            const double v = sl < 2 ? J[(size_t)i * kJAStride + 6 * k + 3 * sl + j] : J[(size_t)V.n * kJAStride + (size_t)i * 6 + 3 * k + j];
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
  if ((rc = materialize_mirror(c)) != BSGPU_OK) return rc;   // (the factors touching the blocks are picked from the host rows)
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
// One undamped assembly + one factorisation; the unit vectors of the blocks ride along as rows of the rhs tile.
// `blocks` (distinct, pose-side, not constant): out = the joint covariance of their tangent coordinates, D x D row-major, D <= 64.
static int covariance_of(bsgpu_ctx* c, const std::vector<int>& blocks, double* out) {
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  std::vector<int> cols;
  for (size_t i = 0; i < blocks.size(); ++i) {
    const int bl = blocks[i];
    if (bl < 0 || bl >= c->nb) return fail(c, BSGPU_ERR_INVALID, "covariance: block out of range");
    for (size_t j = 0; j < i; ++j) if (blocks[j] == bl) return fail(c, BSGPU_ERR_INVALID, "covariance: block named twice");
    if (c->toff[bl] < 0) return fail(c, BSGPU_ERR_INVALID, "covariance: constant block");
    if (c->is_lm[bl]) return fail(c, BSGPU_ERR_UNSUPPORTED, "covariance: landmark blocks are eliminated; only pose-side blocks can be queried");
    for (int k = 0; k < c->tsize[bl]; ++k) cols.push_back(c->plan.spos(c->toff[bl] + k));
  }
  const int D = (int)cols.size();
  if (D <= 0 || D > 64) return fail(c, BSGPU_ERR_UNSUPPORTED, "covariance: the blocks' tangent dimensions must add up to 1..64 (one rhs tile)");
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
  c->spec_J = false;
  int* d_cols = nullptr;
  double* d_out = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_cols, sizeof(int) * cols.size()));
  if (hipMalloc((void**)&d_out, sizeof(double) * D * D) != hipSuccess) { (void)hipFree(d_cols); return fail(c, BSGPU_ERR_DEVICE, "out of device memory"); }
  (void)hipMemcpyAsync(d_cols, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice, s);
  for (int attempt = 0; attempt < 2; ++attempt) {
    launch_cov_units(s, c->d_S, c->npad, c->plan.rhs_row, d_cols, D);
    DenseDev D0{c->d_nreal, c->d_rows_flat, c->d_panels, c->d_Lp, c->d_Vinv,
                c->d_bs_desc, c->d_chain_begin, c->d_chain_end, c->d_tile_sync, c->d_ftasks, c->d_fsync};
    D0.Winv = c->d_Winv; D0.tile_tot = c->d_tile_tot; D0.rhs_rows = D;
    D0.ftasks_plain = c->d_ftasks_plain; D0.tile_tot_plain = c->d_tile_tot_plain; D0.n_ftasks_plain = c->n_ftasks_plain;
    dense_factor(s, c->plan, D0, c->d_S, c->d_scal);
    launch_cov_dots(s, c->d_Lp, c->npad, c->plan.rhs_row, c->plan.T * 64, D, 0, D, d_out);
    (void)hipMemcpyAsync(out, d_out, sizeof(double) * D * D, hipMemcpyDeviceToHost, s);
    rc = fetch_scalars(c);
    if (rc != BSGPU_OK || !(c->h_scal[SC_CHOL_FAIL] == 2.0 && c->d_ftasks)) break;
    // a wait inside the single-launch factorisation timed out (the device is shared — e.g. next to a bsgpu_solve_batch): not a numerical
    // failure.  As solve() does: this context takes the launch-per-step path from here on, and the system is assembled and factored again.
    c->d_ftasks = nullptr;
    assemble(c, o, 1e300, true, true);
  }
  (void)hipFree(d_cols); (void)hipFree(d_out);
  if (rc != BSGPU_OK) return rc;
  if (c->h_scal[SC_CHOL_FAIL] > 0.0 || !std::isfinite(out[0]))
    return fail(c, BSGPU_ERR_NUMERIC, "covariance: J^T J is singular at the current values (gauge freedom or unobserved block)");
  return BSGPU_OK;
}
int bsgpu_covariance(bsgpu_ctx* c, int32_t ba, int32_t bb, double* out) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!out) return fail(c, BSGPU_ERR_INVALID, "null argument");
  int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  if (ba < 0 || bb < 0 || ba >= c->nb || bb >= c->nb) return fail(c, BSGPU_ERR_INVALID, "covariance: block out of range");
  if (ba == bb) return covariance_of(c, {ba}, out);
  const int ta = c->tsize[ba], tb = c->tsize[bb];
  std::vector<double> joint((size_t)(ta + tb) * (ta + tb));
  rc = covariance_of(c, {ba, bb}, joint.data());
  if (rc != BSGPU_OK) return rc;
  for (int i = 0; i < ta; ++i) for (int j = 0; j < tb; ++j) out[i * tb + j] = joint[(size_t)i * (ta + tb) + ta + j];
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_covariance_joint(bsgpu_ctx* c, int32_t n_blocks, const int32_t* blocks, double* out) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (n_blocks <= 0 || !blocks || !out) return fail(c, BSGPU_ERR_INVALID, "covariance_joint: bad argument");
  return covariance_of(c, std::vector<int>(blocks, blocks + n_blocks), out);
} catch (...) { return api_exception(c); }

// The payload of the index-th dense linear prior, replaced in place: the finalized device structure stays (same blocks, rows, columns).
int bsgpu_update_marginal(bsgpu_ctx* c, int32_t index, int32_t n_rows, int32_t n_cols, int32_t n_xbar, const double* A, const double* b,
                          const double* xbar) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (index < 0 || index >= (int)c->marginals.size() || !A || !b || !xbar) return fail(c, BSGPU_ERR_INVALID, "update_marginal: bad argument");
  HostMarginal& mg = c->marginals[index];
  if (n_rows != mg.rows || (size_t)n_rows * (size_t)std::max(0, n_cols) != mg.A.size() || (size_t)n_rows != mg.b.size() || (size_t)n_xbar != mg.xbar.size())
    return fail(c, BSGPU_ERR_INVALID, "update_marginal: the payload's shape is not that of the prior as it was added");
  std::memcpy(mg.A.data(), A, sizeof(double) * mg.A.size());
  std::memcpy(mg.b.data(), b, sizeof(double) * mg.b.size());
  std::memcpy(mg.xbar.data(), xbar, sizeof(double) * mg.xbar.size());
  if (!c->finalized) return BSGPU_OK;
  if (index >= (int)c->marg.size()) return fail(c, BSGPU_ERR_INVALID, "update_marginal: device table out of step");
  HIPCHK(c, hipSetDevice(c->device));
  const MargDev& d = c->marg[index].dev;
  HIPCHK(c, hipStreamSynchronize(c->stream));   // (kernels of an earlier step may still read the payload)
  HIPCHK(c, hipMemcpy(const_cast<double*>(d.A), mg.A.data(), sizeof(double) * mg.A.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(const_cast<double*>(d.b), mg.b.data(), sizeof(double) * mg.b.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(const_cast<double*>(d.xbar), mg.xbar.data(), sizeof(double) * mg.xbar.size(), hipMemcpyHostToDevice));
  c->spec_J = false;
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
// Residuals + Jacobians of EVERY factor type at the current values, `reps` times between two HIP events on the solver's stream
// (what an LM iteration starts with; the measurement behind the evaluation roofline of windows without reprojection factors).
double bsgpu_time_eval_ms(bsgpu_ctx* c, int32_t reps) {
  if (!c) return -1.0;
  if (finalize(c) != BSGPU_OK || reps <= 0) return -1.0;
  if (hipSetDevice(c->device) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  eval_all(c, c->d_x, true, SC_COST_X);   // warm
  (void)hipEventRecord(e0, c->stream);
  for (int i = 0; i < reps; ++i) eval_all(c, c->d_x, true, SC_COST_X);
  (void)hipEventRecord(e1, c->stream);
  if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  c->spec_J = false;
  return (double)ms / reps;
}
// ... and its algorithmic bytes: per factor indices + constants in, residual + tangent Jacobian out; every
// parameter block once
int64_t bsgpu_eval_bytes(const bsgpu_ctx* c) {
  if (!c) return -1;
  // (idx, consts, residuals, tangent Jacobian columns) per type, in doubles / ints as the tables hold them
  static const struct { int idx, consts, res, jcols; } L[BSGPU_F_NUM_TYPES] = {
      {4, 3, 2, 9}, {6, 3, 2, 9}, {10, 287, 15, 30}, {5, 241, 15, 15}, {6, 43, 6, 12}, {4, 43, 6, 12}, {2, 43, 6, 6}, {1, 12, 3, 3},
      {2, 12, 3, 6}, {1, 7, 2, 3}, {6, 6, 2, 13}, {4, 6, 2, 7}};
  int64_t b = (int64_t)c->vis.n * 200 + (int64_t)c->h_x.size() * 8;
  for (int t = 2; t < BSGPU_F_NUM_TYPES && t < kNumInternal; ++t)
    b += (int64_t)c->small[t].n * (4 * L[t].idx + 8 * (L[t].consts + L[t].res + L[t].res * L[t].jcols));
  return b;
}
// block-sparse PCG path (pose graphs above the dense limit): block rows and non-zero 3x3 blocks of J^T J (0, 0 on the dense path)
int bsgpu_bsr_info(bsgpu_ctx* c, int32_t* block_rows, int32_t* nnz_blocks) try {
  if (!c) return BSGPU_ERR_INVALID;
  const int rc = finalize(c);
  if (rc != BSGPU_OK) return rc;
  if (block_rows) *block_rows = c->use_pcg ? c->nbr : 0;
  if (nnz_blocks) *nnz_blocks = c->use_pcg ? c->nblk : 0;
  return BSGPU_OK;
} catch (...) { return api_exception(c); }
int bsgpu_profile_step(bsgpu_ctx* c, const bsgpu_options* o, int32_t reps, double* ms_out, double* work_out) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (!o || !ms_out || reps <= 0) return fail(c, BSGPU_ERR_INVALID, "profile_step: bad argument");
  return profile_step(c, *o, reps, ms_out, work_out);
} catch (...) { return api_exception(c); }
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
    const char* e3 = getenv("BSGPU_SHARED");
    P.build(n, adj, std::max(1, (int)max_chains), 1, !(e3 && atoi(e3) == 0));
  }
  const int npad = P.npad;
  std::vector<double> hS((size_t)npad * npad, 0.0);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) hS[(size_t)P.spos(i) * npad + P.spos(j)] = A[(size_t)i * n + j];
  for (int j = 0; j < n; ++j) hS[(size_t)P.rhs_row * npad + P.spos(j)] = b[j];
  std::vector<uint8_t> real(npad, 0);
  for (int j = 0; j < n; ++j) real[P.spos(j)] = 1;
  for (int i = 0; i < npad; ++i) if (!real[i]) hS[(size_t)i * npad + i] = 1.0;
  double *dS = nullptr, *dLp = nullptr, *dV = nullptr, *dy = nullptr, *dscal = nullptr;
  int *dnreal = nullptr, *drows = nullptr;
  PanelDesc *dpan = nullptr, *dsep = nullptr;
  int *dpot2 = nullptr, *dcb = nullptr, *dce = nullptr, *dsync = nullptr, *dfsync = nullptr, *dtot = nullptr;
  double* dW = nullptr;
  FusedTask* dft = nullptr;
  const char* ef = getenv("BSGPU_CHOL_FUSED");
  const bool fused = !(ef && atoi(ef) == 0) && !P.ftasks.empty();
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return BSGPU_ERR_DEVICE;
  auto up = [](const void* src, size_t bytes, void** dst) {
    if (hipMalloc(dst, bytes ? bytes : 8) != hipSuccess) return false;
    return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = hipMalloc(&dS, sizeof(double) * hS.size()) == hipSuccess && hipMalloc(&dLp, sizeof(double) * hS.size()) == hipSuccess && hipMemset(dLp, 0, sizeof(double) * hS.size()) == hipSuccess &&
            hipMalloc(&dV, sizeof(double) * chol_vinv_stride() * std::max(1, T)) == hipSuccess && hipMalloc(&dy, sizeof(double) * npad) == hipSuccess &&
            hipMalloc(&dscal, sizeof(double) * SC_NUM) == hipSuccess;
  std::vector<int> rf = P.rows_flat; if (rf.empty()) rf.push_back(0);
  ok = ok && up(P.nreal.data(), sizeof(int) * P.nreal.size(), (void**)&dnreal) &&
       up(rf.data(), sizeof(int) * rf.size(), (void**)&drows) &&
       up(P.panels.data(), sizeof(PanelDesc) * P.panels.size(), (void**)&dpan) &&
       up(P.bs_desc.data(), sizeof(int) * P.bs_desc.size(), (void**)&dpot2) &&
       up(P.chain_begin.data(), sizeof(int) * P.chain_begin.size(), (void**)&dcb) && up(P.chain_end.data(), sizeof(int) * P.chain_end.size(), (void**)&dce) &&
       up(P.tile_sync.data(), sizeof(int) * P.tile_sync.size(), (void**)&dsync);
  if (ok && fused) {
    const std::vector<int> zeros((size_t)P.fused_sync_words, 0);
    ok = up(P.ftasks.data(), sizeof(FusedTask) * P.ftasks.size(), (void**)&dft) && up(zeros.data(), sizeof(int) * zeros.size(), (void**)&dfsync) &&
         up(P.tile_tot.data(), sizeof(int) * P.tile_tot.size(), (void**)&dtot) && hipMalloc((void**)&dW, sizeof(double) * 4096 * std::max(1, T)) == hipSuccess;
  }
  int rc = BSGPU_OK;
  int *dbc = nullptr, *drc = nullptr, *dbu = nullptr, *dbur = nullptr;
  if (ok) {
    (void)hipMemcpy(dS, hS.data(), sizeof(double) * hS.size(), hipMemcpyHostToDevice);
    (void)hipMemset(dscal, 0, sizeof(double) * SC_NUM);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    DenseDev D{dnreal, drows, dpan, dLp, dV, dpot2, dcb, dce, dsync, dft, dfsync};
    D.Winv = dW; D.tile_tot = dtot;
    if (P.bs_level_sync && !getenv("BSGPU_BACKSOLVE_LEGACY") &&
        up(P.bs_desc_chain.data(), sizeof(int) * P.bs_desc_chain.size(), (void**)&dbc) && up(P.rows_flat_chain.data(), sizeof(int) * P.rows_flat_chain.size(), (void**)&drc) &&
        up(P.bs_upd.data(), sizeof(int) * P.bs_upd.size(), (void**)&dbu) && up(P.bs_upd_rows.data(), sizeof(int) * P.bs_upd_rows.size(), (void**)&dbur)) {
      D.bs_desc_chain = dbc; D.rows_flat_chain = drc; D.bs_upd = dbu; D.bs_upd_rows = dbur;
    }
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
  (void)hipFree(dnreal); (void)hipFree(drows); (void)hipFree(dpan);
  (void)hipFree(dsep); (void)hipFree(dpot2); (void)hipFree(dcb); (void)hipFree(dce); (void)hipFree(dsync);
  (void)hipFree(dft); (void)hipFree(dfsync); (void)hipFree(dtot); (void)hipFree(dW);
  (void)hipFree(dbc); (void)hipFree(drc); (void)hipFree(dbu); (void)hipFree(dbur);
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

int bsgpu_set_plan_preference(bsgpu_ctx* c, int32_t preference) try {
  if (!c) return BSGPU_ERR_INVALID;
  if (preference != BSGPU_PLAN_LATENCY && preference != BSGPU_PLAN_THROUGHPUT) return fail(c, BSGPU_ERR_INVALID, "bsgpu_set_plan_preference: BSGPU_PLAN_LATENCY or BSGPU_PLAN_THROUGHPUT");
  if (c->plan_pref != preference) { c->plan_pref = preference; c->dim_model = -1; return invalidate_keep_values(c); }   // (a finalized context is planned again at its next finalize, from the point it has reached)
  return BSGPU_OK;
} catch (...) { return api_exception(c); }

}  // extern "C"

