// finalize(): flattening of the described problem to device tables — what [EXT] fuse HashGraph::createProblem does for the
// reference's `graph_->optimize()` (bs_optimizers/src/fixed_lag_smoother.cpp:281), with a deterministic variable index.
#include <atomic>

#include "bsgpu_ctx.h"
#include "band_plan.h"
#include "dim_order.h"

namespace bsg {

int band_part_forced() {
  const char* e = getenv("BSGPU_BAND_PART");   // (read at every finalize: tests/test_gpu_band.py changes it between solves)
  return e ? atoi(e) : 0;
}

namespace {
void eigen_quat_to_rot(const double* q, double* R) {
  const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
  const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0], txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

}  // namespace

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
  std::vector<int> lm_use(nb, 0), other_use(nb, 0), rho_use(nb, 0);   // rho_use: as the inverse-depth slot of an inverse-depth factor
  for (int t = 0; t < BSGPU_F_NUM_TYPES; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    if (t == BSGPU_F_REPROJ && c->mirror0.active && !c->mirror0.materialized) {
      // the table lives in the slot-named mirror (bsgpu_sync_factors_indirect): the same checks and use counts from its per-slot
      // counters — a pass over the slots, not over the rows
      const SlotMirror& m = c->mirror0;
      const size_t ns = std::max({m.use_q.size(), m.use_p.size(), m.use_l.size()});
      for (size_t sl = 0; sl < ns; ++sl) {
        const int uq = sl < m.use_q.size() ? m.use_q[sl] : 0, up = sl < m.use_p.size() ? m.use_p[sl] : 0, ul = sl < m.use_l.size() ? m.use_l[sl] : 0;
        if (!(uq | up | ul)) continue;
        const int b = sl < (size_t)m.n_slots ? m.s2b[sl] : -1;
        if (b < 0 || b >= nb) return fail(c, BSGPU_ERR_INVALID, "sync_factors_indirect: slot out of range or not mapped to a block");
        if (uq && (c->size[b] != 4 || c->manifold[b] != BSGPU_MANIFOLD_QUAT_RIGHT))
          return fail(c, BSGPU_ERR_INVALID, c->size[b] != 4 ? "block size does not match factor slot" : "4-d slot must be a quaternion-manifold block");
        if ((up || ul) && c->size[b] != 3) return fail(c, BSGPU_ERR_INVALID, "block size does not match factor slot");
        other_use[b] += uq + up; lm_use[b] += ul;
      }
      for (size_t cam = 0; cam < m.cam_use.size(); ++cam)
        if (m.cam_use[cam] > 0 && cam >= c->cams.size()) return fail(c, BSGPU_ERR_INVALID, "camera index out of range");
      continue;
    }
    for (int f = 0; f < g.n; ++f) {
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sl = 0; sl < ti.nvar; ++sl) {
        const int b = idx[sl];
        if (b < 0 || b >= nb) return fail(c, BSGPU_ERR_INVALID, "factor references block out of range");
        if (c->size[b] != ti.amb[sl]) return fail(c, BSGPU_ERR_INVALID, "block size does not match factor slot");
        if (ti.amb[sl] == 4 && c->manifold[b] != BSGPU_MANIFOLD_QUAT_RIGHT)
          return fail(c, BSGPU_ERR_INVALID, "4-d slot must be a quaternion-manifold block");
        if (t <= 1 && sl == 2) lm_use[b]++; else other_use[b]++;
        if ((t == BSGPU_F_IDP_REPROJ || t == BSGPU_F_IDP_REPROJ_UNARY) && sl == ti.nvar - 1) rho_use[b]++;
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
  const bool idp_elim = !(getenv("BSGPU_IDP_ELIM") && atoi(getenv("BSGPU_IDP_ELIM")) == 0);
  for (int b = 0; b < nb; ++b) {
    if (c->size[b] > 4 || c->size[b] == 0) return fail(c, BSGPU_ERR_UNSUPPORTED, "block sizes 1..4 only");
    if (c->manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT && c->size[b] != 4) return fail(c, BSGPU_ERR_INVALID, "quaternion block must have size 4");
    c->tsize[b] = (c->manifold[b] == BSGPU_MANIFOLD_QUAT_RIGHT) ? 3 : c->size[b];
    if (c->is_const[b]) continue;
    if (lm_use[b] > 0 && other_use[b] == 0 && c->size[b] == 3 && c->manifold[b] == BSGPU_MANIFOLD_EUCLIDEAN &&
        !(b < (int)c->no_elim.size() && c->no_elim[b])) c->is_lm[b] = 1;
    // a scalar block whose every use is the inverse-depth slot of an inverse-depth factor: eliminated on the landmark side too
    // (k_idp.hip; BSGPU_IDP_ELIM=0 keeps such blocks in the reduced system, as leaf tiles of the factorisation)
    if (idp_elim && lm_use[b] == 0 && rho_use[b] > 0 && rho_use[b] == other_use[b] && c->size[b] == 1 &&
        c->manifold[b] == BSGPU_MANIFOLD_EUCLIDEAN && !(b < (int)c->no_elim.size() && c->no_elim[b])) c->is_lm[b] = 2;
  }
  int to = 0;
  for (int b = 0; b < nb; ++b) if (!c->is_const[b] && !c->is_lm[b]) { c->toff[b] = to; to += c->tsize[b]; }
  c->n_pose = to;
  std::vector<int> lm_index(nb, -1);
  int nl = 0;
  for (int b = 0; b < nb; ++b) if (!c->is_const[b] && c->is_lm[b] == 1) { c->toff[b] = to; to += 3; lm_index[b] = nl++; }
  // (the scalar landmarks last, one tangent column each: in a window without Euclidean landmarks the tangent order is then the block order)
  std::vector<int> idp_index(nb, -1);
  int n_rho = 0;
  const int idp_to0 = to;
  for (int b = 0; b < nb; ++b) if (!c->is_const[b] && c->is_lm[b] == 2) { c->toff[b] = to; to += 1; idp_index[b] = n_rho++; }
  c->n_tan = to; c->n_lm = nl; c->n_idp_lm = n_rho;
  // every binary inverse-depth factor has an eliminated landmark: the factors' own pose-pose terms are assembled with the elimination
  // (idp_pairs_kernel) and the group leaves the generic pose-only assembly.  The unary factor's Jacobian is identically zero.
  bool idp_direct = n_rho > 0 && !getenv("BSGPU_IDP_GENERIC_ASSEMBLY");
  {
    const HostGroup& g = c->groups[BSGPU_F_IDP_REPROJ];
    const int ni = kTypes[BSGPU_F_IDP_REPROJ].nidx;
    for (int f = 0; f < g.n && idp_direct; ++f) idp_direct = idp_index[g.idx[(size_t)f * ni + 4]] >= 0;
  }
  auto skip_generic_assembly = [&](int t) { return (t == BSGPU_F_IDP_REPROJ && idp_direct) || (t == BSGPU_F_IDP_REPROJ_UNARY && n_rho > 0); };
  c->npad = ((c->n_pose + 63) / 64 + 1) * 64;   // real tiles + one tile for the rhs row (dense_plan.h)
  // leaf tiles of the reduced system: tiles made of inverse-depth landmarks only — scalar blocks that no factor couples to each other
  // (an inverse-depth factor has one), so the tiled factorisation can eliminate them first (dense_plan.h build(): leaf)
  c->leaf_tile.assign((c->n_pose + 63) / 64, 0);
  {
    const int T0 = (c->n_pose + 63) / 64;
    std::vector<uint8_t> dim_leaf(c->n_pose, 0);
    bool any = false;
    for (int b = 0; b < nb; ++b)
      if (!c->is_const[b] && !c->is_lm[b] && c->size[b] == 1 && rho_use[b] > 0 && rho_use[b] == other_use[b]) { dim_leaf[c->toff[b]] = 1; any = true; }
    if (any)
      for (int t = 0; t < T0; ++t) {
        bool all = true;
        for (int d = 64 * t; d < std::min(c->n_pose, 64 * t + 64) && all; ++d) all = dim_leaf[d] != 0;
        c->leaf_tile[t] = all ? 1 : 0;
      }
    c->n_leaf_tiles = 0;
    for (uint8_t v : c->leaf_tile) c->n_leaf_tiles += v;
  }
  // (pose-only graphs above kDenseLimit go to the block-sparse PCG unless the exact factorisation is asked for — BSGPU_EXACT_POSE_GRAPH=1 at
  // finalize(): the dense tile storage, 2 x npad^2 doubles, is not allocated on spec; C4 that way: DESIGN.md 3.3)
  const bool exact_pose_graph = getenv("BSGPU_EXACT_POSE_GRAPH") != nullptr && atoi(getenv("BSGPU_EXACT_POSE_GRAPH")) != 0;   // (forced: whatever the plan costs)
  c->dense_ok = (size_t)c->npad <= kDenseLimit || ((nl + n_rho > 0 || c->n_leaf_tiles > 0 || exact_pose_graph) && (size_t)c->npad <= kDenseLimitLandmarks);   // else: block-sparse PCG (pose-only problems)
  // A pose graph above kDenseLimit whose loop closures are LOCAL (a mapper's: poses near each other) dissects into many small supernodes,
  // and its exact step — the reference's own call, SPARSE_NORMAL_CHOLESKY, submap_pose_graph_optimization.cpp:144-146 — is then cheaper
  // than the PCG's (synthetic.pose_graph_local: 2.9 against 5.3 ms per LM iteration).  Whether that is so is only known once the order
  // has been found: the window is taken as dense provisionally, and goes back to the PCG below if the plan says otherwise.
  // BSGPU_EXACT_POSE_GRAPH=0 keeps every such graph on the PCG.
  const bool dense_on_trial = !c->dense_ok && nl + n_rho == 0 && (size_t)c->npad <= kDenseLimitLandmarks && c->marginals.empty() &&
                              !(getenv("BSGPU_EXACT_POSE_GRAPH") && atoi(getenv("BSGPU_EXACT_POSE_GRAPH")) == 0);
  if (dense_on_trial) c->dense_ok = true;
  // The graph of the reduced system's tangent BLOCKS (which pairs of pose-side blocks some factor, some shared landmark or a dense prior
  // couples): what the per-dimension ordering of the factorisation is found on (dim_order.h).  A bit matrix; every place below that marks
  // the natural-tile adjacency marks it too.  Not kept for windows that take the tile-level order (leaf tiles of BSGPU_IDP_ELIM=0,
  // systems beyond the dense limit, BSGPU_DIM_ORDER=0).
  struct BlockGraph {
    int nbk = 0, words = 0;
    std::vector<int> bid_of_t, t0, w;
    std::vector<uint64_t> bits;
    inline void add(int ta, int tb) {
      const int a = bid_of_t[ta], b = bid_of_t[tb];
      bits[(size_t)a * words + (b >> 6)] |= 1ull << (b & 63);
      bits[(size_t)b * words + (a >> 6)] |= 1ull << (a & 63);
    }
  } bg;
  {
    const char* ed = getenv("BSGPU_DIM_ORDER");
    int nbk = 0;
    for (int b = 0; b < nb; ++b) if (c->toff[b] >= 0 && c->toff[b] < c->n_pose) ++nbk;
    if (c->dense_ok && c->n_leaf_tiles == 0 && nbk > 0 && nbk <= 16384 && !(ed && atoi(ed) == 0)) {
      bg.nbk = nbk; bg.words = (nbk + 63) / 64;
      bg.bid_of_t.assign(c->n_pose, -1);
      bg.t0.reserve(nbk); bg.w.reserve(nbk);
      for (int b = 0; b < nb; ++b) if (c->toff[b] >= 0 && c->toff[b] < c->n_pose) {   // (block order = tangent order)
        for (int k = 0; k < c->tsize[b]; ++k) bg.bid_of_t[c->toff[b] + k] = (int)bg.t0.size();
        bg.t0.push_back(c->toff[b]); bg.w.push_back(c->tsize[b]);
      }
      bg.bits.assign((size_t)nbk * bg.words, 0);
    }
  }
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

  // band landmarks on the matrix cores (k_band.hip) in windows of at least kBandMinFactors reprojection factors: a unit of that kernel is a
  // first camera pose's landmarks, and a window of the reference's own size has too few of them per pose to fill the device (measured,
  // LM it/s band / entries: 20 KF x 500 8 410 / 8 850, 50 x 5 000 6 390 / 6 500, 100 x 20 000 4 720 / 4 690, 200 x 50 000 3 310 / 3 140).
  // BSGPU_PAIRS_BAND=0: every pair by entries (the cross-check), =1: band landmarks whatever the size (tests).
  const char* band_env = getenv("BSGPU_PAIRS_BAND");
  const int n_reproj = c->groups[BSGPU_F_REPROJ].n + c->groups[BSGPU_F_REPROJ_ONLINE_CALIB].n;
  // (band_available(): the kernel's ~147 KB of dynamic LDS per workgroup, asked of this device once — a device or partition without it keeps the pair entries)
  const bool band_on = (band_env ? strcmp(band_env, "0") != 0 : n_reproj >= kBandMinFactors) && band_available();
  const bool sort_entries = getenv("BSGPU_PAIR_ENTRIES_SORT") != nullptr;   // (tests: the path windows of more than 2 896 camera poses take)
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
  { const int rc_m = materialize_mirror(c); if (rc_m != BSGPU_OK) return rc_m; }
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
    // band landmarks (band_plan.h): no pair entries — everything they add to the reduced system is pairs_band_kernel's
    std::vector<int> b_cmin(nl, -1), b_mask(nl, 0);
    std::vector<int4> b_rec;
    if (band_on && nl > 0) band_classify_host(nl, lm_start.data(), cam_pose.data(), b_cmin, b_mask, b_rec);
    BandUnits bu;
    band_units(nl, b_cmin.data(), b_mask.data(), V.n_cam_pose, bu);
    std::vector<int4> b_lm(bu.lm.size());
    for (size_t i = 0; i < bu.lm.size(); ++i) b_lm[i] = b_rec[bu.lm[i]];
    V.n_band_lm = (int)bu.lm.size(); V.n_band_units = (int)bu.unit_cam.size();
    lap("band landmarks");
    // pair entries (factor a, factor b) of every landmark, grouped by camera-pose pair (ca <= cb); inside a group the
    // order is landmark-major.  Two passes over the landmarks: count per pair key, then fill in place.
    const uint64_t ncp = (uint64_t)std::max(1, V.n_cam_pose);
    std::vector<int> seg_ci, seg_cj, seg_start, ent_fa, ent_fb;
    if (ncp * ncp <= (uint64_t)8 << 20 && !sort_entries) {
      std::vector<int> start(ncp * ncp + 1, 0);
      for (int l = 0; l < nl; ++l) {
        if (b_cmin[l] >= 0) continue;
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a) {
          const uint64_t ra = (uint64_t)cam_pose[a] * ncp;
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b) if (cam_pose[a] <= cam_pose[b]) start[ra + cam_pose[b] + 1]++;
        }
      }
      for (int f = n_elim; f < nv; ++f) start[(uint64_t)cam_pose[f] * ncp + cam_pose[f] + 1]++;
      for (size_t i = 0; i < ncp * ncp; ++i) start[i + 1] += start[i];
      const size_t n_ent = (size_t)start[ncp * ncp];
      ent_fa.resize(n_ent); ent_fb.resize(n_ent);
      lap("count pair entries");
      // segments (chunks of <= pair_chunk entries of one pair) straight from the counts
      const int chunk = pair_chunk(n_ent);
      for (uint64_t key = 0; key < ncp * ncp; ++key)
        for (int p0 = start[key]; p0 < start[key + 1]; p0 += chunk) { seg_ci.push_back((int)(key / ncp)); seg_cj.push_back((int)(key % ncp)); seg_start.push_back(p0); }
      std::vector<int> pos(start.begin(), start.end() - 1);
      for (int l = 0; l < nl; ++l) {
        if (b_cmin[l] >= 0) continue;
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a) {
          const uint64_t ra = (uint64_t)cam_pose[a] * ncp;
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
            if (cam_pose[a] <= cam_pose[b]) { const int p = pos[ra + cam_pose[b]]++; ent_fa[p] = a; ent_fb[p] = b; }
        }
      }
      for (int f = n_elim; f < nv; ++f) { const int p = pos[(uint64_t)cam_pose[f] * ncp + cam_pose[f]]++; ent_fa[p] = f; ent_fb[p] = f; }
      lap("fill pair entries");
    } else {   // very many camera poses: comparison sort of explicit entries
      struct Ent { uint64_t key; int fa, fb; };
      std::vector<Ent> ents;
      ents.reserve((size_t)nv * 5);
      for (int l = 0; l < nl; ++l) {
        if (b_cmin[l] >= 0) continue;
        for (int a = lm_start[l]; a < lm_start[l + 1]; ++a)
          for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
            if (cam_pose[a] <= cam_pose[b]) ents.push_back({(uint64_t)cam_pose[a] * ncp + cam_pose[b], a, b});
      }
      for (int f = n_elim; f < nv; ++f) ents.push_back({(uint64_t)cam_pose[f] * ncp + cam_pose[f], f, f});
      std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.key < y.key; });
      ent_fa.resize(ents.size()); ent_fb.resize(ents.size());
      const int chunk = pair_chunk(ents.size());
      for (size_t i = 0; i < ents.size(); ++i) {
        if (i == 0 || ents[i].key != ents[i - 1].key || (int)i - seg_start.back() >= chunk) {
          seg_ci.push_back((int)(ents[i].key / ncp)); seg_cj.push_back((int)(ents[i].key % ncp)); seg_start.push_back((int)i);
        }
        ent_fa[i] = ents[i].fa; ent_fb[i] = ents[i].fb;
      }
      lap("sort pair entries");
    }
    seg_start.push_back((int)ent_fa.size());
    V.n_seg = (int)seg_ci.size(); V.n_ent = (int)ent_fa.size();
    V.n_seg_c = 0; V.seg_ci_c = V.seg_cj_c = V.seg_start_c = nullptr;   // (the coarse list of the batch: rebuilt on first use)
    lap("segments");
    V.fac = c->upload(fac); V.pix = c->upload(pix); V.w = c->upload(w);
    V.cam_pose = c->upload(cam_pose); V.lm_of = c->upload(lm_of); V.lm_start = c->upload(lm_start);
    V.cp_tq = c->upload(cp_tq); V.cp_tp = c->upload(cp_tp);
    V.seg_ci = c->upload(seg_ci); V.seg_cj = c->upload(seg_cj); V.seg_start = c->upload(seg_start);
    V.ent_fa = c->upload(ent_fa); V.ent_fb = c->upload(ent_fb);
    V.band_lm = c->upload(b_lm); V.band_unit_start = c->upload(bu.unit_start); V.band_unit_cam = c->upload(bu.unit_cam);
    V.band_lm_id = c->upload(bu.lm);
    // structural tile adjacency of the reduced system (natural 64-wide tiles) for the Cholesky plan
    const int T = (c->n_pose + 63) / 64;
    c->tile_adj.assign((size_t)T * T, 0);
    auto touch = [&](int ra, int rb) {  // tangent rows ra, rb (start of 3-blocks)
      if (ra < 0 || rb < 0) return;
      for (int a = ra; a < ra + 3; a += 2) for (int b = rb; b < rb + 3; b += 2) {
        c->tile_adj[(size_t)(a / 64) * T + b / 64] = 1; c->tile_adj[(size_t)(b / 64) * T + a / 64] = 1;
      }
      if (bg.nbk) bg.add(ra, rb);
    };
    for (int s = 0; s < V.n_seg; ++s) {
      const int i = seg_ci[s], j = seg_cj[s];
      const int ri[2] = {cp_tq[i], cp_tp[i]}, rj[2] = {cp_tq[j], cp_tp[j]};
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) touch(ri[a], rj[b]);
    }
    // (the camera-pose pairs of the band landmarks have no segments: BandUnits::adj)
    for (int i = 0; i < V.n_cam_pose && V.n_band_lm > 0; ++i)
      for (int d = 0; d < kBandCams; ++d) {
        if (!((bu.adj[i] >> d) & 1u)) continue;
        const int j = i + d;
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
    SlotMirror& mir = c->mirror0;
    const bool resident = mir.active && !mir.materialized && mir.dev_valid;
    if (!force_host && c->groups[BSGPU_F_REPROJ_ONLINE_CALIB].n == 0 && g0.n > 0 && (force_dev || g0.n >= kDeviceFlattenMin)) {
      // distinct losses of the reprojection factors (a window has one or two)
      bool ok = true;
      if (mir.active && !mir.materialized && !resident) { const int rc_m = materialize_mirror(c); if (rc_m != BSGPU_OK) return rc_m; }
      if (resident) { for (const auto& u : mir.loss_use) if (u.rows > 0 && ok) { get_loss(u.kind, u.a); ok = losses.size() <= 8; } }
      else for (int f = 0; f < g0.n && ok; ++f) { get_loss(g0.loss_kind[f], g0.loss_a[f]); ok = losses.size() <= 8; }
      if (ok) {
        std::vector<int> bx(c->off.begin(), c->off.end());
        std::vector<unsigned char> bc(c->is_const.begin(), c->is_const.end());
        const int* d_bx = c->upload(bx); const int* d_bt = c->upload(c->toff);
        const unsigned char* d_bc = c->upload(bc); const int* d_bl = c->upload(lm_index);
        const int T = (c->n_pose + 63) / 64;
        bool all_const = false;
        FlattenSegsHost segs_host;
        auto dalloc = [&](size_t bytes) -> void* { return c->alloc<unsigned char>(bytes); };
        const FlattenResident res = {mir.d_idx, mir.d_consts, mir.d_lk, mir.d_la, mir.d_s2b};
        const int st = flatten_visual_device(c->stream, dalloc, g0.n, g0.idx.data(), g0.consts.data(), g0.loss_kind.data(), g0.loss_a.data(),
                                             losses, nb, d_bx, d_bt, d_bc, d_bl, nl, T, c->vis, &c->d_vis_src, c->tile_adj, &all_const,
                                             resident ? &res : nullptr, bg.nbk ? &segs_host : nullptr, band_on);
        if (st < 0) return fail(c, BSGPU_ERR_DEVICE, "device error while flattening the reprojection factors");
        if (st == 0) {
          flattened_on_device = true;
          if (bg.nbk)
            for (size_t sgi = 0; sgi < segs_host.seg_ci.size(); ++sgi) {
              const int i = segs_host.seg_ci[sgi], j = segs_host.seg_cj[sgi];
              const int ri[2] = {segs_host.cp_tq[i], segs_host.cp_tp[i]}, rj[2] = {segs_host.cp_tq[j], segs_host.cp_tp[j]};
              for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (ri[a] >= 0 && rj[b] >= 0) bg.add(ri[a], rj[b]);
            }
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
    V.r = c->alloc<double2>(nv); V.J = c->alloc<double>((size_t)nv * (kJAStride + 6)); V.JB = V.J ? V.J + (size_t)nv * kJAStride : nullptr; V.CR = c->alloc<double>((size_t)nv * 8);
    V.Linv = c->alloc<double>((size_t)std::max(1, nl) * kLmRec); V.z = V.Linv ? V.Linv + 6 : nullptr;
    V.n_cost_part = (nv + 255) / 256;
    V.cost_part = c->alloc<double>(V.n_cost_part);
    V.cost_part_cand = c->alloc<double>(V.n_cost_part);
    V.mcc_part = c->alloc<double>(std::max(1, backsub_mcc_groups(V)));
    if (!V.J || !V.CR || !V.r) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (visual tables)");
    // (bsgpu_internal.h Visual::no_cr: every factor belongs to a band landmark)
    const char* no_cr_env = getenv("BSGPU_NO_CR");
    V.no_cr = !(no_cr_env && atoi(no_cr_env) == 0) && V.n_band_units > 0 && V.n_seg == 0 && V.n_ent == 0 && V.n == V.n_elim && V.band_lm_id != nullptr && V.Linv && V.z;
  }
  lap("visual upload + alloc");
  // ---- pose-only groups
  size_t part_max = std::max<size_t>(V.n_cost_part, 2 * ((size_t)nb + 255) / 256 + 2);
  int* d_toff_asm[kNumInternal] = {nullptr};
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
        if (c->toff[idx[sl]] >= c->n_pose && !(c->is_lm[idx[sl]] == 2 && sl == ti.nvar - 1 && (t == BSGPU_F_IDP_REPROJ || t == BSGPU_F_IDP_REPROJ_UNARY)))
          return fail(c, BSGPU_ERR_UNSUPPORTED, "internal: landmark in a pose-only factor");
        if (!c->is_const[idx[sl]]) active[f] = 1;
      }
      inactive[f] = !active[f];
      if (!active[f]) c->any_inactive = true;
      loss[f] = get_loss(g.loss_kind[f], g.loss_a[f]);
      if (active[f] && !skip_generic_assembly(t)) {
        const int T = (c->n_pose + 63) / 64;
        for (int sa = 0; sa < ti.nvar; ++sa)
          for (int sb = 0; sb < ti.nvar; ++sb) {
            const int ra = c->toff[idx[sa]], rb = c->toff[idx[sb]];
            if (ra < 0 || rb < 0 || ra >= c->n_pose || rb >= c->n_pose) continue;   // (an eliminated inverse-depth slot is not in the reduced system)
            const int wa = c->tsize[idx[sa]], wb = c->tsize[idx[sb]];
            for (int a = ra; a < ra + wa; a += std::max(1, wa - 1)) for (int b = rb; b < rb + wb; b += std::max(1, wb - 1)) c->tile_adj[(size_t)(a / 64) * T + b / 64] = 1;
            if (bg.nbk && sb < sa) bg.add(ra, rb);
          }
      }
    }
    if (timing && g.n > 10000) lap("  group: per-factor loop");
    sg.xoff = c->upload(xoff); sg.toff = c->upload(toff); sg.consts = c->upload(g.consts); sg.loss = c->upload(loss);
    // the assembly into the reduced system sees an eliminated inverse-depth slot as a constant one (the evaluation, the model cost change
    // and bsgpu_evaluate keep the real column)
    d_toff_asm[t] = nullptr;
    if (n_rho > 0 && (t == BSGPU_F_IDP_REPROJ || t == BSGPU_F_IDP_REPROJ_UNARY)) {
      std::vector<int> tm(toff);
      bool any = false;
      for (int& v : tm) if (v >= c->n_pose) { v = -1; any = true; }
      if (any) d_toff_asm[t] = c->upload(tm);
    }
    sg.active = c->upload(active);
    if (has_camera(t)) {
      std::vector<int> camv(g.n);
      for (int f = 0; f < g.n; ++f) camv[f] = g.idx[(size_t)f * ti.nidx + ti.nvar];
      sg.cam = c->upload(camv);
    }
    c->d_small_inactive[t] = c->upload(inactive);
    c->h_small_active[t] = active;
    if (timing && g.n > 10000) lap("  group: uploads");
    sg.r = c->alloc<double>((size_t)g.n * ti.m);
    sg.J = c->alloc<double>((size_t)g.n * ti.m * 3 * ti.nvar);
    c->d_small_part[t] = c->alloc<double>((size_t)g.n * ti.m);
    c->d_small_part_cand[t] = c->alloc<double>(g.n);
    c->d_small_part_mcc[t] = c->alloc<double>((size_t)g.n * ti.m);
    part_max = std::max(part_max, (size_t)g.n * ti.m);
    if (timing && g.n > 10000) lap("  group: allocations");
  }
  lap("pose-only group tables");
  // ---- how the pose-only factors are assembled.  Where many factors of a type add into the same 3x3 blocks of J^T J (C3: 20 000
  // relative-pose factors over 100 keyframes and ONE extrinsics variable) per-factor atomics serialise on those addresses; such a type
  // is assembled by SEGMENTS instead: for every block (tangent offsets ra, rb) the (factor, slot a, slot b) contributions to it, in
  // (type, factor) order, cut into chunks of at most kSegChunk that sixteen lanes sum before one atomic add per entry.  A type
  // whose blocks get only a few contributions each (an IMU chain: one or two) keeps the workgroup-per-factor kernel, which reads J once.
  {
    constexpr int kSegChunk = 64;
    struct Contrib { int ra, rb, tf, ss; };
    std::vector<Contrib> cl, tmp;
    std::vector<int> cnt;
    std::vector<AsmGroup> asm_grp;
    size_t asm_nfac = 0;
    auto sort_by_block = [&](size_t lo, size_t hi) {   // cl[lo, hi) by (ra, rb), stable: LSD counting sort, rb then ra
      const size_t n = hi - lo;
      if (n < 2) return;
      tmp.resize(n);
      for (int pass = 0; pass < 2; ++pass) {
        cnt.assign((size_t)c->n_pose + 2, 0);
        for (size_t i = lo; i < hi; ++i) cnt[(pass == 0 ? cl[i].rb : cl[i].ra) + 1]++;
        for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
        for (size_t i = lo; i < hi; ++i) tmp[cnt[pass == 0 ? cl[i].rb : cl[i].ra]++] = cl[i];
        std::copy(tmp.begin(), tmp.begin() + n, cl.begin() + lo);
      }
    };
    for (int t = 0; t < kNumInternal; ++t) {
      c->small_factorwise[t] = c->small[t];
      if (t < 2) c->small_factorwise[t].n = 0;
      if (d_toff_asm[t]) c->small_factorwise[t].toff = d_toff_asm[t];
      if (skip_generic_assembly(t)) c->small_factorwise[t].n = 0;
    }
    for (int t = 2; t < kNumInternal; ++t) {
      const SmallGroup& sg = c->small[t];
      if (!sg.n || skip_generic_assembly(t)) continue;
      const int nv = sg.nv;
      std::vector<int> toffs((size_t)sg.n * nv);
      HIPCHK(c, hipStreamSynchronize(c->stream));   // (the table went up asynchronously on the context's stream)
      HIPCHK(c, hipMemcpy(toffs.data(), sg.toff, sizeof(int) * toffs.size(), hipMemcpyDeviceToHost));
      for (int& v : toffs) if (v >= c->n_pose) v = -1;   // (eliminated inverse-depth slots: k_idp.hip)
      // factors that name the same variable in every slot (the ~21 lidar constraints between two keyframes of a lidar-inertial window):
      // groups of at least kGroupMin, summed a wave per group (k_small.hip: small_assemble_group); the others by segments / by factors
      std::vector<uint8_t> grouped(sg.n, 0);
      bool any_group = false;
      constexpr int kGroupMin = 4;
      if (sg.w_last == 3 && nv * 3 <= 18 && sg.m <= 6 && (sg.m * nv * 3) % 2 == 0 && sg.n >= 64 && !getenv("BSGPU_NO_GROUP_ASSEMBLY")) {
        std::vector<int> order;
        order.reserve(sg.n);
        for (int f = 0; f < sg.n; ++f) if (c->h_small_active[t][f]) order.push_back(f);
        auto key_less = [&](int a, int b) {
          const int* ka = &toffs[(size_t)a * nv]; const int* kb = &toffs[(size_t)b * nv];
          for (int i = 0; i < nv; ++i) if (ka[i] != kb[i]) return ka[i] < kb[i];
          return a < b;
        };
        std::sort(order.begin(), order.end(), key_less);
        // (a slot that holds the same variable in every factor — the extrinsics of C3 — takes one add per group on its 36 + 12 addresses:
        //  measured the same as leaving those entries to the segments, 25.7 against 25.8 us)
        for (size_t i = 0; i < order.size();) {
          size_t j = i + 1;
          while (j < order.size() && std::equal(&toffs[(size_t)order[i] * nv], &toffs[(size_t)order[i] * nv] + nv, &toffs[(size_t)order[j] * nv])) ++j;
          if ((int)(j - i) >= kGroupMin) {
            // (one pass through LDS per workgroup: a large group is cut, into pieces of equal size)
            const size_t n_in = j - i, pieces = (n_in + kAsmGroupMax - 1) / kAsmGroupMax;
            size_t q0 = i;
            for (size_t pc = 0; pc < pieces; ++pc) {
              AsmGroup rec{};
              rec.type = t; rec.count = (int)(n_in / pieces + (pc < n_in % pieces ? 1 : 0)); rec.m = sg.m; rec.nv = nv;
              rec.te = 0;
              for (int sl = 0; sl < 6; ++sl) {
                rec.toff[sl] = sl < nv ? toffs[(size_t)order[i] * nv + sl] : -1;
                if (rec.toff[sl] >= 0) rec.te = 3 * (sl + 1);
              }
              rec.J = sg.J; rec.r = sg.r;
              for (int q = 0; q < kAsmGroupMax; ++q) rec.fac[q] = order[q0 + std::min(q, rec.count - 1)];
              asm_grp.push_back(rec);
              q0 += rec.count;
            }
            for (size_t q = i; q < j; ++q) grouped[order[q]] = 1;
            asm_nfac += j - i;
            any_group = true;
          }
          i = j;
        }
      }
      const size_t first = cl.size();
      for (int f = 0; f < sg.n; ++f) {
        if (!c->h_small_active[t][f] || grouped[f]) continue;
        for (int sa = 0; sa < nv; ++sa) for (int sb = 0; sb < nv; ++sb) {
          const int ra = toffs[(size_t)f * nv + sa], rb = toffs[(size_t)f * nv + sb];
          if (ra < 0 || rb < 0) continue;
          if (ra < rb) continue;   // (the block above the diagonal is the transpose of the one below: the kernel writes both, §3 of DESIGN.md)
          cl.push_back({ra, rb, (t << 24) | f, (sa << 8) | sb});
        }
      }
      // (stable counting sorts on the two tangent offsets, not a comparison sort: an inverse-depth window brings 2 M contributions)
      sort_by_block(first, cl.size());
      size_t n_blocks = 0;
      for (size_t i = first; i < cl.size(); ++i) if (i == first || cl[i].ra != cl[i - 1].ra || cl[i].rb != cl[i - 1].rb) ++n_blocks;
      if (any_group || (n_blocks && (cl.size() - first) >= 2 * n_blocks)) c->small_factorwise[t].n = 0;   // by segments (lower blocks only: half the contributions of the full pattern; a type with groups: its other factors too)
      else cl.resize(first);                                                                // by factors
    }
    sort_by_block(0, cl.size());
    std::vector<int> seg_start, seg_ra, seg_rb;
    std::vector<int2> contrib(cl.size());
    for (size_t i = 0; i < cl.size(); ++i) {
      if (i == 0 || cl[i].ra != cl[i - 1].ra || cl[i].rb != cl[i - 1].rb || (int)i - seg_start.back() >= kSegChunk) {
        seg_start.push_back((int)i); seg_ra.push_back(cl[i].ra); seg_rb.push_back(cl[i].rb);
      }
      contrib[i] = make_int2(cl[i].tf, cl[i].ss);
    }
    seg_start.push_back((int)cl.size());
    c->n_sa_seg = (int)seg_ra.size();
    c->d_sa_seg_start = c->upload(seg_start); c->d_sa_seg_ra = c->upload(seg_ra); c->d_sa_seg_rb = c->upload(seg_rb);
    c->d_sa_contrib = c->upload(contrib);
    c->n_asm_grp = (int)asm_grp.size();
    if (timing && !asm_grp.empty()) {
      int mx = 0;
      for (const AsmGroup& gq : asm_grp) mx = std::max(mx, gq.count);
      fprintf(stderr, "[bsgpu finalize] same-slot groups: %d groups, %d factors, largest %d\n", (int)asm_grp.size(), (int)asm_nfac, mx);
    }
    c->d_asm_grp = c->upload(asm_grp);
    if (timing) lap("  lists: contributions");
    std::vector<SmallGroup> groups(c->small, c->small + kNumInternal);
    c->d_small_groups = c->upload(groups);
  }
  lap("pose-only assembly lists");
  // ---- inverse-depth landmarks eliminated on the landmark side (k_idp.hip): the binary factors sorted by landmark, the
  // (landmark, camera pose) views, and the pair entries grouped by camera-pose pair — what host_visual() above builds for the
  // Euclidean landmarks, one level up: a view stands for all factors of a landmark that involve that camera pose
  c->idp = IdpElim();
  if (n_rho > 0) {
    const HostGroup& g = c->groups[BSGPU_F_IDP_REPROJ];
    const TypeInfo& ti = kTypes[BSGPU_F_IDP_REPROJ];
    IdpElim& E = c->idp;
    E.n_lm = n_rho; E.to0 = idp_to0;
    std::vector<int> lm_start(n_rho + 1, 0), order;
    for (int f = 0; f < g.n; ++f) { const int l = idp_index[g.idx[(size_t)f * ti.nidx + 4]]; if (l >= 0) lm_start[l + 1]++; }
    for (int l = 0; l < n_rho; ++l) lm_start[l + 1] += lm_start[l];
    order.resize(lm_start[n_rho]);
    {
      std::vector<int> pos(lm_start.begin(), lm_start.end() - 1);
      for (int f = 0; f < g.n; ++f) { const int l = idp_index[g.idx[(size_t)f * ti.nidx + 4]]; if (l >= 0) order[pos[l]++] = f; }
    }
    E.n_fac = (int)order.size();
    // camera poses = distinct (q block, p block) pairs of either side, numbered in ascending (q, p) order
    std::vector<uint64_t> cp_keys;
    cp_keys.reserve(64);
    {
      std::vector<int> seen_p(nb, -1);
      for (int f : order)
        for (int side = 0; side < 2; ++side) {
          const int bq = g.idx[(size_t)f * ti.nidx + 2 * side], bp = g.idx[(size_t)f * ti.nidx + 2 * side + 1];
          if (seen_p[bq] != bp) { seen_p[bq] = bp; cp_keys.push_back(((uint64_t)bq << 32) | (uint32_t)bp); }
        }
      std::sort(cp_keys.begin(), cp_keys.end());
      cp_keys.erase(std::unique(cp_keys.begin(), cp_keys.end()), cp_keys.end());
    }
    const int k = (int)cp_keys.size();
    std::vector<int> cp_tq(k), cp_tp(k), cp_first(nb, -1);
    for (int i = 0; i < k; ++i) {
      const int bq = (int)(cp_keys[i] >> 32), bp = (int)(cp_keys[i] & 0xffffffffu);
      cp_tq[i] = c->toff[bq]; cp_tp[i] = c->toff[bp];
      if (cp_first[bq] < 0) cp_first[bq] = i;
    }
    auto cp_of = [&](int bq, int bp) { int i = cp_first[bq]; while ((int)(cp_keys[i] & 0xffffffffu) != bp) ++i; return i; };
    E.n_cam_pose = k;
    // views of every landmark (a landmark has a handful: linear search), and the factors' two views
    std::vector<int> view_start(n_rho + 1, 0), view_cp;
    std::vector<int2> fview(order.size());
    view_cp.reserve(order.size() + n_rho);
    for (int l = 0; l < n_rho; ++l) {
      const int v0 = (int)view_cp.size();
      for (int p = lm_start[l]; p < lm_start[l + 1]; ++p) {
        const int f = order[p];
        int vv[2];
        for (int side = 0; side < 2; ++side) {
          const int cp = cp_of(g.idx[(size_t)f * ti.nidx + 2 * side], g.idx[(size_t)f * ti.nidx + 2 * side + 1]);
          int v = v0;
          while (v < (int)view_cp.size() && view_cp[v] != cp) ++v;
          if (v == (int)view_cp.size()) view_cp.push_back(cp);
          vv[side] = v;
        }
        fview[p] = make_int2(vv[0], vv[1]);
      }
      view_start[l + 1] = (int)view_cp.size();
    }
    E.n_view = (int)view_cp.size();
    // pair entries (view a, view b, code) of every landmark with cp(a) <= cp(b), grouped by camera-pose pair, landmark-major inside a
    // group.  code (idp_pairs_kernel): the factor whose two poses are that view pair — its cross term A_a^T A_m rides on the entry —
    // or -1; a further factor on the same view pair gets an entry of its own without the Schur term.
    E.direct = idp_direct ? 1 : 0;
    std::vector<int> view_lm(view_cp.size());
    for (int l = 0; l < n_rho; ++l) for (int v = view_start[l]; v < view_start[l + 1]; ++v) view_lm[v] = l;
    std::vector<int> first_code;
    struct Extra { int a, b, code; };
    std::vector<Extra> extras;
    auto for_entries = [&](int l, auto&& fn) {   // the entries of landmark l, in a fixed order
      const int v0 = view_start[l], nv = view_start[l + 1] - v0;
      first_code.assign((size_t)nv * nv, -1);
      extras.clear();
      if (E.direct)
        for (int p = lm_start[l]; p < lm_start[l + 1]; ++p) {
          const int x = fview[p].x, y = fview[p].y;
          if (x == y) continue;   // (both sides on one camera pose: in that view's own block)
          const bool swap = view_cp[x] > view_cp[y];
          const int a = swap ? y : x, b = swap ? x : y;
          const int code = (p << 2) | (swap ? 2 : 0);
          int& fc = first_code[(size_t)(a - v0) * nv + (b - v0)];
          if (fc < 0) fc = code; else extras.push_back({a, b, code | 1});
        }
      for (int a = v0; a < v0 + nv; ++a)
        for (int b = v0; b < v0 + nv; ++b)
          if (view_cp[a] <= view_cp[b] && (a == b || view_cp[a] != view_cp[b] || a < b)) fn(a, b, first_code[(size_t)(a - v0) * nv + (b - v0)]);
      for (const Extra& x : extras) fn(x.a, x.b, x.code);
    };
    const uint64_t ncp = (uint64_t)std::max(1, k);
    constexpr int kIdpChunk = 128;   // entries per segment (one wave of idp_pairs_kernel; 64 .. 256 measured alike, 32 half as fast)
    std::vector<int> seg_ci, seg_cj, seg_start, ent_va, ent_vb, ent_code;
    if (ncp * ncp <= (uint64_t)8 << 20 && !sort_entries) {
      std::vector<int> start(ncp * ncp + 1, 0);
      for (int l = 0; l < n_rho; ++l) for_entries(l, [&](int a, int b, int) { start[(uint64_t)view_cp[a] * ncp + view_cp[b] + 1]++; });
      for (size_t i = 0; i < ncp * ncp; ++i) start[i + 1] += start[i];
      ent_va.resize(start[ncp * ncp]); ent_vb.resize(start[ncp * ncp]); ent_code.resize(start[ncp * ncp]);
      for (uint64_t key = 0; key < ncp * ncp; ++key)
        for (int p0 = start[key]; p0 < start[key + 1]; p0 += kIdpChunk) { seg_ci.push_back((int)(key / ncp)); seg_cj.push_back((int)(key % ncp)); seg_start.push_back(p0); }
      std::vector<int> pos(start.begin(), start.end() - 1);
      for (int l = 0; l < n_rho; ++l)
        for_entries(l, [&](int a, int b, int code) { const int p = pos[(uint64_t)view_cp[a] * ncp + view_cp[b]]++; ent_va[p] = a; ent_vb[p] = b; ent_code[p] = code; });
    } else {
      struct Ent { uint64_t key; int va, vb, code; };
      std::vector<Ent> ents;
      for (int l = 0; l < n_rho; ++l) for_entries(l, [&](int a, int b, int code) { ents.push_back({(uint64_t)view_cp[a] * ncp + view_cp[b], a, b, code}); });
      std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.key < y.key; });
      ent_va.resize(ents.size()); ent_vb.resize(ents.size()); ent_code.resize(ents.size());
      for (size_t i = 0; i < ents.size(); ++i) {
        if (i == 0 || ents[i].key != ents[i - 1].key || (int)i - seg_start.back() >= kIdpChunk) {
          seg_ci.push_back((int)(ents[i].key / ncp)); seg_cj.push_back((int)(ents[i].key % ncp)); seg_start.push_back((int)i);
        }
        ent_va[i] = ents[i].va; ent_vb[i] = ents[i].vb; ent_code[i] = ents[i].code;
      }
    }
    seg_start.push_back((int)ent_va.size());
    E.n_seg = (int)seg_ci.size(); E.n_ent = (int)ent_va.size();
    // the fill of the elimination: every pair of camera poses that share a landmark
    {
      const int T = (c->n_pose + 63) / 64;
      if (c->tile_adj.size() != (size_t)T * T) c->tile_adj.assign((size_t)T * T, 0);
      auto touch = [&](int ra, int rb) {
        if (ra < 0 || rb < 0) return;
        for (int a = ra; a < ra + 3; a += 2) for (int b = rb; b < rb + 3; b += 2) {
          c->tile_adj[(size_t)(a / 64) * T + b / 64] = 1; c->tile_adj[(size_t)(b / 64) * T + a / 64] = 1;
        }
        if (bg.nbk) bg.add(ra, rb);
      };
      for (int sgi = 0; sgi < E.n_seg; ++sgi) {
        const int i = seg_ci[sgi], j = seg_cj[sgi];
        const int ri[2] = {cp_tq[i], cp_tp[i]}, rj[2] = {cp_tq[j], cp_tp[j]};
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) touch(ri[a], rj[b]);
      }
    }
    E.order = c->upload(order); E.lm_start = c->upload(lm_start); E.view_start = c->upload(view_start); E.fview = c->upload(fview);
    E.cp_tq = c->upload(cp_tq); E.cp_tp = c->upload(cp_tp);
    E.seg_ci = c->upload(seg_ci); E.seg_cj = c->upload(seg_cj); E.seg_start = c->upload(seg_start);
    E.ent_va = c->upload(ent_va); E.ent_vb = c->upload(ent_vb); E.ent_code = c->upload(ent_code); E.view_lm = c->upload(view_lm);
    E.view_cp = c->upload(view_cp);
    if (E.direct) {
      // a view of exactly one factor side (every measurement view) needs no stored D: the pair kernel forms it from the factor's row
      std::vector<int> cnt(view_cp.size(), 0), view_code(view_cp.size(), 0);
      for (size_t p = 0; p < order.size(); ++p) {
        const int x = fview[p].x, y = fview[p].y;
        if (x == y) { cnt[x] += 2; continue; }
        if (cnt[x]++ == 0) view_code[x] = (int)(p << 1); 
        if (cnt[y]++ == 0) view_code[y] = (int)(p << 1) | 1;
      }
      int n_multi = 0;
      for (size_t v = 0; v < view_cp.size(); ++v) if (cnt[v] != 1) view_code[v] = -(++n_multi);
      E.view_code = c->upload(view_code);
      E.VD = c->alloc<double>((size_t)std::max(1, n_multi) * 48);
      if (!E.VD) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (inverse-depth landmark tables)");
    }
    E.U = c->alloc<double>((size_t)std::max(1, E.n_view) * 8); E.linv = c->alloc<double>(n_rho); E.z = c->alloc<double>(n_rho);
    E.C = c->alloc<double>((size_t)std::max(1, E.n_fac) * 2);
    if (!E.U || !E.linv || !E.z || !E.C) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (inverse-depth landmark tables)");
    lap("inverse-depth landmark tables");
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
      if (mc.active) {   // a dense prior couples every pair of its blocks
        for (int ta : col_t) for (int tb : col_t) if (ta >= 0 && tb >= 0) c->tile_adj[(size_t)(ta / 64) * T + tb / 64] = 1;
        if (bg.nbk) for (size_t i = 0; i < mg.blocks.size(); ++i) for (size_t j = 0; j < i; ++j) {
          const int bi = mg.blocks[i], bj = mg.blocks[j];
          if (!c->is_const[bi] && !c->is_const[bj]) bg.add(c->toff[bi], c->toff[bj]);
        }
      }
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
  // ---- tiled Cholesky plan: nested-dissection tile order, symbolic factorisation, step schedule
  {
    const char* e = getenv("BSGPU_CHAINS");
    const int max_chains = e ? std::max(1, atoi(e)) : 16;
    {
      const int T0 = (c->n_pose + 63) / 64;
      if (c->tile_adj.size() != (size_t)T0 * T0) c->tile_adj.assign((size_t)T0 * T0, 0);
    }
    const char* e3 = getenv("BSGPU_SHARED");   // panels of one step may update the same tiles (atomics): on unless BSGPU_SHARED=0
    { const char* ex = getenv("BSGPU_CHOL_EXT"); c->plan.allow_ext = !(ex && atoi(ex) == 0); }
    { const char* sp = getenv("BSGPU_CHOL_SPLIT"); if (sp) c->plan.split_depth = std::max(0, std::min(64, atoi(sp))); }   // (dense_plan.h kFusedSplit; default 2)
    // (the LM diagonal and the gradient norms as tasks of the factorisation's launch; BSGPU_POSE_DIAG_LAUNCH=1: their own launch, as before)
    c->plan.diag_tasks = getenv("BSGPU_POSE_DIAG_LAUNCH") == nullptr;
    c->plan.rider_tasks = c->plan.diag_tasks ? (c->nb + 255) / 256 : 0;   // (0: every tile's panel has its own update tasks, also a separator's appendix tile)
    const bool use_leaf = c->n_leaf_tiles > 0 && !getenv("BSGPU_NO_LEAF_TILES");
    bool ordered = false;
    lap("blocks");
    if (bg.nbk && max_chains > 1) {
      // per-dimension order (dim_order.h): nested dissection of the block graph, separators = sets of tangent blocks
      DimOrder ord;
      ord.n_pose = c->n_pose; ord.blk_t0 = bg.t0; ord.blk_w = bg.w;
      ord.adj_ptr.assign(bg.nbk + 1, 0);
      for (int a = 0; a < bg.nbk; ++a) {
        const uint64_t* rowb = &bg.bits[(size_t)a * bg.words];
        for (int wd = 0; wd < bg.words; ++wd) {
          uint64_t m = rowb[wd];
          while (m) { const int b = wd * 64 + __builtin_ctzll(m); m &= m - 1; if (b != a) ord.adj.push_back(b); }
        }
        ord.adj_ptr[a + 1] = (int)ord.adj.size();
      }
      if (const char* ev = getenv("BSGPU_DIM_ORDER_DEPTH")) ord.max_depth = std::max(0, atoi(ev));
      if (const char* ev = getenv("BSGPU_DIM_ABSORB")) ord.absorb = atoi(ev) != 0;   // (0: no separator joins its parent — dim_order.h absorb_separators())
      if (const char* ev = getenv("BSGPU_DIM_T_STEP3")) ord.t_step3 = atof(ev);
      if (const char* ev = getenv("BSGPU_DIM_MERGE")) ord.merge_dims = atoi(ev);
      lap("  order: adjacency lists");
      // The dissection's cost model (chain start, hand-over, a second full tile's hand-over) is a heuristic and no one setting is best at every
      // size: (5, 10, 4) gives C2 and C3 their shortest factorisations, (2, 8, 8) / (3, 9, 8) cut a window of the reference's size into seven
      // supernodes instead of five and are 4 % faster there and 3 - 5 % slower on C2 / C3 (scripts/ab_dim_model.sh).  What ranks the candidates
      // right every time is the ticket order's own replay of the finished task list (DensePlan::est_makespan_us): small systems — where planning is
      // tens of microseconds — are planned under each setting and keep the shortest replay.  BSGPU_DIM_T_CHAIN0 / _T_HOP / _T_HOP_TILE: one setting.
      struct OrdModel { double chain0, hop, hop_tile, step3; int merge, depth; };
      std::vector<OrdModel> models = {{ord.t_chain0, ord.t_hop, ord.t_hop_tile, ord.t_step3, ord.merge_dims, ord.max_depth}};
      {
        const char* e0 = getenv("BSGPU_DIM_T_CHAIN0"); const char* e1 = getenv("BSGPU_DIM_T_HOP"); const char* e2 = getenv("BSGPU_DIM_T_HOP_TILE");
        static const bool cand_off = getenv("BSGPU_DIM_CANDIDATES") && atoi(getenv("BSGPU_DIM_CANDIDATES")) == 0;
        if (e0 || e1 || e2 || getenv("BSGPU_DIM_T_STEP3") || getenv("BSGPU_DIM_MERGE"))
          models[0] = {e0 ? atof(e0) : ord.t_chain0, e1 ? atof(e1) : ord.t_hop, e2 ? atof(e2) : ord.t_hop_tile, ord.t_step3, ord.merge_dims, ord.max_depth};
        else if (!cand_off && !dense_on_trial && c->n_pose <= 2000 && c->plan_pref == BSGPU_PLAN_LATENCY) {
          // ((2, 10, 12; three-tile steps at 3.0, separators up to 40 dimensions joining their parents): what an offline search of 1 440 settings by the
          //  replay found for C3 — 105.2 -> 102.8 us replayed, 97 -> 91.5 us measured, C3 5 875 -> 6 075 LM it/s)
          // ((5, 40, 0; 3.0, 40; two levels deeper): the same search on pose graphs of 200 poses — 400 loop closures 256 -> 228 us replayed, 2 590 -> 3 090 LM it/s; 300: 240 -> 208 us,
          //  18 % fewer tasks, 3 156 -> 3 535 LM it/s)
          models.push_back({2.0, 8.0, 8.0, ord.t_step3, ord.merge_dims, ord.max_depth}); models.push_back({3.0, 9.0, 8.0, ord.t_step3, ord.merge_dims, ord.max_depth});
          models.push_back({2.0, 10.0, 12.0, 3.0, 40, ord.max_depth}); models.push_back({5.0, 40.0, 0.0, 3.0, 40, ord.max_depth + 2});
        }
      }
      const bool compare = models.size() > 1;
      if (compare && c->dim_model >= 0 && c->dim_model < (int)models.size() && c->dim_model_npose == c->n_pose && c->dim_model_age < 32) {
        const OrdModel w = models[c->dim_model];   // (the last comparison's winner, alone)
        models.assign(1, w);
        ++c->dim_model_age;
      }
      const DimOrder ord_base = ord;          // (inputs and settings; build() fills the rest)
      DensePlan plan_base;
      if (models.size() > 1) plan_base = c->plan;   // (its settings: allow_ext, split_depth, diag_tasks ...)
      bool keep = true;
      double best_span = 1e300;
      DensePlan best_plan;
      DimOrder best_ord;
      for (size_t mi = 0; mi < models.size(); ++mi) {
      if (mi > 0 || models.size() > 1) { ord = ord_base; }
      ord.t_chain0 = models[mi].chain0; ord.t_hop = models[mi].hop; ord.t_hop_tile = models[mi].hop_tile; ord.t_step3 = models[mi].step3; ord.merge_dims = models[mi].merge; ord.max_depth = models[mi].depth;
      ord.build();
      if (mi == 0) lap("  order: dissection");
      // tile adjacency in S order from the block graph (a block lies in at most two tiles of its supernode)
      const int To = ord.T;
      std::vector<uint8_t> adjS((size_t)To * To, 0);
      auto mark = [&](int a, int b) {
        const int a0 = ord.dpos[bg.t0[a]] >> 6, a1 = ord.dpos[bg.t0[a] + bg.w[a] - 1] >> 6, b0 = ord.dpos[bg.t0[b]] >> 6, b1 = ord.dpos[bg.t0[b] + bg.w[b] - 1] >> 6;
        for (int x = a0; x <= a1; ++x) for (int y = b0; y <= b1; ++y) { adjS[(size_t)x * To + y] = 1; adjS[(size_t)y * To + x] = 1; }
      };
      for (int a = 0; a < bg.nbk; ++a) {
        mark(a, a);
        for (int e4 = ord.adj_ptr[a]; e4 < ord.adj_ptr[a + 1]; ++e4) if (ord.adj[e4] < a) mark(a, ord.adj[e4]);
      }
      if (models.size() > 1) {   // (small systems, never on trial: plan, replay, keep the shortest)
        DensePlan cand = plan_base;
        cand.build_ordered(c->n_pose, To, ord.dpos, ord.nreal, adjS, ord.piece_ranges, ord.sep_ranges_by_level, !(e3 && atoi(e3) == 0));
        if (timing) fprintf(stderr, "[bsgpu finalize]   cost model (%.1f, %.1f, %.1f; %.1f, %d, %d): %d supernodes, depth %d, task list replayed %.1f us\n", models[mi].chain0, models[mi].hop,
                            models[mi].hop_tile, models[mi].step3, models[mi].merge, models[mi].depth, ord.n_nodes, ord.depth, cand.est_makespan_us);
        if (cand.est_makespan_us < best_span) { best_span = cand.est_makespan_us; best_plan = std::move(cand); best_ord = ord; c->dim_model = (int)mi; c->dim_model_age = 0; c->dim_model_npose = c->n_pose; }
        if (mi + 1 == models.size()) { c->plan = std::move(best_plan); ord = best_ord; ordered = true; }
        continue;
      }
      if (dense_on_trial && ord.depth == 0) keep = false;   // no separator found (C4: uniformly random loop closures): the system fills in
      if (keep && dense_on_trial) {
        // ... and an order that does find separators may still fill in: count the update tasks of the tile-level symbolic factorisation (bit
        // rows: a few milliseconds at 469 tiles) BEFORE the plan — its task list, ticket order and back-substitution tables — is built for a
        // factorisation that will be refused anyway (a filled-in plan of C4's size is 17 million tasks and seconds of finalize)
        const int W = (To + 63) / 64;
        std::vector<uint64_t> rowbits((size_t)To * W, 0);   // rowbits[k]: the row tiles t > k of panel k
        for (int k = 0; k < To; ++k) for (int t = k + 1; t < To; ++t) if (adjS[(size_t)t * To + k]) rowbits[(size_t)k * W + (t >> 6)] |= 1ull << (t & 63);
        double tasks = 0.0;
        for (int k = 0; k < To && tasks < 4.0e7; ++k) {
          const uint64_t* rk = &rowbits[(size_t)k * W];
          int n = 0;
          for (int wd = 0; wd < W; ++wd) n += __builtin_popcountll(rk[wd]);
          tasks += 0.5 * n * (n + 1);
          for (int wd = 0; wd < W; ++wd) {   // fill: every row tile a of panel k gets the later row tiles of panel k
            uint64_t m = rk[wd];
            while (m) {
              const int a = wd * 64 + __builtin_ctzll(m); m &= m - 1;
              uint64_t* ra = &rowbits[(size_t)a * W];
              for (int w2 = a >> 6; w2 < W; ++w2) { uint64_t add = rk[w2]; if (w2 == (a >> 6)) add &= ~((2ull << (a & 63)) - 1); ra[w2] |= add; }
            }
          }
        }
        const double est_flops = tasks * 2.0 * 64.0 * 64.0 * 64.0;
        if (est_flops / 7.0e12 * 1e6 > 4000.0) {
          keep = false;
          if (timing) fprintf(stderr, "[bsgpu finalize] pose graph above the dense limit on trial: ~%.3g update tasks after fill (%.3g flops): block-sparse PCG, no plan built\n", tasks, est_flops);
        }
      }
      if (keep) {
        c->plan.build_ordered(c->n_pose, To, ord.dpos, ord.nreal, adjS, ord.piece_ranges, ord.sep_ranges_by_level, !(e3 && atoi(e3) == 0));
        ordered = true;
        // (the exact step must beat a PCG solve of a few milliseconds: the critical path of the chains, and the flops at the ~3 TFLOP/s the
        // update tasks of a large plan sustain)
        if (dense_on_trial) {
          const double est_us = std::max(ord.est_path_us, c->plan.fused_flops / 7.0e12 * 1e6);   // (7 TFLOP/s: what the thousands of independent update tasks of such a plan sustain — measured on synthetic.pose_graph_local)
          if (timing) fprintf(stderr, "[bsgpu finalize] pose graph above the dense limit on trial: path %.0f us, %.3g flops -> %.0f us estimated per factorisation: %s\n",
                              ord.est_path_us, c->plan.fused_flops, est_us, est_us > 4000.0 ? "block-sparse PCG" : "exact tiled factorisation");
          if (est_us > 4000.0) { keep = false; ordered = false; }
        }
      }
      }   // (cost models)
      if (!keep) c->dense_ok = false;
      lap("  order: tile plan");
      if (timing) fprintf(stderr, "[bsgpu finalize] per-dimension order: %d blocks, %d supernodes, depth %d, estimated path %.0f us\n", bg.nbk, ord.n_nodes, ord.depth, ord.est_path_us);
    }
    if (dense_on_trial && !ordered) c->dense_ok = false;
    if (!c->dense_ok) c->plan.build_skeleton(c->n_pose);
    else if (!ordered)
      c->plan.build(c->n_pose, c->tile_adj, c->dense_ok ? max_chains : 1, 1, !(e3 && atoi(e3) == 0), use_leaf ? &c->leaf_tile : nullptr);
    c->npad = c->plan.npad;
    const int T = c->plan.T;
    if (timing) fprintf(stderr, "[bsgpu finalize] task list replayed: %.1f us\n", c->plan.est_makespan_us);
    if (timing) fprintf(stderr, "[bsgpu finalize] Cholesky plan: %d tiles (%d leaf), %d pieces, %d panel steps, %d back-substitution launches\n", c->plan.T,
                        c->plan.n_leaf_tiles, c->plan.n_pieces, c->plan.n_steps(), (int)c->plan.bs_group_off.size() - 1);
    c->d_dpos = c->upload(c->plan.dpos); c->d_inat = c->upload(c->plan.inat); c->d_nreal = c->upload(c->plan.nreal);
    c->d_rows_flat = c->upload(c->plan.rows_flat);
    c->d_panels = c->upload(c->plan.panels);
    c->d_bs_desc = c->upload(c->plan.bs_desc);
    c->d_bs_desc_chain = c->d_rows_flat_chain = c->d_bs_upd = c->d_bs_upd_rows = nullptr;
    c->d_Winv = nullptr;
    c->d_bs_chain_group = c->d_bs_grp_nchains = c->d_bs_grp_nitems = c->d_bs_items4 = c->d_bs_tile_updated = c->d_bs_sync = c->d_bs_order = nullptr;
    if (c->plan.bs_level_sync && !getenv("BSGPU_BACKSOLVE_LEGACY")) {
      c->d_bs_desc_chain = c->upload(c->plan.bs_desc_chain); c->d_rows_flat_chain = c->upload(c->plan.rows_flat_chain);
      c->d_bs_upd = c->upload(c->plan.bs_upd); c->d_bs_upd_rows = c->upload(c->plan.bs_upd_rows);
      c->d_bs_chain_group = c->upload(c->plan.bs_chain_group); c->d_bs_grp_nchains = c->upload(c->plan.bs_grp_nchains);
      c->d_bs_grp_nitems = c->upload(c->plan.bs_grp_nitems); c->d_bs_items4 = c->upload(c->plan.bs_items4);
      c->d_bs_tile_updated = c->upload(c->plan.bs_tile_updated);
      c->d_bs_order = c->upload(c->plan.bs_order);
      c->d_bs_sync = c->upload(std::vector<int>(16 * (2 + c->plan.bs_grp_nchains.size() + c->plan.bs_items4.size() / 4 + 8), 0));   // (words a 64-byte line apart)
    }
    c->d_tile_sync = c->upload(c->plan.tile_sync);
    {
      // fused single-launch factorisation (default; BSGPU_CHOL_FUSED=0 keeps the launch-per-step path): task list + zeroed counters
      const char* ef = getenv("BSGPU_CHOL_FUSED");
      c->d_ftasks = nullptr; c->d_fsync = nullptr; c->d_tile_tot = nullptr;
      c->d_ftasks_plain = nullptr; c->d_tile_tot_plain = nullptr; c->n_ftasks_plain = 0;
      c->d_ftasks_bulk = nullptr; c->d_tile_tot_bulk = nullptr; c->n_ftasks_bulk = 0;
      c->d_ftasks_rows = nullptr; c->n_ftasks_rows = 0;
      // (one workgroup of 512 threads per task, and a grid holds fewer than 2^32 threads: above 8.38 M tasks — a DENSE system of more than
      // ~23 600 dimensions, which only the exact option on a pose graph produces — the launch is refused by the runtime, so the
      // launch-per-step path runs there; scripts/c4_exact.py)
      if (!(ef && atoi(ef) == 0) && !c->plan.ftasks.empty() && c->plan.ftasks.size() * 512 < ((size_t)1 << 32)) {
        c->d_ftasks = c->upload(c->plan.ftasks);
        c->d_tile_tot = c->upload(c->plan.tile_tot);
        if (!c->plan.ftasks_plain.empty()) {
          c->d_ftasks_plain = c->upload(c->plan.ftasks_plain); c->d_tile_tot_plain = c->upload(c->plan.tile_tot_plain);
          c->n_ftasks_plain = (int)c->plan.ftasks_plain.size();
        }
        if (!c->plan.ftasks_bulk.empty()) {
          c->d_ftasks_bulk = c->upload(c->plan.ftasks_bulk); c->d_tile_tot_bulk = c->upload(c->plan.tile_tot_bulk);
          c->n_ftasks_bulk = (int)c->plan.ftasks_bulk.size();
        }
        if (c->plan.frows_src >= 0 && !c->plan.frow_items.empty()) {   // the list, then the segments' pairs (k_chol.hip chol_fused_rowseg finds them behind the list)
          std::vector<FusedTask> img = c->plan.ftasks_rows;
          const size_t n_pairs = c->plan.frow_items.size() / 2, per = sizeof(FusedTask) / (2 * sizeof(int));
          img.resize(img.size() + (n_pairs + per - 1) / per);
          std::memcpy(reinterpret_cast<int*>(img.data() + c->plan.ftasks_rows.size()), c->plan.frow_items.data(), sizeof(int) * c->plan.frow_items.size());
          c->d_ftasks_rows = c->upload(img); c->n_ftasks_rows = (int)c->plan.ftasks_rows.size();
        }
        c->d_fsync = c->upload(std::vector<int>((size_t)c->plan.fused_sync_words, 0));
        c->d_Winv = c->alloc<double>((size_t)std::max(1, T) * 4096);   // (the chains write every entry of a tile's inverse, zeros above its diagonal blocks included)
      }
    }
    c->d_touched = c->upload(c->plan.touched_tiles); c->n_touched = (int)c->plan.touched_tiles.size();
    c->d_chain_begin = c->upload(c->plan.chain_begin); c->d_chain_end = c->upload(c->plan.chain_end);
    c->d_Vinv = c->alloc<double>((size_t)std::max(1, T) * chol_vinv_stride());
    c->d_ytan = c->alloc<double>(std::max(1, c->n_pose));
    if (c->dense_ok) {
      c->d_Lp = c->alloc<double>((size_t)c->npad * c->npad);
      if (!c->d_Lp) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (L panels)");
      // (the tasks that carry an appendix tile write its 16 real columns of the factor's row tiles and nothing else of those tiles: the
      //  padding's columns must read zero in the back-substitution, and the buffer comes from the pool with whatever it last held)
      bool any_ext = false;
      for (int v : c->plan.fext_of) any_ext = any_ext || v >= 0;
      if (any_ext && hipMemsetAsync(c->d_Lp, 0, sizeof(double) * (size_t)c->npad * c->npad, c->stream) != hipSuccess)
        return fail(c, BSGPU_ERR_DEVICE, "device error while clearing the factor");
    }
  }
  // ---- dense system + vectors
  if (c->dense_ok) {
    c->d_S = c->alloc<double>((size_t)c->npad * c->npad);
    if (!c->d_S) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (reduced system)");
    // cleared once here (the buffer comes from the pool); a step then clears only the tiles anything writes (plan.touched_tiles)
    launch_zero(c->stream, c->d_S, (int64_t)c->npad * c->npad);
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
  {
    c->n_part_upd = (nb + 255) / 256;
    c->n_upd_blocks = 0; c->d_upd_blocks = nullptr; c->d_lm_xoff = nullptr;
    if (c->vis.n_lm > 0 && !getenv("BSGPU_UPDATE_SEPARATE")) {
      // the candidate update rides in the landmark back-substitution (k_reproj.hip: backsub_mcc_kernel): the eliminated Euclidean landmarks
      // are updated by the lanes that compute their step, every other block by extra workgroups of that launch
      std::vector<int> others, lm_xoff(c->vis.n_lm, 0);
      for (int b = 0; b < nb; ++b) {
        if (!c->is_const[b] && c->is_lm[b] == 1) lm_xoff[lm_index[b]] = c->off[b];
        else others.push_back(b);
      }
      if (!others.empty()) {
        c->n_upd_blocks = (int)others.size();
        c->d_upd_blocks = c->upload(others); c->d_lm_xoff = c->upload(lm_xoff);
        c->n_part_upd = (c->n_upd_blocks + 255) / 256 + (c->vis.n_lm * 8 + 255) / 256;   // update units, then the landmark workgroups
      }
    }
    c->pre_cleared = false;
    c->upd_in_mcc = c->n_upd_blocks == 0 && c->vis.n_lm == 0 && !getenv("BSGPU_UPDATE_SEPARATE");
    if (c->upd_in_mcc) c->n_part_upd = (nb + 127) / 128;
    c->d_part_upd = c->alloc<double>(2 * (size_t)c->n_part_upd + 2);
    std::vector<ReduceEntry> tab;
    if (c->vis.n) {
      tab.push_back({c->vis.cost_part, c->vis.n_cost_part, 1, 0, SC_COST_X});
      tab.push_back({c->vis.cost_part_cand, c->vis.n_cost_part, 1, 0, SC_COST_CAND});
      tab.push_back({c->vis.mcc_part, backsub_mcc_groups(c->vis), 1, 0, SC_MCC});   // one partial per workgroup of backsub_mcc_kernel
    }
    for (int t = 2; t < kNumInternal; ++t) {
      if (!c->small[t].n) continue;
      tab.push_back({c->d_small_part[t], small_cost_parts(c->small[t]), 1, 0, SC_COST_X});
      tab.push_back({c->d_small_part_cand[t], small_cost_parts(c->small[t]), 1, 0, SC_COST_CAND});
      tab.push_back({c->d_small_part_mcc[t], (c->small[t].n * c->small[t].m + 127) / 128, 1, 0, SC_MCC});   // one partial per workgroup of small_mcc_kernel
    }
    for (const auto& mc : c->marg) {
      if (!mc.active) continue;
      tab.push_back({mc.part, mc.dev.rows, 1, 0, SC_COST_X});
      tab.push_back({mc.part_cand, mc.dev.rows, 1, 0, SC_COST_CAND});
      tab.push_back({mc.part_mcc, mc.dev.rows, 1, 0, SC_MCC});
    }
    c->n_gpart = (std::max(c->nb, c->npad) + 255) / 256;
    c->d_gpart = c->alloc<double>(2 * (size_t)c->n_gpart);
    HIPCHK(c, hipMemsetAsync(c->d_gpart, 0, sizeof(double) * 2 * (size_t)c->n_gpart, c->stream));
    tab.push_back({c->d_gpart, c->n_gpart, 2, 0, SC_GRAD_MAX, 1});
    tab.push_back({c->d_gpart, c->n_gpart, 2, 1, SC_GRAD_NORM2});
    tab.push_back({c->d_part_upd, c->n_part_upd, 2, 0, SC_STEP_NORM2});
    tab.push_back({c->d_part_upd, c->n_part_upd, 2, 1, SC_X_NORM2});
    c->n_reduce = (int)tab.size();
    c->d_reduce = c->upload(tab);
    c->h_reduce = tab;
    c->d_reduce_counter = c->alloc<int>(1);
    if (c->d_reduce_counter) HIPCHK(c, hipMemsetAsync(c->d_reduce_counter, 0, sizeof(int), c->stream));
    c->d_dec = c->alloc<double>(2 * (size_t)kDecSlots * kDecStride);
    if (c->d_dec) HIPCHK(c, hipMemsetAsync(c->d_dec, 0, sizeof(double) * 2 * kDecSlots * kDecStride, c->stream));
  }
  lap("plan + reduce table");
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipGetLastError());
  lap("device sync");
  c->finalized = true;
  {
    // process-unique: a context re-created at the address of a destroyed one (the submap-refinement pattern: create N windows, solve, destroy,
    // create N fresh ones) must never match tables cached for the old one (bsgpu_batch.cpp)
    static std::atomic<uint64_t> g_finalize_gen{0};
    c->finalize_gen = ++g_finalize_gen;
  }
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
                                                      : "window too large: the reduced camera system exceeds the 49152 dimensions of the tiled Schur path (3 276 keyframes of 15-d states) and the PCG path covers pose-only problems");
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
  // who writes which block: a block with ONE contribution (factor, slot a, slot b) off the diagonal is stored by that factor's wave
  // (bsr_assemble_kernel); the others — every diagonal block — are summed by segments of at most 64 contributions
  std::vector<int> n_contrib(nblk, 0);
  auto slot_of = [&](int ra, int rb) { return (int)(std::lower_bound(keys.begin(), keys.end(), ((uint64_t)(ra / 3) << 32) | (uint32_t)(rb / 3)) - keys.begin()); };
  for (int t = 2; t < kNumInternal; ++t) {
    const HostGroup& g = c->groups[t];
    const TypeInfo& ti = kTypes[t];
    for (int f = 0; f < g.n; ++f) {
      if (!c->h_small_active[t][f]) continue;
      const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
      for (int sa = 0; sa < ti.nvar; ++sa) for (int sb = 0; sb < ti.nvar; ++sb) {
        const int ra = c->toff[idx[sa]], rb = c->toff[idx[sb]];
        if (ra >= 0 && rb >= 0) n_contrib[slot_of(ra, rb)]++;
      }
    }
  }
  struct SegContrib { int slot, tf, ss; };
  std::vector<SegContrib> sc_list;
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
        const int sl = slot_of(ra, rb);
        if (n_contrib[sl] == 1 && ra != rb) slots[((size_t)f * ti.nvar + sa) * ti.nvar + sb] = sl;
        else sc_list.push_back({sl, (t << 24) | f, (sa << 8) | sb});
      }
    }
    c->d_slots[t] = c->upload(slots);
  }
  {
    std::stable_sort(sc_list.begin(), sc_list.end(), [](const SegContrib& x, const SegContrib& y) { return x.slot < y.slot; });
    std::vector<int> seg_start, seg_slot, seg_row;
    std::vector<int2> contrib(sc_list.size());
    for (size_t i = 0; i < sc_list.size(); ++i) {
      if (i == 0 || sc_list[i].slot != sc_list[i - 1].slot || (int)i - seg_start.back() >= 64) {
        seg_start.push_back((int)i); seg_slot.push_back(sc_list[i].slot);
        const int r = (int)(keys[sc_list[i].slot] >> 32), cc = (int)(keys[sc_list[i].slot] & 0xffffffffu);
        seg_row.push_back(r == cc ? 3 * r : -1);
      }
      contrib[i] = make_int2(sc_list[i].tf, sc_list[i].ss);
    }
    seg_start.push_back((int)sc_list.size());
    c->n_bsr_seg = (int)seg_slot.size();
    c->d_bsr_seg_start = c->upload(seg_start); c->d_bsr_seg_slot = c->upload(seg_slot); c->d_bsr_seg_row = c->upload(seg_row);
    c->d_bsr_contrib = c->upload(contrib);
  }
  c->nbr = nbr; c->nblk = nblk;
  // the preconditioner pairs consecutive block rows (2m, 2m+1) — position and orientation of one pose in the pose-graph layouts —
  // into 6x6 diagonal blocks: slot of the coupling block (2m, 2m+1), -1 if the two rows are not coupled (or 2m+1 does not exist)
  const int npair = (nbr + 1) / 2;
  std::vector<int> pair_slot(npair, -1);
  for (int m = 0; m < npair; ++m) {
    if (2 * m + 1 >= nbr) continue;
    const uint64_t key = ((uint64_t)(2 * m) << 32) | (uint32_t)(2 * m + 1);
    auto it = std::lower_bound(keys.begin(), keys.end(), key);
    if (it != keys.end() && *it == key) pair_slot[m] = (int)(it - keys.begin());
  }
  c->d_row_ptr = c->upload(row_ptr); c->d_col = c->upload(col); c->d_diag_slot = c->upload(diag_slot);
  c->d_pair_slot = c->upload(pair_slot);
  c->d_val = c->alloc<double>((size_t)nblk * 9); c->d_Minv = c->alloc<double>((size_t)npair * 36);
  c->d_rhs = c->alloc<double>(c->n_pose);
  c->d_px = c->alloc<double>(c->n_pose); c->d_pr = c->alloc<double>(c->n_pose); c->d_pz = c->alloc<double>(c->n_pose);
  c->d_pp = c->alloc<double>(c->n_pose); c->d_pp1 = c->alloc<double>(c->n_pose); c->d_pq = c->alloc<double>(c->n_pose);
  c->d_ppart = c->alloc<double>((size_t)pcg_spmv_grid(nbr) + 8); c->d_ppart2 = c->alloc<double>(4 * (size_t)pcg_rows_grid(nbr) + 8);
  c->d_psc = c->alloc<double>(pcg_num_scalars());
  if (!c->d_val || !c->d_pq || !c->d_psc) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (block-sparse system)");
  // ---- the solve as one persistent launch (k_pcg.hip pcg_persistent_kernel): contiguous ranges of block rows per workgroup, cut at
  // even rows (a 6x6 preconditioner block stays in one workgroup) and balanced by non-zero blocks; the columns each range names
  c->pcg_persist = PcgPersistDev();
  if (!getenv("BSGPU_PCG_LAUNCHES") && nbr >= 64) {
    int n_cu = 256;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, c->device) == hipSuccess && pr.multiProcessorCount > 0) n_cu = pr.multiProcessorCount; }
    const int G = std::max(1, std::min({n_cu, 256, (nbr + 1) / 2}));
    std::vector<int> wg_row(G + 1, 0);
    const double total = (double)nblk + 4.0 * nbr;   // (a row costs its blocks plus the update's share)
    int r = 0;
    for (int g = 0; g < G; ++g) {
      wg_row[g] = r;
      const double want = total * (g + 1) / G;
      while (r < nbr && ((double)row_ptr[r] + 4.0 * r < want || (r & 1))) ++r;
      if (g == G - 1) r = nbr;
    }
    wg_row[G] = nbr;
    std::vector<int> wg_colptr(G + 1, 0), wg_cols, lcol(nblk, 0), stamp(nbr, -1), local(nbr, 0);
    int max_cols = 0, max_rows = 0;
    bool paired = true;
    for (int g = 0; g < G; ++g) {
      std::vector<int> cols;
      for (int e = row_ptr[wg_row[g]]; e < row_ptr[wg_row[g + 1]]; ++e) if (stamp[col[e]] != g) { stamp[col[e]] = g; cols.push_back(col[e]); }
      std::sort(cols.begin(), cols.end());
      for (size_t i = 0; i < cols.size(); ++i) local[cols[i]] = (int)i;
      paired = paired && cols.size() % 2 == 0;
      for (size_t i = 0; i + 1 < cols.size() && paired; i += 2) paired = (cols[i] & 1) == 0 && cols[i + 1] == cols[i] + 1;
      for (int e = row_ptr[wg_row[g]]; e < row_ptr[wg_row[g + 1]]; ++e) lcol[e] = local[col[e]];
      wg_cols.insert(wg_cols.end(), cols.begin(), cols.end());
      wg_colptr[g + 1] = (int)wg_cols.size();
      max_cols = std::max(max_cols, (int)cols.size());
      max_rows = std::max(max_rows, 3 * (wg_row[g + 1] - wg_row[g]));
    }
    if (max_rows <= pcg_persistent_max_rows() && pcg_persistent_lds(max_cols) <= pcg_persistent_lds_limit()) {
      PcgPersistDev& P = c->pcg_persist;
      P.G = G; P.max_cols = max_cols; P.paired = paired ? 1 : 0;
      P.wg_row = c->upload(wg_row); P.wg_colptr = c->upload(wg_colptr); P.wg_cols = c->upload(wg_cols); P.lcol = c->upload(lcol);
      const size_t sw = pcg_persistent_slot_words(G), zw = pcg_persistent_z_words(nbr);
      P.slots = c->alloc<unsigned long long>(sw + zw + 2);
      if (P.slots) { P.zg = reinterpret_cast<double*>(P.slots + sw); P.abort_w = reinterpret_cast<int*>(P.slots + sw + zw); P.sync_bytes = sizeof(unsigned long long) * (sw + zw + 2); }
      if (!P.wg_row || !P.lcol || !P.slots) P = PcgPersistDev();
      // ---- the coarse space of the two-level preconditioner: the free poses.  A pose = a 3-vector block next to a quaternion block in
      // a factor's variable list ((p, q) in the pose-graph types, (q, p) in the IMU ones); a pose that a one-pose factor holds (absolute
      // pose, IMU prior) is anchored and stays out.  BSGPU_PCG_COARSE=0: block-Jacobi alone (the cross-check).
      const char* ce = getenv("BSGPU_PCG_COARSE");
      if (P.G > 0 && !(ce && !strcmp(ce, "0"))) {
        std::vector<int> p_of_q(c->nb, -1);
        std::vector<unsigned char> anchored(c->nb, 0);
        for (int t = 2; t < kNumInternal; ++t) {
          const HostGroup& g = c->groups[t];
          const TypeInfo& ti = kTypes[t];
          int pairs[8][2], np = 0;
          for (int sl = 0; sl + 1 < ti.nvar && np < 8; ++sl) {
            if (ti.amb[sl] == 3 && ti.amb[sl + 1] == 4) { pairs[np][0] = sl; pairs[np][1] = sl + 1; ++np; ++sl; }
            else if (ti.amb[sl] == 4 && ti.amb[sl + 1] == 3) { pairs[np][0] = sl + 1; pairs[np][1] = sl; ++np; ++sl; }
          }
          for (int f = 0; f < g.n && np > 0; ++f) {
            if (!c->h_small_active[t][f]) continue;
            const int32_t* idx = &g.idx[(size_t)f * ti.nidx];
            for (int k = 0; k < np; ++k) {
              const int bp = idx[pairs[k][0]], bq = idx[pairs[k][1]];
              if (p_of_q[bq] < 0) p_of_q[bq] = bp;
              if (np == 1) { anchored[bq] = 1; anchored[bp] = 1; }
            }
          }
        }
        std::vector<int4> co;
        double cen[3] = {0.0, 0.0, 0.0};
        for (int bq = 0; bq < c->nb; ++bq) {
          const int bp = p_of_q[bq];
          if (bp < 0 || anchored[bq] || anchored[bp] || c->toff[bq] < 0 || c->toff[bp] < 0) continue;
          co.push_back(make_int4(c->off[bp], c->off[bq], c->toff[bp], c->toff[bq]));
          for (int k = 0; k < 3; ++k) cen[k] += c->h_x[c->off[bp] + k];
        }
        if (co.size() >= 2) {
          for (int k = 0; k < 3; ++k) P.centre[k] = cen[k] / (double)co.size();
          P.co = c->upload(co);
          double* scratch = c->alloc<double>(pcg_coarse_scratch_doubles(nbr));
          if (P.co && scratch && hipMemsetAsync(scratch, 0, sizeof(double) * pcg_coarse_scratch_doubles(nbr), c->stream) == hipSuccess) {
            const size_t nw = (size_t)3 * nbr * 6;
            P.W = scratch; P.E = scratch + nw; P.Einv = scratch + pcg_coarse_scratch_doubles(nbr) - 38;
            P.counter = reinterpret_cast<unsigned*>(P.Einv + 36);
            P.n_coarse = (int)co.size();
          }
        }
      }
    }
  }
  c->bsr_built = true;
  return BSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------
// PCG on the reduced camera system: the tile rows of S (what the assembly writes, without the rhs row / column), built on first use
// ---------------------------------------------------------------------------------------------------
int build_spcg(bsgpu_ctx* c) {
  if (c->spcg_built) return BSGPU_OK;
  if (!c->dense_ok || !c->d_S) return fail(c, BSGPU_ERR_UNSUPPORTED, "BSGPU_LINEAR_SCHUR_PCG works on the assembled reduced camera system: the window exceeds its limit");
  const int T = c->plan.T, N = T + 1;
  std::vector<int> row_ptr(T + 1, 0), col;
  for (int t : c->plan.touched_tiles) { const int i = t / N, j = t % N; if (i < T && j < T) row_ptr[i + 1]++; }
  for (int i = 0; i < T; ++i) row_ptr[i + 1] += row_ptr[i];
  col.resize(row_ptr[T]);
  std::vector<int> pos(row_ptr.begin(), row_ptr.end() - 1);
  for (int t : c->plan.touched_tiles) { const int i = t / N, j = t % N; if (i < T && j < T) col[pos[i]++] = j; }
  // chunks of at most spcg_chunk_tiles() tiles of one row: one workgroup of the matrix-vector product each
  std::vector<int> chunk_row, chunk_ptr, row_chunk_ptr(T + 1, 0);
  for (int i = 0; i < T; ++i) {
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; e += spcg_chunk_tiles()) { chunk_row.push_back(i); chunk_ptr.push_back(e); }
    row_chunk_ptr[i + 1] = (int)chunk_row.size();
  }
  chunk_ptr.push_back(row_ptr[T]);
  c->n_schunks = (int)chunk_row.size();
  c->d_schunk_row = c->upload(chunk_row); c->d_schunk_ptr = c->upload(chunk_ptr); c->d_srow_chunk_ptr = c->upload(row_chunk_ptr);
  c->d_tcol = c->upload(col);
  const size_t n = (size_t)T * 64;
  c->d_sMinv = c->alloc<double>((size_t)T * 4096);
  c->d_sx = c->alloc<double>(n); c->d_sr = c->alloc<double>(n); c->d_sz = c->alloc<double>(n); c->d_sp0 = c->alloc<double>(n);
  c->d_sp1 = c->alloc<double>(n); c->d_sq = c->alloc<double>((size_t)c->n_schunks * 64);
  c->d_spart_pq = c->alloc<double>((size_t)c->n_schunks + 8); c->d_spart = c->alloc<double>(4 * (size_t)T + 8); c->d_ssc = c->alloc<double>(pcg_num_scalars());
  if (!c->d_sMinv || !c->d_sq || !c->d_ssc) return fail(c, BSGPU_ERR_DEVICE, "out of device memory (PCG on the reduced system)");
  c->spcg_built = true;
  return BSGPU_OK;
}

}  // namespace bsg
