// CPU check of the fused factorisation's task list (beam_slam_amd/csrc/dense_plan.h): replaying the list IN ORDER, every counter a task
// waits for must already have been advanced far enough by earlier tasks (the dead-lock-freedom argument of chol_fused_kernel), every
// structural tile must be produced exactly once, and a tile-level numeric replay must reproduce a dense Cholesky.
//   g++ -O2 -std=c++17 -I beam_slam_amd/csrc tests/plan/test_plan.cpp -o /tmp/test_plan && /tmp/test_plan
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "dense_plan.h"
#include "dim_order.h"
using namespace bsg;

static int replay(const DensePlan& P, const std::vector<uint8_t>& adjS /* P.T x P.T, S order */, unsigned seed, int max_chains);

static int check(int T, const std::vector<uint8_t>& adj, int max_chains, const std::vector<uint8_t>* leaf, unsigned seed) {
  DensePlan P;
  const int n_pose = T * 64 - (seed % 3 == 0 ? 23 : 0);
  if (seed % 2) { P.diag_tasks = true; P.rider_tasks = (int)(seed % 4); }
  P.split_depth = (int)((seed / 7) % 4);   // (0: no chunks; 1: the default; 2, 3: the last updates of a tile)
  P.build(n_pose, adj, max_chains, 1, true, leaf);
  std::vector<uint8_t> adjS((size_t)P.T * P.T, 0);
  for (int i = 0; i < P.T; ++i) for (int j = 0; j < P.T; ++j) if (adj[(size_t)i * P.T + j]) adjS[(size_t)P.perm[i] * P.T + P.perm[j]] = 1;
  for (int j = 0; j < n_pose; ++j) if (P.inat[P.dpos[j]] != j || P.dpos[j] != P.perm[j >> 6] * 64 + (j & 63)) { printf("  FAIL dpos / inat\n"); return 1; }
  return replay(P, adjS, seed, max_chains);
}

// The per-dimension order (dim_order.h) on a random block graph — keyframes of five 3-d blocks, poses coupled within a band, states
// coupled along the chain, optional hubs (an extrinsics block), far couplings (loop closures) and a dense prior on the first blocks —
// and the plan built on it: the order must be a permutation into whole tiles whose padding sits at the end of a supernode, every
// coupling must stay inside a supernode or go to an ANCESTOR (what makes the pieces independent), and the ticket list must replay.
static int check_ordered(std::mt19937& rng, unsigned seed) {
  const int nkf = 2 + rng() % 120, band = 1 + rng() % 14;
  const bool with_vb = rng() % 4 != 0, hub = rng() % 5 == 0, loops = rng() % 4 == 0, prior = rng() % 4 == 0;
  const int per = with_vb ? 5 : 2;
  DimOrder o;
  int t = 0;
  const int nbk = per * nkf + (hub ? 2 : 0) + (rng() % 3 == 0 ? 1 + rng() % 5 : 0);   // (+ a few scalar blocks at the end)
  for (int b = 0; b < nbk; ++b) { const int w = b < per * nkf + (hub ? 2 : 0) ? 3 : 1; o.blk_t0.push_back(t); o.blk_w.push_back(w); t += w; }
  o.n_pose = t;
  std::vector<std::vector<uint8_t>> A(nbk, std::vector<uint8_t>(nbk, 0));
  auto add = [&](int a, int b) { if (a != b) A[a][b] = A[b][a] = 1; };
  for (int i = 0; i < nkf; ++i) {
    for (int a = 0; a < per; ++a) for (int b = 0; b < per; ++b) add(per * i + a, per * i + b);
    for (int j = i + 1; j <= i + band && j < nkf; ++j) if (rng() % 8) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) add(per * i + a, per * j + b);
    if (with_vb && i + 1 < nkf) for (int a = 0; a < per; ++a) for (int b = 0; b < per; ++b) add(per * i + a, per * (i + 1) + b);
  }
  if (hub) for (int i = 0; i < nkf; ++i) for (int a = 0; a < 2; ++a) for (int h = 0; h < 2; ++h) add(per * i + a, per * nkf + h);
  if (loops) for (int e = 0; e < 1 + (int)(rng() % 6); ++e) { const int a = rng() % nkf, b = rng() % nkf; for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) add(per * a + x, per * b + y); }
  if (prior) { const int np = std::min(nbk, 3 + (int)(rng() % 30)); for (int a = 0; a < np; ++a) for (int b = 0; b < np; ++b) add(a, b); }
  for (int b = per * nkf + (hub ? 2 : 0); b < nbk; ++b) add(b, rng() % (per * nkf));   // the scalar blocks hang off some block
  o.adj_ptr.push_back(0);
  for (int a = 0; a < nbk; ++a) { for (int b = 0; b < nbk; ++b) if (A[a][b]) o.adj.push_back(b); o.adj_ptr.push_back((int)o.adj.size()); }
  o.max_depth = rng() % 7;
  if (rng() % 3 == 0) { o.t_hop = 1.0 + rng() % 20; o.t_step = 0.5 + (rng() % 8) * 0.5; }
  if (rng() % 2 == 0) {   // the settings bsgpu_finalize.cpp plans small systems under (the plan whose task list replays shortest is kept: every one of them must replay)
    static const double M[5][6] = {{5, 10, 4, 4.5, 24, 5}, {2, 8, 8, 4.5, 24, 5}, {3, 9, 8, 4.5, 24, 5}, {2, 10, 12, 3.0, 40, 5}, {5, 40, 0, 3.0, 40, 7}};
    const double* m = M[rng() % 5];
    o.t_chain0 = m[0]; o.t_hop = m[1]; o.t_hop_tile = m[2]; o.t_step3 = m[3]; o.merge_dims = (int)m[4]; o.max_depth = (int)m[5];
  }
  o.build();
  int fails = 0;
  auto fail = [&](const char* what) { if (fails++ < 5) printf("  FAIL (ordered, seed %u) %s\n", seed, what); };
  // permutation, tiles, padding
  std::vector<int> seen((size_t)o.T * 64, 0);
  for (int j = 0; j < o.n_pose; ++j) { if (o.dpos[j] < 0 || o.dpos[j] >= o.T * 64 || seen[o.dpos[j]]++) { fail("dpos is not a permutation"); break; } }
  if ((int)o.nreal.size() != o.T) fail("nreal size");
  for (int tt = 0; tt < o.T && tt < (int)o.nreal.size(); ++tt) for (int q = 0; q < 64; ++q) if ((seen[tt * 64 + q] != 0) != (q < o.nreal[tt])) { fail("padding is not at the end of the tile"); break; }
  for (int b = 0; b < nbk; ++b) for (int k = 1; k < o.blk_w[b]; ++k) if (o.dpos[o.blk_t0[b] + k] != o.dpos[o.blk_t0[b]] + k) fail("a block is not contiguous");
  // supernode of every block; couplings only inside a supernode or towards an ancestor
  std::vector<int> node_of(nbk, -1);
  for (size_t nd = 0; nd < o.nodes.size(); ++nd) for (int v : o.nodes[nd].verts) { if (node_of[v] >= 0) fail("block in two supernodes"); node_of[v] = (int)nd; }
  for (int b = 0; b < nbk; ++b) if (node_of[b] < 0) fail("block in no supernode");
  auto is_anc = [&](int a, int d) { for (int x = o.nodes[d].parent; x >= 0; x = o.nodes[x].parent) if (x == a) return true; return false; };
  for (int a = 0; a < nbk && fails == 0; ++a) for (int b = 0; b < a; ++b) if (A[a][b]) {
    const int na = node_of[a], nb2 = node_of[b];
    if (na != nb2 && !is_anc(na, nb2) && !is_anc(nb2, na)) fail("coupling between two independent supernodes");
  }
  if (fails) return fails;
  std::vector<uint8_t> adjS((size_t)o.T * o.T, 0);
  for (int a = 0; a < nbk; ++a) for (int b = 0; b < nbk; ++b) if (A[a][b] || a == b)
    for (int ka = 0; ka < o.blk_w[a]; ++ka) for (int kb = 0; kb < o.blk_w[b]; ++kb) adjS[(size_t)(o.dpos[o.blk_t0[a] + ka] >> 6) * o.T + (o.dpos[o.blk_t0[b] + kb] >> 6)] = 1;
  DensePlan P;
  if (seed % 2) { P.diag_tasks = true; P.rider_tasks = 1 + (int)(seed % 5); }   // (the LM diagonal / gradient norms carried by the launch)
  P.split_depth = (int)((seed / 7) % 4);
  P.build_ordered(o.n_pose, o.T, o.dpos, o.nreal, adjS, o.piece_ranges, o.sep_ranges_by_level);
  if (!o.sep_ranges_by_level.empty() && !P.bs_level_sync && P.bs_group_off.size() > 2) fail("the by-level back-substitution groups were refused");
  return fails + replay(P, adjS, seed, -1);
}

static long g_bulk_lists = 0, g_rows_lists = 0, g_row_segs = 0, g_row_seg_updates = 0;
static bool g_no_turns = false;   // the list being replayed is one for launches without turns (k_chol.hip FusedCtx::no_turn)
static long g_ext_tasks = 0, g_ext_plans = 0, g_diag_tasks = 0, g_split_tasks = 0, g_split_loaded = 0;
static int replay_list(const DensePlan& P, const std::vector<uint8_t>& adj, unsigned seed, int max_chains) {
  { long n = 0; for (const FusedTask& f : P.ftasks) n += (f.flags & kFusedExt) ? 1 : 0; g_ext_tasks += n; g_ext_plans += n > 0; }
  const int T = P.T;
  const int N = P.T + 1;
  // scalar stand-in: every tile is ONE number; A = M M^T + shift restricted to the structure is not SPD-safe, so use a diagonally
  // dominant matrix on the structure (tile (i,j) coupled iff adj) and compare with a dense Cholesky of the permuted matrix
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> A((size_t)N * N, 0.0);
  for (int i = 0; i < P.T; ++i) for (int j = 0; j < i; ++j) if (adj[(size_t)i * P.T + j] || adj[(size_t)j * P.T + i]) {
    const double v = U(rng);
    A[(size_t)i * N + j] = v; A[(size_t)j * N + i] = v;
  }
  for (int i = 0; i < P.T; ++i) { double s = 1.0; for (int j = 0; j < P.T; ++j) s += std::fabs(A[(size_t)i * N + j]); A[(size_t)i * N + i] = s; }
  for (int j = 0; j < P.T; ++j) { A[(size_t)P.T * N + j] = U(rng); }   // rhs row
  std::vector<double> R = A;   // reference: dense right-looking Cholesky on rows 0..T (row T = rhs, never a pivot)
  for (int k = 0; k < P.T; ++k) {
    const double l = std::sqrt(R[(size_t)k * N + k]);
    R[(size_t)k * N + k] = l;
    for (int i = k + 1; i < N; ++i) R[(size_t)i * N + k] /= l;
    for (int i = k + 1; i < N; ++i) for (int j = k + 1; j <= i; ++j) if (!(i == P.T && j == P.T)) R[(size_t)i * N + j] -= R[(size_t)i * N + k] * R[(size_t)j * N + k];
  }
  // replay
  std::vector<double> S = A, L((size_t)N * N, 0.0);
  std::vector<int> upd((size_t)N * N, 0), potrf_done(N, 0), xpub((size_t)N * N, 0);
  // kFusedSplit chunks: the chunks that reached a tile, the chunks that published their columns of X
  std::vector<int> upd_hi((size_t)N * N, 0), pub_hi((size_t)N * N, 0);
  auto tot_lo = [&](int a, int b) { return P.tile_tot[(size_t)a * N + b] & 0xffff; };
  auto tot_hi = [&](int a, int b) { return (int)((unsigned)P.tile_tot[(size_t)a * N + b] >> 16); };
  int fails = 0;
  auto fail = [&](const char* what, int t) { if (fails++ < 5) printf("  FAIL %s at task %d\n", what, t); };
  for (size_t t = 0; t < P.ftasks.size(); ++t) {
    const FusedTask& f = P.ftasks[t];
    if (f.flags & kFusedRider) { if (f.k < 0 || f.k >= P.rider_tasks) fail("rider unit", (int)t); continue; }
    if (f.flags & kFusedDiagAdd) {   // the LM diagonal of a tile: its first update, nothing to wait for
      if (!P.diag_tasks || f.k != f.ti || f.k != f.tj || f.k >= P.T || f.need_c != 0) fail("diagonal task", (int)t);
      if (upd[(size_t)f.k * N + f.k] != 0) fail("diagonal task is not the tile's first update", (int)t);
      if (potrf_done[f.k]) fail("diagonal task after its tile was factored", (int)t);
      upd[(size_t)f.k * N + f.k]++;
      ++g_diag_tasks;
      continue;
    }
    if (f.flags & kFusedChain) {
      const int b0 = f.k, m = f.ti;
      if (m < 1 || m > kChainMaxTiles) fail("chain length", (int)t);
      for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) {
        const bool present = (f.tj >> (i * (i + 1) / 2 + j)) & 1;
        if (upd[(size_t)(b0 + i) * N + b0 + j] != tot_lo(b0 + i, b0 + j)) fail("chain tile not final", (int)t);
        if (upd_hi[(size_t)(b0 + i) * N + b0 + j] != tot_hi(b0 + i, b0 + j)) fail("chain tile: chunks missing", (int)t);
        g_split_loaded += tot_hi(b0 + i, b0 + j);
        if (!present && S[(size_t)(b0 + i) * N + b0 + j] != 0.0) fail("absent tile is non-zero", (int)t);
      }
      for (int kk = b0; kk < b0 + m; ++kk) {   // dense factorisation of the chain's own tiles
        if (potrf_done[kk]) fail("tile factored twice", (int)t);
        const double l = std::sqrt(S[(size_t)kk * N + kk]);
        L[(size_t)kk * N + kk] = l;
        for (int i = kk + 1; i < b0 + m; ++i) L[(size_t)i * N + kk] = S[(size_t)i * N + kk] / l;
        for (int i = kk + 1; i < b0 + m; ++i) for (int j = kk + 1; j <= i; ++j) S[(size_t)i * N + j] -= L[(size_t)i * N + kk] * L[(size_t)j * N + kk];
        potrf_done[kk] = 1;
      }
      continue;
    }
    if (f.flags & kFusedSplit) {
      // one K-chunk of an update of a tile inside a chain: its share of the product goes to a partial tile of its own, no turn
      const int k = f.k, i = f.ti, j = f.tj, pch = f.tot_c & 0xff, Pn = f.tot_c >> 8;
      const bool diag = i == j, ext = (f.flags & kFusedExt) != 0, ext_chunk = ext && pch == Pn - 1;
      ++g_split_tasks;
      if (Pn != kFusedSplitChunks + (ext ? 1 : 0) || pch < 0 || pch >= Pn) fail("chunk numbering", (int)t);
      if (f.flags & (kFusedXiLp | kFusedXjLp | kFusedXjChain)) fail("a chunk reads published X", (int)t);
      if (!(i < P.T && P.fchain_of_tile[i] >= 0 && P.fchain_of_tile[i] == P.fchain_of_tile[j])) fail("chunk of a tile outside a chain", (int)t);
      if (P.fchain_of_tile[i] == P.fchain_of_tile[k]) fail("chunk inside the chain of its panel", (int)t);
      const int te = k + 1;
      if (!potrf_done[ext_chunk ? te : k]) fail("chunk: factor not out", (int)t);
      if (f.tot_i != P.tile_tot[(size_t)i * N + k] || f.tot_j != P.tile_tot[(size_t)j * N + k]) fail("chunk: tot mismatch", (int)t);
      if (upd[(size_t)i * N + k] < f.tot_i) fail("chunk: panel tile i not ready", (int)t);
      if (!diag && upd[(size_t)j * N + k] < f.tot_j) fail("chunk: panel tile j not ready", (int)t);
      if (ext && (P.fext_of[k] != te || P.fchain_of_tile[te] != P.fchain_of_tile[k])) fail("chunk: appendix of another chain", (int)t);
      if (ext_chunk && upd[(size_t)i * N + te] != tot_lo(i, te)) fail("chunk: appendix columns of tile i not final", (int)t);
      if (ext_chunk && !diag && upd[(size_t)j * N + te] != tot_lo(j, te)) fail("chunk: appendix columns of tile j not final", (int)t);
      if (potrf_done[i]) fail("chunk after its tile was factored", (int)t);
      const double xi = S[(size_t)i * N + k] / L[(size_t)k * N + k], xj = diag ? xi : S[(size_t)j * N + k] / L[(size_t)k * N + k];
      double val;
      if (!ext_chunk) val = xi * xj / kFusedSplitChunks;   // (the stand-in for 16 of the panel's 64 columns)
      else {
        const double xi_e = (S[(size_t)i * N + te] - xi * L[(size_t)te * N + k]) / L[(size_t)te * N + te];
        const double xj_e = diag ? xi_e : (S[(size_t)j * N + te] - xj * L[(size_t)te * N + k]) / L[(size_t)te * N + te];
        val = xi_e * xj_e;
        if (f.flags & kFusedPublishX) L[(size_t)i * N + te] = xi_e;
      }
      if (f.need_c != tot_lo(i, j) || upd[(size_t)i * N + j] != f.need_c) fail("chunk: the tile's updates with a turn of their own are not all in", (int)t);
      if (upd_hi[(size_t)i * N + j] >= tot_hi(i, j)) fail("chunk: more chunks than the tile counts", (int)t);
      S[(size_t)i * N + j] -= val;
      upd_hi[(size_t)i * N + j]++;
      if (f.flags & kFusedPublishX) {
        if (!diag) fail("an off-diagonal chunk publishes X", (int)t);
        if (++pub_hi[(size_t)i * N + k] == Pn) { L[(size_t)i * N + k] = xi; xpub[(size_t)i * N + k] = 1; upd[(size_t)i * N + k]++; }
      }
      continue;
    }
    if (f.flags & kFusedRowSeg) {
      // several updates of row tile i by panel k, each reading the X its two diagonal tasks have published; their products are counted at their tiles in the
      // segment's order (a list for launches WITHOUT turns: the tile's count only has to be below its total)
      const int k = f.k, i = f.ti;
      const bool ext = (f.flags & kFusedExt) != 0;
      ++g_row_segs;
      if (f.tot_j < 1 || f.tot_j > kFusedRowSegMax || f.tj < 0 || 2 * (f.tj + f.tot_j) > (int)P.frow_items.size()) { fail("segment: pairs out of range", (int)t); continue; }
      if (f.tot_i != P.tile_tot[(size_t)i * N + k]) fail("segment: tot mismatch", (int)t);
      if (upd[(size_t)i * N + k] < f.tot_i + 1 || !xpub[(size_t)i * N + k]) fail("segment: X_i not published", (int)t);
      for (int q = 0; q < f.tot_j; ++q) {
        const int j = P.frow_items[2 * (f.tj + q)], totj = P.frow_items[2 * (f.tj + q) + 1];
        ++g_row_seg_updates;
        if (j >= i || j <= k) fail("segment: pair's tile", (int)t);
        if (totj != P.tile_tot[(size_t)j * N + k]) fail("segment: pair's tot", (int)t);
        if (upd[(size_t)j * N + k] < totj + 1 || !xpub[(size_t)j * N + k]) fail("segment: X_j not published", (int)t);
        if (upd[(size_t)i * N + j] >= tot_lo(i, j)) fail("segment: more updates than the tile counts", (int)t);
        double v = L[(size_t)i * N + k] * L[(size_t)j * N + k];
        if (ext) v += L[(size_t)i * N + k + 1] * L[(size_t)j * N + k + 1];
        S[(size_t)i * N + j] -= v;
        upd[(size_t)i * N + j]++;
      }
      continue;
    }
    const int k = f.k, i = f.ti, j = f.tj;
    const bool diag = i == j, solve_i = !(f.flags & kFusedXiLp), solve_j = !diag && !(f.flags & (kFusedXjLp | kFusedXjChain));
    if (!potrf_done[k] && (solve_i || solve_j || (f.flags & kFusedXjChain))) fail("L_kk not out", (int)t);
    if (f.tot_i != P.tile_tot[(size_t)i * N + k] || (!(f.flags & kFusedXjChain) && f.tot_j != P.tile_tot[(size_t)j * N + k])) fail("tot mismatch", (int)t);   // (a task whose tj is in the chain of k waits for the chain's flag, not for a count of tile (tj, k) — a tile inside a chain, whose count may carry chunks)
    if (upd[(size_t)i * N + k] < f.tot_i + (solve_i ? 0 : 1)) fail("panel tile i not ready", (int)t);
    if (!diag && !(f.flags & kFusedXjChain) && upd[(size_t)j * N + k] < f.tot_j + (solve_j ? 0 : 1)) fail("panel tile j not ready", (int)t);
    if ((f.flags & kFusedXjChain) && !(P.fchain_of_tile[j] == P.fchain_of_tile[k] && j > k)) fail("XjChain flag", (int)t);
    if (P.fchain_of_tile[k] >= 0 && i < P.T && P.fchain_of_tile[i] == P.fchain_of_tile[k]) fail("update task inside a chain", (int)t);
    const double xi = solve_i ? S[(size_t)i * N + k] / L[(size_t)k * N + k] : L[(size_t)i * N + k];
    const double xj = diag ? xi : (solve_j ? S[(size_t)j * N + k] / L[(size_t)k * N + k] : L[(size_t)j * N + k]);
    // a panel that carries its chain's appendix tile k + 1 (kFusedExt): X(., k + 1) = (A(., k + 1) - X(., k) L(k + 1, k)) / L(k + 1, k + 1)
    const bool ext = (f.flags & kFusedExt) != 0;
    double xi_e = 0.0, xj_e = 0.0;
    if (ext) {
      const int te = k + 1;
      if (P.fext_of[k] != te || P.fchain_of_tile[te] != P.fchain_of_tile[k]) fail("appendix of another chain", (int)t);
      if (f.flags & kFusedXjChain) fail("appendix task inside the chain", (int)t);
      if (!potrf_done[te] && (solve_i || solve_j)) fail("appendix not factored", (int)t);
      if (solve_i && upd[(size_t)i * N + te] != tot_lo(i, te)) fail("appendix columns of tile i not final", (int)t);
      if (solve_j && upd[(size_t)j * N + te] != tot_lo(j, te)) fail("appendix columns of tile j not final", (int)t);
      xi_e = solve_i ? (S[(size_t)i * N + te] - xi * L[(size_t)te * N + k]) / L[(size_t)te * N + te] : L[(size_t)i * N + te];
      xj_e = diag ? xi_e : (solve_j ? (S[(size_t)j * N + te] - xj * L[(size_t)te * N + k]) / L[(size_t)te * N + te] : L[(size_t)j * N + te]);
    }
    if (!solve_i && !xpub[(size_t)i * N + k]) fail("X_i read before it was published", (int)t);
    if (!diag && (f.flags & kFusedXjLp) && !xpub[(size_t)j * N + k]) fail("X_j read before it was published", (int)t);
    if (f.need_c >= 0) {
      if (!g_no_turns && upd[(size_t)i * N + j] != f.need_c) fail("turn", (int)t);
      if (g_no_turns && upd[(size_t)i * N + j] >= tot_lo(i, j)) fail("more updates than the tile counts", (int)t);
      S[(size_t)i * N + j] -= xi * xj + xi_e * xj_e;
      upd[(size_t)i * N + j]++;
    }
    if (f.flags & kFusedPublishX) { L[(size_t)i * N + k] = xi; xpub[(size_t)i * N + k] = 1; upd[(size_t)i * N + k]++; if (ext) L[(size_t)i * N + k + 1] = xi_e; }
  }
  for (int k = 0; k < P.T; ++k) if (!potrf_done[k]) fail("tile never factored", k);
  double emax = 0.0;
  for (int i = 0; i < N; ++i) for (int j = 0; j <= i && j < P.T; ++j) emax = std::max(emax, std::fabs(L[(size_t)i * N + j] - R[(size_t)i * N + j]));
  if (emax > 1e-9) { printf("  FAIL factor mismatch %.3e\n", emax); ++fails; }
  if (fails) printf("T %d chains %d seed %u: %d failures (%zu tasks, %zu chains)\n", T, max_chains, seed, fails, P.ftasks.size(), P.fchain_begin.size());
  return fails;
}

// the plan's list, and its list without the diagonal / rider tasks (what a launch that carries neither takes)
static int replay(const DensePlan& P, const std::vector<uint8_t>& adj, unsigned seed, int max_chains) {
  int f = replay_list(P, adj, seed, max_chains);
  if (P.frows_src >= 0) {   // ... and the list with the row segments (launches without turns), on its source's tile counts
    DensePlan Q = P;
    Q.ftasks = P.ftasks_rows;
    Q.tile_tot = P.frows_src == 2 ? P.tile_tot_bulk : P.frows_src == 1 ? P.tile_tot_plain : P.tile_tot;
    if (P.frows_src > 0) { Q.diag_tasks = false; Q.rider_tasks = 0; }
    size_t members = 0, src_members = 0;
    for (const FusedTask& t : Q.ftasks) members += (t.flags & kFusedRowSeg) ? (size_t)t.tot_j : 1;
    src_members = (P.frows_src == 2 ? P.ftasks_bulk : P.frows_src == 1 ? P.ftasks_plain : P.ftasks).size();
    if (members != src_members) { printf("  FAIL rows list: %zu updates for %zu tasks of its source\n", members, src_members); ++f; }
    g_no_turns = true;
    f += replay_list(Q, adj, seed, max_chains);
    g_no_turns = false;
    ++g_rows_lists;
  }
  if (P.diag_tasks || P.rider_tasks > 0) {
    if (P.ftasks_plain.empty()) { printf("  FAIL no plain list\n"); return f + 1; }
    DensePlan Q = P;
    Q.ftasks = P.ftasks_plain; Q.tile_tot = P.tile_tot_plain; Q.diag_tasks = false; Q.rider_tasks = 0;
    f += replay_list(Q, adj, seed, max_chains);
  }
  if (!P.ftasks_bulk.empty()) {   // ... and the list without the K-chunks (what a batch of many windows takes)
    DensePlan Q = P;
    Q.ftasks = P.ftasks_bulk; Q.tile_tot = P.tile_tot_bulk; Q.diag_tasks = false; Q.rider_tasks = 0;
    for (const FusedTask& t : Q.ftasks) if (t.flags & (kFusedSplit | kFusedDiagAdd | kFusedRider)) { printf("  FAIL bulk list holds a chunk / diagonal / rider task\n"); return f + 1; }
    f += replay_list(Q, adj, seed, max_chains);
    ++g_bulk_lists;
  }
  return f;
}

int main() {
  int fails = 0, cases = 0;
  std::mt19937 rng(12345);
  for (int rep = 0; rep < 400; ++rep) {
    const int T = 1 + rng() % 60;
    const int band = 1 + rng() % 5;
    std::vector<uint8_t> adj((size_t)T * T, 0);
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) if (std::abs(i - j) <= band) adj[(size_t)i * T + j] = 1;
    const int extra = rng() % 4 == 0 ? rng() % (T + 1) : 0;   // far couplings (loop closures, long tracks)
    for (int e = 0; e < extra; ++e) { const int a = rng() % T, b = rng() % T; adj[(size_t)a * T + b] = adj[(size_t)b * T + a] = 1; }
    if (rng() % 7 == 0) for (int i = 0; i < T; ++i) { adj[(size_t)i * T + 0] = adj[i] = 1; }   // a block coupled to everything (extrinsics)
    std::vector<uint8_t> leaf(T, 0);
    const bool use_leaf = rng() % 5 == 0 && T > 6;
    if (use_leaf) {   // leaf tiles: coupled to a few core tiles, to no other leaf
      const int nl = 1 + rng() % (T / 2);
      for (int t = T - nl; t < T; ++t) {
        leaf[t] = 1;
        for (int j = 0; j < T; ++j) adj[(size_t)t * T + j] = adj[(size_t)j * T + t] = 0;
        adj[(size_t)t * T + t] = 1;
        for (int e = 0; e < 3; ++e) { const int j = rng() % (T - nl); adj[(size_t)t * T + j] = adj[(size_t)j * T + t] = 1; }
      }
    }
    const int chains[] = {1, 2, 4, 8, 16, 32};
    fails += check(T, adj, chains[rng() % 6], use_leaf ? &leaf : nullptr, (unsigned)rng());
    ++cases;
  }
  printf("%d cases, %d failures\n", cases, fails);
  int ocases = 0, ofails = 0;
  for (int rep = 0; rep < 400; ++rep) { ofails += check_ordered(rng, (unsigned)rng()); ++ocases; }
  printf("%d ordered cases, %d failures\n", ocases, ofails);
  printf("appendix tiles: %ld tasks in %ld plans\n", g_ext_tasks, g_ext_plans);
  if (g_ext_plans < 20) { printf("too few plans with an appendix tile\n"); ++fails; }
  printf("split chunks: %ld tasks, %ld partial tiles added by chains\n", g_split_tasks, g_split_loaded);
  if (g_split_tasks < 1000 || g_split_loaded != g_split_tasks) { printf("the split chunks and the chains' partial tiles do not match\n"); ++fails; }
  printf("bulk lists replayed: %ld\n", g_bulk_lists);
  if (g_bulk_lists < 100) { printf("too few bulk lists\n"); ++fails; }
  printf("lists with row segments replayed: %ld (%ld segments, %ld updates in them)\n", g_rows_lists, g_row_segs, g_row_seg_updates);
  if (g_rows_lists < 400 || g_row_seg_updates < 2 * g_row_segs || g_row_segs < 1000) { printf("too few row segments\n"); ++fails; }
  printf("diagonal tasks: %ld\n", g_diag_tasks);
  if (g_diag_tasks < 1000) { printf("too few diagonal tasks\n"); ++fails; }
  fails += ofails;
  return fails ? 1 : 0;
}
