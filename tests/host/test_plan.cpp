// CPU test of the tiled-Cholesky planner (beam_slam_amd/csrc/dense_plan.h): for random block-banded SPD systems the
// plan (tile order, symbolic fill, step schedule, look-ahead flags, stand-alone potrf lists, shared-tile flags,
// back-substitution plan) is EXECUTED on the host with the semantics of the device kernels — every panel of a step
// reads the state from before the step, updates to a tile are only summed (atomics) where the plan says the tile is
// shared — and the result must solve the system.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../beam_slam_amd/csrc/dense_plan.h"

using bsg::DensePlan;
using bsg::PanelDesc;
static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

struct Mat { int n; std::vector<double> a; double& operator()(int i, int j) { return a[(size_t)i * n + j]; } };

static bool run_case(int n_pose, int band_tiles, int max_chains, int min_piece, bool shared, unsigned seed, int n_far = 0, int* n_pieces_out = nullptr,
                     int n_leaf = 0 /* the last n_leaf tiles are coupled to core tiles only (inverse-depth landmark tiles) */) {
  std::mt19937 rng(seed);
  std::normal_distribution<double> N(0, 1);
  const int T = (n_pose + 63) / 64;
  std::vector<uint8_t> adj((size_t)T * T, 0);
  const int Tc = T - n_leaf;
  for (int i = 0; i < Tc; ++i) for (int j = 0; j < Tc; ++j) if (std::abs(i - j) <= band_tiles) adj[(size_t)i * T + j] = 1;
  std::vector<uint8_t> leaf(T, 0);
  for (int t = Tc; t < T; ++t) {   // a leaf tile: itself + a few neighbouring core tiles (the keyframes that see its landmarks)
    leaf[t] = 1;
    adj[(size_t)t * T + t] = 1;
    const int c0 = (int)(rng() % Tc), span = 1 + (int)(rng() % 3);
    for (int j = c0; j < std::min(Tc, c0 + span); ++j) adj[(size_t)t * T + j] = adj[(size_t)j * T + t] = 1;
  }
  // long feature tracks / loop closures: a few tile pairs far outside the band
  for (int f = 0; f < n_far; ++f) { const int i = (int)(rng() % T), j = (int)(rng() % T); adj[(size_t)i * T + j] = adj[(size_t)j * T + i] = 1; }
  // SPD matrix with that tile structure (natural order)
  Mat A{n_pose, std::vector<double>((size_t)n_pose * n_pose, 0.0)};
  for (int i = 0; i < n_pose; ++i) for (int j = 0; j <= i; ++j)
    if (adj[(size_t)(i / 64) * T + j / 64]) { const double v = (i == j) ? 0.0 : 0.3 * N(rng); A(i, j) = v; A(j, i) = v; }
  for (int i = 0; i < n_pose; ++i) { double s = 0; for (int j = 0; j < n_pose; ++j) s += std::fabs(A(i, j)); A(i, i) = s + 1.0; }
  std::vector<double> b(n_pose);
  for (auto& v : b) v = N(rng);
  DensePlan P;
  P.build(n_pose, adj, max_chains, min_piece, shared, n_leaf ? &leaf : nullptr);
  if (n_leaf) CHECK(P.n_leaf_tiles == n_leaf);
  if (n_pieces_out) *n_pieces_out = P.n_pieces;
  const int np = P.npad, NT = P.T + 1;
  // S in solver order with the rhs as row rhs_row; unit pivots on padding
  Mat S{np, std::vector<double>((size_t)np * np, 0.0)};
  for (int i = 0; i < n_pose; ++i) for (int j = 0; j < n_pose; ++j) S(P.spos(i), P.spos(j)) = A(i, j);
  for (int j = 0; j < n_pose; ++j) S(P.rhs_row, P.spos(j)) = b[j];
  for (int i = 0; i < np; ++i) if (S(i, i) == 0.0) S(i, i) = 1.0;
  Mat Lp{np, std::vector<double>((size_t)np * np, 0.0)};
  std::vector<int> factored(NT, 0);
  auto potrf_tile = [&](int t) {
    CHECK(!factored[t]);
    factored[t] = 1;
    const int o = t * 64, nr = P.nreal[t];
    for (int j = 0; j < 64; ++j) {
      if (j >= nr) { for (int i = j; i < 64; ++i) S(o + i, o + j) = (i == j) ? 1.0 : 0.0; continue; }
      double d = S(o + j, o + j);
      for (int k = 0; k < j; ++k) d -= S(o + j, o + k) * S(o + j, o + k);
      if (!(d > 0)) { ++g_fail; std::printf("non-positive pivot in tile %d\n", t); return; }
      const double l = std::sqrt(d);
      S(o + j, o + j) = l;
      for (int i = j + 1; i < 64; ++i) { double s = S(o + i, o + j); for (int k = 0; k < j; ++k) s -= S(o + i, o + k) * S(o + j, o + k); S(o + i, o + j) = s / l; }
    }
  };
  for (int st = 0; st < P.n_steps(); ++st) {
    for (int i = P.potrf_before_step_off[st]; i < P.potrf_before_step_off[st + 1]; ++i) potrf_tile(P.potrf_tiles[i]);
    // all panels of the step read the pre-step state; their tile updates are deltas applied afterwards
    struct Upd { int ti, tj; bool shared; std::vector<double> d; };
    std::vector<Upd> upds;
    std::vector<int> lookahead_tiles;
    for (int pi = P.step_off[st]; pi < P.step_off[st + 1]; ++pi) {
      const PanelDesc& pd = P.panels[pi];
      CHECK(factored[pd.k]);
      const int c0 = pd.k * 64;
      std::vector<std::vector<double>> X(pd.n_rows, std::vector<double>(64 * 64));
      for (int q = 0; q < pd.n_rows; ++q) {
        const int r0 = P.rows_flat[pd.row_off + q] * 64;
        for (int i = 0; i < 64; ++i)
          for (int j = 0; j < 64; ++j) {   // X = A L^-T
            double s = S(r0 + i, c0 + j);
            for (int k = 0; k < j; ++k) s -= X[q][i * 64 + k] * S(c0 + j, c0 + k);
            X[q][i * 64 + j] = s / S(c0 + j, c0 + j);
          }
        for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) Lp(r0 + i, c0 + j) = X[q][i * 64 + j];
      }
      for (int qi = 0; qi < pd.n_rows; ++qi) for (int qj = 0; qj <= qi; ++qj) {
        Upd u; u.ti = P.rows_flat[pd.row_off + qi]; u.tj = P.rows_flat[pd.row_off + qj];
        u.shared = ((pd.shared_mask >> std::min(qi, 31)) & 1) && ((pd.shared_mask >> std::min(qj, 31)) & 1);
        u.d.assign(64 * 64, 0.0);
        for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double s = 0; for (int k = 0; k < 64; ++k) s += X[qi][i * 64 + k] * X[qj][j * 64 + k]; u.d[i * 64 + j] = -s; }
        upds.push_back(std::move(u));
      }
      for (int q = 0; q < pd.n_rows && q < 31; ++q)   // tiles that receive their last update in this step: one arrival per panel that has them
        if ((pd.final_mask >> q) & 1) { const int t = P.rows_flat[pd.row_off + q]; CHECK(t < P.T); lookahead_tiles.push_back(t); }
    }
    // a tile written by more than one panel of this step must be flagged shared by all of them (else the device loses an update)
    std::vector<int> writers((size_t)NT * NT, 0), nonshared((size_t)NT * NT, 0);
    for (const auto& u : upds) { writers[(size_t)u.ti * NT + u.tj]++; if (!u.shared) nonshared[(size_t)u.ti * NT + u.tj]++; }
    for (int t = 0; t < NT * NT; ++t) if (writers[t] > 1 && t != P.T * NT + P.T) CHECK(nonshared[t] == 0);
    for (const auto& u : upds) {
      if (u.ti == P.T && u.tj == P.T) continue;   // the rhs tile's own diagonal is never used
      for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) S(u.ti * 64 + i, u.tj * 64 + j) += u.d[i * 64 + j];
    }
    // the last arriver factors the tile: the arrivals of the step must be exactly what the device waits for (tile_sync), the
    // tile must never be updated again, and every updater must have announced it
    std::sort(lookahead_tiles.begin(), lookahead_tiles.end());
    for (size_t i = 0; i < lookahead_tiles.size();) {
      const int t = lookahead_tiles[i];
      size_t j = i;
      while (j < lookahead_tiles.size() && lookahead_tiles[j] == t) ++j;
      CHECK((int)(j - i) == P.tile_sync[t]);
      CHECK(writers[(size_t)t * NT + t] == (int)(j - i));
      CHECK(P.tile_sync[(size_t)(P.T + 1) + t] == 0);
      potrf_tile(t);
      i = j;
    }
  }
  for (int t = 0; t < P.T; ++t) CHECK(factored[t] == 1);
  // back-substitution following the plan: separators step by step, then the pieces
  std::vector<double> y(np, 0.0);
  for (int j = 0; j < P.T * 64; ++j) y[j] = Lp(P.rhs_row, j);
  auto solve_panel = [&](int kb, const int* rows, int n_rows) {
    const int c0 = kb * 64, nr = P.nreal[kb];
    std::vector<double> rhs(64);
    for (int c = 0; c < 64; ++c) {
      double s = 0;
      for (int q = 0; q < n_rows; ++q) { const int r0 = rows[q] * 64; for (int i = 0; i < 64; ++i) s += Lp(r0 + i, c0 + c) * y[r0 + i]; }
      rhs[c] = y[c0 + c] - s;
    }
    for (int j = 63; j >= 0; --j) {
      double s = rhs[j];
      for (int i = j + 1; i < 64; ++i) s -= S(c0 + i, c0 + j) * y[c0 + i];
      y[c0 + j] = (j < nr) ? s / S(c0 + j, c0 + j) : 0.0;
    }
  };
  std::vector<int> solved(P.T, 0);
  for (size_t g = 0; g + 1 < P.bs_group_off.size(); ++g) {
    std::vector<int> done_in_group;
    for (int c = P.bs_group_off[g]; c < P.bs_group_off[g + 1]; ++c)   // chains of one group run concurrently: they may only
      for (int k = P.chain_end[c] - 1; k >= P.chain_begin[c]; --k) {   // read tiles of EARLIER groups or of their own chain
        const PanelDesc& pd = P.panels[P.panel_of_tile[k]];
        CHECK(pd.k == k);
        for (int q = 0; q < pd.n_rows; ++q) {
          const int t = P.rows_flat[pd.row_off + q];
          CHECK(t == P.T || solved[t] == 1 || (t >= P.chain_begin[c] && t < P.chain_end[c] && t > k));
        }
        solve_panel(k, &P.rows_flat[pd.row_off], pd.n_rows);
        done_in_group.push_back(k);
      }
    for (int k : done_in_group) solved[k] = 1;
  }
  for (int t = 0; t < P.T; ++t) CHECK(solved[t]);
  // residual of A x = b
  double err = 0, nb = 0;
  for (int i = 0; i < n_pose; ++i) {
    double s = 0;
    for (int j = 0; j < n_pose; ++j) s += A(i, j) * y[P.spos(j)];
    err = std::max(err, std::fabs(s - b[i])); nb = std::max(nb, std::fabs(b[i]));
  }
  const bool ok = err <= 1e-10 * std::max(1.0, nb);
  std::printf("n=%4d T=%2d band=%d chains<=%2d min_piece=%d shared=%d -> pieces=%d steps=%2d potrf launches=%d  residual %.2e %s\n", n_pose, P.T, band_tiles,
              max_chains, min_piece, (int)shared, P.n_pieces, P.n_steps(), (int)[&] { int n = 0; for (int s = 0; s < P.n_steps(); ++s) n += P.potrf_before_step_off[s + 1] > P.potrf_before_step_off[s]; return n; }(), err, ok ? "" : "FAILED");
  if (!ok) ++g_fail;
  return ok;
}

int main() {
  unsigned seed = 1;
  for (bool shared : {false, true})
    for (int chains : {1, 2, 4, 8, 16})
      for (int min_piece : {1, 3}) {
        run_case(700, 1, chains, min_piece, shared, seed++);
        run_case(1500, 2, chains, min_piece, shared, seed++);
        run_case(1930, 3, chains, min_piece, shared, seed++);     // partial last tile
      }
  run_case(40, 1, 4, 1, true, seed++);
  run_case(64, 0, 4, 1, true, seed++);
  run_case(900, 20, 8, 1, true, seed++);                          // dense: no dissection possible
  // a banded window with a few couplings far outside the band keeps its independent pieces (the band width that sizes the
  // separators is a percentile, not the maximum) and still factors exactly
  for (int n_far : {1, 3, 6}) {
    int pieces = 0;
    run_case(3000, 3, 16, 1, true, seed++, n_far, &pieces);
    CHECK(pieces >= 4);
  }
  // leaf tiles (inverse-depth landmarks): ordered first, the core dissected on its adjacency including their fill
  for (int chains : {1, 4, 16}) {
    run_case(64 * 30, 2, chains, 1, true, seed++, 0, nullptr, 18);
    run_case(64 * 40 + 17, 3, chains, 1, true, seed++, 0, nullptr, 25);   // partial last (leaf) tile
    run_case(64 * 12, 1, chains, 1, false, seed++, 0, nullptr, 9);
  }
  if (g_fail) { std::printf("FAILED: %d checks\n", g_fail); return 1; }
  std::printf("ALL PLAN TESTS PASSED\n");
  return 0;
}
