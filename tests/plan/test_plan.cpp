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
using namespace bsg;

static int check(int T, const std::vector<uint8_t>& adj, int max_chains, const std::vector<uint8_t>* leaf, unsigned seed) {
  DensePlan P;
  const int n_pose = T * 64 - (seed % 3 == 0 ? 23 : 0);
  P.build(n_pose, adj, max_chains, 1, true, leaf);
  const int N = P.T + 1;
  // scalar stand-in: every tile is ONE number; A = M M^T + shift restricted to the structure is not SPD-safe, so use a diagonally
  // dominant matrix on the structure (tile (i,j) coupled iff adj) and compare with a dense Cholesky of the permuted matrix
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> A((size_t)N * N, 0.0);
  for (int i = 0; i < P.T; ++i) for (int j = 0; j < i; ++j) if (adj[(size_t)i * P.T + j] || adj[(size_t)j * P.T + i]) {
    const double v = U(rng);
    A[(size_t)P.perm[i] * N + P.perm[j]] = v; A[(size_t)P.perm[j] * N + P.perm[i]] = v;
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
  int fails = 0;
  auto fail = [&](const char* what, int t) { if (fails++ < 5) printf("  FAIL %s at task %d\n", what, t); };
  for (size_t t = 0; t < P.ftasks.size(); ++t) {
    const FusedTask& f = P.ftasks[t];
    if (f.flags & kFusedChain) {
      const int b0 = f.k, m = f.ti;
      if (m < 1 || m > kChainMaxTiles) fail("chain length", (int)t);
      for (int i = 0; i < m; ++i) for (int j = 0; j <= i; ++j) {
        const bool present = (f.tj >> (i * (i + 1) / 2 + j)) & 1;
        if (upd[(size_t)(b0 + i) * N + b0 + j] != P.tile_tot[(size_t)(b0 + i) * N + b0 + j]) fail("chain tile not final", (int)t);
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
    const int k = f.k, i = f.ti, j = f.tj;
    const bool diag = i == j, solve_i = !(f.flags & kFusedXiLp), solve_j = !diag && !(f.flags & (kFusedXjLp | kFusedXjChain));
    if (!potrf_done[k] && (solve_i || solve_j || (f.flags & kFusedXjChain))) fail("L_kk not out", (int)t);
    if (f.tot_i != P.tile_tot[(size_t)i * N + k] || f.tot_j != P.tile_tot[(size_t)j * N + k]) fail("tot mismatch", (int)t);
    if (upd[(size_t)i * N + k] < f.tot_i + (solve_i ? 0 : 1)) fail("panel tile i not ready", (int)t);
    if (!diag && !(f.flags & kFusedXjChain) && upd[(size_t)j * N + k] < f.tot_j + (solve_j ? 0 : 1)) fail("panel tile j not ready", (int)t);
    if ((f.flags & kFusedXjChain) && !(P.fchain_of_tile[j] == P.fchain_of_tile[k] && j > k)) fail("XjChain flag", (int)t);
    if (P.fchain_of_tile[k] >= 0 && i < P.T && P.fchain_of_tile[i] == P.fchain_of_tile[k]) fail("update task inside a chain", (int)t);
    const double xi = solve_i ? S[(size_t)i * N + k] / L[(size_t)k * N + k] : L[(size_t)i * N + k];
    const double xj = diag ? xi : (solve_j ? S[(size_t)j * N + k] / L[(size_t)k * N + k] : L[(size_t)j * N + k]);
    if (!solve_i && !xpub[(size_t)i * N + k]) fail("X_i read before it was published", (int)t);
    if (!diag && (f.flags & kFusedXjLp) && !xpub[(size_t)j * N + k]) fail("X_j read before it was published", (int)t);
    if (f.need_c >= 0) {
      if (upd[(size_t)i * N + j] != f.need_c) fail("turn", (int)t);
      S[(size_t)i * N + j] -= xi * xj;
      upd[(size_t)i * N + j]++;
    }
    if (f.flags & kFusedPublishX) { L[(size_t)i * N + k] = xi; xpub[(size_t)i * N + k] = 1; upd[(size_t)i * N + k]++; }
  }
  for (int k = 0; k < P.T; ++k) if (!potrf_done[k]) fail("tile never factored", k);
  double emax = 0.0;
  for (int i = 0; i < N; ++i) for (int j = 0; j <= i && j < P.T; ++j) emax = std::max(emax, std::fabs(L[(size_t)i * N + j] - R[(size_t)i * N + j]));
  if (emax > 1e-9) { printf("  FAIL factor mismatch %.3e\n", emax); ++fails; }
  if (fails) printf("T %d chains %d seed %u: %d failures (%zu tasks, %zu chains)\n", T, max_chains, seed, fails, P.ftasks.size(), P.fchain_begin.size());
  return fails;
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
  return fails ? 1 : 0;
}
