// Offline look at the per-dimension order (beam_slam_amd/csrc/dim_order.h) and the plan built on it, without a GPU:
//   g++ -O2 -std=c++17 -I beam_slam_amd/csrc scripts/dim_order_tool.cpp -o /tmp/dim_order_tool
//   python scripts/dim_order_graph.py c2 > /tmp/c2.graph && /tmp/dim_order_tool /tmp/c2.graph
// Input: "nbk", then nbk lines "t0 w", then "nedges", then edges "a b".
#include <cstdio>
#include <cstdlib>
#include <set>
#include "dense_plan.h"
#include "dim_order.h"
using namespace bsg;
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "r");
  int nbk = 0, ne = 0;
  if (!f || fscanf(f, "%d", &nbk) != 1) return 1;
  DimOrder o;
  for (int b = 0; b < nbk; ++b) { int t0, w; if (fscanf(f, "%d %d", &t0, &w) != 2) return 1; o.blk_t0.push_back(t0); o.blk_w.push_back(w); o.n_pose = t0 + w; }
  if (fscanf(f, "%d", &ne) != 1) return 1;
  std::vector<std::set<int>> adj(nbk);
  for (int e = 0; e < ne; ++e) { int a, b; if (fscanf(f, "%d %d", &a, &b) != 2) return 1; if (a != b) { adj[a].insert(b); adj[b].insert(a); } }
  o.adj_ptr.push_back(0);
  for (int b = 0; b < nbk; ++b) { for (int x : adj[b]) o.adj.push_back(x); o.adj_ptr.push_back((int)o.adj.size()); }
  if (getenv("T_STEP")) o.t_step = atof(getenv("T_STEP"));
  if (getenv("T_HOP")) o.t_hop = atof(getenv("T_HOP"));
  if (getenv("T_STEP2")) o.t_step2 = atof(getenv("T_STEP2"));
  if (getenv("T_STEP3")) o.t_step3 = atof(getenv("T_STEP3"));
  if (getenv("T_CHAIN0")) o.t_chain0 = atof(getenv("T_CHAIN0"));
  if (getenv("T_HOP_TILE")) o.t_hop_tile = atof(getenv("T_HOP_TILE"));
  if (getenv("MERGE")) o.merge_dims = atoi(getenv("MERGE"));
  if (getenv("DEPTH")) o.max_depth = atoi(getenv("DEPTH"));
  if (getenv("HUB")) o.hub_frac = atof(getenv("HUB"));
  if (getenv("ABSORB")) o.absorb = atoi(getenv("ABSORB")) != 0;
  o.build();
  printf("n_pose %d blocks %d: T %d nodes %d depth %d est %.1f us\n", o.n_pose, nbk, o.T, o.n_nodes, o.depth, o.est_path_us);
  for (size_t i = 0; i < o.nodes.size(); ++i) {
    const auto& nd = o.nodes[i];
    printf("  %2zu %*s%s dims %d (blocks %d..%d) parent %d\n", i, 2 * nd.depth, "", nd.is_sep ? "sep" : "piece", nd.dims, nd.verts.empty() ? -1 : nd.verts.front(), nd.verts.empty() ? -1 : nd.verts.back(), nd.parent);
  }
  const int To = o.T;
  std::vector<uint8_t> adjS((size_t)To * To, 0);
  for (int a = 0; a < nbk; ++a) {
    auto mark = [&](int x, int y) { for (int ka = 0; ka < o.blk_w[x]; ++ka) for (int kb = 0; kb < o.blk_w[y]; ++kb) { const int p = o.dpos[o.blk_t0[x] + ka] >> 6, q = o.dpos[o.blk_t0[y] + kb] >> 6; adjS[(size_t)p * To + q] = adjS[(size_t)q * To + p] = 1; } };
    mark(a, a);
    for (int b : adj[a]) mark(a, b);
  }
  {   // couplings between tiles of different PIECES (there must be none: pieces are independent)
    std::vector<int> piece_of_tile(To, -1);
    for (size_t i = 0; i < o.piece_ranges.size(); ++i) for (int t = o.piece_ranges[i].first; t < o.piece_ranges[i].second; ++t) piece_of_tile[t] = (int)i;
    int bad = 0;
    for (int x = 0; x < To; ++x) for (int y = 0; y < x; ++y) if (adjS[(size_t)x * To + y] && piece_of_tile[x] >= 0 && piece_of_tile[y] >= 0 && piece_of_tile[x] != piece_of_tile[y]) { if (bad++ < 5) printf("  piece tiles %d and %d are coupled\n", y, x); }
    printf("couplings between different pieces: %d\n", bad);
  }
  DensePlan P;
  P.build_ordered(o.n_pose, To, o.dpos, o.nreal, adjS, o.piece_ranges, o.sep_ranges_by_level);
  int nchain = 0;
  for (auto& t : P.ftasks) nchain += (t.flags & kFusedChain) ? 1 : 0;
  printf("plan: %d tiles, %zu tasks (%d chains), touched %zu, flops %.3g, level_sync %d, task list replayed %.1f us\n", P.T, P.ftasks.size(), nchain, P.touched_tiles.size(), P.fused_flops, (int)P.bs_level_sync, P.est_makespan_us);
  if (getenv("DUMP_K")) {
    const int k0 = atoi(getenv("DUMP_K"));
    for (size_t t = 0; t < P.ftasks.size(); ++t) { const auto& f = P.ftasks[t]; if (f.k == k0 || f.k == k0 + 1) printf("ticket %zu: k %d ti %d tj %d flags %d need %d tot %d\n", t, f.k, f.ti, f.tj, f.flags, f.need_c, f.tot_c); }
  }
  return 0;
}
