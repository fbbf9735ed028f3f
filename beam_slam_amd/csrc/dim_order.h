// Host-side ordering of the reduced camera system PER TANGENT BLOCK (bsgpu_finalize.cpp; the tile machinery of dense_plan.h then runs
// on the order found here: DensePlan::build_ordered).
//
// Why: the tile-level nested dissection of dense_plan.h build() cuts the window into runs of natural 64-wide tiles, so a separator has to
// be as wide as the band IN TILES — three tiles, 192 pivots, on a visual-inertial window whose keyframes couple to their twelve
// neighbours through shared landmarks: a keyframe's 15 tangent dimensions sit together, and all of them go into the separator.  But only
// the 6 pose dimensions of a keyframe see the landmarks; its velocity and biases couple to the two neighbouring states alone (the IMU
// chain).  On the graph of tangent BLOCKS (3 dimensions each) the separator of the same cut is the pose blocks of twelve keyframes plus
// the velocity / bias blocks of ONE: 81 dimensions instead of 192 — and the factorisation's critical path is pivots-on-the-path.
//
// The dissection is generic (no knowledge of keyframes): the blocks are taken in natural (time) order, a cut position c splits them,
// and the separator is the lighter of {blocks before c with a neighbour at or after c} and {blocks at or after c with a neighbour
// before c}; blocks coupled to most of the set (an extrinsics block every pose touches, a dense prior) go into the separator first.
// A set is split when that shortens the estimated critical path (steps of 16 pivots + a hand-over per chain of three tiles); tiny
// separators are merged into their parent.  Supernodes (pieces, separators) are laid out as runs of whole tiles, padded at the end.
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace bsg {

struct DimOrder {
  // ---- input
  int n_pose = 0;
  std::vector<int> blk_t0, blk_w;      // tangent blocks of the reduced system in natural order: first tangent index, width
  std::vector<int> adj_ptr, adj;       // CSR adjacency over the blocks (symmetric, no self loops)
  // ---- output
  int T = 0;                           // tiles (without the rhs tile)
  std::vector<int> dpos;               // tangent index -> position in S
  std::vector<int> nreal;              // per tile
  std::vector<std::pair<int, int>> piece_ranges;                     // [begin, end) tiles
  std::vector<std::vector<std::pair<int, int>>> sep_ranges_by_level;   // [0] = deepest
  int n_nodes = 0, depth = 0;
  double est_path_us = 0.0;

  // nominal costs (microseconds) of the fused factorisation, from the task stamps of C2 (BSGPU_CHOL_PROBE, scripts/chol_probe.py): a
  // chain of m tiles takes t_chain0 (its tiles' loads) + steps x t_step[m], the hand-over from a chain to the next one on the path (flag,
  // loads, strip solve, product, turn, publication, flag) t_hop.  Measured on C2 with (5, 2, 2.1, 4.5, 8 .. 16): 152-153 us per
  // factorisation for every hand-over value; with t_step3 = 3.5 (two leaves of three tiles survive) 159.5.
  double t_hop_tile = 4.0;
  double t_chain0 = 5.0, t_step = 2.0, t_step2 = 2.1, t_step3 = 4.5, t_hop = 10.0;   // (t_step3: a chain of three tiles is bound by the MFMA rate of its ONE CU — and a leaf that long is the head of the critical path: priced so that the dissection avoids it)
  int max_depth = 5;
  double hub_frac = 0.6;
  int merge_dims = 24;     // separators up to this many dimensions are merged into their parent separator

  double node_cost(int dims) const {
    if (dims <= 0) return 0.0;
    const int tiles = (dims + 63) / 64, chains = (tiles + 2) / 3;
    if (chains > 1) return (chains - 1) * (t_chain0 + 12 * t_step3 + t_hop) + node_cost(dims - 192 * (chains - 1));
    const double ts = tiles == 1 ? t_step : tiles == 2 ? t_step2 : t_step3;
    // (a last tile of at most 16 real columns rides in the tasks of the tile before it — dense_plan.h kFusedExt —, any other second tile costs
    //  the path a hand-over of its own: 151 -> 138 us per factorisation on C2, whose separators are 78 dimensions wide)
    const bool appendix = tiles >= 2 && dims - 64 * (tiles - 1) <= 16;
    return t_chain0 + ts * ((dims + 15) / 16) + t_hop + ((tiles >= 2 && !appendix) ? t_hop_tile : 0.0);
  }

  struct Node { std::vector<int> verts; int parent = -1, depth = 0, dims = 0; bool is_sep = false; };
  std::vector<Node> nodes;
  std::vector<std::pair<int, std::vector<int>>> merges;   // (separator node, blocks that join it): applied at the end — a split further up may still be undone

  // returns the estimated critical path of the subtree built for V (blocks, ascending); nodes are appended to `nodes`
  double dissect(std::vector<int>& V, int parent, int depth_, std::vector<int>& pos) {
    int wV = 0;
    for (int v : V) wV += blk_w[v];
    const double cost_single = node_cost(wV);
    auto make_piece = [&]() {
      Node nd; nd.verts = V; nd.parent = parent; nd.depth = depth_; nd.dims = wV; nd.is_sep = false;
      nodes.push_back(std::move(nd));
      return cost_single;
    };
    if (depth_ >= max_depth || wV < 96 || V.size() < 4) return make_piece();
    // ---- hubs: blocks coupled to most of the set
    for (size_t i = 0; i < V.size(); ++i) pos[V[i]] = (int)i;
    std::vector<int> hubs, rest;
    {
      for (int v : V) {
        int deg = 0;
        for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) if (pos[adj[e]] >= 0) deg += blk_w[adj[e]];
        if (deg >= hub_frac * wV) hubs.push_back(v); else rest.push_back(v);
      }
    }
    for (int v : V) pos[v] = -1;
    if (rest.size() < 4) return make_piece();
    // ---- the cut: over `rest` in natural order
    const int n = (int)rest.size();
    for (int i = 0; i < n; ++i) pos[rest[i]] = i;
    std::vector<int> hi(n), lo(n), pw(n + 1, 0);
    for (int i = 0; i < n; ++i) {
      const int v = rest[i];
      int h = i, l = i;
      for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) { const int p = pos[adj[e]]; if (p >= 0) { h = std::max(h, p); l = std::min(l, p); } }
      hi[i] = h; lo[i] = l; pw[i + 1] = pw[i] + blk_w[v];
    }
    for (int i = 0; i < n; ++i) pos[rest[i]] = -1;
    std::vector<int> dl(n + 2, 0), dr(n + 2, 0);   // difference arrays over the cut position c = 1 .. n-1 (c = number of blocks before the cut)
    for (int i = 0; i < n; ++i) {
      if (hi[i] > i) { dl[i + 1] += blk_w[rest[i]]; dl[hi[i] + 1] -= blk_w[rest[i]]; }   // in L(c) for i < c <= hi[i]
      if (lo[i] < i) { dr[lo[i] + 1] += blk_w[rest[i]]; dr[i + 1] -= blk_w[rest[i]]; }   // in R(c) for lo[i] < c <= i
    }
    const int wR = pw[n];
    int wH = 0;
    for (int v : hubs) wH += blk_w[v];
    // the cut whose ESTIMATED path is shortest: this separator, then the longer of the two sides — each side taken as splitting on with
    // separators of the same weight as long as that pays (the real recursion below decides; this only ranks the cut positions)
    auto side_est = [&](int w, int ws) {
      double best = node_cost(w), acc = 0.0;
      int cur = w;
      for (int l = depth_ + 1; l < max_depth && ws > 0; ++l) {
        cur = (cur - ws) / 2;
        if (cur <= 0) break;
        acc += node_cost(ws);
        best = std::min(best, acc + node_cost(cur));
      }
      return best;
    };
    int best_c = -1, best_side = 0;
    double best_est = 1e300;
    int accL = 0, accR = 0;
    for (int c = 1; c < n; ++c) {
      accL += dl[c]; accR += dr[c];
      const double frac = (double)pw[c] / (double)wR;
      if (frac < 0.15 || frac > 0.85) continue;
      const int side = accL <= accR ? 0 : 1, ws = side == 0 ? accL : accR;
      const int wa = pw[c] - (side == 0 ? ws : 0), wb = (wR - pw[c]) - (side == 1 ? ws : 0);
      if (wa <= 0 || wb <= 0) continue;
      const double est = node_cost(ws + wH) + std::max(side_est(wa, ws), side_est(wb, ws));
      if (est < best_est) { best_est = est; best_c = c; best_side = side; }
    }
    if (best_c < 0) return make_piece();
    std::vector<int> S(hubs), A, Bv;
    for (int i = 0; i < n; ++i) {
      const bool in_sep = best_side == 0 ? (i < best_c && hi[i] >= best_c) : (i >= best_c && lo[i] < best_c);
      if (in_sep) S.push_back(rest[i]);
      else (i < best_c ? A : Bv).push_back(rest[i]);
    }
    if (A.empty() || Bv.empty()) return make_piece();
    std::sort(S.begin(), S.end());
    int wS = 0;
    for (int v : S) wS += blk_w[v];
    // ---- build both sides, keep the split if it shortens the path
    const size_t mark = nodes.size(), mark_m = merges.size();
    int sep_node = -1, child_parent = parent, child_depth = depth_;
    const bool merge_up = wS <= merge_dims && parent >= 0;   // (a tiny separator joins its parent: one hand-over less on the path)
    if (wS > 0 && !merge_up) {
      Node nd; nd.verts = S; nd.parent = parent; nd.depth = depth_; nd.dims = wS; nd.is_sep = true;
      nodes.push_back(std::move(nd));
      sep_node = (int)nodes.size() - 1;
      child_parent = sep_node; child_depth = depth_ + 1;
    }
    const double cA = dissect(A, child_parent, child_depth, pos);
    const double cB = dissect(Bv, child_parent, child_depth, pos);
    double cost_split = std::max(cA, cB);
    if (sep_node >= 0) cost_split += node_cost(wS);
    else if (merge_up) cost_split += t_step * ((wS + 15) / 16);
    if (cost_split + 1e-9 >= cost_single) {   // not worth it: V stays one piece
      nodes.resize(mark); merges.resize(mark_m);
      return make_piece();
    }
    if (merge_up && wS > 0) merges.push_back({parent, S});
    return cost_split;
  }

  // A separator joins its parent separator when ONE chain over both is shorter than two chains and the hand-over between them (round 5): the
  // recursion above only ever cuts a set in two, so a window of the reference's size (300 reduced dimensions) came out as a piece below a
  // 48-dimensional separator below another 48-dimensional separator — 9 + 5 + 9 us of the factorisation's 53 for six 16-pivot steps, which a
  // single two-tile chain does in 15.  Moving a supernode's blocks up into an ancestor keeps every coupling inside a supernode or towards an
  // ancestor.  down[] = the estimated path from the leaves up to and including a node; a merge is taken when it shortens the parent's.
  bool absorb = true;
  void absorb_separators() {
    const int n = (int)nodes.size();
    for (bool changed = true; changed;) {
      changed = false;
      std::vector<double> down(n, 0.0);
      for (int i = n - 1; i >= 0; --i) {   // (a node's children come after it)
        if (nodes[i].dims <= 0) continue;
        down[i] += node_cost(nodes[i].dims);
        const int p = nodes[i].parent;
        if (p >= 0) down[p] = std::max(down[p], down[i]);   // (below(p) so far; p's own cost is added when the loop reaches it)
      }
      for (int c = n - 1; c >= 0 && !changed; --c) {
        Node& C = nodes[c];
        if (!C.is_sep || C.dims <= 0 || C.parent < 0) continue;
        Node& P = nodes[C.parent];
        if (!P.is_sep || P.dims <= 0) continue;
        double below_c = 0.0, other = 0.0;
        for (int k = 0; k < n; ++k) {
          if (nodes[k].dims <= 0) continue;
          if (nodes[k].parent == c) below_c = std::max(below_c, down[k]);
          else if (nodes[k].parent == C.parent && k != c) other = std::max(other, down[k]);
        }
        const double before = std::max(below_c + node_cost(C.dims), other) + node_cost(P.dims);
        const double after = std::max(below_c, other) + node_cost(P.dims + C.dims);
        if (after + 1e-9 >= before) continue;
        P.verts.insert(P.verts.end(), C.verts.begin(), C.verts.end());
        P.dims += C.dims;
        for (int k = 0; k < n; ++k) if (nodes[k].parent == c) nodes[k].parent = C.parent;
        C.verts.clear(); C.dims = 0;
        changed = true;
      }
    }
    // depths again (a node one below its parent), and the estimate
    for (int i = 0; i < n; ++i) nodes[i].depth = nodes[i].parent < 0 ? 0 : nodes[nodes[i].parent].depth + 1;
    std::vector<double> down(n, 0.0);
    double est = 0.0;
    for (int i = n - 1; i >= 0; --i) {
      if (nodes[i].dims <= 0) continue;
      down[i] += node_cost(nodes[i].dims);
      if (nodes[i].parent >= 0) down[nodes[i].parent] = std::max(down[nodes[i].parent], down[i]); else est = std::max(est, down[i]);
    }
    est_path_us = est;
  }

  void build() {
    const int nbk = (int)blk_t0.size();
    nodes.clear();
    std::vector<int> V(nbk), pos(nbk, -1);
    for (int i = 0; i < nbk; ++i) V[i] = i;
    merges.clear();
    est_path_us = nbk ? dissect(V, -1, 0, pos) : 0.0;
    for (const auto& mg : merges) {
      Node& P = nodes[mg.first];
      P.verts.insert(P.verts.end(), mg.second.begin(), mg.second.end());
      for (int v : mg.second) P.dims += blk_w[v];
    }
    if (absorb) absorb_separators();
    for (Node& nd : nodes) std::sort(nd.verts.begin(), nd.verts.end());
    n_nodes = (int)nodes.size();
    // ---- layout: the pieces in the order they were made (left to right), then the separators, deepest first
    depth = 0;
    for (const Node& nd : nodes) if (nd.is_sep && nd.dims > 0) depth = std::max(depth, nd.depth + 1);
    dpos.assign(std::max(1, n_pose), 0);
    nreal.clear(); piece_ranges.clear();
    sep_ranges_by_level.assign(depth, {});
    int tile = 0;
    auto place = [&](const Node& nd) {
      const int t0 = tile;
      int at = tile * 64;
      for (int v : nd.verts) for (int k = 0; k < blk_w[v]; ++k) dpos[blk_t0[v] + k] = at++;
      const int tiles = std::max(1, (nd.dims + 63) / 64);
      for (int t = 0; t < tiles; ++t) nreal.push_back(std::min(64, nd.dims - 64 * t));
      tile += tiles;
      return std::make_pair(t0, tile);
    };
    for (const Node& nd : nodes) if (!nd.is_sep && nd.dims > 0) piece_ranges.push_back(place(nd));
    for (int d = depth - 1; d >= 0; --d)
      for (const Node& nd : nodes) if (nd.is_sep && nd.depth == d && nd.dims > 0) sep_ranges_by_level[depth - 1 - d].push_back(place(nd));
    if (tile == 0) { nreal.push_back(0); tile = 1; piece_ranges.push_back({0, 1}); }   // (an empty reduced system still has one tile)
    T = tile;
  }
};

}  // namespace bsg
