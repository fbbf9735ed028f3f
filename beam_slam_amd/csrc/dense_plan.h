// Host-side plan of the tiled (64 x 64) Cholesky of the reduced camera system: tile ordering, exact
// tile-level symbolic factorisation, and a step schedule that runs independent panels concurrently.
//
// A fixed-lag window gives a block-BANDED reduced system (a keyframe only shares landmarks / IMU factors
// with its neighbours), and a banded Cholesky in natural order is one long dependent chain of panels.
// Ordering the tiles by nested dissection of that chain — [piece 0][piece 1 reversed] ... [separators] —
// makes the pieces independent sub-chains that factor concurrently (one launch handles one panel of every
// piece), and only the separators (as wide as the band) come last.  The ordering is a permutation of whole
// natural tiles; it is internal to the solver: the variable index the C-ABI exposes does not change.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace bsg {

struct PanelDesc {  // device-visible
  int k;          // S tile index of the panel (diagonal tile k)
  int row_off;    // into the flat row-tile list
  int n_rows;     // active row tiles below the diagonal (S tile indices, ascending)
  int final_mask; // bit q (q < 31): row tile q receives its LAST update in this step; the diagonal workgroup (q, q) of the panel that
                  // arrives last among that tile's updaters of the step (tile_sync) factors it right there (look-ahead)
  int self_potrf;   // tile (k,k) is not factored by a look-ahead (it is never updated: the head of a piece): the panel's workgroups do it
  int shared_mask;  // bit q: row tile q of this panel is also a row tile of another panel of the same step, so tiles
                    // (i, j) with both bits set are accumulated with atomics (two panels update them concurrently)
};

// One unit of work of the fused (single-launch) factorisation, k_chol.hip chol_fused_kernel: workgroup-sized, pulled from a
// device-side queue in list order.  The list is a topological order of the tile dependencies (every counter a task waits
// for is advanced by tasks EARLIER in the list), so a queue that hands tasks out in order cannot deadlock whatever the
// residency of the grid.  Two kinds of task:
//   chain   a run of <= kChainMaxTiles consecutive tiles of the ordering (a piece, a separator, or a part of one) is factored as
//           ONE dense matrix inside one workgroup (chol_chain.h): potrf, the solves and the updates among its own tiles never leave
//           the CU.  It waits until every tile of the chain has received its updates from outside (tile_tot), and sets potrf_done[k]
//           for each of its tiles k as the tile's column (L, and the tile inverse W) leaves for memory.
//   update  (k; ti, tj), ti >= tj outside the chain of k (tj may be a LATER tile of k's chain):
//           C(ti, tj) -= X_ti X_tj^T with X_t = A(t, k) L_kk^-T.  X of a tile of k's own chain is read from the factor (the chain
//           wrote it); X of an outside tile is either solved here, or — where that is off the critical path — read from what the
//           DIAGONAL task (k; t, t) of that tile published (one solve per (k, t) instead of one per task).
struct FusedTask {  // device-visible, 32 bytes
  int k;        // panel (S tile index of the diagonal tile); chain task: its first tile
  int ti, tj;   // row tiles of the update C(ti, tj) -= X_ti X_tj^T, ti >= tj; chain task: number of tiles | bit mask of its non-zero tiles
  int flags;    // kFused*
  int tot_i;    // number of updates tile (ti, k) receives in the whole factorisation: it is final (readable) at that count,
                // and X_ti is published (kFusedPublishX of the diagonal task) at that count + 1
  int tot_j;    // ... tile (tj, k)
  int need_c;   // number of updates of tile (ti, tj) by earlier tasks: this task's turn comes at exactly that count
                // (updates of one tile are applied in list order: no atomics, bit-reproducible factor); -1: no update (rhs x rhs)
  int tot_c;    // total number of updates of tile (ti, tj)
};
constexpr int kFusedChain = 1;      // a chain task
constexpr int kFusedXiLp = 2;       // X_ti is read from the factor once the diagonal task (k; ti, ti) has published it
constexpr int kFusedXjLp = 4;       // ... X_tj
constexpr int kFusedXjChain = 8;    // tj belongs to the chain of k: X_tj = L(tj, k) is out when potrf_done[k] is set
constexpr int kFusedPublishX = 16;  // (diagonal task) X_ti goes to the factor write-through and the tile's counter is bumped once more
constexpr int kFusedExt = 32;       // panel k carries its chain's APPENDIX tile k + 1 (a last tile of at most kFusedExtCols real columns): the task also
                                    // forms X(., k + 1) = (A(., k + 1) - X(., k) L(k + 1, k)^T) W_{k+1}^T of its row tiles and its product runs over both —
                                    // the appendix has no update tasks of its own, and the tiles (., k + 1) are not updated by panel k
constexpr int kFusedExtCols = 16;
constexpr int kFusedDiagAdd = 64;   // k = ti = tj = a tile: the LM diagonal of its real columns is added to S (LmDiag, bsgpu_internal.h) — the FIRST update of
                                    // every diagonal tile (need_c = 0), so the chains and the other updates of the tile come after it by the tile's counter
constexpr int kFusedRider = 128;    // k = unit of independent work carried by the launch (the step's gradient norms): nothing waits for it
constexpr int kFusedSplit = 256;    // ONE K-CHUNK of the LAST update of a tile inside a chain (round 5): what a chain — the critical path — waits for last
                                    // was one workgroup, MFMA-bound on its one CU (solve 1.1-2.1 us + rank-64 product 1.7 us, 4.4 + 3 with an appendix).
                                    // Such an update is dealt out to 4 (5 with an appendix) workgroups: chunk p forms the 16 columns 16 p .. of X_ti / X_tj
                                    // (W = L_kk^-1 is lower triangular: K = 16 (p + 1)) and ADDS their rank-16 product to the tile with FP64 atomics once
                                    // every earlier update of the tile has been published (need_c = that count: the chunks share one turn, the last); the
                                    // appendix chunk forms the appendix's 16 columns.  tot_c = chunk | number of chunks << 8.  The tile's counter counts the
                                    // chunks in its upper half: tile_tot = updates with a turn of their own | chunks << 16.  (The order in which the chunks'
                                    // sums reach a tile varies: the factor is reproducible to rounding, like the assembled system it factors.)
constexpr int kFusedRowSeg = 512;   // SEVERAL updates of one row tile by one panel in one task (ftasks_rows; k_chol.hip chol_fused_rowseg): tj = index of the segment's first
                                    // (tile, updates of (tile, k)) pair in frow_items, tot_j = the number of pairs; every product is counted at its tile like a task
constexpr int kFusedRowSegMax = 8;  // pairs per segment, at most
constexpr int kFusedSplitChunks = 4;
constexpr int kChainMaxTiles = 3;   // == chain::kChainMaxTiles (chol_chain.h)

struct DensePlan {
  int n_pose = 0, T = 0 /* real tiles */, npad = 0, rhs_row = 0;
  std::vector<int> perm;       // natural tile -> S tile (build(): the tile-level ordering; empty when the order was given per dimension)
  std::vector<int> dpos;       // natural tangent index (0 .. n_pose-1) -> position in S (solver order); what every kernel that addresses S uses
  std::vector<int> inat;       // position in S (0 .. npad-1) -> natural tangent index, or -1 (padding, the rhs tile)
  std::vector<int> nreal;      // per S tile: number of real columns (64, or n_pose % 64 for the partial tile)
  std::vector<int> touched_tiles;  // every tile (i * (T + 1) + j, both triangles) an assembly or the factorisation may write: the
                                   // structural blocks, their fill, the rhs row and column, the diagonal — what a step has to clear
  std::vector<int> tile_sync;  // device image, 2 x (T + 1): [expected arrivals per tile | arrival counters (zero; the last arriver resets its own)]
  std::vector<int> rows_flat;  // row tiles of every panel (includes the rhs tile T)
  std::vector<PanelDesc> panels;        // in schedule order
  std::vector<int> step_off;            // panels[step_off[s] .. step_off[s+1]) run in one launch
  std::vector<int> step_maxrows;
  std::vector<int> potrf_before_step_off, potrf_tiles;  // standalone potrf launches: tiles to factor before step s
  int n_chains = 1;
  int n_leaf_tiles = 0;   // tiles ordered first as one-panel pieces (build(): leaf)
  // back-substitution: GROUPS of chains, one launch per group; a chain is a run of consecutive S tiles [begin, end) walked
  // by one workgroup from its last tile down.  Root separator first, then the separators level by level (those of one
  // level are independent), then all pieces at once.  Every row tile of a panel lies later in its own chain or in an
  // earlier group (checked; if the structure does not allow it the groups fall back to the reverse step schedule).
  std::vector<int> panel_of_tile;                 // S tile -> index into panels
  std::vector<int> chain_begin, chain_end;        // all chains, group after group
  std::vector<int> bs_group_off;                  // chains [bs_group_off[g], bs_group_off[g+1]) run in launch g
  int n_pieces = 1;
  // per S tile, what the back-substitution needs of its panel in ONE contiguous record (kBsDescInts ints): n_rows, offset of
  // its row list in rows_flat, number of real columns, and the first kBsDescRows row tiles themselves — a chain's workgroup
  // fetches the records of its tiles in one coalesced round instead of three dependent ones (tile -> panel -> row list)
  static constexpr int kBsDescRows = 16, kBsDescInts = 3 + kBsDescRows;
  std::vector<int> bs_desc;
  // fused single-launch factorisation (k_chol.hip chol_fused_kernel)
  std::vector<FusedTask> ftasks;
  std::vector<int> tile_tot;          // (T+1)^2: number of update TASKS per tile (what a chain waits for before it reads its tiles): updates that take a turn of their
                                      // own on the tile | kFusedSplit chunks << 16
  int split_depth = 2;                // how many of the LAST updates of a tile inside a chain are dealt out as kFusedSplit chunks; 0: none (finalize: BSGPU_CHOL_SPLIT)
  int n_split_chunks = 0;
  std::vector<int> fchain_begin, fchain_len, fchain_of_tile;   // the chains of the fused factorisation
  std::vector<int> fext_of;           // T+1: the appendix tile panel k carries (kFusedExt), or -1
  std::vector<FusedTask> ftasks_plain;   // ftasks without the diagonal / rider tasks (empty: the plan has none, ftasks is that list)
  std::vector<int> tile_tot_plain;
  std::vector<FusedTask> ftasks_bulk;    // ... and without the K-chunks either: what a launch takes that is bound by the NUMBER of its workgroups, not by one
  std::vector<int> tile_tot_bulk;        // window's critical path (many windows side by side: bsgpu_batch.cpp); empty: the plan has no chunks
  // ... and with the updates that read two published X gathered into ROW SEGMENTS (kFusedRowSeg: one task per row tile and panel, up to kFusedRowSegMax
  // products): built from the bulk list (or, without chunks, from the plain one; or from ftasks itself) — frows_src says which, its tile_tot is that list's.
  // For launches WITHOUT turns only (the products of a segment reach their tiles in the segment's order, not in the list's).
  std::vector<FusedTask> ftasks_rows;
  std::vector<int> frow_items;           // (tile tj, number of updates of tile (tj, k)) pairs, a segment's pairs side by side
  int frows_src = -1;                    // 2: ftasks_bulk, 1: ftasks_plain, 0: ftasks; -1: no segments (no list)
  // One pass over the source list.  An update (k; ti, tj) that reads both X (kFusedXiLp | kFusedXjLp) joins the open segment of (k, ti) if the diagonal
  // task that publishes X_tj has its ticket BEFORE that segment's (a workgroup only ever waits for tickets taken earlier: k_chol.hip), else it opens a
  // new one at its own place.  A segment sits where its FIRST member sat: its members' products only ever come earlier than in the source list, so every
  // reader of their tiles still has them in front of it; what a member waits for — X_ti (the first member's diagonal task), X_tj (checked) — is in front
  // of the segment.
  void build_row_segments() {
    ftasks_rows.clear(); frow_items.clear(); frows_src = -1;
    const std::vector<FusedTask>& src = !ftasks_bulk.empty() ? ftasks_bulk : !ftasks_plain.empty() ? ftasks_plain : ftasks;
    if (src.empty()) return;
    frows_src = !ftasks_bulk.empty() ? 2 : !ftasks_plain.empty() ? 1 : 0;
    const int N = T + 1;
    std::vector<int> pub_pos((size_t)N * N, -1);       // (tile, panel) -> place of the diagonal task that publishes X(tile, panel) in the new list
    std::vector<int> open_seg((size_t)N * N, -1);      // (row tile, panel) -> place of its open segment
    std::vector<std::vector<int>> seg_items;           // per segment task (by its place; empty for other tasks)
    for (const FusedTask& f : src) {
      const bool member = !(f.flags & (kFusedChain | kFusedRider | kFusedDiagAdd | kFusedSplit | kFusedXjChain | kFusedPublishX)) && (f.flags & kFusedXiLp) &&
                          (f.flags & kFusedXjLp) && f.ti != f.tj && f.need_c >= 0;
      if (!member) {
        if ((f.flags & kFusedPublishX) && !(f.flags & (kFusedChain | kFusedSplit))) pub_pos[(size_t)f.ti * N + f.k] = (int)ftasks_rows.size();
        ftasks_rows.push_back(f); seg_items.emplace_back();
        continue;
      }
      const size_t key = (size_t)f.ti * N + f.k;
      const int os = open_seg[key], pp = pub_pos[(size_t)f.tj * N + f.k];
      if (os >= 0 && pp >= 0 && pp < os && (int)seg_items[os].size() < 2 * kFusedRowSegMax) {
        seg_items[os].push_back(f.tj); seg_items[os].push_back(f.tot_j);
        continue;
      }
      FusedTask g = f;
      g.flags = kFusedRowSeg | (f.flags & kFusedExt);
      g.tj = 0; g.tot_j = 0; g.need_c = 0; g.tot_c = 0;
      open_seg[key] = (int)ftasks_rows.size();
      ftasks_rows.push_back(g); seg_items.emplace_back();
      seg_items.back().push_back(f.tj); seg_items.back().push_back(f.tot_j);
    }
    for (size_t t = 0; t < ftasks_rows.size(); ++t) {
      if (!(ftasks_rows[t].flags & kFusedRowSeg)) continue;
      ftasks_rows[t].tj = (int)frow_items.size() / 2;
      ftasks_rows[t].tot_j = (int)seg_items[t].size() / 2;
      frow_items.insert(frow_items.end(), seg_items[t].begin(), seg_items[t].end());
    }
  }
  bool allow_ext = true;              // (finalize: BSGPU_CHOL_EXT=0 plans every tile's panel by itself)
  bool diag_tasks = false;            // one kFusedDiagAdd task per tile at the head of the list
  int rider_tasks = 0;                // kFusedRider tasks behind them
  int fused_sync_words = 0;   // ints of device scratch: ([queue head | abort | exited workgroups | potrf_done (T+1) | update counts (T+1)^2]) x 16
  double est_makespan_us = 0.0;   // the factorisation's span by the ticket order's own replay of the task list (nominal durations: build_fused_tasks)
  double fused_flops = 0.0;   // FP64 flops of the planned factorisation (trsm + rank-64 updates + potrf of every touched tile), for the MFMA roofline
  // solve offsets
  inline int spos(int j) const { return dpos[j]; }

  // adj: T x T symmetric tile adjacency in NATURAL tile order (adj[i*T+j] != 0 iff block (i,j) of S is structurally non-zero)
  // leaf (optional, T flags): tiles that are coupled to no other leaf tile — a window's inverse-depth landmarks, thousands of scalar
  // blocks that touch a few keyframes each.  They are ordered FIRST: eliminating them is the landmark Schur complement, carried out by
  // the tile machinery (each is the head of a one-panel piece); the nested dissection below then orders the remaining (core) tiles on
  // their adjacency INCLUDING the fill the leaves leave behind.  Without this a window with 20 000 such landmarks would be ordered as
  // one banded chain with the landmarks last — poses eliminated into a dense 20 000-dimensional block.
  void build(int n_pose_, const std::vector<uint8_t>& adj_in, int max_chains, int min_piece_w = 1 /* minimum piece length in units of the band width */,
             bool allow_shared = true /* panels of one step may update the same tiles (atomics) */, const std::vector<uint8_t>* leaf = nullptr) {
    n_pose = n_pose_;
    T = (n_pose + 63) / 64;
    npad = (T + 1) * 64;
    rhs_row = T * 64;
    const std::vector<uint8_t>& adj = adj_in;
    // ---- leaf tiles (validated: mutually uncoupled) and the core sub-chain
    std::vector<int> leaf_tiles, core;
    {
      std::vector<uint8_t> is_leaf(T, 0);
      if (leaf && (int)leaf->size() == T) {
        bool ok = true;
        for (int i = 0; i < T && ok; ++i) if ((*leaf)[i]) for (int j = 0; j < T; ++j) if (j != i && (*leaf)[j] && adj[(size_t)i * T + j]) { ok = false; break; }
        if (ok) for (int i = 0; i < T; ++i) is_leaf[i] = (*leaf)[i] ? 1 : 0;
      }
      for (int i = 0; i < T; ++i) (is_leaf[i] ? leaf_tiles : core).push_back(i);
      if (core.empty()) { core = leaf_tiles; leaf_tiles.clear(); }
    }
    const int Tc = (int)core.size(), n_leaf = (int)leaf_tiles.size();
    n_leaf_tiles = n_leaf;
    // core adjacency in core index space, with the fill of the leaf elimination (two core tiles that share a leaf become coupled)
    std::vector<uint8_t> adjc((size_t)Tc * Tc, 0);
    {
      std::vector<int> core_of(T, -1);
      for (int i = 0; i < Tc; ++i) core_of[core[i]] = i;
      for (int i = 0; i < Tc; ++i) for (int j = 0; j < Tc; ++j) adjc[(size_t)i * Tc + j] = adj[(size_t)core[i] * T + core[j]];
      std::vector<int> nb;
      for (int t : leaf_tiles) {
        nb.clear();
        for (int j = 0; j < T; ++j) if (core_of[j] >= 0 && (adj[(size_t)t * T + j] || adj[(size_t)j * T + t])) nb.push_back(core_of[j]);
        for (int a : nb) for (int b : nb) adjc[(size_t)a * Tc + b] = 1;
      }
    }
    // ---- ordering of the core: nested dissection of a banded chain
    perm.assign(T, 0);
    // band width used to cut the chain into pieces and separators: the distance below which 95 % of the coupled tile pairs lie,
    // not the maximum — one long feature track or a loop closure couples two far-apart keyframes, and sizing the separators
    // for it would leave a single piece (a serial chain of T panels).  The symbolic factorisation below is exact for ANY
    // ordering, so couplings wider than w only add their own fill and dependencies where they occur.
    int w = 0;
    {
      std::vector<int> dist;
      for (int i = 0; i < Tc; ++i) for (int j = 0; j < i; ++j) if (adjc[(size_t)i * Tc + j]) dist.push_back(i - j);
      if (!dist.empty()) {
        std::sort(dist.begin(), dist.end());
        w = dist[std::min(dist.size() - 1, (size_t)(0.95 * (double)dist.size()))];
      }
    }
    std::vector<int> order;  // S order: list of natural tiles
    std::vector<std::pair<int, int>> piece_ranges;                    // S tile ranges of the pieces
    std::vector<std::vector<std::pair<int, int>>> sep_ranges_by_level;  // S tile ranges of the separators, per level
    std::vector<std::pair<int, int>> leaf_ranges;                     // S tile ranges (one tile each) of the leaf tiles
    for (int t : leaf_tiles) { leaf_ranges.push_back({(int)order.size(), (int)order.size() + 1}); order.push_back(t); }
    int chains = 1;
    if (max_chains > 1 && w >= 1) while (chains * 2 <= max_chains && Tc >= (chains * 2) * min_piece_w * w + (chains * 2 - 1) * w) chains *= 2;
    n_chains = chains;
    if (chains == 1) {
      piece_ranges.push_back({(int)order.size(), (int)order.size() + Tc});
      for (int i = 0; i < Tc; ++i) order.push_back(core[i]);
    } else {
      // pieces p = 0..chains-1 separated by chains-1 separators of w tiles
      const int n_sep = chains - 1;
      const int body = Tc - n_sep * w;
      std::vector<int> piece_len(chains, body / chains);
      for (int i = 0; i < body % chains; ++i) piece_len[i]++;
      std::vector<std::pair<int, int>> pieces, seps;  // [begin, end) core tiles
      int pos = 0;
      for (int p = 0; p < chains; ++p) {
        pieces.push_back({pos, pos + piece_len[p]});
        pos += piece_len[p];
        if (p < n_sep) { seps.push_back({pos, pos + w}); pos += w; }
      }
      // a piece is ordered so that the end adjacent to its (higher-level) separator comes last; pieces with
      // separators on both sides are interior: natural order keeps the right neighbour last, the left separator
      // then sees fill along the piece (still correct: the symbolic factorisation below is exact)
      for (int p = 0; p < chains; ++p) {
        piece_ranges.push_back({(int)order.size(), (int)order.size() + piece_len[p]});
        const bool reverse = (p == chains - 1) && chains > 1;  // last piece: its only separator is on the left
        if (!reverse) for (int t = pieces[p].first; t < pieces[p].second; ++t) order.push_back(core[t]);
        else for (int t = pieces[p].second - 1; t >= pieces[p].first; --t) order.push_back(core[t]);
      }
      // separators last, lowest level (most local) first: odd-indexed separators of the recursive bisection
      // are the deepest; order them by increasing "level" so that the root separator is eliminated last
      std::vector<int> sep_level(n_sep, 0);
      for (int i = 0; i < n_sep; ++i) { int lvl = 0, x = i + 1; while ((x & 1) == 0) { x >>= 1; ++lvl; } sep_level[i] = lvl; }
      int max_lvl = 0;
      for (int l : sep_level) max_lvl = std::max(max_lvl, l);
      sep_ranges_by_level.assign(max_lvl + 1, {});
      for (int lvl = 0; lvl <= max_lvl; ++lvl)
        for (int i = 0; i < n_sep; ++i) if (sep_level[i] == lvl) {
          sep_ranges_by_level[lvl].push_back({(int)order.size(), (int)order.size() + (seps[i].second - seps[i].first)});
          for (int t = seps[i].first; t < seps[i].second; ++t) order.push_back(core[t]);
        }
    }
    for (int s = 0; s < T; ++s) perm[order[s]] = s;
    nreal.assign(T + 1, 64);
    if (n_pose % 64) nreal[perm[T - 1]] = n_pose % 64;
    nreal[T] = 0;
    dpos.assign(std::max(1, n_pose), 0);
    inat.assign(npad, -1);
    for (int j = 0; j < n_pose; ++j) { dpos[j] = perm[j >> 6] * 64 + (j & 63); inat[dpos[j]] = j; }
    // structure in S order (tile T = rhs tile, coupled to every panel)
    const int N = T + 1;
    std::vector<uint8_t> B((size_t)N * N, 0);
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) if (adj[(size_t)i * T + j]) B[(size_t)perm[i] * N + perm[j]] = 1;
    finish(B, piece_ranges, sep_ranges_by_level, leaf_ranges, allow_shared);
  }

  // A window that does not take the tiled factorisation at all (a pose graph on the block-sparse PCG): sizes and the identity position
  // tables only.  The full plan of such a system is worthless and enormous — C4's 469 tiles fill in completely: 17 million tasks, four
  // seconds of finalize() (measured, round 4; the plan had been built for every window since round 1).
  void build_skeleton(int n_pose_) {
    *this = DensePlan();
    n_pose = n_pose_; T = (n_pose + 63) / 64; npad = (T + 1) * 64; rhs_row = T * 64;
    perm.resize(T);
    for (int t = 0; t < T; ++t) perm[t] = t;
    nreal.assign(T + 1, 64);
    if (n_pose % 64 && T > 0) nreal[T - 1] = n_pose % 64;
    nreal[T] = 0;
    dpos.assign(std::max(1, n_pose), 0);
    inat.assign(npad, -1);
    for (int j = 0; j < n_pose; ++j) { dpos[j] = j; inat[j] = j; }
    step_off.assign(1, 0);
    bs_group_off.assign(1, 0);
    potrf_before_step_off.assign(1, 0);
    tile_sync.assign(2 * (size_t)(T + 1), 0);
    bs_desc.assign((size_t)std::max(1, T) * kBsDescInts, 0);
    panel_of_tile.assign(T, 0);
  }

  // The order is GIVEN, per dimension (bsgpu_finalize.cpp: dim_order.h — a nested dissection of the block graph of the reduced system whose
  // separators are sets of tangent blocks, not runs of natural tiles): dpos_[j] = position of tangent index j in S; the supernodes (pieces,
  // separators) are runs of whole tiles, each padded to a multiple of 64 at its END (nreal_[t] < 64 on a supernode's last tile: unit
  // pivots, as for the window's last tile in build()).  adjS: T_ x T_ symmetric tile adjacency IN S ORDER.  sep_ranges_by_level[0] = the
  // deepest separators ... back() = the root(s); every row tile of a panel must lie in its own supernode or in an ancestor's (checked by
  // the back-substitution plan below, which falls back to the reverse step schedule otherwise).
  void build_ordered(int n_pose_, int T_, const std::vector<int>& dpos_, const std::vector<int>& nreal_, const std::vector<uint8_t>& adjS,
                     const std::vector<std::pair<int, int>>& piece_ranges, const std::vector<std::vector<std::pair<int, int>>>& sep_ranges_by_level,
                     bool allow_shared = true) {
    n_pose = n_pose_; T = T_; npad = (T + 1) * 64; rhs_row = T * 64;
    perm.clear();
    n_leaf_tiles = 0;
    n_chains = (int)piece_ranges.size();
    dpos = dpos_;
    if (dpos.empty()) dpos.assign(1, 0);
    inat.assign(npad, -1);
    for (int j = 0; j < n_pose; ++j) inat[dpos[j]] = j;
    nreal = nreal_;
    nreal.resize(T + 1, 64);
    nreal[T] = 0;
    const int N = T + 1;
    std::vector<uint8_t> B((size_t)N * N, 0);
    for (int i = 0; i < T; ++i) for (int j = 0; j < T; ++j) if (adjS[(size_t)i * T + j]) B[(size_t)i * N + j] = 1;
    finish(B, piece_ranges, sep_ranges_by_level, {}, allow_shared);
  }

 private:
  // everything below the ordering: exact tile-level symbolic factorisation, step schedule, task list of the fused factorisation,
  // back-substitution plan.  B: N x N (N = T + 1) structure in S order (the rhs tile's couplings are added here).
  void finish(std::vector<uint8_t>& B, const std::vector<std::pair<int, int>>& piece_ranges,
              const std::vector<std::vector<std::pair<int, int>>>& sep_ranges_by_level, const std::vector<std::pair<int, int>>& leaf_ranges,
              bool allow_shared) {
    const int N = T + 1;
    // ---- exact tile-level symbolic factorisation in S order (tile T = rhs tile, coupled to every panel)
    for (int k = 0; k < T; ++k) { B[(size_t)T * N + k] = 1; B[(size_t)k * N + T] = 1; B[(size_t)k * N + k] = 1; }
    std::vector<std::vector<int>> rows(T);
    for (int k = 0; k < T; ++k) {
      for (int t = k + 1; t < N; ++t) if (B[(size_t)t * N + k]) rows[k].push_back(t);
      for (int a : rows[k]) for (int b : rows[k]) B[(size_t)a * N + b] = 1;  // fill
    }
    touched_tiles.clear();
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) if (B[(size_t)i * N + j] || B[(size_t)j * N + i] || i == j) touched_tiles.push_back(i * N + j);
    // ---- schedule: panel k depends on every panel j < k with k in rows(j); two panels sharing a row tile
    // (they would update the same C tiles) must not share a step
    std::vector<int> ready(T, 0);  // earliest step
    std::vector<int> step_of(T, -1);
    std::vector<std::vector<int>> steps;
    std::vector<std::vector<uint8_t>> step_rows;  // per step: which row tiles are written
    for (int k = 0; k < T; ++k) {
      int s = ready[k];
      while (true) {
        if (s >= (int)steps.size()) { steps.push_back({}); step_rows.push_back(std::vector<uint8_t>(N, 0)); }
        bool conflict = step_rows[s][k] != 0;   // its own diagonal tile / column is still being updated in this step
        // (the rhs tile T is shared by every panel: concurrent panels write disjoint column ranges of its row
        //  and only its never-used diagonal tile is written twice)
        if (!allow_shared) for (int t : rows[k]) if (t < T && step_rows[s][t]) conflict = true;
        if (allow_shared && (int)steps[s].size() >= 16) conflict = true;   // descriptor table of a launch (k_chol.hip: kStepMaxPanels)
        if (!conflict) break;
        ++s;
      }
      step_of[k] = s;
      steps[s].push_back(k);
      for (int t : rows[k]) if (t < T) { if (step_rows[s][t] < 255) step_rows[s][t]++; ready[t] = std::max(ready[t], s + 1); }
    }
    // look-ahead: a tile is final once the last step that updates it has run.  Every panel of that step which has the tile
    // as a row updates its diagonal block from its diagonal workgroup; the one that arrives last (a counter per tile, or
    // trivially the only one) factors the tile on the spot, so that the step that uses it as a panel starts from the factor.
    std::vector<int> last_updater_step(N, -1), n_updaters_in_last(N, 0);
    for (int k = 0; k < T; ++k) for (int t : rows[k]) {
      if (step_of[k] > last_updater_step[t]) { last_updater_step[t] = step_of[k]; n_updaters_in_last[t] = 1; }
      else if (step_of[k] == last_updater_step[t]) n_updaters_in_last[t]++;
    }
    std::vector<uint8_t> arrival_ok(N, 1);   // every updater of the tile's last step can name it in its 31-bit mask
    for (int k = 0; k < T; ++k)
      for (size_t q = 0; q < rows[k].size(); ++q) if (step_of[k] == last_updater_step[rows[k][q]] && q >= 31) arrival_ok[rows[k][q]] = 0;
    std::vector<uint8_t> factored_by_lookahead(T, 0);
    tile_sync.assign(2 * (size_t)N, 0);
    for (int t = 0; t < T; ++t) if (last_updater_step[t] >= 0 && arrival_ok[t]) { factored_by_lookahead[t] = 1; tile_sync[t] = n_updaters_in_last[t]; }
    panels.clear(); rows_flat.clear(); step_off.assign(1, 0); step_maxrows.clear();
    for (size_t s = 0; s < steps.size(); ++s) {
      int mr = 0;
      for (int k : steps[s]) {
        PanelDesc d;
        d.k = k; d.row_off = (int)rows_flat.size(); d.n_rows = (int)rows[k].size();
        d.final_mask = 0;
        d.shared_mask = 0;
        for (size_t q = 0; q < rows[k].size(); ++q) {   // bits 0..30 exact, bit 31 = any later row (conservative)
          const int rt = rows[k][q];
          if ((rt < T && step_rows[s][rt] > 1) || (rt == T && steps[s].size() > 1 && allow_shared)) d.shared_mask |= (int)(1u << (q < 31 ? q : 31));
          if (rt < T && q < 31 && factored_by_lookahead[rt] && last_updater_step[rt] == (int)s) d.final_mask |= (int)(1u << q);
        }
        rows_flat.insert(rows_flat.end(), rows[k].begin(), rows[k].end());
        mr = std::max(mr, d.n_rows);
        panels.push_back(d);
      }
      step_off.push_back((int)panels.size());
      step_maxrows.push_back(mr);
    }
    // standalone potrf: every tile not factored by a look-ahead, right before the step of its panel
    potrf_before_step_off.assign(steps.size() + 1, 0);
    potrf_tiles.clear();
    for (size_t s = 0; s < steps.size(); ++s) {
      potrf_before_step_off[s] = (int)potrf_tiles.size();
      for (int k : steps[s]) if (!factored_by_lookahead[k]) potrf_tiles.push_back(k);
    }
    potrf_before_step_off[steps.size()] = (int)potrf_tiles.size();
    for (PanelDesc& d : panels) d.self_potrf = factored_by_lookahead[d.k] ? 0 : 1;
    // ---- task list of the fused factorisation (FusedTask): chains, then the update tasks of their panels, depth by depth
    {
      ftasks.clear(); fchain_begin.clear(); fchain_len.clear();
      fchain_of_tile.assign(N, -1);
      auto add_range = [&](int b0, int e0) {
        const int len = e0 - b0;
        if (len <= 0) return;
        const int nseg = (len + kChainMaxTiles - 1) / kChainMaxTiles;
        int at = b0;
        for (int sg = 0; sg < nseg; ++sg) {
          const int l = len / nseg + (sg < len % nseg ? 1 : 0);
          for (int t = at; t < at + l; ++t) fchain_of_tile[t] = (int)fchain_begin.size();
          fchain_begin.push_back(at); fchain_len.push_back(l);
          at += l;
        }
      };
      for (const auto& r : leaf_ranges) add_range(r.first, r.second);
      for (const auto& r : piece_ranges) add_range(r.first, r.second);
      for (const auto& lv : sep_ranges_by_level) for (const auto& r : lv) add_range(r.first, r.second);
      const int nch = (int)fchain_begin.size();
      // appendix tiles: a separator of 78 dimensions is a tile of 64 and one of 14 — and a panel of its own for the 14 costs the critical
      // path a whole hand-over (flag, loads, solve, product, turn, publication: ~6.5 us, BSGPU_CHOL_PROBE) behind the one of the 64.  The
      // tasks of the panel before it carry those columns instead (kFusedExt) when the two panels have the same row tiles outside the chain.
      fext_of.assign(N, -1);
      std::vector<uint8_t> is_app(N, 0);
      if (allow_ext)
        for (int ch = 0; ch < nch; ++ch) {
          const int b0 = fchain_begin[ch], l = fchain_len[ch];
          if (l < 2) continue;
          const int te = b0 + l - 1, kb = te - 1;
          if (nreal[te] < 1 || nreal[te] > kFusedExtCols || !B[(size_t)te * N + kb]) continue;
          std::vector<int> ra, rb;
          for (int a : rows[kb]) if (a > te) ra.push_back(a);
          for (int a : rows[te]) if (a > te) rb.push_back(a);
          if (ra != rb) continue;
          fext_of[kb] = te; is_app[te] = 1;
        }
      // depth of a chain: one more than the deepest chain that updates one of its tiles (S order is topological: rows(k) > k)
      std::vector<int> order_ch(nch), depth(nch, 0);
      for (int i = 0; i < nch; ++i) order_ch[i] = i;
      std::sort(order_ch.begin(), order_ch.end(), [&](int x, int y) { return fchain_begin[x] < fchain_begin[y]; });
      int max_depth = 0;
      for (int ch : order_ch)
        for (int k = fchain_begin[ch]; k < fchain_begin[ch] + fchain_len[ch]; ++k)
          for (int t : rows[k]) if (t < T && fchain_of_tile[t] != ch) { depth[fchain_of_tile[t]] = std::max(depth[fchain_of_tile[t]], depth[ch] + 1); max_depth = std::max(max_depth, depth[fchain_of_tile[t]]); }
      tile_tot.assign((size_t)N * N, 0);
      std::vector<int> seen((size_t)N * N, 0);
      if (diag_tasks)
        for (int t = 0; t < T; ++t) {
          FusedTask f{t, t, t, kFusedDiagAdd, 0, 0, 0, 0};
          seen[(size_t)t * N + t] = 1; tile_tot[(size_t)t * N + t] = 1;
          ftasks.push_back(f);
        }
      for (int u = 0; u < rider_tasks; ++u) ftasks.push_back(FusedTask{u, 0, 0, kFusedRider, 0, 0, -1, 0});
      fused_flops = 0.0;
      const double tile3 = 64.0 * 64.0 * 64.0;
      for (int d = 0; d <= max_depth; ++d) {
        for (int ch : order_ch) {
          if (depth[ch] != d) continue;
          const int b0 = fchain_begin[ch], l = fchain_len[ch];
          unsigned present = 0;
          for (int i = 0; i < l; ++i) for (int j = 0; j <= i; ++j) if (i == j || B[(size_t)(b0 + i) * N + b0 + j]) present |= 1u << (i * (i + 1) / 2 + j);
          FusedTask f{b0, l, (int)present, kFusedChain, 0, 0, -1, 0};
          ftasks.push_back(f);
          const double nn = 64.0 * l;
          fused_flops += nn * nn * nn / 3.0;
        }
        for (int qq = 0; qq < kChainMaxTiles; ++qq)
          for (int ch : order_ch) {
            if (depth[ch] != d || fchain_len[ch] <= qq) continue;
            const int k = fchain_begin[ch] + qq, cend = fchain_begin[ch] + fchain_len[ch];
            if (is_app[k]) continue;   // (carried by the tasks of panel k - 1)
            const int ext = fext_of[k];
            std::vector<FusedTask> st;
            for (int a : rows[k]) {
              if (a < cend) continue;                     // (both tiles inside the chain: the chain's own work)
              fused_flops += tile3;                        // one triangular solve per outside row tile
              for (int b2 : rows[k]) {
                if (b2 > a) continue;
                if (b2 == ext) continue;   // (the appendix's columns of row tile a are formed inside the tasks (k; a, .), not by an update of tile (a, k + 1))
                FusedTask f{k, a, b2, ext >= 0 ? kFusedExt : 0, 0, 0, -1, 0};
                if (b2 < cend) f.flags |= kFusedXjChain;
                if (a == b2) f.flags |= kFusedPublishX;
                if (!(a == T && b2 == T)) fused_flops += (a == b2 ? 1.0 : 2.0) * tile3;
                st.push_back(f);
              }
            }
            // the diagonal tasks first (the others may read what they publish), then the tiles the next panels need first
            std::stable_sort(st.begin(), st.end(), [](const FusedTask& x, const FusedTask& y) {
              const bool dx = x.ti == x.tj, dy = y.ti == y.tj;
              if (dx != dy) return dx;
              return x.tj != y.tj ? x.tj < y.tj : x.ti < y.ti;
            });
            for (FusedTask& f : st) {
              if (!(f.ti == T && f.tj == T)) { f.need_c = seen[(size_t)f.ti * N + f.tj]++; tile_tot[(size_t)f.ti * N + f.tj]++; }
              ftasks.push_back(f);
            }
          }
      }
      // ---- ticket order: the list above is topological, but a chain sits behind EVERY update task of the depths below it, and a
      // workgroup holds its CU from the moment it takes its ticket: hundreds of early tickets that can only wait would keep the next
      // chain from even starting.  Re-order by the time a task can START — its dependencies' estimated finish times (list
      // scheduling with nominal durations, unlimited workgroups) — which is again a topological order (a task starts after everything
      // it waits for has finished), now one in which tickets are taken roughly when they can run.
      {
        const int nt = (int)ftasks.size();
        // nominal durations (microseconds), from the task stamps of C2 (BSGPU_CHOL_PROBE): a chain of m tiles 5 (its tiles' loads) + its real
        // 16-pivot steps x (2.0 | 2.25 | 3.0); an update task 6.5 from its inputs to its publication; two updates of ONE tile publish at
        // least 1.5 apart (the read-modify-write turn — the solves and products of the tile's updaters overlap)
        const double dur_update = 6.5, t_turn = 1.5, chain0 = 5.0;
        auto chain_steps = [&](int k0, int ntile) { int st = 0; for (int i = 0; i < ntile; ++i) st += (std::min(64, std::max(1, nreal[k0 + i])) + 15) / 16; return st; };
        auto chain_step_us = [](int m) { return m <= 1 ? 2.0 : m == 2 ? 2.25 : 3.0; };
        std::vector<double> fin_tile((size_t)N * N, 0.0), st_diag((size_t)N * N, 0.0), fin_potrf(N, 0.0), start(nt, 0.0), dur(nt, dur_update);
        // explicit predecessors for the backward pass: `preds` must have FINISHED before the task starts (the chain of its panel, the last
        // writers of its input tiles, the diagonal tasks whose X it may read), `turn_pred` must have published before it publishes
        std::vector<int> last_writer((size_t)N * N, -1), diag_task((size_t)N * N, -1), chain_task(N, -1), turn_pred(nt, -1);
        std::vector<std::vector<int>> preds(nt);
        for (int t = 0; t < nt; ++t) {   // (list order: every dependency of task t has been seen)
          const FusedTask& f = ftasks[t];
          if (f.flags & (kFusedDiagAdd | kFusedRider)) {   // (start at once, a few microseconds: a diagonal task is the first writer of its tile)
            start[t] = 0.0; dur[t] = 3.0;
            if (f.flags & kFusedDiagAdd) { fin_tile[(size_t)f.k * N + f.k] = 3.0; last_writer[(size_t)f.k * N + f.k] = t; }
            continue;
          }
          if (f.flags & kFusedChain) {
            double st = 0.0;
            for (int i = 0; i < f.ti; ++i) for (int j = 0; j <= i; ++j) {
              st = std::max(st, fin_tile[(size_t)(f.k + i) * N + f.k + j]);
              if (last_writer[(size_t)(f.k + i) * N + f.k + j] >= 0) preds[t].push_back(last_writer[(size_t)(f.k + i) * N + f.k + j]);
            }
            start[t] = st;
            const double us = chain_step_us(f.ti);
            for (int i = 0; i < f.ti; ++i) { fin_potrf[f.k + i] = st + chain0 + us * chain_steps(f.k, i + 1); chain_task[f.k + i] = t; }
            dur[t] = chain0 + us * chain_steps(f.k, f.ti);
          } else {
            const bool diag = f.ti == f.tj, xj_chain = (f.flags & kFusedXjChain) != 0, ext = (f.flags & kFusedExt) != 0;
            double st = std::max(fin_potrf[f.k], fin_tile[(size_t)f.ti * N + f.k]);
            if (chain_task[f.k] >= 0) preds[t].push_back(chain_task[f.k]);
            if (last_writer[(size_t)f.ti * N + f.k] >= 0) preds[t].push_back(last_writer[(size_t)f.ti * N + f.k]);
            if (ext) {   // (the appendix's factor and the row tiles' appendix columns)
              st = std::max(st, std::max(fin_potrf[f.k + 1], std::max(fin_tile[(size_t)f.ti * N + f.k + 1], fin_tile[(size_t)f.tj * N + f.k + 1])));
              if (last_writer[(size_t)f.ti * N + f.k + 1] >= 0) preds[t].push_back(last_writer[(size_t)f.ti * N + f.k + 1]);
              if (last_writer[(size_t)f.tj * N + f.k + 1] >= 0) preds[t].push_back(last_writer[(size_t)f.tj * N + f.k + 1]);
            }
            if (!diag && !xj_chain) {
              st = std::max(st, fin_tile[(size_t)f.tj * N + f.k]);
              if (last_writer[(size_t)f.tj * N + f.k] >= 0) preds[t].push_back(last_writer[(size_t)f.tj * N + f.k]);
              // (it may read what the diagonal tasks of its two tiles publish: it stays behind them.  A task whose tj is in the chain of k
              // never does — it solves its own strip of X_ti and takes X_tj from the chain's factor)
              st = std::max(st, std::max(st_diag[(size_t)f.ti * N + f.k], st_diag[(size_t)f.tj * N + f.k]));
              if (diag_task[(size_t)f.ti * N + f.k] >= 0) preds[t].push_back(diag_task[(size_t)f.ti * N + f.k]);
              if (diag_task[(size_t)f.tj * N + f.k] >= 0) preds[t].push_back(diag_task[(size_t)f.tj * N + f.k]);
            }
            start[t] = st;
            double fin = st + dur_update + (ext ? 1.0 : 0.0);
            if (f.need_c >= 0) {
              if (last_writer[(size_t)f.ti * N + f.tj] >= 0) { fin = std::max(fin, fin_tile[(size_t)f.ti * N + f.tj] + t_turn); turn_pred[t] = last_writer[(size_t)f.ti * N + f.tj]; }
              fin_tile[(size_t)f.ti * N + f.tj] = fin; last_writer[(size_t)f.ti * N + f.tj] = t;
            }
            dur[t] = fin - st;
            if (diag) { st_diag[(size_t)f.ti * N + f.k] = st; diag_task[(size_t)f.ti * N + f.k] = t; }
          }
        }
        // latest start times (ALAP): a task whose result is only needed at the root can take its ticket late; one that feeds the next
        // chain on the path must not queue behind it.  The ticket key blends the earliest and the latest start: for a dependency d of t
        // (chain, input tile, diagonal task), start[d] <= start[t] and lst[d] <= lst[t], so any blend keeps every dependency in front
        // (equal keys: list order).  The turns on a tile are whatever the final order says (need_c, below).
        double makespan = 0.0;
        for (int t = 0; t < nt; ++t) makespan = std::max(makespan, start[t] + dur[t]);
        est_makespan_us = makespan;
        std::vector<double> lfin(nt, 1e300);
        for (int t = nt - 1; t >= 0; --t) {
          if (lfin[t] > 1e299) lfin[t] = makespan;
          const double ls = lfin[t] - std::min(dur[t], (ftasks[t].flags & kFusedChain) ? dur[t] : dur_update);
          for (int p : preds[t]) lfin[p] = std::min(lfin[p], ls);
          if (turn_pred[t] >= 0) lfin[turn_pred[t]] = std::min(lfin[turn_pred[t]], lfin[t] - t_turn);
        }
        const double beta = 0.5;   // (0 .. 1 measured on C2: 168.9 .. 166.5 us per factorisation — the order matters little once the model's durations are right)
        std::vector<int> ord(nt);
        for (int t = 0; t < nt; ++t) ord[t] = t;
        std::vector<double> key(nt);
        for (int t = 0; t < nt; ++t) {
          const double lst = lfin[t] - ((ftasks[t].flags & kFusedChain) ? dur[t] : dur_update);
          key[t] = start[t] + beta * std::max(0.0, lst - start[t]);
        }
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return key[x] < key[y]; });
        // a stable sort by start time keeps every dependency in front: dep.finish <= start, dep.start < dep.finish; equal start times
        // keep the list order.  The turns on a tile (need_c) follow the NEW order.
        std::vector<FusedTask> nl(nt);
        for (int t = 0; t < nt; ++t) nl[t] = ftasks[ord[t]];
        std::fill(seen.begin(), seen.end(), 0);
        for (FusedTask& f : nl) if (!(f.flags & kFusedChain) && f.need_c >= 0) f.need_c = seen[(size_t)f.ti * N + f.tj]++;
        ftasks.swap(nl);
      }
      const bool fetch_x = true;   // (off-diagonal tasks read the X their diagonal tasks publish: one solve per (panel, row tile) instead of one per task)
      for (FusedTask& f : ftasks) {
        if (f.flags & (kFusedChain | kFusedRider)) continue;
        if (f.flags & kFusedDiagAdd) { f.tot_c = tile_tot[(size_t)f.k * N + f.k]; continue; }
        f.tot_i = tile_tot[(size_t)f.ti * N + f.k];
        f.tot_j = tile_tot[(size_t)f.tj * N + f.k];
        f.tot_c = (f.need_c >= 0) ? tile_tot[(size_t)f.ti * N + f.tj] : 0;
        if (f.ti == f.tj || !fetch_x) continue;
        // An off-diagonal task reads the X its diagonal tasks publish unless it is the one a chain is waiting for last: the final
        // update of a tile INSIDE a chain (then it solves its own strips and starts as soon as L_kk is out), or a task whose tj is
        // in the chain of k (its target is the panel tile of the chain's NEXT panels)
        const int chk = fchain_of_tile[f.k];
        const bool last_panel = f.k + 1 + ((f.flags & kFusedExt) ? 1 : 0) == fchain_begin[chk] + fchain_len[chk];   // (the last panels of BOTH sides of a separator finish together)
        const bool last_of_chain_tile = f.ti < T && fchain_of_tile[f.ti] == fchain_of_tile[f.tj] && (f.need_c + 1 == f.tot_c || last_panel);
        if (last_of_chain_tile || (f.flags & kFusedXjChain)) continue;
        f.flags |= kFusedXiLp | kFusedXjLp;
      }
      // ---- kFusedSplit: the last `split_depth` updates (in list order) of every tile inside a chain become K-chunks.  Two by default: a
      // separator's tiles take their last updates from the last panels of BOTH its children, which finish at about the same time — with only
      // the very last one dealt out, its chunks waited 4-6 us for the other child's update to take its turn (BSGPU_CHOL_PROBE, C2).
      n_split_chunks = 0;
      const std::vector<FusedTask> ftasks_whole = ftasks;   // (before the chunks: ftasks_bulk below)
      const std::vector<int> tile_tot_whole = tile_tot;
      if (split_depth > 0) {
        const int nt = (int)ftasks.size();
        std::vector<std::vector<int>> updaters((size_t)N * N);
        for (int t = 0; t < nt; ++t) {
          const FusedTask& f = ftasks[t];
          if ((f.flags & (kFusedChain | kFusedRider)) || f.need_c < 0) continue;
          updaters[(size_t)f.ti * N + f.tj].push_back(t);
        }
        std::vector<int> n_chunks(nt, 0), turn(nt, 0);
        for (int a = 0; a < T; ++a)
          for (int b = 0; b <= a; ++b) {
            if (fchain_of_tile[a] < 0 || fchain_of_tile[a] != fchain_of_tile[b]) continue;
            const std::vector<int>& u = updaters[(size_t)a * N + b];
            int taken = 0, hi = 0;
            for (int q = (int)u.size() - 1; q >= 0 && taken < split_depth; --q, ++taken) {
              const FusedTask& f = ftasks[u[q]];
              if (f.flags & (kFusedDiagAdd | kFusedXjChain)) break;
              n_chunks[u[q]] = kFusedSplitChunks + ((f.flags & kFusedExt) ? 1 : 0);
              hi += n_chunks[u[q]];
            }
            if (!taken) continue;
            const int lo = (int)u.size() - taken;   // the updates that keep a turn of their own: need_c 0 .. lo - 1
            for (int q = lo; q < (int)u.size(); ++q) turn[u[q]] = lo;
            n_split_chunks += hi;
            tile_tot[(size_t)a * N + b] = lo | (hi << 16);
          }
        if (n_split_chunks > 0) {
          std::vector<FusedTask> nl;
          nl.reserve(ftasks.size() + (size_t)n_split_chunks);
          for (int t = 0; t < nt; ++t) {
            if (n_chunks[t] == 0) { nl.push_back(ftasks[t]); continue; }
            for (int p = 0; p < n_chunks[t]; ++p) {
              FusedTask f = ftasks[t];
              f.flags = (f.flags & (kFusedExt | kFusedPublishX)) | kFusedSplit;   // (a chunk forms its own columns of X: nothing published is read)
              f.need_c = turn[t];   // (the chunks of a tile share ONE turn, after every update that has a turn of its own)
              f.tot_c = p | (n_chunks[t] << 8);
              nl.push_back(f);
            }
          }
          ftasks.swap(nl);
          // (tot_i / tot_j of the other tasks name panel tiles (t, k), t outside the chain of k: never a tile inside a chain — except tot_j of
          //  a task whose tj is in the chain of k, which waits for the chain's flag and not for that count)
        }
      }
      fused_sync_words = 16 * (3 + N + N * N);   // (every word a 64-byte line apart: k_chol.hip fused_sync_stride)
      // the same list WITHOUT the diagonal / rider tasks (what a launch that carries neither takes: a step whose LM diagonal is in S already,
      // the batched launches of many windows — every task is a workgroup with ~160 KB of LDS, also one that only counts itself in)
      ftasks_plain.clear(); tile_tot_plain = tile_tot;
      if (diag_tasks || rider_tasks > 0) {
        for (const FusedTask& f0 : ftasks) {
          if (f0.flags & (kFusedDiagAdd | kFusedRider)) continue;
          FusedTask f = f0;
          if (diag_tasks && !(f.flags & kFusedChain) && f.ti == f.tj && f.ti < T && f.need_c >= 0) { f.need_c--; if (!(f.flags & kFusedSplit)) f.tot_c--; }   // (a chunk's tot_c is its number)
          ftasks_plain.push_back(f);
        }
        if (diag_tasks) for (int t = 0; t < T; ++t) tile_tot_plain[(size_t)t * N + t]--;
      }
      ftasks_bulk.clear(); tile_tot_bulk.clear();
      if (n_split_chunks > 0) {
        tile_tot_bulk = tile_tot_whole;
        for (const FusedTask& f0 : ftasks_whole) {
          if (f0.flags & (kFusedDiagAdd | kFusedRider)) continue;
          FusedTask f = f0;
          if (diag_tasks && !(f.flags & kFusedChain) && f.ti == f.tj && f.ti < T && f.need_c >= 0) { f.need_c--; f.tot_c--; }
          ftasks_bulk.push_back(f);
        }
        if (diag_tasks) for (int t = 0; t < T; ++t) tile_tot_bulk[(size_t)t * N + t]--;
      }
      build_row_segments();
    }
    // ---- back-substitution plan
    panel_of_tile.assign(T, 0);
    for (size_t i = 0; i < panels.size(); ++i) panel_of_tile[panels[i].k] = (int)i;
    n_pieces = (int)piece_ranges.size();
    bs_desc.assign((size_t)std::max(1, T) * kBsDescInts, 0);
    for (int k = 0; k < T; ++k) {
      const PanelDesc& d = panels[panel_of_tile[k]];
      int* r = &bs_desc[(size_t)k * kBsDescInts];
      r[0] = d.n_rows; r[1] = d.row_off; r[2] = nreal[k];
      for (int q = 0; q < d.n_rows && q < kBsDescRows; ++q) r[3 + q] = rows_flat[d.row_off + q];
    }
    auto build_groups = [&](bool by_level) {
      chain_begin.clear(); chain_end.clear(); bs_group_off.assign(1, 0);
      if (by_level) {
        for (int lvl = (int)sep_ranges_by_level.size() - 1; lvl >= 0; --lvl) {
          for (const auto& r : sep_ranges_by_level[lvl]) { chain_begin.push_back(r.first); chain_end.push_back(r.second); }
          if (!sep_ranges_by_level[lvl].empty()) bs_group_off.push_back((int)chain_begin.size());
        }
        for (const auto& r : piece_ranges) { chain_begin.push_back(r.first); chain_end.push_back(r.second); }
        bs_group_off.push_back((int)chain_begin.size());
        if (!leaf_ranges.empty()) {   // the leaf tiles last: every row tile of theirs is a core tile, solved by then
          for (const auto& r : leaf_ranges) { chain_begin.push_back(r.first); chain_end.push_back(r.second); }
          bs_group_off.push_back((int)chain_begin.size());
        }
      } else {   // reverse step schedule, one single-tile chain per panel (always valid)
        for (int st = (int)steps.size() - 1; st >= 0; --st) {
          for (int k : steps[st]) { chain_begin.push_back(k); chain_end.push_back(k + 1); }
          bs_group_off.push_back((int)chain_begin.size());
        }
      }
    };
    build_groups(true);
    {
      std::vector<int> group_of(T, -1), chain_of(T, -1);
      for (size_t g = 0; g + 1 < bs_group_off.size(); ++g)
        for (int c = bs_group_off[g]; c < bs_group_off[g + 1]; ++c)
          for (int t = chain_begin[c]; t < chain_end[c]; ++t) { group_of[t] = (int)g; chain_of[t] = c; }
      bool ok = true;
      for (int k = 0; k < T && ok; ++k) {
        if (group_of[k] < 0) ok = false;
        for (int t : rows[k]) if (t < T && !((chain_of[t] == chain_of[k] && t > k) || group_of[t] < group_of[k])) ok = false;
      }
      if (!ok) build_groups(false);
      bs_level_sync = false;
      bs_desc_chain.clear(); rows_flat_chain.clear(); bs_upd.clear(); bs_upd_rows.clear(); bs_upd_off.assign(1, 0);
      if (ok && bs_group_off.size() > 2) {
        bs_desc_chain.assign((size_t)std::max(1, T) * kBsDescInts, 0);
        for (int k = 0; k < T; ++k) {
          int* r = &bs_desc_chain[(size_t)k * kBsDescInts];
          const int off = (int)rows_flat_chain.size();
          // (its own chain's row tiles AND those of the group just before its own — its parent separator: the chain multiplies them in
          // itself when the parent's y is out, instead of waiting for a round of update items in between; round 4)
          for (int t : rows[k]) if (t < T && (chain_of[t] == chain_of[k] || group_of[t] == group_of[k] - 1)) rows_flat_chain.push_back(t);
          const int n = (int)rows_flat_chain.size() - off;
          r[0] = n; r[1] = off; r[2] = nreal[k];
          for (int q = 0; q < n && q < kBsDescRows; ++q) r[3 + q] = rows_flat_chain[off + q];
        }
        const int G = (int)bs_group_off.size() - 1;
        for (int g = 0; g + 1 < G; ++g) {
          for (int k = 0; k < T; ++k) {
            if (group_of[k] <= g + 1) continue;   // (the next group's chains take this group's rows themselves)
            const int off = (int)bs_upd_rows.size();
            for (int t : rows[k]) if (t < T && group_of[t] == g) bs_upd_rows.push_back(t);
            const int n = (int)bs_upd_rows.size() - off;
            if (n > 0) { bs_upd.push_back(k); bs_upd.push_back(off); bs_upd.push_back(n); }
          }
          bs_upd_off.push_back((int)bs_upd.size() / 3);
        }
        bs_chain_group.assign(chain_begin.size(), 0); bs_grp_nchains.assign(G, 0); bs_grp_nitems.assign(G, 0);
        for (int g = 0; g < G; ++g) {
          bs_grp_nchains[g] = bs_group_off[g + 1] - bs_group_off[g];
          for (int c = bs_group_off[g]; c < bs_group_off[g + 1]; ++c) bs_chain_group[c] = g;
        }
        bs_tile_updated.assign(std::max(1, T), 0);
        bs_items4.clear();
        for (int g = 0; g + 1 < G; ++g) {
          bs_grp_nitems[g] = bs_upd_off[g + 1] - bs_upd_off[g];
          for (int i = bs_upd_off[g]; i < bs_upd_off[g + 1]; ++i) {
            const int k = bs_upd[3 * i];
            bs_items4.push_back(k); bs_items4.push_back(bs_upd[3 * i + 1]); bs_items4.push_back(bs_upd[3 * i + 2]);
            bs_items4.push_back(g | ((bs_tile_updated[k] ? 0 : 1) << 16));
            bs_tile_updated[k] = 1;
          }
        }
        bs_order.clear();
        for (int g = 0; g < G; ++g) {
          for (int c = bs_group_off[g]; c < bs_group_off[g + 1]; ++c) bs_order.push_back(c);
          if (g + 1 < G) for (int i = bs_upd_off[g]; i < bs_upd_off[g + 1]; ++i) bs_order.push_back((int)chain_begin.size() + i);
        }
        if (bs_items4.empty()) bs_items4.assign(4, 0);
        bs_group_maxrows.assign(G, 0);
        for (int k = 0; k < T; ++k) bs_group_maxrows[group_of[k]] = std::max(bs_group_maxrows[group_of[k]], bs_desc_chain[(size_t)k * kBsDescInts]);
        if (rows_flat_chain.empty()) rows_flat_chain.push_back(0);
        if (bs_upd_rows.empty()) bs_upd_rows.push_back(0);
        if (bs_upd.empty()) { bs_upd.assign(3, 0); }
        bs_level_sync = true;
      }
    }
  }
 public:
  // Level-synchronous back-substitution (by-level groups only): a chain's workgroup walks its panels with the row tiles of ITS OWN
  // chain and of the group just before its own (bs_desc_chain / rows_flat_chain, same record format as bs_desc); what the chains of
  // group g contribute to the panels of the groups from g + 2 on — y_k -= sum_{t in group g} L(t,k)^T y_t — is applied by one workgroup per target panel
  // (bs_upd: {k, offset into bs_upd_rows, count} per item, items of group g in [bs_upd_off[g], bs_upd_off[g+1])).  Those products
  // are two thirds of the factor's entries and depend on nothing inside the later chains: as a wide launch they stream at the
  // chip's bandwidth instead of at one workgroup's latency.
  bool bs_level_sync = false;
  std::vector<int> bs_desc_chain, rows_flat_chain, bs_upd, bs_upd_rows, bs_upd_off;
  std::vector<int> bs_chain_group, bs_grp_nchains, bs_grp_nitems, bs_items4, bs_tile_updated;
  std::vector<int> bs_order;   // ticket -> workgroup role (chain index, or n_chains + item index) in dependency order: chains of group 0, items of phase 0, chains of group 1, ...   // single-launch form (k_chol.hip chol_backsolve_fused_kernel)
  std::vector<int> bs_group_maxrows;   // per group: the most own-chain row tiles of any of its panels (level-synchronous form)
  int n_steps() const { return (int)step_off.size() - 1; }
};

}  // namespace bsg
