// Host-side tables of the LANDMARK-MAJOR assembly of the camera blocks (pairs_lm_kernel, k_reproj.hip; built by bsgpu_finalize.cpp).
//
// The camera-pair segments of pairs_kernel gather, for every (factor a, factor b) entry of a landmark, the two factors' rows from wherever
// they lie: 304 bytes and five cache lines per entry, 36 entries per landmark of eight observations — 0.55 GB of line traffic per
// assembly of C2 for 64 MB of rows, and the kernel runs at the rate the L2 hands lines out.  Here the rows of a landmark are read ONCE:
// a workgroup stages the rows of a BATCH of landmarks in LDS (their factors are contiguous in the landmark-sorted arrays) and forms
// W_f = A_f^T C_f (6x3) of every staged factor; each of its lanes OWNS one row of one 6x6 block of the reduced system — block = a pair of
// camera poses of the workgroup's GROUP — walks the batch's entries of that pair and keeps the sums
//     block(a, :) += [f_a == f_b] (A_a^T A_b)(a, :) - W_a(a, :) W_b^T
// in registers, batch after batch; after the group's last batch every lane adds its six sums to the reduced system.  No atomics inside
// the workgroup, no reduction across lanes; the six lanes of a block read the same W_b (one LDS access).
//
// A group is a run of landmarks (ordered by their first camera pose) that see at most kLmcSlots camera poses between them; its pairs are
// numbered by distance, (d, s_a) -> s_b = s_a + d over the group's n_s poses, so that the lanes of a wave own pairs with about as many
// entries each.  Windows with a landmark seen from more than kLmcSlots camera poses keep the pair segments (build() returns false).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace bsg {

constexpr int kLmcSlots = 16;          // camera poses of a group
constexpr int kLmcBatch = 192;         // factors staged at a time (local factor ids fit a byte)
constexpr int kLmcBatchEnt = 2048;     // entries of a batch (8 per lane of the staging pass)
constexpr int kLmcPairs = kLmcSlots * (kLmcSlots + 1) / 2;
constexpr int kLmcPairStride = 144;    // uint16 per batch in pair_start: kLmcPairs + 1 offsets, padded to whole 16-byte pieces
constexpr int kLmcThreads = 512;       // lanes of a workgroup of pairs_lm_kernel

// pair (s_a <= s_b) of a group with n_s camera poses -> its index: distance-major
inline int lmc_pair_index(int n_s, int sa, int sb) { const int d = sb - sa; return d * n_s - d * (d - 1) / 2 + sa; }

struct LmPairsPlan {
  int n_group = 0, n_batch = 0;
  std::vector<int> slots;          // n_group x kLmcSlots camera-pose ids, ascending, -1 beyond the group's count
  std::vector<int> group_batch;    // n_group + 1: the group's batches
  std::vector<int> batch_fac;      // n_batch + 1, into fac
  std::vector<int> batch_ent;      // n_batch + 1, into ent (multiples of 8: a batch's entries are padded to whole 16-byte pieces)
  std::vector<uint32_t> fac;       // staged factor: (factor index << 4) | slot of its camera pose
  std::vector<uint16_t> ent;       // entry: local factor a | local factor b << 8, the batch's entries sorted by pair index
  std::vector<uint16_t> pair_start;   // n_batch x kLmcPairStride: entries [pair_start[p], pair_start[p + 1]) of the batch belong to pair p
  std::vector<uint16_t> lane_start;   // n_group x kLmcPairStride: lanes [lane_start[p], lane_start[p + 1]) of the workgroup own block p

  // lm_start[0 .. n_lm]: the factors of eliminated landmark l (factors sorted by landmark); factors n_elim .. n_fac-1 have a constant
  // landmark (a unit of their own each: only the diagonal entry).  group_max: factors per group (the balance of the launch).
  bool build(int n_lm, int n_fac, int n_elim, const int* lm_start, const int* cam_pose, int n_cam_pose, int group_max) {
    *this = LmPairsPlan();
    if ((size_t)n_fac >= ((size_t)1 << 28)) return false;
    const int n_unit = n_lm + (n_fac - n_elim);
    auto unit_range = [&](int u, int& f0, int& f1) {
      if (u < n_lm) { f0 = lm_start[u]; f1 = lm_start[u + 1]; } else { f0 = n_elim + (u - n_lm); f1 = f0 + 1; }
    };
    auto unit_entries = [&](int f0, int f1) {
      int n = 0;
      for (int a = f0; a < f1; ++a) for (int b = f0; b < f1; ++b) if (cam_pose[a] <= cam_pose[b]) ++n;
      return n;
    };
    // units by their first camera pose (counting sort, stable)
    std::vector<int> first_cam(n_unit), count(n_cam_pose + 2, 0), order(n_unit);
    for (int u = 0; u < n_unit; ++u) {
      int f0, f1;
      unit_range(u, f0, f1);
      if (f1 - f0 > kLmcBatch) return false;
      int mn = n_cam_pose;   // (an empty unit sorts last and stages nothing)
      for (int f = f0; f < f1; ++f) mn = std::min(mn, cam_pose[f]);
      first_cam[u] = mn;
      count[mn + 1]++;
    }
    for (int i = 0; i <= n_cam_pose; ++i) count[i + 1] += count[i];
    for (int u = 0; u < n_unit; ++u) order[count[first_cam[u]]++] = u;
    std::vector<int> stamp(std::max(1, n_cam_pose), -1), slot_of(std::max(1, n_cam_pose), 0);
    std::vector<int> cams;            // camera poses of the open group
    std::vector<int> members;         // its units
    int group_fac = 0;
    bool bad = false;
    group_batch.push_back(0); batch_fac.push_back(0); batch_ent.push_back(0);
    struct E { uint16_t code; uint16_t pair; };
    std::vector<E> be;                // entries of the open batch
    std::vector<int> pc(kLmcPairs + 1);
    std::vector<double> weight(kLmcPairs);
    std::vector<int> lanes(kLmcPairs);
    auto close_group = [&]() {
      if (members.empty()) return;
      std::sort(cams.begin(), cams.end());
      const int ns = (int)cams.size();
      for (int s = 0; s < ns; ++s) slot_of[cams[s]] = s;
      for (int s = 0; s < kLmcSlots; ++s) slots.push_back(s < ns ? cams[s] : -1);
      int in_batch = 0;
      std::fill(weight.begin(), weight.end(), 0.0);
      auto close_batch = [&]() {
        if (in_batch == 0) return;
        for (const E& e : be) weight[e.pair] += e.pair < ns ? 2.0 : 1.0;   // (a diagonal pair's entries: the W product, A^T A and the three vectors)
        // the batch's entries by pair (counting sort, stable: landmark-major inside a pair)
        std::fill(pc.begin(), pc.end(), 0);
        for (const E& e : be) pc[e.pair + 1]++;
        for (int p = 0; p < kLmcPairs; ++p) pc[p + 1] += pc[p];
        const size_t e0 = ent.size();
        ent.resize(e0 + ((be.size() + 7) & ~(size_t)7), 0);
        const size_t ps0 = pair_start.size();
        pair_start.resize(ps0 + kLmcPairStride, (uint16_t)be.size());
        for (int p = 0; p <= kLmcPairs; ++p) pair_start[ps0 + p] = (uint16_t)pc[p];
        for (const E& e : be) ent[e0 + pc[e.pair]++] = e.code;
        be.clear();
        batch_fac.push_back((int)fac.size()); batch_ent.push_back((int)ent.size());
        in_batch = 0;
      };
      for (int u : members) {
        int f0, f1;
        unit_range(u, f0, f1);
        const int ne = unit_entries(f0, f1);
        if (ne > kLmcBatchEnt) { bad = true; break; }
        if (in_batch + (f1 - f0) > kLmcBatch || (int)be.size() + ne > kLmcBatchEnt) close_batch();
        const int base = in_batch;
        for (int f = f0; f < f1; ++f) fac.push_back(((uint32_t)f << 4) | (uint32_t)slot_of[cam_pose[f]]);
        for (int a = f0; a < f1; ++a)
          for (int b = f0; b < f1; ++b)
            if (cam_pose[a] <= cam_pose[b])
              be.push_back({(uint16_t)((base + a - f0) | ((base + b - f0) << 8)), (uint16_t)lmc_pair_index(ns, slot_of[cam_pose[a]], slot_of[cam_pose[b]])});
        in_batch += f1 - f0;
      }
      close_batch();
      {   // lanes in proportion to the pairs' work: at least one for a pair with entries, none for an empty one
        double tot = 0.0;
        int busy = 0;
        for (int p = 0; p < kLmcPairs; ++p) { tot += weight[p]; busy += weight[p] > 0.0; }
        int used = 0;
        for (int p = 0; p < kLmcPairs; ++p) {
          lanes[p] = weight[p] > 0.0 ? std::max(1, (int)(weight[p] / tot * (kLmcThreads - busy))) : 0;
          used += lanes[p];
        }
        // (rounding down left a few lanes: to the pairs with the most work per lane)
        while (used < kLmcThreads && busy > 0) {
          int best = -1;
          for (int p = 0; p < kLmcPairs; ++p) if (lanes[p] > 0 && (best < 0 || weight[p] / lanes[p] > weight[best] / lanes[best])) best = p;
          lanes[best]++; ++used;
        }
        const size_t l0 = lane_start.size();
        lane_start.resize(l0 + kLmcPairStride, (uint16_t)used);
        int at = 0;
        for (int p = 0; p <= kLmcPairs; ++p) { lane_start[l0 + p] = (uint16_t)at; if (p < kLmcPairs) at += lanes[p]; }
      }
      group_batch.push_back((int)batch_fac.size() - 1);
      members.clear(); cams.clear(); group_fac = 0;
    };
    int group_id = 0;   // stamp value of the open group
    for (int u : order) {
      int f0, f1;
      unit_range(u, f0, f1);
      if (f1 == f0) continue;
      // camera poses this unit would add
      int added = 0;
      for (int f = f0; f < f1; ++f) if (stamp[cam_pose[f]] != group_id) { stamp[cam_pose[f]] = group_id; cams.push_back(cam_pose[f]); ++added; }
      if ((int)cams.size() > kLmcSlots || (group_fac > 0 && group_fac + (f1 - f0) > group_max)) {
        cams.resize(cams.size() - added);   // (this unit opens the next group)
        close_group();
        ++group_id;
        for (int f = f0; f < f1; ++f) if (stamp[cam_pose[f]] != group_id) { stamp[cam_pose[f]] = group_id; cams.push_back(cam_pose[f]); }
        if ((int)cams.size() > kLmcSlots) return false;   // one landmark seen from more camera poses than a group holds
      }
      members.push_back(u);
      group_fac += f1 - f0;
      if (bad) return false;
    }
    close_group();
    if (bad) return false;
    n_group = (int)group_batch.size() - 1;
    n_batch = (int)batch_fac.size() - 1;
    return n_group > 0;
  }
};

}  // namespace bsg
