// LmPairsPlan (beam_slam_amd/csrc/lm_pairs_plan.h) on random landmark / camera-pose structures: every (factor a, factor b) pair of a
// landmark with cam(a) <= cam(b) appears exactly once, inside one batch that stages both factors; a group's slots hold every camera
// pose its factors see, ascending; no batch stages more than kLmcBatch factors; factors of constant landmarks give their diagonal
// entry alone; a landmark seen from more than kLmcSlots camera poses makes build() decline.
#include <cstdio>
#include <map>
#include <random>
#include <set>
#include "lm_pairs_plan.h"

using namespace bsg;

static bool check(int n_lm, int n_fac, int n_elim, const std::vector<int>& lm_start, const std::vector<int>& cam, int ncp, int group_max, bool expect_ok) {
  LmPairsPlan P;
  const bool ok = P.build(n_lm, n_fac, n_elim, lm_start.data(), cam.data(), ncp, group_max);
  if (ok != expect_ok) { std::printf("build() returned %d, expected %d\n", (int)ok, (int)expect_ok); return false; }
  if (!ok) return true;
  std::vector<int> lm_of(n_fac, -1);
  for (int l = 0; l < n_lm; ++l) for (int f = lm_start[l]; f < lm_start[l + 1]; ++f) lm_of[f] = l;
  std::map<std::pair<int, int>, int> seen;
  std::vector<int> staged(n_fac, 0);
  if ((int)P.group_batch.size() != P.n_group + 1 || (int)P.batch_fac.size() != P.n_batch + 1 || (int)P.batch_ent.size() != P.n_batch + 1) return false;
  if (P.pair_start.size() != (size_t)P.n_batch * kLmcPairStride) return false;
  if (P.group_batch.back() != P.n_batch || P.batch_fac.back() != (int)P.fac.size() || P.batch_ent.back() != (int)P.ent.size()) return false;
  for (int g = 0; g < P.n_group; ++g) {
    const int* sl = &P.slots[(size_t)g * kLmcSlots];
    for (int s = 1; s < kLmcSlots; ++s) if (sl[s] >= 0 && !(sl[s - 1] >= 0 && sl[s - 1] < sl[s])) { std::printf("slots not ascending\n"); return false; }
    int gf = 0;
    for (int b = P.group_batch[g]; b < P.group_batch[g + 1]; ++b) {
      const int f0 = P.batch_fac[b], nf = P.batch_fac[b + 1] - f0;
      if (nf <= 0 || nf > kLmcBatch) { std::printf("batch of %d factors\n", nf); return false; }
      gf += nf;
      for (int i = 0; i < nf; ++i) {
        const int f = (int)(P.fac[f0 + i] >> 4), s = (int)(P.fac[f0 + i] & 15u);
        if (f < 0 || f >= n_fac || sl[s] != cam[f]) { std::printf("slot of factor %d\n", f); return false; }
        staged[f]++;
      }
      const int e0 = P.batch_ent[b], ne_pad = P.batch_ent[b + 1] - e0;
      const uint16_t* ps = &P.pair_start[(size_t)b * kLmcPairStride];
      const int ne = ps[kLmcPairs];
      if (ne_pad % 8 != 0 || ne > ne_pad || ne_pad - ne >= 8 || ne > kLmcBatchEnt || ps[0] != 0) { std::printf("entry padding\n"); return false; }
      int ns = 0;
      while (ns < kLmcSlots && sl[ns] >= 0) ++ns;
      for (int p = 0; p < kLmcPairs; ++p) if (ps[p] > ps[p + 1]) { std::printf("pair offsets\n"); return false; }
      for (int p = 0; p < kLmcPairs; ++p)
        for (int e = e0 + ps[p]; e < e0 + ps[p + 1]; ++e) {
          const int la = P.ent[e] & 255, lb = P.ent[e] >> 8;
          if (la >= nf || lb >= nf) { std::printf("entry outside its batch\n"); return false; }
          const int sa = (int)(P.fac[f0 + la] & 15u), sb = (int)(P.fac[f0 + lb] & 15u);
          if (sa > sb || lmc_pair_index(ns, sa, sb) != p) { std::printf("entry in the wrong pair\n"); return false; }
        }
      for (int e = e0; e < e0 + ne; ++e) {
        const int la = P.ent[e] & 255, lb = P.ent[e] >> 8;
        if (la >= nf || lb >= nf) { std::printf("entry outside its batch\n"); return false; }
        const int fa = (int)(P.fac[f0 + la] >> 4), fb = (int)(P.fac[f0 + lb] >> 4);
        if (cam[fa] > cam[fb]) { std::printf("entry order\n"); return false; }
        if (fa != fb && (lm_of[fa] < 0 || lm_of[fa] != lm_of[fb])) { std::printf("entry across landmarks\n"); return false; }
        seen[{fa, fb}]++;
      }
    }
    if (gf > group_max && P.group_batch[g + 1] - P.group_batch[g] > 1 && gf > group_max + kLmcBatch) { std::printf("group of %d factors\n", gf); return false; }
  }
  for (int f = 0; f < n_fac; ++f) {
    const bool in_unit = f >= n_elim || lm_of[f] >= 0;
    if (staged[f] != (in_unit ? 1 : 0)) { std::printf("factor %d staged %d times\n", f, staged[f]); return false; }
  }
  size_t want = 0;
  for (int l = 0; l < n_lm; ++l)
    for (int a = lm_start[l]; a < lm_start[l + 1]; ++a)
      for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
        if (cam[a] <= cam[b]) { ++want; if (seen[{a, b}] != 1) { std::printf("pair (%d, %d) seen %d times\n", a, b, seen[{a, b}]); return false; } }
  for (int f = n_elim; f < n_fac; ++f) { ++want; if (seen[{f, f}] != 1) { std::printf("constant-landmark factor %d\n", f); return false; } }
  size_t have = 0;
  for (int b = 0; b < P.n_batch; ++b) have += P.pair_start[(size_t)b * kLmcPairStride + kLmcPairs];
  if (want != have) { std::printf("%zu entries, expected %zu\n", have, want); return false; }
  return true;
}

int main() {
  std::mt19937 rng(20250929);
  int fails = 0, cases = 0;
  for (int it = 0; it < 300; ++it) {
    const int ncp = 1 + (int)(rng() % 60), n_lm = (int)(rng() % 400), n_const = (int)(rng() % 20);
    const int max_track = 1 + (int)(rng() % 16);
    const bool stereo = rng() % 4 == 0, scattered = rng() % 5 == 0;
    std::vector<int> lm_start(1, 0), cam;
    for (int l = 0; l < n_lm; ++l) {
      const int m = (rng() % 10 == 0) ? 0 : 1 + (int)(rng() % std::min(max_track, ncp));
      std::set<int> cs;
      if (scattered) { while ((int)cs.size() < m) cs.insert((int)(rng() % ncp)); }
      else { const int k0 = (int)(rng() % (ncp - m + 1)); for (int j = 0; j < m; ++j) cs.insert(k0 + j); }
      std::vector<int> order(cs.begin(), cs.end());
      std::shuffle(order.begin(), order.end(), rng);
      for (int c : order) { cam.push_back(c); if (stereo && rng() % 2) cam.push_back(c); }
      lm_start.push_back((int)cam.size());
    }
    const int n_elim = (int)cam.size();
    for (int i = 0; i < n_const; ++i) cam.push_back((int)(rng() % ncp));
    const int group_max = kLmcBatch * (1 + (int)(rng() % 4));
    // scattered camera sets can make a GROUP overflow, never a single landmark (<= 16 distinct poses): build() must still succeed
    ++cases;
    if (!check(n_lm, (int)cam.size(), n_elim, lm_start, cam, ncp, group_max, true)) { ++fails; std::printf("case %d failed\n", it); }
  }
  {   // a landmark seen from 17 camera poses: declined
    std::vector<int> lm_start = {0, 17}, cam;
    for (int i = 0; i < 17; ++i) cam.push_back(i);
    ++cases;
    if (!check(1, 17, 17, lm_start, cam, 20, 768, false)) ++fails;
  }
  std::printf("%d cases, %d failures\n", cases, fails);
  return fails ? 1 : 0;
}
