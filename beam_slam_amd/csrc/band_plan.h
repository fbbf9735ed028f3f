// Band landmarks of the visual Schur complement (k_band.hip: pairs_band_kernel): which landmarks qualify, their records, how they are
// dealt out to workgroups, and which camera-pose pairs they couple.  Host side; shared by the host and the device flattening of finalize()
// (the device classifies with the same rule: k_flatten.hip fl_band_kernel).
//   A landmark qualifies when it is seen at least twice, all its camera poses lie within kBandCams consecutive ids and no camera pose sees
//   it twice (a stereo pair on one body pose keeps the general pair entries of pairs_kernel).
//   Units: landmarks grouped by their first camera pose k0 (the row origin of the 78 x 78 block a workgroup accumulates), inside a group by
//   falling span (a sub-batch of a workgroup multiplies only the tile rows its widest landmark reaches), groups cut into parts of `part`.
#pragma once
#include <cstdint>
#include <vector>

#include "bsgpu_internal.h"

namespace bsg {

// the record of a band landmark whose factor rows are [b, e) with camera-pose ids cam_pose[b..e): false if it does not qualify
__host__ __device__ inline bool band_record(int b, int e, const int* cam_pose, int* cmin, int4* rec) {
  *cmin = -1;
  *rec = make_int4(0, 0, -1, -1);
  if (e - b < 2 || e - b > kBandCams) return false;
  int lo = cam_pose[b], hi = cam_pose[b];
  for (int f = b + 1; f < e; ++f) { lo = cam_pose[f] < lo ? cam_pose[f] : lo; hi = cam_pose[f] > hi ? cam_pose[f] : hi; }
  if (hi - lo >= kBandCams) return false;
  unsigned m = 0;
  uint64_t inv = ~(uint64_t)0;   // nibble j: the observation that sees slot j, 15 = none
  for (int f = b; f < e; ++f) {
    const int sl = cam_pose[f] - lo;
    const unsigned bit = 1u << sl;
    if (m & bit) return false;
    m |= bit;
    inv = (inv & ~((uint64_t)15 << (4 * sl))) | ((uint64_t)(f - b) << (4 * sl));
  }
  *cmin = lo;
  *rec = make_int4(b, (int)(m | ((unsigned)(hi - lo + 1) << 16) | ((unsigned)(e - b) << 24)), (int)(uint32_t)inv, (int)(uint32_t)(inv >> 32));
  return true;
}

inline void band_classify_host(int nl, const int* lm_start, const int* cam_pose, std::vector<int>& cmin, std::vector<int>& mask, std::vector<int4>& rec) {
  cmin.assign(nl, -1); mask.assign(nl, 0);
  rec.assign(nl, make_int4(0, 0, -1, -1));
  for (int l = 0; l < nl; ++l)
    if (band_record(lm_start[l], lm_start[l + 1], cam_pose, &cmin[l], &rec[l])) mask[l] = rec[l].y & 0xffff;
}

struct BandUnits {
  std::vector<int> lm;                // band landmarks in unit order
  std::vector<int> unit_start, unit_cam;
  std::vector<uint32_t> adj;          // per camera pose i: bit d set = some band landmark is seen from poses i and i + d (d < kBandCams)
};

// landmarks per unit, at most (BSGPU_BAND_PART forces a value)
int band_part_forced();

// cmin / mask per landmark (cmin < 0: not a band landmark).  One unit per first camera pose unless that leaves most of the device idle: a
// unit ends with up to 6 400 atomic adds into S, whatever its size (two units per first camera pose on C2: 56 us against 50).
inline void band_units(int nl, const int* cmin, const int* mask, int ncp, BandUnits& out) {
  out.lm.clear(); out.unit_start.clear(); out.unit_cam.clear();
  const size_t nc = (size_t)(ncp > 0 ? ncp : 1);
  out.adj.assign(nc, 0);
  // counting sort by (first camera pose, kBandCams - span)
  const size_t nkey = nc * 16;
  std::vector<int> start(nkey + 1, 0);
  auto span_of = [&](int l) { int top = 0; for (unsigned m = (unsigned)mask[l]; m; m >>= 1) ++top; return top; };
  auto key_of = [&](int l) { return (size_t)cmin[l] * 16 + (size_t)(kBandCams - span_of(l)); };
  int n = 0;
  for (int l = 0; l < nl; ++l) if (cmin[l] >= 0) { start[key_of(l) + 1]++; ++n; }
  out.unit_start.push_back(0);
  if (!n) return;
  for (size_t k = 0; k < nkey; ++k) start[k + 1] += start[k];
  out.lm.resize(n);
  for (int l = 0; l < nl; ++l) if (cmin[l] >= 0) out.lm[start[key_of(l)]++] = l;
  out.unit_start.clear();
  int part = band_part_forced();
  if (part <= 0) part = n / 128 > 64 ? n / 128 : 64;
  // the pairs a mask couples, once per distinct (k0, mask): a window has a few masks per first camera pose
  std::vector<std::vector<unsigned>> seen(nc);
  int i = 0;
  while (i < n) {
    const int k0 = cmin[out.lm[i]];
    int j = i;
    while (j < n && cmin[out.lm[j]] == k0) {
      const unsigned m = (unsigned)mask[out.lm[j]];
      bool have = false;
      for (unsigned s : seen[k0]) if (s == m) { have = true; break; }
      if (!have) {
        seen[k0].push_back(m);
        for (int a = 0; a < kBandCams; ++a)
          if ((m >> a) & 1u) out.adj[k0 + a] |= (m >> a);
      }
      ++j;
    }
    // (parts of equal size rather than full parts and a remainder)
    const int parts = (j - i + part - 1) / part;
    for (int p = 0; p < parts; ++p) { out.unit_start.push_back(i + (int)((int64_t)(j - i) * p / parts)); out.unit_cam.push_back(k0); }
    i = j;
  }
  out.unit_start.push_back(n);
}

}  // namespace bsg
