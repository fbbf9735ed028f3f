// Band landmarks of the visual Schur complement (k_reproj.hip: pairs_band_kernel): which landmarks qualify and how they are dealt out
// to workgroups.  Host side, shared by the host and the device flattening of finalize().
//   a landmark qualifies when it is seen at least twice, all its camera poses lie within kBandCams consecutive ids and no camera pose
//   sees it twice (a stereo pair on one body pose keeps the general pair entries);
//   units: landmarks grouped by their first camera pose k0 (the row origin of the 78 x 78 block a workgroup accumulates), inside a
//   group by falling span (neighbouring landmarks of a wave issue the same number of products), groups cut into parts of kBandPart.
#pragma once
#include <cstdint>
#include <vector>

#include "bsgpu_internal.h"

namespace bsg {

// per landmark: cmin[l] = first camera pose or -1 (not a band landmark), mask[l] = bit (cam - cmin) per observation, and the record the
// kernel reads: (first factor row, mask | observations << 16, slot of observation o in nibble o of the 64 bits z | w << 32)
inline void band_classify_host(int nl, const int* lm_start, const int* cam_pose, std::vector<int>& cmin, std::vector<int>& mask,
                               std::vector<int4>& rec) {
  cmin.assign(nl, -1); mask.assign(nl, 0);
  rec.assign(nl, make_int4(0, 0, 0, 0));
  for (int l = 0; l < nl; ++l) {
    const int b = lm_start[l], e = lm_start[l + 1];
    if (e - b < 2 || e - b > kBandCams) continue;
    int lo = cam_pose[b], hi = cam_pose[b];
    for (int f = b + 1; f < e; ++f) { lo = cam_pose[f] < lo ? cam_pose[f] : lo; hi = cam_pose[f] > hi ? cam_pose[f] : hi; }
    if (hi - lo >= kBandCams) continue;
    unsigned m = 0;
    uint64_t nib = 0;
    bool dup = false;
    for (int f = b; f < e; ++f) {
      const unsigned bit = 1u << (cam_pose[f] - lo);
      dup |= (m & bit) != 0; m |= bit;
      nib |= (uint64_t)(cam_pose[f] - lo) << (4 * (f - b));
    }
    if (dup) continue;
    cmin[l] = lo; mask[l] = (int)m;
    rec[l] = make_int4(b, (int)(m | ((unsigned)(e - b) << 16)), (int)(uint32_t)nib, (int)(uint32_t)(nib >> 32));
  }
}

struct BandUnits {
  std::vector<int> lm;                // landmarks in unit order
  std::vector<int> unit_start, unit_cam;
  std::vector<int> cam_units;         // camera pose -> first unit with that k0 (ncp + 1)
};

inline void band_units(int nl, const int* cmin, const int* mask, int ncp, BandUnits& out) {
  out.lm.clear(); out.unit_start.clear(); out.unit_cam.clear();
  out.cam_units.assign((size_t)(ncp > 0 ? ncp : 1) + 1, 0);
  // counting sort by (first camera pose, kBandCams - span)
  const size_t nkey = (size_t)(ncp > 0 ? ncp : 1) * 16;
  std::vector<int> start(nkey + 1, 0);
  auto key_of = [&](int l) {
    int top = 0;
    for (unsigned m = (unsigned)mask[l]; m; m >>= 1) ++top;
    return (size_t)cmin[l] * 16 + (size_t)(kBandCams - top);
  };
  int n = 0;
  for (int l = 0; l < nl; ++l) if (cmin[l] >= 0) { start[key_of(l) + 1]++; ++n; }
  if (!n) { out.unit_start.push_back(0); return; }
  for (size_t k = 0; k < nkey; ++k) start[k + 1] += start[k];
  out.lm.resize(n);
  for (int l = 0; l < nl; ++l) if (cmin[l] >= 0) out.lm[start[key_of(l)]++] = l;
  int i = 0;
  while (i < n) {
    const int k0 = cmin[out.lm[i]];
    int j = i;
    while (j < n && cmin[out.lm[j]] == k0) ++j;
    // (parts of equal size rather than full parts and a remainder)
    const int parts = (j - i + kBandPart - 1) / kBandPart;
    for (int p = 0; p < parts; ++p) { out.unit_start.push_back(i + (int)((int64_t)(j - i) * p / parts)); out.unit_cam.push_back(k0); }
    i = j;
  }
  out.unit_start.push_back(n);
  for (int k0 : out.unit_cam) out.cam_units[k0 + 1]++;
  for (size_t k = 0; k + 1 < out.cam_units.size(); ++k) out.cam_units[k + 1] += out.cam_units[k];
}

}  // namespace bsg
