// Device-side flattening of the reprojection factors of a window (the per-cycle rebuild that the
// reference does on the host in HashGraph::createProblem): what the Schur kernels need — factors sorted by landmark,
// camera-pose ids, per-landmark ranges, the (factor a, factor b) pair entries grouped by camera pair in chunks of pair_chunk(entries),
// the tile adjacency of the reduced system — is built from the RAW factor table with rocPRIM sorts and scans instead of
// host loops over 400 k factors and 2 M pair entries.  It produces exactly the tables of the host path (same order:
// both sorts are stable), which stays as the general path (online-calibration factors, landmark blocks shared with
// other factors, small problems) and as the cross-check (BSGPU_FLATTEN=host).
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <functional>
#include <vector>

#include "bsgpu_device.h"
#include "band_plan.h"

namespace bsg {

namespace {

__global__ void fl_prep_kernel(int n, const int* __restrict__ idx /* n x 4 */, const double* __restrict__ consts /* n x 3 */,
                               const int* __restrict__ loss_kind, const double* __restrict__ loss_a, int n_loss,
                               const int* __restrict__ tab_kind, const double* __restrict__ tab_a, const int* __restrict__ blk_xoff,
                               const unsigned char* __restrict__ blk_const, const int* __restrict__ blk_lm, int nl,
                               unsigned* __restrict__ key, int4* __restrict__ fac, double2* __restrict__ pix, double* __restrict__ w,
                               int* __restrict__ lm_of, int* __restrict__ bq_of, int* __restrict__ used_q, int* __restrict__ p_of_q,
                               int* __restrict__ flags_out /* [0] fallback, [1] some factor fully constant */,
                               const int* __restrict__ slot_map /* the idx columns hold caller slots (device-resident table) */) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const int4 row = reinterpret_cast<const int4*>(idx)[f];
  const int bq = slot_map ? slot_map[row.x] : row.x, bp = slot_map ? slot_map[row.y] : row.y, bl = slot_map ? slot_map[row.z] : row.z;
  const int cam = row.w;
  const int lm = blk_lm[bl];
  const int cq = blk_const[bq], cp = blk_const[bp], cl = blk_const[bl];
  if (lm < 0 && !cl) flags_out[0] = 1;           // a landmark block that is neither eliminated nor constant: host path
  const int fl = (cq ? kFlagQConst : 0) | (cp ? kFlagPConst : 0) | (cl ? kFlagLConst : 0);
  if (fl == 7) flags_out[1] = 1;
  int kind = loss_kind[f];
  double a = loss_a[f];
  if (kind == BSGPU_LOSS_TRIVIAL) a = 1.0;
  int lid = -1;
  for (int i = 0; i < n_loss; ++i) if (tab_kind[i] == kind && tab_a[i] == a) { lid = i; break; }
  if (lid < 0) { flags_out[0] = 1; lid = 0; }
  key[f] = (unsigned)(lm < 0 ? nl : lm);
  fac[f] = make_int4(blk_xoff[bq], blk_xoff[bp], blk_xoff[bl], meta_pack(cam, lid, fl));
  pix[f] = make_double2(consts[3 * f], consts[3 * f + 1]);
  w[f] = consts[3 * f + 2];
  lm_of[f] = lm;
  bq_of[f] = bq;
  used_q[bq] = 1;
  const int old = atomicCAS(&p_of_q[bq], -1, bp);
  if (old != -1 && old != bp) flags_out[0] = 1;  // one orientation block paired with two position blocks: host path
}

__global__ void fl_cp_kernel(int nb, const int* __restrict__ used_q, const int* __restrict__ cp_of_q, const int* __restrict__ p_of_q,
                             const int* __restrict__ blk_toff, int* __restrict__ cp_tq, int* __restrict__ cp_tp) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb || !used_q[b]) return;
  cp_tq[cp_of_q[b]] = blk_toff[b];
  cp_tp[cp_of_q[b]] = blk_toff[p_of_q[b]];
}

__global__ void fl_gather_kernel(int n, const int* __restrict__ order, const int4* __restrict__ fac_in, const double2* __restrict__ pix_in,
                                 const double* __restrict__ w_in, const int* __restrict__ lm_in, const int* __restrict__ bq_in,
                                 const int* __restrict__ cp_of_q, int4* __restrict__ fac, double2* __restrict__ pix,
                                 double* __restrict__ w, int* __restrict__ lm_of, int* __restrict__ cam_pose, int* __restrict__ src,
                                 int* __restrict__ lm_cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int f = order[i];
  fac[i] = fac_in[f]; pix[i] = pix_in[f]; w[i] = w_in[f];
  const int lm = lm_in[f];
  lm_of[i] = lm;
  cam_pose[i] = cp_of_q[bq_in[f]];
  src[i] = f;                       // type 0: (0 << 28) | f
  if (lm >= 0) atomicAdd(&lm_cnt[lm], 1);
}

// band landmarks (band_plan.h: band_record is the rule, on either side): first camera pose or -1, mask of the slots seen, the record
__global__ void fl_band_kernel(int nl, const int* __restrict__ lm_start, const int* __restrict__ cam_pose, int enabled, int* __restrict__ cmin,
                               int* __restrict__ mask, int4* __restrict__ rec) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nl) return;
  int cm = -1;
  int4 rc = make_int4(0, 0, -1, -1);
  if (enabled) band_record(lm_start[l], lm_start[l + 1], cam_pose, &cm, &rc);
  cmin[l] = cm; mask[l] = cm >= 0 ? (rc.y & 0xffff) : 0; rec[l] = rc;
}
__global__ void fl_band_gather_kernel(int n, const int* __restrict__ order, const int4* __restrict__ rec, int4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rec[order[i]];
}

// pair entries of a landmark: (a, b) over its factors with cam(a) <= cam(b), a-major (the host loop's order); a band landmark has none
__global__ void fl_pair_count_kernel(int nl, const int* __restrict__ lm_start, const int* __restrict__ cam_pose, const int* __restrict__ cmin,
                                     int* __restrict__ cnt) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nl) return;
  int n = 0;
  if (cmin[l] < 0)
    for (int a = lm_start[l]; a < lm_start[l + 1]; ++a)
      for (int b = lm_start[l]; b < lm_start[l + 1]; ++b) n += cam_pose[a] <= cam_pose[b];
  cnt[l] = n;
}
__global__ void fl_pair_gen_kernel(int nl, const int* __restrict__ lm_start, const int* __restrict__ cam_pose, const int* __restrict__ cmin,
                                   const int* __restrict__ off, unsigned long long ncp, unsigned long long* __restrict__ key,
                                   unsigned long long* __restrict__ val) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nl || cmin[l] >= 0) return;
  int o = off[l];
  for (int a = lm_start[l]; a < lm_start[l + 1]; ++a)
    for (int b = lm_start[l]; b < lm_start[l + 1]; ++b)
      if (cam_pose[a] <= cam_pose[b]) {
        key[o] = (unsigned long long)cam_pose[a] * ncp + (unsigned long long)cam_pose[b];
        val[o] = ((unsigned long long)(unsigned)a << 32) | (unsigned)b;
        ++o;
      }
}
// factors of constant landmarks: one (f, f) entry each, after all landmark entries
__global__ void fl_pair_tail_kernel(int n_elim, int n, const int* __restrict__ cam_pose, unsigned long long ncp, int base,
                                    unsigned long long* __restrict__ key, unsigned long long* __restrict__ val) {
  const int i = n_elim + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = base + (i - n_elim);
  key[o] = (unsigned long long)cam_pose[i] * ncp + (unsigned long long)cam_pose[i];
  val[o] = ((unsigned long long)(unsigned)i << 32) | (unsigned)i;
}
__global__ void fl_split_kernel(int n, const unsigned long long* __restrict__ key, const unsigned long long* __restrict__ val,
                                int* __restrict__ fa, int* __restrict__ fb, int* __restrict__ run_flag_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  fa[i] = (int)(val[i] >> 32); fb[i] = (int)(val[i] & 0xffffffffull);
  run_flag_idx[i] = (i == 0 || key[i] != key[i - 1]) ? i : 0;     // start index of a run of equal keys (max-scanned next)
}
__global__ void fl_segflag_kernel(int n, const int* __restrict__ run_start, unsigned char* __restrict__ flag, int chunk_mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = ((i - run_start[i]) & chunk_mask) == 0;
}
__global__ void fl_seg_kernel(int n_seg, int n_ent, int* __restrict__ seg_start, const unsigned long long* __restrict__ key,
                              unsigned long long ncp, int* __restrict__ seg_ci, int* __restrict__ seg_cj, const int* __restrict__ cp_tq,
                              const int* __restrict__ cp_tp, unsigned char* __restrict__ tile_adj, int T) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j == 0) seg_start[n_seg] = n_ent;
  if (j >= n_seg) return;
  const unsigned long long k = key[seg_start[j]];
  const int ci = (int)(k / ncp), cj = (int)(k % ncp);
  seg_ci[j] = ci; seg_cj[j] = cj;
  const int ri[2] = {cp_tq[ci], cp_tp[ci]}, rj[2] = {cp_tq[cj], cp_tp[cj]};
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
    if (ri[a] < 0 || rj[b] < 0) continue;
    for (int x = ri[a]; x < ri[a] + 3; x += 2) for (int y = rj[b]; y < rj[b] + 3; y += 2) {
      tile_adj[(size_t)(x / 64) * T + y / 64] = 1; tile_adj[(size_t)(y / 64) * T + x / 64] = 1;
    }
  }
}

// changed rows of the device-resident slot-named table (bsgpu_sync_factors_indirect): packed (row, idx x 4, consts x 3, loss) -> place
__global__ void fl_patch_kernel(int n_ch, const int* __restrict__ rows, const int4* __restrict__ idx4, const double* __restrict__ consts3,
                                const int* __restrict__ lk, const double* __restrict__ la, int4* __restrict__ dst_idx,
                                double* __restrict__ dst_consts, int* __restrict__ dst_lk, double* __restrict__ dst_la) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_ch) return;
  const size_t r = (size_t)rows[i];
  dst_idx[r] = idx4[i];
  dst_consts[3 * r] = consts3[3 * i]; dst_consts[3 * r + 1] = consts3[3 * i + 1]; dst_consts[3 * r + 2] = consts3[3 * i + 2];
  dst_lk[r] = lk[i]; dst_la[r] = la[i];
}

}  // namespace

void launch_patch_factor_rows(hipStream_t s, int n_ch, const int* rows, const int* idx4, const double* consts3, const int* lk, const double* la,
                              int* dst_idx, double* dst_consts, int* dst_lk, double* dst_la) {
  if (n_ch <= 0) return;
  hipLaunchKernelGGL(fl_patch_kernel, dim3((n_ch + 255) / 256), dim3(256), 0, s, n_ch, rows, reinterpret_cast<const int4*>(idx4), consts3, lk, la,
                     reinterpret_cast<int4*>(dst_idx), dst_consts, dst_lk, dst_la);
}

// returns 0 = tables built, 1 = this window needs the host path, < 0 = device error
int flatten_visual_device(hipStream_t s, const std::function<void*(size_t)>& dalloc, int n, const int* h_idx, const double* h_consts,
                          const int* h_loss_kind, const double* h_loss_a, const std::vector<DevLoss>& losses, int nb, const int* d_blk_xoff,
                          const int* d_blk_toff, const unsigned char* d_blk_const, const int* d_blk_lm, int nl, int T, Visual& V,
                          int** d_vis_src, std::vector<unsigned char>& tile_adj, bool* any_all_const, const FlattenResident* res,
                          FlattenSegsHost* segs_out, bool band_enabled) {
  auto A = [&](size_t bytes) { return dalloc(bytes ? bytes : 8); };
#define FL_CHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return -1; } } while (0)
  const int g256 = (n + 255) / 256;
  // (with a device-resident table — bsgpu_sync_factors_indirect — nothing is copied: the 20 MB of a C2 window stay where they are)
  int* d_idx = res ? const_cast<int*>(res->idx) : (int*)A(sizeof(int) * 4 * (size_t)n);
  double* d_consts = res ? const_cast<double*>(res->consts) : (double*)A(sizeof(double) * 3 * (size_t)n);
  int* d_lk = res ? const_cast<int*>(res->loss_kind) : (int*)A(sizeof(int) * (size_t)n);
  double* d_la = res ? const_cast<double*>(res->loss_a) : (double*)A(sizeof(double) * (size_t)n);
  const int n_loss = (int)losses.size();
  std::vector<int> tk(n_loss); std::vector<double> ta(n_loss);
  for (int i = 0; i < n_loss; ++i) { tk[i] = losses[i].kind; ta[i] = losses[i].a; }
  int* d_tk = (int*)A(sizeof(int) * n_loss); double* d_ta = (double*)A(sizeof(double) * n_loss);
  unsigned *d_key = (unsigned*)A(sizeof(unsigned) * (size_t)n), *d_key2 = (unsigned*)A(sizeof(unsigned) * (size_t)n);
  int4* d_fac0 = (int4*)A(sizeof(int4) * (size_t)n); double2* d_pix0 = (double2*)A(sizeof(double2) * (size_t)n);
  double* d_w0 = (double*)A(sizeof(double) * (size_t)n);
  int *d_lm0 = (int*)A(sizeof(int) * (size_t)n), *d_bq0 = (int*)A(sizeof(int) * (size_t)n),
      *d_order = (int*)A(sizeof(int) * (size_t)n);
  int *d_used = (int*)A(sizeof(int) * ((size_t)nb + 1)), *d_pofq = (int*)A(sizeof(int) * (size_t)nb), *d_cpofq = (int*)A(sizeof(int) * ((size_t)nb + 1));
  int* d_flags = (int*)A(sizeof(int) * 4);
  if (!d_idx || !d_consts || !d_lk || !d_la || !d_key || !d_key2 || !d_fac0 || !d_pix0 || !d_w0 || !d_lm0 || !d_bq0 || !d_order ||
      !d_used || !d_pofq || !d_cpofq || !d_flags || !d_tk || !d_ta) return -1;
  if (!res) {
    FL_CHK(hipMemcpyAsync(d_idx, h_idx, sizeof(int) * 4 * (size_t)n, hipMemcpyHostToDevice, s));
    FL_CHK(hipMemcpyAsync(d_consts, h_consts, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    FL_CHK(hipMemcpyAsync(d_lk, h_loss_kind, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    FL_CHK(hipMemcpyAsync(d_la, h_loss_a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
  }
  FL_CHK(hipMemcpyAsync(d_tk, tk.data(), sizeof(int) * n_loss, hipMemcpyHostToDevice, s));
  FL_CHK(hipMemcpyAsync(d_ta, ta.data(), sizeof(double) * n_loss, hipMemcpyHostToDevice, s));
  FL_CHK(hipMemsetAsync(d_used, 0, sizeof(int) * ((size_t)nb + 1), s));
  FL_CHK(hipMemsetAsync(d_pofq, 0xff, sizeof(int) * (size_t)nb, s));
  FL_CHK(hipMemsetAsync(d_flags, 0, sizeof(int) * 4, s));
  hipLaunchKernelGGL(fl_prep_kernel, dim3(g256), dim3(256), 0, s, n, d_idx, d_consts, d_lk, d_la, n_loss, d_tk, d_ta, d_blk_xoff, d_blk_const,
                     d_blk_lm, nl, d_key, d_fac0, d_pix0, d_w0, d_lm0, d_bq0, d_used, d_pofq, d_flags, res ? res->slot_map : nullptr);
  // camera-pose ids: exclusive scan of the used orientation blocks (ascending block index = the host's (q, p) order)
  size_t tmp_bytes = 0, need = 0;
  auto grow = [&](size_t b) { if (b > tmp_bytes) tmp_bytes = b; };
  FL_CHK(rocprim::exclusive_scan(nullptr, need, d_used, d_cpofq, 0, (size_t)nb + 1, rocprim::plus<int>(), s)); grow(need);
  int end_bit = 1;
  while ((1u << end_bit) <= (unsigned)(nl + 1) && end_bit < 32) ++end_bit;
  FL_CHK(rocprim::radix_sort_pairs(nullptr, need, d_key, d_key2, rocprim::counting_iterator<int>(0), d_order, (size_t)n, 0, end_bit, s)); grow(need);
  void* d_tmp = A(tmp_bytes);
  if (!d_tmp) return -1;
  FL_CHK(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_used, d_cpofq, 0, (size_t)nb + 1, rocprim::plus<int>(), s));
  int h_ncp = 0, h_flags[4] = {0, 0, 0, 0};
  FL_CHK(hipMemcpyAsync(&h_ncp, d_cpofq + nb, sizeof(int), hipMemcpyDeviceToHost, s));
  FL_CHK(hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
  FL_CHK(hipStreamSynchronize(s));                                  // sync #1: n_cam_pose, fallback flags
  if (h_flags[0]) return 1;
  *any_all_const = h_flags[1] != 0;
  V.n = n; V.n_lm = nl; V.n_cam_pose = h_ncp;
  V.cp_tq = (int*)A(sizeof(int) * (size_t)h_ncp); V.cp_tp = (int*)A(sizeof(int) * (size_t)h_ncp);
  hipLaunchKernelGGL(fl_cp_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, nb, d_used, d_cpofq, d_pofq, d_blk_toff, V.cp_tq, V.cp_tp);
  FL_CHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_key2, rocprim::counting_iterator<int>(0), d_order, (size_t)n, 0, end_bit, s));
  V.fac = (int4*)A(sizeof(int4) * (size_t)n); V.pix = (double2*)A(sizeof(double2) * (size_t)n); V.w = (double*)A(sizeof(double) * (size_t)n);
  V.lm_of = (int*)A(sizeof(int) * (size_t)n); V.cam_pose = (int*)A(sizeof(int) * (size_t)n);
  *d_vis_src = (int*)A(sizeof(int) * (size_t)n);
  int* d_lmcnt = (int*)A(sizeof(int) * ((size_t)nl + 1));
  V.lm_start = (int*)A(sizeof(int) * ((size_t)nl + 1));
  int *d_pcnt = (int*)A(sizeof(int) * ((size_t)nl + 1)), *d_poff = (int*)A(sizeof(int) * ((size_t)nl + 1));
  if (!V.fac || !V.pix || !V.w || !V.lm_of || !V.cam_pose || !*d_vis_src || !d_lmcnt || !V.lm_start || !d_pcnt || !d_poff) return -1;
  FL_CHK(hipMemsetAsync(d_lmcnt, 0, sizeof(int) * ((size_t)nl + 1), s));
  FL_CHK(hipMemsetAsync(d_pcnt, 0, sizeof(int) * ((size_t)nl + 1), s));
  hipLaunchKernelGGL(fl_gather_kernel, dim3(g256), dim3(256), 0, s, n, d_order, d_fac0, d_pix0, d_w0, d_lm0, d_bq0, d_cpofq, V.fac, V.pix, V.w,
                     V.lm_of, V.cam_pose, *d_vis_src, d_lmcnt);
  size_t need2 = 0;
  FL_CHK(rocprim::exclusive_scan(nullptr, need2, d_lmcnt, V.lm_start, 0, (size_t)nl + 1, rocprim::plus<int>(), s));
  void* d_tmp2 = need2 > tmp_bytes ? A(need2) : d_tmp;
  size_t tmp2_bytes = need2 > tmp_bytes ? need2 : tmp_bytes;
  if (!d_tmp2) return -1;
  FL_CHK(rocprim::exclusive_scan(d_tmp2, tmp2_bytes, d_lmcnt, V.lm_start, 0, (size_t)nl + 1, rocprim::plus<int>(), s));
  int *d_bcmin = (int*)A(sizeof(int) * ((size_t)nl + 1)), *d_bmask = (int*)A(sizeof(int) * ((size_t)nl + 1));
  int4* d_brec = (int4*)A(sizeof(int4) * ((size_t)nl + 1));
  if (!d_bcmin || !d_bmask || !d_brec) return -1;
  std::vector<int> h_bcmin(nl), h_bmask(nl);
  if (nl > 0) {
    hipLaunchKernelGGL(fl_band_kernel, dim3((nl + 255) / 256), dim3(256), 0, s, nl, V.lm_start, V.cam_pose, band_enabled ? 1 : 0, d_bcmin, d_bmask, d_brec);
    hipLaunchKernelGGL(fl_pair_count_kernel, dim3((nl + 255) / 256), dim3(256), 0, s, nl, V.lm_start, V.cam_pose, d_bcmin, d_pcnt);
    FL_CHK(hipMemcpyAsync(h_bcmin.data(), d_bcmin, sizeof(int) * (size_t)nl, hipMemcpyDeviceToHost, s));
    FL_CHK(hipMemcpyAsync(h_bmask.data(), d_bmask, sizeof(int) * (size_t)nl, hipMemcpyDeviceToHost, s));
  }
  FL_CHK(rocprim::exclusive_scan(d_tmp2, tmp2_bytes, d_pcnt, d_poff, 0, (size_t)nl + 1, rocprim::plus<int>(), s));
  int h_nelim = 0, h_npairs = 0;
  FL_CHK(hipMemcpyAsync(&h_nelim, V.lm_start + nl, sizeof(int), hipMemcpyDeviceToHost, s));
  FL_CHK(hipMemcpyAsync(&h_npairs, d_poff + nl, sizeof(int), hipMemcpyDeviceToHost, s));
  FL_CHK(hipStreamSynchronize(s));                                  // sync #2: entry count, band landmarks
  // units of the band kernel: a counting sort of the landmarks on the host (band_plan.h) while the device sorts the pair entries
  BandUnits bu;
  band_units(nl, h_bcmin.data(), h_bmask.data(), h_ncp, bu);
  V.n_band_lm = (int)bu.lm.size(); V.n_band_units = (int)bu.unit_cam.size();
  V.band_lm = (int4*)A(sizeof(int4) * bu.lm.size()); V.band_unit_start = (int*)A(sizeof(int) * bu.unit_start.size());
  V.band_unit_cam = (int*)A(sizeof(int) * bu.unit_cam.size());
  int* d_border = (int*)A(sizeof(int) * bu.lm.size());
  if (!V.band_lm || !V.band_unit_start || !V.band_unit_cam || !d_border) return -1;
  // (pageable sources that live to the end of this function, past the last synchronisation)
  if (!bu.lm.empty()) {
    FL_CHK(hipMemcpyAsync(d_border, bu.lm.data(), sizeof(int) * bu.lm.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(fl_band_gather_kernel, dim3(((int)bu.lm.size() + 255) / 256), dim3(256), 0, s, (int)bu.lm.size(), d_border, d_brec, V.band_lm);
    V.band_lm_id = d_border;   // (kept: the band kernel's look-up of a landmark's Linv and z, Visual::no_cr)
  }
  FL_CHK(hipMemcpyAsync(V.band_unit_start, bu.unit_start.data(), sizeof(int) * bu.unit_start.size(), hipMemcpyHostToDevice, s));
  if (!bu.unit_cam.empty()) FL_CHK(hipMemcpyAsync(V.band_unit_cam, bu.unit_cam.data(), sizeof(int) * bu.unit_cam.size(), hipMemcpyHostToDevice, s));
  V.n_elim = h_nelim;
  const int n_ent = h_npairs + (n - h_nelim);
  V.n_ent = n_ent;
  const unsigned long long ncp = (unsigned long long)(h_ncp > 0 ? h_ncp : 1);
  unsigned long long *d_ek = (unsigned long long*)A(8 * (size_t)n_ent), *d_ev = (unsigned long long*)A(8 * (size_t)n_ent),
                     *d_ek2 = (unsigned long long*)A(8 * (size_t)n_ent), *d_ev2 = (unsigned long long*)A(8 * (size_t)n_ent);
  V.ent_fa = (int*)A(sizeof(int) * (size_t)n_ent); V.ent_fb = (int*)A(sizeof(int) * (size_t)n_ent);
  int *d_runidx = (int*)A(sizeof(int) * (size_t)n_ent), *d_runstart = (int*)A(sizeof(int) * (size_t)n_ent);
  unsigned char* d_segflag = (unsigned char*)A((size_t)n_ent);
  int* d_segsel = (int*)A(sizeof(int) * ((size_t)n_ent + 1));
  int* d_nseg = (int*)A(sizeof(int) * 2);
  if (!d_ek || !d_ev || !d_ek2 || !d_ev2 || !V.ent_fa || !V.ent_fb || !d_runidx || !d_runstart || !d_segflag || !d_segsel || !d_nseg) return -1;
  if (nl > 0 && h_npairs > 0) hipLaunchKernelGGL(fl_pair_gen_kernel, dim3((nl + 255) / 256), dim3(256), 0, s, nl, V.lm_start, V.cam_pose, d_bcmin, d_poff, ncp, d_ek, d_ev);
  if (n > h_nelim) hipLaunchKernelGGL(fl_pair_tail_kernel, dim3((n - h_nelim + 255) / 256), dim3(256), 0, s, h_nelim, n, V.cam_pose, ncp, h_npairs, d_ek, d_ev);
  int kbits = 1;
  while (kbits < 64 && (ncp * ncp) >> kbits) ++kbits;
  size_t need3 = 0, need4 = 0, need5 = 0;
  if (n_ent > 0) {
    FL_CHK(rocprim::radix_sort_pairs(nullptr, need3, d_ek, d_ek2, d_ev, d_ev2, (size_t)n_ent, 0, kbits, s));
    FL_CHK(rocprim::inclusive_scan(nullptr, need4, d_runidx, d_runstart, (size_t)n_ent, rocprim::maximum<int>(), s));
    FL_CHK(rocprim::select(nullptr, need5, rocprim::counting_iterator<int>(0), d_segflag, d_segsel, d_nseg, (size_t)n_ent, s));
    size_t need345 = need3 > need4 ? need3 : need4;
    if (need5 > need345) need345 = need5;
    void* d_tmp3 = need345 > tmp2_bytes ? A(need345) : d_tmp2;
    size_t tmp3_bytes = need345 > tmp2_bytes ? need345 : tmp2_bytes;
    if (!d_tmp3) return -1;
    FL_CHK(rocprim::radix_sort_pairs(d_tmp3, tmp3_bytes, d_ek, d_ek2, d_ev, d_ev2, (size_t)n_ent, 0, kbits, s));
    const int ge = (n_ent + 255) / 256;
    hipLaunchKernelGGL(fl_split_kernel, dim3(ge), dim3(256), 0, s, n_ent, d_ek2, d_ev2, V.ent_fa, V.ent_fb, d_runidx);
    FL_CHK(rocprim::inclusive_scan(d_tmp3, tmp3_bytes, d_runidx, d_runstart, (size_t)n_ent, rocprim::maximum<int>(), s));
    hipLaunchKernelGGL(fl_segflag_kernel, dim3(ge), dim3(256), 0, s, n_ent, d_runstart, d_segflag, pair_chunk((size_t)n_ent) - 1);
    FL_CHK(rocprim::select(d_tmp3, tmp3_bytes, rocprim::counting_iterator<int>(0), d_segflag, d_segsel, d_nseg, (size_t)n_ent, s));
  } else {
    FL_CHK(hipMemsetAsync(d_nseg, 0, sizeof(int) * 2, s));
  }
  int h_nseg = 0;
  FL_CHK(hipMemcpyAsync(&h_nseg, d_nseg, sizeof(int), hipMemcpyDeviceToHost, s));
  FL_CHK(hipStreamSynchronize(s));                                  // sync #3: segment count
  V.n_seg = h_nseg;
  V.n_seg_c = 0; V.seg_ci_c = V.seg_cj_c = V.seg_start_c = nullptr;   // (the coarse list of the batch: rebuilt on first use)
  V.seg_start = d_segsel;   // n_seg selected start indices (+ the end marker written below; the buffer has n_ent + 1 slots)
  V.seg_ci = (int*)A(sizeof(int) * (size_t)(h_nseg > 0 ? h_nseg : 1)); V.seg_cj = (int*)A(sizeof(int) * (size_t)(h_nseg > 0 ? h_nseg : 1));
  unsigned char* d_adj = (unsigned char*)A((size_t)T * T + 8);
  if (!V.seg_ci || !V.seg_cj || !d_adj) return -1;
  FL_CHK(hipMemsetAsync(d_adj, 0, (size_t)T * T + 8, s));
  hipLaunchKernelGGL(fl_seg_kernel, dim3((h_nseg + 256) / 256), dim3(256), 0, s, h_nseg, n_ent, V.seg_start, d_ek2, ncp, V.seg_ci, V.seg_cj, V.cp_tq,
                     V.cp_tp, d_adj, T);
  tile_adj.assign((size_t)T * T, 0);
  if (T > 0) FL_CHK(hipMemcpyAsync(tile_adj.data(), d_adj, (size_t)T * T, hipMemcpyDeviceToHost, s));
  // the camera-pose pairs themselves, for the block-level ordering of the reduced system (dim_order.h); the pairs of the band landmarks
  // have no segments: they come from BandUnits::adj, as further (i, j) pairs and as marks of the tile adjacency
  std::vector<int> h_tq, h_tp;
  const bool want_cp = segs_out || V.n_band_lm > 0;
  if (segs_out) {
    segs_out->seg_ci.resize(h_nseg); segs_out->seg_cj.resize(h_nseg);
    if (h_nseg > 0) {
      FL_CHK(hipMemcpyAsync(segs_out->seg_ci.data(), V.seg_ci, sizeof(int) * (size_t)h_nseg, hipMemcpyDeviceToHost, s));
      FL_CHK(hipMemcpyAsync(segs_out->seg_cj.data(), V.seg_cj, sizeof(int) * (size_t)h_nseg, hipMemcpyDeviceToHost, s));
    }
  }
  if (want_cp && h_ncp > 0) {
    h_tq.resize(h_ncp); h_tp.resize(h_ncp);
    FL_CHK(hipMemcpyAsync(h_tq.data(), V.cp_tq, sizeof(int) * (size_t)h_ncp, hipMemcpyDeviceToHost, s));
    FL_CHK(hipMemcpyAsync(h_tp.data(), V.cp_tp, sizeof(int) * (size_t)h_ncp, hipMemcpyDeviceToHost, s));
  }
  FL_CHK(hipStreamSynchronize(s));
  FL_CHK(hipGetLastError());
  for (int i = 0; i < h_ncp && V.n_band_lm > 0; ++i)
    for (int d = 0; d < kBandCams; ++d) {
      if (!((bu.adj[i] >> d) & 1u)) continue;
      const int j = i + d;
      if (segs_out) { segs_out->seg_ci.push_back(i); segs_out->seg_cj.push_back(j); }
      const int ri[2] = {h_tq[i], h_tp[i]}, rj[2] = {h_tq[j], h_tp[j]};
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        if (ri[a] < 0 || rj[b] < 0) continue;
        for (int x = ri[a]; x < ri[a] + 3; x += 2) for (int y = rj[b]; y < rj[b] + 3; y += 2) {
          tile_adj[(size_t)(x / 64) * T + y / 64] = 1; tile_adj[(size_t)(y / 64) * T + x / 64] = 1;
        }
      }
    }
  if (segs_out) { segs_out->cp_tq = h_tq; segs_out->cp_tp = h_tp; }
#undef FL_CHK
  return 0;
}

}  // namespace bsg
