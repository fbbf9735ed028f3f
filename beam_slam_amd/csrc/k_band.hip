// Band landmarks of the visual Schur complement on the matrix cores (round 5; SURVEY.md §8a: the JᵀJ / Schur step that Ceres' SchurEliminator
// does for bs_optimizers/src/fixed_lag_smoother.cpp:281 `graph_->optimize()`, [EXT] ceres/internal/schur_eliminator_impl.h).
//
// A feature track of a sliding window is seen from consecutive key frames.  With W_a = A_aᵀ C_a (6 x 3; A the pose part of the Jacobian
// row pair, C = B L⁻ᵀ of landmark_kernel) the pair block of observations a, b of one landmark is
//     A_aᵀ ([a = b] I − C_a C_bᵀ) A_b = [a = b] A_aᵀ A_a − W_a W_bᵀ,
// so a landmark whose first camera pose is k0 contributes −Z Zᵀ with Z = its W_a stacked at rows 6 (cam_a − k0): 78 x 3, zero rows where a
// camera pose does not see it.  pairs_kernel (k_reproj.hip) forms these blocks entry by entry — 2.0 M (a, b) entries on C2, 304 B of gathered
// rows each, 54 sums per lane: 0.55 GB through the L1s and two waves per SIMD, 69 us.  Here every observation is read ONCE:
//   * one workgroup (kBandWaves waves) per unit = (k0, a part of its landmarks, ordered by falling span; band_plan.h);
//   * the unit's landmarks go through in sub-batches of 32.  FORM: lane (slot j = camera pose k0 + j, landmark) loads the observation's rows
//     (A 96 B, C and rho 64 B, r 16 B), forms W (36 FMAs) and writes it to LDS as Z[row 6 j + m][column 3 landmark + k]; the lane keeps the
//     sums over ITS camera pose of A_aᵀ A_a, A_aᵀ rho_a, A_aᵀ r_a (33 accumulators; no exchange until the unit ends).  The next sub-batch's
//     rows are requested before this one is multiplied;
//   * MULTIPLY: the 15 lower 16 x 16 tiles of Z Zᵀ are dealt out to the waves (two each), K = 96 per sub-batch on v_mfma_f64_16x16x4, both
//     operands straight from LDS (row pitch 98 doubles: the sixteen rows x four columns of an operand fall on 64 different banks); tile rows
//     beyond the sub-batch's widest landmark are skipped (a track of <= 5 key frames touches 3 of 15 tiles);
//   * at the end of the unit the tiles are SUBTRACTED from S with FP64 atomics (both triangles, as pairs_kernel does), and the per-camera
//     sums reach S's diagonal blocks, the reduced rhs row, the gradient and diag(H) after a transposed butterfly inside each 32-lane half.
// Landmarks that do not qualify (a span of more than kBandCams camera poses, two observations from one camera pose) and the factors of
// constant landmarks keep their pair entries in pairs_kernel; the pose-only factors that rode in the pair launch ride here.
#include <mutex>
#include <vector>
#include <cstdio>
#include <atomic>

#include "bsgpu_device.h"

namespace bsg {

constexpr int kBandWaves = 8;                     // waves of a workgroup (4 waves and two workgroups per compute unit: 59.5 us against 55.6 at one unit per first camera pose)
constexpr int kBandThreads = 64 * kBandWaves;
constexpr int kBandNL = kBandThreads / 16;        // landmarks per sub-batch (sixteen loader threads each)
constexpr int kBandRows = 80;                     // 6 kBandCams = 78 rows in five tiles of 16
constexpr int kBandPitch = 3 * kBandNL + 2;       // doubles per row of Z: half of it odd (the 16 rows x 4 columns of an operand fall on 64 banks)
constexpr int kBandKSteps = 3 * kBandNL / 4;      // K = 4 steps of a sub-batch
constexpr int kBandNT = (15 + kBandWaves - 1) / kBandWaves;   // tiles of the lower triangle per wave: wave, wave + kBandWaves, ...
static_assert((kBandPitch / 2) % 2 == 1 && 3 * kBandNL % 4 == 0 && 64 % kBandNL == 0 && kBandWaves * (64 / kBandNL) >= kBandCams, "band kernel shape");
// 16-byte pieces per landmark in the stage (A: 6 per observation, C | rho: 4, r: 1; odd strides: the 16-byte reads of the forming lanes,
// one landmark each, fall on different banks)
constexpr int kBandStA = 81, kBandStC = 65, kBandStR = 17;   // (>= 16 x 5, 16 x 4, 16: every loader thread stores every piece it requested, no conditions)
constexpr size_t kBandLds = sizeof(double) * kBandRows * kBandPitch + sizeof(int) * kBandRows + 16 * (size_t)kBandNL * (kBandStA + kBandStC + kBandStR + 1);
static_assert(kBandCams * 6 <= kBandRows && kBandCams <= 13, "slot nibbles, tile rows");
static_assert(sizeof(double) * (15 * 30 + 16) + sizeof(int) * 16 <= kBandLds, "a riding pose-only factor's staging");

typedef double band_d4 __attribute__((ext_vector_type(4)));

BSG_DEV void band_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// record of landmark li for the loader (beyond the unit: no observations, no slots)
BSG_DEV int4 band_fetch_rec(const int4* __restrict__ band_lm, int li, int end) {
  int4 t = band_lm[li < end ? li : end - 1];
  if (li >= end) { t.y = 0; t.z = -1; t.w = -1; }
  return t;
}
// The 16-byte pieces of one landmark's rows that a thread of its 16-thread loader group holds between the request and the staging: pieces
// l16 + 16 i of the landmark's A rows (6 per observation), of its C | rho rows (4 per observation), of its residuals (1 per observation).
// Plain named locals and unconditional loads at clamped indices: a struct handed to helpers is an object in memory across the barriers' asm
// (it lived in scratch: 176 bytes per lane), and arrays of loaded values under a lane condition went to scratch in round 3's band kernel.
// (NOCR — Visual::no_cr: the stage's C area takes the landmark's B rows, 3 pieces per observation, in pieces 0 .. 47, and the landmark's Linv (pieces
//  48 .. 50) and z (pieces 51, 52: z0 z1 | z2 pad — the landmark's 80-byte record as it lies in memory) instead of the C | rho rows; LMID = the record's landmark)
#define BSG_BAND_ISSUE(REC, LMID)                                                                                             \
  {                                                                                                                           \
    const int n_ = (int)((unsigned)(REC).y >> 24);                                                                            \
    const int cpo_ = NOCR ? 3 : 4;                                                                                            \
    const int na_ = 6 * n_ - 1 > 0 ? 6 * n_ - 1 : 0, nc_ = cpo_ * n_ - 1 > 0 ? cpo_ * n_ - 1 : 0, nr_ = n_ - 1 > 0 ? n_ - 1 : 0; \
    const double2* Jf_ = J2 + (size_t)(REC).x * (kJAStride / 2);                                                              \
    const double2* Cf_ = C2 + (size_t)(REC).x * cpo_;                                                                         \
    const double2* rf_ = r + (size_t)(REC).x;                                                                                 \
    pa0 = Jf_[min(l16, na_)]; pa1 = Jf_[min(l16 + 16, na_)]; pa2 = Jf_[min(l16 + 32, na_)]; pa3 = Jf_[min(l16 + 48, na_)];     \
    pa4 = Jf_[min(l16 + 64, na_)];                                                                                            \
    pc0 = Cf_[min(l16, nc_)]; pc1 = Cf_[min(l16 + 16, nc_)]; pc2 = Cf_[min(l16 + 32, nc_)];                                    \
    if (NOCR) pc3 = reinterpret_cast<const double2*>(Linv + (size_t)(LMID) * kLmRec)[min(l16, 4)];   /* the landmark's record: Linv | z | pad */ \
    else pc3 = Cf_[min(l16 + 48, nc_)];                                                                                       \
    pr0 = rf_[min(l16, nr_)];                                                                                                 \
    prec = (REC);                                                                                                             \
  }
#define BSG_BAND_STAGE()                                                                                                      \
  {                                                                                                                           \
    double2* a_ = sA + g * kBandStA; double2* c_ = sC + g * kBandStC; double2* rr_ = sR + g * kBandStR;                        \
    a_[l16] = pa0; a_[l16 + 16] = pa1; a_[l16 + 32] = pa2; a_[l16 + 48] = pa3; a_[l16 + 64] = pa4;                             \
    c_[l16] = pc0; c_[l16 + 16] = pc1; c_[l16 + 32] = pc2; c_[l16 + 48] = pc3;                                                 \
    rr_[l16] = pr0;                                                                                                           \
    if (l16 == 0) sRec[g] = prec;                                                                                             \
  }

// The products of one sub-batch for the first NACT tiles of a wave: K = 4 step k + 1's operands are requested from LDS before step k's
// products are issued.  Written with the reads and their waits as asm: left to the compiler the loop was read, wait, multiply in every round
// — also fully unrolled — and the LDS latency was paid at every step (51 of a unit's 88 thousand cycles, measured).  The wait names the
// registers it releases, so the products that use them cannot be moved above it.
template <int NACT>
BSG_DEV void band_multiply(const unsigned (&za)[kBandNT], const unsigned (&zb)[kBandNT], band_d4 (&acc)[kBandNT]) {
  static_assert(NACT >= 1 && NACT <= 4, "tiles per wave");
  if (NACT > kBandNT) return;
  double a[2][4], b[2][4];
#define BSG_DSR(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#pragma unroll
  for (int j = 0; j < NACT && j < kBandNT; ++j) { BSG_DSR(a[0][j], za[j], 0); BSG_DSR(b[0][j], zb[j], 0); }
#pragma unroll
  for (int k = 0; k < kBandKSteps; ++k) {
    const int cur = k & 1, nxt = cur ^ 1;
    if (k + 1 < kBandKSteps) {
#pragma unroll
      for (int j = 0; j < NACT && j < kBandNT; ++j) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[nxt][j]) : "v"(za[j]), "n"(32 * (k + 1 < kBandKSteps ? k + 1 : 0)));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[nxt][j]) : "v"(zb[j]), "n"(32 * (k + 1 < kBandKSteps ? k + 1 : 0)));
      }
      if (NACT == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[cur][0]), "+v"(b[cur][0]));
      if (NACT == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]));
      if (NACT == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]), "+v"(a[cur][2]), "+v"(b[cur][2]));
      if (NACT == 4) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]), "+v"(a[cur][2]), "+v"(b[cur][2]), "+v"(a[cur][3]), "+v"(b[cur][3]));
    } else {
      if (NACT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(b[cur][0]));
      if (NACT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]));
      if (NACT == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]), "+v"(a[cur][2]), "+v"(b[cur][2]));
      if (NACT == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(b[cur][0]), "+v"(a[cur][1]), "+v"(b[cur][1]), "+v"(a[cur][2]), "+v"(b[cur][2]), "+v"(a[cur][3]), "+v"(b[cur][3]));
    }
#pragma unroll
    for (int j = 0; j < NACT && j < kBandNT; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][j], b[cur][j], acc[j], 0, 0, 0);
  }
#undef BSG_DSR
}

// NOCR: CR = the landmark parts of the Jacobian rows (Visual::JB, 48 B per observation), lm_id / Linv / z = the records' landmarks and their inverse factors
template <bool NOCR = false>
__device__ __forceinline__ void pairs_band_kernel_body(const int bsg_bx, const int bsg_gx, int n_units, const int* __restrict__ unit_start, const int* __restrict__ unit_cam, const int4* __restrict__ band_lm, int n_cam_pose, const double* __restrict__ J, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only, const SmallGroupSet& small, int n_small_units, const int* __restrict__ lm_id = nullptr, const double* __restrict__ Linv = nullptr, const double* __restrict__ z = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double bsm[];
  const int tid = threadIdx.x;
  const int u = bsg_bx;
  if (u >= n_units) {
    // the pose-only factors assembled one workgroup per factor (IMU: two or three hundred of them) as extra workgroups of this launch:
    // independent atomics into the same system, and a launch of their own cost ~8 us on the dependent path (as in pairs_kernel).  They come
    // LAST: the units, each as long as the launch, start at once, and these fill the compute units the units leave free
    if (u - n_units < n_small_units)
      small_assemble_unit(small, u - n_units, tid, kBandThreads, bsm, bsm + 15 * 30, reinterpret_cast<int*>(bsm + 15 * 30 + 16), S, ld, rhs_row, grad, hdiag, perm);
    return;
  }
  double* Zs = bsm;
  int* pos = reinterpret_cast<int*>(bsm + kBandRows * kBandPitch);
  double2* sA = reinterpret_cast<double2*>(pos + kBandRows);
  double2* sC = sA + kBandNL * kBandStA;
  double2* sR = sC + kBandNL * kBandStC;
  int4* sRec = reinterpret_cast<int4*>(sR + kBandNL * kBandStR);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = unit_cam[u], beg = unit_start[u], end = unit_start[u + 1];
  // solver position of every row of the unit's block (-1: padding, a constant block, beyond the last camera pose)
  if (tid < kBandRows) {
    const int sl = tid / 6, m = tid - 6 * sl, cam = k0 + sl;
    int p = -1;
    if (sl < kBandCams && cam < n_cam_pose) {
      const int t = m < 3 ? cp_tq[cam] : cp_tp[cam];
      if (t >= 0) p = perm[t + (m < 3 ? m : m - 3)];
    }
    pos[tid] = p;
  }
  for (int q = tid; q < kBandRows * kBandPitch; q += kBandThreads) Zs[q] = 0.0;   // (rows 78, 79 and the pitch's padding are never written again)
  // LOAD: sixteen threads per landmark of a sub-batch fetch its contiguous rows 16 bytes at a time (two lines per request of a group: with one
  // lane per observation every request of a wave looked 64 lines up and the kernel ran at that rate — 56 us) and stage them in LDS
  const int g = tid >> 4, l16 = tid & 15;
  const double2* J2 = reinterpret_cast<const double2*>(J);
  const double2* C2 = reinterpret_cast<const double2*>(CR);
  double2 pa0, pa1, pa2, pa3, pa4, pc0, pc1, pc2, pc3, pr0;
  int4 prec;
  auto fetch_id = [&](int li) { return NOCR ? lm_id[li < end ? li : end - 1] : 0; };
  {
    const int4 rec0 = band_fetch_rec(band_lm, beg + g, end);
    const int id0 = fetch_id(beg + g);
    BSG_BAND_ISSUE(rec0, id0)
  }
  int4 rec_next = band_fetch_rec(band_lm, beg + kBandNL + g, end);
  int id_next = fetch_id(beg + kBandNL + g);
  // FORM: this lane's camera-pose slot and landmark of a sub-batch
  const int slot = wave + kBandWaves * (lane / kBandNL), lmk = lane % kBandNL;
  const bool slot_ok = slot < kBandCams;
  double* zrow = Zs + (size_t)(6 * (slot_ok ? slot : 0)) * kBandPitch + 3 * lmk;
  double AtA[21], Atp[6], Atr[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) AtA[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) { Atp[i] = 0.0; Atr[i] = 0.0; }
  // MULTIPLY: this wave's tiles (R >= C) of the lower triangle, tile t = R (R + 1) / 2 + C: t = wave + kBandWaves j.  Their rows R grow with j,
  // so the tiles a sub-batch reaches are the first `nact` of them.
  const int i16 = lane & 15, kq = lane >> 4;
  int tR[kBandNT], tC[kBandNT];
  unsigned zaddr_a[kBandNT], zaddr_b[kBandNT];   // LDS byte addresses of this lane's operand elements at K = 0
  band_d4 acc[kBandNT];
#pragma unroll
  for (int j = 0; j < kBandNT; ++j) {
    const int t = wave + kBandWaves * j;
    int R = 0;
    while ((R + 1) * (R + 2) / 2 <= t) ++R;
    tR[j] = t < 15 ? R : 99; tC[j] = t < 15 ? t - R * (R + 1) / 2 : 0;
    const int Ra = t < 15 ? R : 0;
    zaddr_a[j] = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(Zs + (size_t)(16 * Ra + i16) * kBandPitch + kq);
    zaddr_b[j] = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(Zs + (size_t)(16 * tC[j] + i16) * kBandPitch + kq);
    acc[j] = band_d4{0.0, 0.0, 0.0, 0.0};
  }
  const int T0 = (6 * ((band_lm[beg].y >> 16) & 0xff) + 15) >> 4;   // tile rows of the unit's widest landmark (its first)
  __syncthreads();   // (the clearing of Z and `pos` before the first sub-batch is written)
  for (int base = beg; base < end; base += kBandNL) {
    BSG_BAND_STAGE()
    // the next sub-batch's rows are on their way while this one is formed and multiplied; its records came a round earlier.  (Unconditional:
    // past the unit's end the records say "no observations" and the requests fall on a row of its last landmark.)
    BSG_BAND_ISSUE(rec_next, id_next)
    rec_next = band_fetch_rec(band_lm, base + 2 * kBandNL + g, end);
    id_next = fetch_id(base + 2 * kBandNL + g);
    band_lds_sync();   // (the stage is complete; every wave is past the previous sub-batch's products)
    {
      const int4 rec = sRec[lmk];
      const unsigned long long inv = (unsigned long long)(unsigned)rec.z | ((unsigned long long)(unsigned)rec.w << 32);
      const int oi = (int)((inv >> (4 * slot)) & 15ull);
      const bool valid = slot_ok && oi != 15;
      // (unconditional reads at a clamped observation, then a select: loads under a lane condition into array elements go to scratch memory;
      //  the stage may hold anything where no row was written — a product with zero would not do)
      const int oc = valid ? oi : 0;
      const double2* a = sA + lmk * kBandStA + 6 * oc;
      const double2* c = sC + lmk * kBandStC + (NOCR ? 3 : 4) * oc;
      const double2 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[NOCR ? 2 : 3];
      const double2 rr = sR[lmk * kBandStR + oc];
#define BSG_SEL(x) (valid ? (x) : 0.0)
      const double A0[6] = {BSG_SEL(a0.x), BSG_SEL(a0.y), BSG_SEL(a1.x), BSG_SEL(a1.y), BSG_SEL(a2.x), BSG_SEL(a2.y)};
      const double A1[6] = {BSG_SEL(a3.x), BSG_SEL(a3.y), BSG_SEL(a4.x), BSG_SEL(a4.y), BSG_SEL(a5.x), BSG_SEL(a5.y)};
      double Ca[3] = {BSG_SEL(c0.x), BSG_SEL(c0.y), BSG_SEL(c1.x)}, Cb[3] = {BSG_SEL(c1.y), BSG_SEL(c2.x), BSG_SEL(c2.y)};
      const double r0 = BSG_SEL(rr.x), r1 = BSG_SEL(rr.y);
      double p0 = BSG_SEL(c3.x), p1 = BSG_SEL(c3.y);
      if (NOCR) {
        // what landmark_kernel's second pass wrote: C = B Linv^T (Linv lower triangular: i00 | i10 i11 | i20 i21 i22), rho = r - C z
        const double2* lz = sC + lmk * kBandStC + 48;
        const double2 l0 = lz[0], l1 = lz[1], l2 = lz[2], z01 = lz[3];
        const double z0 = z01.x, z1 = z01.y, z2 = lz[4].x;
        const double x0 = Ca[0], x1 = Ca[1], x2 = Ca[2], y0 = Cb[0], y1 = Cb[1], y2 = Cb[2];
        Ca[0] = x0 * l0.x; Ca[1] = x0 * l0.y + x1 * l1.x; Ca[2] = x0 * l1.y + x1 * l2.x + x2 * l2.y;
        Cb[0] = y0 * l0.x; Cb[1] = y0 * l0.y + y1 * l1.x; Cb[2] = y0 * l1.y + y1 * l2.x + y2 * l2.y;
        p0 = r0 - (Ca[0] * z0 + Ca[1] * z1 + Ca[2] * z2);
        p1 = r1 - (Cb[0] * z0 + Cb[1] * z1 + Cb[2] * z2);
      }
#undef BSG_SEL
      if (slot_ok) {
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int k = 0; k < 3; ++k) zrow[m * kBandPitch + k] = fma(A0[m], Ca[k], A1[m] * Cb[k]);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) AtA[a * (a + 1) / 2 + b] = fma(A0[a], A0[b], fma(A1[a], A1[b], AtA[a * (a + 1) / 2 + b]));
        Atp[a] = fma(A0[a], p0, fma(A1[a], p1, Atp[a]));
        Atr[a] = fma(A0[a], r0, fma(A1[a], r1, Atr[a]));
      }
    }
    const int top = __builtin_amdgcn_readfirstlane((sRec[0].y >> 16) & 0xff);   // span of the sub-batch's widest landmark (they are ordered by falling span)
    band_lds_sync();   // (Z is complete; every wave is done with the stage)
    if (!(grad_only & 1)) {
      const int T = (6 * top + 15) >> 4;                        // tile rows that hold something (wave-uniform)
      int nact = 0;
#pragma unroll
      for (int j = 0; j < kBandNT; ++j) nact += tR[j] < T;
      // (every K step of the sub-batch: the columns past its last landmark are zero)
      switch (nact) {
        case 1: band_multiply<1>(zaddr_a, zaddr_b, acc); break;
        case 2: band_multiply<2>(zaddr_a, zaddr_b, acc); break;
        case 3: band_multiply<3>(zaddr_a, zaddr_b, acc); break;
        case 4: band_multiply<4>(zaddr_a, zaddr_b, acc); break;
        default: break;
      }
    }
  }
  // the unit's tiles out of S: up to 6 400 atomic adds per unit, which the device completes at ~80 G a second whoever issues them
  if (!(grad_only & 1)) {
#pragma unroll
    for (int j = 0; j < kBandNT; ++j) {
      const int R = tR[j], C = tC[j];
      if (R >= T0) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const double v = acc[j][reg];
        if (v == 0.0) continue;
        const int pr = pos[16 * R + 4 * reg + kq], pc = pos[16 * C + i16];
        if (pr < 0 || pc < 0) continue;
        if (grad_only & 2) {
          // LOWER TRIANGLE ONLY (in solver order): what the tiled factorisation reads — tiles below the diagonal whole, of a diagonal tile the 16 x 16 blocks on
          // and below its block diagonal, of those on it their lower triangles (chol_chain.h load_slot / elim_pivots; the update tasks' operands are tiles below
          // the diagonal).  Half of a unit's adds went to places nothing reads, and the units' adds are a fifth of the launch (~120 G adds a second, all at its end).
          if (R != C) atomicAdd(&S[(size_t)max(pr, pc) * ld + min(pr, pc)], -v);
          else if (pr >= pc) atomicAdd(&S[(size_t)pr * ld + pc], -v);   // (a diagonal tile of the unit's block holds (a, b) and (b, a) itself)
          continue;
        }
        atomicAdd(&S[(size_t)pr * ld + pc], -v);
        if (R != C) atomicAdd(&S[(size_t)pc * ld + pr], -v);
      }
    }
  }
  // the sums over this lane's camera pose: transposed butterfly inside the kBandNL lanes of the slot — lane l of them ends up with the values
  // (32 / kBandNL) l .. of the 32 — the 33rd by itself
  double v[32];
#pragma unroll
  for (int i = 0; i < 21; ++i) v[i] = AtA[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[21 + i] = Atp[i];
#pragma unroll
  for (int i = 0; i < 5; ++i) v[27 + i] = Atr[i];
  double v32 = Atr[5];
#pragma unroll
  for (int o = kBandNL / 2, h = 16; o > 0; o >>= 1, h >>= 1) {
    const bool upper = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const double lo = v[i], hi = v[i + h];
      const double recv = __shfl_xor(upper ? lo : hi, o, 64);
      v[i] = (upper ? hi : lo) + recv;
    }
    v32 += __shfl_xor(v32, o, 64);
  }
  const int cam = k0 + slot;
  if (!slot_ok || cam >= n_cam_pose) return;
  const int tq = cp_tq[cam], tp = cp_tp[cam];
  auto tang = [&](int m) { return m < 3 ? (tq < 0 ? -1 : tq + m) : (tp < 0 ? -1 : tp + m - 3); };
#pragma unroll
  for (int q = 0; q <= 32 / kBandNL; ++q) {
    // value idx of (A^T A lower triangle 0..20 | A^T rho 21..26 | A^T r 27..32)
    const bool last = q == 32 / kBandNL;
    if (last && lmk != 0) break;
    const int idx = last ? 32 : lmk * (32 / kBandNL) + q;
    const double total = last ? v32 : v[q < 32 / kBandNL ? q : 0];
    if (total == 0.0) continue;
    if (idx < 21) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= idx) ++a;
      const int b = idx - a * (a + 1) / 2;
      const int pa = pos[6 * slot + a], pb = pos[6 * slot + b];
      if (pa >= 0 && pb >= 0) {
        if (grad_only & 2) atomicAdd(&S[(size_t)max(pa, pb) * ld + min(pa, pb)], total);
        else {
          atomicAdd(&S[(size_t)pa * ld + pb], total);
          if (a != b) atomicAdd(&S[(size_t)pb * ld + pa], total);
        }
        if (a == b) atomicAdd(&hdiag[tang(a)], total);
      }
    } else if (idx < 27) {
      const int pa = pos[6 * slot + idx - 21];
      if (pa >= 0) atomicAdd(&S[(size_t)rhs_row * ld + pa], total);
    } else {
      const int ra = tang(idx - 27);
      if (ra >= 0) atomicAdd(&grad[ra], total);
    }
  }
}

__global__ __launch_bounds__(kBandThreads) void pairs_band_kernel(int n_units, const int* __restrict__ unit_start, const int* __restrict__ unit_cam, const int4* __restrict__ band_lm, int n_cam_pose, const double* __restrict__ J, const double2* __restrict__ r, const double* __restrict__ CR, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only, SmallGroupSet small, int n_small_units, GoWord go) {
  if (go.p && !(__hip_atomic_load(go.p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0.0)) return;   // (an assembly ahead whose step was not accepted)
  pairs_band_kernel_body((int)blockIdx.x, (int)gridDim.x, n_units, unit_start, unit_cam, band_lm, n_cam_pose, J, r, CR, cp_tq, cp_tp, S, ld, rhs_row, grad, hdiag, perm, grad_only, small, n_small_units);
}
// ... without C rows (Visual::no_cr): the landmark parts of the Jacobian rows instead, and the landmarks' Linv and z
__global__ __launch_bounds__(kBandThreads) void pairs_band_nocr_kernel(int n_units, const int* __restrict__ unit_start, const int* __restrict__ unit_cam, const int4* __restrict__ band_lm, int n_cam_pose, const double* __restrict__ J, const double2* __restrict__ r, const double* __restrict__ JB, const int* __restrict__ cp_tq, const int* __restrict__ cp_tp, double* __restrict__ S, int ld, int rhs_row, double* __restrict__ grad, double* __restrict__ hdiag, const int* __restrict__ perm, int grad_only, SmallGroupSet small, int n_small_units, const int* __restrict__ lm_id, const double* __restrict__ Linv, const double* __restrict__ z, GoWord go) {
  if (go.p && !(__hip_atomic_load(go.p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0.0)) return;
  pairs_band_kernel_body<true>((int)blockIdx.x, (int)gridDim.x, n_units, unit_start, unit_cam, band_lm, n_cam_pose, J, r, JB, cp_tq, cp_tp, S, ld, rhs_row, grad, hdiag, perm, grad_only, small, n_small_units, lm_id, Linv, z);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct pairs_band_kernel_Args {
  int bsg_grid;
  int n_units;
  const int* unit_start;
  const int* unit_cam;
  const int4* band_lm;
  int n_cam_pose;
  const double* J;
  const double2* r;
  const double* CR;
  const int* cp_tq;
  const int* cp_tp;
  double* S;
  int ld;
  int rhs_row;
  double* grad;
  double* hdiag;
  const int* perm;
  int grad_only;
  SmallGroupSet small;
  int n_small_units;
  // (Visual::no_cr — all null otherwise: the landmark parts of the Jacobian rows, the band records' landmarks, their Linv | z records)
  const double* JB;
  const int* lm_id;
  const double* Linv;
  const double* z;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct pairs_band_kernel_ArgsG {
  int bsg_grid;
  int n_units;
  const int __attribute__((address_space(1)))* unit_start;
  const int __attribute__((address_space(1)))* unit_cam;
  const int4 __attribute__((address_space(1)))* band_lm;
  int n_cam_pose;
  const double __attribute__((address_space(1)))* J;
  const double2 __attribute__((address_space(1)))* r;
  const double __attribute__((address_space(1)))* CR;
  const int __attribute__((address_space(1)))* cp_tq;
  const int __attribute__((address_space(1)))* cp_tp;
  double __attribute__((address_space(1)))* S;
  int ld;
  int rhs_row;
  double __attribute__((address_space(1)))* grad;
  double __attribute__((address_space(1)))* hdiag;
  const int __attribute__((address_space(1)))* perm;
  int grad_only;
  SmallGroupSet small;
  int n_small_units;
  const double __attribute__((address_space(1)))* JB;
  const int __attribute__((address_space(1)))* lm_id;
  const double __attribute__((address_space(1)))* Linv;
  const double __attribute__((address_space(1)))* z;
};
static_assert(sizeof(pairs_band_kernel_ArgsG) == sizeof(pairs_band_kernel_Args), "layout");

__global__ __launch_bounds__(kBandThreads) void pairs_band_kernel_batch(const pairs_band_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.y];
  const pairs_band_kernel_ArgsG& a = reinterpret_cast<const pairs_band_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.x >= a.bsg_grid) return;
  // (a batch's systems are read by the fused tiled factorisation next: the units add into the lower triangle only, as a lone step's do — bit 1;
  //  a window's gradient-only step adds no tiles at all)
  const int band_mode = bsg_dyn->grad_only[bsg_w] ? 1 : 2;
  // (a window without C rows — Visual::no_cr — forms them from B, Linv and z as its lone launch does)
  if (a.Linv)
    pairs_band_kernel_body<true>((int)blockIdx.x, a.bsg_grid, a.n_units, (const int*)a.unit_start, (const int*)a.unit_cam, (const int4*)a.band_lm, a.n_cam_pose, (const double*)a.J, (const double2*)a.r, (const double*)a.JB, (const int*)a.cp_tq, (const int*)a.cp_tp, (double*)a.S, a.ld, a.rhs_row, (double*)a.grad, (double*)a.hdiag, (const int*)a.perm, band_mode, a.small, a.n_small_units, (const int*)a.lm_id, (const double*)a.Linv, (const double*)a.z);
  else
  pairs_band_kernel_body((int)blockIdx.x, a.bsg_grid, a.n_units, (const int*)a.unit_start, (const int*)a.unit_cam, (const int4*)a.band_lm, a.n_cam_pose, (const double*)a.J, (const double2*)a.r, (const double*)a.CR, (const int*)a.cp_tq, (const int*)a.cp_tp, (double*)a.S, a.ld, a.rhs_row, (double*)a.grad, (double*)a.hdiag, (const int*)a.perm, band_mode, a.small, a.n_small_units);
}

// The band kernels need kBandLds (~147 KB) of dynamic LDS per workgroup: the opt-in is made once per device (keyed by the whole device id)
// and its outcome is kept — finalize() asks band_available() and plans the pair-entry path on a device or partition that cannot give it.
static int band_attr_state(int dev) {   // 1: available, -1: not
  static std::mutex mu;
  static std::vector<signed char> state;
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0) return -1;
  if ((size_t)dev >= state.size()) state.resize((size_t)dev + 1, 0);
  if (state[dev] == 0) {
    int max_lds = 0;
    bool ok = hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && max_lds >= (int)kBandLds;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(pairs_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLds) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(pairs_band_kernel_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLds) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(pairs_band_nocr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLds) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); fprintf(stderr, "[bsgpu] device %d gives %d B of LDS per workgroup, the band kernel needs %d: pair-entry path\n", dev, max_lds, (int)kBandLds); }
    state[dev] = ok ? 1 : -1;
  }
  return state[dev];
}
bool band_available() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return band_attr_state(dev) > 0;
}
static void band_attr_once() { (void)band_available(); }

void launch_pairs_band(hipStream_t s, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, bool grad_only,
                       const SmallGroupSet* small, int n_small_units, bool lower_only, GoWord go) {
  if (v.n_band_units == 0) return;
  band_attr_once();
  SmallGroupSet none;
  none.n = 0;
  const int riders = small ? n_small_units : 0;
  if (v.no_cr)
    hipLaunchKernelGGL(pairs_band_nocr_kernel, dim3(riders + v.n_band_units), dim3(kBandThreads), kBandLds, s, v.n_band_units, v.band_unit_start, v.band_unit_cam, v.band_lm,
                       v.n_cam_pose, v.J, v.r, v.JB, v.cp_tq, v.cp_tp, S, ld, rhs_row, grad, hdiag, perm, (grad_only ? 1 : 0) | (lower_only ? 2 : 0), small ? *small : none, riders, v.band_lm_id,
                       v.Linv, v.z, go);
  else
  hipLaunchKernelGGL(pairs_band_kernel, dim3(riders + v.n_band_units), dim3(kBandThreads), kBandLds, s, v.n_band_units, v.band_unit_start, v.band_unit_cam, v.band_lm,
                     v.n_cam_pose, v.J, v.r, v.CR, v.cp_tq, v.cp_tp, S, ld, rhs_row, grad, hdiag, perm, (grad_only ? 1 : 0) | (lower_only ? 2 : 0), small ? *small : none, riders, go);
}
void batchargs_pairs_band(BatchArgTable& t, const Visual& v, double* S, int ld, int rhs_row, double* grad, double* hdiag, const int* perm, const SmallGroupSet* small,
                          int n_small_units) {
  pairs_band_kernel_Args a;
  SmallGroupSet none;
  none.n = 0;
  const int riders = small ? n_small_units : 0;
  a.bsg_grid = v.n_band_units > 0 ? riders + v.n_band_units : 0;
  a.n_units = v.n_band_units; a.unit_start = v.band_unit_start; a.unit_cam = v.band_unit_cam; a.band_lm = v.band_lm; a.n_cam_pose = v.n_cam_pose;
  a.J = v.J; a.r = v.r; a.CR = v.CR; a.cp_tq = v.cp_tq; a.cp_tp = v.cp_tp; a.S = S; a.ld = ld; a.rhs_row = rhs_row; a.grad = grad; a.hdiag = hdiag; a.perm = perm;
  a.grad_only = 0; a.small = small ? *small : none; a.n_small_units = riders;
  a.JB = v.no_cr ? v.JB : nullptr; a.lm_id = v.no_cr ? v.band_lm_id : nullptr; a.Linv = v.no_cr ? v.Linv : nullptr; a.z = v.no_cr ? v.z : nullptr;
  t.push(a);
}
void launch_pairs_band_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  band_attr_once();
  hipLaunchKernelGGL(pairs_band_kernel_batch, dim3(t.max_grid, n), dim3(kBandThreads), kBandLds, s, static_cast<const pairs_band_kernel_Args*>(t.dev), dyn, list);
}

}  // namespace bsg
