// Blocked FP64 Cholesky of the reduced camera system on gfx950: one kernel per schedule step; a step runs
// one 64-wide panel of every independent sub-chain of the nested-dissection ordering (dense_plan.h).
//
//   chol_potrf_tile   factor one 64x64 diagonal tile (used for tile 0 and for tiles no panel step
//                     reaches); writes L and the inverses of its four 16x16 diagonal blocks
//   chol_panel_step   workgroup (i, j), j <= i, over the row tiles the envelope keeps active in panel k:
//                       X_i = A_ik L_kk^-T, X_j = A_jk L_kk^-T   (block substitution, MFMA)
//                       C_ij -= X_i X_j^T                          (MFMA)
//                     the workgroup that owns tile (k+1, k+1) factors it right away (look-ahead), so the
//                     dependent chain potrf -> trsm -> update -> potrf costs one launch per panel
//   chol_backsolve    L^T y = y' in ONE workgroup marching up the panels (y' = rhs row carried through
//                     the factorisation as row n_pose)
//
// All matrix products run on v_mfma_f64_16x16x4_f64:  a = A[lane&15][lane>>4], b = B[lane>>4][lane&15],
// d[reg] = D[(lane>>4) + 4 reg][lane&15].  The 16x16 diagonal blocks are factored in registers of one
// wave (row per lane, v_readlane broadcasts), the only scalar dependent chain left.
#include <chrono>
#include <cstring>
#include <thread>

#include "bsgpu_device.h"
#include "chol_chain.h"
#include "dense_plan.h"

namespace bsg {

namespace {

constexpr int NB = 64;
constexpr int LDT = 66;
constexpr int kVinvStride = 1024 + 64;  // per tile: 4 inverse 16x16 diagonal blocks + 64 reciprocal pivots  // LDS row pitch of a 64x64 tile: conflict-free ds_read_b64 for the MFMA fragments
typedef double double4_t __attribute__((ext_vector_type(4)));

BSG_DEV double readlane_d(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// D += sign * A B^T over k in [0, K): A rows = 16 rows at sA (pitch lda), B rows = 16 rows at sB (pitch ldb)
template <int K>
BSG_DEV double4_t mfma_abt(double4_t acc, const double* sA, int lda, const double* sB, int ldb, double sign, int lane) {
  const int r = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int k = 0; k < K; k += 4) {
    const double a = sign * sA[r * lda + k + kq];
    const double b = sB[r * ldb + k + kq];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}
BSG_DEV double4_t load_d(const double* s, int ld, int lane) {
  double4_t v;
  const int r0 = lane >> 4, c = lane & 15;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) v[reg] = s[(r0 + 4 * reg) * ld + c];
  return v;
}
BSG_DEV void store_d(double* s, int ld, int lane, double4_t v) {
  const int r0 = lane >> 4, c = lane & 15;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) s[(r0 + 4 * reg) * ld + c] = v[reg];
}

// 1/sqrt(d) to double precision: v_rsq_f64 seed (~2^-26) + two Newton steps, all FMA — keeps the
// per-column dependent chain of the factorisation short (an IEEE sqrt + divide costs ~3x as much)
BSG_DEV double fast_rsqrt(double d) {
  // v_rsq_f64 seed (relative error ~2^-26) + ONE cubically convergent step, all FMA: the result is within a few ulp.
  // A second (linear) clean-up step would make it almost correctly rounded, but this sits on the dependent chain of
  // every pivot of the factorisation, and the reduced system is assembled with more round-off than that.
  const double y = __builtin_amdgcn_rsq(d);
  const double t = fma(-d * y, y, 1.0);         // 1 - d y^2
  return fma(y * t, fma(t, 0.375, 0.5), y);     // y (1 + t/2 + 3 t^2/8)
}

// Factor the 64x64 tile in sC (lower triangle meaningful, pitch LDT) in place; the four inverse
// diagonal blocks go to sV (4 x 16 x 16, row-major, lower), the 64 reciprocal pivots to sInvD.
// Whole workgroup (256 threads).  Columns >= nreal (rhs row, padding) are unit pivots.
// Returns (in every thread) whether a non-positive / non-finite pivot was met.
template <bool PROBE = false, int NT = 256>
BSG_DEV bool potrf64_lds(double* sC, double* sV, double* sInvD /* 64 */, int tid, int nreal, long long* ts = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  bool bad = false;
  int nts = 0;
  auto stamp = [&]() { if (PROBE && tid == 0) ts[nts++] = wall_clock64(); };
  stamp();
#pragma unroll 1
  for (int b = 0; b < 4; ++b) {
    if (wave == 0) {
      // (1)+(2) right-looking elimination of block column b in registers of one wave: lane l owns tile
      // row 16b + l (the 16 diagonal-block rows first, then every row below), 16 entries each.  The
      // pivot row entries travel by v_readlane from lanes 0..15; rows below get their triangular
      // solve from the very same updates.  Straight-line code (selects, no branches) and the NEXT pivot's
      // reciprocal square root started before the remaining updates of the current one, so that its
      // dependent chain overlaps with their issue slots.
      const int nrows = NB - 16 * b;
      const int row = 16 * b + ((lane < nrows) ? lane : 0);
      double* R = sC + row * LDT + 16 * b;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = R[c];
      // the rhs row's own diagonal entry has collected -|y'|^2 from the trailing updates: unit pivot
      // Columns >= nreal (padding of the last tile) need no special case here: their rows and columns are zero apart
      // from a unit diagonal (pose_diag_kernel / mask_unreal_columns), nothing ever updates them, so the plain
      // recurrence gives pivot 1 and an empty column.
      double d = readlane_d(a[0], 0);
      if (!(d > 0.0) || !(d < 1.7e308)) bad = true;
      double inv = fast_rsqrt(d);
      double my_inv = 0.0;   // lane j keeps pivot j's reciprocal: ONE store after the loop (a predicated LDS store per pivot
                             // splits the loop into sixteen basic blocks, and the scheduler can no longer overlap anything)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        my_inv = (lane == j) ? inv : my_inv;
        a[j] = a[j] * inv;
        if (j < 15) {
          const double l1 = readlane_d(a[j], j + 1);
          a[j + 1] = fma(-a[j], l1, a[j + 1]);
          d = readlane_d(a[j + 1], j + 1);
          if (!(d > 0.0) || !(d < 1.7e308)) bad = true;
          inv = fast_rsqrt(d);
        }
#pragma unroll
        for (int c = j + 2; c < 16; ++c) {
          const double lc = readlane_d(a[j], c);
          a[c] = fma(-a[j], lc, a[c]);
        }
      }
      if (lane < 16) sInvD[16 * b + lane] = my_inv;
      if (lane < nrows) {
#pragma unroll
        for (int c = 0; c < 16; ++c) R[c] = (lane >= 16 || c <= lane) ? a[c] : 0.0;
      }
    }
    __syncthreads();
    stamp();
    // (3) trailing update C_rc -= X_rb X_cb^T for b < c <= r <= 3 (MFMA, one 16x16 block per wave pass)
    const int nb = 3 - b;
    const int npairs = nb * (nb + 1) / 2;
    for (int p = wave; p < npairs; p += NT / 64) {
      int rr = 0, acc_cnt = 0;
      while (acc_cnt + rr + 1 <= p) { acc_cnt += rr + 1; ++rr; }
      const int cc = p - acc_cnt;
      const int Rb = b + 1 + rr, Cc = b + 1 + cc;
      double* Cblk = sC + (16 * Rb) * LDT + 16 * Cc;
      double4_t acc = load_d(Cblk, LDT, lane);
      acc = mfma_abt<16>(acc, sC + (16 * Rb) * LDT + 16 * b, LDT, sC + (16 * Cc) * LDT + 16 * b, LDT, -1.0, lane);
      store_d(Cblk, LDT, lane, acc);
    }
    __syncthreads();
    stamp();
  }
  // inverses of the four diagonal blocks: wave w, lane c < 16 solves L_ww v = e_c
  if (wave < 4 && lane < 16) {
    const double* D = sC + (16 * wave) * LDT + 16 * wave;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s = fma(-D[i * LDT + k], v[k], s);
      v[i] = s * sInvD[16 * wave + i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sV[wave * 256 + i * 16 + lane] = v[i];
  }
  __syncthreads();
  stamp();
  return __syncthreads_or(bad ? 1 : 0) != 0;
}

// columns >= nreal of a diagonal tile (rhs row, padding) become unit pivots with nothing below
template <int NT = 256>
BSG_DEV void mask_unreal_columns(double* sC, int nreal, int tid) {
  if (nreal >= NB) return;
  for (int i = tid; i < NB * NB; i += NT) {
    const int r = i >> 6, c = i & 63;
    if (c >= nreal && c <= r) sC[r * LDT + c] = (r == c) ? 1.0 : 0.0;
  }
}

// L_tt goes to the shadow matrix Lp (what the back-substitution reads) and, when S is given, also in place into S
// (what the NEXT panel step loads; not allowed when other workgroups of this launch still read the raw tile)
BSG_DEV void write_factor(double* S, double* Lp, int ld, int t, const double* sC, const double* sV, const double* sInvD, double* Vinv,
                          int tid) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = tid + 256 * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    const size_t at = (size_t)(t * NB + r) * ld + t * NB + c2;
    if (c2 + 1 <= r) {
      const double2 v = *reinterpret_cast<const double2*>(&sC[r * LDT + c2]);
      *reinterpret_cast<double2*>(&Lp[at]) = v;
      if (S) *reinterpret_cast<double2*>(&S[at]) = v;
    } else if (c2 <= r) {
      Lp[at] = sC[r * LDT + c2];
      if (S) S[at] = sC[r * LDT + c2];
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = (tid + 256 * q) * 2;
    *reinterpret_cast<double2*>(&Vinv[(size_t)t * kVinvStride + i]) = *reinterpret_cast<const double2*>(&sV[i]);
  }
  if (tid < NB) Vinv[(size_t)t * kVinvStride + 1024 + tid] = sInvD[tid];
}

}  // namespace

__global__ __launch_bounds__(256) void chol_potrf_tiles_kernel(double* __restrict__ S, double* __restrict__ Lp, int ld,
                                                               const int* __restrict__ tiles, const int* __restrict__ nreal,
                                                               double* __restrict__ Vinv, double* __restrict__ scal) {
  __shared__ double sC[NB * LDT];
  __shared__ double sV[4 * 256];
  __shared__ double sInvD[NB];
  const int tid = threadIdx.x;
  const int t = tiles[blockIdx.x];
  // all sixteen 16-byte loads of a thread are issued before the first one is consumed (a loop with the
  // triangle test inside serialises them: 16 round trips to L2 / HBM, ~7 us for a 32 KB tile)
  double2 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = tid + 256 * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    v[q] = *reinterpret_cast<const double2*>(&S[(size_t)(t * NB + r) * ld + t * NB + c2]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = tid + 256 * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    sC[r * LDT + c2] = (c2 <= r) ? v[q].x : 0.0;
    sC[r * LDT + c2 + 1] = (c2 + 1 <= r) ? v[q].y : 0.0;
  }
  __syncthreads();
  mask_unreal_columns(sC, nreal[t], tid);
  __syncthreads();
  const bool bad = potrf64_lds(sC, sV, sInvD, tid, nreal[t]);
  if (bad && tid == 0) scal[SC_CHOL_FAIL] = 1.0;
  write_factor(S, Lp, ld, t, sC, sV, sInvD, Vinv, tid);
}

// X = A L^-T by 16-column block substitution, in place in sA (64 x 64, pitch LDT); wave w owns rows
// 16w..16w+15.  sL = L_kk (pitch LDT), sV = its inverse diagonal blocks, sT = 4 x (16 x 17) scratch.
BSG_DEV void trsm_tile(double* sA, const double* sL, const double* sV, double* sT, int lane, int wave) {
  double* rows = sA + (16 * wave) * LDT;
  double* T = sT + wave * (16 * 17);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double4_t acc = load_d(rows + 16 * b, LDT, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < b) acc = mfma_abt<16>(acc, rows + 16 * c, LDT, sL + (16 * b) * LDT + 16 * c, LDT, -1.0, lane);
    store_d(T, 17, lane, acc);
    __builtin_amdgcn_wave_barrier();
    double4_t x = {0.0, 0.0, 0.0, 0.0};
    x = mfma_abt<16>(x, T, 17, sV + b * 256, 16, 1.0, lane);
    __builtin_amdgcn_wave_barrier();
    store_d(rows + 16 * b, LDT, lane, x);
    __builtin_amdgcn_wave_barrier();
  }
}

// The same solve with nothing but constants read from LDS on the dependent path: the strip is kept TRANSPOSED in accumulator
// registers.  With Y_b = X_b^T (16 x 16: row = column inside block b, column = row of the strip)
//     Y_b = V_b (A_b^T - sum_{c<b} L_bc Y_c)
// and the MFMA result layout d[reg] = D[(lane >> 4) + 4 reg][lane & 15] is exactly the B-operand layout of the next product
// (K slice kk <-> register kk), so a solved block feeds the updates of the later blocks — and an updated block its own solve —
// straight from registers; the A operands (V_b, -L_bc) are constants of the panel.  40 MFMAs per 16-row strip and no LDS round
// trip between them (trsm_tile: a store / barrier / reload per block; 3.3 us per tile against 1.2 us, scripts/chol_probe.py).
// (sV: block b at b * kVtPitch * 16, rows kVtPitch doubles apart: with 16 the sixteen lanes of a fragment read hit two banks)
constexpr int kVtPitch = 18;
BSG_DEV void trsm_tile_t(double* sA, const double* sL, const double* sV, int lane, int wave) {
  double* rows = sA + (16 * wave) * LDT;
  const int n = lane & 15, q = lane >> 4;
  double4_t acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) acc[b][reg] = rows[n * LDT + 16 * b + q + 4 * reg];   // A_b^T in result layout
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[b * 16 * kVtPitch + n * kVtPitch + 4 * kk + q], acc[b][kk], y, 0, 0, 0);
#pragma unroll
    for (int c = b + 1; c < 4; ++c)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-sL[(16 * c + n) * LDT + 16 * b + 4 * kk + q], y[kk], acc[c], 0, 0, 0);
    acc[b] = y;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) rows[n * LDT + 16 * b + q + 4 * reg] = acc[b][reg];
}

// The strip X = A L_kk^-T as a PRODUCT with the tile's full inverse W = L_kk^-1 (what the chains leave for every tile: identity rows
// carried through the elimination, chol_chain.h): Y_b = X_b^T = sum_{c <= b} W_bc A_c^T — ten 16 x 16 x 16 block products per strip,
// the same 40 MFMAs as the substitution above, but none of them waits for another one's result: the substitution is four dependent
// block steps (a dependent v_mfma_f64 costs ~175 cycles against 64 of issue: 2.8 us per solve in the task stamps), the product runs at
// the issue rate.  sW: W row-major with pitch LDT (zeros above the diagonal).  Round 2 had measured this form with W built by MFMA
// products inside the task (2.2 us, the gain gone); since round 3 W comes for free.
BSG_DEV void solve_tile_w(double* sA, const double* sW, int lane, int wave) {
  double* rows = sA + (16 * wave) * LDT;
  const int n = lane & 15, q = lane >> 4;
  double4_t a[4], y[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) a[b][reg] = rows[n * LDT + 16 * b + q + 4 * reg];   // A_b^T in result layout = the B operand of the products
    y[b] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int b = c; b < 4; ++b)   // (consecutive MFMAs go to different accumulators)
        y[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sW[(16 * b + n) * LDT + 16 * c + 4 * kk + q], a[c][kk], y[b], 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) rows[n * LDT + 16 * b + q + 4 * reg] = y[b][reg];
}

// descriptors of the (few) panels of one step and their row-tile lists, passed BY VALUE: they arrive with the kernel
// arguments instead of costing two dependent round trips to memory before the first tile load can be issued
constexpr int kStepMaxPanels = 16, kStepMaxRows = 16;
struct StepArgs {
  int k[kStepMaxPanels], n_rows[kStepMaxPanels], final_mask[kStepMaxPanels], shared_mask[kStepMaxPanels], self_potrf[kStepMaxPanels];
  int rows[kStepMaxPanels][kStepMaxRows];
};

template <bool KARG, bool PROBE = false>
__global__ __launch_bounds__(256) void chol_panel_step_kernel(double* __restrict__ S, double* __restrict__ Lp, int ld,
                                                              const PanelDesc* __restrict__ descs,
                                                              const int* __restrict__ rows_flat, const int* __restrict__ nreal,
                                                              double* __restrict__ Vinv, double* __restrict__ scal, int* tile_sync, StepArgs args,
                                                              long long* probe_ts = nullptr) {
  const int bi = blockIdx.y, bj = blockIdx.x, z = blockIdx.z;
  int nts = 0;   // PROBE: wall-clock stamps of workgroup (0,0) for scripts/potrf_probe.hip
  auto stamp = [&]() { if (PROBE && bi == 0 && bj == 0 && threadIdx.x == 0) probe_ts[nts++] = wall_clock64(); };
  stamp();
  int k, n_rows, final_mask, ti, tj, shared_mask, self_potrf;
  if (KARG) {
    k = args.k[z]; n_rows = args.n_rows[z]; final_mask = args.final_mask[z]; shared_mask = args.shared_mask[z]; self_potrf = args.self_potrf[z];
    if (bi >= n_rows || bj > bi) return;
    ti = args.rows[z][bi]; tj = args.rows[z][bj];
  } else {
    const PanelDesc pd = descs[z];
    k = pd.k; n_rows = pd.n_rows; final_mask = pd.final_mask; shared_mask = pd.shared_mask; self_potrf = pd.self_potrf;
    if (bi >= n_rows || bj > bi) return;
    ti = rows_flat[pd.row_off + bi]; tj = rows_flat[pd.row_off + bj];
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sXi = smem;                 // 64 x LDT
  double* sXj = sXi + NB * LDT;       // 64 x LDT
  double* sL = sXj + NB * LDT;        // 64 x LDT
  double* sV = sL + NB * LDT;         // 4 x 256
  double* sT = sV + 4 * 256;          // 4 x 16 x 17
  double* sInvD = sT + 4 * 16 * 17;   // 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ri = ti * NB, rj = tj * NB, c0 = k * NB;
  const bool diag = bi == bj;
  // 16-byte loads (ld, c0 and the LDS pitch are all multiples of 2 doubles).  The upper triangle of
  // L_kk in S holds stale values; trsm_tile only reads its strictly-lower 16x16 blocks.
  // ALL loads of the prologue are issued before the first LDS store: written as load-store pairs (with a branch on `diag` in
  // between) the compiler waited for every load before the next one went out — two dozen dependent round trips to L2 / HBM,
  // 5.7 us of a 24 us step (scripts/potrf_probe.hip); in flight together they cost one.  Unconditional on purpose: a diagonal
  // workgroup has rj == ri and never reads sXj before it reuses it, a self-factoring panel overwrites sV; branches around
  // the loads would keep the staging arrays out of registers (scratch).
  {
    double2 vXi[8], vL[8], vXj[8], vV[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = tid + 256 * q;
      const int r = i >> 5, c2 = (i & 31) * 2;
      vXi[q] = *reinterpret_cast<const double2*>(&S[(size_t)(ri + r) * ld + c0 + c2]);
      vL[q] = *reinterpret_cast<const double2*>(&S[(size_t)(c0 + r) * ld + c0 + c2]);
      vXj[q] = *reinterpret_cast<const double2*>(&S[(size_t)(rj + r) * ld + c0 + c2]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) vV[q] = *reinterpret_cast<const double2*>(&Vinv[(size_t)k * kVinvStride + (tid + 256 * q) * 2]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = tid + 256 * q;
      const int r = i >> 5, c2 = (i & 31) * 2;
      *reinterpret_cast<double2*>(&sXi[r * LDT + c2]) = vXi[q];
      *reinterpret_cast<double2*>(&sL[r * LDT + c2]) = vL[q];
      *reinterpret_cast<double2*>(&sXj[r * LDT + c2]) = vXj[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<double2*>(&sV[(tid + 256 * q) * 2]) = vV[q];
  }
  // C_ij (wave w owns rows 16w.. of the 64x64 tile): fetched now, so that the round trip hides behind the solves
  // (a tile another panel of this step also updates is accumulated with atomics: start from zero)
  const bool shared_tile = ((shared_mask >> (bi < 31 ? bi : 31)) & 1) && ((shared_mask >> (bj < 31 ? bj : 31)) & 1);
  double4_t acc[4];
  const int crow = lane >> 4, ccol = lane & 15;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      acc[t][reg] = shared_tile ? 0.0 : S[(size_t)(ri + 16 * wave + crow + 4 * reg) * ld + rj + 16 * t + ccol];
  __syncthreads();
  if (self_potrf) {
    // tile (k, k) was not factored by a look-ahead (several panels updated it last, or it is the head of a piece):
    // every workgroup of the panel factors its own copy instead of waiting for a launch that does it once;
    // workgroup (0, 0) publishes the factor for the back-substitution (to the shadow matrix: the others still read S)
    mask_unreal_columns(sL, nreal[k], tid);
    __syncthreads();
    const bool bad0 = potrf64_lds(sL, sV, sInvD, tid, nreal[k]);
    if (bi == 0 && bj == 0) {
      if (bad0 && tid == 0) scal[SC_CHOL_FAIL] = 1.0;
      write_factor(nullptr, Lp, ld, k, sL, sV, sInvD, Vinv, tid);
    }
    __syncthreads();
  }
  stamp();
  trsm_tile(sXi, sL, sV, sT, lane, wave);
  if (!diag) trsm_tile(sXj, sL, sV, sT, lane, wave);
  __syncthreads();
  stamp();
  const double* Xj = diag ? sXi : sXj;
  // C_ij -= X_i X_j^T
#pragma unroll
  for (int t = 0; t < 4; ++t)
    acc[t] = mfma_abt<64>(acc[t], sXi + (16 * wave) * LDT, LDT, Xj + (16 * t) * LDT, LDT, -1.0, lane);
  stamp();
  // Tile (ti, ti) receives its last update in this step (dense_plan.h: final_mask): whoever completes it factors it here, so
  // that the step using it as a panel starts from the factor.  A sole updater does so from its registers; with several
  // updaters (atomics) the one that arrives last does, from memory.
  const bool final_tile = diag && bi < 31 && ((final_mask >> bi) & 1);
  const int n_expect = final_tile ? tile_sync[ti] : 0;
  const bool factor_regs = final_tile && n_expect == 1;
  if (!factor_regs) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double* dst = &S[(size_t)(ri + 16 * wave + crow + 4 * reg) * ld + rj + 16 * t + ccol];
        if (shared_tile) atomicAdd(dst, acc[t][reg]); else *dst = acc[t][reg];
      }
  }
  if (diag) {
    // this workgroup publishes the L panel of its row tile — into the shadow matrix Lp, NOT in place:
    // the other workgroups of this launch still read A(i, k) from S
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = tid + 256 * q;
      const int r = i >> 5, c2 = (i & 31) * 2;
      *reinterpret_cast<double2*>(&Lp[(size_t)(ri + r) * ld + c0 + c2]) = *reinterpret_cast<const double2*>(&sXi[r * LDT + c2]);
    }
  }
  bool factor_now = factor_regs;
  double* sC = sXj;   // (a diagonal workgroup: sXj is free)
  if (final_tile && n_expect > 1) {
    // arrival: every thread's atomics device-visible, then one count per workgroup; the last one resets the counter for the
    // next factorisation and reads the finished tile past its (possibly stale) L2 lines
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      int* arrive = tile_sync + (ld / NB) + ti;
      const int last = atomicAdd(arrive, 1) == n_expect - 1;
      if (last) atomicExch(arrive, 0);
      s_last = last;
    }
    __syncthreads();
    if (s_last) {
      __threadfence();   // acquire at agent scope: invalidates what this XCD's caches may hold of the tile; the other updaters released
                         // their atomics with the fence before their arrival.  Plain 16-byte loads, all in flight, after it.
      double2 vt[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = tid + 256 * q;
        vt[q] = *reinterpret_cast<const double2*>(&S[(size_t)(ri + (i >> 5)) * ld + ri + (i & 31) * 2]);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = tid + 256 * q;
        const int r = i >> 5, c2 = (i & 31) * 2;
        sC[r * LDT + c2] = (c2 <= r) ? vt[q].x : 0.0;
        sC[r * LDT + c2 + 1] = (c2 + 1 <= r) ? vt[q].y : 0.0;
      }
      factor_now = true;
    }
  } else if (factor_regs) {
    // (the strictly upper part of the tile is never read by the factorisation, so it is left as the update produced it)
#pragma unroll
    for (int t = 0; t < 4; ++t) store_d(sC + (16 * wave) * LDT + 16 * t, LDT, lane, acc[t]);
  }
  if (factor_now) {
    __syncthreads();
    if (nreal[ti] < NB) { mask_unreal_columns(sC, nreal[ti], tid); __syncthreads(); }
    stamp();
    const bool bad = potrf64_lds<PROBE>(sC, sV, sInvD, tid, nreal[ti], PROBE ? probe_ts + 8 : nullptr);
    stamp();
    if (bad && tid == 0) scal[SC_CHOL_FAIL] = 1.0;
    write_factor(S, Lp, ld, ti, sC, sV, sInvD, Vinv, tid);
    if (PROBE) { __syncthreads(); stamp(); }
  }
}

constexpr size_t kPanelStepLds = sizeof(double) * (3 * NB * LDT + 4 * 256 + 4 * 16 * 17 + 64);

void launch_chol_potrf_tiles(hipStream_t s, double* S, double* Lp, int ld, const int* tiles_dev, int n_tiles, const int* nreal_dev,
                             double* Vinv, double* scal) {
  if (n_tiles <= 0) return;
  hipLaunchKernelGGL(chol_potrf_tiles_kernel, dim3(n_tiles), dim3(256), 0, s, S, Lp, ld, tiles_dev, nreal_dev, Vinv, scal);
}
void launch_chol_panel_step(hipStream_t s, double* S, double* Lp, int ld, const PanelDesc* descs_dev, int n_panels, int max_rows,
                            const int* rows_flat_dev, const int* nreal_dev, double* Vinv, double* scal, int* tile_sync_dev,
                            const PanelDesc* descs_host, const int* rows_flat_host) {
  if (n_panels <= 0 || max_rows <= 0) return;
  StepArgs a;
  std::memset(&a, 0, sizeof(a));
  if (descs_host && rows_flat_host && n_panels <= kStepMaxPanels && max_rows <= kStepMaxRows) {
    for (int p = 0; p < n_panels; ++p) {
      a.k[p] = descs_host[p].k; a.n_rows[p] = descs_host[p].n_rows; a.final_mask[p] = descs_host[p].final_mask;
      a.shared_mask[p] = descs_host[p].shared_mask; a.self_potrf[p] = descs_host[p].self_potrf;
      for (int q = 0; q < descs_host[p].n_rows; ++q) a.rows[p][q] = rows_flat_host[descs_host[p].row_off + q];
    }
    hipLaunchKernelGGL(chol_panel_step_kernel<true>, dim3(max_rows, max_rows, n_panels), dim3(256), kPanelStepLds, s, S, Lp, ld, descs_dev,
                       rows_flat_dev, nreal_dev, Vinv, scal, tile_sync_dev, a);
  } else {
    hipLaunchKernelGGL(chol_panel_step_kernel<false>, dim3(max_rows, max_rows, n_panels), dim3(256), kPanelStepLds, s, S, Lp, ld, descs_dev,
                       rows_flat_dev, nreal_dev, Vinv, scal, tile_sync_dev, a);
  }
}

// ---------------------------------------------------------------------------------------------------
// The whole factorisation in ONE launch: a persistent grid pulls the plan's tasks (dense_plan.h FusedTask) from a device-side
// queue in list order and synchronises through per-tile counters instead of kernel boundaries.
//
//   task (k; ti, tj):  wait  L_kk published, tiles (ti,k) and (tj,k) final, and its turn on tile (ti,tj)
//                      X_i = A_ik L_kk^-T, X_j = A_jk L_kk^-T, C_ij -= X_i X_j^T          (as chol_panel_step_kernel)
//                      publish C_ij, advance the tile's counter; whoever applies the LAST update of a diagonal tile holds the
//                      finished tile in registers and factors it on the spot (look-ahead) and publishes L + its block inverses
//
// Updates of one tile are applied in list order (FusedTask::need_c), so there are no atomics on matrix data and the factor is
// bit-reproducible.  Hand-off between workgroups (MI355X: eight XCDs with private L2s, L1 never refreshed by other CUs):
// producers store write-through (sc1) and drain (s_waitcnt vmcnt(0)) before ONE lane advances the counter with an agent-scope
// atomic; consumers poll the counters with relaxed agent-scope loads from one lane and then read the tiles with sc1 loads.
// Dead-lock freedom: the queue hands tasks out in list order and every counter a task waits for is advanced by earlier tasks, so
// the earliest unfinished task is always held by a running workgroup whose dependencies are complete — whatever the residency
// of the grid.  Every wait is bounded (wall clock): on a time-out the factorisation is flagged failed (SC_CHOL_FAIL) and drains.
// The last workgroup to leave re-zeroes the queue head and the counters: the next launch starts from a clean slate.
// ---------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
BSG_DEV double2 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */);
  double2 d; __builtin_memcpy(&d, &v, 16); return d;
}
BSG_DEV double ld8_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16);
  double d; __builtin_memcpy(&d, &v, 8); return d;
}
BSG_DEV void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double2 d) {
  u32x4_t v; __builtin_memcpy(&v, &d, 16);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}
BSG_DEV void st8_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double d) {
  u32x2_t v; __builtin_memcpy(&v, &d, 8);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, 16);
}
BSG_DEV int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: wait until *p == want (counters only ever grow towards it); false on abort / time-out
BSG_DEV bool wait_count(const int* p, int want, int* abort_w, long long deadline) {
  // (both words are requested together and the wall clock is read every 32nd round: a round of the poll is ONE memory round trip,
  // not three in a row — the waiter sees the counter ~0.5 us sooner)
  // (a tile's counter: the updates that took their turn on it in the lower half, the kFusedSplit chunks that reached it in the upper half;
  //  `want` is encoded the same way — a plain count asks for nothing of the upper half)
  const int want_lo = want & 0xffff, want_hi = (int)((unsigned)want >> 16);
  for (unsigned it = 0;; ++it) {
    // (the abort word is ONE line that every waiting workgroup of the launch would read each round — hundreds of readers on one channel —
    //  so it is looked at every eighth round, together with the counter: still one round trip)
    const bool look = (it & 7) == 7;
    const int v = ld_flag(p), a = look ? ld_flag(abort_w) : 0;
    if ((v & 0xffff) >= want_lo && (int)((unsigned)v >> 16) >= want_hi) return true;
    if (a != 0) return false;
    if ((it & 31) == 31 && (long long)wall_clock64() > deadline) { __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
}
// one WAVE: wait until every flag in [lo, hi) is set (each written once, by a different workgroup: a counter word that 18-25 workgroups
// on eight XCDs add to costs its waiter ~2.7 us after the last add, a flag per writer ~1); false on abort / time-out (wave-uniform)
BSG_DEV bool wait_flags(const int* flags, int lo, int hi, int* abort_w, long long deadline, int stride = 1) {
  const int lane = threadIdx.x & 63;
  unsigned rounds = 0;
  for (;;) {
    int ok = 1;
    for (int i = lo + lane; i < hi; i += 64) ok &= (ld_flag(flags + (size_t)i * stride) != 0) ? 1 : 0;
    if (__all(ok)) return true;
    int stop = 0;
    if (lane == 0 && (++rounds & 7) == 0) {
      if (ld_flag(abort_w) != 0) stop = 1;
      else if ((long long)wall_clock64() > deadline) { __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); stop = 1; }
    }
    if (__builtin_amdgcn_readfirstlane(stop)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}
constexpr long long kFusedTimeoutTicks = 100000000LL / 4;   // s_memrealtime runs at 100 MHz: a quarter of a second

// a value every lane holds identically, moved to scalar registers (the compiler cannot see that it is uniform once it has been
// through memory or a function argument: buffer descriptors built from it would otherwise be applied lane by lane)
template <typename P> BSG_DEV P* uniform_ptr(P* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}
BSG_DEV long long uniform_i64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
constexpr int kExtPitch = 18;   // doubles between two rows of a 16-column appendix block in LDS
// the appendix columns of one 16-row strip (kFusedExt): sE strip (16 x 16) <- (sE strip - X strip L16^T) W16^T, X = the strip's solved
// 64 columns (sX, pitch LDT), L16 = rows 0..15 of L(k + 1, k) (pitch LDT), W16 = the inverse of the appendix's 16 x 16 factor
BSG_DEV void solve_ext_strip(const double* sX, double* sE, const double* sL16, const double* sW16, int lane, int strip) {
  double* rows = sE + (16 * strip) * kExtPitch;
  double4_t t = load_d(rows, kExtPitch, lane);
  t = mfma_abt<64>(t, sX + (16 * strip) * LDT, LDT, sL16, LDT, -1.0, lane);
  __builtin_amdgcn_wave_barrier();
  store_d(rows, kExtPitch, lane, t);
  __builtin_amdgcn_wave_barrier();
  double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
  x = mfma_abt<16>(x, rows, kExtPitch, sW16, kExtPitch, 1.0, lane);
  __builtin_amdgcn_wave_barrier();
  store_d(rows, kExtPitch, lane, x);
}
BSG_DEV bool diag_tile_of(const FusedTask& tk) { return tk.ti == tk.tj; }
struct FusedCtx {
  double *S, *Lp, *Vinv, *scal;
  double* Winv;   // per tile: the full inverse of its factor: its diagonal 16x16 blocks are what the tasks' triangular solves multiply by,
                  // the whole of it what the back-substitution multiplies by
  const FusedTask* tasks;
  int n_tasks;
  const int* tile_tot;
  const int* nreal;
  int ld, n_vinv_tiles;
  int rhs_strips;   // 16-row strips of the rhs tile that hold anything (1: the solve's single rhs row; 4: every row may be used)
  int *abort_w, *potrf_done, *upd;
  int fs;   // ints between two words of the sync area (a cache line apart: words of one line that different workgroups write or add to are
            // serialised at the memory side — measured on the PCG slots, k_pcg.hip — and the queue head / exit counter take ~2 000 atomics)
  long long deadline;
  long long* probe_ts;
  int no_turn;      // updates add their products with FP64 atomics as soon as they have them, in no particular order (1: a diagonal tile's LM-diagonal task is in
                    // the list and goes first, 2: no such task); 0: every update waits for its turn on the tile and rewrites it (bit-reproducible factor)
};
// One UPDATE task of the fused factorisation (dense_plan.h FusedTask); returns false when a wait was aborted (wave-uniform).
// 512 threads: waves 0-3 solve the strips of X_i, waves 4-7 those of X_j at the same time; the rank-64 update is two 16x16 blocks
// per wave.  Stamps (PROBE): 1 got task, 2 dependencies met, 3 tiles in LDS, 4 solves done, 5 product done and turn taken, 6 published.
template <bool PROBE, int NT>
BSG_DEV bool chol_fused_update(const FusedCtx& C, int t, const FusedTask& tk, double* smem) {
  constexpr int NQ = 2048 / NT;          // 16-byte pieces of a 64x64 tile per thread
  constexpr int TPW = 16 / (NT / 64);    // 16x16 blocks of the update per wave
  double* const S = uniform_ptr(C.S); double* const Lp = uniform_ptr(C.Lp);
  const int ld = __builtin_amdgcn_readfirstlane(C.ld);
  int* const abort_w = uniform_ptr(C.abort_w); int* const potrf_done = uniform_ptr(C.potrf_done); int* const upd = uniform_ptr(C.upd);
  const int fs = __builtin_amdgcn_readfirstlane(C.fs);
  const long long deadline = uniform_i64(C.deadline);
  long long* const probe_ts = uniform_ptr(C.probe_ts);
  double* sXi = smem;                 // 64 x LDT
  double* sXj = sXi + NB * LDT;       // 64 x LDT
  double* sL = sXj + NB * LDT;        // 64 x LDT
  // an APPENDIX tile's columns (kFusedExt, dense_plan.h): 16 columns of the two row tiles, the 16 x 64 block L(k + 1, k), the 16 x 16 inverse W_{k+1}
  double* sEi = sL + NB * LDT;       // 64 x kExtPitch
  double* sEj = sEi + NB * kExtPitch;
  double* sL16 = sEj + NB * kExtPitch;   // 16 x LDT
  double* sW16 = sL16 + 16 * LDT;        // 16 x kExtPitch
  int* s_ctl = reinterpret_cast<int*>(sW16 + 16 * kExtPitch);   // 4 ints
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = ld / NB;
  // (buffer resources per TILE ROW — 64 rows of S or of the factor: a resource counts its bytes and its offsets in 32 bits, and the
  // matrix passes 4 GB at 23 170 dimensions; with one resource over the whole matrix every step above that size came out invalid)
  auto tile_rows = [&](double* base, int row0) {
    return __builtin_amdgcn_make_buffer_rsrc(base + (size_t)row0 * ld, 0, (int)((size_t)NB * ld * sizeof(double)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(C.Winv), 0, (int)((size_t)(N - 1) * 4096 * sizeof(double)), 0x00020000);
  const int crow = lane >> 4, ccol = lane & 15;
  const int rs = wave & 3, tt0 = (wave >> 2) * TPW;   // this wave's row strip and first column block of the update
  auto stamp = [&](int slot) { if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + slot] = wall_clock64(); };
  stamp(1);
  const int k = tk.k, ti = tk.ti, tj = tk.tj;
  const bool diag = ti == tj;
  const bool do_update = tk.need_c >= 0;
  // NO TURNS (round 6): the turn on a tile was a read-modify-write behind every earlier updater of the tile — 2-4 us each, one after the other, and where the
  // last panels of two children meet in a separator's tile (the critical path) the chunks waited for two or three of them (BSGPU_CHOL_PROBE: 8 and 12 us on C2's
  // first two levels).  Every update now ADDS (agent-scope FP64 atomics, as the K-chunks and the assembly do) the moment its product exists; only the
  // LM-diagonal task of a diagonal tile, which rewrites the tile, still goes first.  The tile's counter counts the updates as before.
  const int no_turn = __builtin_amdgcn_readfirstlane(C.no_turn);
  const int need_c = !do_update ? -1 : no_turn == 0 ? tk.need_c : (no_turn == 1 && diag_tile_of(tk) ? (tk.need_c < 1 ? tk.need_c : 1) : 0);
  const bool solve_i = !(tk.flags & kFusedXiLp);
  const bool solve_j = !diag && !(tk.flags & (kFusedXjLp | kFusedXjChain));
  const bool need_L = solve_i || solve_j;
  const bool ext = (tk.flags & kFusedExt) != 0;   // (NT == 512 only: chol_fused_kernel)
  const int te = k + 1;
  const int ri = __builtin_amdgcn_readfirstlane(ti * NB), rj = __builtin_amdgcn_readfirstlane(tj * NB), c0 = __builtin_amdgcn_readfirstlane(k * NB);
  const __amdgpu_buffer_rsrc_t rS_i = tile_rows(S, ri), rS_j = tile_rows(S, rj), rL_i = tile_rows(Lp, ri), rL_j = tile_rows(Lp, rj);
  // rows of the rhs tile (tile N-1 as row tile) beyond its used strips are zero and stay zero: no solve, no product, no traffic for them
  const int strips_i = (ti == N - 1) ? __builtin_amdgcn_readfirstlane(C.rhs_strips) : 4;
  const bool strip_on = rs < strips_i;
  double2 vXi[NQ], vL[NQ], vXj[NQ];
  // The panel tiles first: they have usually had their last update long before L_kk is out, so their loads travel while the
  // workgroup waits for the factor; only L_kk and its block inverses are requested after it.  A strip that is not solved here is
  // read from the factor: X of an outside tile once its diagonal task has published it (the tile's counter one past its last
  // update), X of a tile of k's own chain with potrf_done[k].
  if (tid == 0) {
    bool ok = wait_count(&upd[(ti * N + k) * fs], tk.tot_i + (solve_i ? 0 : 1), abort_w, deadline);
    if (!diag && !(tk.flags & kFusedXjChain)) ok = ok && wait_count(&upd[(tj * N + k) * fs], tk.tot_j + (solve_j ? 0 : 1), abort_w, deadline);
    if (ext) {   // (the appendix columns of a tile this task solves must be final too; a published X carries them)
      const int* tot = uniform_ptr(C.tile_tot);
      if (solve_i) ok = ok && wait_count(&upd[(ti * N + te) * fs], tot[(size_t)ti * N + te], abort_w, deadline);
      if (solve_j) ok = ok && wait_count(&upd[(tj * N + te) * fs], tot[(size_t)tj * N + te], abort_w, deadline);
    }
    s_ctl[1] = ok ? 1 : 0;
    // ... and the C tile, if it is already this task's turn on it (on the critical path it is: the tile's earlier updaters are
    // panels that finished long ago): its values wait in registers through the solves, and the product accumulates onto them
    s_ctl[3] = (do_update && !no_turn && (tk.need_c == 0 || (ld_flag(&upd[(ti * N + tj) * fs]) & 0xffff) >= tk.need_c)) ? 1 : 0;
  }
  __syncthreads();
  const bool c_pre = __builtin_amdgcn_readfirstlane(s_ctl[3]) != 0;
  double4_t cpre[TPW];
  if (c_pre && strip_on) {
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        cpre[u][reg] = ld8_sc1(rS_i, (unsigned)(((size_t)(16 * rs + crow + 4 * reg) * ld + rj + 16 * (tt0 + u) + ccol) * sizeof(double)));
  }
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + NT * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    vXi[q] = ld16_sc1(solve_i ? rS_i : rL_i, (unsigned)(((size_t)r * ld + c0 + c2) * sizeof(double)));
    if (!diag && !(tk.flags & kFusedXjChain)) vXj[q] = ld16_sc1(solve_j ? rS_j : rL_j, (unsigned)(((size_t)r * ld + c0 + c2) * sizeof(double)));
  }
  double2 vEi = double2{0.0, 0.0}, vEj = double2{0.0, 0.0}, vL16 = double2{0.0, 0.0}, vW16 = double2{0.0, 0.0};
  const int er = tid >> 3, ec2 = (tid & 7) * 2;   // this thread's piece of a 64 x 16 appendix block
  if (ext) {
    vEi = ld16_sc1(solve_i ? rS_i : rL_i, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)));
    if (!diag) vEj = ld16_sc1(solve_j ? rS_j : rL_j, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)));
  }
  __syncthreads();   // (s_ctl[1] is rewritten below)
  if (need_L || (tk.flags & kFusedXjChain)) {
    if (tid == 0) s_ctl[1] = wait_count(&potrf_done[(ext ? te : k) * fs], 1, abort_w, deadline) ? 1 : 0;   // (a chain sets its tiles' flags in order)
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  }
  stamp(2);
  if (tk.flags & kFusedXjChain) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      vXj[q] = ld16_sc1(rL_j, (unsigned)(((size_t)(i >> 5) * ld + c0 + (i & 31) * 2) * sizeof(double)));
    }
  }
  if (need_L) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      vL[q] = ld16_sc1(rW, (unsigned)(((size_t)k * 4096 + (size_t)(i >> 5) * 64 + (i & 31) * 2) * sizeof(double)));   // W = L_kk^-1, the whole tile (solve_tile_w)
    }
    if (ext) {   // rows 0..15 of L(k + 1, k) and the 16 x 16 inverse of the appendix's factor
      const __amdgpu_buffer_rsrc_t rL_e = tile_rows(Lp, te * NB);
      vL16 = ld16_sc1(rL_e, (unsigned)(((size_t)(tid >> 5) * ld + c0 + (tid & 31) * 2) * sizeof(double)));
      if (tid < 128) vW16 = ld16_sc1(rW, (unsigned)(((size_t)te * 4096 + (size_t)er * 64 + ec2) * sizeof(double)));
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + NT * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    *reinterpret_cast<double2*>(&sXi[r * LDT + c2]) = vXi[q];
    if (need_L) *reinterpret_cast<double2*>(&sL[r * LDT + c2]) = vL[q];
    if (!diag) *reinterpret_cast<double2*>(&sXj[r * LDT + c2]) = vXj[q];
  }
  if (ext) {
    *reinterpret_cast<double2*>(&sEi[er * kExtPitch + ec2]) = vEi;
    if (!diag) *reinterpret_cast<double2*>(&sEj[er * kExtPitch + ec2]) = vEj;
    if (need_L) {
      *reinterpret_cast<double2*>(&sL16[(tid >> 5) * LDT + (tid & 31) * 2]) = vL16;
      if (tid < 128) *reinterpret_cast<double2*>(&sW16[er * kExtPitch + ec2]) = vW16;
    }
  }
  __syncthreads();
  stamp(3);
  if (need_L) {
    if (NT == 256) {
      if (solve_i && wave < strips_i) solve_tile_w(sXi, sL, lane, wave);
      if (solve_j) solve_tile_w(sXj, sL, lane, wave);
    } else {   // (waves 0-3: the strips of X_i; waves 4-7: those of X_j, at the same time)
      if (wave < 4) { if (solve_i && wave < strips_i) solve_tile_w(sXi, sL, lane, wave); }
      else if (solve_j) solve_tile_w(sXj, sL, lane, wave - 4);
    }
    __syncthreads();
    if (ext) {   // X(., k + 1) = (A(., k + 1) - X(., k) L(k + 1, k)^T) W_{k+1}^T, strip by strip
      if (wave < 4) { if (solve_i && wave < strips_i) solve_ext_strip(sXi, sEi, sL16, sW16, lane, wave); }
      else if (solve_j) solve_ext_strip(sXj, sEj, sL16, sW16, lane, wave - 4);
      __syncthreads();
    }
  }
  stamp(4);
  if (tk.flags & kFusedPublishX) {
    // the L panel of this row tile — what the back-substitution reads, and what the off-diagonal tasks of this panel multiply with —
    // leaves NOW, before this task's own product and its turn on the diagonal tile: the tasks that read it were waiting through both
    // (7-10 us after the chain's flag on every level of the critical path, BSGPU_CHOL_PROBE; round 4)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      const int r = i >> 5, c2 = (i & 31) * 2;
      st16_sc1(rL_i, (unsigned)(((size_t)r * ld + c0 + c2) * sizeof(double)), *reinterpret_cast<const double2*>(&sXi[r * LDT + c2]));
    }
    // (... with the appendix's 16 columns: the rest of that tile of the factor is the padding's zeros, which nothing ever writes)
    if (ext) st16_sc1(rL_i, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)), *reinterpret_cast<const double2*>(&sEi[er * kExtPitch + ec2]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) atomicAdd(&upd[(ti * N + k) * fs], 1);
  }
  const double* Xj = diag ? sXi : sXj;
  double4_t acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) acc[u] = double4_t{0.0, 0.0, 0.0, 0.0};
  // The C tile: already in registers (requested before the wait for L_kk), or — if the turn had not come by then — one more look
  // now: a tile requested here still travels under the 256 MFMAs instead of being waited for after them.
  bool c_early = false;
  if (do_update) {
    if (c_pre) {
      c_early = true;
      if (strip_on) {
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[u] = cpre[u];
      }
    } else if (!no_turn) {
      if (tid == 0) s_ctl[3] = ((ld_flag(&upd[(ti * N + tj) * fs]) & 0xffff) >= tk.need_c) ? 1 : 0;
      __syncthreads();
      c_early = __builtin_amdgcn_readfirstlane(s_ctl[3]) != 0;
      if (c_early && strip_on) {
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            acc[u][reg] = ld8_sc1(rS_i, (unsigned)(((size_t)(16 * rs + crow + 4 * reg) * ld + rj + 16 * (tt0 + u) + ccol) * sizeof(double)));
      }
    }
    if (strip_on) {
#pragma unroll
      for (int u = 0; u < TPW; ++u)
        acc[u] = mfma_abt<64>(acc[u], sXi + (16 * rs) * LDT, LDT, Xj + (16 * (tt0 + u)) * LDT, LDT, -1.0, lane);
      if (ext) {
        const double* Ej = diag ? sEi : sEj;
#pragma unroll
        for (int u = 0; u < TPW; ++u)
          acc[u] = mfma_abt<16>(acc[u], sEi + (16 * rs) * kExtPitch, kExtPitch, Ej + (16 * (tt0 + u)) * kExtPitch, kExtPitch, -1.0, lane);
      }
    }
    if (no_turn) {
      if (need_c > 0) {   // (a diagonal tile: its LM-diagonal task rewrites it — long done by the time any product exists)
        if (tid == 0) s_ctl[1] = wait_count(&upd[(ti * N + tj) * fs], need_c, abort_w, deadline) ? 1 : 0;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
      }
      stamp(5);
      if (strip_on) {
        double* const Sg = S + (size_t)(ri + 16 * rs + crow) * ld + rj + 16 * tt0 + ccol;
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            (void)__hip_atomic_fetch_add(Sg + (size_t)(4 * reg) * ld + 16 * u, acc[u][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
    if (!c_early) {
      // this task's turn on the tile: every earlier update of it has been published
      if (tid == 0) s_ctl[1] = wait_count(&upd[(ti * N + tj) * fs], tk.need_c, abort_w, deadline) ? 1 : 0;
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
      if (strip_on) {
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            acc[u][reg] += ld8_sc1(rS_i, (unsigned)(((size_t)(16 * rs + crow + 4 * reg) * ld + rj + 16 * (tt0 + u) + ccol) * sizeof(double)));
      }
    }
    stamp(5);
    if (strip_on) {
#pragma unroll
      for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          st8_sc1(rS_i, (unsigned)(((size_t)(16 * rs + crow + 4 * reg) * ld + rj + 16 * (tt0 + u) + ccol) * sizeof(double)), acc[u][reg]);
    }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && do_update) atomicAdd(&upd[(ti * N + tj) * fs], 1);
  stamp(6);
  return true;
}

// runtime-K form of mfma_abt (K a multiple of 4), FOUR accumulators: consecutive MFMAs of one accumulator are ~175 cycles apart (the result of
// one is the addend of the next), of different ones 64 — a chunk's products are chains of up to 16
BSG_DEV double4_t mfma_abt_rt(double4_t acc, const double* sA, int lda, const double* sB, int ldb, double sign, int K, int lane) {
  const int r = lane & 15, kq = lane >> 4;
  double4_t a1 = double4_t{0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1;
  int k = 0;
  for (; k + 16 <= K; k += 16) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[r * lda + k + kq], sB[r * ldb + k + kq], acc, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[r * lda + k + 4 + kq], sB[r * ldb + k + 4 + kq], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[r * lda + k + 8 + kq], sB[r * ldb + k + 8 + kq], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[r * lda + k + 12 + kq], sB[r * ldb + k + 12 + kq], a3, 0, 0, 0);
  }
  for (; k < K; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[r * lda + k + kq], sB[r * ldb + k + kq], acc, 0, 0, 0);
  return (acc + a1) + (a2 + a3);
}
// D += A B over k in [k0, k1) (multiples of 16 apart): A = 16 rows at sA (pitch lda), B = rows k of sB (pitch ldb), 16 columns from column c0
// (B is NOT transposed); four accumulators as above
BSG_DEV double4_t mfma_ab_rt(double4_t acc, const double* sA, int lda, const double* sB, int ldb, int c0, int k0, int k1, int lane) {
  const int r = lane & 15, kq = lane >> 4;
  double4_t a1 = double4_t{0.0, 0.0, 0.0, 0.0}, a2 = a1, a3 = a1;
  for (int k = k0; k < k1; k += 16) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r * lda + k + kq], sB[(k + kq) * ldb + c0 + r], acc, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r * lda + k + 4 + kq], sB[(k + 4 + kq) * ldb + c0 + r], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r * lda + k + 8 + kq], sB[(k + 8 + kq) * ldb + c0 + r], a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r * lda + k + 12 + kq], sB[(k + 12 + kq) * ldb + c0 + r], a3, 0, 0, 0);
  }
  return (acc + a1) + (a2 + a3);
}
// ONE K-CHUNK of the last update of a tile inside a chain (dense_plan.h kFusedSplit): chunk p of P forms the 16 columns 16 p .. of
// X_ti = A(ti, k) W^T (W = L_kk^-1 is lower triangular: K = 16 (p + 1)) and of X_tj and adds -X_ti X_tj^T (rank 16) to the tile with FP64
// atomics, once the tile's earlier updates are all in (tk.need_c: the chunks share the last turn).  The appendix chunk (p = P - 1 of a panel
// that carries an appendix) forms the appendix's columns X_e = E W16^T - A M^T with M = W16 L(k+1, k) W — the lower-left block of the
// inverse of the 80-column factor, 16 x 64, formed here (20 MFMAs on the path instead of the whole strips' solves).  A diagonal task's
// chunks publish their columns of X to the factor; the last one to have done so bumps the panel tile's counter (what the readers of X wait for).
template <bool PROBE>
BSG_DEV bool chol_fused_split(const FusedCtx& C, int t, const FusedTask& tk, double* smem) {
  constexpr int NT = 512, NQ = 2048 / NT;
  double* const S = uniform_ptr(C.S); double* const Lp = uniform_ptr(C.Lp);
  const int ld = __builtin_amdgcn_readfirstlane(C.ld);
  int* const abort_w = uniform_ptr(C.abort_w); int* const potrf_done = uniform_ptr(C.potrf_done); int* const upd = uniform_ptr(C.upd);
  const int fs = __builtin_amdgcn_readfirstlane(C.fs);
  const long long deadline = uniform_i64(C.deadline);
  long long* const probe_ts = uniform_ptr(C.probe_ts);
  double* sXi = smem;                    // A(ti, k), 64 x LDT
  double* sXj = sXi + NB * LDT;          // A(tj, k)
  double* sL = sXj + NB * LDT;           // rows 16 p .. of W (a chunk) | W (the appendix chunk)
  double* sEi = sL + NB * LDT;           // 64 x kExtPitch: the chunk's columns of X_ti (the appendix chunk: E_ti in, X_e out)
  double* sEj = sEi + NB * kExtPitch;
  double* sL16 = sEj + NB * kExtPitch;   // 16 x LDT: rows 0..15 of L(k + 1, k); then M
  double* sW16 = sL16 + 16 * LDT;        // 16 x kExtPitch
  int* s_ctl = reinterpret_cast<int*>(sW16 + 16 * kExtPitch);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = ld / NB;
  auto tile_rows = [&](double* base, int row0) {
    return __builtin_amdgcn_make_buffer_rsrc(base + (size_t)row0 * ld, 0, (int)((size_t)NB * ld * sizeof(double)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(C.Winv), 0, (int)((size_t)(N - 1) * 4096 * sizeof(double)), 0x00020000);
  const int crow = lane >> 4, ccol = lane & 15;
  auto stamp = [&](int sl) { if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + sl] = wall_clock64(); };
  stamp(1);
  const int k = tk.k, ti = tk.ti, tj = tk.tj;
  const int p = tk.tot_c & 0xff, P = tk.tot_c >> 8;
  const bool diag = ti == tj;
  const bool ext = (tk.flags & kFusedExt) != 0, ext_chunk = ext && p == P - 1;
  const int te = k + 1;
  const int ri = __builtin_amdgcn_readfirstlane(ti * NB), rj = __builtin_amdgcn_readfirstlane(tj * NB), c0 = __builtin_amdgcn_readfirstlane(k * NB);
  const __amdgpu_buffer_rsrc_t rS_i = tile_rows(S, ri), rS_j = tile_rows(S, rj), rL_i = tile_rows(Lp, ri);
  if (tid == 0) {
    bool ok = wait_count(&upd[(ti * N + k) * fs], tk.tot_i, abort_w, deadline);
    if (!diag) ok = ok && wait_count(&upd[(tj * N + k) * fs], tk.tot_j, abort_w, deadline);
    if (ext_chunk) {
      const int* tot = uniform_ptr(C.tile_tot);
      ok = ok && wait_count(&upd[(ti * N + te) * fs], tot[(size_t)ti * N + te], abort_w, deadline);
      if (!diag) ok = ok && wait_count(&upd[(tj * N + te) * fs], tot[(size_t)tj * N + te], abort_w, deadline);
    }
    s_ctl[1] = ok ? 1 : 0;
  }
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  double2 vXi[NQ], vXj[NQ], vL[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + NT * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    vXi[q] = ld16_sc1(rS_i, (unsigned)(((size_t)r * ld + c0 + c2) * sizeof(double)));
    if (!diag) vXj[q] = ld16_sc1(rS_j, (unsigned)(((size_t)r * ld + c0 + c2) * sizeof(double)));
  }
  double2 vEi = double2{0.0, 0.0}, vEj = double2{0.0, 0.0}, vL16 = double2{0.0, 0.0}, vW16 = double2{0.0, 0.0}, vWc = double2{0.0, 0.0};
  const int er = tid >> 3, ec2 = (tid & 7) * 2;   // this thread's piece of a 64 x 16 block
  if (ext_chunk) {
    vEi = ld16_sc1(rS_i, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)));
    if (!diag) vEj = ld16_sc1(rS_j, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)));
  }
  __syncthreads();   // (s_ctl[1] is rewritten below)
  // (a chain sets its tiles' flags in order.  The appendix chunk too starts with tile k's flag: W and L(k + 1, k) are out with it, and all of
  //  its work but one 16-wide product — see below — needs nothing of the appendix's own factor, which is out a step of the chain later)
  if (tid == 0) s_ctl[1] = wait_count(&potrf_done[k * fs], 1, abort_w, deadline) ? 1 : 0;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  if (!ext_chunk) stamp(2);
  if (!ext_chunk) {
    // rows 16 p .. 16 p + 15 of W: one 16-byte piece per thread
    vWc = ld16_sc1(rW, (unsigned)(((size_t)k * 4096 + (size_t)(16 * p + (tid >> 5)) * 64 + (tid & 31) * 2) * sizeof(double)));
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      vL[q] = ld16_sc1(rW, (unsigned)(((size_t)k * 4096 + (size_t)(i >> 5) * 64 + (i & 31) * 2) * sizeof(double)));
    }
    const __amdgpu_buffer_rsrc_t rL_e = tile_rows(Lp, te * NB);
    vL16 = ld16_sc1(rL_e, (unsigned)(((size_t)(tid >> 5) * ld + c0 + (tid & 31) * 2) * sizeof(double)));
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = tid + NT * q;
    const int r = i >> 5, c2 = (i & 31) * 2;
    *reinterpret_cast<double2*>(&sXi[r * LDT + c2]) = vXi[q];
    if (!diag) *reinterpret_cast<double2*>(&sXj[r * LDT + c2]) = vXj[q];
    if (ext_chunk) *reinterpret_cast<double2*>(&sL[r * LDT + c2]) = vL[q];
  }
  if (!ext_chunk) *reinterpret_cast<double2*>(&sL[(tid >> 5) * LDT + (tid & 31) * 2]) = vWc;
  else {
    *reinterpret_cast<double2*>(&sEi[er * kExtPitch + ec2]) = vEi;
    if (!diag) *reinterpret_cast<double2*>(&sEj[er * kExtPitch + ec2]) = vEj;
    *reinterpret_cast<double2*>(&sL16[(tid >> 5) * LDT + (tid & 31) * 2]) = vL16;
  }
  __syncthreads();
  if (!ext_chunk) stamp(3);
  if (!ext_chunk) {
    // (waves 0-3: the strips of the chunk of X_i; waves 4-7: those of X_j, at the same time)
    if (wave < 4 || !diag) {
      const double* A = wave < 4 ? sXi : sXj;
      double* E = wave < 4 ? sEi : sEj;
      const int strip = wave & 3;
      double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
      x = mfma_abt_rt(x, A + (16 * strip) * LDT, LDT, sL, LDT, 1.0, 16 * (p + 1), lane);
      store_d(E + (16 * strip) * kExtPitch, kExtPitch, lane, x);
    }
    __syncthreads();
  } else {
    // X_e = E W16^T - A M^T with M = W16 L16 W  ==  (E - A P^T) W16^T with P = L16 W (16 x 64).  P and Y = E - A P^T need the FIRST tile of the
    // chain only (W = L_kk^-1 and L16 = rows 0..15 of L(k + 1, k) leave the chain with tile k's flag): they are formed while the chain runs its
    // last 16-pivot step, and what follows the chain's end is one load of W16 (2 KB) and one 16-wide product per strip — the appendix chunk was
    // the last of its panel's chunks by 4 us on every level of the critical path (BSGPU_CHOL_PROBE, round 6: its two dependent 16 x 64 products
    // and the strips behind them all came after the flag of the appendix).
    // P(:, 16 c ..) = sum_{m >= 16 c} L16(:, m) W(m, 16 c ..)   (W lower triangular), block column c by wave c
    double4_t pb = double4_t{0.0, 0.0, 0.0, 0.0};
    if (wave < 4) pb = mfma_ab_rt(pb, sL16, LDT, sL, LDT, 16 * wave, 16 * wave, 64, lane);
    __syncthreads();
    if (wave < 4) store_d(sL16 + 16 * wave, LDT, lane, pb);   // P over L16 (16 x 64, pitch LDT)
    __syncthreads();
    if (wave < 4 || !diag) {   // Y strips = E - A P^T, in place
      const double* A = wave < 4 ? sXi : sXj;
      double* E = wave < 4 ? sEi : sEj;
      const int strip = wave & 3;
      double4_t y = load_d(E + (16 * strip) * kExtPitch, kExtPitch, lane);
      y = mfma_abt_rt(y, A + (16 * strip) * LDT, LDT, sL16, LDT, -1.0, 64, lane);
      __builtin_amdgcn_wave_barrier();
      store_d(E + (16 * strip) * kExtPitch, kExtPitch, lane, y);
    }
    // ... and now the appendix's own factor: its 16 x 16 inverse
    if (tid == 0) s_ctl[1] = wait_count(&potrf_done[te * fs], 1, abort_w, deadline) ? 1 : 0;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
    stamp(2);
    if (tid < 128) vW16 = ld16_sc1(rW, (unsigned)(((size_t)te * 4096 + (size_t)er * 64 + ec2) * sizeof(double)));
    if (tid < 128) *reinterpret_cast<double2*>(&sW16[er * kExtPitch + ec2]) = vW16;
    __syncthreads();
    stamp(3);
    if (wave < 4 || !diag) {   // X_e strips = Y W16^T
      double* E = (wave < 4 ? sEi : sEj) + (16 * (wave & 3)) * kExtPitch;
      double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
      x = mfma_abt_rt(x, E, kExtPitch, sW16, kExtPitch, 1.0, 16, lane);
      __builtin_amdgcn_wave_barrier();
      store_d(E, kExtPitch, lane, x);
    }
    __syncthreads();
  }
  stamp(4);
  if (tk.flags & kFusedPublishX) {
    // this chunk's 16 columns of the L panel of row tile ti (the appendix chunk: the appendix's columns)
    st16_sc1(rL_i, (unsigned)(((size_t)er * ld + c0 + (ext_chunk ? NB : 16 * p) + ec2) * sizeof(double)), *reinterpret_cast<const double2*>(&sEi[er * kExtPitch + ec2]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      // (the chunks count themselves in the upper half of the panel tile's counter; the last one advances the lower half: X is published)
      const int old = atomicAdd(&upd[(ti * N + k) * fs], 1 << 16);
      if ((int)((unsigned)old >> 16) == P - 1) atomicAdd(&upd[(ti * N + k) * fs], 1);
    }
  }
  const double* Ej = diag ? sEi : sEj;
  const int rs = wave & 3, tt0 = (wave >> 2) * 2;
  double4_t acc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    acc[u] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (!(diag && tt0 + u > rs))   // (blocks above the diagonal of a diagonal tile: the chain never reads them)
      acc[u] = mfma_abt_rt(acc[u], sEi + (16 * rs) * kExtPitch, kExtPitch, Ej + (16 * (tt0 + u)) * kExtPitch, kExtPitch, -1.0, 16, lane);
  }
  // the chunks' turn: every earlier update of the tile has been published (on the critical path: long ago) — with no turns (chol_fused_update) only
  // a diagonal tile's LM-diagonal task, which rewrites the tile
  const int no_turn = __builtin_amdgcn_readfirstlane(C.no_turn);
  const int need_c = no_turn == 0 ? tk.need_c : (no_turn == 1 && diag ? (tk.need_c < 1 ? tk.need_c : 1) : 0);
  if (tid == 0) s_ctl[1] = wait_count(&upd[(ti * N + tj) * fs], need_c, abort_w, deadline) ? 1 : 0;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  stamp(5);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (diag && tt0 + u > rs) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      (void)__hip_atomic_fetch_add(&S[(size_t)(ri + 16 * rs + crow + 4 * reg) * ld + rj + 16 * (tt0 + u) + ccol], acc[u][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) atomicAdd(&upd[(ti * N + tj) * fs], 1 << 16);
  stamp(6);
  return true;
}

// A ROW SEGMENT (dense_plan.h kFusedRowSeg; launches without turns only): the updates (k; ti, tj_0), (k; ti, tj_1), ... of ONE row tile by one panel, each of
// which reads the X its two diagonal tasks have published — X_ti is fetched once and stays in LDS, the X_tj follow one after the other (the next one's
// loads are issued before this one's product), every product is added to its tile with FP64 atomics and counted there exactly as a task of its own would.
// What it saves is what an update task spends around its 1.3 - 2 us of products: the workgroup's start, its ticket and record (2.1 us), X_ti (0.9), the
// publication's barrier — many factorisations side by side (bsgpu_solve_batch) are bound by the number of these workgroups, each of which holds a CU.
// items: (tj, updates of tile (tj, k)) pairs behind the task list (tk.tj = index of the segment's first pair, tk.tot_j = their number).
template <bool PROBE>
BSG_DEV bool chol_fused_rowseg(const FusedCtx& C, int t, const FusedTask& tk, double* smem) {
  constexpr int NT = 512, NQ = 2048 / NT, TPW = 2;
  double* const S = uniform_ptr(C.S); double* const Lp = uniform_ptr(C.Lp);
  const int ld = __builtin_amdgcn_readfirstlane(C.ld);
  int* const abort_w = uniform_ptr(C.abort_w); int* const upd = uniform_ptr(C.upd);
  const int fs = __builtin_amdgcn_readfirstlane(C.fs);
  const long long deadline = uniform_i64(C.deadline);
  long long* const probe_ts = uniform_ptr(C.probe_ts);
  const int* const items = reinterpret_cast<const int*>(uniform_ptr(C.tasks) + __builtin_amdgcn_readfirstlane(C.n_tasks));
  double* sXi = smem;                 // 64 x LDT
  double* sXj = sXi + NB * LDT;       // 64 x LDT
  double* sEi = sXj + 2 * NB * LDT;   // 64 x kExtPitch (the layout of chol_fused_update: the W tile's area stays unused)
  double* sEj = sEi + NB * kExtPitch;
  int* s_ctl = reinterpret_cast<int*>(sEj + NB * kExtPitch + 16 * LDT + 16 * kExtPitch);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = ld / NB;
  auto tile_rows = [&](double* base, int row0) {
    return __builtin_amdgcn_make_buffer_rsrc(base + (size_t)row0 * ld, 0, (int)((size_t)NB * ld * sizeof(double)), 0x00020000);
  };
  const int crow = lane >> 4, ccol = lane & 15;
  const int rs = wave & 3, tt0 = (wave >> 2) * TPW;
  auto stamp = [&](int slot) { if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + slot] = wall_clock64(); };
  stamp(1);
  const int k = tk.k, ti = tk.ti, first = tk.tj, len = tk.tot_j;
  const bool ext = (tk.flags & kFusedExt) != 0;
  const int ri = __builtin_amdgcn_readfirstlane(ti * NB), c0 = __builtin_amdgcn_readfirstlane(k * NB);
  const __amdgpu_buffer_rsrc_t rL_i = tile_rows(Lp, ri);
  const int strips_i = (ti == N - 1) ? __builtin_amdgcn_readfirstlane(C.rhs_strips) : 4;
  const bool strip_on = rs < strips_i;
  const int er = tid >> 3, ec2 = (tid & 7) * 2;
  if (tid == 0) s_ctl[1] = wait_count(&upd[(ti * N + k) * fs], tk.tot_i + 1, abort_w, deadline) ? 1 : 0;   // X_ti is published
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  stamp(2);
  {
    double2 vXi[NQ], vEi = double2{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      vXi[q] = ld16_sc1(rL_i, (unsigned)(((size_t)(i >> 5) * ld + c0 + (i & 31) * 2) * sizeof(double)));
    }
    if (ext) vEi = ld16_sc1(rL_i, (unsigned)(((size_t)er * ld + c0 + NB + ec2) * sizeof(double)));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      *reinterpret_cast<double2*>(&sXi[(i >> 5) * LDT + (i & 31) * 2]) = vXi[q];
    }
    if (ext) *reinterpret_cast<double2*>(&sEi[er * kExtPitch + ec2]) = vEi;
  }
  stamp(3);
  // The X_tj go through TWO LDS tiles in turn, and a round waits for nothing it has just issued:
  //   round s:  X_tj(s) registers -> LDS tile s & 1 | LDS barrier | X_tj(s + 1) requested | products of s | their atomic adds
  // The vector-memory counter completes in order, so at the top of round s "at most the adds of round s - 1 outstanding" (8 per lane) says that
  // X_tj(s) has arrived AND the adds of round s - 2 are in: that tile's counter is advanced behind this round's barrier (every wave has passed the
  // same wait).  A round costs its products and one barrier; the drain of the adds (0.8 us) and the loads' latency run under the next round's products.
  double* const sXjb[2] = {sXj, sXj + NB * LDT};   // (the second tile: chol_fused_update's W area)
  double* const sEjb[2] = {sEj, sEj + NB * kExtPitch};   // (... its L16 / W16 area: 64 x kExtPitch doubles fit in 16 x LDT + 16 x kExtPitch? no — see the static_assert)
  static_assert(16 * LDT + 16 * kExtPitch >= NB * kExtPitch, "second appendix stage");
  int tj = __builtin_amdgcn_readfirstlane(items[2 * first]), totj = __builtin_amdgcn_readfirstlane(items[2 * first + 1]);
  double2 vXj[NQ], vEj;
  auto request_j = [&](int tjx) {   // 5 vector-memory loads per lane, whatever the panel (the waits below count them)
    const __amdgpu_buffer_rsrc_t rL_j = tile_rows(Lp, tjx * NB);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      vXj[q] = ld16_sc1(rL_j, (unsigned)(((size_t)(i >> 5) * ld + c0 + (i & 31) * 2) * sizeof(double)));
    }
    vEj = ld16_sc1(rL_j, (unsigned)(((size_t)er * ld + c0 + (ext ? NB : 0) + ec2) * sizeof(double)));
    asm volatile("" ::: "memory");
  };
  __syncthreads();
  if (tid == 0) s_ctl[1] = wait_count(&upd[(tj * N + k) * fs], totj + 1, abort_w, deadline) ? 1 : 0;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return false;
  request_j(tj);
  int tj_prev = -1, tj_prev2 = -1;
  bool ok_all = true;
  for (int s = 0; s < len; ++s) {
    const int b = s & 1;
    if (strip_on && s > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + NT * q;
      *reinterpret_cast<double2*>(&sXjb[b][(i >> 5) * LDT + (i & 31) * 2]) = vXj[q];
    }
    if (ext) *reinterpret_cast<double2*>(&sEjb[b][er * kExtPitch + ec2]) = vEj;
    const int tj_now = tj;
    const bool more = s + 1 < len;
    if (more) {
      tj = __builtin_amdgcn_readfirstlane(items[2 * (first + s + 1)]); totj = __builtin_amdgcn_readfirstlane(items[2 * (first + s + 1) + 1]);
      if (tid == 0) s_ctl[1] = wait_count(&upd[(tj * N + k) * fs], totj + 1, abort_w, deadline) ? 1 : 0;   // (published long ago, as a rule: one look)
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (LDS only: the adds in flight stay in flight)
    if (tid == 0 && tj_prev2 >= 0) atomicAdd(&upd[(ti * N + tj_prev2) * fs], 1);   // the adds of round s - 2 are in (every wave has waited for its own)
    if (more) {
      if (__builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) { ok_all = false; tj_prev2 = -1; break; }   // (round s - 2's tile has just been counted)
      request_j(tj);
    }
    const int rj = __builtin_amdgcn_readfirstlane(tj_now * NB);
    if (strip_on) {
      double4_t acc[TPW];
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        acc[u] = double4_t{0.0, 0.0, 0.0, 0.0};
        acc[u] = mfma_abt<64>(acc[u], sXi + (16 * rs) * LDT, LDT, sXjb[b] + (16 * (tt0 + u)) * LDT, LDT, -1.0, lane);
        if (ext) acc[u] = mfma_abt<16>(acc[u], sEi + (16 * rs) * kExtPitch, kExtPitch, sEjb[b] + (16 * (tt0 + u)) * kExtPitch, kExtPitch, -1.0, lane);
      }
      double* const Sg = S + (size_t)(ri + 16 * rs + crow) * ld + rj + 16 * tt0 + ccol;
#pragma unroll
      for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          (void)__hip_atomic_fetch_add(Sg + (size_t)(4 * reg) * ld + 16 * u, acc[u][reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::: "memory");
    }
    tj_prev2 = tj_prev; tj_prev = tj_now;
  }
  // the last two rounds' tiles (and, after a wait that was given up, whatever was added)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (tj_prev2 >= 0) atomicAdd(&upd[(ti * N + tj_prev2) * fs], 1);
    if (tj_prev >= 0) atomicAdd(&upd[(ti * N + tj_prev) * fs], 1);
  }
  if (!ok_all) return false;
  stamp(6);
  return true;
}

// A CHAIN task: wait for the chain's tiles to have received their updates from outside, then factor them as one dense matrix
// (chol_chain.h).  Afterwards the block inverses of its tiles also go to Vinv (what the launch-per-level back-substitution of very
// large windows reads; nobody in this launch does).
template <bool PROBE>
BSG_DEV bool chol_fused_chain(const FusedCtx& C, int t, const FusedTask& tk, double* smem) {
  const int tid = threadIdx.x;
  const int ld = __builtin_amdgcn_readfirstlane(C.ld), N = ld / NB, fs = __builtin_amdgcn_readfirstlane(C.fs);
  const int b0 = tk.k, m = tk.ti;
  int* const upd = uniform_ptr(C.upd); int* const abort_w = uniform_ptr(C.abort_w);
  const int* const tile_tot = uniform_ptr(C.tile_tot);
  const long long deadline = uniform_i64(C.deadline);
  long long* const probe_ts = uniform_ptr(C.probe_ts);
  int* s_ctl = reinterpret_cast<int*>(smem);
  if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + 1] = wall_clock64();
  if (tid < 64) {   // one lane per tile of the chain
    bool ok = true;
    int ii = 0, jj = 0, left = tid;
    while (ii < m && left > ii) { left -= ii + 1; ++ii; }
    jj = left;
    if (ii < m && ((unsigned)tk.tj >> tid) & 1u) {
      const int a = b0 + ii, b = b0 + jj;
      ok = wait_count(&upd[(a * N + b) * fs], tile_tot[(size_t)a * N + b], abort_w, deadline);
    }
    const int all_ok = __all(ok ? 1 : 0);
    if (tid == 0) s_ctl[0] = all_ok;
  }
  __syncthreads();
  const bool go = __builtin_amdgcn_readfirstlane(s_ctl[0]) != 0;
  __syncthreads();
  if (!go) return false;
  if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + 2] = wall_clock64();
  chain::ChainArgs A;
  A.S = uniform_ptr(C.S); A.Lp = uniform_ptr(C.Lp); A.Winv = uniform_ptr(C.Winv); A.ld = ld; A.c0 = b0; A.m = m; A.present = (unsigned)tk.tj;
  A.nreal = uniform_ptr(C.nreal); A.tile_flag = uniform_ptr(C.potrf_done); A.flag_stride = fs;
  A.Vinv = uniform_ptr(C.Vinv); A.vinv_stride = kVinvStride;
  const bool bad = chain::chain_factor<false>(A, smem, nullptr);
  if (bad && tid == 0) uniform_ptr(C.scal)[SC_CHOL_FAIL] = 1.0;
  if (PROBE && tid == 0) probe_ts[(size_t)t * 8 + 6] = wall_clock64();
  return true;
}

constexpr int kFusedThreads = 512;
constexpr size_t kFusedLds = sizeof(double) * (3 * NB * LDT + 2 * NB * kExtPitch + 16 * LDT + 16 * kExtPitch + 8) > sizeof(double) * chain::chain_lds_doubles() ? sizeof(double) * (3 * NB * LDT + 2 * NB * kExtPitch + 16 * LDT + 16 * kExtPitch + 8)
                                                                                                                   : sizeof(double) * chain::chain_lds_doubles();

template <bool PROBE>
__device__ __forceinline__ void chol_fused_kernel_body(const int bsg_bx, const int bsg_gx, double* __restrict__ S, double* __restrict__ Lp, int ld, const FusedTask* __restrict__ tasks, int n_tasks, const int* __restrict__ tile_tot, const int* __restrict__ nreal, double* __restrict__ Vinv, double* __restrict__ scal, int* sync, double* Winv, int fs, int rhs_strips, const LmDiag& diag, const GradNormRide& gn, long long* probe_ts) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_head[4];
  const int tid = threadIdx.x;
  const int N = ld / NB;
  int* head = sync; int* abort_w = sync + fs; int* exited = sync + 2 * fs;   // (layout: [head | abort | exited | potrf_done (N) | update counts (N x N)] x fs ints)
  FusedCtx C;
  C.S = S; C.Lp = Lp; C.Vinv = Vinv; C.scal = scal; C.tasks = tasks; C.n_tasks = n_tasks; C.tile_tot = tile_tot; C.nreal = nreal; C.ld = ld; C.n_vinv_tiles = N - 1;
  C.Winv = Winv; C.rhs_strips = rhs_strips & 0xff; C.no_turn = (rhs_strips >> 8) & 3;   // (the launch's update mode rides in the argument's second byte: fused_update_mode())
  C.abort_w = abort_w; C.potrf_done = sync + 3 * fs; C.upd = sync + (3 + N) * fs; C.fs = fs;
  C.deadline = (long long)wall_clock64() + kFusedTimeoutTicks + 20LL * n_tasks;   // (+ 0.2 us per task: a dense 30 000-dimensional factorisation is 17 M tasks and half a second)
  C.probe_ts = probe_ts;
  // ONE task per workgroup, taken from a ticket counter: the k-th workgroup to start running gets task k, so the tasks are started in
  // list order whatever order the hardware dispatches the grid in — every counter a task waits for is advanced by a task that was
  // started earlier (no dead-lock, whatever the residency).
  long long t_deq = 0;
  if (PROBE) t_deq = wall_clock64();
  if (tid == 0) s_head[0] = atomicAdd(head, 1);
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane(s_head[0]);
  if (t < n_tasks) {
    if (PROBE && tid == 0) { probe_ts[(size_t)t * 8] = t_deq; probe_ts[(size_t)t * 8 + 7] = bsg_bx; }
    FusedTask tk = tasks[t];
    tk.k = __builtin_amdgcn_readfirstlane(tk.k); tk.ti = __builtin_amdgcn_readfirstlane(tk.ti); tk.tj = __builtin_amdgcn_readfirstlane(tk.tj);
    tk.flags = __builtin_amdgcn_readfirstlane(tk.flags); tk.tot_i = __builtin_amdgcn_readfirstlane(tk.tot_i); tk.tot_j = __builtin_amdgcn_readfirstlane(tk.tot_j);
    tk.need_c = __builtin_amdgcn_readfirstlane(tk.need_c); tk.tot_c = __builtin_amdgcn_readfirstlane(tk.tot_c);
    if (tk.flags & kFusedRider) {
      // a unit of the step's gradient norms (grad_norms_kernel's work: nothing in this launch waits for it, the end-of-step reduction reads it)
      if (gn.nb > 0 && tk.k * 256 < gn.nb) grad_norms_unit<kFusedThreads>(tk.k, tid, gn, smem, smem + 8);
    } else if (tk.flags & kFusedDiagAdd) {
      // the LM diagonal of tile k's real columns (pose_diag_kernel's work), the tile's first update: the stores go out write-through like every
      // update's, then the tile's counter
      if (tid < NB && diag.hdiag) {
        const int pos = tk.k * NB + tid;
        const int j = diag.inat[pos];
        if (j >= 0) {
          const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(S) + (size_t)tk.k * NB * ld, 0, (int)((size_t)NB * ld * sizeof(double)), 0x00020000);
          const unsigned off = (unsigned)(((size_t)tid * ld + pos) * sizeof(double));
          st8_sc1(rS, off, ld8_sc1(rS, off) + lm_diag_value(j, diag));
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) atomicAdd(&C.upd[(tk.k * N + tk.k) * fs], 1);
    } else if (tk.flags & kFusedChain) (void)chol_fused_chain<PROBE>(C, t, tk, smem);
    else if (tk.flags & kFusedSplit) (void)chol_fused_split<PROBE>(C, t, tk, smem);
    else if (tk.flags & kFusedRowSeg) (void)chol_fused_rowseg<PROBE>(C, t, tk, smem);
    else (void)chol_fused_update<PROBE, kFusedThreads>(C, t, tk, smem);
  }
  // leave: the last workgroup out re-zeroes the queue and the counters for the next factorisation
  __syncthreads();
  if (tid == 0) {
    if (ld_flag(abort_w) != 0) scal[SC_CHOL_FAIL] = 2.0;
    s_head[2] = (atomicAdd(exited, 1) == bsg_gx - 1) ? 1 : 0;
  }
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_head[2]) != 0) {
    const int nw = 3 + N + N * N;
    for (int i = tid; i < nw; i += kFusedThreads) __hip_atomic_store(&sync[i * fs], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <bool PROBE>
__global__ __launch_bounds__(kFusedThreads) void chol_fused_kernel(double* __restrict__ S, double* __restrict__ Lp, int ld, const FusedTask* __restrict__ tasks, int n_tasks, const int* __restrict__ tile_tot, const int* __restrict__ nreal, double* __restrict__ Vinv, double* __restrict__ scal, int* sync, double* Winv, int fs, int rhs_strips, LmDiag diag, GradNormRide gn, long long* probe_ts = nullptr) {
  chol_fused_kernel_body<PROBE>((int)blockIdx.x, (int)gridDim.x, S, Lp, ld, tasks, n_tasks, tile_tot, nreal, Vinv, scal, sync, Winv, fs, rhs_strips, diag, gn, probe_ts);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct chol_fused_kernel_Args {
  int bsg_grid;
  double* S;
  double* Lp;
  int ld;
  const FusedTask* tasks;
  int n_tasks;
  const int* tile_tot;
  const int* nreal;
  double* Vinv;
  double* scal;
  int* sync;
  double* Winv;
  int fs;
  int rhs_strips;
  long long* probe_ts;
  LmDiag diag;        // (radius and the step's flags: per round, BatchDyn)
  GradNormRide gn;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct chol_fused_kernel_ArgsG {
  int bsg_grid;
  double __attribute__((address_space(1)))* S;
  double __attribute__((address_space(1)))* Lp;
  int ld;
  const FusedTask __attribute__((address_space(1)))* tasks;
  int n_tasks;
  const int __attribute__((address_space(1)))* tile_tot;
  const int __attribute__((address_space(1)))* nreal;
  double __attribute__((address_space(1)))* Vinv;
  double __attribute__((address_space(1)))* scal;
  int __attribute__((address_space(1)))* sync;
  double __attribute__((address_space(1)))* Winv;
  int fs;
  int rhs_strips;
  long long __attribute__((address_space(1)))* probe_ts;
  LmDiag diag;
  GradNormRide gn;
};
static_assert(sizeof(chol_fused_kernel_ArgsG) == sizeof(chol_fused_kernel_Args), "layout");

template <bool PROBE>
__global__ __launch_bounds__(kFusedThreads) void chol_fused_kernel_batch(const chol_fused_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.x];
  const chol_fused_kernel_ArgsG& a = reinterpret_cast<const chol_fused_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.y >= a.bsg_grid) return;   // (windows interleaved in dispatch order: x = window, y = the window's workgroup — the workgroups of ALL windows take their tickets side by side)
  LmDiag diag = a.diag;   // (small: patched copies; the rest of the entry is read in place)
  diag.inv_radius = 1.0 / bsg_dyn->radius[bsg_w]; diag.compute_scale = bsg_dyn->first[bsg_w]; diag.compute_dcl = bsg_dyn->new_J[bsg_w];
  GradNormRide gn = a.gn;
  if (!bsg_dyn->new_J[bsg_w]) gn.nb = 0;
  chol_fused_kernel_body<PROBE>((int)blockIdx.y, a.bsg_grid, (double*)a.S, (double*)a.Lp, a.ld, (const FusedTask*)a.tasks, a.n_tasks, (const int*)a.tile_tot, (const int*)a.nreal, (double*)a.Vinv, (double*)a.scal, (int*)a.sync, (double*)a.Winv, a.fs, a.rhs_strips, diag, gn, (long long*)a.probe_ts);
}
// ints between two words of the sync area: 16 = a 64-byte line each (packed words of one line that different workgroups write are
// serialised at the memory side: 270 -> 237 us per C2 factorisation when they were moved apart, round 2)
int fused_sync_stride() {
  return 16;
}

// how the updates of a launch reach their tiles (FusedCtx::no_turn): 0 turns (BSGPU_CHOL_NOTURN=0), 1 atomics behind a diagonal tile's LM-diagonal task, 2 atomics
static int fused_update_mode(bool diag_tasks_in_list) {
  static const bool turns = getenv("BSGPU_CHOL_NOTURN") && atoi(getenv("BSGPU_CHOL_NOTURN")) == 0;
  return turns ? 0 : diag_tasks_in_list ? 1 : 2;
}
void launch_chol_fused(hipStream_t s, double* S, double* Lp, int ld, const FusedTask* tasks_dev, int n_tasks, const int* tile_tot_dev, const int* nreal_dev,
                       double* Vinv, double* scal, int* sync_dev, double* Winv, int rhs_rows, const LmDiag& diag, const GradNormRide& gn, bool diag_tasks_in_list) {
  const int rhs_strips = (rhs_rows <= 0 ? 4 : std::min(4, (rhs_rows + 15) / 16)) | (fused_update_mode(diag_tasks_in_list) << 8);
  if (n_tasks <= 0) return;
  const int grid = n_tasks;   // one workgroup per task (about 100 KB of LDS each: one per CU is resident, the rest queue behind them)
  // BSGPU_CHOL_PROBE=<file>: the 20th factorisation of the process runs the stamped variant and dumps, per task, the wall-clock
  // stamps (100 MHz) dequeue / got task / dependencies met / tiles in LDS / solves done / update done / published, and its workgroup
  static const char* probe_file = getenv("BSGPU_CHOL_PROBE");
  static int probe_calls = 0;
  if (probe_file && ++probe_calls == 20) {
    long long* ts = nullptr;
    std::vector<long long> h((size_t)n_tasks * 8, 0);
    std::vector<FusedTask> ht(n_tasks);
    if (hipMalloc((void**)&ts, sizeof(long long) * h.size()) == hipSuccess) {
      (void)hipMemset(ts, 0, sizeof(long long) * h.size());
      hipLaunchKernelGGL((chol_fused_kernel<true>), dim3(grid), dim3(kFusedThreads), kFusedLds, s, S, Lp, ld, tasks_dev, n_tasks, tile_tot_dev, nreal_dev, Vinv,
                         scal, sync_dev, Winv, fused_sync_stride(), rhs_strips, diag, gn, ts);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h.data(), ts, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
      (void)hipMemcpy(ht.data(), tasks_dev, sizeof(FusedTask) * ht.size(), hipMemcpyDeviceToHost);
      (void)hipFree(ts);
      if (FILE* f = fopen(probe_file, "w")) {
        fprintf(f, "# task k ti tj flags need_c tot_c | t_dequeue t_got t_deps t_loaded t_solved t_updated t_published wg\n");
        for (int i = 0; i < n_tasks; ++i) {
          fprintf(f, "%d %d %d %d %d %d %d", i, ht[i].k, ht[i].ti, ht[i].tj, ht[i].flags, ht[i].need_c, ht[i].tot_c);
          for (int q = 0; q < 8; ++q) fprintf(f, " %lld", h[(size_t)i * 8 + q]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
      return;
    }
  }
  hipLaunchKernelGGL((chol_fused_kernel<false>), dim3(grid), dim3(kFusedThreads), kFusedLds, s, S, Lp, ld, tasks_dev, n_tasks, tile_tot_dev, nreal_dev, Vinv, scal,
                     sync_dev, Winv, fused_sync_stride(), rhs_strips, diag, gn, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// backward substitution L^T y = y', one launch per schedule step in reverse order, one 1024-thread
// workgroup per panel of the step:
//   rhs = y'[k] - sum_{t in rows(k)} L(t, k)^T y[t]   (16 row groups x 64 columns, loads issued up front)
//   solve L_kk^T y_k = rhs                              (one wave, four 16-row block steps with the diagonal-block inverses)
// y is in S (solver) order; rows >= nreal of a tile are kept zero.
// ---------------------------------------------------------------------------------------------------
// The back-substitution is latency work: per panel a handful of dependent round trips to L2 / HBM.  So every load a
// panel needs is put in flight at once — the L_kk tile, and the row-tile entries in chunks of CH tiles (a loop that
// loads and consumes one row tile at a time pays one round trip per tile) — and the piece-walking kernel keeps y and
// its panels' descriptors / row lists in LDS, so that nothing but L is fetched inside the walk.
constexpr int kBsChunk = 6;
constexpr int kBsChunkDeep = 3;   // row tiles per panel of the two-panel-deep walk (chol_backsolve_chain_kernel)
// workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it
BSG_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// one workgroup per chain (a separator, or a piece of the nested-dissection ordering), walking its panels from its last
// tile down to its first; everything a chain depends on outside itself was solved by an earlier launch (dense_plan.h)
constexpr int kBsMaxRows = DensePlan::kBsDescRows;   // row lists up to this length are staged in LDS
// CH: row tiles of a panel whose entries are loaded up front (further ones go through the two-at-a-time loop, each a round trip).
// DEEP: the loads of TWO panels ahead are in flight (two register sets, the walk unrolled by two) — for chains whose panels have
// at most CH row tiles each (the level-synchronous form of a banded window: CH = 3), where a step is otherwise as long as one
// memory round trip although its arithmetic takes a third of that.
template <int CH>
struct BsPanelRegs {
  double dl[4], l[CH][4], vinv;
  int r0[CH];
};
// what the single-launch form (chol_backsolve_fused_kernel) adds to a chain's walk: the turn to wait for, where y is shared
struct BsFused {
  int* abort_w;            // sticky failure flag
  const int* wait_flags;   // null: nothing to wait for; else the walk starts when every flag in [wait_lo, wait_hi) is set (the update items
  int wait_lo, wait_hi, flag_stride;   // of the phase two groups up: they wrote this chain's start values)
  const int* wait_word;    // ... and when *wait_word has reached wait_word_val: the chains of the group just before (null: the root group)
  int wait_word_val;
  int* done_word;          // bumped once the chain's y is out
  const int* tile_updated; // per tile: an earlier phase has written y there (else the start value is y_init)
  long long deadline;
  long long* ts;           // debugging: 16 wall-clock stamps per workgroup (BSGPU_BACKSOLVE_PROBE), or null
  int ts_row;
};
template <bool Y_IN_LDS, int CH, bool DEEP, bool FUSED, bool USE_W>
BSG_DEV void bs_chain_walk(const double* Lp, const double* Vinv /* USE_W: the tiles' full inverses, 64 x 64 row-major each */, int ld, const int* __restrict__ bs_desc, int b, int e,
                           const int* __restrict__ rows_flat, double* y, int npad, int max_len, const double* __restrict__ y_init,
                           const int* __restrict__ iperm, int n_pose, double* __restrict__ y_tan, double* __restrict__ delta,
                           const BsFused& F) {
  constexpr bool y_in_lds = Y_IN_LDS;
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double* sL = dyn;                         // 64 x 65
  double* sp = sL + NB * (NB + 1);          // 16 x 64
  double* sV = sp + 16 * NB;                // 4 x 16 x 16: inverses of the diagonal 16x16 blocks of L_kk
  // the solution vector: in LDS next to the tiles when it fits (npad <= 12 288) — otherwise the walk reads and writes y itself (a
  // row is read only after the workgroup that solved it — this one, or one of an earlier launch — has written it; L1 is per CU
  // and write-through, and no line of y holds rows of two tiles)
  double* sy = y_in_lds ? sV + 1024 : y;    // npad
  int* s_nrows = reinterpret_cast<int*>(sV + 1024 + (y_in_lds ? npad : 0));   // max_len
  int* s_rowoff = s_nrows + max_len;                  // max_len
  int* s_nr = s_rowoff + max_len;                     // max_len
  int* s_rows = s_nr + max_len;                       // max_len x kBsMaxRows
  const int tid = threadIdx.x;
  const int len = e - b;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)npad * sizeof(double)), 0x00020000);
  (void)ry;
  if (!FUSED) {
    // y_init (first launch of a solve, a single chain): the forward-substituted rhs row of the factor, copied to y on the way
    if (y_init) { for (int i = tid; i < npad; i += 1024) { const double v = (i < npad - NB) ? y_init[i] : 0.0; sy[i] = v; y[i] = v; } }   // (the last tile is the rhs tile itself)
    else if (y_in_lds) { for (int i = tid; i < npad; i += 1024) sy[i] = y[i]; }
  }
  // the records of this chain's tiles (DensePlan::bs_desc: one coalesced round, not tile -> panel -> row list)
  for (int i = tid; i < len * DensePlan::kBsDescInts; i += 1024) {
    const int p = i / DensePlan::kBsDescInts, q = i - p * DensePlan::kBsDescInts;
    const int v = bs_desc[(size_t)b * DensePlan::kBsDescInts + i];
    if (q == 0) s_nrows[p] = v; else if (q == 1) s_rowoff[p] = v; else if (q == 2) s_nr[p] = v; else s_rows[p * kBsMaxRows + (q - 3)] = v;
  }
  __syncthreads();
  // Software pipeline over the chain: the tiles of panel k-1 (k-2 with DEEP) do not depend on y, so their loads are issued BEFORE
  // the barrier + 64-pivot triangular solve of panel k and are in flight underneath it (the barrier is an LDS-only one: a
  // __syncthreads() would wait for those loads).
  const int c = tid & 63, part = tid >> 6;
  auto issue = [&](int k, BsPanelRegs<CH>& R) {
    const int p = k - b, n_rows = s_nrows[p], c0 = k * NB;
    const int* rows = (n_rows <= kBsMaxRows) ? (s_rows + p * kBsMaxRows) : (rows_flat + s_rowoff[p]);
    if (USE_W) {   // W = L_kk^-1 in the layout of the row tiles: rows 4 part .. 4 part + 3, column c
#pragma unroll
      for (int q = 0; q < 4; ++q) R.dl[q] = Vinv[(size_t)k * NB * NB + (4 * part + q) * NB + c];
      R.vinv = 0.0;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + 1024 * q;
        const int r = i >> 6, cc = i & 63;
        R.dl[q] = (cc <= r) ? Lp[(size_t)(c0 + r) * ld + c0 + cc] : 0.0;
      }
      R.vinv = Vinv[(size_t)k * kVinvStride + tid];
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool ok = u < n_rows;
      R.r0[u] = (ok ? rows[u] : k) * NB + 4 * part;
#pragma unroll
      for (int i = 0; i < 4; ++i) R.l[u][i] = ok ? Lp[(size_t)(R.r0[u] + i) * ld + c0 + c] : 0.0;
    }
  };
  // one panel: R holds its loaded entries; `next` (>= b, or < b for none) is the panel whose loads go out into R once R is consumed
  auto step = [&](int k, BsPanelRegs<CH>& R, int next) {
    const int p = k - b, n_rows = s_nrows[p], nr = s_nr[p], c0 = k * NB;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = fma(R.l[u][i], sy[R.r0[u] + i], acc);
    if (n_rows > CH) {   // (wide separators / dense windows only)
      const int* rows = (n_rows <= kBsMaxRows) ? (s_rows + p * kBsMaxRows) : (rows_flat + s_rowoff[p]);
      for (int q0 = CH; q0 < n_rows; q0 += 2) {
        double l2[2][4];
        int r2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool ok = q0 + u < n_rows;
          r2[u] = (ok ? rows[q0 + u] : k) * NB + 4 * part;
#pragma unroll
          for (int i = 0; i < 4; ++i) l2[u][i] = ok ? Lp[(size_t)(r2[u] + i) * ld + c0 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = fma(l2[u][i], sy[r2[u] + i], acc);
      }
    }
    if (USE_W) {
      // y_k = W^T (y'_k - sum): two 16-part reductions, every thread busy, no dependent chain (the substitution with L_kk^T below
      // is ~130 dependent FMAs of one wave, each behind its own LDS read: 2.6 - 4.4 us a panel against ~1)
      const double w0 = R.dl[0], w1 = R.dl[1], w2 = R.dl[2], w3 = R.dl[3];
      sp[part * NB + c] = acc;
      if (next >= b) issue(next, R);
      if (y_in_lds) lds_barrier(); else __syncthreads();
      double* st = sL;   // 64 doubles
      if (tid < NB) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += sp[q * NB + tid];
        st[tid] = (tid < nr) ? (sy[c0 + tid] - sum) : 0.0;
      }
      if (y_in_lds) lds_barrier(); else __syncthreads();
      const double a2 = w0 * st[4 * part] + w1 * st[4 * part + 1] + w2 * st[4 * part + 2] + w3 * st[4 * part + 3];
      sp[part * NB + c] = a2;
      if (y_in_lds) lds_barrier(); else __syncthreads();
      if (tid < NB) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += sp[q * NB + tid];
        v = (tid < nr) ? v : 0.0;
        if (FUSED) st8_sc1(ry, (unsigned)((c0 + tid) * sizeof(double)), v); else y[c0 + tid] = v;
        if (y_in_lds) sy[c0 + tid] = v;
        if (y_tan) {
          const int j = iperm[k * NB + tid];
          if (j >= 0) { y_tan[j] = v; delta[j] = -v; }
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 1024 * q;
      sL[(i >> 6) * (NB + 1) + (i & 63)] = R.dl[q];
    }
    sp[part * NB + c] = acc;
    sV[tid] = R.vinv;
    if (next >= b) issue(next, R);
    if (y_in_lds) lds_barrier(); else __syncthreads();
    if (tid < NB) {
      // L_kk^T y = t on one wave, lane = row.  Blocked by 16 with the inverses of the diagonal blocks (V_b = L_bb^-1, kept
      // by the factorisation for its own triangular solves): four dependent block steps of independent loads and
      // straight-line FMAs, instead of 64 dependent pivots each with its own LDS round trip and branch.
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += sp[q * NB + tid];
      double v = (tid < nr) ? (sy[c0 + tid] - sum) : 0.0;
      const int blk = tid >> 4, li = tid & 15;
#pragma unroll
      for (int bb = 3; bb >= 0; --bb) {
        double yb = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) yb = fma(sV[bb * 256 + m * 16 + li], readlane_d(v, 16 * bb + m), yb);   // (V_b^T t_b)_i
        v = (blk == bb) ? yb : v;
        if (bb > 0) {
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            const double upd = fma(-sL[(16 * bb + m) * (NB + 1) + tid], readlane_d(v, 16 * bb + m), v);
            v = (tid < 16 * bb) ? upd : v;
          }
        }
      }
      v = (tid < nr) ? v : 0.0;
      if (FUSED) st8_sc1(ry, (unsigned)((c0 + tid) * sizeof(double)), v); else y[c0 + tid] = v;
      if (y_in_lds) sy[c0 + tid] = v;
      if (y_tan) {   // the solution in tangent (natural) order and the step -y, written where the tile is solved
        const int j = iperm[k * NB + tid];
        if (j >= 0) { y_tan[j] = v; delta[j] = -v; }
      }
    }
    }
    if (y_in_lds) lds_barrier(); else __syncthreads();   // (global y: the stores must have left the wave before the other waves read them)
  };
  // (single launch) the first loads are out; now wait for the turn, then take the chain's start values: what the earlier phases left
  // in y (write-through stores, read past this CU's caches), or the rhs row where nothing has been applied yet
  int n_stamp = 4;
  auto stamp = [&](int slot) { if (FUSED && F.ts && tid == 0) F.ts[(size_t)F.ts_row * 16 + slot] = wall_clock64(); };
  auto fused_enter = [&]() {
    if (!FUSED) return true;
    stamp(1);
    int* s_ok = reinterpret_cast<int*>(s_rows + (size_t)max_len * kBsMaxRows);
    if (tid < 64) {   // (the first wave polls, a flag per lane)
      bool ok = !F.wait_flags || F.wait_hi <= F.wait_lo || wait_flags(F.wait_flags, F.wait_lo, F.wait_hi, F.abort_w, F.deadline, F.flag_stride);
      if (ok && F.wait_word) {
        int okw = 1;
        if (tid == 0) okw = wait_count(F.wait_word, F.wait_word_val, F.abort_w, F.deadline) ? 1 : 0;
        ok = __builtin_amdgcn_readfirstlane(okw) != 0;
      }
      if (tid == 0) *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(*s_ok) == 0) return false;
    stamp(2);
    for (int i = b * NB + tid; i < e * NB; i += 1024)
      sy[i] = F.tile_updated[i >> 6] ? ld8_sc1(ry, (unsigned)(i * sizeof(double))) : y_init[i];
    // the row tiles outside the chain (its parent separator's: solved by now): their y, for the products of the walk
    for (int p = 0; p < len; ++p) {
      const int n_rows = s_nrows[p];
      const int* rows = (n_rows <= kBsMaxRows) ? (s_rows + p * kBsMaxRows) : (rows_flat + s_rowoff[p]);
      for (int q = tid >> 6; q < n_rows; q += 16) {
        const int t = rows[q];
        if (t < b || t >= e) sy[t * NB + (tid & 63)] = ld8_sc1(ry, (unsigned)((t * NB + (tid & 63)) * sizeof(double)));
      }
    }
    __syncthreads();
    stamp(3);
    return true;
  };
  bool entered = true;
  if (DEEP) {
    BsPanelRegs<CH> A, B;
    if (len > 0) issue(e - 1, A);
    if (len > 1) issue(e - 2, B);
    entered = fused_enter();
    if (entered)
      for (int k = e - 1; k >= b; k -= 2) {
        step(k, A, k - 2);
        if (n_stamp < 12) stamp(n_stamp++);
        if (k - 1 >= b) { step(k - 1, B, k - 3); if (n_stamp < 12) stamp(n_stamp++); }
      }
  } else {
    BsPanelRegs<CH> A;
    if (len > 0) issue(e - 1, A);
    entered = fused_enter();
    if (entered)
      for (int k = e - 1; k >= b; --k) step(k, A, k - 1);
  }
  if (FUSED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && entered) atomicAdd(F.done_word, 1);
    stamp(12);
  }
}
template <bool Y_IN_LDS, int CH, bool DEEP, bool USE_W>
__device__ __forceinline__ void chol_backsolve_chain_kernel_body(const int bsg_bx, const int bsg_gx, const double* S, const double* Lp, const double* Vinv, int ld, const int* __restrict__ bs_desc, const int* __restrict__ chain_begin, const int* __restrict__ chain_end, const int* __restrict__ rows_flat, double* y, int npad, int max_len, const double* __restrict__ y_init, const int* __restrict__ iperm, int n_pose, double* __restrict__ y_tan, double* __restrict__ delta) {
  (void)S;
  const BsFused none{};
  bs_chain_walk<Y_IN_LDS, CH, DEEP, false, USE_W>(Lp, Vinv, ld, bs_desc, chain_begin[bsg_bx], chain_end[bsg_bx], rows_flat, y, npad, max_len,
                                           y_init, iperm, n_pose, y_tan, delta, none);
}
template <bool Y_IN_LDS, int CH, bool DEEP, bool USE_W>
__global__ __launch_bounds__(1024) void chol_backsolve_chain_kernel(const double* S, const double* Lp, const double* Vinv, int ld, const int* __restrict__ bs_desc, const int* __restrict__ chain_begin, const int* __restrict__ chain_end, const int* __restrict__ rows_flat, double* y, int npad, int max_len, const double* __restrict__ y_init, const int* __restrict__ iperm, int n_pose, double* __restrict__ y_tan, double* __restrict__ delta) {
  chol_backsolve_chain_kernel_body<Y_IN_LDS, CH, DEEP, USE_W>((int)blockIdx.x, (int)gridDim.x, S, Lp, Vinv, ld, bs_desc, chain_begin, chain_end, rows_flat, y, npad, max_len, y_init, iperm, n_pose, y_tan, delta);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct chol_backsolve_chain_kernel_Args {
  int bsg_grid;
  const double* S;
  const double* Lp;
  const double* Vinv;
  int ld;
  const int* bs_desc;
  const int* chain_begin;
  const int* chain_end;
  const int* rows_flat;
  double* y;
  int npad;
  int max_len;
  const double* y_init;
  const int* iperm;
  int n_pose;
  double* y_tan;
  double* delta;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct chol_backsolve_chain_kernel_ArgsG {
  int bsg_grid;
  const double __attribute__((address_space(1)))* S;
  const double __attribute__((address_space(1)))* Lp;
  const double __attribute__((address_space(1)))* Vinv;
  int ld;
  const int __attribute__((address_space(1)))* bs_desc;
  const int __attribute__((address_space(1)))* chain_begin;
  const int __attribute__((address_space(1)))* chain_end;
  const int __attribute__((address_space(1)))* rows_flat;
  double __attribute__((address_space(1)))* y;
  int npad;
  int max_len;
  const double __attribute__((address_space(1)))* y_init;
  const int __attribute__((address_space(1)))* iperm;
  int n_pose;
  double __attribute__((address_space(1)))* y_tan;
  double __attribute__((address_space(1)))* delta;
};
static_assert(sizeof(chol_backsolve_chain_kernel_ArgsG) == sizeof(chol_backsolve_chain_kernel_Args), "layout");

template <bool Y_IN_LDS, int CH, bool DEEP, bool USE_W>
__global__ __launch_bounds__(1024) void chol_backsolve_chain_kernel_batch(const chol_backsolve_chain_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.x];
  const chol_backsolve_chain_kernel_ArgsG& a = reinterpret_cast<const chol_backsolve_chain_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.y >= a.bsg_grid) return;   // (windows interleaved in dispatch order: x = window, y = the window's workgroup — the workgroups of ALL windows take their tickets side by side)
  chol_backsolve_chain_kernel_body<Y_IN_LDS, CH, DEEP, USE_W>((int)blockIdx.y, a.bsg_grid, (double*)a.S, (double*)a.Lp, (double*)a.Vinv, a.ld, (const int*)a.bs_desc, (const int*)a.chain_begin, (const int*)a.chain_end, (const int*)a.rows_flat, (double*)a.y, a.npad, a.max_len, (const double*)a.y_init, (const int*)a.iperm, a.n_pose, (double*)a.y_tan, (double*)a.delta);
}

// ---------------------------------------------------------------------------------------------------
// The whole level-synchronous back-substitution in ONE launch: a workgroup per chain and one per (phase, target panel) update item,
// all resident at once (the launcher checks the grid against the number of CUs), each doing what does not depend on y — its records,
// the loads of its first tiles — before it waits for its turn:
//   chains of group g     wait for the flags of every item of phase g-1   (nothing for the root group)
//   items of phase g      wait for done_chain[g] == chains of group g
// y travels between workgroups with write-through stores and loads that pass the CU's caches (as the factor does in
// chol_fused_kernel); waits are bounded (abort flag + deadline -> SC_CHOL_FAIL = 2); the last workgroup out clears the counters.
// Seven launches of 13 + 5 us each become one: the prologues overlap and a hand-over costs ~2 us instead of a launch boundary.
// sync (words a cache line apart): [0] abort, [1] exited, [2 .. 2+G) done_chain (counters: at most eight writers), [2+G .. 2+G+n_items) one flag per
// update item, then the ticket
// ---------------------------------------------------------------------------------------------------
template <int CH, bool DEEP>
__device__ __forceinline__ void chol_backsolve_fused_kernel_body(const int bsg_bx, const int bsg_gx, const double* Lp, const double* Winv, int ld, const int* __restrict__ bs_desc, const int* __restrict__ chain_begin, const int* __restrict__ chain_end, const int* __restrict__ rows_flat, int n_chains, const int* __restrict__ chain_group, const int* __restrict__ grp_nchains, const int* __restrict__ grp_nitems, int G, const int* __restrict__ items, const int* __restrict__ upd_rows, const int* __restrict__ tile_updated, double* y, int npad, int max_len, const double* __restrict__ y_init, const int* __restrict__ iperm, int n_pose, double* __restrict__ y_tan, double* __restrict__ delta, int* sync, double* __restrict__ scal, long long* ts, const int* __restrict__ order, int fs) {
  const int tid = threadIdx.x;
  // (every word of the sync area `fs` ints — a cache line — apart: see chol_fused_kernel)
  int* abort_w = sync; int* exited = sync + fs; int* done_chain = sync + 2 * fs; int* item_flag = sync + (size_t)(2 + G) * fs;
  const int n_items_total = bsg_gx - n_chains;
  // roles are handed out by a ticket in dependency order (the k-th workgroup to START gets role order[k]): whoever a workgroup waits
  // for holds an earlier ticket and is therefore running — no dead-lock even when the grid is not resident at once (several
  // contexts sharing the GPU), as in chol_fused_kernel
  __shared__ int s_role;
  if (tid == 0) s_role = order[atomicAdd(sync + (size_t)(2 + G + n_items_total) * fs, 1)];
  __syncthreads();
  const int bid = __builtin_amdgcn_readfirstlane(s_role);
  if (ts && tid == 0) ts[(size_t)bid * 16] = wall_clock64();
  const long long deadline = (long long)wall_clock64() + kFusedTimeoutTicks;
  if (bid < n_chains) {
    const int ch = bid, g = chain_group[ch];
    BsFused F;
    F.abort_w = abort_w; F.deadline = deadline; F.tile_updated = tile_updated;
    // start values: what the update items of phases <= g - 2 left (the phases follow each other: an item waits for the phase before
    // its own); the rows of group g - 1 the chain multiplies in itself, once that group's chains are done
    int hi = 0;
    for (int q = 0; q + 2 <= g; ++q) hi += grp_nitems[q];   // (every phase up to g - 2: a phase without items must not cut the order)
    F.wait_flags = g > 1 ? item_flag : nullptr; F.flag_stride = fs;
    F.wait_lo = 0; F.wait_hi = g > 1 ? hi : 0;
    F.wait_word = g > 0 ? done_chain + (size_t)(g - 1) * fs : nullptr; F.wait_word_val = g > 0 ? grp_nchains[g - 1] : 0;
    F.done_word = done_chain + (size_t)g * fs; F.ts = ts; F.ts_row = bid;
    bs_chain_walk<true, CH, DEEP, true, true>(Lp, Winv, ld, bs_desc, chain_begin[ch], chain_end[ch], rows_flat, y, npad, max_len, y_init, iperm, n_pose,
                                        y_tan, delta, F);
  } else {
    // an update item: y_k = (y_k or the rhs row) - sum_t L(t,k)^T y_t over the row tiles t of panel k that group `phase` solved
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    double* sp = dyn;                                   // 16 x 64
    int* s_ok = reinterpret_cast<int*>(dyn + 16 * NB);
    const int* it = items + 4 * ((size_t)bid - n_chains);
    const int k = it[0], off = it[1], n = it[2], phase = it[3] & 0xffff, first = it[3] >> 16;
    const int c = tid & 63, part = tid >> 6, c0 = k * NB;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((size_t)npad * sizeof(double)), 0x00020000);
    double l[kBsChunk][4];
    int r0[kBsChunk];
#pragma unroll
    for (int u = 0; u < kBsChunk; ++u) {
      const bool ok = u < n;
      r0[u] = (ok ? upd_rows[off + u] : k) * NB + 4 * part;
#pragma unroll
      for (int i = 0; i < 4; ++i) l[u][i] = ok ? Lp[(size_t)(r0[u] + i) * ld + c0 + c] : 0.0;
    }
    if (ts && tid == 0) ts[(size_t)bid * 16 + 1] = wall_clock64();
    // its turn: the chains of group `phase` are done — and so are the items of the phase before (two phases may update the same panel)
    if (tid < 64) {
      int hi_prev = 0;
      for (int q = 0; q < phase; ++q) hi_prev += grp_nitems[q];   // (every earlier phase)
      bool okp = hi_prev == 0 || wait_flags(item_flag, 0, hi_prev, abort_w, deadline, fs);
      int okw = 1;
      if (okp && tid == 0) okw = wait_count(done_chain + (size_t)phase * fs, grp_nchains[phase], abort_w, deadline) ? 1 : 0;
      okp = okp && __builtin_amdgcn_readfirstlane(okw) != 0;
      if (tid == 0) *s_ok = okp ? 1 : 0;
    }
    __syncthreads();
    if (ts && tid == 0) ts[(size_t)bid * 16 + 2] = wall_clock64();
    const bool ok_turn = __builtin_amdgcn_readfirstlane(*s_ok) != 0;
    if (ok_turn) {
      double acc = 0.0;
      double base = 0.0;   // (requested with the row values, not after the reduction: one round trip less)
      if (tid < NB) base = first ? y_init[c0 + tid] : ld8_sc1(ry, (unsigned)((c0 + tid) * sizeof(double)));
#pragma unroll
      for (int u = 0; u < kBsChunk; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fma(l[u][i], (u < n) ? ld8_sc1(ry, (unsigned)((r0[u] + i) * sizeof(double))) : 0.0, acc);
      for (int q0 = kBsChunk; q0 < n; q0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (q0 + u >= n) continue;
          const int r2 = upd_rows[off + q0 + u] * NB + 4 * part;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = fma(Lp[(size_t)(r2 + i) * ld + c0 + c], ld8_sc1(ry, (unsigned)((r2 + i) * sizeof(double))), acc);
        }
      }
      sp[part * NB + c] = acc;
      __syncthreads();
      if (tid < NB) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += sp[q * NB + tid];
        st8_sc1(ry, (unsigned)((c0 + tid) * sizeof(double)), base - sum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(item_flag + (size_t)(bid - n_chains) * fs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ts && tid == 0) ts[(size_t)bid * 16 + 12] = wall_clock64();
    }
  }
  // leave: the last workgroup out clears the counters for the next solve
  __syncthreads();
  __shared__ int s_last;
  if (tid == 0) {
    if (ld_flag(abort_w) != 0) scal[SC_CHOL_FAIL] = 2.0;
    s_last = (atomicAdd(exited, 1) == bsg_gx - 1) ? 1 : 0;
  }
  __syncthreads();
  if (s_last) for (int i = tid; i < 2 + G + n_items_total + 1; i += 1024) __hip_atomic_store(&sync[(size_t)i * fs], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int CH, bool DEEP>
__global__ __launch_bounds__(1024) void chol_backsolve_fused_kernel(const double* Lp, const double* Winv, int ld, const int* __restrict__ bs_desc, const int* __restrict__ chain_begin, const int* __restrict__ chain_end, const int* __restrict__ rows_flat, int n_chains, const int* __restrict__ chain_group, const int* __restrict__ grp_nchains, const int* __restrict__ grp_nitems, int G, const int* __restrict__ items, const int* __restrict__ upd_rows, const int* __restrict__ tile_updated, double* y, int npad, int max_len, const double* __restrict__ y_init, const int* __restrict__ iperm, int n_pose, double* __restrict__ y_tan, double* __restrict__ delta, int* sync, double* __restrict__ scal, long long* ts, const int* __restrict__ order, int fs) {
  chol_backsolve_fused_kernel_body<CH, DEEP>((int)blockIdx.x, (int)gridDim.x, Lp, Winv, ld, bs_desc, chain_begin, chain_end, rows_flat, n_chains, chain_group, grp_nchains, grp_nitems, G, items, upd_rows, tile_updated, y, npad, max_len, y_init, iperm, n_pose, y_tan, delta, sync, scal, ts, order, fs);
}
// one launch over several windows (bsgpu_batch.cpp): blockIdx.y picks the window of list `bsg_list`, its arguments come from memory
struct chol_backsolve_fused_kernel_Args {
  int bsg_grid;
  const double* Lp;
  const double* Winv;
  int ld;
  const int* bs_desc;
  const int* chain_begin;
  const int* chain_end;
  const int* rows_flat;
  int n_chains;
  const int* chain_group;
  const int* grp_nchains;
  const int* grp_nitems;
  int G;
  const int* items;
  const int* upd_rows;
  const int* tile_updated;
  double* y;
  int npad;
  int max_len;
  const double* y_init;
  const int* iperm;
  int n_pose;
  double* y_tan;
  double* delta;
  int* sync;
  double* scal;
  long long* ts;
  const int* order;
  int fs;
};
// (the same entry as the kernel reads it: its pointers are GLOBAL pointers — read as generic ones every load through them would be a FLAT
// instruction, which also counts against the LDS counter and serialises the kernels that overlap gathers with LDS traffic)
struct chol_backsolve_fused_kernel_ArgsG {
  int bsg_grid;
  const double __attribute__((address_space(1)))* Lp;
  const double __attribute__((address_space(1)))* Winv;
  int ld;
  const int __attribute__((address_space(1)))* bs_desc;
  const int __attribute__((address_space(1)))* chain_begin;
  const int __attribute__((address_space(1)))* chain_end;
  const int __attribute__((address_space(1)))* rows_flat;
  int n_chains;
  const int __attribute__((address_space(1)))* chain_group;
  const int __attribute__((address_space(1)))* grp_nchains;
  const int __attribute__((address_space(1)))* grp_nitems;
  int G;
  const int __attribute__((address_space(1)))* items;
  const int __attribute__((address_space(1)))* upd_rows;
  const int __attribute__((address_space(1)))* tile_updated;
  double __attribute__((address_space(1)))* y;
  int npad;
  int max_len;
  const double __attribute__((address_space(1)))* y_init;
  const int __attribute__((address_space(1)))* iperm;
  int n_pose;
  double __attribute__((address_space(1)))* y_tan;
  double __attribute__((address_space(1)))* delta;
  int __attribute__((address_space(1)))* sync;
  double __attribute__((address_space(1)))* scal;
  long long __attribute__((address_space(1)))* ts;
  const int __attribute__((address_space(1)))* order;
  int fs;
};
static_assert(sizeof(chol_backsolve_fused_kernel_ArgsG) == sizeof(chol_backsolve_fused_kernel_Args), "layout");

template <int CH, bool DEEP>
__global__ __launch_bounds__(1024) void chol_backsolve_fused_kernel_batch(const chol_backsolve_fused_kernel_Args* __restrict__ bsg_A, const BatchDyn* __restrict__ bsg_dyn, int bsg_list) {
  const int bsg_w = bsg_dyn->idx[bsg_list][blockIdx.x];
  const chol_backsolve_fused_kernel_ArgsG& a = reinterpret_cast<const chol_backsolve_fused_kernel_ArgsG*>(bsg_A)[bsg_w];
  if ((int)blockIdx.y >= a.bsg_grid) return;   // (windows interleaved in dispatch order: x = window, y = the window's workgroup — the workgroups of ALL windows take their tickets side by side)
  chol_backsolve_fused_kernel_body<CH, DEEP>((int)blockIdx.y, a.bsg_grid, (double*)a.Lp, (double*)a.Winv, a.ld, (const int*)a.bs_desc, (const int*)a.chain_begin, (const int*)a.chain_end, (const int*)a.rows_flat, a.n_chains, (const int*)a.chain_group, (const int*)a.grp_nchains, (const int*)a.grp_nitems, a.G, (const int*)a.items, (const int*)a.upd_rows, (const int*)a.tile_updated, (double*)a.y, a.npad, a.max_len, (const double*)a.y_init, (const int*)a.iperm, a.n_pose, (double*)a.y_tan, (double*)a.delta, (int*)a.sync, (double*)a.scal, (long long*)a.ts, (const int*)a.order, a.fs);
}
// false: not launched (the grid would not be resident at once, or y does not fit LDS) — the caller takes the launch-per-level path
bool launch_chol_backsolve_fused(hipStream_t s, const double* Lp, const double* Winv, int ld, const int* bs_desc_dev, const int* chain_begin_dev,
                                 const int* chain_end_dev, const int* rows_flat_dev, int n_chains, const int* chain_group_dev,
                                 const int* grp_nchains_dev, const int* grp_nitems_dev, int G, const int* items_dev, int n_items,
                                 const int* upd_rows_dev, const int* tile_updated_dev, double* y, int npad, int max_chain_len, int max_rows,
                                 const double* y_init, const int* iperm_dev, int n_pose, double* y_tan, double* delta, int* sync_dev, double* scal,
                                 const int* order_dev) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  static const bool off = getenv("BSGPU_BACKSOLVE_FUSED") && atoi(getenv("BSGPU_BACKSOLVE_FUSED")) == 0;
  const size_t lds = chol_backsolve_chain_lds(npad, max_chain_len) + 16;
  (void)n_cu;   // (roles are ticketed in dependency order: the grid need not be resident at once; beyond ~2 workgroups per CU the
                // launch-per-level form is the better one)
  if (off || !Winv || !order_dev || n_chains + n_items > 2 * n_cu || lds > (size_t)160 * 1024 - 256 /* (4 bytes of static LDS) */ || getenv("BSGPU_BACKSOLVE_GLOBAL_Y")) return false;
  const bool deep = max_rows > 0 && max_rows <= kBsChunkDeep && max_chain_len > 1;
  // BSGPU_BACKSOLVE_PROBE=<file>: the 20th solve of the process is stamped (100 MHz wall clock): per workgroup start / loads out /
  // turn / start values / after each panel / done
  static const char* probe_file = getenv("BSGPU_BACKSOLVE_PROBE");
  static int probe_calls = 0;
  long long* ts = nullptr;
  const int grid = n_chains + n_items;
  if (probe_file && ++probe_calls == 20) { (void)hipMalloc((void**)&ts, sizeof(long long) * 16 * grid); (void)hipMemsetAsync(ts, 0, sizeof(long long) * 16 * grid, s); }
  if (deep)
    hipLaunchKernelGGL((chol_backsolve_fused_kernel<kBsChunkDeep, true>), dim3(n_chains + n_items), dim3(1024), lds, s, Lp, Winv, ld, bs_desc_dev,
                       chain_begin_dev, chain_end_dev, rows_flat_dev, n_chains, chain_group_dev, grp_nchains_dev, grp_nitems_dev, G, items_dev,
                       upd_rows_dev, tile_updated_dev, y, npad, max_chain_len, y_init, iperm_dev, n_pose, y_tan, delta, sync_dev, scal, ts, order_dev, fused_sync_stride());
  else
    hipLaunchKernelGGL((chol_backsolve_fused_kernel<kBsChunk, false>), dim3(n_chains + n_items), dim3(1024), lds, s, Lp, Winv, ld, bs_desc_dev,
                       chain_begin_dev, chain_end_dev, rows_flat_dev, n_chains, chain_group_dev, grp_nchains_dev, grp_nitems_dev, G, items_dev,
                       upd_rows_dev, tile_updated_dev, y, npad, max_chain_len, y_init, iperm_dev, n_pose, y_tan, delta, sync_dev, scal, ts, order_dev, fused_sync_stride());
  if (ts) {
    std::vector<long long> h((size_t)16 * grid);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), ts, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    (void)hipFree(ts);
    if (FILE* f = fopen(probe_file, "w")) {
      long long t0 = 0;
      for (int i = 0; i < grid; ++i) if (h[(size_t)i * 16] && (!t0 || h[(size_t)i * 16] < t0)) t0 = h[(size_t)i * 16];
      fprintf(f, "# workgroup role | stamps in us from the first start: start, loads out, turn, start values, panels..., done(12)\n");
      for (int i = 0; i < grid; ++i) {
        fprintf(f, "%d %s", i, i < n_chains ? "chain" : "update");
        for (int q = 0; q < 13; ++q) fprintf(f, " %.2f", h[(size_t)i * 16 + q] ? (h[(size_t)i * 16 + q] - t0) / 100.0 : -1.0);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  return true;
}

void launch_chol_backsolve_chains(hipStream_t s, const double* S, const double* Lp, const double* Vinv, int ld,
                                  const int* bs_desc_dev, const int* chain_begin_dev,
                                  const int* chain_end_dev, int n_chains, const int* rows_flat_dev, double* y,
                                  int npad, int max_chain_len, const double* y_init, const int* iperm_dev, int n_pose, double* y_tan,
                                  double* delta, int max_rows, const double* Winv) {
  if (n_chains <= 0) return;
  size_t lds = chol_backsolve_chain_lds(npad, max_chain_len);
  // Winv (the tiles' full inverses, left by the fused factorisation): a panel's own solve is two 16-part reductions instead of a
  // substitution with 64 dependent pivots (bs_chain_walk)
  static const bool no_w = getenv("BSGPU_BACKSOLVE_NO_W") != nullptr;
  if (no_w) Winv = nullptr;
  const char* fg = getenv("BSGPU_BACKSOLVE_GLOBAL_Y");   // (tests: force the path windows above 12 288 reduced dimensions take)
  const int y_in_lds = (lds <= (size_t)160 * 1024 && !(fg && atoi(fg) != 0)) ? 1 : 0;
  if (!y_in_lds) lds = chol_backsolve_chain_lds(0, max_chain_len);
  // max_rows: the most row tiles any panel of these chains has (0: unknown).  Few enough: the two-panel-deep variant.
  const bool deep = max_rows > 0 && max_rows <= kBsChunkDeep && max_chain_len > 1;
#define BSG_LAUNCH_CHAIN_W(YL, CH, DEEP, W)                                                                                                \
  hipLaunchKernelGGL((chol_backsolve_chain_kernel<YL, CH, DEEP, W>), dim3(n_chains), dim3(1024), lds, s, S, Lp, W ? Winv : Vinv, ld, bs_desc_dev, \
                     chain_begin_dev, chain_end_dev, rows_flat_dev, y, npad, max_chain_len, n_chains == 1 ? y_init : nullptr, iperm_dev, \
                     n_pose, y_tan, delta)
#define BSG_LAUNCH_CHAIN(YL, CH, DEEP) do { if (Winv) BSG_LAUNCH_CHAIN_W(YL, CH, DEEP, true); else BSG_LAUNCH_CHAIN_W(YL, CH, DEEP, false); } while (0)
  if (y_in_lds) { if (deep) BSG_LAUNCH_CHAIN(true, kBsChunkDeep, true); else BSG_LAUNCH_CHAIN(true, kBsChunk, false); }
  else { if (deep) BSG_LAUNCH_CHAIN(false, kBsChunkDeep, true); else BSG_LAUNCH_CHAIN(false, kBsChunk, false); }
#undef BSG_LAUNCH_CHAIN_W
#undef BSG_LAUNCH_CHAIN
}

// Between two groups of chains (level-synchronous back-substitution, dense_plan.h): y_k -= sum_t L(t,k)^T y_t over the row tiles t
// of panel k that the group just solved.  One 1024-thread workgroup per target panel, the same thread layout as the walk (16 row
// groups x 64 columns), the tiles' entries loaded kBsChunk tiles at a time; nothing here depends on anything but finished y.
__global__ __launch_bounds__(1024) void chol_backsolve_update_kernel(const double* __restrict__ Lp, int ld, const int* __restrict__ items,
                                                                     const int* __restrict__ upd_rows, double* __restrict__ y) {
  __shared__ double sp[16 * NB];
  const int tid = threadIdx.x, c = tid & 63, part = tid >> 6;
  const int k = items[3 * blockIdx.x], off = items[3 * blockIdx.x + 1], n = items[3 * blockIdx.x + 2];
  const int c0 = k * NB;
  double acc = 0.0;
  for (int q0 = 0; q0 < n; q0 += kBsChunk) {
    double l[kBsChunk][4], yv[kBsChunk][4];
#pragma unroll
    for (int u = 0; u < kBsChunk; ++u) {
      const bool ok = q0 + u < n;
      const int r0 = (ok ? upd_rows[off + q0 + u] : k) * NB + 4 * part;
#pragma unroll
      for (int i = 0; i < 4; ++i) { l[u][i] = ok ? Lp[(size_t)(r0 + i) * ld + c0 + c] : 0.0; yv[u][i] = ok ? y[r0 + i] : 0.0; }
    }
#pragma unroll
    for (int u = 0; u < kBsChunk; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = fma(l[u][i], yv[u][i], acc);
  }
  sp[part * NB + c] = acc;
  __syncthreads();
  if (tid < NB) {
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += sp[q * NB + tid];
    y[c0 + tid] -= sum;
  }
}
void launch_chol_backsolve_update(hipStream_t s, const double* Lp, int ld, const int* items_dev, int n_items, const int* upd_rows_dev, double* y) {
  if (n_items <= 0) return;
  hipLaunchKernelGGL(chol_backsolve_update_kernel, dim3(n_items), dim3(1024), 0, s, Lp, ld, items_dev, upd_rows_dev, y);
}

// ---- the factorisation and the back-substitution of several windows in one launch each (bsgpu_batch.cpp): every window keeps its own
// task list, ticket and counters — a workgroup of window w takes window w's next ticket
void batchargs_chol_fused(BatchArgTable& t, double* S, double* Lp, int ld, const FusedTask* tasks_dev, int n_tasks, const int* tile_tot_dev, const int* nreal_dev,
                          double* Vinv, double* scal, int* sync_dev, double* Winv, int rhs_rows, const LmDiag& diag, const GradNormRide& gn, bool diag_tasks_in_list) {
  chol_fused_kernel_Args a;
  a.bsg_grid = n_tasks;
  a.S = S; a.Lp = Lp; a.ld = ld; a.tasks = tasks_dev; a.n_tasks = n_tasks; a.tile_tot = tile_tot_dev; a.nreal = nreal_dev; a.Vinv = Vinv; a.scal = scal;
  a.sync = sync_dev; a.Winv = Winv; a.fs = fused_sync_stride(); a.rhs_strips = (rhs_rows <= 0 ? 4 : std::min(4, (rhs_rows + 15) / 16)) | (fused_update_mode(diag_tasks_in_list) << 8); a.probe_ts = nullptr;
  a.diag = diag; a.gn = gn;
  t.push(a);
  t.lds = kFusedLds;
}
void launch_chol_fused_batch(hipStream_t s, const BatchArgTable& t, const BatchDyn* dyn, int list, int n) {
  if (n <= 0 || t.max_grid <= 0) return;
  hipLaunchKernelGGL((chol_fused_kernel_batch<false>), dim3(n, t.max_grid), dim3(kFusedThreads), kFusedLds, s, static_cast<const chol_fused_kernel_Args*>(t.dev), dyn, list);
}
size_t chol_backsolve_chain_lds(int npad, int max_chain_len);
// tabs[0..3]: single-launch form (shallow, deep), one-chain form (shallow, deep).  Returns the index of the table this window's
// back-substitution uses (the others get an empty entry), or -1 when neither form covers its plan (the caller takes the lone path).
int batchargs_backsolve(BatchArgTable* tabs, const DensePlan& P, const DenseDev& D, double* y, const int* iperm, int n_pose, double* y_tan, double* delta) {
  const int ld = P.npad;
  const double* rhs_row = D.Lp + (size_t)P.rhs_row * ld;
  const int G = (int)P.bs_group_off.size() - 1;
  const bool level_sync = P.bs_level_sync && D.bs_desc_chain && D.bs_upd;
  int max_len = 1, max_rows = 0;
  for (size_t i = 0; i < P.chain_begin.size(); ++i) max_len = std::max(max_len, P.chain_end[i] - P.chain_begin[i]);
  int form = -1;
  chol_backsolve_fused_kernel_Args f;
  chol_backsolve_chain_kernel_Args c;
  f.bsg_grid = 0; c.bsg_grid = 0;
  size_t lds = 0;
  if (level_sync && D.bs_items4 && D.bs_sync && D.scal && D.Winv && D.ftasks && D.fsync && D.bs_order && !getenv("BSGPU_BACKSOLVE_GLOBAL_Y")) {
    for (int g = 0; g < G; ++g) max_rows = std::max(max_rows, P.bs_group_maxrows[g]);
    max_rows = std::max(1, max_rows);
    const int n_chains = (int)P.chain_begin.size(), n_items = (int)P.bs_upd.size() / 3 * (P.bs_upd_off.back() > 0 ? 1 : 0);
    lds = chol_backsolve_chain_lds(P.npad, max_len) + 16;
    if (lds <= (size_t)160 * 1024 - 256 && n_chains + n_items <= 512) {
      const bool deep = max_rows <= kBsChunkDeep && max_len > 1;
      form = deep ? 1 : 0;
      f.bsg_grid = n_chains + n_items;
      f.Lp = D.Lp; f.Winv = D.Winv; f.ld = ld; f.bs_desc = D.bs_desc_chain; f.chain_begin = D.chain_begin; f.chain_end = D.chain_end; f.rows_flat = D.rows_flat_chain;
      f.n_chains = n_chains; f.chain_group = D.bs_chain_group; f.grp_nchains = D.bs_grp_nchains; f.grp_nitems = D.bs_grp_nitems; f.G = G; f.items = D.bs_items4;
      f.upd_rows = D.bs_upd_rows; f.tile_updated = D.bs_tile_updated; f.y = y; f.npad = P.npad; f.max_len = max_len; f.y_init = rhs_row; f.iperm = iperm; f.n_pose = n_pose;
      f.y_tan = y_tan; f.delta = delta; f.sync = D.bs_sync; f.scal = D.scal; f.ts = nullptr; f.order = D.bs_order; f.fs = fused_sync_stride();
    }
  } else if (!level_sync && G == 1 && P.chain_begin.size() == 1 && D.ftasks && D.fsync && D.tile_tot && D.Winv && !getenv("BSGPU_BACKSOLVE_GLOBAL_Y") &&
             !getenv("BSGPU_BACKSOLVE_NO_W")) {
    lds = chol_backsolve_chain_lds(P.npad, max_len);
    if (lds <= (size_t)160 * 1024) {
      const bool deep = false;   // (max_rows unknown for the plain row lists: launch_chol_backsolve_chains passes 0)
      form = 2 + (deep ? 1 : 0);
      c.bsg_grid = 1;
      c.S = nullptr; c.Lp = D.Lp; c.Vinv = D.Winv; c.ld = ld; c.bs_desc = D.bs_desc; c.chain_begin = D.chain_begin; c.chain_end = D.chain_end; c.rows_flat = D.rows_flat;
      c.y = y; c.npad = P.npad; c.max_len = max_len; c.y_init = rhs_row; c.iperm = iperm; c.n_pose = n_pose; c.y_tan = y_tan; c.delta = delta;
    }
  }
  for (int q = 0; q < 2; ++q) { chol_backsolve_fused_kernel_Args e = f; if (q != form) e.bsg_grid = 0; tabs[q].push(e); if (q == form) tabs[q].lds = std::max(tabs[q].lds, lds); }
  for (int q = 2; q < 4; ++q) { chol_backsolve_chain_kernel_Args e = c; if (q != form) e.bsg_grid = 0; tabs[q].push(e); if (q == form) tabs[q].lds = std::max(tabs[q].lds, lds); }
  return form;
}
void launch_backsolve_batch(hipStream_t s, const BatchArgTable* tabs, const BatchDyn* dyn, const int* n_in_form /* 4 */) {
  if (n_in_form[0] > 0) hipLaunchKernelGGL((chol_backsolve_fused_kernel_batch<kBsChunk, false>), dim3(n_in_form[0], tabs[0].max_grid), dim3(1024), tabs[0].lds, s, static_cast<const chol_backsolve_fused_kernel_Args*>(tabs[0].dev), dyn, BL_BS_FUSED);
  if (n_in_form[1] > 0) hipLaunchKernelGGL((chol_backsolve_fused_kernel_batch<kBsChunkDeep, true>), dim3(n_in_form[1], tabs[1].max_grid), dim3(1024), tabs[1].lds, s, static_cast<const chol_backsolve_fused_kernel_Args*>(tabs[1].dev), dyn, BL_BS_FUSED + 1);
  if (n_in_form[2] > 0) hipLaunchKernelGGL((chol_backsolve_chain_kernel_batch<true, kBsChunk, false, true>), dim3(n_in_form[2], tabs[2].max_grid), dim3(1024), tabs[2].lds, s, static_cast<const chol_backsolve_chain_kernel_Args*>(tabs[2].dev), dyn, BL_BS_CHAIN);
  if (n_in_form[3] > 0) hipLaunchKernelGGL((chol_backsolve_chain_kernel_batch<true, kBsChunkDeep, true, true>), dim3(n_in_form[3], tabs[3].max_grid), dim3(1024), tabs[3].lds, s, static_cast<const chol_backsolve_chain_kernel_Args*>(tabs[3].dev), dyn, BL_BS_CHAIN + 1);
}
size_t chol_backsolve_chain_lds(int npad, int max_chain_len) {
  return sizeof(double) * (NB * (NB + 1) + 16 * NB + 1024 + (size_t)npad) + sizeof(int) * (size_t)max_chain_len * (3 + kBsMaxRows);
}
int chol_vinv_stride() { return kVinvStride; }
void chol_prepare() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_panel_step_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPanelStepLds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_panel_step_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPanelStepLds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<true, kBsChunkDeep, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<true, kBsChunkDeep, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<false, kBsChunkDeep, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<false, kBsChunkDeep, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<true, kBsChunk, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<true, kBsChunk, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<false, kBsChunk, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_chain_kernel<false, kBsChunk, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_fused_kernel<kBsChunkDeep, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_backsolve_fused_kernel<kBsChunk, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds);
}

}  // namespace bsg
