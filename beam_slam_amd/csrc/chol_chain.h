// Dense FP64 Cholesky of a CHAIN of up to MAXM consecutive 64x64 tiles of the reduced camera system inside ONE 512-thread
// workgroup (k_chol.hip chol_fused_kernel, task kind kFusedChain; dense_plan.h says what a chain is: a piece or a separator of the
// nested-dissection order, whose tiles are dense among themselves).
//
// Why: the tiled factorisation's critical path is the chain potrf(k) -> solve -> update -> potrf(k+1) ...; as one workgroup-sized
// task per tile each link cost a cross-workgroup hand-over, a reload of L_kk, a 64-wide solve, a rank-64 product and a 64x64 potrf
// on one wave (17.7 us).  Here the whole chain lives in ONE workgroup and is factored right-looking in 16-column
// steps:
//   * the trailing matrix stays in REGISTERS — 16x16 blocks in the v_mfma_f64_16x16x4 result layout, dealt out to the eight waves
//     column by column (block (r, c) of the enumeration goes to wave idx % 8, slot idx / 8) — and is only ever touched by MFMA;
//   * the current 16-column panel goes through LDS (rows x 16, pitch 18): the owners of its blocks store them, the ELIMINATION
//     takes it with one row per lane: every row of 16 lanes holds a replica of the 16x16 diagonal block next to 64 sub-diagonal
//     rows, so the pivot row travels by DPP row_newbcast (v_mov_b64_dpp / v_fmac_f64_dpp) instead of two v_readlane per value,
//     and the next pivot d' = y - x^2 / d is formed from broadcast scalars while the column updates of the current one issue —
//     the dependent chain per pivot is rsq + Newton + one FMA, the rest is throughput;
//   * identity rows ride through the elimination of a tile as extra sub-diagonal rows: X = I L_kk^-T, i.e. the tile's full inverse
//     W = L_kk^-1 (what the back-substitution multiplies by, and whose diagonal 16x16 blocks are the V_b of the tasks' triangular
//     solves) costs a few more rows instead of a dependent MFMA chain after the factorisation;
//   * after each step the finished panel leaves for Lp (rows of the chain) and Winv (identity rows, transposed) with write-through
//     stores; the flag of a tile is set one step later, when the stores have drained behind the next step's work.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace bsg {
namespace chain {

constexpr int PP = 18;   // doubles between two rows of the LDS panel: conflict-free ds_read_b64 of the MFMA operands (16 rows x 4 k)
typedef double double4_c __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_c __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_c __attribute__((ext_vector_type(2)));

#define BSG_CHAIN_DEV __device__ __forceinline__

// lane N of every row of 16 lanes, to the whole row (DP-ALU DPP control row_newbcast; gfx90a and later)
template <int N> BSG_CHAIN_DEV double bcast16(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xf, 0xf, true); }

BSG_CHAIN_DEV double rsqrt_fast(double d) {   // v_rsq_f64 seed (~2^-26) + one cubically convergent step
  const double y = __builtin_amdgcn_rsq(d);
  const double t = fma(-d * y, y, 1.0);
  return fma(y * t, fma(t, 0.375, 0.5), y);
}

// acc += -l * bcast_C(ld) for the diagonal replica and the sub-diagonal row, columns C = J+1 .. 15
#ifndef BSG_CHAIN_ASM_DPP
#define BSG_CHAIN_ASM_DPP 1
#endif
template <int J, int C>
BSG_CHAIN_DEV void upd_cols(double (&ad)[16], double (&ab)[16], double ld, double nld, double nlb) {
  if constexpr (C < 16) {
#if BSG_CHAIN_ASM_DPP
    // one instruction per value: dst += bcast_C(ld) * (-l).  The hazard VALU write -> DPP read of ld needs two wait states: the
    // s_nop rides in the first statement of a pivot (inline asm is opaque to the compiler's hazard recogniser)
    if constexpr (C == J + 1)
      asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(ad[C]) : "v"(ld), "v"(nld), "n"(C));
    else
      asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(ad[C]) : "v"(ld), "v"(nld), "n"(C));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(ab[C]) : "v"(ld), "v"(nlb), "n"(C));
#else
    const double s = bcast16<C>(ld);
    ad[C] = fma(nld, s, ad[C]);
    ab[C] = fma(nlb, s, ab[C]);
#endif
    upd_cols<J, C + 1>(ad, ab, ld, nld, nlb);
  }
}

// pivots J .. 15 of a 16-column panel: ad = row (lane & 15) of the diagonal block (one replica per row of 16 lanes), ab = this
// lane's sub-diagonal row; d = the pivot (uniform).  Leaves the scaled columns (the factor) in ad / ab; returns 1 / L_15,15.
template <int J>
BSG_CHAIN_DEV double elim_pivots(double (&ad)[16], double (&ab)[16], double d) {
  if constexpr (J < 15) {
    // what the NEXT pivot needs of the unscaled data goes out before this pivot's reciprocal square root is known
    const double x = bcast16<J + 1>(ad[J]), y = bcast16<J + 1>(ad[J + 1]);
    const double rs = rsqrt_fast(d);
    const double ld = ad[J] * rs, lb = ab[J] * rs;
    const double dn = fma(-(x * x), rs * rs, y);
    upd_cols<J, J + 1>(ad, ab, ld, -ld, -lb);
    ad[J] = ld; ab[J] = lb;
    return elim_pivots<J + 1>(ad, ab, dn);
  } else {
    const double rs = rsqrt_fast(d);
    ad[J] = ad[J] * rs; ab[J] = ab[J] * rs;
    return rs;
  }
}

// workgroup barrier that orders LDS traffic only: global loads / stores issued before it stay in flight across it (__syncthreads()
// waits for them: the chain's tile loads and the panels' stores would be paid at every step)
BSG_CHAIN_DEV void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

BSG_CHAIN_DEV double2 lds_ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }

BSG_CHAIN_DEV void st16_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double2 d) {
  u32x4_c v; __builtin_memcpy(&v, &d, 16);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16 /* sc1 */);
}
BSG_CHAIN_DEV double ld8_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x2_c v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16);
  double d; __builtin_memcpy(&d, &v, 8); return d;
}

struct ChainArgs {
  const double* S;    // the assembled reduced system (tiles read with sc1 loads: other workgroups of this launch have updated them)
  double* Lp;         // the factor (shadow matrix), write-through
  double* Winv;       // per tile: L_kk^-1, row-major 64 x 64, write-through
  int ld;             // row pitch of S / Lp in doubles
  int c0, m;          // first tile and number of tiles of the chain
  unsigned present;   // bit i (i + 1) / 2 + j: tile (c0 + i, c0 + j) of the chain is structurally non-zero (else it is zero and stays zero)
  const int* nreal;   // per tile: number of real columns (the window's last tile may be partial; the others: unit pivots)
  int* tile_flag;     // potrf_done words of the chain's tiles: tile_flag[(c0 + k) * flag_stride] = 1 once column tile k is out
  int flag_stride;
  double* Vinv;       // per tile (vinv_stride doubles): the inverses of the four diagonal 16x16 blocks of its factor + 64 reciprocal
  int vinv_stride;    // pivots — read by later launches only (plain stores); null: not wanted
};

// 512 threads, three kinds of waves, so that the hardware interleaves the pivot chain (VALU latency) of panel b with the rank-16
// updates (MFMA) of panel b-1 — different waves on the same SIMDs:
//   waves 0 .. 2   ELIMINATE: 4 row blocks of 16 lanes each = the 12 row blocks a three-tile chain (+ identity rows) has at most
//   waves 4 .. 7   own the trailing matrix (one per SIMD) and UPDATE it; their code is fully unrolled over the steps (template on the
//                  chain length M): which slot is touched when is static, only the ROW a slot stands for depends on the wave —
//                  column block c's rows c + ((u + c) & 3) + 4 i go to wave u, slot (c, i) — so every LDS address is one of a few
//                  per-lane bases plus an immediate, there is no bookkeeping, and the compiler pipelines the operand fetches
//                  under the MFMAs by itself
//   wave 3         the identity rows' blocks, and every store to global memory (panel b-1 -> Lp / Winv while panel b is eliminated)
// A step b (one 16-column panel) is two phases, a barrier after each:
//   A  the blocks of column b receive the update of panel b-1 and go to panel buffer b % 2 (with the identity rows that enter);
//   B  panel b is eliminated in place while every other live block receives the update of panel b-1 from the other buffer.
// Returns whether a pivot was not positive / finite (every thread).  ts (PROBE): shader-clock stamps, 2 per step + 3.
constexpr int kChainMaxTiles = 3;
constexpr int kChainPanelRows = 16 * (4 * kChainMaxTiles + 4);   // rows of a panel buffer: the chain's rows, then the 64 identity rows
// (slots whose row does not exist read past their panel buffer: the allocation covers that)
constexpr int chain_lds_doubles() { return 2 * kChainPanelRows * PP + 16 * 12 * PP; }

BSG_CHAIN_DEV double4_c chain_mfma4(double4_c acc, const double* pa, const double* pb) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk], pb[4 * kk], acc, 0, 0, 0);
  return acc;
}

// ---- the updating waves' static schedule
template <int NCB> struct ChainSlots {
  static constexpr int K(int c) { return (NCB - c + 3) / 4; }
  static constexpr int first(int c) { int s = 0; for (int i = 0; i < c; ++i) s += K(i); return s; }
  static constexpr int total = first(NCB);
  static constexpr int rem(int c) { return NCB - c - 4 * (K(c) - 1); }   // wave offsets (u + c) & 3 < rem(c) have a row in the last slot of column c
};
template <int NCB>
struct ChainUpdater {
  using SL = ChainSlots<NCB>;
  double4_c acc[SL::total];
  int rowsel[4];        // (u + j) & 3
  int offA[2][4];       // per panel buffer, per j: doubles to row block rowsel[j], this lane's operand position
  int offB[2];          // ... to row block 0
  int offS[2][4];       // ... staging position (result layout) of row block rowsel[j]
  double* P0;
  int mask_lo, mask_nr; // the partial tile's first row block (or < 0) and its number of real columns
  int nb_real;          // column blocks that hold real columns (chain_factor): the steps of the others are not run
  template <int C, int I> BSG_CHAIN_DEV bool valid() const { return (I < SL::K(C) - 1) || (SL::rem(C) == 4) || (rowsel[C & 3] < SL::rem(C)); }
  // slot (C, I) takes the update of panel PB (in buffer PB & 1)
  template <int PB, int C, int I> BSG_CHAIN_DEV void update() {
    // (blocks whose row or column is padding are zero and stay zero: no products for them — a 144-dimensional piece in three tiles
    // has 45 live blocks of 78)
    if (valid<C, I>() && C < nb_real && C + rowsel[C & 3] + 4 * I < nb_real) {
      const double* pa = P0 + offA[PB & 1][C & 3] + 16 * (C + 4 * I) * PP;
      const double* pb = P0 + offB[PB & 1] + 16 * C * PP;
      acc[SL::first(C) + I] = chain_mfma4(acc[SL::first(C) + I], pa, pb);
    }
  }
  template <int PB, int C, int I> BSG_CHAIN_DEV void update_col() {
    if constexpr (I < SL::K(C)) { update<PB, C, I>(); update_col<PB, C, I + 1>(); }
  }
  template <int PB, int C> BSG_CHAIN_DEV void update_from() {   // columns C .. NCB-1
    if constexpr (C < NCB) { update_col<PB, C, 0>(); update_from<PB, C + 1>(); }
  }
  template <int B, int I> BSG_CHAIN_DEV void stage(int q, int n) {
    if constexpr (I < SL::K(B)) {
      if (valid<B, I>()) {
        double4_c v = acc[SL::first(B) + I];
        const int r = B + rowsel[B & 3] + 4 * I;
        if (I == 0 && r == B && r >= mask_lo && r < mask_lo + 4) {   // a diagonal block of the partial tile: columns >= nreal are unit pivots with nothing below
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * (B & 3) + q + 4 * reg, col = 16 * (B & 3) + n;
            if (col >= mask_nr && col <= row) v[reg] = (row == col) ? 1.0 : 0.0;
          }
        }
        double* ps = P0 + offS[B & 1][B & 3] + 16 * (B + 4 * I) * PP;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ps[4 * reg * PP] = v[reg];
      }
      stage<B, I + 1>(q, n);
    }
  }
  template <int B, class F> BSG_CHAIN_DEV void step(int q, int n, F& load_rest) {
    if constexpr (B < NCB) {
      if (B >= nb_real) return;                          // (uniform: the chain's last tile ends in padding — every kind of wave stops here)
      if constexpr (B > 0) update_col<B - 1, B, 0>();   // phase A: column B takes its last update ...
      stage<B, 0>(q, n);                                 // ... and is staged
      lds_sync();
      if constexpr (B == 0) load_rest();                 // (the rest of the chain's tiles travel under the first elimination)
      if constexpr (B > 0) update_from<B - 1, B + 1>(); // phase B
      lds_sync();
      step<B + 1>(q, n, load_rest);
    }
  }
};

template <int NCB>
BSG_CHAIN_DEV void chain_update_waves(const ChainArgs& A, double* smem, int u, int q, int n, int c0, int ld, __amdgpu_buffer_rsrc_t rS, int nb_real) {
  using SL = ChainSlots<NCB>;
  ChainUpdater<NCB> U;
  U.nb_real = nb_real;
  constexpr int PROWS = kChainPanelRows;
  U.P0 = smem;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    U.rowsel[j] = (u + j) & 3;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      U.offA[pb][j] = pb * PROWS * PP + (16 * U.rowsel[j] + n) * PP + q;
      U.offS[pb][j] = pb * PROWS * PP + (16 * U.rowsel[j] + q) * PP + n;
    }
  }
  U.offB[0] = n * PP + q; U.offB[1] = PROWS * PP + n * PP + q;
  U.mask_lo = -1; U.mask_nr = 64;
  for (int t = 0; t < NCB / 4; ++t) { const int nr = A.nreal[c0 + t]; if (nr < 64) { U.mask_lo = 4 * t; U.mask_nr = nr; } }
  U.mask_lo = __builtin_amdgcn_readfirstlane(U.mask_lo); U.mask_nr = __builtin_amdgcn_readfirstlane(U.mask_nr);
  // the chain's tiles into registers: every load goes out now, column block 0 first
  auto load_slot = [&](auto Cc, auto Ic) {
    constexpr int C = decltype(Cc)::value, I = decltype(Ic)::value;
    double4_c v = {0.0, 0.0, 0.0, 0.0};
    const int r = C + U.rowsel[C & 3] + 4 * I;
    if (r < NCB) {
      const int ti = r >> 2, tj = C >> 2;
      if ((A.present >> (ti * (ti + 1) / 2 + tj)) & 1u) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          v[reg] = ld8_wt(rS, (unsigned)(((size_t)(16 * r + q + 4 * reg) * ld + c0 * 64 + 16 * C + n) * sizeof(double)));
      }
    }
    U.acc[SL::first(C) + I] = v;
  };
  auto load_all = [&](auto self, auto Cc, auto Ic, auto Ce) {   // columns C .. CEND-1
    constexpr int C = decltype(Cc)::value, I = decltype(Ic)::value, CEND = decltype(Ce)::value;
    if constexpr (C < CEND) {
      if constexpr (I < SL::K(C)) { load_slot(Cc, Ic); self(self, Cc, std::integral_constant<int, I + 1>{}, Ce); }
      else self(self, std::integral_constant<int, C + 1>{}, std::integral_constant<int, 0>{}, Ce);
    }
  };
  // column block 0 first: the first elimination only needs that one
  load_all(load_all, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  auto load_rest = [&]() { load_all(load_all, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NCB>{}); };
  U.template step<0>(q, n, load_rest);
}

template <bool PROBE>
BSG_CHAIN_DEV bool chain_factor(const ChainArgs& A, double* smem, long long* ts) {
  constexpr int NW = 8, NE = 3;
  constexpr int PROWS = kChainPanelRows;
  double* const Pbuf0 = smem;
  double* const Pbuf1 = smem + PROWS * PP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int m = __builtin_amdgcn_readfirstlane(A.m), c0 = __builtin_amdgcn_readfirstlane(A.c0), ld = __builtin_amdgcn_readfirstlane(A.ld);
  const int NCB = 4 * m;
  // A chain whose LAST tile is partial (a supernode of the per-dimension order is padded to whole tiles at its end; the window's last
  // tile in the tile-level order): the 16-column blocks made of padding only are unit pivots with nothing below — their steps are
  // not run (an 81-dimensional separator is six steps, not eight), their part of the factor is written as constants at the end.
  const int nr_last = __builtin_amdgcn_readfirstlane(A.nreal[c0 + m - 1]);
  const int nb_real = nr_last < 64 ? 4 * (m - 1) + (nr_last + 15) / 16 : NCB;
  // (the resources start at the chain's first row: 32-bit sizes and offsets, and the matrix passes 4 GB at 23 170 dimensions)
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.S) + (size_t)c0 * 64 * ld, 0, (int)((size_t)kChainMaxTiles * 64 * ld * sizeof(double)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(A.Lp + (size_t)c0 * 64 * ld, 0, (int)((size_t)kChainMaxTiles * 64 * ld * sizeof(double)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(A.Winv + (size_t)c0 * 4096, 0, (int)((size_t)m * 4096 * sizeof(double)), 0x00020000);
  int nts = 0;
  auto stamp = [&]() { if (PROBE && tid == 0) ts[nts++] = (long long)__builtin_readcyclecounter(); };
  stamp();
  bool bad = false;
  // panel p (finished, in LDS) to Lp / Winv
  auto panel_out = [&](int p, const double* P, int t0, int nthreads) {
    const int k = p >> 2, pq = p & 3;
    const int total = (NCB - p) * 128;
    for (int i = t0; i < total; i += nthreads) {
      const int prow = 16 * p + (i >> 3), piece = i & 7;
      st16_wt(rL, (unsigned)(((size_t)prow * ld + c0 * 64 + 16 * p + 2 * piece) * sizeof(double)), lds_ld2(&P[prow * PP + 2 * piece]));
    }
    // W[16 pq + c][j] = X[j][16 pq + c]: 16 rows x 64 columns of Winv[k], two columns per thread
    for (int i = t0; i < 512; i += nthreads) {
      const int c = i >> 5, j2 = (i & 31) * 2;
      double2 w = {0.0, 0.0};
      if ((j2 >> 4) <= pq) { w.x = P[(16 * NCB + j2) * PP + c]; w.y = P[(16 * NCB + j2 + 1) * PP + c]; }
      st16_wt(rW, (unsigned)(((size_t)k * 4096 + (16 * pq + c) * 64 + j2) * sizeof(double)), w);
    }
    if (A.Vinv) {   // V_pq = (L_pq,pq)^-1 = the diagonal block of W: V[i][c] = X[16 pq + c][16 pq + i]
      double* V = A.Vinv + (size_t)(c0 + k) * A.vinv_stride;
      for (int i = t0; i < 256; i += nthreads) V[pq * 256 + i] = P[(16 * (NCB + pq) + (i & 15)) * PP + (i >> 4)];
      for (int i = t0; i < 16; i += nthreads) V[1024 + 16 * pq + i] = P[(16 * (NCB + pq) + i) * PP + i];
    }
  };
  // The kinds of waves run different loops (the accumulators of the one and the rows of the other never share a register
  // allocation); all pass the same two barriers per step.
  if (wave < NE) {
    // =============================== eliminating waves ===============================
    stamp();
#pragma unroll 1
    for (int b = 0; b < nb_real; ++b) {
      const int k = b >> 2, bq = b & 3;
      double* const P = (b & 1) ? Pbuf1 : Pbuf0;   // panel b
      lds_sync();
      stamp();
      // phase B: row blocks b+1 .. nb_real-1 of the chain (rows of padding are zero and stay zero), then the identity row blocks 0 .. bq; four per wave
      const int nc = nb_real - 1 - b, nrb = nc + bq + 1;
      if (4 * wave < nrb) {
        const int j = 4 * wave + q;
        const bool active = j < nrb;
        const int prow = (!active ? 16 * b : (j < nc ? 16 * (b + 1 + j) : 16 * (NCB + (j - nc)))) + n;   // (idle lanes shadow the diagonal rows)
        double ad[16], ab[16];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          const double2 v = lds_ld2(&P[(16 * b + n) * PP + 2 * c2]);
          ad[2 * c2] = v.x; ad[2 * c2 + 1] = v.y;
        }
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          const double2 v = lds_ld2(&P[prow * PP + 2 * c2]);
          ab[2 * c2] = v.x; ab[2 * c2 + 1] = v.y;
        }
        const double d0 = bcast16<0>(ad[0]);
        const double rs_last = elim_pivots<0>(ad, ab, d0);
        if (!(rs_last > 0.0) || !(rs_last < 1.7e308)) bad = true;   // (a bad pivot anywhere before turns everything after it into NaN)
        if (active) {
#pragma unroll
          for (int c2 = 0; c2 < 8; ++c2) *reinterpret_cast<double2*>(&P[prow * PP + 2 * c2]) = double2{ab[2 * c2], ab[2 * c2 + 1]};
        }
        if (wave == 0 && q == 0) {
#pragma unroll
          for (int c2 = 0; c2 < 8; ++c2)
            *reinterpret_cast<double2*>(&P[(16 * b + n) * PP + 2 * c2]) = double2{(2 * c2 <= n) ? ad[2 * c2] : 0.0, (2 * c2 + 1 <= n) ? ad[2 * c2 + 1] : 0.0};
        }
      }
      lds_sync();
      if (bq == 0 && b > 0 && tid == 0)   // (wave 3 drained the previous tile's stores before this barrier)
        __hip_atomic_store(&A.tile_flag[(size_t)(c0 + k - 1) * A.flag_stride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stamp();
    }
  } else if (wave == NE) {
    // =============================== the identity rows and the stores ===============================
    // blocks (wq, wc), wq < wc, of the identity rows' trailing part inside the current tile: w01 w02 w03 w12 w13 w23
    double4_c w01, w02, w03, w12, w13, w23;
#pragma unroll 1
    for (int k = 0; k < m; ++k) {
      w01 = w02 = w03 = w12 = w13 = w23 = double4_c{0.0, 0.0, 0.0, 0.0};
      auto sub = [&](auto BQc) {
        constexpr int BQ = decltype(BQc)::value;
        const int b = 4 * k + BQ;
        if (b >= nb_real) return;
        double* const P = (BQ & 1) ? Pbuf1 : Pbuf0;          // panel b  (4 k is even)
        const double* const Pp = (BQ & 1) ? Pbuf0 : Pbuf1;    // panel b - 1
        auto wupd = [&](double4_c& acc, int wq, int wc) {
          return chain_mfma4(acc, Pp + (16 * (NCB + wq) + n) * PP + q, Pp + (16 * (4 * k + wc) + n) * PP + q);
        };
        auto wstage = [&](const double4_c& v, int wq) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) P[(16 * (NCB + wq) + q + 4 * reg) * PP + n] = v[reg];
        };
        // phase A: the identity rows that enter with this column block; the blocks of column BQ take their last update (panel b-1)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) P[(16 * (NCB + BQ) + q + 4 * reg) * PP + n] = (q + 4 * reg == n) ? 1.0 : 0.0;
        if constexpr (BQ == 1) { w01 = wupd(w01, 0, 1); wstage(w01, 0); }
        if constexpr (BQ == 2) { w02 = wupd(w02, 0, 2); w12 = wupd(w12, 1, 2); wstage(w02, 0); wstage(w12, 1); }
        if constexpr (BQ == 3) { w03 = wupd(w03, 0, 3); w13 = wupd(w13, 1, 3); w23 = wupd(w23, 2, 3); wstage(w03, 0); wstage(w13, 1); wstage(w23, 2); }
        lds_sync();
        // phase B: the live blocks take the update of panel b-1 (same tile); panel b-1 leaves
        if constexpr (BQ == 1) { w02 = wupd(w02, 0, 2); w03 = wupd(w03, 0, 3); }
        if constexpr (BQ == 2) { w03 = wupd(w03, 0, 3); w13 = wupd(w13, 1, 3); }
        if (b > 0) panel_out(b - 1, Pp, lane, 64);
        if (BQ == 0 && b > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile is out: the flag follows the barrier
        lds_sync();
      };
      sub(std::integral_constant<int, 0>{}); sub(std::integral_constant<int, 1>{}); sub(std::integral_constant<int, 2>{}); sub(std::integral_constant<int, 3>{});
    }
  } else {
    // =============================== updating waves ===============================
    const int u = wave - 4;
    if (m == 1) chain_update_waves<4>(A, smem, u, q, n, c0, ld, rS, nb_real);
    else if (m == 2) chain_update_waves<8>(A, smem, u, q, n, c0, ld, rS, nb_real);
    else chain_update_waves<12>(A, smem, u, q, n, c0, ld, rS, nb_real);
  }
  panel_out(nb_real - 1, ((nb_real - 1) & 1) ? Pbuf1 : Pbuf0, tid, 64 * NW);
  // the blocks of padding: L = I on their diagonal, zero below; W = I; reciprocal pivots 1
  for (int p = nb_real; p < NCB; ++p) {
    const int k = p >> 2, pq = p & 3;
    for (int i = tid; i < (NCB - p) * 128; i += 64 * NW) {
      const int prow = 16 * p + (i >> 3), piece = i & 7;
      const int cc = 16 * p + 2 * piece;
      st16_wt(rL, (unsigned)(((size_t)prow * ld + c0 * 64 + cc) * sizeof(double)), double2{prow == cc ? 1.0 : 0.0, prow == cc + 1 ? 1.0 : 0.0});
    }
    for (int i = tid; i < 512; i += 64 * NW) {
      const int c = i >> 5, j2 = (i & 31) * 2;
      st16_wt(rW, (unsigned)(((size_t)k * 4096 + (16 * pq + c) * 64 + j2) * sizeof(double)), double2{16 * pq + c == j2 ? 1.0 : 0.0, 16 * pq + c == j2 + 1 ? 1.0 : 0.0});
    }
    if (A.Vinv) {
      double* V = A.Vinv + (size_t)(c0 + k) * A.vinv_stride;
      for (int i = tid; i < 256; i += 64 * NW) V[pq * 256 + i] = ((i & 15) == (i >> 4)) ? 1.0 : 0.0;
      for (int i = tid; i < 16; i += 64 * NW) V[1024 + 16 * pq + i] = 1.0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
  if (tid == 0) __hip_atomic_store(&A.tile_flag[(size_t)(c0 + m - 1) * A.flag_stride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  stamp();
  return any_bad;
}

}  // namespace chain
}  // namespace bsg
