// Block-sparse (BSR, 3x3 blocks) normal equations + block-Jacobi preconditioned conjugate gradients for
// pose-only problems whose reduced system is too large for the dense exact path (BASELINE config 4:
// 5 000-pose / 50 000-constraint global-mapper pose graph, 30 000 tangent dims — the shape of
// bs_models/src/lib/global_mapping/submap_pose_graph_optimization.cpp:22-150).  The reference solves it
// exactly ([EXT] Ceres SPARSE_NORMAL_CHOLESKY); this path is an inexact Newton step whose inner
// tolerance is tight enough (default 1e-10 relative) for the LM trajectory to agree to the stated
// final-cost tolerance.  Every tangent block of the pose-only factor types has size 3, hence 3x3 blocks.
//
//   bsr_assemble        wave / factor: the J^T J blocks only that factor writes, plain stores (slots precomputed)
//   bsr_assemble_seg    the shared blocks (all diagonal ones) by segments of <= 64 contributions, with gradient and diag(J^T J)
//   bsr_finish_diag     LM diagonal (Jacobi scaling folded in, as in the dense path) + inverses of the 6x6 diagonal blocks of
//                       consecutive block-row pairs (position + orientation of a pose)
//   pcg_spmv            beta, stop test and p = z + beta p folded in; q = A p (one wave / block row), partials of p.q
//   pcg_update          alpha from the partials; x += a p; r -= a q; z = M^-1 r; partials of r.z, r.r
// HBM-bound: one PCG iteration streams the BSR values once (C4: 30 MB) plus six n-vectors.
#include <atomic>

#include "bsgpu_device.h"

namespace bsg {

// device scalars of the PCG recurrence
enum { PC_RZ = 0, PC_RR = 1, PC_RR0 = 2, PC_DONE = 3, PC_ITERS = 4, PC_PQ = 5, PC_NUM = 8 };

// Blocks that exactly ONE (factor, slot a, slot b) writes — the off-diagonal blocks of a pose graph, nearly all of them — are stored by
// that factor's wave, without atomics (slots >= 0); the others (every diagonal block: a pose's ~20 constraints add into it) are summed by
// segments, bsr_assemble_seg_kernel (their entries of `slots` are -1 here).  Factor-wise atomics for everything were 7.2 M FP64 atomics
// per LM step on the 5 000-pose graph (95 us); now 0.3 M.
__global__ __launch_bounds__(64) void bsr_assemble_kernel(SmallGroup g, const int* __restrict__ slots, double* __restrict__ val) {
  __shared__ double sJ[15 * 30];
  const int f = blockIdx.x, lane = threadIdx.x;
  if (!g.active[f]) return;
  const int m = g.m, nv = g.nv, tw = 3 * nv;
  const double* J = g.J + (size_t)f * m * tw;
  for (int i = lane; i < m * tw; i += 64) sJ[i] = J[i];
  __syncthreads();
  const int* sl = slots + (size_t)f * nv * nv;
  for (int p = lane; p < tw * tw; p += 64) {
    const int a = p / tw, b = p % tw;
    const int slot = sl[(a / 3) * nv + (b / 3)];
    if (slot < 0) continue;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += sJ[k * tw + a] * sJ[k * tw + b];
    val[(size_t)slot * 9 + (a % 3) * 3 + (b % 3)] = acc;
  }
}
// the shared blocks: sixteen lanes per segment (the contributions (type, factor, slot a, slot b) to one block, at most 64 of them); a
// diagonal block's segment also sums the gradient J^T r and diag(J^T J) of its three rows
BSG_DEV double bsr_sum16(double v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void bsr_assemble_seg_kernel(const SmallGroup* __restrict__ groups, int n_seg, const int* __restrict__ seg_start,
                                                              const int* __restrict__ seg_slot, const int* __restrict__ seg_row,
                                                              const int2* __restrict__ contrib, double* __restrict__ val, double* __restrict__ rhs,
                                                              double* __restrict__ grad, double* __restrict__ hdiag) {
  const int seg = (int)blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  if (seg >= n_seg) return;
  const int beg = seg_start[seg], end = seg_start[seg + 1];
  const int row = seg_row[seg];                       // tangent row of a diagonal block, -1 otherwise
  double acc[9], gs[3], hs[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) { gs[i] = 0.0; hs[i] = 0.0; }
  for (int e = beg + lane; e < end; e += 16) {
    const int2 cb = contrib[e];
    const int t = cb.x >> 24, f = cb.x & ((1 << 24) - 1), sa = cb.y >> 8, sb = cb.y & 255;
    const SmallGroup& g = groups[t];
    const int m = g.m, tw = 3 * g.nv;
    const double* J = g.J + (size_t)f * m * tw;
    const double* r = g.r + (size_t)f * m;
    const bool dg = row >= 0 && sa == sb;
    for (int k = 0; k < m; ++k) {
      const double a0 = J[k * tw + 3 * sa], a1 = J[k * tw + 3 * sa + 1], a2 = J[k * tw + 3 * sa + 2];
      const double b0 = J[k * tw + 3 * sb], b1 = J[k * tw + 3 * sb + 1], b2 = J[k * tw + 3 * sb + 2];
      acc[0] += a0 * b0; acc[1] += a0 * b1; acc[2] += a0 * b2;
      acc[3] += a1 * b0; acc[4] += a1 * b1; acc[5] += a1 * b2;
      acc[6] += a2 * b0; acc[7] += a2 * b1; acc[8] += a2 * b2;
      if (dg) {
        const double rk = r[k];
        gs[0] += a0 * rk; gs[1] += a1 * rk; gs[2] += a2 * rk;
        hs[0] += a0 * a0; hs[1] += a1 * a1; hs[2] += a2 * a2;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = bsr_sum16(acc[i]);
  if (row >= 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { gs[i] = bsr_sum16(gs[i]); hs[i] = bsr_sum16(hs[i]); }
  }
  const int slot = seg_slot[seg];
  if (lane < 9) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) v = (lane == i) ? acc[i] : v;
    atomicAdd(&val[(size_t)slot * 9 + lane], v);     // (a block of more than 64 contributions has several segments)
  } else if (row >= 0 && lane < 12) {
    const int i = lane - 9;
    double g0 = 0.0, h0 = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { g0 = (i == q) ? gs[q] : g0; h0 = (i == q) ? hs[q] : h0; }
    atomicAdd(&rhs[row + i], g0);
    atomicAdd(&grad[row + i], g0);
    atomicAdd(&hdiag[row + i], h0);
  }
}
void launch_bsr_assemble(hipStream_t s, const SmallGroup& g, const int* slots, double* val) {
  if (g.n == 0) return;
  hipLaunchKernelGGL(bsr_assemble_kernel, dim3(g.n), dim3(64), 0, s, g, slots, val);
}
void launch_bsr_assemble_seg(hipStream_t s, const SmallGroup* groups_dev, int n_seg, const int* seg_start, const int* seg_slot, const int* seg_row,
                             const int2* contrib, double* val, double* rhs, double* grad, double* hdiag) {
  if (n_seg <= 0) return;
  hipLaunchKernelGGL(bsr_assemble_seg_kernel, dim3((n_seg + 15) / 16), dim3(256), 0, s, groups_dev, n_seg, seg_start, seg_slot, seg_row, contrib, val, rhs,
                     grad, hdiag);
}

// per PAIR of block rows (2m, 2m+1): scale / clamped LM diagonal (same algebra as pose_diag_kernel), lambda added to the two
// diagonal blocks, and the inverse of the 6x6 block [[D_a, B], [B^T, D_b]] for the block-Jacobi preconditioner (B = the coupling
// block of the two rows — position and orientation of one pose in a pose graph; 0 when the rows are not coupled, which leaves two
// 3x3 inverses).  3x3 blocks alone took ~315 iterations per LM step on the 5 000-pose graph (profiles/r02_c4_kernel_stats.csv).
BSG_DEV void lm_diag_block(int br, double* __restrict__ D, const double* __restrict__ hdiag, double inv_radius, int compute_scale,
                           int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* __restrict__ scale, double* __restrict__ dcl) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 3 * br + i;
    const double h = hdiag[j];
    const double sc = compute_scale ? (jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0) : scale[j];
    const double d = compute_dcl ? fmin(fmax(sc * sc * h, lm_lo), lm_hi) / (sc * sc) : dcl[j];
    if (compute_scale) scale[j] = sc;
    if (compute_dcl) dcl[j] = d;
    D[4 * i] += d * inv_radius;
  }
}
__global__ __launch_bounds__(256) void bsr_finish_diag_kernel(int nbr, const int* __restrict__ diag_slot, const int* __restrict__ pair_slot,
                                                              double* __restrict__ val, const double* __restrict__ hdiag, double inv_radius,
                                                              int compute_scale, int compute_dcl, int jacobi, double lm_lo,
                                                              double lm_hi, double* __restrict__ scale, double* __restrict__ dcl,
                                                              double* __restrict__ Minv) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int npair = (nbr + 1) / 2;
  if (m >= npair) return;
  const int ra = 2 * m, rb = 2 * m + 1;
  const bool has_b = rb < nbr;
  double* Da = val + (size_t)diag_slot[ra] * 9;
  lm_diag_block(ra, Da, hdiag, inv_radius, compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl);
  double A[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i * 6 + j] = Da[i * 3 + j];
  if (has_b) {
    double* Db = val + (size_t)diag_slot[rb] * 9;
    lm_diag_block(rb, Db, hdiag, inv_radius, compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) A[(3 + i) * 6 + 3 + j] = Db[i * 3 + j];
    const int ps = pair_slot[m];
    if (ps >= 0) {
      const double* B = val + (size_t)ps * 9;   // block (2m, 2m+1)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { A[i * 6 + 3 + j] = B[i * 3 + j]; A[(3 + j) * 6 + i] = B[i * 3 + j]; }
    }
  } else {
#pragma unroll
    for (int i = 3; i < 6; ++i) A[i * 6 + i] = 1.0;
  }
  // 6x6 SPD inverse: Cholesky A = L L^T, W = L^-1, A^-1 = W^T W
  double L[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k < j) d -= L[j * 6 + k] * L[j * 6 + k];
    const double ljj = sqrt(d), inv = 1.0 / ljj;
    L[j * 6 + j] = ljj;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i <= j) continue;
      double v = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < j) v -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = v * inv;
    }
  }
  double W[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) W[i] = 0.0;
#pragma unroll
  for (int cix = 0; cix < 6; ++cix) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < cix) continue;
      double v = (i == cix) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k >= cix && k < i) v -= L[i * 6 + k] * W[k * 6 + cix];
      W[i * 6 + cix] = v / L[i * 6 + i];
    }
  }
  double* M = Minv + (size_t)m * 36;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k >= i && k >= j) v += W[k * 6 + i] * W[k * 6 + j];
      M[i * 6 + j] = v;
    }
}
void launch_bsr_finish_diag(hipStream_t s, int nbr, const int* diag_slot, const int* pair_slot, double* val, const double* hdiag, double radius,
                            int compute_scale, int compute_dcl, int jacobi, double lm_lo, double lm_hi, double* scale,
                            double* dcl, double* Minv) {
  const int npair = (nbr + 1) / 2;
  hipLaunchKernelGGL(bsr_finish_diag_kernel, dim3((npair + 255) / 256), dim3(256), 0, s, nbr, diag_slot, pair_slot, val, hdiag, 1.0 / radius,
                     compute_scale, compute_dcl, jacobi, lm_lo, lm_hi, scale, dcl, Minv);
}

// z = M^-1 r for the pair of block rows (2m, 2m+1) (rows >= nbr do not exist: their r is taken as 0 and nothing is written)
// One PCG iteration is TWO launches (a third of the time of an iteration is otherwise the ~5 us dependency latency of
// each extra launch; the SpMV itself streams 30 MB in ~6 us):
//   S_k (pcg_spmv_kernel)    every workgroup reduces the r.z / r.r partials of iteration k itself (same order => same
//                            bits everywhere), so beta_k and the stop test need no launch of their own; the direction
//                            p_k = z_k + beta_k p_{k-1} is formed on the fly for the gathered columns and stored for the
//                            workgroup's own rows into the OTHER p buffer (rows are gathered by other workgroups while
//                            they are written, hence the double buffer); q = A p_k, partials of p_k.q
//   U_k (pcg_update_kernel)  alpha_k = rz_k / (p_k.q) from the partials; x += alpha p_k; r -= alpha q; z = M^-1 r;
//                            partials of r.z, r.r for iteration k+1 into the other half of `part` (S_{k+1} needs both
//                            rz_{k+1} and rz_k)
// The stop flag is sticky in the device scalars (written by workgroup 0 of S_k, read from the next launch on), the
// host looks at it every 20 iterations.
struct PcgTotals { double rz, rr; };
// sum of the (r.z, r.r) partials, by the calling WAVE (every lane gets the totals; same order everywhere => same bits)
BSG_DEV PcgTotals pcg_totals_wave(const double* __restrict__ part, int n_part) {
  const int lane = threadIdx.x & 63;
  double a = 0, c = 0;
  for (int i = lane; i < n_part; i += 64) { a += part[2 * i]; c += part[2 * i + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
  PcgTotals t;
  t.rz = a; t.rr = c;
  return t;
}
BSG_DEV PcgTotals pcg_totals(const double* __restrict__ part, int n_part, double* sred) {
  double a = 0, c = 0;
  for (int i = threadIdx.x; i < n_part; i += 256) { a += part[2 * i]; c += part[2 * i + 1]; }
  PcgTotals t;
  t.rz = block_sum_256(a, sred);
  t.rr = block_sum_256(c, sred);
  return t;
}

// x = 0, r = b, z = M^-1 r, both p buffers = 0 (p_0 = z_0 + 0 * p_{-1}); partials of r.z, r.r for iteration 0.  One thread per ROW
// (six threads share a 6x6 block of the preconditioner: a thread per block left 2 500 threads on the chip for C4 and read its 288
// bytes of the block alone); 252 = 42 x 6 rows per workgroup, so that no block straddles two workgroups.
constexpr int kPcgRowsPerWg = 252;
__global__ __launch_bounds__(256) void pcg_init_kernel(int nbr, const double* __restrict__ b, const double* __restrict__ Minv,
                                                       double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                       double* __restrict__ p0, double* __restrict__ p1, double* __restrict__ part) {
  __shared__ double sred[4];
  const int n = 3 * nbr;
  const int j = blockIdx.x * kPcgRowsPerWg + threadIdx.x;
  double rz = 0.0, rr = 0.0;
  if (threadIdx.x < kPcgRowsPerWg && j < n) {
    const int m = j / 6, i = j - 6 * m, base = 6 * m;
    const double* M = Minv + (size_t)m * 36 + 6 * i;
    double zv = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) zv += M[k] * ((base + k < n) ? b[base + k] : 0.0);
    const double rv = b[j];
    x[j] = 0; p0[j] = 0; p1[j] = 0; r[j] = rv; z[j] = zv;
    rz = rv * zv; rr = rv * rv;
  }
  const double a = block_sum_256(rz, sred), c = block_sum_256(rr, sred);
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = c; }
}
__global__ void pcg_init_scalars_kernel(const double* __restrict__ part, int np, double* __restrict__ sc) {
  __shared__ double sred[4];
  const PcgTotals t = pcg_totals(part, np, sred);
  if (threadIdx.x == 0) { sc[PC_RZ] = t.rz; sc[PC_RR] = t.rr; sc[PC_RR0] = t.rr; sc[PC_DONE] = (t.rr == 0.0) ? 1.0 : 0.0; sc[PC_ITERS] = 0.0; }
}

constexpr int kSpmvLanes = 16;
// S_k
__global__ __launch_bounds__(256) void pcg_spmv_kernel(int nbr, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                       const double* __restrict__ val, const double* __restrict__ z,
                                                       const double* __restrict__ p_prev, double* __restrict__ p_cur,
                                                       double* __restrict__ q, const double* __restrict__ part_cur,
                                                       const double* __restrict__ part_prev, int n_part, int first,
                                                       double* __restrict__ part_pq, double* __restrict__ sc, double tol2) {
  __shared__ double sred[4];
  // every wave derives beta_k and the stop test itself from the partials (a few dozen numbers): no workgroup barrier
  const PcgTotals cur = pcg_totals_wave(part_cur, n_part);
  const double rz_prev = first ? 0.0 : pcg_totals_wave(part_prev, n_part).rz;
  const bool done = sc[PC_DONE] != 0.0 || !(cur.rr > tol2 * sc[PC_RR0]) || !(cur.rz > 0.0);
  const double beta = (!first && rz_prev != 0.0) ? cur.rz / rz_prev : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (done) sc[PC_DONE] = 1.0;
    else { sc[PC_ITERS] += 1.0; sc[PC_RZ] = cur.rz; sc[PC_RR] = cur.rr; }
  }
  // kLanes lanes per block row (a row of C4 has ~40 blocks of 72 bytes)
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int br = gid / kSpmvLanes, sub = gid % kSpmvLanes;
  double a0 = 0, a1 = 0, a2 = 0, pq = 0;
  if (br < nbr && !done) {
    for (int e = row_ptr[br] + sub; e < row_ptr[br + 1]; e += kSpmvLanes) {
      const double* B = val + (size_t)e * 9;
      const int c = 3 * col[e];
      const double p0 = z[c] + beta * p_prev[c], p1 = z[c + 1] + beta * p_prev[c + 1], p2 = z[c + 2] + beta * p_prev[c + 2];
      a0 += B[0] * p0 + B[1] * p1 + B[2] * p2;
      a1 += B[3] * p0 + B[4] * p1 + B[5] * p2;
      a2 += B[6] * p0 + B[7] * p1 + B[8] * p2;
    }
  }
#pragma unroll
  for (int o = kSpmvLanes / 2; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, kSpmvLanes); a1 += __shfl_xor(a1, o, kSpmvLanes); a2 += __shfl_xor(a2, o, kSpmvLanes); }
  if (br < nbr && sub == 0 && !done) {
    const int c = 3 * br;
    const double p0 = z[c] + beta * p_prev[c], p1 = z[c + 1] + beta * p_prev[c + 1], p2 = z[c + 2] + beta * p_prev[c + 2];
    p_cur[c] = p0; p_cur[c + 1] = p1; p_cur[c + 2] = p2;
    q[c] = a0; q[c + 1] = a1; q[c + 2] = a2;
    pq = a0 * p0 + a1 * p1 + a2 * p2;
  }
  const double t = block_sum_256(pq, sred);
  if (threadIdx.x == 0) part_pq[blockIdx.x] = t;
}

// U_k (one thread per row, as pcg_init_kernel)
__global__ __launch_bounds__(256) void pcg_update_kernel(int nbr, const double* __restrict__ part_pq, int n_part_pq,
                                                         const double* __restrict__ Minv, const double* __restrict__ p,
                                                         const double* __restrict__ q, double* __restrict__ x,
                                                         double* __restrict__ r, double* __restrict__ z,
                                                         double* __restrict__ part_next, const double* __restrict__ sc) {
  __shared__ double sred[4];
  __shared__ double s_alpha;
  double a = 0;
  for (int i = threadIdx.x; i < n_part_pq; i += 256) a += part_pq[i];
  const double pq = block_sum_256(a, sred);
  if (threadIdx.x == 0) s_alpha = (pq > 0.0) ? sc[PC_RZ] / pq : 0.0;   // sc[PC_RZ] = rz_k, stored by S_k
  __syncthreads();
  if (sc[PC_DONE] != 0.0) return;   // (uniform; the partials of the stopping iteration stay as they are)
  const double alpha = s_alpha;
  const int n = 3 * nbr;
  const int j = blockIdx.x * kPcgRowsPerWg + threadIdx.x;
  const bool live = threadIdx.x < kPcgRowsPerWg && j < n;
  double rz = 0.0, rr = 0.0, rv = 0.0, zv = 0.0;
  if (live) {
    const int m = j / 6, i = j - 6 * m, base = 6 * m;
    const double* M = Minv + (size_t)m * 36 + 6 * i;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double rk = (base + k < n) ? r[base + k] - alpha * q[base + k] : 0.0;   // (the block's six rows: same workgroup, read before the barrier)
      zv += M[k] * rk;
      rv = (k == i) ? rk : rv;
    }
  }
  __syncthreads();
  if (live) {
    x[j] += alpha * p[j];
    r[j] = rv; z[j] = zv;
    rz = rv * zv; rr = rv * rv;
  }
  const double t1 = block_sum_256(rz, sred), t2 = block_sum_256(rr, sred);
  if (threadIdx.x == 0) { part_next[2 * blockIdx.x] = t1; part_next[2 * blockIdx.x + 1] = t2; }
}

int pcg_rows_grid(int nbr) { return (3 * nbr + kPcgRowsPerWg - 1) / kPcgRowsPerWg; }

// ---------------------------------------------------------------------------------------------------
// The whole PCG solve as ONE launch (round 2).  The two-launch iteration above spends ~5 us per launch on the dependent-launch
// latency (208 iterations x 2 per LM step on C4); here a grid of at most one workgroup per CU stays resident, every workgroup owns
// a contiguous range of block rows (whole 6x6 preconditioner blocks, balanced by non-zero blocks) and keeps x, r, p, q, z of its rows
// in LDS for the whole solve.  What crosses workgroups per iteration is z (one sc1 store per row, gathered by the workgroups whose
// blocks name the column) and the two dot products:
//   * a workgroup keeps its own copy of p for the columns its blocks name (LDS): p_j = z_j + beta p_j is formed locally from the
//     gathered z_j, so p itself never travels;
//   * a dot product is a SLOT per workgroup in device memory, empty (all bits set) until its owner stores the partial: every
//     workgroup polls all slots and sums them in slot order (the same bits everywhere) — partial and arrival are ONE 8-byte store,
//     no counter, no atomics.  Three slot sets rotate; a workgroup re-empties its slot of the next set when it arrives (everybody
//     has finished reading that set: they arrived at the barrier in between).
// Two such barriers per iteration (p.q, then r.z / r.r).  Hand-off rules as in k_chol.hip (guide G16): stores that other workgroups
// read are agent-scope (sc1, write-through), every storing thread fences before the workgroup barrier that precedes the slot store,
// readers use agent-scope loads (no fences: an agent-scope fence writes back / invalidates the L2, and the matrix would be re-fetched every iteration).  Every poll is bounded by the wall clock: on a time-out (the grid not co-resident: another
// persistent kernel on the device) the abort word is raised, the kernel drains with PC_DONE = -1 and the host runs the launch-per-
// iteration path instead.
// ---------------------------------------------------------------------------------------------------
constexpr int kPcgPersistThreads = 512;
constexpr int kPcgPersistMaxRows = 504;          // scalar rows of one workgroup (one thread each in the update)
constexpr unsigned long long kSlotEmpty = ~0ull;
constexpr int kPcgSlotWords = 8;                // 8-byte words of a workgroup's slot: one 64-byte line (slots of different workgroups in one line
                                                 // serialise at the memory side: 8-byte spacing 3.65 us per reduction, 64-byte 2.7; measured in round 2)
constexpr int kPcgCoarse = 6;                   // coarse vectors of the two-level preconditioner (rigid motions of the whole graph)
constexpr int kTailCap = 384;     // blocks of a workgroup beyond the register-resident ones that are kept in LDS (the rest: from memory)
constexpr int kGatherPoses = 2;   // poses (pairs of block columns, 48 bytes = three 16-byte loads) of a thread in the in-flight gather   // 8-byte words between the slots of two workgroups
constexpr int kRowLanes = 8;                     // lanes per block row in the product (64 rows at a time: a workgroup of C4 has ~40)
constexpr int kRegRows = 1, kRegBlocks = 6;     // register-resident blocks of a lane: rows of its lane group x blocks of the row
BSG_DEV double ld_agent(const double* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)v);
}
BSG_DEV void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct PcgBarrier {
  unsigned long long* slots;   // [3 sets][G workgroups][kPcgSlotWords]: the partials of one workgroup are ONE 64-byte line (one store request)
  int* abort_w;
  int G;
  long long deadline;
  double* sbuf;                // LDS: kPcgSlotWords * G partials + kPcgSlotWords (unused) + 1 flag
  double* tot;                 // LDS: the totals of the last reduction (kPcgSlotWords)
};
// stores this workgroup's partial(s) for barrier `b` and returns the totals over all workgroups; false on abort (uniform)
struct PcgNoSide {
  BSG_DEV void issue(int) const {}
  BSG_DEV bool poll() const { return true; }     // true: nothing outstanding
  BSG_DEV void finish(int) const {}
};
// `side`: work whose loads travel with the looks at the slots (the gather of the published z: it needs the slots no more than the
// slots need it) — issue() requests, poll() re-requests what was still empty and says whether anything is outstanding, finish() stores.
// Slots and side job are polled in ONE loop: a round is one memory round trip whatever it waits for.
// NV <= kPcgSlotWords values per workgroup (the p.q reduction of the two-level preconditioner carries W^T q with it: seven): lane v of
// the first wave stores value v — one instruction, one line —, every looking thread reads the NV words of one workgroup with 16-byte
// loads and wave v sums value v in slot order (the same bits everywhere).
// The workgroup's partials are read from LDS (`part`, NV doubles) and the totals are left in LDS (B.tot): no registers held across the wait.
template <int NV, class Side>
BSG_DEV bool pcg_grid_reduce(const PcgBarrier& B, int b, const double* part, Side& side) {
  static_assert(NV >= 1 && NV <= kPcgSlotWords, "a workgroup's partials are one line");
  const int tid = threadIdx.x, G = B.G, wg = blockIdx.x;
  const int set = b % 3, nxt = (b + 1) % 3;
  unsigned long long* cur = B.slots + (size_t)set * G * kPcgSlotWords;
  unsigned long long* nx = B.slots + (size_t)nxt * G * kPcgSlotWords;
  if (tid < kPcgSlotWords) {
    if (tid < NV) __hip_atomic_store(cur + (size_t)wg * kPcgSlotWords + tid, (unsigned long long)__double_as_longlong(part[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(nx + (size_t)wg * kPcgSlotWords + tid, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  double* flag = B.sbuf + kPcgSlotWords * G + kPcgSlotWords;
  if (tid == 0) *flag = 0.0;
  __syncthreads();
  // (8-byte agent-scope loads of every word.  16-byte sc1 buffer loads of the line — what the z gather uses — were tried here: a look
  // that found the line empty could go on returning the stale line for ever, a different set of looking threads each run; the gather,
  // whose lines are pushed out by its own traffic, has not shown it.  Watching ONE word of the line and fetching the others once it is
  // there costs a second round trip (the seven-word reduction 4.65 us against 3.66), two looking threads per line changed nothing.
  // Measured in round 3.)
  const bool poller = tid < G;
  const unsigned long long* line = cur + (size_t)tid * kPcgSlotWords;
  unsigned long long w[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) w[k] = poller ? __hip_atomic_load(line + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  side.issue(tid);
  bool aborted = false;
  for (unsigned it = 0;; ++it) {
    const bool side_done = side.poll();
    bool have = true;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (w[k] == kSlotEmpty) { have = false; w[k] = __hip_atomic_load(line + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (have && side_done) break;
    if ((it & 15) == 15) {
      if (__hip_atomic_load(B.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) { aborted = true; break; }   // (1 = raised; the word starts with all bits set, like the slots)
      if ((long long)wall_clock64() > B.deadline) { __hip_atomic_store(B.abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (aborted) *flag = 1.0;
  if (poller) {
#pragma unroll
    for (int k = 0; k < NV; ++k) B.sbuf[k * G + tid] = __longlong_as_double((long long)w[k]);
  }
  side.finish(tid);
  __syncthreads();
  {
    const int wv = tid >> 6, ln = tid & 63;
    if (wv < NV) {
      double a2 = 0.0;
      for (int i = ln; i < G; i += 64) a2 += B.sbuf[wv * G + i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a2 += __shfl_xor(a2, o, 64);
      if (ln == 0) B.tot[wv] = a2;
    }
  }
  __syncthreads();
  const bool ok = *flag == 0.0;
  __syncthreads();   // (sbuf is rewritten by the next reduction)
  return ok;
}
// sums of NV <= 8 values over the 512 threads (smem >= 64 doubles), totals left in LDS (`out`).  In a wave the values are summed
// TRANSPOSED (as wave_sum_transpose64): three exchange steps over the lane bits 0-2 leave lane l with the 8-lane partial of value l & 7
// (4 + 2 + 1 exchanges), three plain steps over the bits 3-5 complete it — 10 exchanges instead of 6 per value; then thread k adds the
// eight waves' sums of value k in one order.
template <int NV>
BSG_DEV void block_sum_vec(const double (&vin)[NV], double* smem, double* out /* LDS, NV */) {
  static_assert(NV <= 8, "eight values per lane group");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = k < NV ? vin[k] : 0.0;
#pragma unroll
  for (int o = 4, h = 4; h >= 1; o >>= 1, h >>= 1) {
    const bool upper = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const double lo = v[i], hi = v[i + h];
      const double recv = __shfl_xor(upper ? lo : hi, o, 64);
      v[i] = (upper ? hi : lo) + recv;
    }
  }
  double t = v[0];
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
  if (lane < 8) smem[w * 8 + lane] = t;
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    out[k] = ((smem[k] + smem[8 + k]) + (smem[16 + k] + smem[24 + k])) + ((smem[32 + k] + smem[40 + k]) + (smem[48 + k] + smem[56 + k]));
  }
  __syncthreads();
}
BSG_DEV double block_sum_512(double v, double* smem /* >= 8 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  const double t = ((smem[0] + smem[1]) + (smem[2] + smem[3])) + ((smem[4] + smem[5]) + (smem[6] + smem[7]));
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(kPcgPersistThreads) void pcg_persistent_kernel(
    int nbr, const int* __restrict__ row_ptr, const double* __restrict__ val, const double* __restrict__ Minv, const double* __restrict__ b,
    double* __restrict__ x, double* __restrict__ zg, const int* __restrict__ wg_row, const int* __restrict__ wg_colptr,
    const int* __restrict__ wg_cols, const int* __restrict__ lcol, unsigned long long* __restrict__ slots, int* __restrict__ abort_w,
    double* __restrict__ sc, double tol2, int max_it, long long timeout_ticks, int max_cols, int paired, const double* __restrict__ Wc,
    const double* __restrict__ Einv, long long* __restrict__ probe) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, wg = blockIdx.x, G = gridDim.x;
  double* sx = lds;                               // own rows: x, r, p, q, z
  double* sr = sx + kPcgPersistMaxRows;
  double* sp = sr + kPcgPersistMaxRows;
  double* sq = sp + kPcgPersistMaxRows;
  double* sz = sq + kPcgPersistMaxRows;
  double* sred = sz + kPcgPersistMaxRows;         // 64 | E^-1
  double* sE = sred + 64;       // 36 | W^T r (kPcgCoarse) | c = E^-1 W^T r (kPcgCoarse)
  double* swr = sE + 36;
  double* scc = swr + kPcgCoarse;
  double* spart = scc + kPcgCoarse;               // this workgroup's partials of a reduction | the totals of the last one: 2 x kPcgSlotWords
  double* stot = spart + kPcgSlotWords;
  double* sW = stot + kPcgSlotWords;                  // the own rows of W: kPcgCoarse x kPcgPersistMaxRows (registers are taken: 54 of matrix blocks per lane)
  double* sbar = sW + kPcgCoarse * kPcgPersistMaxRows;   // kPcgSlotWords G + kPcgSlotWords + 1
  double* pc = sbar + kPcgSlotWords * 256 + kPcgSlotWords + 4;   // p of the named columns: 3 * max_cols
  double* zc = pc + 3 * max_cols;                 // their z as gathered: 3 * max_cols
  double* tb = zc + 3 * max_cols;                 // blocks that do not fit the registers ("tails"): 9 * kTailCap
  int* tl = reinterpret_cast<int*>(tb + 9 * kTailCap);   // ... their column offsets into pc: kTailCap
  int* toff = tl + kTailCap;                      // first tail slot of each own block row: kPcgPersistMaxRows / 3 + 1
  const int r0 = wg_row[wg], r1 = wg_row[wg + 1], nrow = 3 * (r1 - r0), n = 3 * nbr;
  const size_t zset = 6 * (size_t)((nbr + 1) / 2);   // doubles of one z set (whole poses: a set starts 16-byte aligned)
  const int c0 = wg_colptr[wg], nc = wg_colptr[wg + 1] - c0;
  PcgBarrier B;
  B.slots = slots; B.abort_w = abort_w; B.G = G; B.deadline = (long long)wall_clock64() + timeout_ticks; B.sbuf = sbar; B.tot = stot;
  const int row = 3 * r0 + tid;                   // this thread's scalar row in the update
  const bool live = tid < nrow;
  const int m = row / 6, mi = row - 6 * m, mbase = 6 * m - 3 * r0;   // its 6x6 preconditioner block (inside the workgroup: r0 is even)
  double Mrow[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Mrow[k] = live ? Minv[(size_t)m * 36 + 6 * mi + k] : 0.0;
  // Two-level preconditioner M^-1 = B^-1 + W E^-1 W^T (B: the 6x6 blocks, W: the kPcgCoarse rigid motions of the whole graph — what the
  // anchor alone holds against, the six smallest eigenvalues of the block-Jacobi-preconditioned matrix —, E = W^T A W).  W^T r is kept
  // by its recurrence W^T r -= alpha W^T q, and W^T q travels WITH p.q in the same 64-byte slot: no third reduction per iteration.
  const bool coarse = Wc != nullptr;
#pragma unroll
  for (int k = 0; k < kPcgCoarse; ++k)
    if (tid < kPcgPersistMaxRows) sW[k * kPcgPersistMaxRows + tid] = (coarse && live) ? Wc[(size_t)row * kPcgCoarse + k] : 0.0;
  if (tid < 36) sE[tid] = coarse ? Einv[tid] : 0.0;
  if (tid < 2 * kPcgCoarse) swr[tid] = 0.0;        // (swr | scc)
  for (int i = tid; i < 3 * nc; i += kPcgPersistThreads) pc[i] = 0.0;
  if (live) { sx[tid] = 0.0; sp[tid] = 0.0; sr[tid] = b[row]; }
  __syncthreads();
  int bar = 0;
  bool ok = true;
  auto coarse_coeffs = [&](double alpha_, const double* tq) {   // W^T r -= alpha W^T q;  c = E^-1 (W^T r)   (uniform: two barriers)
    if (tid < kPcgCoarse) swr[tid] -= alpha_ * tq[tid];
    __syncthreads();
    if (tid < kPcgCoarse) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < kPcgCoarse; ++k) a += sE[tid * kPcgCoarse + k] * swr[k];
      scc[tid] = a;
    }
    __syncthreads();
  };
  auto coarse_z = [&]() {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < kPcgCoarse; ++k) a += sW[k * kPcgPersistMaxRows + tid] * scc[k];
    return a;
  };
  if (coarse) {   // W^T r of the start (r = b): one reduction more per solve
    double v6[kPcgCoarse];
#pragma unroll
    for (int k = 0; k < kPcgCoarse; ++k) v6[k] = live ? sW[k * kPcgPersistMaxRows + tid] * sr[tid] : 0.0;
    block_sum_vec<kPcgCoarse>(v6, sred, spart);
    PcgNoSide no_side0;
    ok = pcg_grid_reduce<kPcgCoarse>(B, bar++, spart, no_side0);
    coarse_coeffs(-1.0, stot);
  }
  double part[2] = {0.0, 0.0};
  if (live && tid < kPcgPersistMaxRows) {
    double zv = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) zv += Mrow[k] * ((6 * m + k < n) ? sr[mbase + k] : 0.0);
    if (coarse) zv += coarse_z();
    sz[tid] = zv;
    st_agent(zg + row, zv);                         // z of iteration k lives in set k % 2 of zg (both sets start empty)
    part[0] = sr[tid] * zv; part[1] = sr[tid] * sr[tid];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the z stores have left; no fence: its cache write-back / invalidate would evict the matrix)
  block_sum_vec<2>(part, sred, spart);
  // the z of the named columns: an entry is empty (all bits set) until its owner has stored it
  // The gather of the named columns' z.  The two block columns (2m, 2m + 1) of a pose are named together (`paired`: host check) and are
  // 48 contiguous bytes of z: three 16-byte loads per pose.  (Measured: 8-byte loads 4.6 us for the phase, 16-byte 3.1; a 64-byte line
  // per pose fetched by a quad — a third of the requests — 3.9: the phase moves ~25 MB of uncached sectors per iteration over the
  // fabric, it is not bound by the request count.)  Unpaired lists and poses beyond kGatherPoses x 512: one 8-byte load per value
  // after the reduction.
  const __amdgpu_buffer_rsrc_t rz_res = __builtin_amdgcn_make_buffer_rsrc(zg, 0, (int)(sizeof(double) * 2 * zset), 0x00020000);
  struct Gather {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t res;
    const int* cols; double* zc;
    size_t set_base; int nc, paired;
    unsigned long long v[kGatherPoses][6];
    unsigned off[kGatherPoses];
    int t;
    BSG_DEV void load(int u, int k) {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(res, off[u] + 16 * k, 0, 16 /* sc1 */);
      v[u][2 * k] = ((unsigned long long)w.y << 32) | w.x; v[u][2 * k + 1] = ((unsigned long long)w.w << 32) | w.z;
    }
    BSG_DEV void issue(int tid_) {
      t = tid_;
      const int np = nc >> 1;
#pragma unroll
      for (int u = 0; u < kGatherPoses; ++u) {
        const int pi = t + u * kPcgPersistThreads;
#pragma unroll
        for (int k = 0; k < 6; ++k) v[u][k] = 0ull;
        if (paired && pi < np) {
          off[u] = (unsigned)((set_base + 3 * (size_t)cols[2 * pi]) * sizeof(double));
#pragma unroll
          for (int k = 0; k < 3; ++k) load(u, k);
        }
      }
    }
    BSG_DEV bool poll() {
      bool all = true;
#pragma unroll
      for (int u = 0; u < kGatherPoses; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (v[u][2 * k] == kSlotEmpty || v[u][2 * k + 1] == kSlotEmpty) { load(u, k); all = false; }
      return all;
    }
    BSG_DEV void finish(int) {
      const int np = nc >> 1;
#pragma unroll
      for (int u = 0; u < kGatherPoses; ++u) {
        const int pi = t + u * kPcgPersistThreads;
        if (paired && pi < np) {
#pragma unroll
          for (int k = 0; k < 6; ++k) zc[6 * pi + k] = __longlong_as_double((long long)v[u][k]);
        }
      }
    }
  };
  auto gather_rest = [&](int it_) {
    const unsigned long long* zcur = reinterpret_cast<const unsigned long long*>(zg + (size_t)(it_ & 1) * zset);
    const int first = paired ? 2 * kGatherPoses * kPcgPersistThreads : 0;
    for (int cI = first + tid; cI < nc; cI += kPcgPersistThreads) {
      const unsigned long long* zp = zcur + 3 * (size_t)wg_cols[c0 + cI];
#pragma unroll
      for (int k = 0; k < 3; ++k) zc[3 * cI + k] = __longlong_as_double((long long)__hip_atomic_load(zp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  };
  // this lane's blocks of its group's first kRegRows rows, for the whole solve (a block missing from the row: zeros, column 0)
  const int grp = tid / kRowLanes, sub = tid % kRowLanes;
  constexpr int kRowGroups = kPcgPersistThreads / kRowLanes;
  double rb[kRegRows][kRegBlocks][9];
  int rl[kRegRows][kRegBlocks];
#pragma unroll
  for (int rr_ = 0; rr_ < kRegRows; ++rr_) {
    const int br = r0 + grp + rr_ * kRowGroups;
#pragma unroll
    for (int k = 0; k < kRegBlocks; ++k) {
      const int e = (br < r1) ? row_ptr[br] + sub + kRowLanes * k : 0;
      const bool have = br < r1 && e < row_ptr[br + 1];
#pragma unroll
      for (int i = 0; i < 9; ++i) rb[rr_][k][i] = have ? val[(size_t)e * 9 + i] : 0.0;
      rl[rr_][k] = have ? 3 * lcol[e] : 0;
    }
  }
  // the blocks the registers do not hold, into LDS (slot = toff[row] + position in the row's tail), as far as kTailCap goes
  auto tail_start = [&](int br) {
    const bool reg_row = (br - r0) < kRegRows * kRowGroups;
    return min(row_ptr[br] + (reg_row ? kRowLanes * kRegBlocks : 0), row_ptr[br + 1]);
  };
  if (tid == 0) {
    int o = 0;
    for (int br = r0; br < r1; ++br) { toff[br - r0] = o; o += row_ptr[br + 1] - tail_start(br); }
    toff[r1 - r0] = o;
  }
  __syncthreads();
  for (int br = r0 + grp; br < r1; br += kRowGroups) {
    const int ts = tail_start(br), o = toff[br - r0];
    for (int e = ts + sub; e < row_ptr[br + 1]; e += kRowLanes) {
      const int slot = o + (e - ts);
      if (slot < kTailCap) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tb[9 * slot + i] = val[(size_t)e * 9 + i];
        tl[slot] = 3 * lcol[e];
      }
    }
  }
  __syncthreads();
  int iters = 0;
  double rz = 0.0, rr = 0.0, rz_prev = 0.0, rr0 = 0.0;
  auto stamp = [&](int it, int k) { if (probe && wg == 0 && tid == 0 && it < 64) probe[it * 8 + k] = (long long)wall_clock64(); };
  for (int it = 0; ok; ++it) {
    stamp(it, 0);
    {
      Gather gth;
      gth.res = rz_res; gth.cols = wg_cols + c0; gth.zc = zc; gth.set_base = (size_t)(it & 1) * zset; gth.nc = nc; gth.paired = paired;
      ok = pcg_grid_reduce<2>(B, bar++, spart, gth);
      rz = stot[0]; rr = stot[1];
    }
    if (!ok) break;
    if (!paired || nc > 2 * kGatherPoses * kPcgPersistThreads) {   // (workgroup-uniform)
      gather_rest(it);   // (every z is there: each workgroup stored its z before its slot)
      __syncthreads();   // (zc is read across threads below)
    }
    stamp(it, 1);
    if (probe && it == 11 && tid == 0) probe[512 + 4 * wg + 3] = (long long)wall_clock64();
    if (it == 0) rr0 = rr;
    if (it >= max_it || !(rr > tol2 * rr0) || !(rz > 0.0)) break;
    const double beta = (it > 0 && rz_prev != 0.0) ? rz / rz_prev : 0.0;
    rz_prev = rz;
    ++iters;
    // p of the named columns from the published z (and of the own rows)
    for (int i = tid; i < 3 * nc; i += kPcgPersistThreads) pc[i] = zc[i] + beta * pc[i];
    if (live) sp[tid] = sz[tid] + beta * sp[tid];
    __syncthreads();
    stamp(it, 2);
    // q = A p for the own rows: kRowLanes lanes per block row; the blocks come from registers (loaded once per solve), what does not
    // fit (rows beyond kRegRows per lane group, blocks beyond kRowLanes x kRegBlocks of a row) from memory as before
    {
      auto row_tail = [&](int br, int e_from, double& a0, double& a1, double& a2) {
        const int e_end = row_ptr[br + 1];
        if (e_from >= e_end) return;
        const int ts = tail_start(br), o = toff[br - r0];
        for (int e = e_from; e < e_end; e += kRowLanes) {
          const int slot = o + (e - ts);
          const bool in_lds = slot < kTailCap;
          const double* Bv = in_lds ? tb + 9 * slot : val + (size_t)e * 9;
          const double* pv = pc + (in_lds ? tl[slot] : 3 * lcol[e]);
          const double p0 = pv[0], p1 = pv[1], p2 = pv[2];
          a0 += Bv[0] * p0 + Bv[1] * p1 + Bv[2] * p2;
          a1 += Bv[3] * p0 + Bv[4] * p1 + Bv[5] * p2;
          a2 += Bv[6] * p0 + Bv[7] * p1 + Bv[8] * p2;
        }
      };
      auto row_done = [&](int br, double a0, double a1, double a2) {
#pragma unroll
        for (int o = kRowLanes / 2; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, kRowLanes); a1 += __shfl_xor(a1, o, kRowLanes); a2 += __shfl_xor(a2, o, kRowLanes); }
        if (sub == 0) {
          const int j = 3 * (br - r0);
          sq[j] = a0; sq[j + 1] = a1; sq[j + 2] = a2;
        }
      };
#pragma unroll
      for (int rr_ = 0; rr_ < kRegRows; ++rr_) {
        const int br = r0 + grp + rr_ * kRowGroups;
        if (br < r1) {
          double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
          for (int k = 0; k < kRegBlocks; ++k) {
            const double* pv = pc + rl[rr_][k];
            const double p0 = pv[0], p1 = pv[1], p2 = pv[2];
            a0 += rb[rr_][k][0] * p0 + rb[rr_][k][1] * p1 + rb[rr_][k][2] * p2;
            a1 += rb[rr_][k][3] * p0 + rb[rr_][k][4] * p1 + rb[rr_][k][5] * p2;
            a2 += rb[rr_][k][6] * p0 + rb[rr_][k][7] * p1 + rb[rr_][k][8] * p2;
          }
          row_tail(br, row_ptr[br] + sub + kRowLanes * kRegBlocks, a0, a1, a2);
          row_done(br, a0, a1, a2);
        }
      }
      for (int br = r0 + grp + kRegRows * kRowGroups; br < r1; br += kRowGroups) {
        double a0 = 0, a1 = 0, a2 = 0;
        row_tail(br, row_ptr[br] + sub, a0, a1, a2);
        row_done(br, a0, a1, a2);
      }
    }
    __syncthreads();   // (q of the own rows is complete)
    // p.q and W^T q: one partial each per workgroup, one slot line
    double pq_all = 0.0;
    if (coarse) {
      double v7[1 + kPcgCoarse];
      v7[0] = live ? sp[tid] * sq[tid] : 0.0;
#pragma unroll
      for (int k = 0; k < kPcgCoarse; ++k) v7[1 + k] = live ? sW[k * kPcgPersistMaxRows + tid] * sq[tid] : 0.0;
      block_sum_vec<1 + kPcgCoarse>(v7, sred, spart);
      stamp(it, 3);
      if (probe && it == 10 && tid == 0) probe[512 + 4 * wg] = (long long)wall_clock64();
      PcgNoSide no_side;
      ok = pcg_grid_reduce<1 + kPcgCoarse>(B, bar++, spart, no_side);
      if (!ok) break;
      pq_all = stot[0];
      coarse_coeffs((pq_all > 0.0) ? rz / pq_all : 0.0, stot + 1);
    } else {
      double v1[1] = {live ? sp[tid] * sq[tid] : 0.0};
      block_sum_vec<1>(v1, sred, spart);
      stamp(it, 3);
      if (probe && it == 10 && tid == 0) probe[512 + 4 * wg] = (long long)wall_clock64();
      PcgNoSide no_side;
      ok = pcg_grid_reduce<1>(B, bar++, spart, no_side);
      if (!ok) break;
      pq_all = stot[0];
    }
    stamp(it, 4);
    if (probe && it == 10 && tid == 0) probe[512 + 4 * wg + 1] = (long long)wall_clock64();
    const double alpha = (pq_all > 0.0) ? rz / pq_all : 0.0;
    // x += alpha p; r -= alpha q; z = M^-1 r; publish z; partials of r.z, r.r
    if (live) { sx[tid] += alpha * sp[tid]; sr[tid] -= alpha * sq[tid]; }
    __syncthreads();
    part[0] = 0.0; part[1] = 0.0;
    if (live) {
      double zv = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) zv += Mrow[k] * ((6 * m + k < n) ? sr[mbase + k] : 0.0);
      if (coarse) zv += coarse_z();
      sz[tid] = zv;
      st_agent(zg + (size_t)((it + 1) & 1) * zset + row, zv);
      st_agent(zg + (size_t)(it & 1) * zset + row, __longlong_as_double((long long)kSlotEmpty));   // (everybody has gathered z of this iteration: they arrived at the p.q barrier)
      part[0] = sr[tid] * zv; part[1] = sr[tid] * sr[tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the z stores have left; no fence: its cache write-back / invalidate would evict what the L2 holds)
    block_sum_vec<2>(part, sred, spart);               // (its barriers order every thread's drain before the slot store)
    stamp(it, 5);
    if (probe && it == 10 && tid == 0) probe[512 + 4 * wg + 2] = (long long)wall_clock64();
  }
  if (live) x[row] = sx[tid];
  if (wg == 0 && tid == 0) {
    sc[PC_ITERS] = (double)iters; sc[PC_RZ] = rz; sc[PC_RR] = rr; sc[PC_RR0] = rr0;
    sc[PC_DONE] = ok ? 1.0 : -1.0;
  }
}

// ---------------------------------------------------------------------------------------------------
// The coarse space of the two-level preconditioner, rebuilt per LM step (the poses move): column a < 3 of W translates every free
// pose along axis a, column 3 + a rotates the graph about axis a through the centre c — position rows e_a x (p - c), orientation rows
// R^T e_a (the tangent of the quaternion blocks is the local one: q (+) d = q exp(d)).  Anchored poses (those an absolute factor
// holds) and blocks that are not part of a pose keep zero rows.  Then A W (block rows x 6), E = W^T (A W) and its inverse.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcg_coarse_w_kernel(int n_co, const int4* __restrict__ co /* x off p, x off q, tangent off p, q */,
                                                           const double* __restrict__ x, double cx, double cy, double cz, double* __restrict__ W) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_co) return;
  const int4 o = co[k];
  const double px = x[o.x] - cx, py = x[o.x + 1] - cy, pz = x[o.x + 2] - cz;
  double R[9];
  const double q[4] = {x[o.y], x[o.y + 1], x[o.y + 2], x[o.y + 3]};
  quat_to_rot_normalized(q, R);
  double* Wp = W + (size_t)o.z * kPcgCoarse;
  double* Wq = W + (size_t)o.w * kPcgCoarse;
  // e_a x p: a = 0: (0, -pz, py); a = 1: (pz, 0, -px); a = 2: (-py, px, 0)
  const double lev[3][3] = {{0.0, -pz, py}, {pz, 0.0, -px}, {-py, px, 0.0}};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      Wp[i * kPcgCoarse + a] = (i == a) ? 1.0 : 0.0;
      Wp[i * kPcgCoarse + 3 + a] = lev[a][i];
      Wq[i * kPcgCoarse + a] = 0.0;
      Wq[i * kPcgCoarse + 3 + a] = R[3 * a + i];       // (R^T e_a)_i = R[a][i]
    }
  }
}
// E = W^T A W without materialising A W: eight lanes per block row take the row's blocks in turn (3 x 6 partial sums of (A W)'s rows
// each), the lane group's sums meet in its first lane, which adds W_row^T (A W)_row into the workgroup's 6 x 6 in LDS; one atomic add
// per entry and workgroup into E; the last workgroup to finish inverts E (Cholesky with a pivot floor: a direction W does not span —
// every pose anchored — drops out, pseudo-inverse) and clears the accumulator for the next LM step.
constexpr int kCoarseRowLanes = 8;
__global__ __launch_bounds__(256) void pcg_coarse_e_kernel(int nbr, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                           const double* __restrict__ val, const double* __restrict__ W, double* __restrict__ E,
                                                           unsigned* __restrict__ counter, double* __restrict__ Einv) {
  __shared__ double sE[36];
  __shared__ int last;
  const int tid = threadIdx.x, sub = tid % kCoarseRowLanes;
  const int br = (blockIdx.x * 256 + tid) / kCoarseRowLanes;
  if (tid < 36) sE[tid] = 0.0;
  __syncthreads();
  double aw[3][kPcgCoarse];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < kPcgCoarse; ++a) aw[i][a] = 0.0;
  if (br < nbr) {
    for (int e = row_ptr[br] + sub; e < row_ptr[br + 1]; e += kCoarseRowLanes) {
      const double* Bv = val + (size_t)e * 9;
      const double2* w2v = reinterpret_cast<const double2*>(W + (size_t)3 * col[e] * kPcgCoarse);   // (144-byte rows: 16-byte pieces)
      double b9[9], w[3 * kPcgCoarse];
#pragma unroll
      for (int i = 0; i < 9; ++i) b9[i] = Bv[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) { const double2 t = w2v[i]; w[2 * i] = t.x; w[2 * i + 1] = t.y; }
#pragma unroll
      for (int a = 0; a < kPcgCoarse; ++a) {
        const double w0 = w[a], w1 = w[kPcgCoarse + a], w2 = w[2 * kPcgCoarse + a];
#pragma unroll
        for (int i = 0; i < 3; ++i) aw[i][a] += b9[3 * i] * w0 + b9[3 * i + 1] * w1 + b9[3 * i + 2] * w2;
      }
    }
  }
#pragma unroll
  for (int o = kCoarseRowLanes / 2; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int a = 0; a < kPcgCoarse; ++a) aw[i][a] += __shfl_xor(aw[i][a], o, kCoarseRowLanes);
  // W_row^T (A W)_row of the lane group's row in its first lane, summed over the wave's eight rows by exchanges (into ONE LDS address per
  // entry from eight lanes at once the adds serialised: 24 of the kernel's 42 us; measured), then one LDS add per wave and entry
  double ec[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) ec[i] = 0.0;
  if (br < nbr && sub == 0) {
    const double* wr = W + (size_t)3 * br * kPcgCoarse;
#pragma unroll
    for (int a = 0; a < kPcgCoarse; ++a) {
      const double w0 = wr[a], w1 = wr[kPcgCoarse + a], w2 = wr[2 * kPcgCoarse + a];
#pragma unroll
      for (int j = 0; j < kPcgCoarse; ++j) ec[a * kPcgCoarse + j] = w0 * aw[0][j] + w1 * aw[1][j] + w2 * aw[2][j];
    }
  }
#pragma unroll
  for (int o = kCoarseRowLanes; o < 64; o <<= 1)
#pragma unroll
    for (int i = 0; i < 36; ++i) ec[i] += __shfl_xor(ec[i], o, 64);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 36; ++i) __hip_atomic_fetch_add(&sE[i], ec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // (a partial per workgroup, summed by the last one in workgroup order: added into 36 shared addresses with atomics — 313 adds per
  // address — the kernel took 42 us; measured)
  // (agent-scope stores, drained before the count goes up; NO fence here: an agent-scope fence writes the L2 back and invalidates it, and
  // 313 workgroups doing that made this kernel 45 us; the one workgroup that sums fences once)
  if (tid < 36) __hip_atomic_store(&E[(size_t)blockIdx.x * 36 + tid], sE[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  __threadfence();
  __shared__ double sP[7 * 36];
  if (tid < 7 * 36) {
    const int k = tid % 36, g0 = tid / 36;
    double e = 0.0;
    const double* Ev = E;      // (plain loads behind the fence: they overlap; agent-scope atomic loads were waited for one by one, 36 us)
#pragma unroll 8
    for (int g = g0; g < (int)gridDim.x; g += 7) e += Ev[(size_t)g * 36 + k];
    sP[tid] = e;
  }
  __syncthreads();
  if (tid < 36) sE[tid] = ((sP[tid] + sP[36 + tid]) + (sP[72 + tid] + sP[108 + tid])) + ((sP[144 + tid] + sP[180 + tid]) + sP[216 + tid]);
  __syncthreads();
  if (tid == 0) {
    *counter = 0u;
    // (every loop unrolled: the arrays stay in registers — with run-time loop bounds they lived in scratch memory and this thread alone
    // took 30 us)
    double A[36], L[36], Wi[36];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) A[i * 6 + j] = 0.5 * (sE[i * 6 + j] + sE[j * 6 + i]);
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) dmax = fmax(dmax, A[i * 6 + i]);
    bool dead[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) { L[i] = 0.0; Wi[i] = 0.0; }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = A[j * 6 + j];
#pragma unroll
      for (int k = 0; k < 6; ++k) if (k < j) d -= L[j * 6 + k] * L[j * 6 + k];
      dead[j] = !(d > 1e-12 * dmax) || !(dmax > 0.0);
      const double ljj = dead[j] ? 1.0 : sqrt(d);
      L[j * 6 + j] = ljj;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i <= j) continue;
        double v = A[i * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) v -= L[i * 6 + k] * L[j * 6 + k];
        L[i * 6 + j] = dead[j] ? 0.0 : v / ljj;
      }
    }
#pragma unroll
    for (int cix = 0; cix < 6; ++cix) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i < cix) continue;
        double v = (i == cix) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k >= cix && k < i) v -= L[i * 6 + k] * Wi[k * 6 + cix];
        Wi[i * 6 + cix] = v / L[i * 6 + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k >= i && k >= j && !dead[k]) v += Wi[k * 6 + i] * Wi[k * 6 + j];
        Einv[i * 6 + j] = (dead[i] || dead[j]) ? 0.0 : v;
      }
  }
}
size_t pcg_coarse_scratch_doubles(int nbr) { return (size_t)3 * nbr * kPcgCoarse + 36 * (size_t)((nbr * kCoarseRowLanes + 255) / 256) + 36 + 2; }
void launch_pcg_coarse(hipStream_t s, const PcgPersistDev& P, int nbr, const int* row_ptr, const int* col, const double* val, const double* x) {
  if (P.n_coarse <= 0) return;
  hipLaunchKernelGGL(pcg_coarse_w_kernel, dim3((P.n_coarse + 255) / 256), dim3(256), 0, s, P.n_coarse, P.co, x, P.centre[0], P.centre[1], P.centre[2], P.W);
  hipLaunchKernelGGL(pcg_coarse_e_kernel, dim3((nbr * kCoarseRowLanes + 255) / 256), dim3(256), 0, s, nbr, row_ptr, col, val, P.W, P.E, P.counter, P.Einv);
}

size_t pcg_persistent_lds(int max_cols) {
  return sizeof(double) * ((5 + kPcgCoarse) * (size_t)kPcgPersistMaxRows + 64 + 36 + 2 * kPcgCoarse + 2 * kPcgSlotWords + kPcgSlotWords * 256 + kPcgSlotWords + 4 + 6 * (size_t)max_cols + 9 * (size_t)kTailCap) +
         sizeof(int) * ((size_t)kTailCap + kPcgPersistMaxRows / 3 + 2);
}
size_t pcg_persistent_lds_limit() { return 150 * 1024; }
int pcg_persistent_max_rows() { return kPcgPersistMaxRows; }
size_t pcg_persistent_z_words(int nbr) { return 2 * 6 * (size_t)((nbr + 1) / 2); }
size_t pcg_persistent_slot_words(int G) { return 3 * (size_t)G * kPcgSlotWords; }
bool launch_pcg_persistent(hipStream_t s, const PcgPersistDev& P, int nbr, const int* row_ptr, const double* val, const double* Minv, const double* b,
                           double* x, double* zg, double* sc, double tol2, int max_it) {
  const size_t lds = pcg_persistent_lds(P.max_cols);
  // (the attribute is per DEVICE: a context on a second GPU of the process needs it set there too)
  static std::atomic<unsigned long long> attr_devices{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pcg_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pcg_persistent_lds_limit()) != hipSuccess) { (void)hipGetLastError(); return false; }
    attr_devices.fetch_or(bit, std::memory_order_release);
  }
  if (lds > pcg_persistent_lds_limit() || P.G < 1 || P.G > 256) return false;
  // slots, both z sets and the abort word are ONE allocation (bsgpu_finalize.cpp), emptied by one fill: all bits set
  if (hipMemsetAsync(P.slots, 0xff, P.sync_bytes, s) != hipSuccess) { (void)hipGetLastError(); return false; }
  const long long timeout_ticks = 100000000LL / 5;   // s_memrealtime: 100 MHz; a fifth of a second for the whole solve
  // BSGPU_PCG_PROBE=1: workgroup 0 stamps the phases of its first 64 iterations with the 100 MHz wall clock; printed once
  static const bool want_probe = getenv("BSGPU_PCG_PROBE") != nullptr;
  static long long* d_probe = nullptr;
  static int probe_left = 3;
  long long* probe = nullptr;
  if (want_probe && probe_left > 0) {
    if (!d_probe && hipMalloc((void**)&d_probe, sizeof(long long) * (512 + 1024)) != hipSuccess) { (void)hipGetLastError(); d_probe = nullptr; }
    if (d_probe) { (void)hipMemsetAsync(d_probe, 0, sizeof(long long) * (512 + 1024), s); probe = d_probe; }
  }
  hipLaunchKernelGGL(pcg_persistent_kernel, dim3(P.G), dim3(kPcgPersistThreads), lds, s, nbr, row_ptr, val, Minv, b, x, zg, P.wg_row, P.wg_colptr,
                     P.wg_cols, P.lcol, P.slots, P.abort_w, sc, tol2, max_it, timeout_ticks, P.max_cols, P.paired, P.n_coarse > 0 ? P.W : nullptr, P.Einv, probe);
  if (probe) {
    long long h[512 + 1024];
    if (hipMemcpyAsync(h, d_probe, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
      double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
      for (int it = 4; it < 63; ++it) {
        if (!h[it * 8 + 5] || !h[(it + 1) * 8]) break;
        for (int k = 0; k < 5; ++k) acc[k] += (double)(h[it * 8 + k + 1] - h[it * 8 + k]);
        acc[5] += (double)(h[(it + 1) * 8] - h[it * 8]);
        ++n;
      }
      if (n) fprintf(stderr, "[pcg probe] per iteration (us, workgroup 0, %d iterations): reduce rz %.2f | gather %.2f | spmv %.2f | reduce pq %.2f | update %.2f | total %.2f\n", n,
                     acc[0] / n / 100, acc[1] / n / 100, acc[2] / n / 100, acc[3] / n / 100, acc[4] / n / 100, acc[5] / n / 100);
    }
      {   // spread over the workgroups at iteration 10: arrival at the p.q reduction, exit from it, arrival at the r.z reduction, exit from it
        long long lo[4], hi[4];
        for (int k = 0; k < 4; ++k) { lo[k] = 1LL << 62; hi[k] = 0; }
        for (int g = 0; g < P.G; ++g) for (int k = 0; k < 4; ++k) { const long long v = h[512 + 4 * g + k]; if (v) { lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); } }
        fprintf(stderr, "[pcg probe] iteration 10 over %d workgroups (us from the first arrival at p.q): arrive p.q %.2f..%.2f | leave %.2f..%.2f | arrive r.z %.2f..%.2f | leave %.2f..%.2f\n",
                P.G, 0.0, (hi[0] - lo[0]) / 100.0, (lo[1] - lo[0]) / 100.0, (hi[1] - lo[0]) / 100.0, (lo[2] - lo[0]) / 100.0, (hi[2] - lo[0]) / 100.0,
                (lo[3] - lo[0]) / 100.0, (hi[3] - lo[0]) / 100.0);
      }
    --probe_left;
  }
  return hipGetLastError() == hipSuccess;
}
// ---------------------------------------------------------------------------------------------------
// PCG on the REDUCED camera system (Ceres: ITERATIVE_SCHUR with the SCHUR_JACOBI preconditioner the reference's
// beam_slam_launch/config/optimization/ceres_config.json:11-12 names).  The operator is the assembled Schur complement itself — the
// 64x64 tiles of S that the assembly writes (solver order, both triangles; DensePlan::touched_tiles without the rhs row/column) —
// the preconditioner the inverses of its diagonal tiles (block-Jacobi with 64-wide blocks: about four keyframes each, a superset
// of Ceres' per-camera blocks).  Same two-launch iteration as above; one workgroup per tile row, every sum in a fixed order.
// ---------------------------------------------------------------------------------------------------
// Minv[I] = S(I,I)^-1 by Gauss-Jordan in LDS (SPD + LM damping: no pivoting; padding rows are unit rows and stay so)
__global__ __launch_bounds__(256) void spcg_tile_inverse_kernel(const double* __restrict__ S, int ld, double* __restrict__ Minv) {
  __shared__ double a[64 * 65];
  __shared__ double colk[64];
  const int I = blockIdx.x, tid = threadIdx.x, r = tid >> 2, qd = tid & 3;
  for (int i = tid; i < 4096; i += 256) a[(i >> 6) * 65 + (i & 63)] = S[(size_t)(I * 64 + (i >> 6)) * ld + I * 64 + (i & 63)];
  __syncthreads();
  for (int k = 0; k < 64; ++k) {
    const double inv = 1.0 / a[k * 65 + k];
    if (tid < 64) colk[tid] = a[tid * 65 + k];
    __syncthreads();
    if (r == k) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { const int cc = 16 * qd + j; a[k * 65 + cc] = (cc == k) ? inv : a[k * 65 + cc] * inv; }
    }
    __syncthreads();
    if (r != k) {
      const double f = colk[r];
#pragma unroll
      for (int j = 0; j < 16; ++j) { const int cc = 16 * qd + j; a[r * 65 + cc] = (cc == k) ? -f * inv : a[r * 65 + cc] - f * a[k * 65 + cc]; }
    }
    __syncthreads();
  }
  for (int i = tid; i < 4096; i += 256) Minv[(size_t)I * 4096 + i] = a[(i >> 6) * 65 + (i & 63)];
}
// y_r = sum_c M[r][c] v[c] for the calling workgroup's 64x64 tile M (row-major, leading dimension ldm) and the 64-vector v in LDS:
// four lanes per row, sixteen columns each; every lane of a row's quad returns the row's sum
BSG_DEV double tile_row_dot(const double* __restrict__ M, int ldm, const double* v) {
  const int r = threadIdx.x >> 2, qd = threadIdx.x & 3;
  const double4* row = reinterpret_cast<const double4*>(M + (size_t)r * ldm + 16 * qd);
  const double4 m0 = row[0], m1 = row[1], m2 = row[2], m3 = row[3];
  const double* w = v + 16 * qd;
  double acc = m0.x * w[0] + m0.y * w[1] + m0.z * w[2] + m0.w * w[3];
  acc += m1.x * w[4] + m1.y * w[5] + m1.z * w[6] + m1.w * w[7];
  acc += m2.x * w[8] + m2.y * w[9] + m2.z * w[10] + m2.w * w[11];
  acc += m3.x * w[12] + m3.y * w[13] + m3.z * w[14] + m3.w * w[15];
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  return acc;
}
// x = 0, r = b (the rhs row of S), z = Minv r, p buffers = 0; partials of r.z and r.r
__global__ __launch_bounds__(256) void spcg_init_kernel(const double* __restrict__ b, const double* __restrict__ Minv, double* __restrict__ x,
                                                        double* __restrict__ r, double* __restrict__ z, double* __restrict__ p0,
                                                        double* __restrict__ p1, double* __restrict__ part) {
  __shared__ double sv[64];
  __shared__ double sred[4];
  const int I = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) { const double v = b[I * 64 + tid]; sv[tid] = v; r[I * 64 + tid] = v; x[I * 64 + tid] = 0.0; p0[I * 64 + tid] = 0.0; p1[I * 64 + tid] = 0.0; }
  __syncthreads();
  const double zr = tile_row_dot(Minv + (size_t)I * 4096, 64, sv);
  double rz = 0.0, rr = 0.0;
  if ((tid & 3) == 0) { const double rv = sv[tid >> 2]; z[I * 64 + (tid >> 2)] = zr; rz = rv * zr; rr = rv * rv; }
  const double t1 = block_sum_256(rz, sred), t2 = block_sum_256(rr, sred);
  if (tid == 0) { part[2 * I] = t1; part[2 * I + 1] = t2; }
}
// S_k: beta_k and the stop test from the partials; one workgroup per CHUNK of up to kSpcgChunk tiles of one tile row I (a separator
// row of a nested-dissection order has dozens of tiles: a workgroup per row would walk them one memory round trip at a time).
// p_k = z_k + beta_k p_{k-1} is formed for the gathered tiles; the chunk's part of q_I goes to qpart[chunk] and its part of
// p_I . q_I to part_pq[chunk] — p.q needs no complete q; the owner of the row's first chunk stores p_I.
constexpr int kSpcgChunk = 4;
__global__ __launch_bounds__(256) void spcg_matvec_kernel(const double* __restrict__ S, int ld, const int* __restrict__ chunk_row,
                                                          const int* __restrict__ chunk_ptr, const int* __restrict__ tcol,
                                                          const double* __restrict__ z, const double* __restrict__ p_prev,
                                                          double* __restrict__ p_cur, double* __restrict__ qpart,
                                                          const double* __restrict__ part_cur, const double* __restrict__ part_prev,
                                                          int n_part, int first, double* __restrict__ part_pq, double* __restrict__ sc,
                                                          double tol2) {
  __shared__ double sp[kSpcgChunk][64];
  __shared__ double sred[4];
  const PcgTotals cur = pcg_totals_wave(part_cur, n_part);
  const double rz_prev = first ? 0.0 : pcg_totals_wave(part_prev, n_part).rz;
  const bool done = sc[PC_DONE] != 0.0 || !(cur.rr > tol2 * sc[PC_RR0]) || !(cur.rz > 0.0);
  const double beta = (!first && rz_prev != 0.0) ? cur.rz / rz_prev : 0.0;
  const int ch = blockIdx.x, tid = threadIdx.x;
  if (ch == 0 && tid == 0) {
    if (done) sc[PC_DONE] = 1.0;
    else { sc[PC_ITERS] += 1.0; sc[PC_RZ] = cur.rz; sc[PC_RR] = cur.rr; }
  }
  if (done) { if (tid == 0) part_pq[ch] = 0.0; return; }
  const int I = chunk_row[ch], e0 = chunk_ptr[ch], e1 = chunk_ptr[ch + 1];
  {
    const int u = tid >> 6, c = tid & 63;   // (256 threads: the four vectors at once)
    if (e0 + u < e1) { const int J = tcol[e0 + u]; sp[u][c] = z[J * 64 + c] + beta * p_prev[J * 64 + c]; }
  }
  __syncthreads();
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < kSpcgChunk; ++u)
    if (e0 + u < e1) acc += tile_row_dot(S + (size_t)(I * 64) * ld + tcol[e0 + u] * 64, ld, sp[u]);
  double pq = 0.0;
  if ((tid & 3) == 0) {
    const int j = I * 64 + (tid >> 2);
    const double pv = z[j] + beta * p_prev[j];
    if (ch == 0 || chunk_row[ch - 1] != I) p_cur[j] = pv;
    qpart[(size_t)ch * 64 + (tid >> 2)] = acc;
    pq = pv * acc;
  }
  const double t = block_sum_256(pq, sred);
  if (tid == 0) part_pq[ch] = t;
}
// U_k: alpha_k from the p.q partials; q_I = the sum of the row's chunk parts (in order); x += alpha p, r -= alpha q, z = Minv r;
// partials of r.z, r.r for iteration k + 1
__global__ __launch_bounds__(256) void spcg_update_kernel(const double* __restrict__ part_pq, int n_chunks, const int* __restrict__ row_chunk_ptr,
                                                          const double* __restrict__ qpart, const double* __restrict__ Minv,
                                                          const double* __restrict__ p, double* __restrict__ x, double* __restrict__ r,
                                                          double* __restrict__ z, double* __restrict__ part_next, const double* __restrict__ sc) {
  __shared__ double sv[64];
  __shared__ double sred[4];
  __shared__ double s_alpha;
  double a = 0;
  for (int i = threadIdx.x; i < n_chunks; i += 256) a += part_pq[i];
  const double pq = block_sum_256(a, sred);
  if (threadIdx.x == 0) s_alpha = (pq > 0.0) ? sc[PC_RZ] / pq : 0.0;
  __syncthreads();
  if (sc[PC_DONE] != 0.0) return;
  const double alpha = s_alpha;
  const int I = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    const int j = I * 64 + tid;
    double qv = 0.0;
    for (int ch = row_chunk_ptr[I]; ch < row_chunk_ptr[I + 1]; ++ch) qv += qpart[(size_t)ch * 64 + tid];
    x[j] += alpha * p[j];
    const double rv = r[j] - alpha * qv;
    r[j] = rv; sv[tid] = rv;
  }
  __syncthreads();
  const double zr = tile_row_dot(Minv + (size_t)I * 4096, 64, sv);
  double rz = 0.0, rr = 0.0;
  if ((tid & 3) == 0) { const double rv = sv[tid >> 2]; z[I * 64 + (tid >> 2)] = zr; rz = rv * zr; rr = rv * rv; }
  const double t1 = block_sum_256(rz, sred), t2 = block_sum_256(rr, sred);
  if (tid == 0) { part_next[2 * I] = t1; part_next[2 * I + 1] = t2; }
}
// the solution (solver order) as the step in natural order: y_tan = x, delta = -x
__global__ __launch_bounds__(64) void spcg_finish_kernel(const double* __restrict__ x, const int* __restrict__ iperm, int n_pose,
                                                         double* __restrict__ y_tan, double* __restrict__ delta) {
  const int j = iperm[blockIdx.x * 64 + threadIdx.x];
  if (j >= 0) { const double v = x[blockIdx.x * 64 + threadIdx.x]; y_tan[j] = v; delta[j] = -v; }
}
void launch_spcg_prepare(hipStream_t s, int T, const double* S, int ld, double* Minv) {
  hipLaunchKernelGGL(spcg_tile_inverse_kernel, dim3(T), dim3(256), 0, s, S, ld, Minv);
}
void launch_spcg_init(hipStream_t s, int T, const double* b, const double* Minv, double* x, double* r, double* z, double* p0, double* p1,
                      double* part, double* sc) {
  hipLaunchKernelGGL(spcg_init_kernel, dim3(T), dim3(256), 0, s, b, Minv, x, r, z, p0, p1, part);
  hipLaunchKernelGGL(pcg_init_scalars_kernel, dim3(1), dim3(256), 0, s, part, T, sc);
}
// iteration k: p buffers and the two halves of `part` (2 T doubles each) alternate
void launch_spcg_iteration(hipStream_t s, int k, int T, const double* S, int ld, int n_chunks, const int* chunk_row, const int* chunk_ptr,
                           const int* row_chunk_ptr, const int* tcol, const double* Minv, double* x, double* r, double* z, double* p0,
                           double* p1, double* qpart, double* part_pq, double* part, double* sc, double tol2) {
  double* p_cur = (k & 1) ? p1 : p0;
  double* p_prev = (k & 1) ? p0 : p1;
  double* part_cur = part + (size_t)(k & 1) * 2 * T;
  double* part_other = part + (size_t)((k + 1) & 1) * 2 * T;
  hipLaunchKernelGGL(spcg_matvec_kernel, dim3(n_chunks), dim3(256), 0, s, S, ld, chunk_row, chunk_ptr, tcol, z, p_prev, p_cur, qpart, part_cur,
                     part_other, T, k == 0 ? 1 : 0, part_pq, sc, tol2);
  hipLaunchKernelGGL(spcg_update_kernel, dim3(T), dim3(256), 0, s, part_pq, n_chunks, row_chunk_ptr, qpart, Minv, p_cur, x, r, z, part_other, sc);
}
int spcg_chunk_tiles() { return kSpcgChunk; }
void launch_spcg_finish(hipStream_t s, int T, const double* x, const int* iperm, int n_pose, double* y_tan, double* delta) {
  hipLaunchKernelGGL(spcg_finish_kernel, dim3(T), dim3(64), 0, s, x, iperm, n_pose, y_tan, delta);
}

int pcg_spmv_grid(int nbr);
void launch_pcg_init(hipStream_t s, int nbr, const double* b, const double* Minv, double* x, double* r, double* z, double* p0, double* p1,
                     double* part, double* sc) {
  const int grid = pcg_rows_grid(nbr);
  hipLaunchKernelGGL(pcg_init_kernel, dim3(grid), dim3(256), 0, s, nbr, b, Minv, x, r, z, p0, p1, part);
  hipLaunchKernelGGL(pcg_init_scalars_kernel, dim3(1), dim3(256), 0, s, part, grid, sc);
}
// iteration k (0-based since the init): p buffers and the two halves of `part` (each 2 * ceil(nbr / 256) doubles) alternate
void launch_pcg_iteration(hipStream_t s, int k, int nbr, const int* row_ptr, const int* col, const double* val, const double* Minv,
                          double* x, double* r, double* z, double* p0, double* p1, double* q, double* part_pq, double* part, double* sc,
                          double tol2) {
  const int g16 = pcg_spmv_grid(nbr), g1 = pcg_rows_grid(nbr);
  double* p_cur = (k & 1) ? p1 : p0;
  double* p_prev = (k & 1) ? p0 : p1;
  double* part_cur = part + (size_t)(k & 1) * 2 * g1;
  double* part_other = part + (size_t)((k + 1) & 1) * 2 * g1;
  hipLaunchKernelGGL(pcg_spmv_kernel, dim3(g16), dim3(256), 0, s, nbr, row_ptr, col, val, z, p_prev, p_cur, q, part_cur, part_other, g1,
                     k == 0 ? 1 : 0, part_pq, sc, tol2);
  hipLaunchKernelGGL(pcg_update_kernel, dim3(g1), dim3(256), 0, s, nbr, part_pq, g16, Minv, p_cur, q, x, r, z, part_other, sc);
}
int pcg_spmv_grid(int nbr) { return (int)(((size_t)nbr * kSpmvLanes + 255) / 256); }
int pcg_num_scalars() { return PC_NUM; }
int pcg_done_slot() { return PC_DONE; }
int pcg_iters_slot() { return PC_ITERS; }

}  // namespace bsg
