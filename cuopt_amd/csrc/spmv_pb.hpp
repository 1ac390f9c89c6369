// The pb layout's SpMV skeleton (device code).  Included by kernels_pb.hip only; the layout's view struct and constants, which the
// host side needs too, stay in pdlp_kernels.hpp.
#pragma once
#include "pdlp_kernels.hpp"

namespace pdlp {

// LDS hazards of the gather-free kernels (round-6 audit): no stage is double-buffered.  Phase P fills its slice xs[] of the vector,
// barrier, reads only.  Phase R parks the bin's image lp[] from registers (requested before the row descriptors), barrier, sums its
// rows from it into strip[] (one lane per row), barrier, streams the epilogue, barrier, reuses lp[] as reduction scratch.  The products
// travel from P to R through HBM across a kernel boundary.
template <int THREADS>
__device__ __forceinline__ void pb_products_block(const PbView& V, const double* __restrict__ vec, double* xs)
{
  constexpr int U = 4;
  const int w     = xcd_remap((int)blockIdx.x, V.nwg);
  if (w >= V.nwg) return;
  const int panel = V.wg_panel[w];
  const int c0    = panel << V.panel_shift;
  const int len   = min(1 << V.panel_shift, V.cols - c0);
  constexpr int kFill = 8;
  for (int b0 = 0; b0 < len; b0 += kFill * THREADS) {
    double v[kFill];
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int i = b0 + u * THREADS + (int)threadIdx.x;
      v[u]        = i < len ? vec[c0 + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int i = b0 + u * THREADS + (int)threadIdx.x;
      if (i < len) xs[i] = v[u];
    }
  }
  __syncthreads();
  const int e0 = V.wg_e0[w], e1 = V.wg_e0[w + 1];
  const int gmask = (1 << V.gshift) - 1;
  for (int e = e0 + 2 * (int)threadIdx.x; e < e1; e += 2 * THREADS * U) {
    pb_vec2d a[U];
    uint32_t j[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      a[u] = (pb_vec2d)(0.0), j[u] = 0, dst[u] = 0;
      if (ee < e1) {
        a[u]   = __builtin_nontemporal_load(reinterpret_cast<const pb_vec2d*>(V.val + ee));
        j[u]   = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(V.lidx + ee));
        dst[u] = __builtin_nontemporal_load(V.piece_dst + (ee >> V.gshift));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      if (ee < e1) {
        pb_vec2d p;
        p.x = a[u].x * xs[j[u] & 0xFFFFu];
        p.y = a[u].y * xs[j[u] >> 16];
        *reinterpret_cast<pb_vec2d*>(V.prod + (((int64_t)dst[u] << V.gshift) + (ee & gmask))) = p;
      }
    }
  }
}

// phase R of one workgroup (kPbThreads threads, kPbLdsBytes of dynamic LDS at `lds`)
template <class Epi>
__device__ __forceinline__ void pb_rows_block(const PbView& V, Epi& epi, double* __restrict__ partials, double* lds)
{
  constexpr int THREADS = kPbThreads, WAVES = THREADS / 64, GR = kPbMaxRows / 64 / WAVES, U = (kPbCap / 2 + THREADS - 1) / THREADS;
  static_assert(GR * WAVES * 64 == kPbMaxRows, "whole groups per wave");
  double* lp       = lds;
  double* strip    = lds + kPbCap + 128;
  const int lane   = threadIdx.x & 63;
  const int wave   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b      = xcd_remap((int)blockIdx.x, V.B);
  if (b >= V.B) return;
  const int e0     = V.bin_e0[b];
  const int nunits = (V.bin_e0[b + 1] - e0) >> 1;  // 16-byte units of the image
  const double* __restrict__ prod = V.prod;
  pb_vec2d stage[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int un = threadIdx.x + u * THREADS;
    stage[u]     = (pb_vec2d)(0.0);
    if (un < nunits) stage[u] = __builtin_nontemporal_load(reinterpret_cast<const pb_vec2d*>(prod + e0 + 2 * un));
  }
  const int row0  = V.bin_row0[b];
  const int brows = V.bin_row0[b + 1] - row0;
  const int ng    = (brows + 63) >> 6;
  uint32_t d[GR];
  int eg[GR];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int g = wave + q * WAVES;
    const int i = g * 64 + lane;
    d[q]        = i < brows ? V.sr[row0 + i] : 0u;
    eg[q]       = g < ng ? V.grp_pos[V.bin_grp[b] + g] : 0;
  }
  uint32_t p[GR][kPbKU];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int cnt = (int)(d[q] >> 16);
    int e         = __builtin_amdgcn_readfirstlane(eg[q]);
#pragma unroll
    for (int u = 0; u < kPbKU; ++u) {
      const int at = e;
      e += __builtin_popcountll(__ballot(cnt > u));
      p[q][u] = 0;
      if (cnt > u) p[q][u] = __builtin_nontemporal_load(V.pos + at + lane);
    }
    eg[q] = e;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int un = threadIdx.x + u * THREADS;
    if (un < nunits) *reinterpret_cast<pb_vec2d*>(lp + 2 * un) = stage[u];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int g = wave + q * WAVES;
    if (g < ng) {
      const int cnt  = (int)(d[q] >> 16);
      const int lrow = (int)(d[q] & 0xFFFFu);
      double sum     = 0.0;
#pragma unroll
      for (int u = 0; u < kPbKU; ++u)
        if (cnt > u) sum = sum + lp[p[q][u]];
      const int kmax = __builtin_amdgcn_readfirstlane(cnt);  // sorted: lane 0 holds the longest row of the group
      int e          = eg[q];
      for (int k = kPbKU; k < kmax; ++k) {  // rows longer than the prefetched diagonals
        const int at = e;
        e += __builtin_popcountll(__ballot(cnt > k));
        if (cnt > k) sum = sum + lp[V.pos[at + lane]];
      }
      if (g * 64 + lane < brows) strip[lrow] = sum;
    }
  }
  __syncthreads();
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  for (int i = threadIdx.x; i < brows; i += THREADS) epi.row(row0 + i, dense_plus(V.dense_add, row0 + i, strip[i]), acc);
  if constexpr (Epi::NQ > 0) {
    __syncthreads();  // every wave is done with the image: its first bytes become the reduction scratch
    block_reduce<typename Epi::Op, Epi::NQ, WAVES>(acc, lp);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * V.B + b] = acc[q];
    }
  }
}

// phase R of one WIDE bin (kPbwThreads threads, kPbwLdsBytes of dynamic LDS at `lds`).  LDS hazards: acc[] is zeroed, barrier; a step's
// phase reads and writes only accumulators of pairwise different rows (the levels), a barrier ends every phase; the epilogue reads acc[]
// behind the last phase's barrier (serial rows: one lane writes the row's accumulator behind the last step, a barrier follows); the
// reduction scratch is a region of its own.  Products and slot words are requested kPbwAhead steps
// ahead into a ring of registers (straight-line code, counted vmcnt waits; the barriers leave vector-memory loads in flight).
template <class Epi>
__device__ __forceinline__ void pbw_rows_block(const PbView& V, Epi& epi, double* __restrict__ partials, double* lds)
{
  constexpr int T = kPbwThreads, WAVES = T / 64, PE = kPbwAhead;
  double* acc   = lds;
  double* red   = lds + kPbwRows;                                   // [256] reduction scratch
  uint8_t* lvs  = reinterpret_cast<uint8_t*>(lds + kPbwRows + 256);  // [kPbwMaxSteps]
  const int b   = xcd_remap((int)blockIdx.x, V.B);
  if (b >= V.B) return;
  const int tid   = (int)threadIdx.x;
  const int e0    = V.bin_e0[b];
  const int ns    = (V.bin_e0[b + 1] - e0) >> 10;
  const int row0  = V.bin_row0[b];
  const int brows = V.bin_row0[b + 1] - row0;
  const double* __restrict__ prod = V.prod + e0 + tid;
  // (the slot words travel as the 32-bit word that holds the lane's own and its neighbour's: a 16-bit load would be widened by a separate
  //  instruction that the optimiser sinks away from the load -- to the loop's latch, behind a vmcnt(0))
  const uint32_t* __restrict__ rib = reinterpret_cast<const uint32_t*>(V.rib + e0) + (tid >> 1);
  const int half = (tid & 1) * 16;
  double rv[PE];
  uint32_t rr[PE];
#pragma unroll
  for (int d = 0; d < PE; ++d) {
    rv[d] = __builtin_nontemporal_load(prod + (size_t)d * kPbwStep);
    rr[d] = __builtin_nontemporal_load(rib + (size_t)d * (kPbwStep / 2));
  }
  for (int i = tid; i < kPbwRows; i += T) acc[i] = 0.0;
  {
    const uint8_t* lv = V.step_lv + (e0 >> 10);
#pragma unroll 1
    for (int i = tid; i < ns; i += T) lvs[i] = lv[i];
  }
  __syncthreads();
  int s0 = 0;
  do {  // (a do-while and no exit from the middle of the ring: the loop's head has one incoming state, the counted waits stay counted)
#pragma unroll
    for (int d = 0; d < PE; ++d) {
      const int s        = s0 + d;
      const bool live    = s < ns;  // (uniform; the ring's tail requests read the next bin's slots, or the slack behind the last one)
      const uint32_t w   = (rr[d] >> half) & 0xFFFFu;
      const double p     = rv[d];
      const int row      = (int)(w & (kPbwRows - 1));
      const unsigned lvl = w >> 13;
      const int top      = __builtin_amdgcn_readfirstlane(live ? (int)lvs[s] : 0);
      if (live && lvl == 0u) acc[row] = acc[row] + p;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int f = 1; f <= top; ++f) {  // (uniform trip count; no load inside)
        if (lvl == (unsigned)f) acc[row] = acc[row] + p;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      // (behind the step's last use of the slot: the ring's registers are rewritten in place)
      rv[d] = __builtin_nontemporal_load(prod + (size_t)(s + PE) * kPbwStep);
      rr[d] = __builtin_nontemporal_load(rib + (size_t)(s + PE) * (kPbwStep / 2));
      __builtin_amdgcn_sched_barrier(0);
    }
    s0 += PE;
  } while (s0 < ns);
  if (V.nser) {  // (uniform) serial rows: no step touched their accumulators; the products were written by phase P's kernel
    for (int q = V.ser_ptr[b] + tid; q < V.ser_ptr[b + 1]; q += T) {
      double sum   = 0.0;
      const int e1 = V.ser_eptr[q + 1];
      for (int e = V.ser_eptr[q]; e < e1; e += 8) {  // (eight scattered products in flight, added in order)
        double pr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pr[u] = e + u < e1 ? V.prod[V.ser_slot[e + u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) sum = e + u < e1 ? sum + pr[u] : sum;
      }
      acc[V.ser_row[q] - row0] = sum;
    }
    __syncthreads();
  }
  double accq[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) accq[q] = Epi::Op::identity();
  for (int i = tid; i < brows; i += T) epi.row(row0 + i, dense_plus(V.dense_add, row0 + i, acc[i]), accq);
  if constexpr (Epi::NQ > 0) {
    block_reduce<typename Epi::Op, Epi::NQ, WAVES>(accq, red);
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * V.B + b] = accq[q];
    }
  }
}

}  // namespace pdlp
