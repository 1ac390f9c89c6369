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

}  // namespace pdlp
