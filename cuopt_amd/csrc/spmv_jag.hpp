// The jag layout's SpMV skeleton (device code).  Included by kernels_jag.hip only; the layout's view struct and constants, which the
// host side needs too, stay in pdlp_kernels.hpp.
#pragma once
#include "pdlp_kernels.hpp"

namespace pdlp {

// LDS hazards of jag_block (round-6 audit): nothing is double-buffered.  The column set xwin[] is filled once and is read-only behind
// the first barrier; the strip psum[] is zeroed before that barrier, each of its entries is then written by exactly one lane (the
// lane of the row's pass; rows of their own workgroups get their mark from one thread, never a pass), and read by the epilogue behind
// the second barrier; the reduction scratch reuses xwin[] behind a third barrier.  No barrier inside the loop over the diagonals.
template <class Epi, int WAVES>
__device__ __forceinline__ void jag_block(const JagView& J, const double* __restrict__ vec, Epi& epi,
                                          double* __restrict__ partials)
{
  extern __shared__ __attribute__((aligned(16))) double jag_lds[];
  double* xwin   = jag_lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // grid: the row blocks (blockIdx & 7 = XCD -> a contiguous range of row blocks per XCD), padded to a multiple of 8, then the
  // long rows round-robin over the XCDs (inside one xcd_remap over both, the last XCDs would get long rows only);
  // partials keep the order [row blocks..., long rows...]
  const int nparts = J.nblk + J.nlong;
  const int nb8    = (J.nblk + 7) & ~7;
  int blk;
  if ((int)blockIdx.x < nb8) {
    blk = xcd_remap((int)blockIdx.x, J.nblk);
    if (blk >= J.nblk) return;
  } else {
    blk = J.nblk + ((int)blockIdx.x - nb8);
    if (blk >= nparts) return;
  }
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  if (blk >= J.nblk) {
    // ---- a row longer than kLongRow: the whole workgroup strides over it (fixed tree, compared with a tolerance like
    // every long row of the other layouts); such rows are spread over the grid instead of queueing up in one wave ----
    const int r  = J.lr_row[blk - J.nblk];
    const int k0 = J.off[r], k1 = J.off[r + 1];
    double part[1] = {0.0};
    // 16 entries per thread in flight: a row of up to 8192 nonzeros costs two dependent round trips (entries, then gathers),
    // not two per 2048 -- these workgroups run behind the row blocks and their latency is the kernel's tail
    constexpr int kLongU = 16;
    for (int k = k0 + (int)threadIdx.x; k < k1; k += kLongU * (WAVES * 64)) {
      double a[kLongU];
      int j[kLongU];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) {
        a[u] = 0.0, j[u] = 0;
        if (k + u * (WAVES * 64) < k1) {
          a[u] = __builtin_nontemporal_load(J.csr_val + k + u * (WAVES * 64));
          j[u] = __builtin_nontemporal_load(J.idx + k + u * (WAVES * 64));
        }
      }
      double xv[kLongU];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) xv[u] = k + u * (WAVES * 64) < k1 ? vec[j[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < kLongU; ++u) part[0] = part[0] + a[u] * xv[u];
    }
    block_reduce<SumOp, 1, WAVES>(part, xwin);
    if (threadIdx.x == 0) epi.row(r, dense_plus(J.dense_add, r, part[0]), acc);
    if constexpr (Epi::NQ > 0) {
      if (threadIdx.x == 0) {  // one row: thread 0's accumulators are the workgroup's
#pragma unroll
        for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nparts + blk] = acc[q];
      }
    }
    return;
  }
  double* psum        = jag_lds + jag_window(WAVES);  // row sums of the workgroup's rows, natural order
  const int g         = blk * WAVES + wave;
  const int row0      = J.row0[blk];
  const int brows     = J.row0[blk + 1] - row0;
  const int wbase     = J.win[2 * blk];
  const unsigned wlen = (unsigned)J.win[2 * blk + 1];
  // fill the LDS column set, eight requests per thread in flight (a plain loop pays one round trip per element)
  constexpr int kFill = 8, T = WAVES * 64;
  if (wlen) {
    for (unsigned b0 = 0; b0 < wlen; b0 += kFill * T) {
      double v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const unsigned i = b0 + u * T + threadIdx.x;
        v[u] = i < wlen ? vec[wbase + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const unsigned i = b0 + u * T + threadIdx.x;
        if (i < wlen) xwin[i] = v[u];
      }
    }
  } else {
    const int s0 = J.set_ptr[blk], ns = J.set_ptr[blk + 1] - s0;
    for (int b0 = 0; b0 < ns; b0 += kFill * T) {
      int c[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int i = b0 + u * T + (int)threadIdx.x;
        c[u] = i < ns ? __builtin_nontemporal_load(J.set_col + s0 + i) : -1;
      }
      double v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) v[u] = c[u] >= 0 ? vec[c[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int i = b0 + u * T + (int)threadIdx.x;
        if (i < ns) xwin[i] = v[u];
      }
    }
  }
  for (int i = threadIdx.x; i < brows; i += (WAVES * 64)) psum[i] = 0.0;  // rows without nonzeros
  __syncthreads();
  // rows longer than kLongRow belong to their own workgroups (above): mark them so that the epilogue skips them.  AFTER the
  // barrier: the zero fill above touches the same strip entries from other waves; the passes below never write these entries
  // (they hold rows of <= kLongRow nonzeros only) and the barrier before the epilogue publishes the marks.
  for (int q = J.lr_ptr[blk] + (int)threadIdx.x; q < J.lr_ptr[blk + 1]; q += (WAVES * 64))
    psum[J.lr_row[q] - row0] = __longlong_as_double(kJagNotMine);
  {
    int e         = __builtin_amdgcn_readfirstlane(J.tile_e[g]);
    const int sr0 = __builtin_amdgcn_readfirstlane(J.tile_sr[g]);
    const int ns  = __builtin_amdgcn_readfirstlane(J.tile_sr[g + 1]) - sr0;
    for (int p0 = 0; p0 < ns; p0 += 64) {
      const int i      = p0 + lane;
      const bool have  = i < ns;
      const unsigned d = have ? J.sr[sr0 + i] : 0u;
      const int cnt    = have ? (int)(d >> 16) + 1 : 0;
      const int lrow   = (int)(d & 0xFFFFu);
      double sum       = 0.0;
      const int kmax   = __builtin_amdgcn_readfirstlane(cnt);  // sorted: lane 0 holds the longest row of the pass
      for (int k0 = 0; k0 < kmax; k0 += kJagU) {
        int at[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) {  // diagonal k holds one entry per row longer than k: a prefix of the lanes
          at[u] = e;
          e += __builtin_popcountll(__ballot(cnt > k0 + u));
        }
        double a[kJagU];
        unsigned j[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) {
          a[u] = 0.0, j[u] = 0;
          if (cnt > k0 + u) {
            a[u] = __builtin_nontemporal_load(J.val + at[u] + lane);
            j[u] = __builtin_nontemporal_load(J.slot + at[u] + lane);
          }
        }
        // every gather is an LDS read.  Lanes past their row's end add +0.0 * 0.0: a sum that started at +0.0 is never
        // -0.0, so this changes no bit
        double xv[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) xv[u] = cnt > k0 + u ? xwin[j[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < kJagU; ++u) sum = sum + a[u] * xv[u];
      }
      if (have) psum[lrow] = sum;
    }
  }
  __syncthreads();  // the strip is complete: the fused epilogue streams the workgroup's rows in natural order
  for (int i = threadIdx.x; i < brows; i += (WAVES * 64)) {
    const int row = row0 + i;
    if (row < J.rows && __double_as_longlong(psum[i]) != kJagNotMine) epi.row(row, dense_plus(J.dense_add, row, psum[i]), acc);
  }
  if constexpr (Epi::NQ > 0) {
    __syncthreads();  // every wave is done with the window: its first bytes become the reduction scratch
    block_reduce<typename Epi::Op, Epi::NQ, WAVES>(acc, xwin);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nparts + blk] = acc[q];
    }
  }
}

}  // namespace pdlp
