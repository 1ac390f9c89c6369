// The stream layout's SpMV skeleton (device code).  Included by kernels_stream.hip only; the layout's view struct and constants, which the
// host side needs too, stay in pdlp_kernels.hpp.
#pragma once
#include "pdlp_kernels.hpp"

namespace pdlp {

// ------------------------------------------------------------------------------------------------
// CSR "stream" SpMV skeleton.  One workgroup owns the contiguous row range
// [row_blocks[b], row_blocks[b+1]) whose nonzeros (<= kNnzBlock unless it is a single long row)
// are loaded fully coalesced (lane i <-> nonzero k0+i), multiplied by the gathered vector entry
// and parked in LDS; then lane r adds up row r's products in CSR order and hands (row, sum) to the
// epilogue.  Epilogue concept:
//   struct E { static constexpr int NQ; using Op; __device__ void row(int r, double sum, double (&acc)[NQ]); }
// After the rows, acc[] is reduced over the workgroup and written to partials[q * nb + b].
// ------------------------------------------------------------------------------------------------
template <class Epi>
__device__ __forceinline__ void csr_stream_block(int nb, const int32_t* __restrict__ row_blocks,
                                                 const int32_t* __restrict__ offsets,
                                                 const int32_t* __restrict__ indices,
                                                 const double* __restrict__ values,
                                                 const double* __restrict__ vec, Epi& epi,
                                                 double* __restrict__ partials, const double* __restrict__ dense_add = nullptr,
                                                 int block = -1 /* >= 0: this row block (batched launches over several matrices) */)
{
  __shared__ __attribute__((aligned(32))) double prod[kNnzTile];
  __shared__ double red[4 * (Epi::NQ > 0 ? Epi::NQ : 1) + 4];
  const int b = block >= 0 ? block : xcd_remap(blockIdx.x, nb);
  if (b >= nb) return;
  const int r0 = row_blocks[b], r1 = row_blocks[b + 1];
  const int k0 = row_blocks[nb + 1 + b], k1 = row_blocks[nb + 2 + b];  // = offsets[r0], offsets[r1] (build_row_blocks)
  const int cnt = k1 - k0;
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();

  // The tile starts at k0 rounded DOWN to a multiple of 4 nonzeros, so every lane's 4 values / 4 indices are one
  // aligned 32-byte / 16-byte vector load (the arrays are padded by 8 entries; the up-to-3 foreign entries at
  // either end are multiplied like the others and never summed).  All loads of the workgroup are issued
  // before the first LDS write.  Measured on a banded 1e7-nnz matrix: 39.9 -> 29.5 us (59 % of the HBM roofline).
  const int base = k0 & ~3;
  if (cnt <= kNnzBlock) {
    constexpr int kPasses = kNnzTile / (4 * kBlock);
    // extents of the lane's first rows, requested together with the matrix stream: their latency would otherwise sit
    // between the barrier and the row sums (every workgroup of a mid-size LP is one dependent chain of round trips)
    constexpr int kPre = 4;
    int ext0[kPre], ext1[kPre];
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
      int r   = r0 + threadIdx.x + q * kBlock;
      r       = r < r1 ? r : r1 - 1;
      ext0[q] = offsets[r];
      ext1[q] = offsets[r + 1];
    }
    vec4d a[kPasses];
    vec4i j[kPasses];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int k = base + 4 * (p * kBlock + threadIdx.x);
      if (k < k1) {
        a[p] = __builtin_nontemporal_load(reinterpret_cast<const vec4d*>(values + k));
        j[p] = __builtin_nontemporal_load(reinterpret_cast<const vec4i*>(indices + k));
      } else {
        a[p] = (vec4d)(0.0);
        j[p] = (vec4i)(0);
      }
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const vec4d g = {vec[j[p].x], vec[j[p].y], vec[j[p].z], vec[j[p].w]};
      *reinterpret_cast<vec4d*>(&prod[4 * (p * kBlock + threadIdx.x)]) = a[p] * g;
    }
    __syncthreads();
    int q = 0;
    for (int r = r0 + threadIdx.x; r < r1; r += kBlock, ++q) {
      int s, e;
      switch (q) {  // register arrays want static indices
        case 0: s = ext0[0], e = ext1[0]; break;
        case 1: s = ext0[1], e = ext1[1]; break;
        case 2: s = ext0[2], e = ext1[2]; break;
        case 3: s = ext0[3], e = ext1[3]; break;
        default: s = offsets[r], e = offsets[r + 1]; break;
      }
      s -= base, e -= base;
      double sum = 0.0;
      if (e - s <= kLongRow) {
        for (int k = s; k < e; ++k) sum = sum + prod[k];
      } else {  // 4 interleaved chains: a medium-long row must not serialise the workgroup
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = s;
        for (; k + 3 < e; k += 4) {
          s0 += prod[k], s1 += prod[k + 1], s2 += prod[k + 2], s3 += prod[k + 3];
        }
        for (; k < e; ++k) s0 += prod[k];
        sum = (s0 + s1) + (s2 + s3);
      }
      epi.row(r, dense_plus(dense_add, r, sum), acc);
    }
  } else {
    // a single row longer than the LDS tile: strided partial sums + workgroup tree
    double part[1] = {0.0};
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
      const double a = __builtin_nontemporal_load(values + k);
      const int j    = __builtin_nontemporal_load(indices + k);
      part[0] += a * vec[j];
    }
    block_reduce<SumOp, 1>(part, red);
    if (threadIdx.x == 0) epi.row(r0, dense_plus(dense_add, r0, part[0]), acc);
    __syncthreads();
  }
  if constexpr (Epi::NQ > 0) {
    block_reduce<typename Epi::Op, Epi::NQ>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nb + b] = acc[q];
    }
  }
}

}  // namespace pdlp
