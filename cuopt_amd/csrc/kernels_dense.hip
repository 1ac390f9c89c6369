// gfx950 kernels of the dense layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"

// rows of A, stage 1: one workgroup per chunk of a segment; lane <-> entry, values and vector are coalesced streams
__global__ void __launch_bounds__(kBlock)
k_dense_rows(DenseView D, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop)
{
  __shared__ double red[8];
  if (in_loop && !loop_active(ctl)) return;
  const double* __restrict__ vec = pick_vector(ctl, v0, v1, mode);
  const int sg = D.ch_seg[blockIdx.x], k0 = D.ch_k0[blockIdx.x];
  const int len = min(kDenseChunk, D.seg_len[sg] - k0);
  const double* __restrict__ a = D.val + D.seg_ptr[sg] + k0;
  const double* __restrict__ x = vec + D.seg_c0[sg] + k0;
  double c[kDenseChunk / kBlock];
#pragma unroll
  for (int u = 0; u < kDenseChunk / kBlock; ++u) {
    const int k = threadIdx.x + u * kBlock;
    c[u]        = k < len ? __builtin_nontemporal_load(a + k) * x[k] : 0.0;
  }
  double acc[1] = {0.0};
#pragma unroll
  for (int u = 0; u < kDenseChunk / kBlock; ++u) acc[0] += c[u];
  block_reduce<SumOp, 1>(acc, red);
  if (threadIdx.x == 0) D.ch_part[blockIdx.x] = acc[0];
}

// stage 2: a lane per owning row adds up its chunks in order
__global__ void __launch_bounds__(kBlock)
k_dense_rows_finish(DenseView D, int nrows, const pdlpdev_ctl* __restrict__ ctl, int in_loop, double* __restrict__ add)
{
  if (in_loop && !loop_active(ctl)) return;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nrows) return;
  double acc = 0.0;
  for (int q = D.row_ch[b]; q < D.row_ch[b + 1]; ++q) acc += D.ch_part[q];
  add[D.row[b]] = acc;
}

// rows of A^T (columns of A): lane <-> column of a 256-column tile; the segments that overlap the tile in ascending row order
__global__ void __launch_bounds__(kBlock)
k_dense_cols(DenseView D, int n, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode,
             int in_loop, double* __restrict__ add)
{
  if (in_loop && !loop_active(ctl)) return;
  const double* __restrict__ vec = pick_vector(ctl, v0, v1, mode);
  const int tile = D.tile_id[blockIdx.x];
  const int j    = tile * kBlock + (int)threadIdx.x;
  double acc     = 0.0;
  for (int q = D.tile_ptr[blockIdx.x]; q < D.tile_ptr[blockIdx.x + 1]; ++q) {
    const int sg = D.tile_seg[q];
    const int c0 = D.seg_c0[sg];
    if (j >= c0 && j < c0 + D.seg_len[sg]) acc += __builtin_nontemporal_load(D.val + D.seg_ptr[sg] + (j - c0)) * vec[D.seg_row[sg]];
  }
  if (j < n) add[j] = acc;
}
