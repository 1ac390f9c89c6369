// Ruiz + Pock-Chambolle scaling of the device layer (split out of pdlp_device.hip in round 6; initial_scaling.cu:94-307): the row /
// column norm kernels in their three shapes (a lane per row, row blocks, long rows), the matrix scaling kernels, and the two entry
// points that enqueue them.
#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"
#include "pdlp_core_internal.hpp"

// ================================================================================================
// kernels: setup (scaling, norms).  One lane per row: these run a handful of times per solve.
// ================================================================================================
// Ruiz inf-norms of D_r A D_c.  Rows come from A, columns from A^T (so no atomics and a
// deterministic result; the value is the same as the reference's atomicMax version,
// initial_scaling.cu:94-122, because max is order independent).  TRANSPOSED selects which of
// (d_self, d_other) multiplies first so the product rounds exactly like (a * D_r) * D_c.
template <bool TRANSPOSED, bool POW>
__global__ void __launch_bounds__(kBlock) k_row_norm(int rows, const int32_t* __restrict__ off,
                                                     const int32_t* __restrict__ idx,
                                                     const double* __restrict__ val,
                                                     const double* __restrict__ d_row,
                                                     const double* __restrict__ d_col,
                                                     double exponent, double* __restrict__ out)
{
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
    if (off[r + 1] - off[r] > kLongRow) continue;  // k_row_norm_long: one lane walking thousands of nonzeros stalls its wave
    double acc = 0.0;
    for (int k = off[r]; k < off[r + 1]; ++k) {
      const int j = idx[k];
      double v;
      if (!TRANSPOSED)
        v = fabs((val[k] * d_row[r]) * d_col[j]);
      else
        v = fabs((val[k] * d_row[j]) * d_col[r]);
      if (POW)
        acc = acc + (exponent == 1.0 ? v : pow(v, exponent));  // Pock-Chambolle, :176-252
      else
        acc = v > acc ? v : acc;
    }
    out[r] = acc;
  }
}
// rows longer than kLongRow: one workgroup per row (the block-angular LP's 200 linking rows of 5000 nonzeros cost 3.9 ms per
// call, 42 ms of set-up, when a single lane walked each).  The maximum is order independent.  The Pock-Chambolle SUM stays
// bit-identical to the sequential one (the scaling vectors are compared with the oracle bit for bit): the workgroup loads and
// transforms 256 entries at a time, coalesced, into LDS, and ONE lane adds them up in order -- the adds are the only serial part.
template <bool TRANSPOSED, bool POW>
__global__ void __launch_bounds__(kBlock) k_row_norm_long(const int32_t* __restrict__ rows_long, const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                          const double* __restrict__ d_row, const double* __restrict__ d_col,
                                                          double exponent, double* __restrict__ out)
{
  __shared__ double buf[kBlock];
  __shared__ double red[8];
  const int r = rows_long[blockIdx.x], k1 = off[r + 1];
  double acc[1] = {0.0};
  for (int k0 = off[r]; k0 < k1; k0 += kBlock) {
    const int k = k0 + (int)threadIdx.x;
    double v    = 0.0;
    if (k < k1) {
      const int j = idx[k];
      v           = !TRANSPOSED ? fabs((val[k] * d_row[r]) * d_col[j]) : fabs((val[k] * d_row[j]) * d_col[r]);
    }
    if (POW) {
      buf[threadIdx.x] = exponent == 1.0 ? v : pow(v, exponent);
      __syncthreads();
      if (threadIdx.x == 0) {
        const int cnt = k1 - k0 < kBlock ? k1 - k0 : kBlock;
        for (int i = 0; i < cnt; ++i) acc[0] = acc[0] + buf[i];
      }
      __syncthreads();
    } else {
      acc[0] = v > acc[0] ? v : acc[0];
    }
  }
  if (!POW) block_reduce<MaxOp, 1>(acc, red);
  if (threadIdx.x == 0) out[r] = acc[0];
}
// The same norms with the matrix stream coalesced (round 3: the lane-per-row kernels above walk 12-byte entries 120 bytes apart
// and cost 240-275 us per call on a 1e7-nonzero matrix, 22 calls per solve = 5.7 ms of the set-up): a workgroup owns a row block of
// the stream layout (<= kNnzBlock nonzeros), lane <-> nonzero loads the value and the gathered scale factor into LDS, then lane <->
// row folds its entries in CSR order with the row's own factor -- the same products in the same order, so the same bits.
// Rows of more than kLongRow nonzeros are left to k_row_norm_long as before.
template <bool TRANSPOSED, bool POW>
__global__ void __launch_bounds__(kBlock) k_row_norm_blocks(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
                                                            const int32_t* __restrict__ idx, const double* __restrict__ val,
                                                            const double* __restrict__ d_row, const double* __restrict__ d_col,
                                                            double exponent, double* __restrict__ out)
{
  __shared__ double tp[kNnzTile];
  __shared__ double tq[TRANSPOSED ? 1 : kNnzTile];
  const int b = blockIdx.x;
  const int r0 = rb[b], r1 = rb[b + 1], k0 = rb[nb + 1 + b], k1 = rb[nb + 2 + b];
  if (k1 - k0 > kNnzBlock) return;  // a single row longer than the tile
  const double* __restrict__ other = TRANSPOSED ? d_row : d_col;
  for (int k = k0 + (int)threadIdx.x; k < k1; k += kBlock) {
    const double a = __builtin_nontemporal_load(val + k);
    const double g = other[__builtin_nontemporal_load(idx + k)];
    if (TRANSPOSED) tp[k - k0] = a * g;
    else tp[k - k0] = a, tq[k - k0] = g;
  }
  __syncthreads();
  const double* __restrict__ self = TRANSPOSED ? d_col : d_row;
  for (int r = r0 + (int)threadIdx.x; r < r1; r += kBlock) {
    const int a0 = off[r], a1 = off[r + 1];
    if (a1 - a0 > kLongRow) continue;
    const double ds = self[r];
    double acc      = 0.0;
    for (int k = a0; k < a1; ++k) {
      const double v = TRANSPOSED ? fabs(tp[k - k0] * ds) : fabs((tp[k - k0] * ds) * tq[k - k0]);
      if (POW)
        acc = acc + (exponent == 1.0 ? v : pow(v, exponent));
      else
        acc = v > acc ? v : acc;
    }
    out[r] = acc;
  }
}
// ... and the in-place scaling of the values (k_scale_matrix: 370 us per matrix): products formed by the row's lane in LDS, written
// back as a coalesced stream; entries of rows longer than kLongRow are left alone (k_scale_matrix_long)
__global__ void __launch_bounds__(kBlock) k_scale_matrix_blocks(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
                                                                const int32_t* __restrict__ idx, double* __restrict__ val,
                                                                const double* __restrict__ d_self, const double* __restrict__ d_other)
{
  __shared__ double tp[kNnzTile];
  __shared__ double tq[kNnzTile];
  const int b = blockIdx.x;
  const int r0 = rb[b], r1 = rb[b + 1], k0 = rb[nb + 1 + b], k1 = rb[nb + 2 + b];
  if (k1 - k0 > kNnzBlock) return;
  for (int k = k0 + (int)threadIdx.x; k < k1; k += kBlock) tp[k - k0] = val[k], tq[k - k0] = d_other[__builtin_nontemporal_load(idx + k)];
  __syncthreads();
  for (int r = r0 + (int)threadIdx.x; r < r1; r += kBlock) {
    const int a0 = off[r], a1 = off[r + 1];
    if (a1 - a0 > kLongRow) {
      for (int k = a0; k < a1; ++k) tq[k - k0] = -1.0;  // (scale factors are positive: "not mine")
      continue;
    }
    const double ds = d_self[r];
    for (int k = a0; k < a1; ++k) tp[k - k0] = tp[k - k0] * ds * tq[k - k0];
  }
  __syncthreads();
  for (int k = k0 + (int)threadIdx.x; k < k1; k += kBlock)
    if (tq[k - k0] != -1.0) val[k] = tp[k - k0];
}
// a_divides_sqrt_b_bounded, utils.cuh:122-129
__global__ void __launch_bounds__(kBlock) k_div_sqrt(int n, double* __restrict__ d,
                                                     const double* __restrict__ norm)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    if (norm[i] > 0.0) d[i] = d[i] / sqrt(norm[i]);
}
// scale the CSR values in place: A[k] = (A[k]*D_r[i])*D_c[j]; A^T[k] = (A^T[k]*D_c[j])*D_r[i]
// (two separate kernels in the reference, initial_scaling.cu:310-345, with exactly these orders)
__global__ void __launch_bounds__(kBlock) k_scale_matrix(int rows, const int32_t* __restrict__ off,
                                                         const int32_t* __restrict__ idx,
                                                         double* __restrict__ val,
                                                         const double* __restrict__ d_self,
                                                         const double* __restrict__ d_other)
{
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
    if (off[r + 1] - off[r] > kLongRow) continue;  // k_scale_matrix_long
    const double ds = d_self[r];
    for (int k = off[r]; k < off[r + 1]; ++k) val[k] = val[k] * ds * d_other[idx[k]];
  }
}
__global__ void __launch_bounds__(kBlock) k_scale_matrix_long(const int32_t* __restrict__ rows_long, const int32_t* __restrict__ off,
                                                              const int32_t* __restrict__ idx, double* __restrict__ val,
                                                              const double* __restrict__ d_self, const double* __restrict__ d_other)
{
  const int r     = rows_long[blockIdx.x];
  const double ds = d_self[r];
  for (int k = off[r] + (int)threadIdx.x; k < off[r + 1]; k += kBlock) val[k] = val[k] * ds * d_other[idx[k]];
}

extern "C" {

// ---- setup ----------------------------------------------------------------------------------------
int pdlpdev_scaling_compute(pdlpdev_ctx* ctx, int do_ruiz, int ruiz_iterations, int do_pc, double alpha)
{
  roctx::Range range("pdlp: Ruiz + Pock-Chambolle scaling");
  HIP_TRY(hipSetDevice(ctx->device));
  const int m = ctx->m, n = ctx->n;
  hipStream_t s = ctx->stream;
  k_fill<<<grid_for(m), kBlock, 0, s>>>(m, ctx->dr, 1.0);
  k_fill<<<grid_for(n), kBlock, 0, s>>>(n, ctx->dc, 1.0);
  auto pass = [&](bool pow_mode, double e_row, double e_col) -> int {
    // (the row blocks were cut on the hot CSR: usable when that is the full one)
    const bool blocks_a = ctx->ha_off == ctx->a_off && ctx->a_nb > 0, blocks_t = ctx->hat_off == ctx->at_off && ctx->at_nb > 0;
    if (!pow_mode) {
      if (blocks_a) k_row_norm_blocks<false, false><<<ctx->a_nb, kBlock, 0, s>>>(ctx->a_nb, ctx->a_rb, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_m);
      else
      k_row_norm<false, false><<<grid_for(m), kBlock, 0, s>>>(m, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_m);
      if (ctx->a_nlong) k_row_norm_long<false, false><<<ctx->a_nlong, kBlock, 0, s>>>(ctx->a_long, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_m);
      if (blocks_t) k_row_norm_blocks<true, false><<<ctx->at_nb, kBlock, 0, s>>>(ctx->at_nb, ctx->at_rb, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_n);
      else
      k_row_norm<true, false><<<grid_for(n), kBlock, 0, s>>>(n, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_n);
      if (ctx->at_nlong) k_row_norm_long<true, false><<<ctx->at_nlong, kBlock, 0, s>>>(ctx->at_long, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, 1.0, ctx->tmp_n);
    } else {
      if (blocks_a) k_row_norm_blocks<false, true><<<ctx->a_nb, kBlock, 0, s>>>(ctx->a_nb, ctx->a_rb, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, e_row, ctx->tmp_m);
      else
      k_row_norm<false, true><<<grid_for(m), kBlock, 0, s>>>(m, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, e_row, ctx->tmp_m);
      if (ctx->a_nlong) k_row_norm_long<false, true><<<ctx->a_nlong, kBlock, 0, s>>>(ctx->a_long, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc, e_row, ctx->tmp_m);
      if (blocks_t) k_row_norm_blocks<true, true><<<ctx->at_nb, kBlock, 0, s>>>(ctx->at_nb, ctx->at_rb, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, e_col, ctx->tmp_n);
      else
      k_row_norm<true, true><<<grid_for(n), kBlock, 0, s>>>(n, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, e_col, ctx->tmp_n);
      if (ctx->at_nlong) k_row_norm_long<true, true><<<ctx->at_nlong, kBlock, 0, s>>>(ctx->at_long, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dr, ctx->dc, e_col, ctx->tmp_n);
    }
    LAUNCH_CHECK();
    // row-block sharding: a column's norm is spread over the ranks
    TRY(allreduce(ctx, ctx->tmp_n, (size_t)n, pow_mode ? rccl::kSum : rccl::kMax));
    k_div_sqrt<<<grid_for(m), kBlock, 0, s>>>(m, ctx->dr, ctx->tmp_m);
    k_div_sqrt<<<grid_for(n), kBlock, 0, s>>>(n, ctx->dc, ctx->tmp_n);
    LAUNCH_CHECK();
    return 0;
  };
  if (do_ruiz)
    for (int it = 0; it < ruiz_iterations; ++it) TRY(pass(false, 0, 0));
  if (do_pc) TRY(pass(true, alpha, 2.0 - alpha));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int pdlpdev_scale_problem(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->scaled) return fail(-1, "problem already scaled");
  hipStream_t s = ctx->stream;
  if (ctx->ha_off == ctx->a_off && ctx->a_nb > 0) k_scale_matrix_blocks<<<ctx->a_nb, kBlock, 0, s>>>(ctx->a_nb, ctx->a_rb, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc);
  else
  k_scale_matrix<<<grid_for(ctx->m), kBlock, 0, s>>>(ctx->m, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc);
  if (ctx->a_nlong) k_scale_matrix_long<<<ctx->a_nlong, kBlock, 0, s>>>(ctx->a_long, ctx->a_off, ctx->a_idx, ctx->a_val, ctx->dr, ctx->dc);
  if (ctx->hat_off == ctx->at_off && ctx->at_nb > 0) k_scale_matrix_blocks<<<ctx->at_nb, kBlock, 0, s>>>(ctx->at_nb, ctx->at_rb, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dc, ctx->dr);
  else
  k_scale_matrix<<<grid_for(ctx->n), kBlock, 0, s>>>(ctx->n, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dc, ctx->dr);
  if (ctx->at_nlong) k_scale_matrix_long<<<ctx->at_nlong, kBlock, 0, s>>>(ctx->at_long, ctx->at_off, ctx->at_idx, ctx->at_val, ctx->dc, ctx->dr);
  k_scale_vectors<<<grid_for(std::max(ctx->m, ctx->n)), kBlock, 0, s>>>(ctx->n, ctx->m, ctx->c, ctx->lb, ctx->ub, ctx->dc, ctx->lo, ctx->hi, ctx->dr);
  LAUNCH_CHECK();
  ctx->scaled = true;
  TRY(sync_panel_values(ctx));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
