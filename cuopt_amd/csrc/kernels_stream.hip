// gfx950 kernels of the stream layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "pdlp_layouts.hpp"
#include "spmv_stream.hpp"

__global__ void __launch_bounds__(kBlock)
k_spmv_a_dual(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
              double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
              const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
              const p2pdev::Push* __restrict__ push, const double* __restrict__ dadd)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size,
                 ctl->pending_avg != 0, ycopy, push};
  csr_stream_block(nb, rb, off, idx, val, xbar, e, part, dadd);
  if (push) p2pdev::count_exchange(push);
}

__global__ void __launch_bounds__(kBlock)
k_spmv_at_step(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
               const int32_t* __restrict__ idx, const double* __restrict__ val,
               const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, const double* __restrict__ x0,
               const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ part, const double* __restrict__ dadd)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  csr_stream_block(nb, rb, off, idx, val, cur ? y0 : y1 /* y' */, e, part, dadd);
}

// plain SpMV (A^T y at start / after restart-to-average; parity hook; multi-GPU partial products)
__global__ void __launch_bounds__(kBlock)
k_spmv_plain(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
             const int32_t* __restrict__ idx, const double* __restrict__ val,
             const double* __restrict__ vec, double* __restrict__ out, const double* __restrict__ dadd)
{
  StoreEpilogue e{out};
  csr_stream_block(nb, rb, off, idx, val, vec, e, nullptr, dadd);
}

// variants that pick the ping-pong buffer on the device
__global__ void __launch_bounds__(kBlock)
k_spmv_at_cur(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ out_override, int use_next, const double* __restrict__ dadd)
{
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  csr_stream_block(nb, rb, off, idx, val, cur ? y1 : y0, e, nullptr, dadd);
}

// the same for several contexts in one launch (the small-LP batch, kernels_resident.hip)
__global__ void __launch_bounds__(kBlock) k_spmv_at_cur_batch(const StreamAtCurArgs* __restrict__ args, const int2* __restrict__ blk)
{
  const int2 b            = blk[blockIdx.x];
  const StreamAtCurArgs A = args[b.x];
  const int cur           = A.ctl->cur;
  StoreEpilogue e{cur ? A.aty1 : A.aty0};
  csr_stream_block(A.nb, A.rb, A.off, A.idx, A.val, cur ? A.y1 : A.y0, e, nullptr, nullptr, b.y);
}
int launch_stream_at_cur_batch(hipStream_t s, const StreamAtCurArgs* args, const int2* blk, int blocks)
{
  k_spmv_at_cur_batch<<<blocks, kBlock, 0, s>>>(args, blk);
  HIP_TRY(hipGetLastError());
  return 0;
}

__global__ void __launch_bounds__(kBlock)
k_eval_primal(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0,
              const double* __restrict__ x1, const double* __restrict__ avgx,
              const double* __restrict__ y0, const double* __restrict__ y1,
              const double* __restrict__ avgy, const double* __restrict__ dr,
              const double* __restrict__ lo_u, const double* __restrict__ hi_u, double eps_rel,
              double* __restrict__ linf_rows, double* __restrict__ ax_out, double* __restrict__ part, const double* __restrict__ dadd)
{
  const int cur = ctl->cur;
  const double* xv = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalPrimalEpilogue e{yv, dr, lo_u, hi_u, eps_rel, linf_rows, ax_out};
  csr_stream_block(nb, rb, off, idx, val, xv, e, part, dadd);
}

__global__ void __launch_bounds__(kBlock)
k_eval_dual(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
            const int32_t* __restrict__ idx, const double* __restrict__ val,
            const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0,
            const double* __restrict__ x1, const double* __restrict__ avgx,
            const double* __restrict__ y0, const double* __restrict__ y1,
            const double* __restrict__ avgy, EvalDualCore core, double* __restrict__ part, const double* __restrict__ dadd)
{
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalDualEpilogue e{core};
  csr_stream_block(nb, rb, off, idx, val, yv, e, part, dadd);
}

// ================================================================================================
// host side of the layout
// ================================================================================================
// ================================================================================================
// host side of the device layer
// ================================================================================================
// Greedy partition of the rows into stream blocks: at most kNnzBlock nonzeros and
// kMaxRowsPerBlock rows per block; a row longer than the LDS tile gets a block of its own.
std::vector<int32_t> build_row_blocks(int32_t rows, const int32_t* off)
{
  const int64_t tile = kNnzBlock;  // (smaller tiles for small LPs were measured: no gain at 1e6 nnz)
  std::vector<int32_t> rb;
  rb.push_back(0);
  int32_t start = 0;
  // rows while they fit the tile (at most kMaxRowsPerBlock): `end` = the last offset within off[start] + tile.  Not a walk over every
  // row (2 ms at 1e6 rows, twice) and not a plain binary search either (5 000 blocks x 10 dependent misses on a cold 4 MB array were
  // 2 ms as well): a guess from the mean row length, corrected by a short walk inside the same cache lines; a bisection only where
  // the guess is far off (skewed row lengths).
  const int64_t nnz = off[rows];
  const int32_t step = (int32_t)std::max<int64_t>(1, std::min<int64_t>(kMaxRowsPerBlock, tile * (int64_t)rows / std::max<int64_t>(nnz, 1)));
  while (start < rows) {
    const int32_t last = (int32_t)std::min<int64_t>(rows, (int64_t)start + kMaxRowsPerBlock);
    const int64_t lim  = (int64_t)off[start] + tile;
    int32_t end = std::min(last, start + step);
    int walked  = 0;
    while (end < last && (int64_t)off[end + 1] <= lim && walked < 64) ++end, ++walked;
    while (end > start && (int64_t)off[end] > lim && walked < 64) --end, ++walked;
    if (walked >= 64)  // far off: bisect
      end = (int32_t)(std::upper_bound(off + start, off + last + 1, lim, [](int64_t v, int32_t o) { return v < (int64_t)o; }) - off) - 1;
    if (end == start) end = start + 1;  // long row: alone
    rb.push_back(end);
    start = end;
  }
  // second half: the nonzero position where each block starts (off[rb[b]]), so that a workgroup learns its row range
  // AND its nonzero range in one round trip instead of two dependent ones
  const size_t nb1 = rb.size();
  rb.resize(2 * nb1);
  for (size_t b = 0; b < nb1; ++b) rb[nb1 + b] = off[rb[b]];
  return rb;
}
