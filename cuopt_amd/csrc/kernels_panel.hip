// gfx950 kernels of the panel layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"
#include "spmv_panel.hpp"

// panel-layout twins of (2) and (3): same epilogues, slab-major gather (pdlp_kernels.hpp)
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_a_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size,
                 ctl->pending_avg != 0, ycopy, push};
  panel_block<SEG>(P, xbar, e, part);
  if (push) p2pdev::count_exchange(push);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_step(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  panel_block<SEG>(P, cur ? y0 : y1 /* y' */, e, part);
}

// panel twin of k_spmv_at_cur (A^T y of the iterate / of the trial iterate, optionally into `out_override`)
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_cur(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next)
{
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  panel_block<SEG>(P, cur ? y1 : y0, e, nullptr);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_plain(PanelView P, const double* __restrict__ vec, double* __restrict__ out)
{
  StoreEpilogue e{out};
  panel_block<SEG>(P, vec, e, nullptr);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_primal(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part)
{
  const int cur = ctl->cur;
  const double* xv = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalPrimalEpilogue e{yv, dr, lo_u, hi_u, eps_rel, linf_rows, ax_out};
  panel_block<SEG>(P, xv, e, part);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part)
{
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalDualEpilogue e{core};
  panel_block<SEG>(P, yv, e, part);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_panel_a_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
template __global__ void k_panel_a_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
template __global__ void k_panel_at_step<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
template __global__ void k_panel_at_step<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
template __global__ void k_panel_at_cur<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
template __global__ void k_panel_at_cur<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
template __global__ void k_panel_plain<true>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_panel_plain<false>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_panel_eval_primal<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_panel_eval_primal<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_panel_eval_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
template __global__ void k_panel_eval_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);

// ================================================================================================
// host side of the layout
// ================================================================================================
int64_t gather_working_set(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx)
{
  const int64_t nnz = off[rows], window = 512 * 1024;
  if (rows <= 0 || cols <= 0 || nnz <= 0) return 0;
  const int samples = nnz <= window ? 1 : (int)std::min<int64_t>(4, (nnz + window - 1) / window);
  std::vector<uint8_t> seen((size_t)(cols >> 4) + 1);
  int64_t total = 0;
  for (int s = 0; s < samples; ++s) {
    const int64_t first = samples == 1 ? 0 : (nnz - window) * s / (samples - 1);
    const int64_t last  = std::min(nnz, first + window);
    std::fill(seen.begin(), seen.end(), 0);
    int64_t lines = 0;
    for (int64_t k = first; k < last; ++k) {
      uint8_t& b = seen[(size_t)(idx[k] >> 4)];
      lines += !b;
      b = 1;
    }
    total += lines * 128;
  }
  return total / samples;
}

// the same estimate from the windows alone (`sparse` holds the index entries of the windows back to back): for a matrix whose index
// array lives on the device (kernels_setup.hip analysis_fetch_idx_windows cuts the windows exactly as above)
int64_t gather_working_set_windows(int32_t cols, const int32_t* sparse, const std::vector<std::pair<int64_t, int64_t>>& windows)
{
  if (windows.empty() || cols <= 0) return 0;
  std::vector<uint8_t> seen((size_t)(cols >> 4) + 1);
  int64_t total = 0;
  size_t at = 0;
  for (const auto& w : windows) {
    std::fill(seen.begin(), seen.end(), 0);
    int64_t lines = 0;
    for (int64_t k = w.first; k < w.second; ++k, ++at) {
      uint8_t& b = seen[(size_t)(sparse[at] >> 4)];
      lines += !b;
      b = 1;
    }
    total += lines * 128;
  }
  return total / (int64_t)windows.size();
}

// Geometry, long-tail decision, own rows and the cut into panels: everything build_panels decides from the ROW OFFSETS alone (shared
// with the device-side construction, kernels_setup.hip build_panels_device).  false: the layout does not apply.
bool panel_plan(PanelHost* out, int32_t rows, int32_t cols, const int32_t* off, int64_t slab_bytes, bool force,
                const std::vector<int32_t>* dense_first_seg, std::vector<char>* is_own_out, int64_t* own_nnz_out, int64_t* own_from_out)
{
  PanelHost& P = *out;
  const int64_t nnz = rows > 0 ? off[rows] : 0;
  if (rows <= 0 || cols <= 0 || nnz <= 0) return false;
  // worth it only when the gathered vector overflows an XCD's L2 (4 MiB, shared with the matrix stream)
  if (!force && (int64_t)cols * 8 <= 2 * (int64_t)1048576) return false;
  int S = (int)(((int64_t)cols * 8 + slab_bytes - 1) / slab_bytes);
  S     = std::max(1, std::min(S, 16));
  const int32_t slab_w = (cols + S - 1) / S;
  // panels: two 512-thread workgroups per CU = 512 resident panels, and ALL panels should be resident at once (they walk
  // the slabs in lockstep; a 513th panel runs alone afterwards: the power-law LP with 11.4 M nonzeros took 104 us per SpMV
  // with 570 panels of 20 K nonzeros and takes 75 with 512 of 22 K).  So the panel size follows the matrix, up to
  // 60 K nonzeros (beyond that the row-sum strip, kPanelMaxRows, and the 16-bit tile pointers set the limits).
  const int64_t cap = std::max<int64_t>(2048, cuopt_amd::tune_int("panel_nnz", 60000));  // (tests cut small matrices into many panels)
  // A panel takes rows while they fit under the target (a row above the target is a panel of its own): all panels run at
  // once, so the LARGEST one sets the kernel time -- letting a panel overshoot by its last row made panels of 42 K nonzeros
  // next to the average 22 K on the power-law LP (rows of up to 20 000 nonzeros) and cost 26 of its 117 us.  The target grows
  // (proportionally first, then in 1 % steps) until the panels fit the 512 resident slots again.
  // Long-tailed row lengths: row sums dealt by nonzero (panel_seg_block; every row at rtol 1e-12 instead of bit-exact short rows).
  // auto: when more than 2 % of the nonzeros sit in rows of more than kLongRow entries -- a structural, reproducible rule;
  // CUOPT_AMD_TUNE=panel_seg=0|1 forces it off / on (tests, sweeps).
  int64_t long_nnz = 0;  // nonzeros in rows the row-per-lane kernel sums wave by wave
  int32_t max_len  = 0;
  {
    // (one pass over the offsets on the host pool's threads: at 1e6 rows every serial pass of this function was ~1 ms of the set-up)
    constexpr int kParts = 16;
    int64_t part_long[kParts] = {0};
    int32_t part_max[kParts]  = {0};
    cuopt_amd::parallel_tasks(kParts, [&](int t) {
      const int32_t a = (int32_t)((int64_t)rows * t / kParts), b = (int32_t)((int64_t)rows * (t + 1) / kParts);
      int64_t ln = 0;
      int32_t mx = 0;
      for (int32_t i = a; i < b; ++i) {
        const int32_t len = off[i + 1] - off[i];
        if (len > kLongRow) ln += len;
        mx = std::max(mx, len);
      }
      part_long[t] = ln, part_max[t] = mx;
    }, (int64_t)rows * 8);
    for (int t = 0; t < kParts; ++t) long_nnz += part_long[t], max_len = std::max(max_len, part_max[t]);
  }
  {
    const long long want = cuopt_amd::tune_int("panel_seg", -1);
    P.seg = want == 1 || (want != 0 && long_nnz * 50 > nnz);
    if (slab_w > (1 << kSegColBits)) P.seg = false;
  }
  // Rows beyond kPanelOwnRow nonzeros get a workgroup each behind the panels (a lane, or a wave, would walk them for ever).  The
  // long-tail variant deals every row by nonzero, so a row leaves the panels only when it is longer than a whole panel should be
  // (it would be the one panel everybody waits for): up to there it is ordinary work, and no resident slot is spent on it.
  const int64_t own_from = P.seg ? std::max<int64_t>(kPanelOwnRow, std::min<int64_t>(cap, (nnz + 511) / 512)) : kPanelOwnRow;
  std::vector<char>& is_own = *is_own_out;
  is_own.assign((size_t)rows, 0);
  int64_t own_nnz = 0;
  const bool any_own = max_len > own_from || dense_first_seg != nullptr;
  if (any_own)
    for (int32_t i = 0; i < rows; ++i)
      if (off[i + 1] - off[i] > own_from || (dense_first_seg && (*dense_first_seg)[i] >= 0))  // (rows that own dense segments: their
        is_own[i] = 1, P.own_row.push_back(i), own_nnz += off[i + 1] - off[i];                // workgroup adds the segments too)
  auto cut = [&](int64_t tgt) {
    P.row0.assign(1, 0);
    int32_t start = 0;
    if (!any_own) {
      // every row counts: the greedy rule below as a search over the offsets (same panels: rows are taken while they fit under the
      // target, a first non-empty row above the target is taken alone)
      while (start < rows) {
        const int32_t last = (int32_t)std::min<int64_t>(rows, (int64_t)start + kPanelMaxRows);
        const int64_t lim  = (int64_t)off[start] + tgt;
        int32_t end = (int32_t)(std::upper_bound(off + start, off + last + 1, lim, [](int64_t v, int32_t o) { return v < (int64_t)o; }) - off) - 1;
        if (end < last && off[end] == off[start]) ++end;  // (nothing but empty rows so far: the row that does not fit goes in alone)
        end = std::max(end, start + 1);
        P.row0.push_back(end);
        start = end;
      }
      return;
    }
    while (start < rows) {
      int32_t end = start;
      int64_t cnt = 0;
      while (end < rows && end - start < kPanelMaxRows) {
        const int64_t len = is_own[end] ? 0 : off[end + 1] - off[end];
        if (cnt > 0 && cnt + len > tgt) break;
        cnt += len;
        ++end;
      }
      P.row0.push_back(end);
      start = end;
    }
  };
  int64_t tgt = std::max<int64_t>(2048, std::min<int64_t>(cap, (nnz - own_nnz + 511) / 512));
  cut(tgt);
  // (the rows with a workgroup of their own share the 512 resident slots with the panels: behind a full house they would run alone)
  const int slots = std::max(64, 512 - (int)P.own_row.size());
  tgt             = std::max<int64_t>(2048, std::min<int64_t>(cap, (nnz - own_nnz + slots - 1) / slots));
  cut(tgt);
  for (int it = 0; it < 64 && (int)P.row0.size() - 1 > slots && tgt < cap; ++it) {
    const int64_t w = (int64_t)P.row0.size() - 1;
    tgt = std::min<int64_t>(cap, it == 0 ? (int64_t)((double)tgt * (double)w / (double)slots * 1.002) + 1 : tgt + tgt / 100 + 1);
    cut(tgt);
  }
  const int W = (int)P.row0.size() - 1;
  P.W = W, P.S = S, P.slab_w = slab_w;
  if (!any_own) P.any_long = max_len > kLongRow;
  else for (int32_t i = 0; i < rows && !P.any_long; ++i) P.any_long = !is_own[i] && off[i + 1] - off[i] > kLongRow;
  P.own_ptr.assign((size_t)W + 1, 0);
  for (int w = 0, q = 0; w < W; ++w) {
    while (q < (int)P.own_row.size() && P.own_row[q] < P.row0[w + 1]) ++q;
    P.own_ptr[w + 1] = q;
  }
  *own_nnz_out = own_nnz, *own_from_out = own_from;
  return true;
}

PanelHost build_panels(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx,
                              int64_t slab_bytes, bool force, const std::vector<int32_t>* dense_first_seg)
{
  PanelHost P;
  std::vector<char> is_own;
  int64_t own_nnz = 0, own_from = 0;
  if (!panel_plan(&P, rows, cols, off, slab_bytes, force, dense_first_seg, &is_own, &own_nnz, &own_from)) return P;
  const int64_t nnz = off[rows];
  const int W = P.W, S = P.S;
  const int32_t slab_w = P.slab_w;
  // pass 1: nonzeros per (panel, slab) -- panels are independent, so both passes run over host threads
  std::vector<int64_t> count((size_t)W * S + 1, 0);
  cuopt_amd::parallel_tasks(W, [&](int w) {
    int64_t* cw = &count[(size_t)w * S];
    for (int32_t i = P.row0[w]; i < P.row0[w + 1]; ++i)
      if (!is_own[i])
        for (int64_t t = off[i]; t < off[i + 1]; ++t) cw[idx[t] / slab_w] += 1;
  }, nnz);
  P.tile_ptr.resize((size_t)W * S + 1);
  int64_t pos = 0;
  for (size_t i = 0; i < (size_t)W * S; ++i) {
    if (!P.seg && count[i] >= 65536) return P;  // 16-bit row pointers would overflow: keep the CSR stream layout
    P.tile_ptr[i] = (int32_t)pos;
    pos += count[i];
  }
  P.tile_ptr[(size_t)W * S] = (int32_t)pos;
  // pass 2: placement + per-tile row pointers
  P.nnz = (size_t)(nnz - own_nnz), P.rowptr_size = P.seg ? 0 : (size_t)S * ((size_t)rows + W);
  P.perm.reset(P.nnz), P.col.reset(P.nnz);
  if (P.seg) {
    // placement in (slab, row, CSR) order as below, every entry carrying its row within the panel next to its column inside the slab
    cuopt_amd::parallel_tasks(W, [&](int w) {
      const int32_t a = P.row0[w], b = P.row0[w + 1];
      std::vector<int32_t> cursor(S);
      for (int s2 = 0; s2 < S; ++s2) cursor[s2] = P.tile_ptr[(size_t)w * S + s2];
      for (int32_t i = a; i < b; ++i) {
        if (is_own[i]) continue;
        for (int32_t t = off[i]; t < off[i + 1]; ++t) {
          const int s2    = idx[t] / slab_w;
          const int32_t q = cursor[s2]++;
          P.perm[q] = t, P.col[q] = (int32_t)(((uint32_t)(i - a) << kSegColBits) | (uint32_t)(idx[t] - s2 * slab_w));
        }
      }
    }, nnz);
    P.ok = true;
    return P;
  }
  P.rowptr.reset(P.rowptr_size);
  P.rp_base.resize((size_t)W * S);
  cuopt_amd::parallel_tasks(W, [&](int w) {
    const int32_t a = P.row0[w], b = P.row0[w + 1], nr = b - a;
    const int64_t rp = (int64_t)S * ((int64_t)a + w);  // rowptr entries of all earlier panels
    std::vector<int32_t> cursor(S);
    for (int s2 = 0; s2 < S; ++s2) {
      cursor[s2]                    = P.tile_ptr[(size_t)w * S + s2];
      P.rp_base[(size_t)w * S + s2] = rp + (int64_t)s2 * (nr + 1);
    }
    for (int32_t i = a; i < b; ++i) {
      for (int s2 = 0; s2 < S; ++s2)
        P.rowptr[(size_t)(P.rp_base[(size_t)w * S + s2] + (i - a))] = (uint16_t)(cursor[s2] - P.tile_ptr[(size_t)w * S + s2]);
      if (is_own[i]) continue;
      for (int32_t t = off[i]; t < off[i + 1]; ++t) {
        const int s2 = idx[t] / slab_w;
        const int32_t q = cursor[s2]++;
        P.perm[q] = t, P.col[q] = idx[t];
      }
    }
    for (int s2 = 0; s2 < S; ++s2)
      P.rowptr[(size_t)(P.rp_base[(size_t)w * S + s2] + nr)] = (uint16_t)(cursor[s2] - P.tile_ptr[(size_t)w * S + s2]);
  }, nnz);
  P.ok = true;
  return P;
}

int upload_panels(pdlpdev_ctx* c, pdlpdev_ctx::Panels* dst, const PanelHost& h, const int32_t* d_off, const int32_t* d_idx,
                         const double* d_val)
{
  if (!h.ok) return 0;
  int32_t *row0 = nullptr, *tile_ptr = nullptr, *col = nullptr;
  uint16_t* rowptr = nullptr;
  int64_t* rp_base = nullptr;
  TRY(upload_i32(c, &row0, h.row0.data(), h.row0.size()));
  TRY(upload_i32(c, &tile_ptr, h.tile_ptr.data(), h.tile_ptr.size()));
  TRY(upload_i32(c, &col, h.col.get(), h.nnz));
  TRY(upload_i32(c, &dst->perm, h.perm.get(), h.nnz));
  TRY(dev_alloc(c, &rowptr, h.rowptr_size));
  HIP_TRY(hipMemcpyAsync(rowptr, h.rowptr.get(), h.rowptr_size * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
  TRY(dev_alloc(c, &rp_base, h.rp_base.size()));
  HIP_TRY(hipMemcpyAsync(rp_base, h.rp_base.data(), h.rp_base.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  TRY(dev_alloc(c, &dst->val, h.nnz));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors die with the caller's PanelHost
  dst->v  = PanelView{h.W, h.S, h.any_long ? 1 : 0, row0, tile_ptr, rowptr, rp_base, col, dst->val};
  dst->v.seg = h.seg ? 1 : 0, dst->v.slab_w = h.slab_w;
  dst->nent = (int64_t)h.nnz;
  if (!h.own_row.empty()) {  // W becomes the number of workgroups / partials: the panels, then a workgroup per own row
    int32_t *own_row = nullptr, *own_ptr = nullptr;
    TRY(upload_i32(c, &own_row, h.own_row.data(), h.own_row.size()));
    TRY(upload_i32(c, &own_ptr, h.own_ptr.data(), h.own_ptr.size()));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dst->v.NP = h.W, dst->v.W = h.W + (int)h.own_row.size();
    dst->v.own_row = own_row, dst->v.own_ptr = own_ptr;
    dst->v.csr_off = d_off, dst->v.csr_idx = d_idx, dst->v.csr_val = d_val;
  }
  dst->on = true;
  return 0;
}

// CUOPT_AMD_SPMV_LAYOUT=timed: both layouts of a matrix are timed on the device (plain SpMV, 1 warm-up + 3 launches each)
// and the slower one is dropped.  This is how the structural rule of 'auto' (gather_working_set) was calibrated; it is
// not the default because two close timings make the choice -- and with it the grouping of the reduction partials, the
// step sizes and the iteration count -- differ from run to run.
int pick_layout(pdlpdev_ctx* c, pdlpdev_ctx::Panels* pn, int rows, int nb, const int32_t* rb, const int32_t* off,
                       const int32_t* idx, const double* val, const double* vec, double* out, const char* name)
{
  if (!pn->on) return 0;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float ms_stream = 1e30f, ms_panel = 1e30f;
  for (int round = 0; round < 2; ++round)  // interleaved rounds, minimum per layout: robust to one-off stalls
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 4; ++rep) {
        if (rep == 1) HIP_TRY(hipEventRecord(e0, c->stream));
        if (which == 0)
          k_spmv_plain<<<stream_grid(nb), kBlock, 0, c->stream>>>(nb, rb, off, idx, val, vec, out, (const double*)nullptr);
        else
          (pn->v.seg ? k_panel_plain<true> : k_panel_plain<false>)<<<pn->v.W, kPanelThreads, 0, c->stream>>>(pn->v, vec, out);
      }
      HIP_TRY(hipEventRecord(e1, c->stream));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      if (which == 0)
        ms_stream = std::min(ms_stream, ms);
      else
        ms_panel = std::min(ms_panel, ms);
    }
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  if (getenv("CUOPT_AMD_TIMING"))
    fprintf(stderr, "[cuopt_amd setup]   layout %-3s: stream %.1f us, panels %.1f us -> %s\n", name, ms_stream * 1e3 / 3,
            ms_panel * 1e3 / 3, ms_stream <= ms_panel ? "stream" : "panels");
  if (ms_stream <= ms_panel) pn->on = false;
  (void)rows;
  return 0;
}
