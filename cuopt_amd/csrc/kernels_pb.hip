// gfx950 kernels of the pb layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "pdlp_layouts.hpp"
#include "spmv_pb.hpp"

// gather-free twins: phase P (one kernel, the gathered vector chosen on the device like the other layouts do) and phase R with
// the same epilogues
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_pb_products(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (in_loop && !loop_active(ctl)) return;
  // mode 0: v0.  1: cur ? v0 : v1 (the trial iterate of a ping-pong pair).  2: cur ? v1 : v0 (the current one).
  const double* vec = v0;
  if (mode != 0) {
    const bool cur = ctl->cur != 0;
    vec            = (cur == (mode == 1)) ? v0 : v1;
  }
  pb_products_block<THREADS>(V, vec, pb_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_a_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size, ctl->pending_avg != 0, ycopy, push};
  if constexpr (WIDE) pbw_rows_block(V, e, part, pb_lds);
  else pb_rows_block(V, e, part, pb_lds);
  if (push) p2pdev::count_exchange(push);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_at_step(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  if constexpr (WIDE) pbw_rows_block(V, e, part, pb_lds);
  else pb_rows_block(V, e, part, pb_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_at_cur(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  if constexpr (WIDE) pbw_rows_block(V, e, nullptr, pb_lds);
  else pb_rows_block(V, e, nullptr, pb_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads) k_pb_plain(PbView V, double* __restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  StoreEpilogue e{out};
  if constexpr (WIDE) pbw_rows_block(V, e, nullptr, pb_lds);
  else pb_rows_block(V, e, nullptr, pb_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_eval_primal(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur    = ctl->cur;
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalPrimalEpilogue e{yv, dr, lo_u, hi_u, eps_rel, linf_rows, ax_out};
  if constexpr (WIDE) pbw_rows_block(V, e, part, pb_lds);
  else pb_rows_block(V, e, part, pb_lds);
}

template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_eval_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  EvalDualEpilogue e{core};
  if constexpr (WIDE) pbw_rows_block(V, e, part, pb_lds);
  else pb_rows_block(V, e, part, pb_lds);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_pb_products<512>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
template __global__ void k_pb_products<1024>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
template __global__ void k_pb_a_dual<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push);
template __global__ void k_pb_a_dual<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push);
template __global__ void k_pb_at_step<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part);
template __global__ void k_pb_at_step<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part);
template __global__ void k_pb_at_cur<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next);
template __global__ void k_pb_at_cur<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next);
template __global__ void k_pb_plain<false>(PbView V, double* __restrict__ out);
template __global__ void k_pb_plain<true>(PbView V, double* __restrict__ out);
template __global__ void k_pb_eval_primal<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part);
template __global__ void k_pb_eval_primal<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part);
template __global__ void k_pb_eval_dual<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part);
template __global__ void k_pb_eval_dual<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part);

// ================================================================================================
// host side of the layout
// ================================================================================================
PbHost build_pb(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int cus, bool forced)
{
  if (pb_wants_wide(cols)) {  // wide bins first; a matrix they cannot hold goes on to the image-in-LDS bins
    PbHost W = build_pb_wide(rows, cols, off, idx, cus, forced);
    if (W.ok) return W;
  }
  PbHost H;
  H.rows = rows, H.cols = cols;
  const int64_t nnz = rows > 0 ? off[rows] : 0;
  H.nnz = nnz;
  if (rows <= 0 || cols <= 0 || nnz <= 0) { H.why = "empty matrix"; return H; }
  int longest = 0;
  for (int32_t r = 0; r < rows; ++r) longest = std::max(longest, off[r + 1] - off[r]);
  // a row is summed by ONE lane, left to right: fine up to a few hundred entries, a serial chain beyond
  if (longest > (forced ? kPbCap / 2 : 256)) { H.why = "a row with " + std::to_string(longest) + " nonzeros"; return H; }
  H.panel_shift = cols > (1 << 21) ? 14 : 13;
  H.p_threads   = H.panel_shift == 14 ? 1024 : 512;
  const int SP  = 1 << H.panel_shift;
  const int S   = (cols + SP - 1) >> H.panel_shift;
  H.S           = S;
  const int threads = cuopt_amd::host_threads();
  // bins: consecutive rows, <= target nonzeros and <= kPbMaxRows rows; the target is lowered until every bin's padded image fits
  double target = 0.93 * kPbCap;
  int G = 8;
  std::vector<int32_t> row0;
  for (int iter = 0; iter < 24; ++iter) {
    row0.assign(1, 0);
    while (row0.back() < rows) {
      const int32_t r0 = row0.back();
      const int64_t lim = (int64_t)off[r0] + (int64_t)target;
      int32_t r1 = (int32_t)(std::upper_bound(off + r0, off + rows + 1, (int32_t)std::min<int64_t>(lim, nnz)) - off) - 1;
      r1 = std::min(std::max(r1, r0 + 1), std::min(rows, r0 + kPbMaxRows));
      row0.push_back(r1);
    }
    const int B = (int)row0.size() - 1;
    if (iter == 0) G = nnz / ((int64_t)S * B) >= 24 ? 8 : 4;
    std::vector<int> worst(threads * 4, 0);
    cuopt_amd::parallel_tasks((int)worst.size(), [&](int t) {
      std::vector<int> cnt(S, 0);
      std::vector<int> touched;
      int w = 0;
      for (int b = t; b < B; b += (int)worst.size()) {
        touched.clear();
        int padded = 0;
        for (int k = off[row0[b]]; k < off[row0[b + 1]]; ++k) {
          const int s_ = idx[k] >> H.panel_shift;
          if (cnt[s_] % G == 0) padded += G;
          if (cnt[s_]++ == 0) touched.push_back(s_);
        }
        for (int s_ : touched) cnt[s_] = 0;
        w = std::max(w, padded);
      }
      worst[t] = w;
    }, nnz);
    const int maxpad = *std::max_element(worst.begin(), worst.end());
    if (maxpad <= kPbCap) break;
    if (iter == 23) { H.why = "bins do not converge"; return H; }
    target *= std::min(0.97, 0.99 * (double)kPbCap / (double)maxpad);
  }
  H.gshift   = G == 8 ? 3 : 2;
  H.bin_row0 = row0;
  const int B = (int)row0.size() - 1;
  H.B         = B;
  // chunk sizes (bin-major), the bins' images, P order (panel-major) starts
  cuopt_amd::PoolArray<int32_t> cnt((size_t)B * S), lstart((size_t)B * S);
  H.bin_e0.assign(B + 1, 0);
  std::vector<int32_t> bin_size(B);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    for (int b = t; b < B; b += threads * 4) {
      int32_t* c = cnt.get() + (size_t)b * S;
      std::fill(c, c + S, 0);
      for (int k = off[row0[b]]; k < off[row0[b + 1]]; ++k) c[idx[k] >> H.panel_shift]++;
      int32_t at = 0;
      int32_t* l = lstart.get() + (size_t)b * S;
      for (int s_ = 0; s_ < S; ++s_) {
        l[s_] = at;
        at += (c[s_] + G - 1) / G * G;
      }
      bin_size[b] = at;
    }
  }, nnz);
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    H.bin_e0[b] = (int32_t)total;
    total += bin_size[b];
    if (total >= ((int64_t)1 << 31) - 65536) { H.why = "more than 2^31 padded entries"; return H; }
  }
  H.bin_e0[B] = (int32_t)total;
  H.np        = total;
  cuopt_amd::PoolArray<int32_t> pstart((size_t)S * B + 1);
  {
    int64_t at = 0;
    for (int s_ = 0; s_ < S; ++s_)
      for (int b = 0; b < B; ++b) {
        pstart[(size_t)s_ * B + b] = (int32_t)at;
        at += (cnt[(size_t)b * S + s_] + G - 1) / G * G;
      }
    pstart[(size_t)S * B] = (int32_t)at;
  }
  H.perm.reset((size_t)total + 64), H.lidx.reset((size_t)total + 64), H.piece_dst.reset((size_t)(total >> H.gshift) + 64);
  H.pos.reset((size_t)nnz + 128), H.sr.reset((size_t)rows + 64);
  H.bin_grp.assign(B + 1, 0);
  for (int b = 0; b < B; ++b) H.bin_grp[b + 1] = H.bin_grp[b] + (row0[b + 1] - row0[b] + 63) / 64;
  H.grp_pos.assign((size_t)H.bin_grp[B] + 1, 0);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    // padding slots first (a chunk's tail), then the entries
    for (int64_t i = (int64_t)t * total / (threads * 4), e = (int64_t)(t + 1) * total / (threads * 4); i < e; ++i) H.perm[i] = -1, H.lidx[i] = 0;
  }, total);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    std::vector<int32_t> cur(S);
    std::vector<uint16_t> epos;
    std::vector<int32_t> order;
    for (int b = t; b < B; b += threads * 4) {
      const int32_t r0 = row0[b], nr = row0[b + 1] - r0, k0 = off[r0];
      std::fill(cur.begin(), cur.end(), 0);
      epos.resize((size_t)(off[r0 + nr] - k0));
      const int32_t* l = lstart.get() + (size_t)b * S;
      for (int32_t r = r0; r < r0 + nr; ++r)
        for (int k = off[r]; k < off[r + 1]; ++k) {
          const int s_     = idx[k] >> H.panel_shift;
          const int rank   = cur[s_]++;
          const int32_t pp = pstart[(size_t)s_ * B + b] + rank;
          H.perm[pp]       = k;
          H.lidx[pp]       = (uint16_t)(idx[k] & (SP - 1));
          epos[k - k0]     = (uint16_t)(l[s_] + rank);
        }
      // pieces of this bin's chunks
      for (int s_ = 0; s_ < S; ++s_) {
        const int np_ = (cnt[(size_t)b * S + s_] + G - 1) / G;
        const int32_t p0 = pstart[(size_t)s_ * B + b] >> H.gshift, d0 = (H.bin_e0[b] + l[s_]) >> H.gshift;
        for (int i = 0; i < np_; ++i) H.piece_dst[p0 + i] = d0 + i;
      }
      // rows sorted by length (descending, stable), groups of 64, jagged diagonals of positions
      order.resize(nr);
      for (int i = 0; i < nr; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return off[r0 + x + 1] - off[r0 + x] > off[r0 + y + 1] - off[r0 + y]; });
      int32_t at = k0;  // the bin's positions start where its nonzeros start
      for (int i = 0; i < nr; ++i) H.sr[r0 + i] = ((uint32_t)(off[r0 + order[i] + 1] - off[r0 + order[i]]) << 16) | (uint32_t)order[i];
      for (int g0 = 0, g = 0; g0 < nr; g0 += 64, ++g) {
        H.grp_pos[H.bin_grp[b] + g] = at;
        const int g1   = std::min(nr, g0 + 64);
        const int kmax = off[r0 + order[g0] + 1] - off[r0 + order[g0]];
        for (int k = 0; k < kmax; ++k)
          for (int i = g0; i < g1; ++i) {
            const int32_t r = r0 + order[i];
            if (off[r + 1] - off[r] <= k) break;
            H.pos[at++] = epos[off[r] + k - k0];
          }
      }
    }
  }, nnz);
  H.grp_pos[H.bin_grp[B]] = (int32_t)nnz;
  for (int i = 0; i < 128; ++i) H.pos[(size_t)nnz + i] = 0;
  // P workgroups: every panel's entries in Q parts (pieces are not split)
  const int Q = std::max(1, std::min(16, (4 * cus + S - 1) / S));
  for (int s_ = 0; s_ < S; ++s_) {
    const int64_t e0 = pstart[(size_t)s_ * B], e1 = pstart[(size_t)(s_ + 1) * B];
    const int64_t per = std::max<int64_t>(G, ((e1 - e0 + Q - 1) / Q + G - 1) / G * G);
    for (int64_t e = e0; e < e1; e += per) {
      H.wg_e0.push_back((int32_t)e);
      H.wg_panel.push_back(s_);
    }
  }
  H.wg_e0.push_back((int32_t)total);
  H.ok = true;
  return H;
}

// ---- wide bins: the accumulators in LDS, the image streamed (pdlp_kernels.hpp: kPbw*) -----------------------------------------
bool pb_wants_wide(int32_t cols)
{
  const long long t = cuopt_amd::tune_int("pb_wide", -1);
  if (t >= 0) return t != 0;
  // 16384-column panels are taken from 2 M columns on: from there the (panel, image-in-LDS bin) chunks are shorter than ~32 entries
  return cols > (1 << 21);
}

int pbw_piece_shift(int64_t nnz, int S, int B) { return nnz / ((int64_t)S * B) >= 96 ? 4 : 3; }

PbHost build_pb_wide(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int cus, bool forced)
{
  PbHost H;
  H.wide = true;
  H.rows = rows, H.cols = cols;
  const int64_t nnz = rows > 0 ? off[rows] : 0;
  H.nnz = nnz;
  if (rows <= 0 || cols <= 0 || nnz <= 0) { H.why = "empty matrix"; return H; }
  int longest = 0;
  for (int32_t r = 0; r < rows; ++r) longest = std::max(longest, off[r + 1] - off[r]);
  // (a row of L entries inside one panel takes ceil(L / (kPbwMaxLevel + 1)) steps of its bin: fine for the unstructured matrices this
  //  layout is for, hopeless for long rows -- those stay with the image-in-LDS bins or the panels)
  if (longest > (forced ? kPbCap / 2 : 256)) { H.why = "a row with " + std::to_string(longest) + " nonzeros"; return H; }
  H.panel_shift = cols > (1 << 21) ? 14 : 13;
  H.p_threads   = H.panel_shift == 14 ? 1024 : 512;
  const int SP  = 1 << H.panel_shift;
  const int S   = (cols + SP - 1) >> H.panel_shift;
  H.S           = S;
  const int threads = cuopt_amd::host_threads();
  const int B       = (rows + kPbwRows - 1) / kPbwRows;
  H.B               = B;
  // pieces of 16 entries (whole 128-byte lines for phase P's stores) unless the chunks are shorter than 96 entries (then 8: at 2e9
  // nonzeros -- chunks of 74 -- 16-entry pieces pad by 11 % and push the image past 2^31 entries)
  H.gshift    = pbw_piece_shift(nnz, S, B);
  const int G = 1 << H.gshift;
  H.bin_row0.resize((size_t)B + 1);
  for (int b = 0; b <= B; ++b) H.bin_row0[b] = (int32_t)std::min<int64_t>(rows, (int64_t)b * kPbwRows);
  const std::vector<int32_t>& row0 = H.bin_row0;
  // chunk sizes (bin-major), the bins' images (whole steps), P order (panel-major) starts
  cuopt_amd::PoolArray<int32_t> cnt((size_t)B * S), lstart((size_t)B * S);
  H.bin_e0.assign((size_t)B + 1, 0);
  std::vector<int32_t> bin_size(B);
  std::vector<int> too_long(threads * 4, 0);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    for (int b = t; b < B; b += threads * 4) {
      int32_t* c = cnt.get() + (size_t)b * S;
      std::fill(c, c + S, 0);
      for (int k = off[row0[b]]; k < off[row0[b + 1]]; ++k) c[idx[k] >> H.panel_shift]++;
      int64_t at = 0;
      int32_t* l = lstart.get() + (size_t)b * S;
      for (int s_ = 0; s_ < S; ++s_) {
        l[s_] = (int32_t)at;
        at += (c[s_] + G - 1) / G * G;
        if (c[s_] > 65535) too_long[t] = 1;
      }
      at = (at + kPbwStep - 1) / kPbwStep * kPbwStep;
      if (at > (int64_t)kPbwMaxSteps * kPbwStep) too_long[t] = 1, at = 0;
      bin_size[b] = (int32_t)at;
    }
  }, nnz);
  if (*std::max_element(too_long.begin(), too_long.end())) { H.why = "a chunk of more than 65535 entries or a bin of more than 4096 steps"; return H; }
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    H.bin_e0[b] = (int32_t)total;
    total += bin_size[b];
    if (total >= ((int64_t)1 << 31) - 65536) { H.why = "more than 2^31 padded entries"; return H; }
  }
  H.bin_e0[B] = (int32_t)total;
  H.np        = total;
  cuopt_amd::PoolArray<int32_t> pstart((size_t)S * B + 1);
  int64_t ptotal = 0;
  {
    for (int s_ = 0; s_ < S; ++s_)
      for (int b = 0; b < B; ++b) {
        pstart[(size_t)s_ * B + b] = (int32_t)ptotal;
        ptotal += (cnt[(size_t)b * S + s_] + G - 1) / G * G;
      }
    pstart[(size_t)S * B] = (int32_t)ptotal;
  }
  // (P order holds the chunks only -- ptotal entries -- the image also the bins' rounding to whole steps: ptotal <= total)
  H.perm.reset((size_t)total + 64), H.lidx.reset((size_t)total + 64), H.piece_dst.reset((size_t)(total >> H.gshift) + 64);
  H.rib.reset((size_t)total + 64);
  H.step_lv.assign((size_t)(total >> 10) + 1, 0);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    for (int64_t i = (int64_t)t * total / (threads * 4), e = (int64_t)(t + 1) * total / (threads * 4); i < e; ++i) H.perm[i] = -1, H.lidx[i] = 0, H.rib[i] = 0xFFFFu;
    if (t == 0) for (int64_t i = 0; i < (total >> H.gshift); ++i) H.piece_dst[i] = 0;
  }, total);
  // per bin: the serial rows it found (a row with more than kPbwMaxLevel + 1 entries inside one step) and their entries' slots
  std::vector<std::vector<int32_t>> bin_ser(B);
  std::vector<std::vector<std::vector<int32_t>>> bin_ser_slots(B);
  cuopt_amd::parallel_tasks(threads * 4, [&](int t) {
    std::vector<int32_t> cur(S);
    std::vector<int32_t> stamp(kPbwRows), seen(kPbwRows), slot_of;
    std::vector<char> serial(kPbwRows);
    for (int b = t; b < B; b += threads * 4) {
      const int32_t r0 = row0[b], nr = row0[b + 1] - r0, k0 = off[r0];
      std::fill(cur.begin(), cur.end(), 0);
      slot_of.resize((size_t)(off[r0 + nr] - k0));
      const int32_t* l = lstart.get() + (size_t)b * S;
      for (int32_t r = r0; r < r0 + nr; ++r)
        for (int k = off[r]; k < off[r + 1]; ++k) {
          const int s_     = idx[k] >> H.panel_shift;
          const int rank   = cur[s_]++;
          const int32_t pp = pstart[(size_t)s_ * B + b] + rank;
          H.perm[pp]       = k;
          H.lidx[pp]       = (uint16_t)(idx[k] & (SP - 1));
          slot_of[k - k0]  = H.bin_e0[b] + l[s_] + rank;
          H.rib[(size_t)slot_of[k - k0]] = (uint16_t)(r - r0);
        }
      for (int s_ = 0; s_ < S; ++s_) {
        const int np_ = (cnt[(size_t)b * S + s_] + G - 1) / G;
        const int32_t p0 = pstart[(size_t)s_ * B + b] >> H.gshift, d0 = (H.bin_e0[b] + l[s_]) >> H.gshift;
        for (int i = 0; i < np_; ++i) H.piece_dst[p0 + i] = d0 + i;
      }
      // levels: how many earlier slots of the same step carry the same row (a row beyond the last level: serial)
      std::fill(stamp.begin(), stamp.end(), -1);
      std::fill(serial.begin(), serial.end(), 0);
      bool any = false;
      for (int64_t st = H.bin_e0[b] >> 10; st < (H.bin_e0[b + 1] >> 10); ++st) {
        int top = 0;
        for (int i = 0; i < kPbwStep; ++i) {
          uint16_t& w = H.rib[(size_t)st * kPbwStep + i];
          if (w == 0xFFFFu) continue;
          const int row = w;
          if (stamp[row] != (int32_t)st) stamp[row] = (int32_t)st, seen[row] = 0;
          const int lv = seen[row]++;
          if (lv > kPbwMaxLevel) { serial[row] = 1, any = true; continue; }
          top = std::max(top, lv);
          w   = (uint16_t)(row | lv << 13);
        }
        H.step_lv[(size_t)st] = (uint8_t)top;
      }
      if (any)
        for (int i = 0; i < nr; ++i)
          if (serial[i]) {
            bin_ser[b].push_back(r0 + i);
            bin_ser_slots[b].emplace_back();
            for (int k = off[r0 + i]; k < off[r0 + i + 1]; ++k) {
              H.rib[(size_t)slot_of[k - k0]] = 0xFFFFu;  // its slots read as padding: the steps never touch the row
              bin_ser_slots[b].back().push_back(slot_of[k - k0]);
            }
          }
    }
  }, nnz);
  H.ser_ptr.assign((size_t)B + 1, 0);
  H.ser_eptr.assign(1, 0);
  for (int b = 0; b < B; ++b) {
    for (size_t q = 0; q < bin_ser[b].size(); ++q) {
      H.ser_row.push_back(bin_ser[b][q]);
      H.ser_slot.insert(H.ser_slot.end(), bin_ser_slots[b][q].begin(), bin_ser_slots[b][q].end());
      H.ser_eptr.push_back((int32_t)H.ser_slot.size());
    }
    H.ser_ptr[b + 1] = (int32_t)H.ser_row.size();
  }
  // (one lane per serial row: fine for a few long or clustered rows among short ones, a serial chain for a matrix made of them)
  if ((int64_t)H.ser_slot.size() * 10 > nnz) { H.why = "more than a tenth of the nonzeros in rows with more than 7 entries inside one step of their bin"; return H; }
  // P workgroups: every panel's entries in Q parts (pieces are not split)
  const int Q = std::max(1, std::min(16, (4 * cus + S - 1) / S));
  for (int s_ = 0; s_ < S; ++s_) {
    const int64_t e0 = pstart[(size_t)s_ * B], e1 = pstart[(size_t)(s_ + 1) * B];
    const int64_t per = std::max<int64_t>(G, ((e1 - e0 + Q - 1) / Q + G - 1) / G * G);
    for (int64_t e = e0; e < e1; e += per) {
      H.wg_e0.push_back((int32_t)e);
      H.wg_panel.push_back(s_);
    }
  }
  H.wg_e0.push_back((int32_t)ptotal);
  H.ok = true;
  return H;
}

int upload_pb(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, const PbHost& h)
{
  if (!h.ok) return 0;
  int32_t *piece_dst = nullptr, *wg_e0 = nullptr, *wg_panel = nullptr, *bin_row0 = nullptr, *bin_e0 = nullptr, *bin_grp = nullptr, *grp_pos = nullptr;
  uint16_t *lidx = nullptr, *pos = nullptr, *rib = nullptr;
  uint8_t* step_lv = nullptr;
  uint32_t* sr = nullptr;
  double* prod = nullptr;
  TRY(upload_i32(c, &dst->perm, h.perm.get(), (size_t)h.np, 64));
  TRY(upload_i32(c, &piece_dst, h.piece_dst.get(), (size_t)(h.np >> h.gshift), 64));
  TRY(upload_i32(c, &wg_e0, h.wg_e0.data(), h.wg_e0.size()));
  TRY(upload_i32(c, &wg_panel, h.wg_panel.data(), h.wg_panel.size()));
  TRY(upload_i32(c, &bin_row0, h.bin_row0.data(), h.bin_row0.size()));
  TRY(upload_i32(c, &bin_e0, h.bin_e0.data(), h.bin_e0.size()));
  TRY(dev_alloc(c, &lidx, (size_t)h.np + 64));
  HIP_TRY(hipMemcpyAsync(lidx, h.lidx.get(), (size_t)h.np * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
  const size_t slack = h.wide ? (size_t)kPbwAhead * kPbwStep : 0;
  if (h.wide) {
    TRY(dev_alloc(c, &rib, (size_t)h.np + slack + 64));
    HIP_TRY(hipMemcpyAsync(rib, h.rib.get(), (size_t)h.np * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    TRY(dev_alloc(c, &step_lv, (size_t)(h.np >> 10) + 64));
    HIP_TRY(hipMemcpyAsync(step_lv, h.step_lv.data(), (size_t)(h.np >> 10), hipMemcpyHostToDevice, c->stream));
  } else {
    TRY(upload_i32(c, &bin_grp, h.bin_grp.data(), h.bin_grp.size()));
    TRY(upload_i32(c, &grp_pos, h.grp_pos.data(), h.grp_pos.size()));
    TRY(dev_alloc(c, &pos, (size_t)h.nnz + 128));
    HIP_TRY(hipMemcpyAsync(pos, h.pos.get(), ((size_t)h.nnz + 128) * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    TRY(dev_alloc(c, &sr, (size_t)h.rows + 64));
    HIP_TRY(hipMemcpyAsync(sr, h.sr.get(), (size_t)h.rows * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  TRY(dev_alloc(c, &dst->val, (size_t)h.np + 64));
  TRY(dev_alloc(c, &prod, (size_t)h.np + slack + 256));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host arrays die with the caller's PbHost
  dst->v = PbView{h.rows, h.cols, h.S, h.B, h.gshift, h.panel_shift, (int)h.wg_panel.size(), dst->val, lidx, piece_dst, wg_e0, wg_panel,
                  bin_row0, bin_e0, sr, bin_grp, grp_pos, pos, prod};
  dst->v.wide = h.wide ? 1 : 0, dst->v.rib = rib, dst->v.step_lv = step_lv;
  if (h.wide && !h.ser_row.empty()) {
    int32_t *ser_ptr = nullptr, *ser_row = nullptr, *ser_eptr = nullptr, *ser_slot = nullptr;
    TRY(upload_i32(c, &ser_ptr, h.ser_ptr.data(), h.ser_ptr.size()));
    TRY(upload_i32(c, &ser_row, h.ser_row.data(), h.ser_row.size()));
    TRY(upload_i32(c, &ser_eptr, h.ser_eptr.data(), h.ser_eptr.size()));
    TRY(upload_i32(c, &ser_slot, h.ser_slot.data(), h.ser_slot.size()));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dst->v.nser = (int)h.ser_row.size(), dst->v.ser_ptr = ser_ptr, dst->v.ser_row = ser_row, dst->v.ser_eptr = ser_eptr, dst->v.ser_slot = ser_slot;
  }
  dst->np = h.np, dst->p_threads = h.p_threads, dst->pad = (double)h.np / (double)h.nnz;
  dst->on = true;
  return 0;
}

// ---- a CPU walk through the host construction (tests without a GPU): phase P and phase R of the wide bins, slot by slot, exactly as
// the kernels order them; out = M x, info = {padded entries, bins, panels, highest level, serial rows}.  Returns 1 when the layout cannot hold the
// matrix (why: stderr), 2 on an inconsistency of the arrays.
extern "C" int pdlpdev_debug_pb_wide_host(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, const double* val, const double* x, double* out,
                                          int64_t* info)
{
  const PbHost H = build_pb_wide(rows, cols, off, idx, 256, true);
  if (!H.ok) {
    fprintf(stderr, "build_pb_wide: %s\n", H.why.c_str());
    return 1;
  }
  std::vector<double> prod((size_t)H.np + 64, 0.0);
  std::vector<char> written((size_t)H.np + 64, 0);
  const int gmask = (1 << H.gshift) - 1;
  for (size_t w = 0; w + 1 < H.wg_e0.size(); ++w)
    for (int64_t e = H.wg_e0[w]; e < H.wg_e0[w + 1]; ++e) {
      const int64_t col = ((int64_t)H.wg_panel[w] << H.panel_shift) + H.lidx[e];
      const double a    = H.perm[e] >= 0 ? val[H.perm[e]] : 0.0;
      const int64_t at  = ((int64_t)H.piece_dst[e >> H.gshift] << H.gshift) + (e & gmask);
      if (at >= H.np || written[at] || (H.perm[e] >= 0 && (col >= cols || idx[H.perm[e]] != col))) return 2;
      written[at] = 1;
      prod[at]    = a * x[col < cols ? col : 0];
    }
  int top_all = 0;
  std::vector<double> acc(kPbwRows);
  for (int b = 0; b < H.B; ++b) {
    std::fill(acc.begin(), acc.end(), 0.0);
    if (H.bin_e0[b] % kPbwStep) return 2;
    for (int64_t st = H.bin_e0[b] >> 10; st < (H.bin_e0[b + 1] >> 10); ++st) {
      const int top = H.step_lv[(size_t)st];
      top_all       = std::max(top_all, top);
      for (int f = 0; f <= kPbwMaxLevel; ++f)
        for (int i = 0; i < kPbwStep; ++i) {
          const uint16_t w = H.rib[(size_t)st * kPbwStep + i];
          if ((w >> 13) != f) continue;
          if (f > top || !written[(size_t)st * kPbwStep + i]) return 2;
          acc[w & (kPbwRows - 1)] = acc[w & (kPbwRows - 1)] + prod[(size_t)st * kPbwStep + i];
        }
    }
    if (!H.ser_row.empty())
      for (int q = H.ser_ptr[b]; q < H.ser_ptr[b + 1]; ++q) {
        double sum = 0.0;
        for (int e = H.ser_eptr[q]; e < H.ser_eptr[q + 1]; ++e) {
          if (!written[(size_t)H.ser_slot[e]] || H.rib[(size_t)H.ser_slot[e]] != 0xFFFFu) return 2;
          sum = sum + prod[(size_t)H.ser_slot[e]];
        }
        if (acc[H.ser_row[q] - H.bin_row0[b]] != 0.0) return 2;  // (a step touched a serial row)
        acc[H.ser_row[q] - H.bin_row0[b]] = sum;
      }
    for (int i = 0; i < H.bin_row0[b + 1] - H.bin_row0[b]; ++i) out[H.bin_row0[b] + i] = acc[i];
  }
  if (info) info[0] = H.np, info[1] = H.B, info[2] = H.S, info[3] = top_all, info[4] = (int64_t)H.ser_row.size();
  return 0;
}
