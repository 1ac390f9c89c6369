// gfx950 kernels of the jag layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"
#include "spmv_jag.hpp"

// jagged-layout twins of (2) and (3) and of the plain / ping-pong SpMV: same epilogues, LDS column sets
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_a_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size,
                 ctl->pending_avg != 0, ycopy, push};
  jag_block<decltype(e), WAVES>(J, xbar, e, part);
  if (push) p2pdev::count_exchange(push);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_step(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  jag_block<decltype(e), WAVES>(J, cur ? y0 : y1 /* y' */, e, part);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_cur(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next)
{
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  jag_block<decltype(e), WAVES>(J, cur ? y1 : y0, e, nullptr);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_plain(JagView J, const double* __restrict__ vec, double* __restrict__ out)
{
  StoreEpilogue e{out};
  jag_block<decltype(e), WAVES>(J, vec, e, nullptr);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_primal(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
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
  jag_block<decltype(e), WAVES>(J, xv, e, part);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part)
{
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalDualEpilogue e{core};
  jag_block<decltype(e), WAVES>(J, yv, e, part);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_jag_a_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
template __global__ void k_jag_a_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
template __global__ void k_jag_at_step<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
template __global__ void k_jag_at_step<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
template __global__ void k_jag_at_cur<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
template __global__ void k_jag_at_cur<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
template __global__ void k_jag_plain<8>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_jag_plain<16>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_jag_eval_primal<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_jag_eval_primal<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_jag_eval_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
template __global__ void k_jag_eval_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);

// ================================================================================================
// host side of the layout
// ================================================================================================
// open-addressing set of column indices with O(1) clear (a stamp per slot)
struct ColumnSet {
  std::vector<int32_t> key, payload;
  std::vector<uint32_t> stamp;
  uint32_t now = 0, mask;
  explicit ColumnSet(int capacity_log2)
      : key((size_t)1 << capacity_log2), payload((size_t)1 << capacity_log2), stamp((size_t)1 << capacity_log2, 0), mask((1u << capacity_log2) - 1) {}
  int32_t& at(int32_t c)  // payload of a column that is in the set
  {
    uint32_t h = ((uint32_t)c * 2654435761u) & mask;
    while (key[h] != c || stamp[h] != now) h = (h + 1) & mask;
    return payload[h];
  }
  void clear() { ++now; }
  bool contains(int32_t c) const
  {
    uint32_t h = ((uint32_t)c * 2654435761u) & mask;
    while (stamp[h] == now) {
      if (key[h] == c) return true;
      h = (h + 1) & mask;
    }
    return false;
  }
  bool insert(int32_t c)  // true when c was not there
  {
    uint32_t h = ((uint32_t)c * 2654435761u) & mask;
    while (stamp[h] == now) {
      if (key[h] == c) return false;
      h = (h + 1) & mask;
    }
    stamp[h] = now, key[h] = c;
    return true;
  }
};

// one set per host thread and capacity (the tasks of a parallel_tasks call share their thread's)
static ColumnSet& thread_column_set(int capacity_log2, int which)
{
  thread_local std::unique_ptr<ColumnSet> sets[2][2];
  std::unique_ptr<ColumnSet>& p = sets[capacity_log2 == 16][which];
  if (!p) p.reset(new ColumnSet(capacity_log2));
  return *p;
}


// one wave per group of up to G rows, `waves` groups per workgroup: keep a few hundred workgroups on the chip
// (below ~1.3e5 rows the layout has fewer workgroups than the chip has CUs and the CSR stream kernel's many small workgroups
// win by 8 % on banded 7e4- and 1e5-row LPs; from 2e5 rows on the jagged layout wins: 28.9 k vs 26.9 k it/s)
bool jag_geometry(int32_t rows, int mode, int* G_out, int* waves_out, int* wcap_out, int* brows_out)
{
  int G = rows >= 786432 ? 256 : rows >= 196608 ? 128 : rows >= 131072 ? 64 : 0;
  if (mode == 1 && G == 0) G = 64;
  if (G == 0) return false;
  // 8 waves / 8192 columns by default.  The wide geometry (CUOPT_AMD_TUNE=jag_waves=16: 16384 columns, one workgroup per CU) measured
  // 1-3 % faster on the banded, block-angular and multi-band LPs with the column-set version of this layout -- inside the noise
  // of two runs, so the default stays with the geometry every profile of this round was taken with.
  int waves = 8;
  if (cuopt_amd::tune_int("jag_waves", 8) == 16) waves = 16;
  *G_out = G, *waves_out = waves, *wcap_out = jag_window(waves), *brows_out = waves * G;
  return true;
}

// A workgroup's rows: consecutive, at most `row_cap`, and as many as keep their DISTINCT columns within the LDS window
// (rows longer than kLongRow do not count: they gather from global memory in workgroups of their own).  Greedy from
// `first`; returns the end of the block.
static int32_t jag_block_end(ColumnSet& set, const int32_t* off, const int32_t* idx, int wcap, int32_t first, int32_t limit, int32_t row_cap)
{
  set.clear();
  int32_t distinct = 0, r = first;
  const int32_t last = (int32_t)std::min<int64_t>((int64_t)first + row_cap, limit);
  for (; r < last; ++r) {
    const int32_t len = off[r + 1] - off[r];
    if (len > kLongRow) continue;
    if (distinct + len > wcap) {  // may overflow: count the new columns before inserting any
      int32_t fresh = 0;
      for (int32_t k = off[r]; k < off[r + 1]; ++k) fresh += !set.contains(idx[k]);
      if (distinct + fresh > wcap) break;
    }
    for (int32_t k = off[r]; k < off[r + 1]; ++k) distinct += set.insert(idx[k]);
  }
  return std::max(r, first + 1);  // a row of <= kLongRow nonzeros always fits an empty set
}

// cost of a block in gather equivalents: what filling its LDS set costs (a contiguous range is a coalesced copy, a list
// costs one request per run of consecutive columns) against the gathers it serves.  Also decides range vs list.
struct BlockSet {
  int32_t wbase = 0, wlen = 0;  // contiguous range, or ...
  std::vector<int32_t> cols;    // ... sorted distinct columns
  int64_t refs = 0, cost = 0;
};
static BlockSet jag_block_set(ColumnSet& set, const int32_t* off, const int32_t* idx, int wcap, int32_t r0, int32_t r1)
{
  BlockSet B;
  int32_t lo = std::numeric_limits<int32_t>::max(), hi = -1;
  for (int32_t r = r0; r < r1; ++r) {
    if (off[r + 1] - off[r] > kLongRow) continue;
    for (int32_t k = off[r]; k < off[r + 1]; ++k) lo = std::min(lo, idx[k]), hi = std::max(hi, idx[k]);
    B.refs += off[r + 1] - off[r];
  }
  if (hi < 0) return B;
  if ((int64_t)hi - lo + 1 <= wcap) {
    B.wbase = lo, B.wlen = hi - lo + 1;
    B.cost  = 1 + B.wlen / 16;
    return B;
  }
  set.clear();
  for (int32_t r = r0; r < r1; ++r)
    if (off[r + 1] - off[r] <= kLongRow)
      for (int32_t k = off[r]; k < off[r + 1]; ++k)
        if (set.insert(idx[k])) B.cols.push_back(idx[k]);
  std::sort(B.cols.begin(), B.cols.end());  // only the distinct columns (<= the LDS window) are sorted
  int64_t runs = 0;
  for (size_t i = 0; i < B.cols.size(); ++i) runs += i == 0 || B.cols[i] != B.cols[i - 1] + 1;
  B.cost = runs + (int64_t)B.cols.size() / 16;
  return B;
}

// The sampled estimate of build_jag on a MINI CSR: `nsamples` blocks of `brows` consecutive rows each (what the device's analysis
// pass extracts from a matrix under a candidate permutation, kernels_setup.hip).  Returns the saving 1 - cost / refs.
double jag_estimate_on_samples(int nsamples, int32_t brows, int wcap, const int32_t* soff, const int32_t* sidx)
{
  std::vector<int64_t> refs(nsamples, 0), cost(nsamples, 0);
  cuopt_amd::parallel_tasks(nsamples, [&](int t) {
    ColumnSet& set = thread_column_set(wcap > 8192 ? 16 : 15, 0);
    const int32_t first = t * brows;
    const int32_t last  = jag_block_end(set, soff, sidx, wcap, first, first + brows, brows);
    const BlockSet B    = jag_block_set(set, soff, sidx, wcap, first, last);
    refs[t] = B.refs, cost[t] = B.cost;
  }, (int64_t)soff[(size_t)nsamples * brows]);
  int64_t r = 0, c = 0;
  for (int t = 0; t < nsamples; ++t) r += refs[t], c += cost[t];
  return r ? 1.0 - (double)c / (double)r : 0.0;
}

// `mode`: 0 = use the layout when filling the LDS column sets costs at most half of the gathers they serve, 1 = always
JagHost build_jag(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int mode, int cus)
{
  JagHost H;
  const int64_t nnz = rows > 0 ? off[rows] : 0;
  if (rows <= 0 || cols <= 0 || nnz <= 0) return H;
  int G = 0, waves = 8, wcap = 0, brows = 0;
  if (!jag_geometry(rows, mode, &G, &waves, &wcap, &brows)) return H;
  const int slots = cus * (waves == 16 ? 1 : 2);  // workgroups resident at once: 80 KiB of LDS each (160 KiB with 16 waves)
  auto block_end = [&](ColumnSet& set, int32_t first, int32_t limit, int32_t row_cap) {
    return jag_block_end(set, off, idx, wcap, first, limit, row_cap);
  };
  auto block_set = [&](int32_t r0, int32_t r1, ColumnSet& set) { return jag_block_set(set, off, idx, wcap, r0, r1); };
  if (mode == 0) {  // estimate on ~48 blocks first: a random matrix is turned away after a few milliseconds
    const int samples = (int)std::min<int64_t>(48, std::max<int64_t>(1, rows / brows));
    std::vector<int64_t> refs(samples, 0), cost(samples, 0);
    cuopt_amd::parallel_tasks(samples, [&](int t) {
      ColumnSet& set = thread_column_set(waves == 16 ? 16 : 15, 0);
      const int32_t first = (int32_t)((int64_t)rows * t / samples);
      const int32_t last  = block_end(set, first, rows, brows);
      const BlockSet B    = block_set(first, last, set);
      refs[t] = B.refs, cost[t] = B.cost;
    }, nnz);
    int64_t r = 0, c = 0;
    for (int t = 0; t < samples; ++t) r += refs[t], c += cost[t];
    H.saving = r ? 1.0 - (double)c / (double)r : 0.0;
    if (H.saving < 0.35) return H;
  }
  // the partition: chunks of rows are cut independently (a chunk boundary is a block boundary), in parallel
  const int32_t chunk_rows = 4 * brows;
  const int nchunks        = (int)(((int64_t)rows + chunk_rows - 1) / chunk_rows);
  auto partition = [&](int32_t row_cap) {
    std::vector<std::vector<int32_t>> cuts(nchunks);
    cuopt_amd::parallel_tasks(nchunks, [&](int t) {
      ColumnSet& set = thread_column_set(waves == 16 ? 16 : 15, 0);
      const int32_t c0 = (int32_t)((int64_t)t * chunk_rows), c1 = (int32_t)std::min<int64_t>((int64_t)c0 + chunk_rows, rows);
      for (int32_t r = c0; r < c1;) cuts[t].push_back(r = block_end(set, r, c1, row_cap));
    }, nnz);
    std::vector<int32_t> row0(1, 0);
    for (auto& v : cuts) row0.insert(row0.end(), v.begin(), v.end());
    return row0;
  };
  H.row0 = partition(brows);
  // All workgroups of a round run at once (`slots` of them fit the chip) and the kernel lasts rounds x the largest block.  When
  // the column sets cut the blocks short (block-angular LP: 672 blocks on 512 slots, the second round a third full), smaller
  // blocks that fill the same number of rounds are strictly better: 2 x t(980 rows) instead of 2 x t(1490 rows).
  if (slots > 0 && (int)H.row0.size() - 1 > slots) {
    const int nb = (int)H.row0.size() - 1, rounds = (nb + slots - 1) / slots;
    if ((double)nb < 0.9 * (double)rounds * (double)slots) {
      int32_t cap = (int32_t)std::ceil((double)rows / (0.97 * (double)rounds * (double)slots));
      cap         = std::max<int32_t>(64, (cap + 63) & ~63);  // whole passes of 64 rows
      if (cap < brows) {
        std::vector<int32_t> alt = partition(cap);
        if (((int)alt.size() - 1 + slots - 1) / slots <= rounds) H.row0.swap(alt);
      }
    }
  }
  const int nblk    = (int)H.row0.size() - 1;
  const int ngroups = nblk * waves;  // every wave of every workgroup has a (possibly empty) share of the sorted passes
  H.rows = rows, H.waves = waves, H.ngroups = ngroups, H.nblk = nblk;
  // Per workgroup: rows with 1..kLongRow nonzeros sorted by length (descending, stable), cut into passes of 64, the
  // passes dealt to the waves in snake order (0..7, 7..0, ...): every wave gets the same share of long and short
  // passes, and a pass holds rows of nearly equal length.  pass 1 sizes everything (and builds the column sets), pass 2 fills.
  H.tile_e.assign((size_t)ngroups + 1, 0), H.tile_sr.assign((size_t)ngroups + 1, 0), H.lr_ptr.assign((size_t)nblk + 1, 0);
  H.set_ptr.assign((size_t)nblk + 1, 0), H.win.assign((size_t)2 * nblk, 0);
  auto wave_of_pass = [waves](int p) { return ((p / waves) & 1) ? waves - 1 - (p % waves) : p % waves; };
  // sorted order of a workgroup's short rows (local row numbers), number of them returned
  auto sort_block = [&](int b, std::vector<int32_t>& order) -> int32_t {
    const int32_t r0 = H.row0[b], r1 = H.row0[b + 1];
    int32_t bucket[kLongRow + 2] = {0};
    for (int32_t r = r0; r < r1; ++r) {
      const int32_t len = off[r + 1] - off[r];
      if (len >= 1 && len <= kLongRow) bucket[kLongRow - len]++;  // longest first
    }
    int32_t start[kLongRow + 2];
    int32_t run = 0;
    for (int i = 0; i <= kLongRow; ++i) start[i] = run, run += bucket[i];
    order.resize((size_t)run);
    for (int32_t r = r0; r < r1; ++r) {
      const int32_t len = off[r + 1] - off[r];
      if (len >= 1 && len <= kLongRow) order[start[kLongRow - len]++] = r - r0;
    }
    return run;
  };
  std::vector<int32_t> gsr((size_t)nblk * waves, 0);
  std::vector<int64_t> gent((size_t)nblk * waves, 0);
  std::vector<int32_t> nlong(nblk, 0);
  std::vector<BlockSet> sets(nblk);
  cuopt_amd::parallel_tasks(nblk, [&](int b) {
    std::vector<int32_t> order;
    ColumnSet& set = thread_column_set(waves == 16 ? 16 : 15, 0);
    const int32_t ns = sort_block(b, order);
    const int32_t r0 = H.row0[b], r1 = H.row0[b + 1];
    for (int32_t r = r0; r < r1; ++r) nlong[b] += off[r + 1] - off[r] > kLongRow;
    for (int32_t i0 = 0, p = 0; i0 < ns; i0 += 64, ++p) {
      const int w = wave_of_pass(p);
      for (int32_t i = i0; i < std::min(ns, i0 + 64); ++i) {
        gsr[(size_t)b * waves + w] += 1;
        gent[(size_t)b * waves + w] += off[r0 + order[i] + 1] - off[r0 + order[i]];
      }
    }
    sets[b] = block_set(r0, r1, set);
  }, nnz);
  int64_t refs = 0, cost = 0;
  for (int b = 0; b < nblk; ++b) {
    refs += sets[b].refs, cost += sets[b].cost;
    H.win[2 * b] = sets[b].wbase, H.win[2 * b + 1] = sets[b].wlen;
    H.set_ptr[b + 1] = H.set_ptr[b] + (int32_t)sets[b].cols.size();
    H.lr_ptr[b + 1]  = H.lr_ptr[b] + nlong[b];
  }
  H.saving = refs ? 1.0 - (double)cost / (double)refs : 0.0;
  if (mode == 0 && H.saving < 0.5) return H;
  for (int g = 0; g < ngroups; ++g) {
    H.tile_sr[g + 1] = H.tile_sr[g] + gsr[g];
    H.tile_e[g + 1]  = (int32_t)(H.tile_e[g] + gent[g]);
  }
  H.nsr = (size_t)H.tile_sr[ngroups], H.nent = (size_t)H.tile_e[ngroups];
  H.sr.reset(H.nsr + 1), H.slot.reset(H.nent + 1), H.perm.reset(H.nent + 1);
  H.lr_row.assign((size_t)H.lr_ptr[nblk], 0);
  H.set_col.assign((size_t)H.set_ptr[nblk], 0);
  cuopt_amd::parallel_tasks(nblk, [&](int b) {
    std::vector<int32_t> order;
    const int32_t ns = sort_block(b, order);
    const int32_t r0 = H.row0[b], r1 = H.row0[b + 1];
    const BlockSet& B = sets[b];
    std::copy(B.cols.begin(), B.cols.end(), H.set_col.begin() + H.set_ptr[b]);
    ColumnSet& map = thread_column_set(waves == 16 ? 16 : 15, 1);  // column -> slot of a list-mode set
    if (!B.wlen) {
      map.clear();
      for (size_t i = 0; i < B.cols.size(); ++i) {
        map.insert(B.cols[i]);
        map.at(B.cols[i]) = (int32_t)i;
      }
    }
    auto slot_of = [&](int32_t c) -> uint16_t { return B.wlen ? (uint16_t)(c - B.wbase) : (uint16_t)map.at(c); };
    int32_t nl = H.lr_ptr[b];
    for (int32_t r = r0; r < r1; ++r)
      if (off[r + 1] - off[r] > kLongRow) H.lr_row[nl++] = r;
    int32_t srpos[16];
    int64_t epos[16];
    for (int w = 0; w < waves; ++w) {
      const int g = b * waves + w;
      srpos[w] = H.tile_sr[g], epos[w] = H.tile_e[g];
    }
    for (int32_t i0 = 0, p = 0; i0 < ns; i0 += 64, ++p) {
      const int w = wave_of_pass(p);
      const int32_t i1 = std::min(ns, i0 + 64);
      for (int32_t i = i0; i < i1; ++i) {
        const int32_t len = off[r0 + order[i] + 1] - off[r0 + order[i]];
        H.sr[srpos[w]++] = ((uint32_t)(len - 1) << 16) | (uint32_t)order[i];
      }
      const int32_t kmax = off[r0 + order[i0] + 1] - off[r0 + order[i0]];
      int64_t e = epos[w];
      for (int32_t k = 0; k < kmax; ++k)
        for (int32_t i = i0; i < i1; ++i) {
          const int32_t r = r0 + order[i];
          if (off[r + 1] - off[r] <= k) break;  // sorted: the rest of the pass is shorter still
          H.slot[e] = slot_of(idx[off[r] + k]), H.perm[e] = off[r] + k, ++e;
        }
      epos[w] = e;
    }
  }, nnz);
  H.ok = true;
  return H;
}

int upload_jag(pdlpdev_ctx* c, pdlpdev_ctx::Jag* dst, const JagHost& h, const int32_t* d_off, const int32_t* d_idx,
                      const double* d_val)
{
  dst->saving = h.saving;
  if (!h.ok) return 0;
  int32_t *row0 = nullptr, *tile_e = nullptr, *tile_sr = nullptr, *win = nullptr, *set_ptr = nullptr, *set_col = nullptr,
          *lr_ptr = nullptr, *lr_row = nullptr;
  uint32_t* sr   = nullptr;
  uint16_t* slot = nullptr;
  TRY(upload_i32(c, &row0, h.row0.data(), h.row0.size()));
  TRY(upload_i32(c, &tile_e, h.tile_e.data(), h.tile_e.size()));
  TRY(upload_i32(c, &tile_sr, h.tile_sr.data(), h.tile_sr.size()));
  TRY(upload_i32(c, &win, h.win.data(), h.win.size()));
  TRY(upload_i32(c, &set_ptr, h.set_ptr.data(), h.set_ptr.size()));
  TRY(upload_i32(c, &set_col, h.set_col.data(), h.set_col.size(), 8));
  TRY(upload_i32(c, &lr_ptr, h.lr_ptr.data(), h.lr_ptr.size()));
  TRY(upload_i32(c, &lr_row, h.lr_row.data(), h.lr_row.size()));
  TRY(upload_i32(c, &dst->perm, h.perm.get(), h.nent, 8));
  TRY(dev_alloc(c, &sr, h.nsr + 8));
  HIP_TRY(hipMemcpyAsync(sr, h.sr.get(), h.nsr * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  TRY(dev_alloc(c, &slot, h.nent + 64));
  HIP_TRY(hipMemcpyAsync(slot, h.slot.get(), h.nent * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
  TRY(dev_alloc(c, &dst->val, h.nent + 8));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the host arrays die with the caller's JagHost
  dst->v    = JagView{h.rows, h.waves, h.ngroups, h.nblk, (int)h.lr_row.size(), row0, tile_e, tile_sr, sr, slot, dst->val,
                      win, set_ptr, set_col, lr_ptr, lr_row, d_off, d_idx, d_val};
  dst->nent = (int64_t)h.nent;
  dst->on   = true;
  return 0;
}
