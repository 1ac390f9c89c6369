// The SpMV layouts constructed ON THE DEVICE from the resident CSR (round 5): slab-major panels, the gather-free layout, jagged rows with
// LDS column sets -- each bit-identical to its host construction (kernels_panel.hip build_panels, kernels_pb.hip build_pb,
// kernels_jag.hip build_jag), which stay the tests' references.  gfx950, wave64.
#include <hip/hip_runtime.h>

#include "pdlp_setup.hpp"
#include "setup_primitives.hpp"

namespace {

// ================================================================================================
// slab-major panels on the device
// ================================================================================================
__global__ void __launch_bounds__(kT) k_panel_count(const int32_t* __restrict__ row0, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ idx, int S, int32_t slab_w, int64_t own_from,
                                                    int32_t* __restrict__ count /* [W][S] */)
{
  __shared__ int c[16];
  if (threadIdx.x < 16) c[threadIdx.x] = 0;
  __syncthreads();
  const int w = blockIdx.x;
  const int32_t a = row0[w], b = row0[w + 1];
  for (int32_t i = a + threadIdx.x; i < b; i += kT) {  // thread <-> row, as in the placement
    const int32_t k0 = off[i], k1 = off[i + 1];
    if ((int64_t)(k1 - k0) > own_from) continue;
    int s_cur = -1, run = 0;
    for (int32_t k = k0; k < k1; ++k) {
      const int s2 = idx[k] / slab_w;
      if (s2 != s_cur) {
        if (run) atomicAdd(&c[s_cur], run);
        s_cur = s2, run = 0;
      }
      ++run;
    }
    if (run) atomicAdd(&c[s_cur], run);
  }
  __syncthreads();
  if ((int)threadIdx.x < S) count[(size_t)w * S + threadIdx.x] = c[threadIdx.x];
}

// placement in (slab, row, CSR) order.  LDS: per (slab, row) the number of entries, then its exclusive prefix over the rows of the slab
// (= the uint16 row pointers the row kernel reads); thread <-> row.
template <bool SEG>
__global__ void __launch_bounds__(512) k_panel_place(const int32_t* __restrict__ row0, const int32_t* __restrict__ off,
                                                     const int32_t* __restrict__ idx, int S, int32_t slab_w, int64_t own_from,
                                                     const int32_t* __restrict__ tile_ptr, const int64_t* __restrict__ rp_base,
                                                     uint16_t* __restrict__ rowptr, int32_t* __restrict__ perm, int32_t* __restrict__ col)
{
  extern __shared__ unsigned short cnt[];  // [S][nr + 1]
  __shared__ int scratch[9];
  const int w = blockIdx.x;
  const int32_t a = row0[w], nr = row0[w + 1] - a;
  const int stride = nr + 1;
  for (int i = threadIdx.x; i < S * stride; i += 512) cnt[i] = 0;
  __syncthreads();
  for (int32_t r = threadIdx.x; r < nr; r += 512) {
    const int32_t k0 = off[a + r], k1 = off[a + r + 1];
    if ((int64_t)(k1 - k0) > own_from) continue;
    for (int32_t k = k0; k < k1; ++k) cnt[(idx[k] / slab_w) * stride + r] += 1;  // (slab, row) is this thread's alone
  }
  __syncthreads();
  // exclusive prefix over the rows, slab by slab (nr + 1 entries: the last one is the tile's size)
  for (int s2 = 0; s2 < S; ++s2) {
    int carry = 0;
    for (int r0 = 0; r0 < stride; r0 += 512) {
      const int r = r0 + threadIdx.x;
      const int v = r < nr ? (int)cnt[s2 * stride + r] : 0;
      int total   = 0;
      const int pre = block_exclusive_scan<512>(v, scratch, &total);
      if (r < stride) cnt[s2 * stride + r] = (unsigned short)(carry + pre);
      carry += total;
    }
  }
  __syncthreads();
  if (!SEG)
    for (int s2 = 0; s2 < S; ++s2) {
      uint16_t* dst = rowptr + rp_base[(size_t)w * S + s2];
      for (int r = threadIdx.x; r < stride; r += 512) dst[r] = cnt[s2 * stride + r];
    }
  __syncthreads();  // (the row pointers are out: the prefixes turn into cursors)
  for (int32_t r = threadIdx.x; r < nr; r += 512) {
    const int32_t k0 = off[a + r], k1 = off[a + r + 1];
    if ((int64_t)(k1 - k0) > own_from) continue;
    for (int32_t k = k0; k < k1; ++k) {  // CSR order inside (slab, row), whatever the column order of the row is
      const int32_t c = idx[k];
      const int s2    = c / slab_w;
      const int32_t q = tile_ptr[(size_t)w * S + s2] + (int32_t)(cnt[s2 * stride + r]++);
      perm[q] = k;
      col[q]  = SEG ? (int32_t)(((uint32_t)r << kSegColBits) | (uint32_t)(c - s2 * slab_w)) : c;
    }
  }
}


}  // namespace

// ================================================================================================
// slab-major panels from a device-resident CSR (returns 1: not built here, use build_panels on the host)
// ================================================================================================
int build_panels_device(pdlpdev_ctx* c, pdlpdev_ctx::Panels* dst, int32_t rows, int32_t cols, const int32_t* h_off,
                        const int32_t* d_off, const int32_t* d_idx, const double* d_val, int64_t slab_bytes, bool force)
{
  PanelHost h;
  std::vector<char> is_own;
  int64_t own_nnz = 0, own_from = 0;
  const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto plap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cuopt_amd setup]     panels: %-18s %6.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  if (!panel_plan(&h, rows, cols, h_off, slab_bytes, force, nullptr, &is_own, &own_nnz, &own_from)) return 0;
  plap("plan");
  const int W = h.W, S = h.S;
  const int64_t nnz = h_off[rows];
  h.nnz = (size_t)(nnz - own_nnz), h.rowptr_size = h.seg ? 0 : (size_t)S * ((size_t)rows + W);
  int32_t *row0 = nullptr, *tile_ptr = nullptr, *col = nullptr, *count = nullptr;
  uint16_t* rowptr = nullptr;
  int64_t* rp_base = nullptr;
  TRY(upload_i32(c, &row0, h.row0.data(), h.row0.size()));
  TRY(dev_alloc(c, &count, (size_t)W * S + 1));
  k_panel_count<<<W, kT, 0, c->stream>>>(row0, d_off, d_idx, S, h.slab_w, own_from, count);
  std::vector<int32_t> hc((size_t)W * S);
  HIP_TRY(hipMemcpyAsync(hc.data(), count, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  plap("count");
  h.tile_ptr.resize((size_t)W * S + 1);
  int64_t pos = 0;
  for (size_t i = 0; i < (size_t)W * S; ++i) {
    if (!h.seg && hc[i] >= 65536) return 0;  // 16-bit row pointers would overflow: keep the CSR stream layout
    if (hc[i] >= 65536) return 1;            // (long-tail variant with a tile beyond the kernel's 16-bit cursors: the host constructs it)
    h.tile_ptr[i] = (int32_t)pos;
    pos += hc[i];
  }
  h.tile_ptr[(size_t)W * S] = (int32_t)pos;
  h.rp_base.assign((size_t)W * S, 0);
  if (!h.seg)
    for (int w = 0; w < W; ++w) {
      const int64_t a = h.row0[w], nr = h.row0[w + 1] - a, rp = (int64_t)S * (a + w);
      for (int s2 = 0; s2 < S; ++s2) h.rp_base[(size_t)w * S + s2] = rp + (int64_t)s2 * (nr + 1);
    }
  TRY(upload_i32(c, &tile_ptr, h.tile_ptr.data(), h.tile_ptr.size()));
  {
    // columns, permutation, row pointers and values out of ONE allocation (every entry is written by the kernels below / k_permute:
    // no memset; four hipMalloc + memset pairs of 12-80 MB were ~1.5 ms per side)
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_col = pad((h.nnz + 8) * sizeof(int32_t)), b_rp = pad((h.rowptr_size + 8) * sizeof(uint16_t)), b_val = pad((h.nnz + 8) * sizeof(double));
    char* block = nullptr;
    HIP_TRY(hipMalloc((void**)&block, 2 * b_col + b_rp + b_val));
    c->allocs.push_back(block);
    c->bytes += (int64_t)(2 * b_col + b_rp + b_val);
    dst->val  = (double*)block;
    col       = (int32_t*)(block + b_val);
    dst->perm = (int32_t*)(block + b_val + b_col);
    rowptr    = (uint16_t*)(block + b_val + 2 * b_col);
  }
  TRY(dev_alloc(c, &rp_base, h.rp_base.size()));
  HIP_TRY(hipMemcpyAsync(rp_base, h.rp_base.data(), h.rp_base.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  plap("alloc");
  {
    const size_t lds = (size_t)S * (kPanelMaxRows + 1) * sizeof(unsigned short);
    static std::mutex mu;
    static std::vector<int> done;
    {
      std::lock_guard<std::mutex> lock(mu);
      if (std::find(done.begin(), done.end(), c->device) == done.end()) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_panel_place<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (kPanelMaxRows + 1) * 2));
        HIP_TRY(hipFuncSetAttribute((const void*)k_panel_place<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * (kPanelMaxRows + 1) * 2));
        done.push_back(c->device);
      }
    }
    if (h.seg) k_panel_place<true><<<W, 512, lds, c->stream>>>(row0, d_off, d_idx, S, h.slab_w, own_from, tile_ptr, rp_base, rowptr, dst->perm, col);
    else k_panel_place<false><<<W, 512, lds, c->stream>>>(row0, d_off, d_idx, S, h.slab_w, own_from, tile_ptr, rp_base, rowptr, dst->perm, col);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));  // (the host vectors die here)
  plap("place");
  dst->v = PanelView{h.W, h.S, h.any_long ? 1 : 0, row0, tile_ptr, rowptr, rp_base, col, dst->val};
  dst->v.seg = h.seg ? 1 : 0, dst->v.slab_w = h.slab_w;
  dst->nent  = (int64_t)h.nnz;
  if (!h.own_row.empty()) {
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

// ================================================================================================
// gather-free layout from a device-resident CSR (returns 1: not built here -- nothing was allocated, the host constructs it; 0: built or
// "does not fit" with dst->on false and *why set).  Every array is bit-identical to build_pb's (kernels_pb.hip), which stays the tests'
// reference construction:
//   * the bins (consecutive rows, a nonzero target lowered until every bin's padded image fits) are cut on the HOST from the offsets --
//     a chain of binary searches, one per bin; whether a target fits is decided on the device (the padded size of every bin);
//   * a workgroup per bin SORTS the bin's entries by (source panel, position in the bin) -- unique 32-bit words, a bitonic sort in LDS --
//     which is the order phase P stores them in: an entry's rank inside its (bin, panel) chunk is its distance from the chunk's first
//     position, the chunk's place in the bin's image the sum of the padded chunks in front of it (one scan of the run ends);
//   * chunk sizes go to a B x S table (16-bit), its transposed exclusive scan gives the panel-major starts of phase P;
//   * a second kernel per bin sorts the rows by length (descending, ties by row: what std::stable_sort leaves) and lays the entries'
//     positions out along the jagged diagonals of every group of 64 rows -- a wave per group, one ballot per diagonal.
// ================================================================================================
namespace {
constexpr int kPbBinT  = 1024;
constexpr int kPbSortN = 16384;  // words a bin's sort may take (>= kPbCap)
static_assert(kPbCap <= kPbSortN && kPbMaxRows <= 1024, "a bin's entries / rows fit the LDS sorts");

__global__ void __launch_bounds__(kT) k_max_row_len(int32_t rows, const int32_t* __restrict__ off, int* __restrict__ out)
{
  int best = 0;
  for (int64_t r = (int64_t)blockIdx.x * kT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kT) best = max(best, off[r + 1] - off[r]);
  for (int d = 1; d < 64; d <<= 1) best = max(best, __shfl_xor(best, d, 64));
  if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(out, best);
}

// One bin.  MODE 0: its padded size only (the target search).  1: + the chunk sizes.  2: placement (phase P order, local columns,
// pieces, every entry's position in the bin's image).
constexpr int kPbItems = (kPbCap + kPbBinT - 1) / kPbBinT;  // positions per thread
template <int MODE>
__global__ void __launch_bounds__(kPbBinT)
k_pb_bin(const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int panel_shift, int gshift, int S, int B,
         int32_t* __restrict__ bin_size, int* __restrict__ maxpad, uint16_t* __restrict__ cnt, const int32_t* __restrict__ pstart,
         const int32_t* __restrict__ bin_e0, int32_t* __restrict__ perm, uint16_t* __restrict__ lidx, int32_t* __restrict__ piece_dst,
         uint16_t* __restrict__ epos)
{
  __shared__ uint32_t w[kPbSortN];
  __shared__ int scratch[kPbBinT / 64 + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = row0[b], r1 = row0[b + 1];
  const int k0 = off[r0], n = off[r1] - k0;
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += kPbBinT) w[i] = i < n ? ((uint32_t)(idx[k0 + i] >> panel_shift) << 14 | (uint32_t)i) : 0xFFFFFFFFu;
  __syncthreads();
  lds_bitonic_sort<kPbBinT>(w, n2);
  const int G = 1 << gshift;
  // thread t looks after the sorted positions [t * kPbItems, (t + 1) * kPbItems)
  const int p0 = tid * kPbItems;
  // (1) the first position of every position's chunk: a max-scan of the chunk starts
  int rs[kPbItems];
  int last = 0;
#pragma unroll
  for (int q = 0; q < kPbItems; ++q) {
    const int i = p0 + q;
    const bool start = i < n && (i == 0 || (w[i] >> 14) != (w[i - 1] >> 14));
    last  = start ? i : last;
    rs[q] = last;  // (so far: the latest start inside this thread's range, 0 if none)
  }
  const int before = block_exclusive_max<kPbBinT>(last, scratch);
#pragma unroll
  for (int q = 0; q < kPbItems; ++q) rs[q] = rs[q] > before ? rs[q] : before;
  // (2) the padding in front of every position: a chunk's padding counts from its last entry on
  int padv[kPbItems], sum = 0;
#pragma unroll
  for (int q = 0; q < kPbItems; ++q) {
    const int i = p0 + q;
    const bool end = i < n && (i == n - 1 || (w[i] >> 14) != (w[i + 1] >> 14));
    padv[q] = end ? (G - ((i - rs[q] + 1) & (G - 1))) & (G - 1) : 0;
    sum += padv[q];
  }
  int total = 0;
  int P = block_exclusive_scan<kPbBinT>(sum, scratch, &total);
  if (MODE <= 1) {
    if (tid == 0) {
      bin_size[b] = n + total;
      atomicMax(maxpad, n + total);
    }
    if (MODE == 0) return;
  }
#pragma unroll
  for (int q = 0; q < kPbItems; ++q) {
    const int i = p0 + q;
    if (i < n) {
      const int s_ = (int)(w[i] >> 14), e = (int)(w[i] & 0x3FFF), k = k0 + e;
      const bool end = i == n - 1 || (w[i] >> 14) != (w[i + 1] >> 14);
      if (MODE == 1) {
        if (end) cnt[(size_t)b * S + s_] = (uint16_t)(i - rs[q] + 1);
      } else {
        const int32_t ps = pstart[(size_t)s_ * B + b];
        const int32_t pp = ps + (i - rs[q]);
        perm[pp]         = k;
        lidx[pp]         = (uint16_t)(idx[k] & ((1 << panel_shift) - 1));
        epos[k]          = (uint16_t)(i + P);  // = the chunk's place in the image (its first position + the padding in front) + the rank
        if (end) {
          const int np_ = (i - rs[q] + 1 + G - 1) >> gshift;
          const int32_t q0 = ps >> gshift, d0 = (bin_e0[b] + rs[q] + P) >> gshift;
          for (int t = 0; t < np_; ++t) piece_dst[q0 + t] = d0 + t;
        }
      }
    }
    P += padv[q];
  }
}

// A bin's padded size (and, WRITE, its chunk sizes) from an LDS histogram over the panels: what the target search and the chunk table
// need, without the sort -- when the panel counters fit (S <= kPbHistMax)
constexpr int kPbHistMax = 12288;
template <bool WRITE>
__global__ void __launch_bounds__(512)
k_pb_bin_hist(const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int panel_shift, int gshift, int S,
              int32_t* __restrict__ bin_size, int* __restrict__ maxpad, uint16_t* __restrict__ cnt)
{
  extern __shared__ int hist[];
  __shared__ int scratch[9];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int k0 = off[row0[b]], k1 = off[row0[b + 1]];
  for (int i = tid; i < S; i += 512) hist[i] = 0;
  __syncthreads();
  for (int k = k0 + tid; k < k1; k += 512) atomicAdd(&hist[idx[k] >> panel_shift], 1);
  __syncthreads();
  const int G = 1 << gshift;
  int sum = 0;
  for (int i = tid; i < S; i += 512) {
    const int c_ = hist[i];
    sum += (c_ + G - 1) / G * G;
    if (WRITE) cnt[(size_t)b * S + i] = (uint16_t)c_;
  }
  int total = 0;
  (void)block_exclusive_scan<512>(sum, scratch, &total);
  if (tid == 0) {
    bin_size[b] = total;
    atomicMax(maxpad, total);
  }
}

// padded chunk sizes, panel-major (the order phase P stores the chunks in)
__global__ void __launch_bounds__(kT) k_pb_chunk_sizes(const uint16_t* __restrict__ cnt, int S, int B, int gshift, int32_t* __restrict__ out)
{
  const int G = 1 << gshift;
  const int64_t total = (int64_t)S * B;
  for (int64_t t = (int64_t)blockIdx.x * kT + threadIdx.x; t < total; t += (int64_t)gridDim.x * kT) {
    const int64_t s_ = t / B, b = t % B;
    out[t] = ((int)cnt[(size_t)b * S + s_] + G - 1) / G * G;
  }
}
__global__ void __launch_bounds__(kT) k_pb_panel_starts(const int32_t* __restrict__ pstart, int S, int B, int32_t* __restrict__ out)
{
  for (int s_ = blockIdx.x * kT + threadIdx.x; s_ <= S; s_ += gridDim.x * kT) out[s_] = pstart[(size_t)s_ * B];
}

// One bin's rows by length (descending, ties by row), the groups of 64 and the jagged diagonals of the entries' positions
__global__ void __launch_bounds__(1024)
k_pb_rows(const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const uint16_t* __restrict__ epos, const int32_t* __restrict__ bin_grp,
          uint32_t* __restrict__ sr, int32_t* __restrict__ grp_pos, uint16_t* __restrict__ pos)
{
  __shared__ uint32_t w[1024];
  __shared__ int scratch[17];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = row0[b], nr = row0[b + 1] - r0;
  int n2 = 1;
  while (n2 < nr) n2 <<= 1;
  if (tid < n2) w[tid] = tid < nr ? ((uint32_t)(0xFFFF - (off[r0 + tid + 1] - off[r0 + tid])) << 10 | (uint32_t)tid) : 0xFFFFFFFFu;
  __syncthreads();
  lds_bitonic_sort<1024>(w, n2);
  const bool live = tid < nr;
  const int row   = live ? (int)(w[tid] & 1023) : 0;
  const int len   = live ? 0xFFFF - (int)(w[tid] >> 10) : 0;
  if (live) sr[r0 + tid] = (uint32_t)len << 16 | (uint32_t)row;
  const int at = off[r0] + block_exclusive_scan<1024>(len, scratch, nullptr);  // where this row's group would start if it were a group's first
  const int lane = tid & 63, g = tid >> 6;
  if (lane == 0 && live) grp_pos[bin_grp[b] + g] = at;
  // the group's diagonals: diagonal k holds the k-th entry of every row that has one -- a prefix of the (sorted) group
  int D          = __shfl(at, 0, 64);
  const int kmax = __shfl(len, 0, 64);
  const int k_row = off[r0 + row];
  for (int k = 0; k < kmax; ++k) {
    const unsigned long long mask = __ballot(len > k);
    if (len > k) pos[D + lane] = epos[k_row + k];
    D += __popcll(mask);
  }
}
}  // namespace

// ================================================================================================
// gather-free layout with WIDE bins from a device-resident CSR (pdlp_kernels.hpp kPbw*, host reference: build_pb_wide in kernels_pb.hip;
// returns 1: not built here, 0: built, or "does not fit" with dst->on false and *why set).  Bins are kPbwRows consecutive rows, so nothing
// is searched for:
//   * k_pbw_count: a workgroup per bin, an LDS histogram over the panels -> the B x S chunk table (16-bit) and the bin's image size;
//   * the table's transposed exclusive scan gives the panel-major starts of phase P (the kernels of the other geometry);
//   * k_pbw_place: a workgroup per bin.  An entry's rank inside its (bin, panel) chunk is the number of earlier entries of the bin, in
//     (row, column) order, that fall into the same panel: the bin's entries are cut into 8 contiguous SEGMENTS, every segment is
//     counted per panel (LDS atomics: order does not matter for a count), the counts are scanned per panel over the segments, and then
//     ONE WAVE per segment walks it 64 entries at a time -- the lanes of equal panel find each other with 12 ballots, the earlier ones
//     among them are the rank inside the tile, the segment's running count per panel (LDS, private to the wave) the rest;
//   * k_pbw_levels: a workgroup per step of the image; the level of a slot = the number of earlier slots of the step with the same row:
//     rounds of ds_min over a tag per row (the earliest pending slot of every row wins the round's level).  Rows still pending after
//     the last level are SERIAL rows: flagged, collected, ordered on the host (a handful), their slots erased from the steps and
//     listed (k_pbw_serial) from the slots phase 'place' noted per entry.
// ================================================================================================
namespace {
constexpr int kPbwSeg = 8;
constexpr int kPbwMaxPanels = 4096;  // 36 bytes of LDS per panel in k_pbw_place

__global__ void __launch_bounds__(kT) k_pbw_rowin(int32_t rows, const int32_t* __restrict__ off, uint16_t* __restrict__ rowin)
{
  for (int64_t r = (int64_t)blockIdx.x * kT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kT) {
    const uint16_t v = (uint16_t)(r & (kPbwRows - 1));
    for (int k = off[r]; k < off[r + 1]; ++k) rowin[k] = v;
  }
}

__global__ void __launch_bounds__(512)
k_pbw_count(int32_t rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int panel_shift, int gshift, int S, int32_t* __restrict__ bin_size,
            int* __restrict__ flags /* [0] longest chunk, [1] largest bin */, uint16_t* __restrict__ cnt)
{
  extern __shared__ int hist[];
  __shared__ int scratch[9];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = b * kPbwRows, r1 = min(rows, r0 + kPbwRows);
  const int k0 = off[r0], k1 = off[r1];
  for (int i = tid; i < S; i += 512) hist[i] = 0;
  __syncthreads();
  for (int k = k0 + tid; k < k1; k += 512) atomicAdd(&hist[idx[k] >> panel_shift], 1);
  __syncthreads();
  const int gm = (1 << gshift) - 1;
  int sum = 0, longest = 0;
  for (int i = tid; i < S; i += 512) {
    const int c_ = hist[i];
    sum += (c_ + gm) & ~gm;
    longest = max(longest, c_);
    cnt[(size_t)b * S + i] = (uint16_t)c_;
  }
  int total = 0;
  (void)block_exclusive_scan<512>(sum, scratch, &total);
  for (int d = 1; d < 64; d <<= 1) longest = max(longest, __shfl_xor(longest, d, 64));
  if ((tid & 63) == 0 && longest > 0) atomicMax(flags, longest);
  if (tid == 0) {
    total = (total + kPbwStep - 1) / kPbwStep * kPbwStep;
    bin_size[b] = total;
    atomicMax(flags + 1, total);
  }
}

__global__ void __launch_bounds__(1024)
k_pbw_place(int32_t rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, const uint16_t* __restrict__ rowin, int panel_shift, int gshift, int S, int B,
            const int32_t* __restrict__ pstart, const int32_t* __restrict__ bin_e0, int32_t* __restrict__ perm, uint16_t* __restrict__ lidx,
            int32_t* __restrict__ piece_dst, uint16_t* __restrict__ rib, int32_t* __restrict__ eslot)
{
  extern __shared__ uint32_t sm[];
  __shared__ int scratch[17];
  uint32_t* tab = sm;                         // [kPbwSeg][S]: counts, then the segments' first ranks, then running ranks
  int* lst      = (int*)(sm + kPbwSeg * S);   // [S] the chunk's place in the bin's image
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r0 = b * kPbwRows, r1 = min(rows, r0 + kPbwRows);
  const int k0 = off[r0], n = off[r1] - k0;
  const int seglen = max(64, ((n + kPbwSeg * 64 - 1) / (kPbwSeg * 64)) * 64);
  for (int i = tid; i < kPbwSeg * S; i += 1024) tab[i] = 0u;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&tab[(i / seglen) * S + (idx[k0 + i] >> panel_shift)], 1u);
  __syncthreads();
  constexpr int KPT = kPbwMaxPanels / 1024;
  const uint32_t gm = (1u << gshift) - 1u;
  int padv[KPT], sum = 0;
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int key = tid * KPT + q;
    padv[q]       = 0;
    if (key < S) {
      uint32_t run = 0;
      for (int g = 0; g < kPbwSeg; ++g) {
        const uint32_t c_ = tab[g * S + key];
        tab[g * S + key]  = run;
        run += c_;
      }
      padv[q] = (int)((run + gm) & ~gm);
    }
    sum += padv[q];
  }
  int at        = block_exclusive_scan<1024>(sum, scratch, nullptr);
  const int e0b = bin_e0[b];
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int key = tid * KPT + q;
    if (key < S) {
      lst[key] = at;
      const int np_    = padv[q] >> gshift;
      const int32_t p0 = pstart[(size_t)key * B + b] >> gshift, d0 = (e0b + at) >> gshift;
      for (int i = 0; i < np_; ++i) piece_dst[p0 + i] = d0 + i;
    }
    at += padv[q];
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  if (wave >= kPbwSeg) return;
  uint32_t* mine   = tab + wave * S;
  const int segb   = wave * seglen, sege = min(n, segb + seglen);
  const int cmask  = (1 << panel_shift) - 1;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int i0 = segb; i0 < sege; i0 += 64) {
    const int i      = i0 + lane;
    const bool valid = i < sege;
    const int k      = k0 + (valid ? i : segb);
    const int col    = idx[k];
    const int key    = col >> panel_shift;
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 12; ++bit) {
      const bool one = (key >> bit) & 1;
      const unsigned long long bal = __ballot(one);
      mask &= one ? bal : ~bal;
    }
    if (valid) {
      const uint32_t base = mine[key];
      const int rank      = (int)base + __popcll(mask & lt);
      if ((mask & lt) == 0ull) mine[key] = base + (uint32_t)__popcll(mask);  // (the first lane of the group; behind every lane's read: one wave's LDS operations execute in order)
      const int32_t pp = pstart[(size_t)key * B + b] + rank;
      perm[pp] = k;
      lidx[pp] = (uint16_t)(col & cmask);
      rib[(size_t)e0b + lst[key] + rank] = rowin[k];
      eslot[k] = e0b + lst[key] + rank;  // (where the entry's product lands: the serial rows' lists are made of these)
    }
  }
}

__global__ void __launch_bounds__(1024)
k_pbw_levels(uint16_t* __restrict__ rib, uint8_t* __restrict__ step_lv, const int32_t* __restrict__ bin_e0, int B, uint8_t* __restrict__ serial, int* __restrict__ fail)
{
  __shared__ uint32_t tag[kPbwRows];
  const int t = threadIdx.x;
  for (int i = t; i < kPbwRows; i += 1024) tag[i] = 0xFFFFFFFFu;
  __syncthreads();
  uint16_t* w      = rib + (size_t)blockIdx.x * kPbwStep + t;
  const uint16_t v = *w;
  const int row    = v & (kPbwRows - 1);
  bool pending     = v != 0xFFFFu;
  int lvl = 0, round = 0;
  for (;;) {
    if (pending) atomicMin(&tag[row], (uint32_t)t);
    __syncthreads();
    const bool won = pending && tag[row] == (uint32_t)t;
    __syncthreads();
    if (won) tag[row] = 0xFFFFFFFFu, lvl = round, pending = false;
    if (!__syncthreads_or(pending)) break;
    if (++round > kPbwMaxLevel) {
      // rows still pending have more than kPbwMaxLevel + 1 entries in this step: serial rows (the step's bin: the last one that
      // starts at or before the step)
      if (pending) {
        const int32_t at = (int32_t)(blockIdx.x * (unsigned)kPbwStep);
        int lo = 0, hi = B;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (bin_e0[mid] <= at) lo = mid; else hi = mid;
        }
        serial[(size_t)lo * kPbwRows + row] = 1;
        *fail = 1;
      }
      break;
    }
  }
  if (v != 0xFFFFu) *w = (uint16_t)(row | lvl << 13);
  if (t == 0) step_lv[blockIdx.x] = (uint8_t)min(round, kPbwMaxLevel);
}

// the flagged rows, in any order (the host sorts the few there are)
__global__ void __launch_bounds__(kT) k_pbw_collect(int32_t rows, const uint8_t* __restrict__ serial, int32_t cap, int32_t* __restrict__ list, int* __restrict__ count)
{
  for (int64_t r = (int64_t)blockIdx.x * kT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * kT)
    if (serial[r]) {
      const int at = atomicAdd(count, 1);
      if (at < cap) list[at] = (int32_t)r;
    }
}
// a serial row's slots read as padding; its list keeps where its products land, in column order
__global__ void __launch_bounds__(kT)
k_pbw_serial(int32_t nser, const int32_t* __restrict__ ser_row, const int32_t* __restrict__ ser_eptr, const int32_t* __restrict__ off, const int32_t* __restrict__ eslot,
             uint16_t* __restrict__ rib, int32_t* __restrict__ ser_slot)
{
  const int q = blockIdx.x;
  if (q >= nser) return;
  const int r = ser_row[q], k0 = off[r], n = off[r + 1] - k0, e0 = ser_eptr[q];
  for (int i = threadIdx.x; i < n; i += kT) {
    const int32_t slot = eslot[k0 + i];
    ser_slot[e0 + i]   = slot;
    rib[slot]          = 0xFFFFu;
  }
}
}  // namespace

int build_pb_wide_device(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                         int cus, bool forced, std::string* why)
{
  const int64_t nnz = rows > 0 ? h_off[rows] : 0;
  if (rows <= 0 || cols <= 0 || nnz <= 0) { *why = "empty matrix"; return 0; }
  const int panel_shift = cols > (1 << 21) ? 14 : 13;
  const int S           = (int)(((int64_t)cols + (1 << panel_shift) - 1) >> panel_shift);
  if (S > kPbwMaxPanels) return 1;
  hipStream_t s = c->stream;
  const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto plap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cuopt_amd setup]     gather-free (wide bins): %-14s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<void*> tmp;  // (the context's scratch blocks this construction holds: handed back, not freed, on every way out)
  struct Release {
    pdlpdev_ctx* c;
    std::vector<void*>& v;
    ~Release() { for (void* p : v) scratch_release(c, p); }
  } release_tmp{c, tmp};
  auto talloc = [&](void** p, size_t bytes) -> int {
    TRY(scratch_take(c, p, bytes));
    tmp.push_back(*p);
    return 0;
  };
  int* d_scal = nullptr;  // [0] longest row, [1] longest chunk, [2] largest bin, [3] level overflow
  TRY(talloc((void**)&d_scal, 4 * sizeof(int)));
  HIP_TRY(hipMemsetAsync(d_scal, 0, 4 * sizeof(int), s));
  k_max_row_len<<<grid_of(rows), kT, 0, s>>>(rows, d_off, d_scal);
  const int B = (rows + kPbwRows - 1) / kPbwRows;
  uint16_t* d_cnt = nullptr;
  int32_t* d_bin_size = nullptr;
  TRY(talloc((void**)&d_cnt, (size_t)B * S * sizeof(uint16_t)));
  TRY(talloc((void**)&d_bin_size, (size_t)B * sizeof(int32_t)));
  const int gshift = pbw_piece_shift(nnz, S, B), G = 1 << gshift;
  k_pbw_count<<<B, 512, (size_t)S * sizeof(int), s>>>(rows, d_off, d_idx, panel_shift, gshift, S, d_bin_size, d_scal + 1, d_cnt);
  HIP_TRY(hipGetLastError());
  int h_scal[4] = {0, 0, 0, 0};
  std::vector<int32_t> bin_size(B), bin_e0((size_t)B + 1, 0), row0((size_t)B + 1);
  HIP_TRY(hipMemcpyAsync(h_scal, d_scal, sizeof(h_scal), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(bin_size.data(), d_bin_size, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_scal[0] > (forced ? kPbCap / 2 : 256)) { *why = "a row with " + std::to_string(h_scal[0]) + " nonzeros"; return 0; }
  if (h_scal[1] > 65535 || h_scal[2] > kPbwMaxSteps * kPbwStep) { *why = "a chunk of more than 65535 entries or a bin of more than 4096 steps"; return 0; }
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    row0[b]   = (int32_t)((int64_t)b * kPbwRows);
    bin_e0[b] = (int32_t)total;
    total += bin_size[b];
    if (total >= ((int64_t)1 << 31) - 65536) { *why = "more than 2^31 padded entries"; return 0; }
  }
  row0[B] = rows, bin_e0[B] = (int32_t)total;
  plap("chunk counts");
  const int64_t cells = (int64_t)S * B;
  int32_t *d_sizes = nullptr, *d_pstart = nullptr, *d_bs = nullptr;
  TRY(talloc((void**)&d_sizes, (size_t)cells * sizeof(int32_t)));
  TRY(talloc((void**)&d_pstart, ((size_t)cells + 1) * sizeof(int32_t)));
  TRY(talloc((void**)&d_bs, ((size_t)(cells + 1) / 4096 + 2) * sizeof(int32_t)));
  k_pb_chunk_sizes<<<grid_of(cells), kT, 0, s>>>(d_cnt, S, B, gshift, d_sizes);
  TRY(dev_exclusive_scan(s, d_sizes, d_pstart, cells, d_bs));
  uint16_t* d_rowin = nullptr;
  TRY(talloc((void**)&d_rowin, ((size_t)nnz + 64) * sizeof(uint16_t)));
  k_pbw_rowin<<<grid_of(rows), kT, 0, s>>>(rows, d_off, d_rowin);
  plap("chunk table");
  int32_t *piece_dst = nullptr, *wg_e0 = nullptr, *wg_panel = nullptr, *bin_row0 = nullptr, *d_bin_e0 = nullptr;
  uint16_t *lidx = nullptr, *rib = nullptr;
  uint8_t* step_lv = nullptr;
  double* prod = nullptr;
  const size_t slack = (size_t)kPbwAhead * kPbwStep;
  TRY(dev_alloc(c, &dst->perm, (size_t)total + 64));
  TRY(dev_alloc(c, &piece_dst, (size_t)(total >> gshift) + 64));
  TRY(dev_alloc(c, &lidx, (size_t)total + 64));
  TRY(dev_alloc(c, &rib, (size_t)total + slack + 64));
  TRY(dev_alloc(c, &step_lv, (size_t)(total >> 10) + 64));
  HIP_TRY(hipMemsetAsync(dst->perm, 0xFF, (size_t)total * sizeof(int32_t), s));  // padding slots: -1 (their lidx: 0, dev_alloc's zero fill)
  HIP_TRY(hipMemsetAsync(rib, 0xFF, (size_t)total * sizeof(uint16_t), s));        // padding slots: level 7
  TRY(upload_i32(c, &bin_row0, row0.data(), row0.size()));
  TRY(upload_i32(c, &d_bin_e0, bin_e0.data(), bin_e0.size()));
  plap("alloc");
  int32_t* d_eslot = nullptr;
  uint8_t* d_serial = nullptr;
  TRY(talloc((void**)&d_eslot, ((size_t)nnz + 64) * sizeof(int32_t)));
  TRY(talloc((void**)&d_serial, (size_t)B * kPbwRows));
  HIP_TRY(hipMemsetAsync(d_serial, 0, (size_t)B * kPbwRows, s));
  HIP_TRY(hipFuncSetAttribute((const void*)k_pbw_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kPbwSeg + 1) * kPbwMaxPanels * sizeof(uint32_t))));
  k_pbw_place<<<B, 1024, (size_t)(kPbwSeg + 1) * S * sizeof(uint32_t), s>>>(rows, d_off, d_idx, d_rowin, panel_shift, gshift, S, B, d_pstart, d_bin_e0, dst->perm, lidx, piece_dst, rib, d_eslot);
  HIP_TRY(hipGetLastError());
  plap("place");
  if (total > 0) k_pbw_levels<<<(unsigned)(total >> 10), 1024, 0, s>>>(rib, step_lv, d_bin_e0, B, d_serial, d_scal + 3);
  HIP_TRY(hipGetLastError());
  // P workgroups: every panel's entries in Q parts (pieces are not split)
  int32_t* d_pan = nullptr;
  TRY(talloc((void**)&d_pan, ((size_t)S + 1) * sizeof(int32_t)));
  k_pb_panel_starts<<<grid_of(S + 1), kT, 0, s>>>(d_pstart, S, B, d_pan);
  std::vector<int32_t> pan((size_t)S + 1);
  HIP_TRY(hipMemcpyAsync(pan.data(), d_pan, pan.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_scal + 3, d_scal + 3, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  plap("levels");
  int32_t *ser_ptr = nullptr, *ser_row = nullptr, *ser_eptr = nullptr, *ser_slot = nullptr;
  int nser = 0;
  if (h_scal[3]) {
    // serial rows (a row with more than 7 entries inside one step of its bin): collected on the device, ordered and priced on the host
    int32_t* d_list = nullptr;
    int* d_count = nullptr;
    const int32_t cap = (int32_t)std::min<int64_t>(rows, 1 << 22);
    TRY(talloc((void**)&d_list, (size_t)cap * sizeof(int32_t)));
    TRY(talloc((void**)&d_count, sizeof(int)));
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int), s));
    k_pbw_collect<<<grid_of(rows), kT, 0, s>>>(rows, d_serial, cap, d_list, d_count);
    int count = 0;
    HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const char* too_many = "more than a tenth of the nonzeros in rows with more than 7 entries inside one step of their bin";
    if (count > cap) { *why = too_many; return 0; }  // (the arrays stay with the context until it is destroyed)
    std::vector<int32_t> h_row((size_t)count), h_ptr((size_t)B + 1, 0), h_eptr((size_t)count + 1, 0);
    HIP_TRY(hipMemcpy(h_row.data(), d_list, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::sort(h_row.begin(), h_row.end());
    int64_t ser_nnz = 0;
    for (int q = 0; q < count; ++q) {
      ser_nnz += h_off[h_row[q] + 1] - h_off[h_row[q]];
      h_eptr[q + 1] = (int32_t)ser_nnz;
      h_ptr[h_row[q] / kPbwRows + 1]++;
    }
    for (int b = 0; b < B; ++b) h_ptr[b + 1] += h_ptr[b];
    if (ser_nnz * 10 > nnz) { *why = too_many; return 0; }
    nser = count;
    TRY(upload_i32(c, &ser_ptr, h_ptr.data(), h_ptr.size()));
    TRY(upload_i32(c, &ser_row, h_row.data(), h_row.size()));
    TRY(upload_i32(c, &ser_eptr, h_eptr.data(), h_eptr.size()));
    TRY(dev_alloc(c, &ser_slot, (size_t)ser_nnz + 1));
    k_pbw_serial<<<nser, kT, 0, s>>>(nser, ser_row, ser_eptr, d_off, d_eslot, rib, ser_slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));  // (the host vectors die here)
    plap("serial rows");
  }
  std::vector<int32_t> h_wg_e0, h_wg_panel;
  const int Q = std::max(1, std::min(16, (4 * cus + S - 1) / S));
  for (int s_ = 0; s_ < S; ++s_) {
    const int64_t e0 = pan[s_], e1 = pan[s_ + 1];
    const int64_t per = std::max<int64_t>(G, ((e1 - e0 + Q - 1) / Q + G - 1) / G * G);
    for (int64_t e = e0; e < e1; e += per) {
      h_wg_e0.push_back((int32_t)e);
      h_wg_panel.push_back(s_);
    }
  }
  h_wg_e0.push_back(pan[S]);
  TRY(upload_i32(c, &wg_e0, h_wg_e0.data(), h_wg_e0.size()));
  TRY(upload_i32(c, &wg_panel, h_wg_panel.data(), h_wg_panel.size()));
  TRY(dev_alloc(c, &dst->val, (size_t)total + 64));
  TRY(dev_alloc(c, &prod, (size_t)total + slack + 256));
  HIP_TRY(hipStreamSynchronize(s));  // (the host vectors die here)
  dst->v = PbView{rows, cols, S, B, gshift, panel_shift, (int)h_wg_panel.size(), dst->val, lidx, piece_dst, wg_e0, wg_panel,
                  bin_row0, d_bin_e0, nullptr, nullptr, nullptr, nullptr, prod};
  dst->v.wide = 1, dst->v.rib = rib, dst->v.step_lv = step_lv;
  dst->v.nser = nser, dst->v.ser_ptr = ser_ptr, dst->v.ser_row = ser_row, dst->v.ser_eptr = ser_eptr, dst->v.ser_slot = ser_slot;
  dst->np = total, dst->p_threads = panel_shift == 14 ? 1024 : 512, dst->pad = (double)total / (double)nnz;
  dst->on = true;
  plap("P workgroups");
  return 0;
}

int build_pb_device(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                    int cus, bool forced, std::string* why)
{
  if (pb_wants_wide(cols)) {  // wide bins first; a matrix they cannot hold (or cannot be built for here) goes on to the image-in-LDS bins
    std::string why_wide;
    const int rc = build_pb_wide_device(c, dst, rows, cols, h_off, d_off, d_idx, cus, forced, &why_wide);
    if (rc < 0 || (rc == 0 && dst->on)) return rc;
    if (rc == 1) return 1;  // (the host builds: build_pb takes the same decision)
  }
  const int64_t nnz = rows > 0 ? h_off[rows] : 0;
  if (rows <= 0 || cols <= 0 || nnz <= 0) { *why = "empty matrix"; return 0; }
  hipStream_t s = c->stream;
  const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto plap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cuopt_amd setup]     gather-free: %-14s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<void*> tmp;  // (the context's scratch blocks this construction holds: handed back, not freed, on every way out)
  struct Release {
    pdlpdev_ctx* c;
    std::vector<void*>& v;
    ~Release() { for (void* p : v) scratch_release(c, p); }
  } release_tmp{c, tmp};
  auto talloc = [&](void** p, size_t bytes) -> int {
    TRY(scratch_take(c, p, bytes));
    tmp.push_back(*p);
    return 0;
  };
  int* d_scal = nullptr;  // [0] longest row, [1] largest padded bin
  TRY(talloc((void**)&d_scal, 2 * sizeof(int)));
  HIP_TRY(hipMemsetAsync(d_scal, 0, 2 * sizeof(int), s));
  k_max_row_len<<<grid_of(rows), kT, 0, s>>>(rows, d_off, d_scal);
  int h_scal[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h_scal, d_scal, sizeof(h_scal), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int longest = h_scal[0];
  // a row is summed by ONE lane, left to right: fine up to a few hundred entries, a serial chain beyond
  if (longest > (forced ? kPbCap / 2 : 256)) { *why = "a row with " + std::to_string(longest) + " nonzeros"; return 0; }
  const int panel_shift = cols > (1 << 21) ? 14 : 13;
  const int S           = (int)(((int64_t)cols + (1 << panel_shift) - 1) >> panel_shift);
  if (S >= (1 << 18)) return 1;  // (the sort's words hold 18 bits of panel number)
  const bool use_hist = S <= kPbHistMax && cuopt_amd::tune_int("pb_hist", 1) != 0;  // (the panel counters of a bin fit in LDS; tests switch it off)
  // bins: consecutive rows, <= target nonzeros and <= kPbMaxRows rows; the target is lowered until every bin's padded image fits
  double target = 0.93 * kPbCap;
  int G = 8, gshift = 3;
  std::vector<int32_t> row0;
  int32_t *d_row0 = nullptr, *d_bin_size = nullptr;
  int B = 0;
  for (int iter = 0; iter < 24; ++iter) {
    row0.assign(1, 0);
    while (row0.back() < rows) {
      const int32_t r0 = row0.back();
      const int64_t lim = (int64_t)h_off[r0] + (int64_t)target;
      // (std::upper_bound over the whole offset array is two dozen cache misses per bin on a cold 40 MB array -- 30 ms at 1e7 rows:
      //  the answer lies within kPbMaxRows rows of r0, and near r0 + target / (nonzeros per row): bracket it from there)
      const int32_t key = (int32_t)std::min<int64_t>(lim, nnz);
      const int32_t cap = (int32_t)std::min<int64_t>((int64_t)rows + 1, (int64_t)r0 + kPbMaxRows + 2);
      int32_t lo_ = r0, hi_ = cap;  // the first index in [lo_, hi_) whose offset exceeds the key (hi_: none in the bracket)
      {
        int32_t g = (int32_t)std::min<int64_t>((int64_t)cap - 1, (int64_t)r0 + (int64_t)(target * (double)rows / (double)nnz));
        int32_t step = 8;
        if (h_off[g] > key) {
          hi_ = g;
          while (hi_ - step > r0 && h_off[hi_ - step] > key) hi_ -= step, step *= 2;
          lo_ = std::max(r0, hi_ - step);
        } else {
          lo_ = g + 1;
          while (lo_ + step < cap && h_off[lo_ + step - 1] <= key) lo_ += step, step *= 2;
          hi_ = std::min(cap, lo_ + step);
        }
      }
      int32_t r1 = (int32_t)(std::upper_bound(h_off + lo_, h_off + hi_, key) - h_off) - 1;
      r1 = std::min(std::max(r1, r0 + 1), std::min(rows, r0 + kPbMaxRows));
      row0.push_back(r1);
    }
    B = (int)row0.size() - 1;
    if (iter == 0) {
      G      = nnz / ((int64_t)S * B) >= 24 ? 8 : 4;
      gshift = G == 8 ? 3 : 2;
      // (a lower target means more bins, at most one per row)
      TRY(talloc((void**)&d_row0, ((size_t)rows + 1) * sizeof(int32_t)));
      TRY(talloc((void**)&d_bin_size, ((size_t)rows + 1) * sizeof(int32_t)));
    }
    HIP_TRY(hipMemcpyAsync(d_row0, row0.data(), ((size_t)B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_scal + 1, 0, sizeof(int), s));
    if (use_hist) k_pb_bin_hist<false><<<B, 512, (size_t)S * sizeof(int), s>>>(d_row0, d_off, d_idx, panel_shift, gshift, S, d_bin_size, d_scal + 1, nullptr);
    else k_pb_bin<0><<<B, kPbBinT, 0, s>>>(d_row0, d_off, d_idx, panel_shift, gshift, S, B, d_bin_size, d_scal + 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_scal + 1, d_scal + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int maxpad = h_scal[1];
    if (maxpad <= kPbCap) break;
    if (iter == 23) { *why = "bins do not converge"; return 0; }
    target *= std::min(0.97, 0.99 * (double)kPbCap / (double)maxpad);
  }
  plap("bins");
  // chunk sizes (bin-major, 16-bit), the bins' images, the panel-major starts of phase P
  uint16_t* d_cnt = nullptr;
  TRY(talloc((void**)&d_cnt, (size_t)B * S * sizeof(uint16_t)));
  HIP_TRY(hipMemsetAsync(d_scal + 1, 0, sizeof(int), s));
  if (use_hist) {
    k_pb_bin_hist<true><<<B, 512, (size_t)S * sizeof(int), s>>>(d_row0, d_off, d_idx, panel_shift, gshift, S, d_bin_size, d_scal + 1, d_cnt);
  } else {
    HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)B * S * sizeof(uint16_t), s));
    k_pb_bin<1><<<B, kPbBinT, 0, s>>>(d_row0, d_off, d_idx, panel_shift, gshift, S, B, d_bin_size, d_scal + 1, d_cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  }
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> bin_size(B), bin_e0((size_t)B + 1, 0);
  HIP_TRY(hipMemcpyAsync(bin_size.data(), d_bin_size, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  int64_t total = 0;
  for (int b = 0; b < B; ++b) {
    bin_e0[b] = (int32_t)total;
    total += bin_size[b];
    if (total >= ((int64_t)1 << 31) - 65536) { *why = "more than 2^31 padded entries"; return 0; }
  }
  bin_e0[B] = (int32_t)total;
  const int64_t cells = (int64_t)S * B;
  int32_t *d_sizes = nullptr, *d_pstart = nullptr, *d_bs = nullptr;
  TRY(talloc((void**)&d_sizes, (size_t)cells * sizeof(int32_t)));
  TRY(talloc((void**)&d_pstart, ((size_t)cells + 1) * sizeof(int32_t)));
  TRY(talloc((void**)&d_bs, ((size_t)(cells + 1) / 4096 + 2) * sizeof(int32_t)));
  k_pb_chunk_sizes<<<grid_of(cells), kT, 0, s>>>(d_cnt, S, B, gshift, d_sizes);
  TRY(dev_exclusive_scan(s, d_sizes, d_pstart, cells, d_bs));
  plap("chunk table");
  // the layout's own arrays
  int32_t *piece_dst = nullptr, *wg_e0 = nullptr, *wg_panel = nullptr, *bin_row0 = nullptr, *d_bin_e0 = nullptr, *bin_grp = nullptr, *grp_pos = nullptr;
  uint16_t *lidx = nullptr, *pos = nullptr, *d_epos = nullptr;
  uint32_t* sr = nullptr;
  double* prod = nullptr;
  TRY(dev_alloc(c, &dst->perm, (size_t)total + 64));
  TRY(dev_alloc(c, &piece_dst, (size_t)(total >> gshift) + 64));
  TRY(dev_alloc(c, &lidx, (size_t)total + 64));
  TRY(dev_alloc(c, &pos, (size_t)nnz + 128));
  TRY(dev_alloc(c, &sr, (size_t)rows + 64));
  TRY(dev_alloc(c, &dst->val, (size_t)total + 64));
  TRY(dev_alloc(c, &prod, (size_t)total + 256));
  TRY(talloc((void**)&d_epos, ((size_t)nnz + 64) * sizeof(uint16_t)));
  HIP_TRY(hipMemsetAsync(dst->perm, 0xFF, (size_t)total * sizeof(int32_t), s));  // padding slots: -1 (their lidx: 0, dev_alloc's zero fill)
  TRY(upload_i32(c, &bin_row0, row0.data(), row0.size()));
  TRY(upload_i32(c, &d_bin_e0, bin_e0.data(), bin_e0.size()));
  std::vector<int32_t> h_bin_grp((size_t)B + 1, 0);
  for (int b = 0; b < B; ++b) h_bin_grp[b + 1] = h_bin_grp[b] + (row0[b + 1] - row0[b] + 63) / 64;
  TRY(upload_i32(c, &bin_grp, h_bin_grp.data(), h_bin_grp.size()));
  TRY(dev_alloc(c, &grp_pos, (size_t)h_bin_grp[B] + 1));
  plap("alloc");
  k_pb_bin<2><<<B, kPbBinT, 0, s>>>(bin_row0, d_off, d_idx, panel_shift, gshift, S, B, nullptr, nullptr, nullptr, d_pstart, d_bin_e0, dst->perm, lidx, piece_dst, d_epos);
  HIP_TRY(hipGetLastError());
  plap("place");
  k_pb_rows<<<B, 1024, 0, s>>>(bin_row0, d_off, d_epos, bin_grp, sr, grp_pos, pos);
  HIP_TRY(hipGetLastError());
  const int32_t nnz32 = (int32_t)nnz;
  HIP_TRY(hipMemcpyAsync(grp_pos + h_bin_grp[B], &nnz32, sizeof(int32_t), hipMemcpyHostToDevice, s));
  plap("rows");
  // P workgroups: every panel's entries in Q parts (pieces are not split)
  int32_t* d_pan = nullptr;
  TRY(talloc((void**)&d_pan, ((size_t)S + 1) * sizeof(int32_t)));
  k_pb_panel_starts<<<grid_of(S + 1), kT, 0, s>>>(d_pstart, S, B, d_pan);
  std::vector<int32_t> pan((size_t)S + 1);
  HIP_TRY(hipMemcpyAsync(pan.data(), d_pan, pan.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<int32_t> h_wg_e0, h_wg_panel;
  const int Q = std::max(1, std::min(16, (4 * cus + S - 1) / S));
  for (int s_ = 0; s_ < S; ++s_) {
    const int64_t e0 = pan[s_], e1 = pan[s_ + 1];
    const int64_t per = std::max<int64_t>(G, ((e1 - e0 + Q - 1) / Q + G - 1) / G * G);
    for (int64_t e = e0; e < e1; e += per) {
      h_wg_e0.push_back((int32_t)e);
      h_wg_panel.push_back(s_);
    }
  }
  h_wg_e0.push_back((int32_t)total);
  TRY(upload_i32(c, &wg_e0, h_wg_e0.data(), h_wg_e0.size()));
  TRY(upload_i32(c, &wg_panel, h_wg_panel.data(), h_wg_panel.size()));
  HIP_TRY(hipStreamSynchronize(s));  // (the host vectors die here)
  const int p_threads = panel_shift == 14 ? 1024 : 512;
  dst->v = PbView{rows, cols, S, B, gshift, panel_shift, (int)h_wg_panel.size(), dst->val, lidx, piece_dst, wg_e0, wg_panel,
                  bin_row0, d_bin_e0, sr, bin_grp, grp_pos, pos, prod};
  dst->np = total, dst->p_threads = p_threads, dst->pad = (double)total / (double)nnz;
  dst->on = true;
  plap("P workgroups");
  return 0;
}

// ================================================================================================
// jagged rows + LDS column sets from a device-resident CSR (returns 1: not built here -- nothing was allocated, the host constructs it;
// 0: built, or "not worth it" with dst->on false and dst->saving set).  Every array is bit-identical to build_jag's (kernels_jag.hip),
// which stays the tests' reference construction:
//   * the partition: chunks of 4 * brows rows are cut independently, a workgroup per chunk grows block after block exactly like the
//     sampled estimate does (k_jag_estimate: rows in order while their DISTINCT columns fit the LDS window, an open-addressing table in
//     LDS, the host's two-step check near the limit) and prices every block's column set;
//   * per block: its short rows sorted by length (descending, ties by row: a bitonic sort of unique words), passes of 64 dealt to the
//     waves in snake order; the sizes go to the host for the prefix sums (a few thousand numbers), a second kernel fills the row
//     descriptors, the 16-bit LDS slots and the permutation along the jagged diagonals (a wave per pass, one ballot per diagonal);
//     a list-mode block's distinct columns are collected through the LDS table, sorted there, and searched for every entry's slot.
// ================================================================================================
namespace {
constexpr int kJagCutT = 1024, kJagStride = 512;
struct JagBlockMeta {
  int32_t end;           // first row behind the block
  int32_t wbase, wlen;   // contiguous column set (wlen 0: a list)
  int32_t ncols;         // distinct columns of a list-mode block
  long long refs, cost;
};

// One chunk of rows [c0, c1): blocks cut greedily from c0 on (jag_block_end with `row_cap` rows at most), each priced (jag_block_set).
__global__ void __launch_bounds__(kJagCutT)
k_jag_cut_chunk(int32_t chunk_rows, int32_t row_cap, int32_t wcap, int32_t rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                JagBlockMeta* __restrict__ meta /* [rows]: the blocks of chunk t from meta[t * chunk_rows] on */, int32_t* __restrict__ count)
{
  extern __shared__ int32_t tab[];  // 2 * wcap slots
  __shared__ int32_t pre[kJagStride + 1];
  __shared__ int distinct, stop, lo, hi, runs, parallel, parallel64, end_row, scratch[kJagCutT / 64 + 1];
  __shared__ long long refs;
  const uint32_t mask = (uint32_t)(2 * wcap - 1);
  const int t = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * chunk_rows, c1 = min((int64_t)rows, c0 + chunk_rows);
  int nblocks = 0;
  for (int64_t first = c0; first < c1;) {
    for (int i = t; i < 2 * wcap; i += kJagCutT) tab[i] = kEstEmpty;
    if (t == 0) distinct = 0, stop = 0, lo = 0x7fffffff, hi = -1, runs = 0, refs = 0;
    __syncthreads();
    const int64_t last = min(c1, first + row_cap);
    if (t == 0) end_row = (int)last;
    __syncthreads();
    for (int64_t q0 = first; q0 < last && !stop;) {
      // a stride of up to kJagStride rows whose lengths cannot overflow the window whatever they contain goes in at once (a band's
      // block: four strides instead of 32 chunks); else a chunk of 64 rows, in parallel or row by row as in k_jag_estimate
      int nr = (int)min((int64_t)kJagStride, last - q0);
      bool par;
      {
        int len = 0;
        if (t < nr) {
          len = off[q0 + t + 1] - off[q0 + t];
          len = len > kLongRow ? 0 : len;  // (long rows do not count)
        }
        int total_all = 0;
        const int ex = block_exclusive_scan<kJagCutT>(t < kJagStride ? len : 0, scratch, &total_all);
        if (t < nr) pre[t] = ex;
        if (t == 0) pre[nr] = total_all, parallel = distinct + total_all <= wcap;
        __syncthreads();
        par = parallel != 0;
        if (!par && nr > kEstChunk) {  // (the first 64 rows alone: their prefix sums are already there)
          nr = kEstChunk;
          // (a second word: a wave that has not yet read the stride's answer must not find this one in its place)
          if (t == 0) parallel64 = distinct + pre[nr] <= wcap;
          __syncthreads();
          par = parallel64 != 0;
        }
      }
      const int total = pre[nr];
      if (par) {
        int added = 0, mn = 0x7fffffff, mx = -1;
        // two lanes per row (rows are short: no search for an entry's row, a handful of inserts per lane)
        for (int r = t >> 1; r < nr; r += kJagCutT / 2) {
          const int len = pre[r + 1] - pre[r];
          const int k0  = off[q0 + r];
          for (int k = t & 1; k < len; k += 2) {
            const int32_t c = idx[k0 + k];
            mn = c < mn ? c : mn, mx = c > mx ? c : mx;
            added += est_insert(tab, mask, c);
          }
        }
        // (one LDS atomic per wave: every entry hammering the same two words was most of this kernel's time)
        for (int d = 1; d < 64; d <<= 1) {
          added += __shfl_xor(added, d, 64);
          mn = min(mn, __shfl_xor(mn, d, 64)), mx = max(mx, __shfl_xor(mx, d, 64));
        }
        if ((t & 63) == 0) {
          if (added) atomicAdd(&distinct, added);
          if (mx >= 0) atomicMin(&lo, mn), atomicMax(&hi, mx);
        }
        if (t == 0) refs += total;
        __syncthreads();
      } else {
        // near the limit: row by row, the host's rule (count the new columns before inserting any) -- by ONE wave, without
        // workgroup barriers (a row has at most kLongRow = 128 entries: two per lane; the table's operations of one wave are ordered)
        if (t < 64) {
          int dist = distinct, mn = 0x7fffffff, mx = -1;
          long long rf = 0;
          int stopped = 0, stop_at = 0;
          for (int i = 0; i < nr && !stopped; ++i) {
            const int len = pre[i + 1] - pre[i];
            if (len == 0) continue;
            const int k0 = off[q0 + i];
            const int32_t c0_ = t < len ? idx[k0 + t] : 0, c1_ = t + 64 < len ? idx[k0 + t + 64] : 0;
            if (dist + len > wcap) {
              int fr = (t < len && !est_contains(tab, mask, c0_)) + (t + 64 < len && !est_contains(tab, mask, c1_));
              for (int d = 1; d < 64; d <<= 1) fr += __shfl_xor(fr, d, 64);
              if (dist + fr > wcap) {
                stopped = 1, stop_at = i;
                break;
              }
            }
            int add = 0;
            if (t < len) add += est_insert(tab, mask, c0_), mn = min(mn, c0_), mx = max(mx, c0_);
            if (t + 64 < len) add += est_insert(tab, mask, c1_), mn = min(mn, c1_), mx = max(mx, c1_);
            for (int d = 1; d < 64; d <<= 1) add += __shfl_xor(add, d, 64);
            dist += add;
            rf += len;
          }
          for (int d = 1; d < 64; d <<= 1) mn = min(mn, __shfl_xor(mn, d, 64)), mx = max(mx, __shfl_xor(mx, d, 64));
          if (t == 0) {
            distinct = dist, refs += rf;
            if (mx >= 0) lo = min(lo, mn), hi = max(hi, mx);
            if (stopped) stop = 1, end_row = (int)(q0 + stop_at);
          }
        }
      }
      __syncthreads();
      q0 += nr;
    }
    __syncthreads();
    const int64_t end = max((int64_t)end_row, first + 1);  // (a row of <= kLongRow nonzeros always fits an empty set)
    long long cost = 0;
    int32_t wbase = 0, wlen = 0, ncols = 0;
    if (hi >= 0) {
      if ((long long)hi - lo + 1 <= wcap) {
        wbase = lo, wlen = hi - lo + 1;
        cost  = 1 + wlen / 16;
      } else {
        int mine = 0;
        for (int i = t; i < 2 * wcap; i += kJagCutT) {
          const int32_t c = tab[i];
          if (c != kEstEmpty && (c == 0 || !est_contains(tab, mask, c - 1))) ++mine;
        }
        if (mine) atomicAdd(&runs, mine);
        __syncthreads();
        cost  = (long long)runs + distinct / 16;
        ncols = distinct;
      }
    }
    if (t == 0) meta[c0 + nblocks] = JagBlockMeta{(int32_t)end, wbase, wlen, ncols, refs, cost};
    ++nblocks;
    first = end;
    __syncthreads();
  }
  if (t == 0) count[blockIdx.x] = nblocks;
}

constexpr int kJagSortN = 4096;  // rows of a block (16 waves x 256)
__device__ __forceinline__ int jag_wave_of_pass(int p, int waves) { return ((p / waves) & 1) ? waves - 1 - (p % waves) : p % waves; }

// the block's short rows as sorted words ((kLongRow - len) << 12 | local row; rows without a place in the passes sort behind them):
// returns their number
__device__ __forceinline__ int jag_sort_rows(uint32_t* w, int r0, int nr, const int32_t* __restrict__ off)
{
  __shared__ int ns_s;
  int n2 = 1;
  while (n2 < nr) n2 <<= 1;
  if (threadIdx.x == 0) ns_s = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < n2; i += kJagCutT) {
    uint32_t word = 0xFFFFFFFFu;
    if (i < nr) {
      const int len = off[r0 + i + 1] - off[r0 + i];
      if (len >= 1 && len <= kLongRow) word = (uint32_t)(kLongRow - len) << 12 | (uint32_t)i, ++mine;
    }
    w[i] = word;
  }
  if (mine) atomicAdd(&ns_s, mine);
  __syncthreads();
  lds_bitonic_sort<kJagCutT>(w, n2);
  return ns_s;
}

// sizes: rows and entries of every (block, wave) share, the block's long rows
__global__ void __launch_bounds__(kJagCutT)
k_jag_block_sizes(const int32_t* __restrict__ row0, const int32_t* __restrict__ off, int waves, int32_t* __restrict__ gsr, int32_t* __restrict__ gent,
                  int32_t* __restrict__ nlong)
{
  __shared__ uint32_t w[kJagSortN];
  __shared__ int sr_s[16], ent_s[16], long_s;
  const int b = blockIdx.x, t = threadIdx.x;
  const int r0 = row0[b], nr = row0[b + 1] - r0;
  if (t < 16) sr_s[t] = 0, ent_s[t] = 0;
  if (t == 0) long_s = 0;
  __syncthreads();
  int nl = 0;
  for (int i = t; i < nr; i += kJagCutT) nl += off[r0 + i + 1] - off[r0 + i] > kLongRow;
  if (nl) atomicAdd(&long_s, nl);
  const int ns = jag_sort_rows(w, r0, nr, off);
  for (int base = 0; base < ns; base += kJagCutT) {
    const int i = base + t;
    const int len = i < ns ? kLongRow - (int)(w[i] >> 12) : 0;
    int cnt = i < ns ? 1 : 0, ent = len;
    for (int d = 1; d < 64; d <<= 1) cnt += __shfl_xor(cnt, d, 64), ent += __shfl_xor(ent, d, 64);
    if ((t & 63) == 0 && cnt) {
      const int wv = jag_wave_of_pass(i >> 6, waves);
      atomicAdd(&sr_s[wv], cnt), atomicAdd(&ent_s[wv], ent);
    }
  }
  __syncthreads();
  if (t < waves) gsr[(size_t)b * waves + t] = sr_s[t], gent[(size_t)b * waves + t] = ent_s[t];
  if (t == 0) nlong[b] = long_s;
}

// fill: row descriptors, LDS slots and permutation along the jagged diagonals; a list-mode block's column list; the long rows
__global__ void __launch_bounds__(kJagCutT)
k_jag_block_fill(const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int waves, int32_t wcap,
                 const int32_t* __restrict__ win, const int32_t* __restrict__ set_ptr, const int32_t* __restrict__ tile_sr, const int32_t* __restrict__ tile_e,
                 const int32_t* __restrict__ lr_ptr, uint32_t* __restrict__ sr, uint16_t* __restrict__ slot, int32_t* __restrict__ perm,
                 int32_t* __restrict__ set_col, int32_t* __restrict__ lr_row)
{
  extern __shared__ int32_t lds[];  // list mode: 2 * wcap table slots, then wcap sorted columns
  __shared__ uint32_t w[kJagSortN];
  __shared__ int sr_base[64], e_base[64], pass_cnt[64], pass_ent[64], ncols_s, scratch[17];
  const int b = blockIdx.x, t = threadIdx.x;
  const int r0 = row0[b], nr = row0[b + 1] - r0;
  const int32_t wbase = win[2 * b], wlen = win[2 * b + 1];
  int32_t* tab  = lds;
  int32_t* cols = lds + 2 * wcap;
  const uint32_t mask = (uint32_t)(2 * wcap - 1);
  int ncols = 0;
  if (!wlen) {
    // the distinct columns of the block's short rows, sorted
    for (int i = t; i < 2 * wcap; i += kJagCutT) tab[i] = kEstEmpty;
    if (t == 0) ncols_s = 0;
    __syncthreads();
    for (int i = 0; i < nr; ++i) {  // (a row per trip: rows are short, the lanes take its entries)
      const int k0 = off[r0 + i], len = off[r0 + i + 1] - k0;
      if (len > kLongRow) continue;
      if (t < len) (void)est_insert(tab, mask, idx[k0 + t]);
    }
    __syncthreads();
    for (int i = t; i < 2 * wcap; i += kJagCutT)
      if (tab[i] != kEstEmpty) cols[atomicAdd(&ncols_s, 1)] = tab[i];
    __syncthreads();
    ncols = ncols_s;
    int n2 = 1;
    while (n2 < ncols) n2 <<= 1;
    for (int i = ncols + t; i < n2; i += kJagCutT) cols[i] = 0x7fffffff;
    __syncthreads();
    lds_bitonic_sort<kJagCutT>((uint32_t*)cols, n2);  // (columns are non-negative: the unsigned order is theirs)
    for (int i = t; i < ncols; i += kJagCutT) set_col[set_ptr[b] + i] = cols[i];
  }
  // long rows, in row order
  {
    int carry = 0;
    for (int base = 0; base < nr; base += kJagCutT) {
      const int i = base + t;
      const int is_long = i < nr && off[r0 + i + 1] - off[r0 + i] > kLongRow;
      int total = 0;
      const int pre = block_exclusive_scan<kJagCutT>(is_long, scratch, &total);
      if (is_long) lr_row[lr_ptr[b] + carry + pre] = r0 + i;
      carry += total;
    }
  }
  const int ns = jag_sort_rows(w, r0, nr, off);
  const int npass = (ns + 63) >> 6;
  if (t < 64) pass_cnt[t] = 0, pass_ent[t] = 0;
  __syncthreads();
  for (int base = 0; base < ns; base += kJagCutT) {
    const int i = base + t;
    int cnt = i < ns ? 1 : 0, ent = i < ns ? kLongRow - (int)(w[i] >> 12) : 0;
    for (int d = 1; d < 64; d <<= 1) cnt += __shfl_xor(cnt, d, 64), ent += __shfl_xor(ent, d, 64);
    if ((t & 63) == 0 && cnt) pass_cnt[i >> 6] = cnt, pass_ent[i >> 6] = ent;
  }
  __syncthreads();
  if (t == 0) {
    int srpos[16], epos[16];
    for (int wv = 0; wv < waves; ++wv) srpos[wv] = tile_sr[(size_t)b * waves + wv], epos[wv] = tile_e[(size_t)b * waves + wv];
    for (int p = 0; p < npass; ++p) {
      const int wv = jag_wave_of_pass(p, waves);
      sr_base[p] = srpos[wv], e_base[p] = epos[wv];
      srpos[wv] += pass_cnt[p], epos[wv] += pass_ent[p];
    }
  }
  __syncthreads();
  const int lane = t & 63;
  for (int p = t >> 6; p < npass; p += kJagCutT / 64) {
    const int i    = p * 64 + lane;
    const bool live = i < ns;
    const int row  = live ? (int)(w[i] & 4095) : 0;
    const int len  = live ? kLongRow - (int)(w[i] >> 12) : 0;
    if (live) sr[sr_base[p] + lane] = (uint32_t)(len - 1) << 16 | (uint32_t)row;
    const int k_row = off[r0 + row];
    const int kmax  = __shfl(len, 0, 64);
    int e = e_base[p];
    for (int k = 0; k < kmax; ++k) {
      const unsigned long long m_ = __ballot(len > k);
      if (len > k) {
        const int32_t c = idx[k_row + k];
        int s_;
        if (wlen) {
          s_ = c - wbase;
        } else {
          int a = 0, z = ncols;  // the slot of c in the sorted list
          while (z - a > 1) {
            const int mid = (a + z) >> 1;
            if (cols[mid] <= c) a = mid; else z = mid;
          }
          s_ = a;
        }
        slot[e + lane] = (uint16_t)s_;
        perm[e + lane] = k_row + k;
      }
      e += __popcll(m_);
    }
  }
}
}  // namespace

int build_jag_device(pdlpdev_ctx* c, pdlpdev_ctx::Jag* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                     const double* d_val, int mode, int cus)
{
  const int64_t nnz = rows > 0 ? h_off[rows] : 0;
  if (rows <= 0 || cols <= 0 || nnz <= 0) return 0;
  int G = 0, waves = 8, wcap = 0, brows = 0;
  if (!jag_geometry(rows, mode, &G, &waves, &wcap, &brows)) return 0;
  if (waves != 8) return 1;  // (the 16-wave geometry's table + list do not fit one workgroup's LDS: CUOPT_AMD_TUNE=jag_waves=16 builds on the host)
  hipStream_t s = c->stream;
  const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto plap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cuopt_amd setup]     jagged: %-18s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  std::vector<void*> tmp;
  struct Free {
    std::vector<void*>& v;
    ~Free() { for (void* p : v) (void)hipFree(p); }
  } free_tmp{tmp};
  auto talloc = [&](void** p, size_t bytes) -> int {
    HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 256)));
    tmp.push_back(*p);
    return 0;
  };
  {
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(done.begin(), done.end(), c->device) == done.end()) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_jag_cut_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 * 4));
      HIP_TRY(hipFuncSetAttribute((const void*)k_jag_block_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8192 * 4));
      done.push_back(c->device);
    }
  }
  const int slots = cus * 2;  // workgroups resident at once: 80 KiB of LDS each
  const int32_t chunk_rows = 4 * brows;
  const int nchunks        = (int)(((int64_t)rows + chunk_rows - 1) / chunk_rows);
  JagBlockMeta* d_meta = nullptr;
  int32_t* d_count     = nullptr;
  TRY(talloc((void**)&d_meta, ((size_t)rows + 1) * sizeof(JagBlockMeta)));
  TRY(talloc((void**)&d_count, (size_t)nchunks * sizeof(int32_t)));
  std::vector<JagBlockMeta> blocks;
  auto partition = [&](int32_t row_cap, std::vector<JagBlockMeta>* out) -> int {
    k_jag_cut_chunk<<<nchunks, kJagCutT, (size_t)2 * wcap * sizeof(int32_t), s>>>(chunk_rows, row_cap, wcap, rows, d_off, d_idx, d_meta, d_count);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> count(nchunks);
    HIP_TRY(hipMemcpyAsync(count.data(), d_count, (size_t)nchunks * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    out->clear();
    for (int t = 0; t < nchunks; ++t) {
      const size_t at = out->size();
      out->resize(at + count[t]);
      HIP_TRY(hipMemcpyAsync(out->data() + at, d_meta + (size_t)t * chunk_rows, (size_t)count[t] * sizeof(JagBlockMeta), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
  };
  TRY(partition(brows, &blocks));
  // (as build_jag: when the column sets cut the blocks short of a whole round of resident workgroups, smaller blocks that fill the same
  // number of rounds are strictly better)
  if (slots > 0 && (int)blocks.size() > slots) {
    const int nb = (int)blocks.size(), rounds = (nb + slots - 1) / slots;
    if ((double)nb < 0.9 * (double)rounds * (double)slots) {
      int32_t cap = (int32_t)std::ceil((double)rows / (0.97 * (double)rounds * (double)slots));
      cap         = std::max<int32_t>(64, (cap + 63) & ~63);
      if (cap < brows) {
        std::vector<JagBlockMeta> alt;
        TRY(partition(cap, &alt));
        if (((int)alt.size() + slots - 1) / slots <= rounds) blocks.swap(alt);
      }
    }
  }
  plap("partition");
  const int nblk = (int)blocks.size(), ngroups = nblk * waves;
  std::vector<int32_t> row0((size_t)nblk + 1, 0), win((size_t)2 * nblk, 0), set_ptr((size_t)nblk + 1, 0);
  long long refs = 0, cost = 0;
  for (int b = 0; b < nblk; ++b) {
    row0[b + 1] = blocks[b].end;
    refs += blocks[b].refs, cost += blocks[b].cost;
    win[2 * b] = blocks[b].wbase, win[2 * b + 1] = blocks[b].wlen;
    set_ptr[b + 1] = set_ptr[b] + blocks[b].ncols;
  }
  dst->saving = refs ? 1.0 - (double)cost / (double)refs : 0.0;
  if (mode == 0 && dst->saving < 0.5) return 0;
  int32_t *d_row0 = nullptr, *d_gsr = nullptr, *d_gent = nullptr, *d_nlong = nullptr;
  TRY(upload_i32(c, &d_row0, row0.data(), row0.size()));
  TRY(talloc((void**)&d_gsr, (size_t)ngroups * sizeof(int32_t)));
  TRY(talloc((void**)&d_gent, (size_t)ngroups * sizeof(int32_t)));
  TRY(talloc((void**)&d_nlong, (size_t)nblk * sizeof(int32_t)));
  k_jag_block_sizes<<<nblk, kJagCutT, 0, s>>>(d_row0, d_off, waves, d_gsr, d_gent, d_nlong);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> gsr(ngroups), gent(ngroups), nlong(nblk);
  HIP_TRY(hipMemcpyAsync(gsr.data(), d_gsr, (size_t)ngroups * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(gent.data(), d_gent, (size_t)ngroups * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(nlong.data(), d_nlong, (size_t)nblk * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<int32_t> tile_e((size_t)ngroups + 1, 0), tile_sr((size_t)ngroups + 1, 0), lr_ptr((size_t)nblk + 1, 0);
  for (int g = 0; g < ngroups; ++g) {
    tile_sr[g + 1] = tile_sr[g] + gsr[g];
    tile_e[g + 1]  = (int32_t)((int64_t)tile_e[g] + gent[g]);
  }
  for (int b = 0; b < nblk; ++b) lr_ptr[b + 1] = lr_ptr[b] + nlong[b];
  const size_t nsr = (size_t)tile_sr[ngroups], nent = (size_t)tile_e[ngroups];
  plap("sizes");
  int32_t *tile_e_d = nullptr, *tile_sr_d = nullptr, *win_d = nullptr, *set_ptr_d = nullptr, *set_col_d = nullptr, *lr_ptr_d = nullptr, *lr_row_d = nullptr;
  uint32_t* sr_d    = nullptr;
  uint16_t* slot_d  = nullptr;
  TRY(upload_i32(c, &tile_e_d, tile_e.data(), tile_e.size()));
  TRY(upload_i32(c, &tile_sr_d, tile_sr.data(), tile_sr.size()));
  TRY(upload_i32(c, &win_d, win.data(), win.size()));
  TRY(upload_i32(c, &set_ptr_d, set_ptr.data(), set_ptr.size()));
  TRY(upload_i32(c, &lr_ptr_d, lr_ptr.data(), lr_ptr.size()));
  TRY(dev_alloc(c, &set_col_d, (size_t)set_ptr[nblk] + 8));
  TRY(dev_alloc(c, &lr_row_d, (size_t)lr_ptr[nblk] + 1));
  TRY(dev_alloc(c, &dst->perm, nent + 8));
  TRY(dev_alloc(c, &sr_d, nsr + 8));
  TRY(dev_alloc(c, &slot_d, nent + 64));
  TRY(dev_alloc(c, &dst->val, nent + 8));
  k_jag_block_fill<<<nblk, kJagCutT, (size_t)3 * wcap * sizeof(int32_t), s>>>(d_row0, d_off, d_idx, waves, wcap, win_d, set_ptr_d, tile_sr_d, tile_e_d, lr_ptr_d, sr_d,
                                                                               slot_d, dst->perm, set_col_d, lr_row_d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));  // (the host vectors die here)
  plap("fill");
  dst->v    = JagView{rows, waves, ngroups, nblk, lr_ptr[nblk], d_row0, tile_e_d, tile_sr_d, sr_d, slot_d, dst->val,
                      win_d, set_ptr_d, set_col_d, lr_ptr_d, lr_row_d, d_off, d_idx, d_val};
  dst->nent = (int64_t)nent;
  dst->on   = true;
  return 0;
}

