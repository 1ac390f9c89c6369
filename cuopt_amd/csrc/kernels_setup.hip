// Device-side set-up (see pdlp_setup.hpp): radix sort / scan primitives, CSR transpose, the structure-finding analysis pass and the
// permuted CSR pair, all from ONE upload of A (the layouts' constructions: kernels_layout_build.hip).  gfx950, wave64.
#include <system_error>
#include <thread>
#include <hip/hip_runtime.h>

#include "pdlp_setup.hpp"
#include "setup_primitives.hpp"

namespace {

// ================================================================================================
// exclusive scan of int32: out[i] = sum_{j < i} in[j] for i in [0, n]  (n + 1 outputs; out[n] = total)
// ================================================================================================
constexpr int kScanTile = 1024 * 4;

__global__ void __launch_bounds__(1024) k_scan_local(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                     int32_t* __restrict__ block_sums)
{
  __shared__ int scratch[17];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 4;
  int v[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i = base + q;
    v[q] = i < n ? in[i] : 0;
    s += v[q];
  }
  int total = 0;
  int pre = block_exclusive_scan<1024>(s, scratch, &total);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i = base + q;
    if (i <= n) out[i] = pre;
    pre += v[q];
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_scan_sums(int32_t* __restrict__ block_sums, int nb)
{
  __shared__ int scratch[17];
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_sums[i] : 0;
    int total = 0;
    const int pre = block_exclusive_scan<1024>(v, scratch, &total);
    if (i < nb) block_sums[i] = carry + pre;
    carry += total;
  }
}

__global__ void __launch_bounds__(1024) k_scan_add(int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ block_sums)
{
  const int add = block_sums[blockIdx.x];
  if (add == 0) return;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (base + q <= n) out[base + q] += add;
}

}  // namespace

// block_sums: >= (n + 1) / 4096 + 1 ints
int dev_exclusive_scan(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, int32_t* block_sums)
{
  const int nb = (int)((n + 1 + kScanTile - 1) / kScanTile);
  k_scan_local<<<nb, 1024, 0, s>>>(in, out, n, block_sums);
  if (nb > 1) {
    k_scan_sums<<<1, 1024, 0, s>>>(block_sums, nb);
    k_scan_add<<<nb, 1024, 0, s>>>(out, n, block_sums);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

namespace {

// ================================================================================================
// stable LSD radix sort of (key, value) pairs of uint32, 8 bits per pass
// ================================================================================================
constexpr int kRsTile = 4096;  // items per workgroup: 4 waves x 16 rounds x 64 lanes, wave w owns items [w * 1024, (w + 1) * 1024) of the tile

__global__ void __launch_bounds__(kT) k_rs_hist(const uint32_t* __restrict__ keys, int64_t n, int shift, int nblk,
                                                int32_t* __restrict__ hist /* [256][nblk] */)
{
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll 4
  for (int r = 0; r < kRsTile / kT; ++r) {
    const int64_t i = base + r * kT + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of the digit's row of the histogram in place, its total to totals[digit]
__global__ void __launch_bounds__(1024) k_rs_scan_digit(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ totals)
{
  __shared__ int scratch[17];
  int32_t* row = hist + (size_t)blockIdx.x * nblk;
  int carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblk ? row[i] : 0;
    int total = 0;
    const int pre = block_exclusive_scan<1024>(v, scratch, &total);
    if (i < nblk) row[i] = carry + pre;
    carry += total;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// vals_in == nullptr: the value of item i is i
__global__ void __launch_bounds__(kT) k_rs_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                   uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, int64_t n, int shift,
                                                   int nblk, const int32_t* __restrict__ hist, const int32_t* __restrict__ totals)
{
  __shared__ int wcount[4][256];
  __shared__ int gbase[256];
  __shared__ int scratch[5];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
#pragma unroll
  for (int q = 0; q < 4; ++q) wcount[q][t] = 0;
  {
    // where digit t starts in the output: exclusive scan of the digit totals, + this workgroup's offset inside the digit
    const int pre = block_exclusive_scan<kT>(totals[t], scratch, nullptr);
    gbase[t]      = pre + hist[(size_t)t * nblk + blockIdx.x];
  }
  __syncthreads();
  constexpr int R = kRsTile / kT;  // 16 rounds
  uint32_t key[R], val[R];
  int rk[R];
  const int64_t base = (int64_t)blockIdx.x * kRsTile + (int64_t)w * (kRsTile / 4);
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i  = base + r * 64 + lane;
    const bool valid = i < n;
    key[r]           = valid ? kin[i] : 0xFFFFFFFFu;
    val[r]           = valid ? (vin ? vin[i] : (uint32_t)i) : 0u;
    const unsigned d = (key[r] >> shift) & 255u;
    // lanes of this wave that hold the same digit (a ballot per bit of the digit)
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit              = (d >> b) & 1u;
      const unsigned long long bl = __ballot(valid && bit);
      same &= bit ? bl : ~bl;
    }
    const int rank = __popcll(same & below), cnt = __popcll(same);
    int pre = 0;
    if (valid && rank == 0) pre = atomicAdd(&wcount[w][d], cnt);  // the group's first lane reserves the group's slots
    pre   = __shfl(pre, same ? __ffsll((long long)same) - 1 : 0, 64);
    rk[r] = pre + rank;  // position among this wave's items of digit d, in item order
  }
  __syncthreads();
  {
    // offsets of the waves inside (workgroup, digit): wave order = item order
    const int c0 = wcount[0][t], c1 = wcount[1][t], c2 = wcount[2][t];
    wcount[0][t] = 0, wcount[1][t] = c0, wcount[2][t] = c0 + c1, wcount[3][t] = c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = base + r * 64 + lane;
    if (i < n) {
      const unsigned d = (key[r] >> shift) & 255u;
      const int64_t p  = (int64_t)gbase[d] + wcount[w][d] + rk[r];
      kout[p] = key[r], vout[p] = val[r];
    }
  }
}

struct SortBufs {
  uint32_t *k[2] = {nullptr, nullptr}, *v[2] = {nullptr, nullptr};
  int32_t *hist = nullptr, *totals = nullptr;
  int64_t capacity = 0;
};

int sort_bufs_take(DevArena& ar, SortBufs* B, int64_t n)
{
  const int nblk = (int)((n + kRsTile - 1) / kRsTile) + 1;
  B->k[0] = ar.take<uint32_t>((size_t)n), B->k[1] = ar.take<uint32_t>((size_t)n);
  B->v[0] = ar.take<uint32_t>((size_t)n), B->v[1] = ar.take<uint32_t>((size_t)n);
  B->hist = ar.take<int32_t>((size_t)256 * nblk), B->totals = ar.take<int32_t>(256);
  B->capacity = n;
  if (!B->k[0] || !B->k[1] || !B->v[0] || !B->v[1] || !B->hist || !B->totals) return fail(-2, "device set-up: workspace too small for the sort buffers");
  return 0;
}

int bits_for(uint32_t max_key)
{
  int b = 1;
  while (b < 32 && (max_key >> b)) ++b;
  return b;
}

// Sorts n pairs by the low `bits` bits of the keys (stable).  keys_in / vals_in are not modified (vals_in null: iota); the result is in
// B.k[*slot], B.v[*slot].
int radix_sort_pairs(hipStream_t s, int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, SortBufs& B, int bits, int* slot)
{
  if (n > B.capacity) return fail(-2, "device set-up: sort buffers too small");
  const int passes = std::max(1, (bits + 7) / 8);
  const int nblk   = (int)std::max<int64_t>(1, (n + kRsTile - 1) / kRsTile);
  const uint32_t* kin = keys_in;
  const uint32_t* vin = vals_in;
  for (int p = 0; p < passes; ++p) {
    const int o = p & 1;
    k_rs_hist<<<nblk, kT, 0, s>>>(kin, n, 8 * p, nblk, B.hist);
    k_rs_scan_digit<<<256, 1024, 0, s>>>(B.hist, nblk, B.totals);
    k_rs_scatter<<<nblk, kT, 0, s>>>(kin, vin, B.k[o], B.v[o], n, 8 * p, nblk, B.hist, B.totals);
    kin = B.k[o], vin = B.v[o];
  }
  *slot = (passes - 1) & 1;
  HIP_TRY(hipGetLastError());
  return 0;
}

// ================================================================================================
// CSR -> CSR of the transpose
// ================================================================================================
// off_out[j] = first position of key >= j in the ascending keys (j in [0, nkeys]): the row offsets of a CSR from its sorted row ids
__global__ void __launch_bounds__(kT) k_offsets_from_sorted(const uint32_t* __restrict__ keys, int64_t n, int32_t nkeys, int32_t* __restrict__ off_out)
{
  for (int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x; q <= n; q += (int64_t)gridDim.x * kT) {
    const int64_t prev = q == 0 ? -1 : (int64_t)keys[q - 1];
    const int64_t cur  = q == n ? (int64_t)nkeys : (int64_t)keys[q];
    for (int64_t j = prev + 1; j <= cur; ++j) off_out[j] = (int32_t)q;
  }
}

__device__ __forceinline__ int32_t row_of(const int32_t* __restrict__ off, int32_t rows, int64_t k)
{
  // largest r with off[r] <= k (rows may be empty: upper bound - 1)
  int32_t lo = 0, hi = rows;  // invariant: off[lo] <= k < off[hi]
  while (hi - lo > 1) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if ((int64_t)off[mid] <= k) lo = mid; else hi = mid;
  }
  return lo;
}

// row id of every nonzero, in CSR order: a workgroup owns 4096 consecutive entries, narrows the row range once (two searches), and
// its threads search inside it (coalesced, a dozen steps over lines the whole workgroup shares)
__global__ void __launch_bounds__(kT) k_row_ids(int64_t nnz, const int32_t* __restrict__ off, int32_t rows, uint32_t* __restrict__ rowid)
{
  __shared__ int32_t range[2];
  const int64_t k0 = (int64_t)blockIdx.x * 4096, k1 = min(nnz, k0 + 4096);
  if (threadIdx.x < 2) range[threadIdx.x] = row_of(off, rows, threadIdx.x == 0 ? k0 : k1 - 1);
  __syncthreads();
  const int32_t r_lo = range[0], r_hi = range[1] + 1;  // off[r_lo] <= k < off[r_hi] for every k of the workgroup
  for (int64_t k = k0 + threadIdx.x; k < k1; k += kT) {
    int32_t lo = r_lo, hi = r_hi;
    while (hi - lo > 1) {
      const int32_t mid = lo + ((hi - lo) >> 1);
      if ((int64_t)off[mid] <= k) lo = mid; else hi = mid;
    }
    rowid[k] = (uint32_t)lo;
  }
}

// one entry of the transpose per thread: src[q] = position in the source CSR
__global__ void __launch_bounds__(kT) k_transpose_finish(const uint32_t* __restrict__ src, int64_t nnz, const uint32_t* __restrict__ rowid,
                                                         const double* __restrict__ a_val, int32_t* __restrict__ t_idx,
                                                         double* __restrict__ t_val)
{
  for (int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * kT) {
    const uint32_t k = src[q];
    t_idx[q] = (int32_t)rowid[k];
    t_val[q] = a_val[k];
  }
}


// A (rows x cols, CSR on the device) -> its transpose; rows ascending inside every row of the result (what cusparseCsr2cscEx2
// gives the reference, problem.cu:277-309): the stable sort by column keeps the source order, and the source is in row order
int dev_transpose(hipStream_t s, DevArena& ar, int32_t rows, int32_t cols, int64_t nnz, const DevCsr& A, const DevCsr& T)
{
  const size_t mark = ar.mark();
  SortBufs B;
  TRY(sort_bufs_take(ar, &B, std::max<int64_t>(nnz, 1)));
  int slot = 0;
  if (nnz > 0) TRY(radix_sort_pairs(s, nnz, (const uint32_t*)A.idx, nullptr, B, bits_for((uint32_t)std::max(cols - 1, 1)), &slot));
  k_offsets_from_sorted<<<grid_of(nnz + 1), kT, 0, s>>>(B.k[slot], nnz, cols, T.off);
  if (nnz > 0) {
    uint32_t* rowid = B.k[slot ^ 1];  // (the sort's other key buffer is free)
    k_row_ids<<<(int)((nnz + 4095) / 4096), kT, 0, s>>>(nnz, A.off, rows, rowid);
    k_transpose_finish<<<grid_of(nnz), kT, 0, s>>>(B.v[slot], nnz, rowid, A.val, T.idx, T.val);
  }
  HIP_TRY(hipGetLastError());
  ar.release(mark);  // (stream order: the next user of the arena is enqueued behind these kernels)
  return 0;
}

// ================================================================================================
// the analysis pass: breadth-first levels and seeded cells on the bipartite row-column graph
// ================================================================================================
constexpr int kStage = 4096;  // vertices a workgroup stages in LDS before it reserves room in the next frontier

// appends the workgroup's staged vertices to the next frontier (one reservation per workgroup and flush)
__device__ __forceinline__ void flush_stage(int* stage, int* stage_n, int32_t* __restrict__ out, int32_t* __restrict__ out_count)
{
  __shared__ int base;
  __syncthreads();
  const int cnt = min(*stage_n, kStage);
  if (threadIdx.x == 0) base = cnt ? atomicAdd(out_count, cnt) : 0;
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += kT) out[base + i] = stage[i];
  __syncthreads();
  if (threadIdx.x == 0) *stage_n = 0;
  __syncthreads();
}

// pos_out[v] = mean of the neighbours' positions (visited, non-hub neighbours only; CSR order: reproducible), own position kept
// when there is none; hubs and unvisited vertices keep -1
__global__ void __launch_bounds__(kT) k_barycentre(int32_t count, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                   const float* __restrict__ pos_self, const float* __restrict__ pos_other,
                                                   float* __restrict__ pos_out)
{
  for (int64_t v = (int64_t)blockIdx.x * kT + threadIdx.x; v < count; v += (int64_t)gridDim.x * kT) {
    const float own = pos_self[v];
    float out       = own;
    if (own >= 0.0f) {
      float sum = 0.0f;
      int cnt   = 0;
      for (int32_t k = off[v]; k < off[v + 1]; ++k) {
        const float p = pos_other[idx[k]];
        if (p >= 0.0f) sum += p, ++cnt;
      }
      if (cnt) out = sum / (float)cnt;
    }
    pos_out[v] = out;
  }
}

// start of the barycentre sweeps: the coordinate of a vertex's cell along the quotient graph's chain (-1: hub / not reached)
__global__ void __launch_bounds__(kT) k_cell_pos(int32_t count, const int32_t* __restrict__ label, const float* __restrict__ coord, float* __restrict__ pos)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const int32_t l = label[i];
    pos[i]          = l < 0 ? -1.0f : coord[l];
  }
}

__global__ void __launch_bounds__(kT) k_pos_to_key(int32_t count, const float* __restrict__ pos, float scale, uint32_t last_key, uint32_t* __restrict__ key)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const float p = pos[i];
    key[i]        = p >= 0.0f ? min((uint32_t)(p * scale), last_key - 1u) : last_key;
  }
}

// ---- seeded cells: a breadth-first search from many seeds at once; a vertex joins the cell that reaches it first, ties to the
// smaller cell id (atomicMin on depth << 24 | cell: all claims of one round carry the same depth) -> a reproducible partition
constexpr uint32_t kCellUnvisited = 0xFFFFFFFFu, kCellHub = 0u;

__global__ void __launch_bounds__(kT) k_cell_init(int32_t count, const int32_t* __restrict__ off, int32_t hub_len, uint32_t* __restrict__ key)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const int32_t len = off[i + 1] - off[i];
    key[i]            = len > hub_len ? kCellHub : kCellUnvisited;
  }
}

// seed t (cell id t + 1): the first row of [t * spacing, (t + 1) * spacing) that has entries and is no hub
__global__ void __launch_bounds__(kT) k_cell_seed(int32_t rows, int32_t spacing, int32_t ncells, const int32_t* __restrict__ off,
                                                  uint32_t* __restrict__ key, int32_t* __restrict__ frontier, int32_t* __restrict__ counts)
{
  const int32_t t = blockIdx.x * kT + threadIdx.x;
  int32_t seed    = -1;
  if (t < ncells) {
    const int64_t r0 = (int64_t)t * spacing, r1 = min((int64_t)rows, r0 + spacing);
    for (int64_t r = r0; r < r1 && r < r0 + 64; ++r)
      if (off[r + 1] > off[r] && key[r] == kCellUnvisited) {
        seed = (int32_t)r;
        break;
      }
  }
  const unsigned long long have = __ballot(seed >= 0);
  if (have) {
    const int lane = threadIdx.x & 63;
    int base       = 0;
    if (lane == __ffsll((long long)have) - 1) base = atomicAdd(counts, __popcll(have));
    base = __shfl(base, __ffsll((long long)have) - 1, 64);
    if (seed >= 0) {
      key[seed] = (uint32_t)(t + 1);  // depth 0
      frontier[base + __popcll(have & (lane == 0 ? 0ull : (~0ull >> (64 - lane))))] = seed;
    }
  }
}

__global__ void __launch_bounds__(kT) k_cell_expand(const int32_t* __restrict__ fr, const int32_t* __restrict__ fr_count,
                                                    const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                    const uint32_t* __restrict__ key_self, uint32_t* __restrict__ key_other,
                                                    int32_t* __restrict__ out, int32_t* __restrict__ out_count)
{
  __shared__ int stage[kStage];
  __shared__ int stage_n;
  if (threadIdx.x == 0) stage_n = 0;
  __syncthreads();
  const int32_t count = *fr_count;
  for (int32_t i0 = blockIdx.x * kT; i0 < count; i0 += gridDim.x * kT) {
    const int32_t i = i0 + threadIdx.x;
    if (i < count) {
      const int32_t v     = fr[i];
      const uint32_t kv   = key_self[v];
      const uint32_t cand = (((kv >> 24) + 1u) << 24) | (kv & 0xFFFFFFu);
      if ((kv >> 24) < 254u)
        for (int32_t k = off[v]; k < off[v + 1]; ++k) {
          const int32_t u = idx[k];
          if (key_other[u] <= cand) continue;  // a hub (0), an earlier round, or a smaller cell of this round
          if (atomicMin(&key_other[u], cand) == kCellUnvisited) {
            const int p = atomicAdd(&stage_n, 1);
            if (p < kStage) stage[p] = u;
            else out[atomicAdd(out_count, 1)] = u;
          }
        }
    }
    flush_stage(stage, &stage_n, out, out_count);
  }
}

// label of a vertex: its cell's id, or (map != null) what the map makes of it -- the cell's group; -1: hub / not reached
__global__ void __launch_bounds__(kT) k_cell_label(int32_t count, const uint32_t* __restrict__ cell, const int32_t* __restrict__ map,
                                                   int32_t* __restrict__ label)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const uint32_t c = cell[i];
    label[i]         = (c == kCellHub || c == kCellUnvisited) ? -1 : map ? map[c & 0xFFFFFFu] : (int32_t)(c & 0xFFFFFFu);
  }
}
__global__ void __launch_bounds__(kT) k_map_label(int32_t count, const int32_t* __restrict__ label, const int32_t* __restrict__ map,
                                                  int32_t* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const int32_t l = label[i];
    out[i]          = l < 0 ? -1 : map[l];
  }
}

// One sweep of majority voting: a labelled vertex takes the label most of its neighbours carry (the first 32 entries vote; ties: the
// own label, then the smaller one).  The searches of neighbouring cells leak across linking columns -- a tenth of the rows of a
// block-angular LP end up in a foreign block's cell -- and a vertex whose neighbours sit elsewhere almost unanimously is moved there.
__global__ void __launch_bounds__(kT) k_vote(int32_t count, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                             const int32_t* __restrict__ label_self, const int32_t* __restrict__ label_other,
                                             int32_t* __restrict__ label_out)
{
  for (int64_t v = (int64_t)blockIdx.x * kT + threadIdx.x; v < count; v += (int64_t)gridDim.x * kT) {
    const int32_t own = label_self[v];
    int32_t best      = own;
    if (own >= 0) {
      int32_t l[32];
      int nl = 0;
      const int32_t k0 = off[v], k1 = min(off[v + 1], k0 + 32);
      for (int32_t k = k0; k < k1; ++k) {
        const int32_t x = label_other[idx[k]];
        if (x >= 0) l[nl++] = x;
      }
      int best_cnt = 0;
      for (int i = 0; i < nl; ++i) best_cnt += l[i] == own;
      for (int i = 0; i < nl; ++i) {
        int c = 0;
        for (int j = 0; j < nl; ++j) c += l[j] == l[i];
        if (c > best_cnt || (c == best_cnt && best != own && l[i] < best)) best = l[i], best_cnt = c;
      }
    }
    label_out[v] = best;
  }
}

// sort key in group mode: (group of the vertex's cell, the cell's rank, depth) -- 12 + 12 + 8 bits (at most 4096 cells)
__global__ void __launch_bounds__(kT) k_group_key(int32_t count, const uint32_t* __restrict__ cell, const int32_t* __restrict__ label,
                                                  const int32_t* __restrict__ group_of, const int32_t* __restrict__ rank, uint32_t* __restrict__ key)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) {
    const int32_t l = label[i];
    if (l < 0) key[i] = 0xFFFFFFFFu;
    else key[i] = ((uint32_t)group_of[l] << 20) | ((uint32_t)rank[l] << 8) | min(cell[i] >> 24, 255u);
  }
}

// quotient graph of the cells: W[a * K + b] = nonzeros whose row carries label a and whose column carries label b (ids 1 ... K - 1).
// A few thousand pairs carry almost all the weight (the cells of one block, neighbours along a band -- or, on a random matrix, the
// few labels the vote left): every workgroup first adds into a 2048-slot LDS table keyed by the pair, only what collides there and
// the table's final sums go to global atomics (one hot address was 3 ms of serialised atomics on the 1e6 x 1e6 random LP).
constexpr int kQuotSlots = 2048;
__global__ void __launch_bounds__(kT) k_cell_quotient(int32_t rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                      const int32_t* __restrict__ label_r, const int32_t* __restrict__ label_c, int32_t K,
                                                      int32_t* __restrict__ W)
{
  __shared__ int32_t key[kQuotSlots], val[kQuotSlots];
  for (int i = threadIdx.x; i < kQuotSlots; i += kT) key[i] = -1, val[i] = 0;
  __syncthreads();
  auto add = [&](int32_t a, int32_t b, int32_t run) {
    const int32_t pair = a * K + b;  // (K <= 4096: fits)
    const uint32_t h   = ((uint32_t)pair * 2654435761u) >> 21;  // 11 bits
    const int32_t seen = atomicCAS(&key[h], -1, pair);
    if (seen == -1 || seen == pair) atomicAdd(&val[h], run);
    else atomicAdd(&W[(size_t)pair], run);
  };
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < rows; i += (int64_t)gridDim.x * kT) {
    const int32_t a = label_r[i];
    if (a < 0) continue;
    int32_t last_b = -1, run = 0;
    for (int32_t k = off[i]; k < off[i + 1]; ++k) {
      const int32_t b = label_c[idx[k]];
      if (b < 0) continue;
      if (b != last_b) {
        if (run) add(a, last_b, run);
        last_b = b, run = 0;
      }
      ++run;
    }
    if (run) add(a, last_b, run);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kQuotSlots; i += kT)
    if (key[i] >= 0 && val[i]) atomicAdd(&W[(size_t)key[i]], val[i]);
}

__global__ void __launch_bounds__(kT) k_invert_perm(int32_t count, const uint32_t* __restrict__ new2old, int32_t* __restrict__ old2new)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += (int64_t)gridDim.x * kT) old2new[new2old[i]] = (int32_t)i;
}

// ---- sampled row blocks of P M Q for the jagged layout's cost estimate -------------------------------------------------------
// row q of the sample = new row first[q / brows] + q % brows of the permuted matrix (null maps: identity); long rows count 0
__global__ void __launch_bounds__(kT) k_sample_lens(int32_t nq, int32_t brows, const int32_t* __restrict__ first, int32_t rows,
                                                    const uint32_t* __restrict__ row_new2old, const int32_t* __restrict__ off,
                                                    int32_t* __restrict__ lens)
{
  for (int32_t q = blockIdx.x * kT + threadIdx.x; q < nq; q += gridDim.x * kT) {
    const int64_t r_new = (int64_t)first[q / brows] + q % brows;
    int32_t len         = 0;
    if (r_new < rows) {
      const int32_t r = row_new2old ? (int32_t)row_new2old[r_new] : (int32_t)r_new;
      len             = off[r + 1] - off[r];
      if (len > kLongRow) len = 0;
    }
    lens[q] = len;
  }
}

__global__ void __launch_bounds__(kT) k_sample_fill(int32_t nq, int32_t brows, const int32_t* __restrict__ first, int32_t rows,
                                                    const uint32_t* __restrict__ row_new2old, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ idx, const int32_t* __restrict__ col_old2new,
                                                    const int32_t* __restrict__ soff, int32_t* __restrict__ sidx)
{
  for (int32_t q = blockIdx.x * kT + threadIdx.x; q < nq; q += gridDim.x * kT) {
    const int32_t len = soff[q + 1] - soff[q];
    if (len == 0) continue;
    const int64_t r_new = (int64_t)first[q / brows] + q % brows;
    const int32_t r     = row_new2old ? (int32_t)row_new2old[r_new] : (int32_t)r_new;
    const int32_t k0    = off[r];
    for (int32_t k = 0; k < len; ++k) {
      const int32_t c   = idx[k0 + k];
      sidx[soff[q] + k] = col_old2new ? col_old2new[c] : c;
    }
  }
}

// ---- the jagged layout's sampled cost estimate, on the device -------------------------------------------------------------------
// One workgroup per sampled row block restates build_jag's estimate (kernels_jag.hip jag_block_end + jag_block_set) exactly: rows are
// taken in order while the DISTINCT columns they touch fit the LDS window (a row that would overflow ends the block; rows of more than
// kLongRow entries do not count), then the block's column set is priced -- a contiguous range as one coalesced copy, a list as one
// request per run of consecutive columns -- against the gathers it serves.  The set is an open-addressing table in LDS (2 x window
// slots); chunks of 64 rows whose lengths cannot overflow the window whatever they contain are inserted in parallel, the rows near the
// limit one by one with the host's two-step check (count the new columns first).  Runs need no sort: a column starts a run iff its
// predecessor is not in the set.  out[2 * sample] = gathers served, out[2 * sample + 1] = cost.
constexpr int kEstThreads = 1024;

__global__ void __launch_bounds__(kEstThreads) k_jag_estimate(int32_t brows, int32_t wcap, const int32_t* __restrict__ first, int32_t rows,
                                                             const uint32_t* __restrict__ row_new2old, const int32_t* __restrict__ off,
                                                             const int32_t* __restrict__ idx, const int32_t* __restrict__ col_old2new,
                                                             long long* __restrict__ out)
{
  extern __shared__ int32_t tab[];  // 2 * wcap slots
  __shared__ int32_t pre[kEstChunk + 1], rowid[kEstChunk];
  __shared__ int distinct, fresh, stop, lo, hi, runs, parallel;
  __shared__ long long refs;
  const uint32_t mask = (uint32_t)(2 * wcap - 1);
  const int t = threadIdx.x;
  for (int i = t; i < 2 * wcap; i += kEstThreads) tab[i] = kEstEmpty;
  if (t == 0) distinct = 0, stop = 0, lo = 0x7fffffff, hi = -1, runs = 0, refs = 0;
  __syncthreads();
  const int64_t r0 = first[blockIdx.x];
  const int64_t r1 = min((int64_t)rows, r0 + brows);
  auto old_row = [&](int64_t r_new) { return row_new2old ? (int32_t)row_new2old[r_new] : (int32_t)r_new; };
  auto column  = [&](int32_t k) { const int32_t c = idx[k]; return col_old2new ? col_old2new[c] : c; };
  for (int64_t c0 = r0; c0 < r1 && !stop; c0 += kEstChunk) {
    const int nr = (int)min((int64_t)kEstChunk, r1 - c0);
    if (t < nr) {
      const int32_t r = old_row(c0 + t);
      rowid[t]        = r;
      const int32_t len = off[r + 1] - off[r];
      pre[t + 1]        = len > kLongRow ? 0 : len;  // (long rows do not count)
    }
    if (t == 0) pre[0] = 0;
    __syncthreads();
    if (t == 0) {
      for (int i = 0; i < nr; ++i) pre[i + 1] += pre[i];
      parallel = distinct + pre[nr] <= wcap;  // (decided by one thread: `distinct` moves while the chunk is inserted)
    }
    __syncthreads();
    const int total = pre[nr];
    if (parallel) {
      // no row of the chunk can overflow the window: all entries at once
      int added = 0;
      for (int e = t; e < total; e += kEstThreads) {
        int a = 0, b = nr;  // the row of entry e: largest i with pre[i] <= e
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (pre[mid] <= e) a = mid; else b = mid;
        }
        const int32_t c = column(off[rowid[a]] + (e - pre[a]));
        atomicMin(&lo, c), atomicMax(&hi, c);
        added += est_insert(tab, mask, c);
      }
      if (added) atomicAdd(&distinct, added);
      if (t == 0) refs += total;
      __syncthreads();
    } else {
      // near the limit: row by row, the host's rule (count the new columns before inserting any)
      for (int i = 0; i < nr; ++i) {
        const int len = pre[i + 1] - pre[i];
        if (len == 0) continue;  // (uniform: pre is shared)
        const int32_t c = t < len ? column(off[rowid[i]] + t) : 0;
        if (t == 0) fresh = 0;
        __syncthreads();
        const bool check = distinct + len > wcap;  // (`distinct` rests between the barriers around the inserts)
        if (check && t < len && !est_contains(tab, mask, c)) atomicAdd(&fresh, 1);
        __syncthreads();
        if (t == 0 && check && distinct + fresh > wcap) stop = 1;
        __syncthreads();
        if (stop) break;
        if (t < len) {
          atomicMin(&lo, c), atomicMax(&hi, c);
          if (est_insert(tab, mask, c)) atomicAdd(&distinct, 1);
        }
        if (t == 0) refs += len;
        __syncthreads();
      }
    }
    __syncthreads();
  }
  __syncthreads();
  long long cost = 0;
  if (hi >= 0) {
    if ((long long)hi - lo + 1 <= wcap) {
      cost = 1 + (hi - lo + 1) / 16;
    } else {
      int mine = 0;
      for (int i = t; i < 2 * wcap; i += kEstThreads) {
        const int32_t c = tab[i];
        if (c != kEstEmpty && (c == 0 || !est_contains(tab, mask, c - 1))) ++mine;
      }
      if (mine) atomicAdd(&runs, mine);
      __syncthreads();
      cost = (long long)runs + distinct / 16;
    }
  }
  if (t == 0) out[2 * blockIdx.x] = refs, out[2 * blockIdx.x + 1] = cost;
}

// ---- the permuted CSR pair ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kT) k_new_labels(int64_t nnz, const uint32_t* __restrict__ rowid, const int32_t* __restrict__ idx,
                                                   const int32_t* __restrict__ row_old2new, const int32_t* __restrict__ col_old2new,
                                                   uint32_t* __restrict__ newrow, uint32_t* __restrict__ newcol)
{
  for (int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * kT) {
    newrow[k] = (uint32_t)row_old2new[rowid[k]];
    newcol[k] = (uint32_t)col_old2new[idx[k]];
  }
}

__global__ void __launch_bounds__(kT) k_gather_f64(int64_t n, const double* __restrict__ src, const uint32_t* __restrict__ at, double* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) out[i] = src[at[i]];
}
__global__ void __launch_bounds__(kT) k_gather_u32(int64_t n, const uint32_t* __restrict__ src, const uint32_t* __restrict__ at, uint32_t* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) out[i] = src[at[i]];
}

__global__ void __launch_bounds__(kT) k_emit_csr(int64_t n, const uint32_t* __restrict__ src_pos, const uint32_t* __restrict__ label,
                                                 const double* __restrict__ val, int32_t* __restrict__ out_idx, double* __restrict__ out_val)
{
  for (int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x; q < n; q += (int64_t)gridDim.x * kT) {
    const uint32_t k = src_pos[q];
    out_idx[q] = (int32_t)label[k], out_val[q] = val[k];
  }
}

}  // namespace

namespace {
// the analysis' phase timer (CUOPT_AMD_TIMING prints the laps)
struct Lap {
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  std::string* out;
  hipStream_t s;
  explicit Lap(std::string* o, hipStream_t st) : out(o), s(st) {}
  double operator()(const char* what, bool sync = true)
  {
    if (sync) (void)hipStreamSynchronize(s);
    const auto now  = std::chrono::steady_clock::now();
    const double ms = 1e3 * std::chrono::duration<double>(now - last).count();
    last            = now;
    char buf[96];
    snprintf(buf, sizeof(buf), "%s %.2f ms; ", what, ms);
    *out += buf;
    return ms;
  }
};
}  // namespace

// ================================================================================================
// host-side structure mirrors
// ================================================================================================
const int32_t* analysis_host_off(pdlpdev_analysis* an)
{
  if (!an->permuted) return an->h_off;
  if (!an->have_hp_off) {  // (the offsets alone: 4 MB at 1e6 rows; the indices -- 40 MB -- only when a host construction asks for them)
    an->hp_off.reset((size_t)an->m + 1);
    (void)hipMemcpyAsync(an->hp_off.get(), an->A.off, ((size_t)an->m + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream);
    (void)hipStreamSynchronize(an->stream);
    an->have_hp_off = true;
  }
  return an->hp_off.get();
}
const int32_t* analysis_host_idx(pdlpdev_analysis* an)
{
  if (!an->permuted) return an->h_idx;
  if (!an->have_hp) {
    an->hp_idx.reset((size_t)std::max<int64_t>(an->nnz, 1));
    (void)hipMemcpyAsync(an->hp_idx.get(), an->A.idx, (size_t)an->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream);
    (void)hipStreamSynchronize(an->stream);
    an->have_hp = true;
  }
  return an->hp_idx.get();
}
// the indices of a few rows only (the dense-segment scan looks at rows of 256+ entries): the other entries of the returned array are
// unspecified.  A later analysis_host_idx fetches everything.
const int32_t* analysis_host_idx_rows(pdlpdev_analysis* an, const int32_t* h_off, const std::vector<int32_t>& rows)
{
  if (!an->permuted) return an->h_idx;
  if (an->have_hp) return an->hp_idx.get();
  an->hp_idx.reset((size_t)std::max<int64_t>(an->nnz, 1));
  for (int32_t r : rows)
    (void)hipMemcpyAsync(an->hp_idx.get() + h_off[r], an->A.idx + h_off[r], (size_t)(h_off[r + 1] - h_off[r]) * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream);
  (void)hipStreamSynchronize(an->stream);
  return an->hp_idx.get();
}
const int32_t* analysis_host_t_off(pdlpdev_analysis* an)
{
  if (!an->have_hpt_off) {
    an->hpt_off.reset((size_t)an->n + 1);
    (void)hipMemcpyAsync(an->hpt_off.get(), an->At.off, ((size_t)an->n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream);
    (void)hipStreamSynchronize(an->stream);
    an->have_hpt_off = true;
  }
  return an->hpt_off.get();
}
const int32_t* analysis_host_t_idx(pdlpdev_analysis* an)
{
  if (!an->have_hpt_idx) {
    an->hpt_idx.reset((size_t)std::max<int64_t>(an->nnz, 1));
    (void)hipMemcpyAsync(an->hpt_idx.get(), an->At.idx, (size_t)an->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream);
    (void)hipStreamSynchronize(an->stream);
    an->have_hpt_idx = true;
  }
  return an->hpt_idx.get();
}

// ================================================================================================
// the ordering search
// ================================================================================================
namespace {

struct Side {  // one side of the bipartite graph: its vertices' adjacency
  int32_t count;
  const int32_t *off, *idx;
};

// jagged-layout estimate of P M Q for M = A (side 0) or A^T (side 1) under candidate maps (null: identity); < 0: not applicable
int estimate_saving(pdlpdev_analysis* an, int side, const uint32_t* d_row_new2old, const int32_t* d_col_old2new, double* saving)
{
  const int32_t rows = side == 0 ? an->m : an->n;
  const DevCsr& M    = side == 0 ? an->A : an->At;
  int G = 0, waves = 8, wcap = 0, brows = 0;
  *saving = 0.0;
  if (!jag_geometry(rows, 0, &G, &waves, &wcap, &brows)) return 0;
  const int samples = (int)std::min<int64_t>(48, std::max<int64_t>(1, rows / brows));
  std::vector<int32_t> first(samples);
  for (int t = 0; t < samples; ++t) first[t] = (int32_t)((int64_t)rows * t / samples);
  const int32_t nq = samples * brows;
  DevArena& ar     = an->arena;
  const size_t mark = ar.mark();
  if (cuopt_amd::tune_int("estimate_host", 0) == 0) {
    // on the device: one workgroup per sampled block (k_jag_estimate); 2 x 48 numbers come back
    int32_t* d_first  = ar.take<int32_t>(samples);
    long long* d_out  = ar.take<long long>((size_t)2 * samples);
    if (!d_first || !d_out) return fail(-2, "device set-up: workspace too small for the layout estimate");
    hipStream_t s = an->stream;
    const size_t lds = (size_t)2 * wcap * sizeof(int32_t);
    static std::mutex mu;
    static std::vector<int> done;
    {
      std::lock_guard<std::mutex> lock(mu);
      if (std::find(done.begin(), done.end(), an->device) == done.end()) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_jag_estimate, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16384 * 4));
        done.push_back(an->device);
      }
    }
    HIP_TRY(hipMemcpyAsync(d_first, first.data(), samples * sizeof(int32_t), hipMemcpyHostToDevice, s));
    k_jag_estimate<<<samples, kEstThreads, lds, s>>>(brows, wcap, d_first, rows, d_row_new2old, M.off, M.idx, d_col_old2new, d_out);
    std::vector<long long> h((size_t)2 * samples);
    HIP_TRY(hipMemcpyAsync(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    ar.release(mark);
    long long r = 0, c = 0;
    for (int t = 0; t < samples; ++t) r += h[2 * t], c += h[2 * t + 1];
    *saving = r ? 1.0 - (double)c / (double)r : 0.0;
    return 0;
  }
  int32_t* d_first = ar.take<int32_t>(samples);
  int32_t* d_lens  = ar.take<int32_t>((size_t)nq + 1);
  int32_t* d_soff  = ar.take<int32_t>((size_t)nq + 1);
  int32_t* d_bs    = ar.take<int32_t>((size_t)nq / kScanTile + 2);
  int32_t* d_sidx  = ar.take<int32_t>((size_t)nq * kLongRow);
  if (!d_first || !d_lens || !d_soff || !d_bs || !d_sidx) return fail(-2, "device set-up: workspace too small for the layout estimate");
  hipStream_t s = an->stream;
  HIP_TRY(hipMemcpyAsync(d_first, first.data(), samples * sizeof(int32_t), hipMemcpyHostToDevice, s));
  k_sample_lens<<<grid_of(nq), kT, 0, s>>>(nq, brows, d_first, rows, d_row_new2old, M.off, d_lens);
  TRY(dev_exclusive_scan(s, d_lens, d_soff, nq, d_bs));
  k_sample_fill<<<grid_of(nq), kT, 0, s>>>(nq, brows, d_first, rows, d_row_new2old, M.off, M.idx, d_col_old2new, d_soff, d_sidx);
  std::vector<int32_t> soff((size_t)nq + 1);
  HIP_TRY(hipMemcpyAsync(soff.data(), d_soff, ((size_t)nq + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<int32_t> sidx((size_t)std::max(soff[nq], 1));
  HIP_TRY(hipMemcpyAsync(sidx.data(), d_sidx, (size_t)soff[nq] * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  ar.release(mark);
  if (d_col_old2new)  // a permuted row's columns arrive in the old order
    cuopt_amd::parallel_tasks(samples, [&](int t) {
      for (int32_t q = t * brows; q < (t + 1) * brows; ++q) std::sort(sidx.begin() + soff[q], sidx.begin() + soff[q + 1]);
    }, soff[nq]);
  *saving = jag_estimate_on_samples(samples, brows, wcap, soff.data(), sidx.data());
  return 0;
}

int keys_to_perm(pdlpdev_analysis* an, SortBufs& SB, int32_t count, const uint32_t* d_keys, int bits, uint32_t* d_new2old, int32_t* d_old2new)
{
  int slot = 0;
  TRY(radix_sort_pairs(an->stream, count, d_keys, nullptr, SB, bits, &slot));
  HIP_TRY(hipMemcpyAsync(d_new2old, SB.v[slot], (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToDevice, an->stream));
  k_invert_perm<<<grid_of(count), kT, 0, an->stream>>>(count, d_new2old, d_old2new);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// Looks for a row / column order under which the jagged layout applies.  On success (an->method != 0) the device holds the maps
// {row new2old, col new2old, row old2new, col old2new} (arena memory, valid until the caller releases its mark).
//
// One search serves band-like and block-like structure.  CELLS are grown breadth-first from seeds spaced one row block apart on the
// bipartite row-column graph (rows / columns of more than kLongRow entries neither expand nor get claimed): a handful of rounds
// whatever the graph's diameter (a full level-synchronous search of a band of half-width 2000 over a million rows is ~1100 dependent
// levels: 17-60 ms as launches, no better as one resident workgroup -- measured, profiles/r05_setup.txt).  The QUOTIENT graph of
// the cells (a few hundred nodes) comes to the host: cells joined by strong edges form groups, and the groups' shape decides --
//  * long chains (a band, a staircase: the cells line up along the diagonal): the position of a cell along its chain (breadth-first
//    levels from a pseudo-peripheral cell) is the start value of a few BARYCENTRE sweeps over the real graph (a vertex moves to the
//    mean position of its neighbours): the fuzzy cell boundaries dissolve into a smooth one-dimensional embedding -- method 1;
//  * small tight groups (diagonal blocks behind linking rows / columns): two sweeps of MAJORITY voting at group level pull back the
//    tenth of the vertices that a neighbouring block's cell reached first through a linking column -- method 2.
// The candidate is accepted only if the jagged layout's own sampled cost estimate passes for P A Q and for its transpose.
static int find_ordering(pdlpdev_analysis* an, uint32_t** d_row_new2old, uint32_t** d_col_new2old, int32_t** d_row_old2new,
                         int32_t** d_col_old2new, Lap& lap)
{
  const int32_t m = an->m, n = an->n;
  DevArena& ar = an->arena;
  hipStream_t s = an->stream;
  an->method = 0;
  int G = 0, waves = 8, wcap = 0, brows = 0;
  if (!jag_geometry(m, 0, &G, &waves, &wcap, &brows)) return 0;
  const int32_t spacing = (int32_t)std::max<long long>(64, cuopt_amd::tune_int("reorder_cell_rows", brows));
  const int32_t ncells  = (int32_t)(((int64_t)m + spacing - 1) / spacing);
  const int32_t K       = ncells + 1;
  if (K > 4096) return 0;  // (beyond 8 M rows the quotient graph would need a sparse form: not built)
  const int rounds_cap = 48;
  int32_t* fr_r   = ar.take<int32_t>(m);
  int32_t* fr_c   = ar.take<int32_t>(n);
  int32_t* counts = ar.take<int32_t>((size_t)rounds_cap + 2);
  uint32_t* row_n2o = ar.take<uint32_t>(m);
  uint32_t* col_n2o = ar.take<uint32_t>(n);
  int32_t* row_o2n  = ar.take<int32_t>(m);
  int32_t* col_o2n  = ar.take<int32_t>(n);
  uint32_t* cell_r  = ar.take<uint32_t>(m);
  uint32_t* cell_c  = ar.take<uint32_t>(n);
  uint32_t* key_r   = ar.take<uint32_t>(m);
  uint32_t* key_c   = ar.take<uint32_t>(n);
  int32_t* lab_r    = ar.take<int32_t>(m);
  int32_t* lab_c    = ar.take<int32_t>(n);
  int32_t* W        = ar.take<int32_t>((size_t)K * K);
  int32_t* d_rank   = ar.take<int32_t>((size_t)K);
  int32_t* d_group  = ar.take<int32_t>((size_t)K);
  float* d_coord    = ar.take<float>((size_t)K);
  SortBufs SB;
  TRY(sort_bufs_take(ar, &SB, std::max(m, n)));
  if (!fr_r || !fr_c || !counts || !row_n2o || !col_n2o || !row_o2n || !col_o2n || !cell_r || !cell_c || !key_r || !key_c || !lab_r || !lab_c || !W || !d_rank ||
      !d_group || !d_coord)
    return fail(-2, "device set-up: workspace too small for the ordering search");
  *d_row_new2old = row_n2o, *d_col_new2old = col_n2o, *d_row_old2new = row_o2n, *d_col_old2new = col_o2n;
  const double accept = 0.5;  // the full construction's own threshold (build_jag)

  // ---- cells
  HIP_TRY(hipMemsetAsync(counts, 0, ((size_t)rounds_cap + 2) * sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(W, 0, (size_t)K * K * sizeof(int32_t), s));
  k_cell_init<<<grid_of(m), kT, 0, s>>>(m, an->A.off, kLongRow, cell_r);
  k_cell_init<<<grid_of(n), kT, 0, s>>>(n, an->At.off, kLongRow, cell_c);
  k_cell_seed<<<(ncells + kT - 1) / kT, kT, 0, s>>>(m, spacing, ncells, an->A.off, cell_r, fr_r, counts);
  int level = 0;
  std::vector<int32_t> csz;
  for (;;) {
    const int upto = std::min(level + 8, rounds_cap);
    for (; level < upto; ++level) {
      if ((level & 1) == 0)
        k_cell_expand<<<512, kT, 0, s>>>(fr_r, counts + level, an->A.off, an->A.idx, cell_r, cell_c, fr_c, counts + level + 1);
      else
        k_cell_expand<<<512, kT, 0, s>>>(fr_c, counts + level, an->At.off, an->At.idx, cell_c, cell_r, fr_r, counts + level + 1);
    }
    csz.resize((size_t)level + 1);
    HIP_TRY(hipMemcpyAsync(csz.data(), counts, ((size_t)level + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (csz.back() == 0 || level >= rounds_cap) break;
  }
  an->cell_rounds = level;
  {
    // The cells as the searches left them: if no two of them belong together already (no pair shares 5 % of the lighter one's
    // nonzeros), there is no structure for the vote to sharpen -- the uniformly random matrix leaves here.  (After the vote a random
    // graph WOULD show strong edges: label propagation manufactures communities.)
    k_cell_label<<<grid_of(m), kT, 0, s>>>(m, cell_r, nullptr, lab_r);
    k_cell_label<<<grid_of(n), kT, 0, s>>>(n, cell_c, nullptr, lab_c);
    k_cell_quotient<<<grid_of(m), kT, 0, s>>>(m, an->A.off, an->A.idx, lab_r, lab_c, K, W);
    std::vector<int32_t> hw0((size_t)K * K);
    HIP_TRY(hipMemcpyAsync(hw0.data(), W, hw0.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(W, 0, (size_t)K * K * sizeof(int32_t), s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<int64_t> tot0(K, 0);
    auto sym0 = [&](int a, int b) { return (int64_t)hw0[(size_t)a * K + b] + (a == b ? 0 : hw0[(size_t)b * K + a]); };
    for (int a = 1; a < K; ++a)
      for (int b = 1; b < K; ++b) tot0[a] += a == b ? hw0[(size_t)a * K + a] : sym0(a, b);
    bool any = false;
    for (int a = 1; a < K && !any; ++a)
      for (int b = a + 1; b < K && !any; ++b) {
        const int64_t w = sym0(a, b);
        any = w > 0 && w * 20 >= std::min(tot0[a], tot0[b]);
      }
    if (!any) {
      lap("cells");
      return 0;
    }
  }
  // Two sweeps of majority voting at CELL level before the cells are compared: a search that reached a neighbouring block through a
  // linking column first holds a foothold there (a tenth of the rows of a block-angular LP sit in a foreign block's cell, and a cell
  // with footholds in many blocks would glue their groups together); a vertex whose neighbours carry another label almost
  // unanimously takes it.  On a band the same sweeps sharpen the cells' fuzzy borders.
  {
    int32_t* lab_r2 = (int32_t*)fr_r;  // (the frontiers are free now)
    int32_t* lab_c2 = (int32_t*)fr_c;
    int32_t *lr = lab_r, *lc = lab_c;
    k_cell_label<<<grid_of(m), kT, 0, s>>>(m, cell_r, nullptr, lr);
    k_cell_label<<<grid_of(n), kT, 0, s>>>(n, cell_c, nullptr, lc);
    const int votes = (int)cuopt_amd::tune_int("reorder_votes", 2);
    for (int it = 0; it < votes; ++it) {
      k_vote<<<grid_of(n), kT, 0, s>>>(n, an->At.off, an->At.idx, lc, lr, lab_c2);
      k_vote<<<grid_of(m), kT, 0, s>>>(m, an->A.off, an->A.idx, lr, lab_c2, lab_r2);
      std::swap(lr, lab_r2), std::swap(lc, lab_c2);
    }
    if (lr != lab_r) {  // (an odd number of sweeps: the labels end in the spare buffers)
      HIP_TRY(hipMemcpyAsync(lab_r, lr, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemcpyAsync(lab_c, lc, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
  }
  k_cell_quotient<<<grid_of(m), kT, 0, s>>>(m, an->A.off, an->A.idx, lab_r, lab_c, K, W);
  std::vector<int32_t> hw((size_t)K * K);
  HIP_TRY(hipMemcpyAsync(hw.data(), W, hw.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  lap("cells");

  // ---- the quotient graph on the host: strong edges (>= 5 % of the lighter cell's nonzeros), groups, a chain coordinate per group
  auto sym = [&](int a, int b) { return (int64_t)hw[(size_t)a * K + b] + (a == b ? 0 : hw[(size_t)b * K + a]); };
  std::vector<int64_t> tot(K, 0);
  for (int a = 1; a < K; ++a)
    for (int b = 1; b < K; ++b) tot[a] += a == b ? hw[(size_t)a * K + a] : sym(a, b);
  std::vector<std::vector<std::pair<int64_t, int>>> strong(K);  // (-weight, neighbour): sorted = the heaviest first, ties to the smaller id
  int64_t strong_edges = 0;
  for (int a = 1; a < K; ++a) {
    if (tot[a] == 0) continue;
    for (int b = 1; b < K; ++b) {
      if (b == a || tot[b] == 0) continue;
      const int64_t w = sym(a, b);
      if (w > 0 && w * 20 >= std::min(tot[a], tot[b])) strong[a].emplace_back(-w, b);
    }
    std::sort(strong[a].begin(), strong[a].end());
    strong_edges += (int64_t)strong[a].size();
  }
  if (strong_edges == 0) return 0;  // no two cells belong together: a matrix without structure to find (the uniformly random case)
  auto levels_from = [&](int a0, std::vector<int32_t>& lev, std::vector<int32_t>& members) {  // breadth-first over the strong edges
    members.assign(1, a0);
    lev[a0] = 0;
    for (size_t head = 0; head < members.size(); ++head) {
      const int a = members[head];
      for (auto& e : strong[a])
        if (lev[e.second] < 0) lev[e.second] = lev[a] + 1, members.push_back(e.second);
    }
  };
  std::vector<int32_t> rank(K, 0), group_of(K, 0), lev(K, -1), members, order;
  std::vector<float> coord(K, 0.0f);
  order.reserve(K);
  int ngroups = 0;
  int64_t cells_in_chains = 0, cells_total = 0;
  float base = 0.0f;
  for (int a0 = 1; a0 < K; ++a0) {
    if (lev[a0] >= 0 || tot[a0] == 0) continue;
    levels_from(a0, lev, members);
    // again from the deepest cell (the smallest id among them): the levels run along the chain from one of its ends
    int far = a0;
    for (int c : members)
      if (lev[c] > lev[far] || (lev[c] == lev[far] && c < far)) far = c;
    for (int c : members) lev[c] = -1;
    levels_from(far, lev, members);
    int depth = 0;
    for (int c : members) depth = std::max(depth, (int)lev[c]);
    const int g = ngroups++;
    for (int c : members) group_of[c] = g, coord[c] = base + (float)lev[c], order.push_back(c);
    base += (float)depth + 2.0f;
    cells_total += (int64_t)members.size();
    if (depth >= 8) cells_in_chains += (int64_t)members.size();
  }
  for (size_t i = 0; i < order.size(); ++i) rank[order[i]] = (int32_t)i;
  HIP_TRY(hipMemcpyAsync(d_rank, rank.data(), (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_group, group_of.data(), (size_t)K * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_coord, coord.data(), (size_t)K * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  an->bfs_levels = (int)base;
  const bool chains = cells_in_chains * 2 >= cells_total;
  lap("quotient");

  for (int attempt = 0; attempt < 2 && an->method == 0; ++attempt) {
    const bool chain_mode = attempt == 0 ? chains : !chains;
    if (attempt == 1 && !(chain_mode ? cells_in_chains > 0 : true)) break;
    double* saving = chain_mode ? an->saving_levels : an->saving_cells;
    if (chain_mode) {
      // positions along the chains, then barycentre sweeps each way
      float* pos_r  = (float*)key_r;
      float* pos_c  = (float*)key_c;
      float* pos_r2 = (float*)fr_r;  // (the frontiers are free now)
      float* pos_c2 = (float*)fr_c;
      k_cell_pos<<<grid_of(m), kT, 0, s>>>(m, lab_r, d_coord, pos_r);
      k_cell_pos<<<grid_of(n), kT, 0, s>>>(n, lab_c, d_coord, pos_c);
      const int sweeps = (int)cuopt_amd::tune_int("reorder_sweeps", 3);
      for (int it = 0; it < sweeps; ++it) {
        k_barycentre<<<grid_of(n), kT, 0, s>>>(n, an->At.off, an->At.idx, pos_c, pos_r, pos_c2);
        k_barycentre<<<grid_of(m), kT, 0, s>>>(m, an->A.off, an->A.idx, pos_r, pos_c2, pos_r2);
        std::swap(pos_r, pos_r2), std::swap(pos_c, pos_c2);
      }
      const float scale       = 256.0f;
      const uint32_t last_key = (uint32_t)((base + 2.0f) * scale);
      uint32_t* kr            = (uint32_t*)pos_r2;
      uint32_t* kc            = (uint32_t*)pos_c2;
      k_pos_to_key<<<grid_of(m), kT, 0, s>>>(m, pos_r, scale, last_key, kr);
      k_pos_to_key<<<grid_of(n), kT, 0, s>>>(n, pos_c, scale, last_key, kc);
      TRY(keys_to_perm(an, SB, m, kr, bits_for(last_key), row_n2o, row_o2n));
      TRY(keys_to_perm(an, SB, n, kc, bits_for(last_key), col_n2o, col_o2n));
      lap("chain -> order");
    } else {
      // keys (group of the vertex's cell, the cell's rank, depth): the cells of a group are neighbours, the groups follow each other
      uint32_t* kr = (uint32_t*)fr_r;
      uint32_t* kc = (uint32_t*)fr_c;
      k_group_key<<<grid_of(m), kT, 0, s>>>(m, cell_r, lab_r, d_group, d_rank, kr);
      k_group_key<<<grid_of(n), kT, 0, s>>>(n, cell_c, lab_c, d_group, d_rank, kc);
      TRY(keys_to_perm(an, SB, m, kr, 32, row_n2o, row_o2n));
      TRY(keys_to_perm(an, SB, n, kc, 32, col_n2o, col_o2n));
      lap("groups -> order");
    }
    TRY(estimate_saving(an, 0, row_n2o, col_o2n, &saving[0]));
    if (saving[0] >= accept) TRY(estimate_saving(an, 1, col_n2o, row_o2n, &saving[1]));
    lap("estimate");
    if (saving[0] >= accept && saving[1] >= accept) an->method = chain_mode ? 1 : 2;
  }
  return 0;
}

// P A Q and its transpose on the device from A and the maps; replaces an->A / an->At
static int build_permuted_pair(pdlpdev_analysis* an, const int32_t* d_row_o2n, const int32_t* d_col_o2n)
{
  const int32_t m = an->m, n = an->n;
  const int64_t nnz = an->nnz;
  DevArena& ar = an->arena;
  hipStream_t s = an->stream;
  const size_t mark = ar.mark();
  uint32_t* newrow = ar.take<uint32_t>((size_t)nnz);
  uint32_t* newcol = ar.take<uint32_t>((size_t)nnz);
  uint32_t* gkeys  = ar.take<uint32_t>((size_t)nnz);
  SortBufs SB;
  TRY(sort_bufs_take(ar, &SB, nnz));
  if (!newrow || !newcol || !gkeys) return fail(-2, "device set-up: workspace too small for the permuted matrices");
  // the new matrices (the old A^T is dropped first: its memory is reused through the allocator)
  DevCsr NA, NT;
  HIP_TRY(hipMalloc((void**)&NA.off, ((size_t)m + 1) * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&NA.idx, ((size_t)nnz + 8) * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&NA.val, ((size_t)nnz + 8) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&NT.off, ((size_t)n + 1) * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&NT.idx, ((size_t)nnz + 8) * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void**)&NT.val, ((size_t)nnz + 8) * sizeof(double)));
  HIP_TRY(hipMemsetAsync(NA.idx + nnz, 0, 8 * sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(NA.val + nnz, 0, 8 * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(NT.idx + nnz, 0, 8 * sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(NT.val + nnz, 0, 8 * sizeof(double), s));
  k_row_ids<<<(int)((nnz + 4095) / 4096), kT, 0, s>>>(nnz, an->A.off, m, gkeys);  // (gkeys: free until the first gather)
  k_new_labels<<<grid_of(nnz), kT, 0, s>>>(nnz, gkeys, an->A.idx, d_row_o2n, d_col_o2n, newrow, newcol);
  const int bits_r = bits_for((uint32_t)std::max(m - 1, 1)), bits_c = bits_for((uint32_t)std::max(n - 1, 1));
  int slot = 0;
  // by new column (source order inside), then stable by new row: (new row, new column) order = P A Q
  TRY(radix_sort_pairs(s, nnz, newcol, nullptr, SB, bits_c, &slot));
  k_gather_u32<<<grid_of(nnz), kT, 0, s>>>(nnz, newrow, SB.v[slot], gkeys);
  // (the values of one sort are the input values of the next; its first pass writes to slot 0, so they move to a buffer of their own)
  uint32_t* carry = ar.take<uint32_t>((size_t)nnz);
  if (!carry) return fail(-2, "device set-up: workspace too small for the permuted matrices");
  HIP_TRY(hipMemcpyAsync(carry, SB.v[slot], (size_t)nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  TRY(radix_sort_pairs(s, nnz, gkeys, carry, SB, bits_r, &slot));
  k_offsets_from_sorted<<<grid_of(nnz + 1), kT, 0, s>>>(SB.k[slot], nnz, m, NA.off);
  k_emit_csr<<<grid_of(nnz), kT, 0, s>>>(nnz, SB.v[slot], newcol, an->A.val, NA.idx, NA.val);
  // ... then stable by new column again: (new column, new row) order = its transpose
  HIP_TRY(hipMemcpyAsync(carry, SB.v[slot], (size_t)nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
  k_gather_u32<<<grid_of(nnz), kT, 0, s>>>(nnz, newcol, carry, gkeys);
  TRY(radix_sort_pairs(s, nnz, gkeys, carry, SB, bits_c, &slot));
  k_offsets_from_sorted<<<grid_of(nnz + 1), kT, 0, s>>>(SB.k[slot], nnz, n, NT.off);
  k_emit_csr<<<grid_of(nnz), kT, 0, s>>>(nnz, SB.v[slot], newrow, an->A.val, NT.idx, NT.val);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  // swap in.  (The matrices in the caller's order stay in `owned`: six hipFree calls here are six device synchronisations in the
  // middle of the set-up; they go with the object -- up to 5e7 nonzeros at the solver's end, pdlp_solver.cpp -- unless they are large)
  if (nnz > 50000000)
    for (void* p : {(void*)an->A.off, (void*)an->A.idx, (void*)an->A.val, (void*)an->At.off, (void*)an->At.idx, (void*)an->At.val}) {
      an->owned.erase(std::remove(an->owned.begin(), an->owned.end(), p), an->owned.end());
      (void)hipFree(p);
    }
  an->A = NA, an->At = NT;
  for (void* p : {(void*)NA.off, (void*)NA.idx, (void*)NA.val, (void*)NT.off, (void*)NT.idx, (void*)NT.val}) an->owned.push_back(p);
  an->have_hp = an->have_hp_off = an->have_hpt_off = an->have_hpt_idx = false;
  ar.release(mark);
  return 0;
}

// The analysis' workspace (a few hundred MB at 1e7 nonzeros) outlives the analysis in a one-slot cache per process: releasing it is a
// hipFree -- a device synchronisation, ~3 ms of a 25 ms set-up -- and the next analysis of a similar LP would ask for the same block
// again.  Blocks beyond 2 GB are returned at once.
namespace {
struct SpareArena {
  int device = -1;
  char* base = nullptr;
  size_t cap = 0;
};
std::mutex g_spare_mutex;
SpareArena g_spare;
constexpr size_t kSpareArenaMax = (size_t)2 << 30;
char* take_spare_arena(int device, size_t bytes, size_t* cap)
{
  std::lock_guard<std::mutex> lock(g_spare_mutex);
  if (g_spare.base && g_spare.device == device && g_spare.cap >= bytes && g_spare.cap <= 4 * bytes + ((size_t)64 << 20)) {
    char* p = g_spare.base;
    *cap    = g_spare.cap;
    g_spare = SpareArena{};
    return p;
  }
  return nullptr;
}
// true: kept (the caller must not free it)
bool give_spare_arena(int device, char* base, size_t cap)
{
  if (!base || cap > kSpareArenaMax) return false;
  char* old = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_spare_mutex);
    if (g_spare.base && g_spare.cap >= cap) return false;  // (the larger one stays)
    old     = g_spare.base;
    g_spare = SpareArena{device, base, cap};
  }
  if (old) (void)hipFree(old);
  return true;
}
}  // namespace

extern "C" {

// The workspace leaves a consumed analysis at once (to the one-slot cache, or back to the runtime when it is too large to keep): what
// is left of the object -- small buffers -- may then wait for a convenient moment (the host driver: the solver's end).
void pdlpdev_analysis_release_workspace(pdlpdev_analysis* an)
{
  if (!an || !an->arena.base) return;
  (void)hipSetDevice(an->device);
  if (an->stream) (void)hipStreamSynchronize(an->stream);
  an->owned.erase(std::remove(an->owned.begin(), an->owned.end(), (void*)an->arena.base), an->owned.end());
  if (!give_spare_arena(an->device, an->arena.base, an->arena.cap)) (void)hipFree(an->arena.base);
  an->arena = DevArena{};
}

void pdlpdev_analysis_destroy(pdlpdev_analysis* an)
{
  if (!an) return;
  (void)hipSetDevice(an->device);
  if (an->stream) (void)hipStreamSynchronize(an->stream);
  pdlpdev_analysis_release_workspace(an);
  for (void* p : an->owned) (void)hipFree(p);
  for (double* p : an->pref_dev)
    if (p) (void)hipFree(p);
  if (an->bundle_owned && an->stream) {
    // the stream goes back to the pool the contexts draw from (an HSA queue costs ~2 ms to create)
    if (!(an->pinned && an->chunk && give_recycled(Recycled{an->device, an->stream, an->pinned, an->chunk}))) {
      if (an->pinned) (void)hipHostFree(an->pinned);
      if (an->chunk) (void)hipFree(an->chunk);
      (void)hipStreamDestroy(an->stream);
    }
  }
  delete an;
}

// Uploads A (host CSR, m x n), builds A^T on the device and -- flags bit 0 -- looks for a row / column order under which the jagged
// layout applies (accepted only when the layout's own cost estimate passes for BOTH matrices; then A and A^T on the device are the
// permuted pair).  The host arrays must stay valid until the analysis is consumed (pdlpdev_create_from_analysis) or destroyed.
int pdlpdev_analyze(pdlpdev_analysis** out, int device, int32_t m, int32_t n, const int32_t* a_off, const int32_t* a_idx,
                    const double* a_val, int flags)
{
  return pdlpdev_analyze_with_vectors(out, device, m, n, a_off, a_idx, a_val, flags, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// ... and, when given, sends the problem vectors (c, lb, ub: n; lo, hi: m -- in the caller's order) to the device on a helper thread
// while the analysis' kernels run; pdlpdev_create_from_analysis picks them up when it is handed the same host pointers and the
// analysis did not permute the matrix.
int pdlpdev_analyze_with_vectors(pdlpdev_analysis** out, int device, int32_t m, int32_t n, const int32_t* a_off, const int32_t* a_idx,
                                 const double* a_val, int flags, const double* c, const double* lo, const double* hi, const double* lb,
                                 const double* ub)
{
  roctx::Range range("pdlp: device analysis (upload, transpose, ordering)");
  if (!out || m < 0 || n < 0 || !a_off) return fail(-1, "pdlpdev_analyze: bad argument");
  if (pdlpdev_device_count() <= device)
    return fail(-5, "pdlpdev_analyze: no HIP device %d visible (this solver has no CPU fallback)", device);
  HIP_TRY(hipSetDevice(device));
  pdlpdev_analysis* an = new pdlpdev_analysis();
  *out       = an;
  an->device = device, an->m = m, an->n = n, an->nnz = a_off[m];
  an->h_off = a_off, an->h_idx = a_idx, an->h_val = a_val;
  {
    Recycled r;
    if (take_recycled(device, &r)) {
      an->stream = r.stream, an->pinned = r.pinned, an->chunk = r.chunk;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&an->stream, hipStreamNonBlocking));
      HIP_TRY(hipHostMalloc((void**)&an->pinned, kScalars * sizeof(double) + sizeof(pdlpdev_ctl)));
      HIP_TRY(hipMalloc((void**)&an->chunk, kArenaChunk));
    }
  }
  hipStream_t s     = an->stream;
  const int64_t nnz = an->nnz;
  Lap lap(&an->laps, s);
  auto dmalloc = [&](void** p, size_t bytes) -> int {
    HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 256)));
    an->owned.push_back(*p);
    return 0;
  };
  TRY(dmalloc((void**)&an->A.off, ((size_t)m + 1) * sizeof(int32_t)));
  TRY(dmalloc((void**)&an->A.idx, ((size_t)nnz + 8) * sizeof(int32_t)));
  TRY(dmalloc((void**)&an->A.val, ((size_t)nnz + 8) * sizeof(double)));
  TRY(dmalloc((void**)&an->At.off, ((size_t)n + 1) * sizeof(int32_t)));
  TRY(dmalloc((void**)&an->At.idx, ((size_t)nnz + 8) * sizeof(int32_t)));
  TRY(dmalloc((void**)&an->At.val, ((size_t)nnz + 8) * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(an->A.off, a_off, ((size_t)m + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (nnz) {
    HIP_TRY(hipMemcpyAsync(an->A.idx, a_idx, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(an->A.val, a_val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, s));
  }
  HIP_TRY(hipMemsetAsync(an->A.idx + nnz, 0, 8 * sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(an->A.val + nnz, 0, 8 * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(an->At.idx + nnz, 0, 8 * sizeof(int32_t), s));
  HIP_TRY(hipMemsetAsync(an->At.val + nnz, 0, 8 * sizeof(double), s));
  // workspace: the permuted pair's three sorts are the largest user (7 arrays of nnz words + histograms); the ordering search needs
  // ~14 arrays of max(m, n) words
  {
    bool reorder = (flags & 1) != 0;
    {
      // (the ordering search keeps a dense quotient graph of m / row-block cells: beyond 4096 cells -- 8 M rows -- it does not run,
      // and its 5 extra arrays of nnz words are not reserved: 20 GB at 1e9 nonzeros)
      int G = 0, waves = 0, wcap = 0, brows = 0;
      if (!jag_geometry(m, 0, &G, &waves, &wcap, &brows) || (int64_t)m / brows + 2 > 4096) reorder = false;
    }
    const size_t words = (size_t)std::max<int64_t>(nnz, 1);
    const size_t verts = (size_t)std::max(m, n) + 64;
    size_t bytes = 4 * (words * (reorder ? 9 : 4) + (words / kRsTile + 2) * 256 + 4096) + (reorder ? 4 * verts * 20 + 48ull * 4096 * kLongRow * 4 + 4096ull * 4096 * 4 : 0) + (1 << 20);
    size_t spare_cap = 0;
    if (char* spare = take_spare_arena(device, bytes, &spare_cap)) {  // (the last analysis' workspace: see SpareArena)
      an->arena.base = spare, an->arena.cap = spare_cap;
      an->owned.push_back(spare);
    } else {
      an->arena.cap = bytes;
      TRY(dmalloc((void**)&an->arena.base, bytes));
    }
  }
  an->ms_upload = lap("upload A");
  // (PCIe is idle from here on: the vectors cross it now, next to the transposition and the ordering search)
  struct Prefetch {
    std::thread worker;
    ~Prefetch() { if (worker.joinable()) worker.join(); }
  } prefetch;
  if (c || lo || hi || lb || ub) {
    const double* src[5]  = {c, lb, ub, lo, hi};
    const size_t count[5] = {(size_t)n, (size_t)n, (size_t)n, (size_t)m, (size_t)m};
    for (int i = 0; i < 5; ++i) an->pref_src[i] = src[i];
    auto upload_vectors = [an, device, count] {
      if (hipSetDevice(device) != hipSuccess) return;
      for (int i = 0; i < 5; ++i) {
        if (!an->pref_src[i] || count[i] == 0) continue;
        double* d = nullptr;
        if (hipMalloc((void**)&d, count[i] * sizeof(double)) != hipSuccess) continue;
        if (hipMemcpy(d, an->pref_src[i], count[i] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
          (void)hipFree(d);
          continue;
        }
        an->pref_dev[i] = d;  // (read after the join below)
      }
      (void)hipGetLastError();
    };
    try {
      prefetch.worker = std::thread(upload_vectors);
    } catch (const std::system_error&) {  // (no thread to be had: create uploads the vectors itself, as it does without the prefetch)
    }
  }
  TRY(dev_transpose(s, an->arena, m, n, nnz, an->A, an->At));
  an->ms_transpose = lap("transpose");
  if ((flags & 1) && nnz > 0) {
    const size_t mark = an->arena.mark();
    // the order the matrix came in: nothing to look for when the jagged layout already applies
    TRY(estimate_saving(an, 0, nullptr, nullptr, &an->saving_natural[0]));
    if (an->saving_natural[0] >= 0.35 || !(flags & 1)) TRY(estimate_saving(an, 1, nullptr, nullptr, &an->saving_natural[1]));
    else an->saving_natural[1] = -2.0;  // (not evaluated: the ordering search runs anyway; create reads "< 0.35" as "no jagged layout")
    an->estimated = true;
    lap("estimate 0");
    int G = 0, waves = 0, wcap = 0, brows = 0;
    const bool big_enough = jag_geometry(m, 0, &G, &waves, &wcap, &brows) && jag_geometry(n, 0, &G, &waves, &wcap, &brows);
    if (big_enough && !(an->saving_natural[0] >= 0.35 && an->saving_natural[1] >= 0.35)) {
      uint32_t *rn2o = nullptr, *cn2o = nullptr;
      int32_t *ro2n = nullptr, *co2n = nullptr;
      TRY(find_ordering(an, &rn2o, &cn2o, &ro2n, &co2n, lap));
      if (an->method != 0) {
        an->row_new2old.resize(m), an->col_new2old.resize(n);
        HIP_TRY(hipMemcpyAsync(an->row_new2old.data(), rn2o, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(an->col_new2old.data(), cn2o, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        TRY(build_permuted_pair(an, ro2n, co2n));
        an->permuted = true;
        an->ms_permute = lap("permuted pair");
        // the vectors that travelled ahead follow the matrix into the new order, on the device (the maps are still in the arena):
        // the host never gathers them, the context never uploads them
        if (prefetch.worker.joinable()) prefetch.worker.join();
        bool all = true;
        for (int i = 0; i < 5; ++i) {
          if (!an->pref_dev[i]) { all = all && !an->pref_src[i]; continue; }
          const size_t count = i < 3 ? (size_t)n : (size_t)m;
          double* d = nullptr;
          HIP_TRY(hipMalloc((void**)&d, std::max<size_t>(count, 1) * sizeof(double)));  // (freed with pref_dev; the unpermuted copy it replaces: with `owned`)
          k_gather_f64<<<grid_of((int64_t)count), kT, 0, s>>>((int64_t)count, an->pref_dev[i], i < 3 ? cn2o : rn2o, d);
          an->owned.push_back(an->pref_dev[i]);
          an->pref_dev[i] = d;
        }
        HIP_TRY(hipGetLastError());
        an->pref_permuted = all;
        if (all) lap("vectors");
      }
    }
    an->arena.release(mark);
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (prefetch.worker.joinable()) prefetch.worker.join();
  if (getenv("CUOPT_AMD_TIMING"))
    fprintf(stderr, "[cuopt_amd setup]   analysis: %s| natural %.2f/%.2f levels(%d) %.2f/%.2f cells(%d) %.2f/%.2f -> method %d\n", an->laps.c_str(),
            an->saving_natural[0], an->saving_natural[1], an->bfs_levels, an->saving_levels[0], an->saving_levels[1], an->cell_rounds,
            an->saving_cells[0], an->saving_cells[1], an->method);
  return 0;
}

// out = {permuted, method, natural A, natural A^T, levels A, levels A^T, cells A, cells A^T (x 1e4), search levels, cell rounds}
int pdlpdev_analysis_info(pdlpdev_analysis* an, int32_t out[10])
{
  if (!an) return fail(-1, "pdlpdev_analysis_info: null");
  out[0] = an->permuted, out[1] = an->method;
  out[2] = (int32_t)(1e4 * an->saving_natural[0]), out[3] = (int32_t)(1e4 * an->saving_natural[1]);
  out[4] = (int32_t)(1e4 * an->saving_levels[0]), out[5] = (int32_t)(1e4 * an->saving_levels[1]);
  out[6] = (int32_t)(1e4 * an->saving_cells[0]), out[7] = (int32_t)(1e4 * an->saving_cells[1]);
  out[8] = an->bfs_levels, out[9] = an->cell_rounds;
  return 0;
}

// 1: the analysis reordered the matrix AND holds the problem vectors it was given in that order on the device (a context created from it
// takes them from there whichever host arrays -- the caller's own -- it is handed); 0: the caller permutes and passes them
int pdlpdev_analysis_vectors_in_order(pdlpdev_analysis* an) { return an && an->permuted && an->pref_permuted ? 1 : 0; }

// the maps of an accepted ordering: row_new2old[m], col_new2old[n] (either may be null); returns 1 when permuted, 0 when not
int pdlpdev_analysis_maps(pdlpdev_analysis* an, int32_t* row_new2old, int32_t* col_new2old)
{
  if (!an) return fail(-1, "pdlpdev_analysis_maps: null");
  if (!an->permuted) return 0;
  if (row_new2old) memcpy(row_new2old, an->row_new2old.data(), (size_t)an->m * sizeof(int32_t));
  if (col_new2old) memcpy(col_new2old, an->col_new2old.data(), (size_t)an->n * sizeof(int32_t));
  return 1;
}

// the matrices the device holds, as host CSR (parity tests; the sharded path slices the permuted matrix): which = 0 A, 1 A^T; any
// pointer may be null
int pdlpdev_analysis_download(pdlpdev_analysis* an, int which, int32_t* off, int32_t* idx, double* val)
{
  if (!an) return fail(-1, "pdlpdev_analysis_download: null");
  const DevCsr& M    = which ? an->At : an->A;
  const int32_t rows = which ? an->n : an->m;
  HIP_TRY(hipSetDevice(an->device));
  if (off) HIP_TRY(hipMemcpyAsync(off, M.off, ((size_t)rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream));
  if (idx && an->nnz) HIP_TRY(hipMemcpyAsync(idx, M.idx, (size_t)an->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream));
  if (val && an->nnz) HIP_TRY(hipMemcpyAsync(val, M.val, (size_t)an->nnz * sizeof(double), hipMemcpyDeviceToHost, an->stream));
  HIP_TRY(hipStreamSynchronize(an->stream));
  return 0;
}

// test hook: sorts n (key, value) pairs on the device (stable, by the low `bits` bits); vals null: iota
int pdlpdev_debug_sort_pairs(int device, int64_t n, const uint32_t* keys, const uint32_t* vals, int bits, uint32_t* keys_out, uint32_t* vals_out)
{
  HIP_TRY(hipSetDevice(device));
  DevArena ar;
  ar.cap = (size_t)(6 * n + (n / kRsTile + 2) * 256 + 4096) * 4 + (1 << 16);
  HIP_TRY(hipMalloc((void**)&ar.base, ar.cap));
  uint32_t* dk = ar.take<uint32_t>((size_t)n);
  uint32_t* dv = ar.take<uint32_t>((size_t)n);
  SortBufs B;
  int rc = sort_bufs_take(ar, &B, n);
  int slot = 0;
  if (rc == 0 && dk && dv) {
    (void)hipMemcpy(dk, keys, (size_t)n * 4, hipMemcpyHostToDevice);
    if (vals) (void)hipMemcpy(dv, vals, (size_t)n * 4, hipMemcpyHostToDevice);
    rc = radix_sort_pairs(nullptr, n, dk, vals ? dv : nullptr, B, bits, &slot);
    if (rc == 0) {
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(keys_out, B.k[slot], (size_t)n * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(vals_out, B.v[slot], (size_t)n * 4, hipMemcpyDeviceToHost);
    }
  }
  (void)hipFree(ar.base);
  return rc;
}

// test hook: exclusive scan of n ints -> n + 1 outputs
int pdlpdev_debug_scan(int device, int64_t n, const int32_t* in, int32_t* out)
{
  HIP_TRY(hipSetDevice(device));
  int32_t *din = nullptr, *dout = nullptr, *bs = nullptr;
  HIP_TRY(hipMalloc((void**)&din, (size_t)std::max<int64_t>(n, 1) * 4));
  HIP_TRY(hipMalloc((void**)&dout, (size_t)(n + 1) * 4));
  HIP_TRY(hipMalloc((void**)&bs, (size_t)(n / kScanTile + 2) * 4));
  (void)hipMemcpy(din, in, (size_t)n * 4, hipMemcpyHostToDevice);
  int rc = dev_exclusive_scan(nullptr, din, dout, n, bs);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(out, dout, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost);
  (void)hipFree(din), (void)hipFree(dout), (void)hipFree(bs);
  return rc;
}

}  // extern "C"

// ================================================================================================
// gather_working_set (kernels_panel.hip) on a device-resident index array: the 128-byte lines of the gathered vector that up to four
// windows of 512 K consecutive nonzeros touch -- same windows, same count, no index array on the host
// ================================================================================================
namespace {
// one workgroup per window, the bitmap of touched lines in LDS (cols / 16 bits: 7.8 KB at 1e6 columns; a global bitmap with
// atomicOr serialised on a few thousand words: 1.4 ms per call)
constexpr int kWsThreads = 1024;
__global__ void __launch_bounds__(kWsThreads) k_ws_lines(const int32_t* __restrict__ idx, int64_t nnz, int64_t window, int samples, int words,
                                                         int32_t* __restrict__ lines)
{
  extern __shared__ uint32_t bitmap[];
  __shared__ int total;
  const int s = blockIdx.x;
  for (int i = threadIdx.x; i < words; i += kWsThreads) bitmap[i] = 0u;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const int64_t first = samples == 1 ? 0 : (nnz - window) * s / (samples - 1);
  const int64_t last  = min(nnz, first + window);
  int mine = 0;
  for (int64_t k = first + threadIdx.x; k < last; k += kWsThreads) {
    const uint32_t line = (uint32_t)idx[k] >> 4, bit = 1u << (line & 31u);
    if (!(atomicOr(&bitmap[line >> 5], bit) & bit)) ++mine;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) lines[s] = total;
}
}  // namespace

// returns 1 when the bitmap does not fit a workgroup's LDS (more than ~2e7 columns: the caller counts on the host)
int gather_working_set_device(pdlpdev_ctx* c, const int32_t* d_idx, int64_t nnz, int32_t cols, int64_t* bytes)
{
  *bytes = 0;
  if (cols <= 0 || nnz <= 0) return 0;
  const int64_t window = 512 * 1024;
  const int samples    = nnz <= window ? 1 : (int)std::min<int64_t>(4, (nnz + window - 1) / window);
  const size_t words   = ((size_t)(cols >> 4) >> 5) + 1;
  if (words * 4 > 150 * 1024) return 1;
  int32_t* lines = nullptr;
  TRY(dev_alloc(c, &lines, (size_t)samples));
  {
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(done.begin(), done.end(), c->device) == done.end()) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_ws_lines, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      done.push_back(c->device);
    }
  }
  k_ws_lines<<<samples, kWsThreads, words * 4, c->stream>>>(d_idx, nnz, window, samples, (int)words, lines);
  int32_t h[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h, lines, samples * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int64_t total = 0;
  for (int s = 0; s < samples; ++s) total += (int64_t)h[s] * 128;
  *bytes = total / samples;
  return 0;
}

// ================================================================================================
// up to four windows of a device-resident index array (gather_working_set of a matrix the host does not hold)
// ================================================================================================
int analysis_fetch_idx_windows(pdlpdev_analysis* an, int transposed, int64_t nnz, std::vector<int32_t>* sparse,
                               std::vector<std::pair<int64_t, int64_t>>* windows)
{
  const int64_t window = 512 * 1024;
  const int samples    = nnz <= window ? 1 : (int)std::min<int64_t>(4, (nnz + window - 1) / window);
  windows->clear();
  sparse->clear();
  const int32_t* d_idx = transposed ? an->At.idx : an->A.idx;
  for (int s = 0; s < samples; ++s) {
    const int64_t first = samples == 1 ? 0 : (nnz - window) * s / (samples - 1);
    const int64_t last  = std::min(nnz, first + window);
    windows->emplace_back(first, last);
  }
  size_t total = 0;
  for (auto& w : *windows) total += (size_t)(w.second - w.first);
  sparse->resize(total);
  size_t at = 0;
  for (auto& w : *windows) {
    HIP_TRY(hipMemcpyAsync(sparse->data() + at, d_idx + w.first, (size_t)(w.second - w.first) * sizeof(int32_t), hipMemcpyDeviceToHost, an->stream));
    at += (size_t)(w.second - w.first);
  }
  HIP_TRY(hipStreamSynchronize(an->stream));
  return 0;
}

// ================================================================================================
// synthetic LP generated ON THE DEVICE (scale checks near the reference's stated capacity, docs/cuopt/source/faq.rst:368-370: the
// host generator of cuopt_amd/synthetic.py needs minutes and tens of GB at 1e9 nonzeros).  Same recipe -- a known primal-dual optimal
// pair by construction, equalities on the first half of the rows, '>=' rows with slack on the second -- with the columns of a row
// drawn one per stratum of n / k columns (distinct and ascending by construction).  Deterministic in (seed, m, n, k) except for the
// low bits of c = A^T y* + z* (atomic adds); the known optimum is computed from the c that is returned.
// ================================================================================================
namespace {
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double unit(unsigned long long h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
__device__ __forceinline__ double normal(unsigned long long h)
{
  const double u1 = 1.0 - unit(mix64(h)), u2 = unit(mix64(h ^ 0xD1B54A32D192ED03ull));
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
__global__ void __launch_bounds__(kT) k_gen_cols(int32_t n, unsigned long long seed, double* __restrict__ xs, double* __restrict__ c)
{
  for (int64_t j = (int64_t)blockIdx.x * kT + threadIdx.x; j < n; j += (int64_t)gridDim.x * kT) {
    const unsigned long long h = mix64(seed ^ (0x1000000000ull + (unsigned long long)j));
    const double x = unit(h) < 0.5 ? 0.0 : 1.0 - unit(mix64(h ^ 1));
    xs[j] = x;
    c[j]  = x > 0.0 ? 0.0 : unit(mix64(h ^ 2));  // z*: the reduced cost of a variable at its bound
  }
}
__global__ void __launch_bounds__(kT) k_gen_rows(int32_t m, int32_t n, int32_t k, unsigned long long seed, const double* __restrict__ xs,
                                                 int32_t* __restrict__ off, int32_t* __restrict__ idx, double* __restrict__ val,
                                                 double* __restrict__ ys, double* __restrict__ lo, double* __restrict__ hi, double* __restrict__ c)
{
  const int64_t stratum = n / k;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < m; i += (int64_t)gridDim.x * kT) {
    const unsigned long long hr = mix64(seed ^ (0x2000000000ull + (unsigned long long)i));
    double y = normal(hr);
    const bool equality = i < m / 2;
    if (!equality) y = unit(mix64(hr ^ 7)) < 0.3 ? 0.0 : fabs(y);
    ys[i] = y;
    double ax = 0.0;
    const int64_t base = i * (int64_t)k;
    for (int t = 0; t < k; ++t) {
      const unsigned long long he = mix64(seed ^ ((unsigned long long)i * 64ull + (unsigned long long)t) ^ 0x3000000000000ull);
      const int64_t width = t == k - 1 ? (int64_t)n - stratum * t : stratum;
      const int32_t j     = (int32_t)(stratum * t + (int64_t)(he % (unsigned long long)width));
      const double a      = normal(he ^ 0x55);
      idx[base + t] = j, val[base + t] = a;
      ax += a * xs[j];
      if (y != 0.0) atomicAdd(&c[j], a * y);  // c = A^T y* + z*
    }
    off[i] = (int32_t)base;
    if (i == m - 1) off[m] = (int32_t)(base + k);
    if (equality) {
      lo[i] = ax, hi[i] = ax;
    } else {
      lo[i] = ax - (y > 0.0 ? 0.0 : unit(mix64(hr ^ 9))), hi[i] = INFINITY;
    }
  }
}
}  // namespace

extern "C" int pdlpdev_synthetic_lp(int device, int32_t m, int32_t n, int32_t k, uint64_t seed, int32_t* offsets, int32_t* indices, double* values,
                                    double* c, double* lo, double* hi, double* x_star, double* y_star)
{
  if (m <= 0 || n <= 0 || k <= 0 || k > n || (int64_t)m * k >= ((int64_t)1 << 31)) return fail(-1, "pdlpdev_synthetic_lp: bad sizes (m * k must stay below 2^31)");
  if (pdlpdev_device_count() <= device) return fail(-5, "pdlpdev_synthetic_lp: no HIP device %d visible", device);
  HIP_TRY(hipSetDevice(device));
  const int64_t nnz = (int64_t)m * k;
  int32_t *d_off = nullptr, *d_idx = nullptr;
  double *d_val = nullptr, *d_c = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_xs = nullptr, *d_ys = nullptr;
  HIP_TRY(hipMalloc((void**)&d_off, ((size_t)m + 1) * 4)); HIP_TRY(hipMalloc((void**)&d_idx, (size_t)nnz * 4)); HIP_TRY(hipMalloc((void**)&d_val, (size_t)nnz * 8));
  HIP_TRY(hipMalloc((void**)&d_c, (size_t)n * 8)); HIP_TRY(hipMalloc((void**)&d_xs, (size_t)n * 8));
  HIP_TRY(hipMalloc((void**)&d_lo, (size_t)m * 8)); HIP_TRY(hipMalloc((void**)&d_hi, (size_t)m * 8)); HIP_TRY(hipMalloc((void**)&d_ys, (size_t)m * 8));
  k_gen_cols<<<grid_of(n), kT>>>(n, seed, d_xs, d_c);
  k_gen_rows<<<grid_of(m), kT>>>(m, n, k, seed, d_xs, d_off, d_idx, d_val, d_ys, d_lo, d_hi, d_c);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(offsets, d_off, ((size_t)m + 1) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(indices, d_idx, (size_t)nnz * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(values, d_val, (size_t)nnz * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c, d_c, (size_t)n * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(lo, d_lo, (size_t)m * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hi, d_hi, (size_t)m * 8, hipMemcpyDeviceToHost));
  if (x_star) HIP_TRY(hipMemcpy(x_star, d_xs, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (y_star) HIP_TRY(hipMemcpy(y_star, d_ys, (size_t)m * 8, hipMemcpyDeviceToHost));
  for (void* p : {(void*)d_off, (void*)d_idx, (void*)d_val, (void*)d_c, (void*)d_lo, (void*)d_hi, (void*)d_xs, (void*)d_ys}) (void)hipFree(p);
  return 0;
}
