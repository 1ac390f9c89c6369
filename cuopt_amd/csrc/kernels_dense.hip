// gfx950 kernels of the dense layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "pdlp_layouts.hpp"

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

// ================================================================================================
// host side of the layout
// ================================================================================================
void find_dense_segments(int32_t m, int32_t n, const int32_t* off, const int32_t* idx, DenseHost* D)
{
  const int64_t nnz = off[m];
  for (int32_t r = 0; r < m; ++r) {
    bool owner = false;
    if (off[r + 1] - off[r] < kDenseMin) continue;  // (a 1e7-nonzero matrix of short rows: this scan was 8 ms of the set-up)
    // The C API neither sorts rows nor merges duplicates, and the sparse remainder is cut out by COLUMN RANGE below: a row whose
    // indices are not strictly increasing (a stray duplicate of a column inside a run would vanish from both copies: round-3
    // advisor) keeps all its entries in the sparse layouts.
    bool canonical = true;
    for (int k = off[r]; k + 1 < off[r + 1] && canonical; ++k) canonical = idx[k + 1] > idx[k];
    if (!canonical) continue;
    for (int k = off[r]; k < off[r + 1];) {
      int e = k;
      while (e + 1 < off[r + 1] && idx[e + 1] == idx[e] + 1) ++e;
      const int len = e - k + 1;
      if (len >= kDenseMin) {
        if (!owner) {
          D->row.push_back(r);
          D->row_seg.push_back((int32_t)D->seg_row.size());
          owner = true;
        }
        D->seg_row.push_back(r), D->seg_c0.push_back(idx[k]), D->seg_len.push_back(len), D->seg_ptr.push_back((int32_t)D->nent);
        for (int q = k; q <= e; ++q) D->perm.push_back(q);
        D->nent += len;
      }
      k = e + 1;
    }
  }
  D->row_seg.push_back((int32_t)D->seg_row.size());
  D->seg_ptr.push_back((int32_t)D->nent);
  // worth a second code path only when the segments carry a visible share of the matrix
  const int want = (int)cuopt_amd::tune_int("dense", -1);  // CUOPT_AMD_TUNE=dense=..: 0 off, 1 on whenever a segment exists, default: >= 2 % of the nonzeros
  D->on = want != 0 && D->nent > 0 && (want == 1 || D->nent * 50 >= nnz) && D->seg_row.size() <= 65536;
  if (!D->on) return;
  for (size_t b = 0; b < D->row.size(); ++b) {
    D->row_ch.push_back((int32_t)D->ch_seg.size());
    for (int32_t q = D->row_seg[b]; q < D->row_seg[b + 1]; ++q)
      for (int32_t k0 = 0; k0 < D->seg_len[q]; k0 += kDenseChunk) D->ch_seg.push_back(q), D->ch_k0.push_back(k0);
  }
  D->row_ch.push_back((int32_t)D->ch_seg.size());
  // sparse remainder of A
  D->first_seg.assign(m, -1);
  for (size_t b = 0; b < D->row.size(); ++b) D->first_seg[D->row[b]] = D->row_seg[b];
  D->s_off.assign((size_t)m + 1, 0);
  D->s_idx.reserve((size_t)(nnz - D->nent)), D->s_perm.reserve((size_t)(nnz - D->nent));
  auto covered = [&](int32_t r, int32_t c) {
    const int32_t f = D->first_seg[r];
    if (f < 0) return false;
    for (int32_t q = f; q < (int32_t)D->seg_row.size() && D->seg_row[q] == r; ++q)
      if (c >= D->seg_c0[q] && c < D->seg_c0[q] + D->seg_len[q]) return true;
    return false;
  };
  for (int32_t r = 0; r < m; ++r) {
    for (int k = off[r]; k < off[r + 1]; ++k)
      if (!covered(r, idx[k])) D->s_idx.push_back(idx[k]), D->s_perm.push_back(k);
    D->s_off[r + 1] = (int32_t)D->s_idx.size();
  }
  // 256-column tiles of A^T's side
  const int ntiles_all = (n + kBlock - 1) / kBlock;
  std::vector<int32_t> cnt(ntiles_all, 0);
  for (size_t q = 0; q < D->seg_row.size(); ++q)
    for (int t = D->seg_c0[q] / kBlock; t <= (D->seg_c0[q] + D->seg_len[q] - 1) / kBlock; ++t) cnt[t]++;
  std::vector<int32_t>& slot = D->tile_slot;
  slot.assign(ntiles_all, -1);
  D->tile_ptr.push_back(0);
  for (int t = 0; t < ntiles_all; ++t)
    if (cnt[t]) {
      slot[t] = (int32_t)D->tile_id.size();
      D->tile_id.push_back(t);
      D->tile_ptr.push_back(D->tile_ptr.back() + cnt[t]);
    }
  D->tile_seg.assign((size_t)D->tile_ptr.back(), 0);
  std::vector<int32_t> cur(D->tile_ptr.begin(), D->tile_ptr.end() - 1);
  for (size_t q = 0; q < D->seg_row.size(); ++q)  // segments in ascending row order: the order A^T's rows list them in
    for (int t = D->seg_c0[q] / kBlock; t <= (D->seg_c0[q] + D->seg_len[q] - 1) / kBlock; ++t) D->tile_seg[cur[slot[t]]++] = (int32_t)q;
}

// the sparse remainder of A^T (entry (j, i) goes when row i of A holds column j in a segment)
void strip_transpose(const DenseHost& Din, DenseHost* D, int32_t n, const int32_t* t_off, const int32_t* t_idx)
{
  (void)Din;
  D->st_off.assign((size_t)n + 1, 0);
  for (int32_t j = 0; j < n; ++j) {
    for (int k = t_off[j]; k < t_off[j + 1]; ++k) {
      const int32_t r = t_idx[k], f = D->first_seg[r];
      bool cov = false;
      for (int32_t q = f; f >= 0 && q < (int32_t)D->seg_row.size() && D->seg_row[q] == r; ++q)
        if (j >= D->seg_c0[q] && j < D->seg_c0[q] + D->seg_len[q]) { cov = true; break; }
      if (!cov) D->st_idx.push_back(r), D->st_perm.push_back(k);
    }
    D->st_off[j + 1] = (int32_t)D->st_idx.size();
  }
}
