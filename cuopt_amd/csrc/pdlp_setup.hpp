// Device-side set-up of the PDLP solver (kernels_setup.hip): everything that is O(nnz) in front of the first PDHG step runs on the
// GPU from ONE upload of A -- the explicit transpose (the reference: raft::sparse::linalg::csr_transpose on the device,
// cpp/src/mip/problem/problem.cu:277-309), the analysis pass that looks for structure the matrix arrives without (the reference
// delegates its analysis to cusparseSpMV_preprocess, cpp/src/linear_programming/cusparse_view.cu:92-115,254-265; here: breadth-first
// level orderings and seeded cells on the bipartite row-column graph, gated by the jagged layout's own cost estimate), the permuted
// CSR pair, and the slab-major panel construction.  Hand-written primitives (stable LSD radix sort of key/value pairs, exclusive
// scan); no rocPRIM / hipCUB / rocSPARSE.
#pragma once
#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"

struct DevCsr {
  int32_t* off = nullptr;
  int32_t* idx = nullptr;
  double* val  = nullptr;
};

// one hipMalloc, bump allocation, freed as a whole (the analysis' temporaries: sort buffers, levels, frontiers)
struct DevArena {
  char* base  = nullptr;
  size_t cap  = 0, used = 0, peak = 0;
  template <class T>
  T* take(size_t count)
  {
    const size_t b = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (used + b > cap) return nullptr;
    T* p = (T*)(base + used);
    used += b;
    peak = std::max(peak, used);
    return p;
  }
  size_t mark() const { return used; }
  void release(size_t m) { used = m; }
};

struct pdlpdev_analysis {
  int device = 0;
  hipStream_t stream = nullptr;
  double* pinned = nullptr;  // with the stream and `chunk`: the bundle contexts recycle (pdlp_ctx.hpp Recycled); a context that is created
  char* chunk    = nullptr;  // from the analysis takes all three over (bundle_owned = false from then on)
  bool bundle_owned = true;
  int32_t m = 0, n = 0;
  int64_t nnz = 0;
  DevCsr A, At;  // device-resident CSR of A (m x n) and of A^T (n x m): PERMUTED when `permuted`; owned until a context adopts them
  bool adopted = false;
  // the caller's host CSR (valid until the analysis is destroyed or consumed) -- the UNPERMUTED matrix
  const int32_t *h_off = nullptr, *h_idx = nullptr;
  const double* h_val  = nullptr;
  // ordering
  bool permuted = false;
  bool estimated = false;  // saving_natural holds the jagged layout's sampled estimate of the matrices as they came
  int method    = 0;  // 0 none, 1 breadth-first levels + barycentre sweeps (band-like), 2 seeded cells (block-like)
  std::vector<int32_t> row_new2old, col_new2old;
  double saving_natural[2] = {0, 0}, saving_levels[2] = {0, 0}, saving_cells[2] = {0, 0};  // jagged-layout estimate {A, A^T}
  int bfs_levels = 0, cell_rounds = 0;
  int64_t bfs_reached = 0;
  // host mirrors of the structure the device holds (lazily downloaded; the unpermuted A is the caller's own arrays)
  cuopt_amd::PoolArray<int32_t> hp_off, hp_idx, hpt_off, hpt_idx;
  bool have_hp = false, have_hp_off = false, have_hpt_off = false, have_hpt_idx = false;
  // temporaries
  DevArena arena;
  std::vector<void*> owned;  // hipMalloc'ed blocks this object frees (the arena; A / A^T unless adopted)
  std::string laps;          // "phase ms; phase ms; ..." of the last analyze call (CUOPT_AMD_TIMING prints it)
  double ms_transpose = 0, ms_order = 0, ms_permute = 0, ms_upload = 0;
  // the problem vectors (c, lo, hi, lb, ub as the caller holds them), uploaded by a helper thread while the analysis' kernels run:
  // PCIe is idle once A is over.  A context created from the analysis copies them device to device (pdlp_create.hip).
  const double* pref_src[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double* pref_dev[5]       = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // (a permuted analysis gathers them into the new order on the device -- pref_permuted -- so that the host never does)
  bool pref_permuted = false;
  const double* prefetched(const double* host) const
  {
    if ((permuted && !pref_permuted) || !host) return nullptr;
    for (int i = 0; i < 5; ++i)
      if (pref_src[i] == host && pref_dev[i]) return pref_dev[i];
    return nullptr;
  }
};

// ---- defined in kernels_setup.hip ------------------------------------------------------------------------------------------------
// host structure of the (possibly permuted) matrices, downloaded on first use
const int32_t* analysis_host_off(pdlpdev_analysis* an);   // A: m + 1
const int32_t* analysis_host_idx(pdlpdev_analysis* an);   // A: nnz
const int32_t* analysis_host_idx_rows(pdlpdev_analysis* an, const int32_t* h_off, const std::vector<int32_t>& rows);
const int32_t* analysis_host_t_off(pdlpdev_analysis* an); // A^T: n + 1
const int32_t* analysis_host_t_idx(pdlpdev_analysis* an); // A^T: nnz
// up to four windows of the device's index array (gather_working_set of a matrix whose indices live on the device)
int analysis_fetch_idx_windows(pdlpdev_analysis* an, int transposed, int64_t nnz, std::vector<int32_t>* host_idx_sparse,
                               std::vector<std::pair<int64_t, int64_t>>* windows);
// gather_working_set on a device-resident index array (same windows, same count)
int gather_working_set_device(pdlpdev_ctx* c, const int32_t* d_idx, int64_t nnz, int32_t cols, int64_t* bytes);
// slab-major panels built on the device from a resident CSR; same arrays, bit for bit, as build_panels + upload_panels
// jagged rows + LDS column sets from the device-resident CSR (kernels_setup.hip): 0 built / not worth it (dst->on says which, dst->saving
// the layout's estimate), 1 not handled here (the host construction takes over), < 0 error
int build_jag_device(pdlpdev_ctx* c, pdlpdev_ctx::Jag* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                     const double* d_val, int mode, int cus);
// gather-free layout from the device-resident CSR (kernels_setup.hip): 0 built or does not fit (dst->on says which, *why the reason),
// 1 not handled here (nothing allocated: the host construction takes over), < 0 error
int build_pb_wide_device(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                         int cus, bool forced, std::string* why);
int build_pb_device(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, int32_t rows, int32_t cols, const int32_t* h_off, const int32_t* d_off, const int32_t* d_idx,
                    int cus, bool forced, std::string* why);
int build_panels_device(pdlpdev_ctx* c, pdlpdev_ctx::Panels* dst, int32_t rows, int32_t cols, const int32_t* h_off,
                        const int32_t* d_off, const int32_t* d_idx, const double* d_val, int64_t slab_bytes, bool force);
// the first half of build_panels (kernels_panel.hip): geometry, long-tail decision, own rows, the cut into panels -- from the row
// offsets alone; returns false when the layout does not apply
bool panel_plan(PanelHost* P, int32_t rows, int32_t cols, const int32_t* off, int64_t slab_bytes, bool force,
                const std::vector<int32_t>* dense_first_seg, std::vector<char>* is_own, int64_t* own_nnz, int64_t* own_from);
// jagged layout (kernels_jag.hip): geometry for a matrix of `rows` rows, and the cost estimate on a MINI CSR that holds `nsamples`
// row blocks of `brows` consecutive rows each (columns ascending inside a row)
bool jag_geometry(int32_t rows, int mode, int* G, int* waves, int* wcap, int* brows);
double jag_estimate_on_samples(int nsamples, int32_t brows, int wcap, const int32_t* soff, const int32_t* sidx);
