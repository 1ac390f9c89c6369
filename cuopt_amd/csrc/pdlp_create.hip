// Context creation and destruction of the device layer (pdlp_device.h: pdlpdev_create*, pdlpdev_destroy): uploads (or adoption of an
// analysed matrix's device arrays), the choice and construction of the SpMV layouts of both matrices, the vectors' slab.  The kernels,
// the loop and everything that runs after creation: pdlp_device.hip.
#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"

// (defined in pdlp_device.hip)
__global__ void k_fill(int64_t n, double* __restrict__ d, double v);
int sync_panel_values(pdlpdev_ctx* c);
int resident_tier(int m, int n, int64_t nnz);

static thread_local int g_create_sharded = 0;  // pdlpdev_create_hint: the next context will run behind a communicator

static int create_impl(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                       const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                       const int32_t* at_indices, const double* at_values,
                       void (*transpose_ready)(void*), void* user, const double* c, const double* lo,
                       const double* hi, const double* lb, const double* ub, pdlpdev_analysis* an);

extern "C" {

const char* pdlpdev_last_error(void) { return g_err.c_str(); }
void pdlpdev_create_hint(int sharded) { g_create_sharded = sharded; }
static thread_local pdlpdev_ctx* g_create_stream_donor = nullptr;
void pdlpdev_create_share_stream(pdlpdev_ctx* donor) { g_create_stream_donor = donor; }
int pdlpdev_resident_size(int32_t m, int32_t n, int64_t nnz) { return resident_tier(m, n, nnz) >= 0 ? 1 : 0; }

int pdlpdev_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pdlpdev_device_info(int dev, char* name, int len, int* compute_units, int64_t* hbm_bytes)
{
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (name && len > 0) snprintf(name, (size_t)len, "%s (%s)", p.name, p.gcnArchName);
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return 0;
}

int pdlpdev_create(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                   const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                   const int32_t* at_indices, const double* at_values, const double* c,
                   const double* lo, const double* hi, const double* lb, const double* ub)
{
  return pdlpdev_create_overlapped(out, device, m, n, a_offsets, a_indices, a_values, at_offsets, at_indices, at_values,
                                   nullptr, nullptr, c, lo, hi, lb, ub);
}

int pdlpdev_create_overlapped(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                              const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                              const int32_t* at_indices, const double* at_values,
                              void (*transpose_ready)(void*), void* user, const double* c, const double* lo,
                              const double* hi, const double* lb, const double* ub)
{
  if (!a_offsets || !at_offsets) return fail(-1, "pdlpdev_create: bad argument");
  return create_impl(out, device, m, n, a_offsets, a_indices, a_values, at_offsets, at_indices, at_values, transpose_ready, user, c, lo,
                     hi, lb, ub, nullptr);
}

// The context of an analysed matrix (pdlpdev_analyze): A and A^T are already on the device (the analysis' arrays are adopted, nothing
// of the matrix crosses PCIe again), the layouts are built from them -- panels on the device, the others on the host from the
// structure it holds or fetches.  c / lo / hi / lb / ub are in the order of the matrices the device holds (the caller applies
// pdlpdev_analysis_maps when the analysis permuted them).  The analysis must be destroyed afterwards (pdlpdev_analysis_destroy).
int pdlpdev_create_from_analysis(pdlpdev_ctx** out, pdlpdev_analysis* an, const double* c, const double* lo, const double* hi,
                                 const double* lb, const double* ub)
{
  if (!an || an->adopted) return fail(-1, "pdlpdev_create_from_analysis: no (or an already consumed) analysis");
  const int32_t* a_off = analysis_host_off(an);
  // (a permuted matrix's indices live on the device: they come to the host only if a host construction asks for them -- 88 MB at 1e7
  // nonzeros; the caller's own array otherwise)
  const int32_t* a_idx = an->permuted ? nullptr : analysis_host_idx(an);
  const int32_t* t_off = analysis_host_t_off(an);
  return create_impl(out, an->device, an->m, an->n, a_off, a_idx, nullptr, t_off, nullptr, nullptr, nullptr, nullptr, c, lo, hi, lb, ub, an);
}
}  // extern "C"

static int create_impl(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                       const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                       const int32_t* at_indices, const double* at_values,
                       void (*transpose_ready)(void*), void* user, const double* c, const double* lo,
                       const double* hi, const double* lb, const double* ub, pdlpdev_analysis* an)
{
  roctx::Range range("pdlp: device set-up (upload, layouts)");
  if (!out || m < 0 || n < 0 || !a_offsets || !at_offsets) return fail(-1, "pdlpdev_create: bad argument");
  if (pdlpdev_device_count() <= device)
    return fail(-5, "pdlpdev_create: no HIP device %d visible (this solver has no CPU fallback)", device);
  HIP_TRY(hipSetDevice(device));
  const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  auto tlast = std::chrono::steady_clock::now();
  pdlpdev_ctx* ctx = nullptr;
  auto lap = [&](const char* what) {
    if (!timing) return;
    // (the context's own stream: a device-wide synchronisation would break a graph capture another thread's solver is in the middle of)
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    else (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cuopt_amd setup]   dev: %-22s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - tlast).count());
    tlast = now;
  };
  ctx         = new pdlpdev_ctx();
  ctx->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->cus = cus;
  }
  ctx->m = m, ctx->n = n, ctx->nnz = a_offsets[m];
  *out = ctx;
  {
    Recycled r;
    if (an && an->bundle_owned && an->stream && an->pinned && an->chunk) {  // the analysis' stream, pinned block and chunk move here
      ctx->stream = an->stream, ctx->scal_h = an->pinned, ctx->arena = an->chunk, ctx->first_chunk = an->chunk;
      an->bundle_owned = false;
      HIP_TRY(hipMemsetAsync(ctx->arena, 0, kArenaChunk, ctx->stream));
    } else if (g_create_stream_donor && g_create_stream_donor->device == device) {  // pdlpdev_create_share_stream
      ctx->stream = g_create_stream_donor->stream, ctx->stream_borrowed = true;
      g_create_stream_donor = nullptr;
      HIP_TRY(hipHostMalloc((void**)&ctx->scal_h, kScalars * sizeof(double) + sizeof(pdlpdev_ctl)));
      HIP_TRY(hipMalloc((void**)&ctx->arena, kArenaChunk));
      HIP_TRY(hipMemsetAsync(ctx->arena, 0, kArenaChunk, ctx->stream));
      ctx->first_chunk = ctx->arena;
    } else if (take_recycled(device, &r)) {
      ctx->stream = r.stream, ctx->scal_h = r.pinned, ctx->arena = r.chunk, ctx->first_chunk = r.chunk;
      HIP_TRY(hipMemsetAsync(ctx->arena, 0, kArenaChunk, ctx->stream));
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      HIP_TRY(hipHostMalloc((void**)&ctx->scal_h, kScalars * sizeof(double) + sizeof(pdlpdev_ctl)));  // one pinned block
      HIP_TRY(hipMalloc((void**)&ctx->arena, kArenaChunk));
      HIP_TRY(hipMemsetAsync(ctx->arena, 0, kArenaChunk, ctx->stream));
      ctx->first_chunk = ctx->arena;
    }
    ctx->ctl_h = (pdlpdev_ctl*)(ctx->scal_h + kScalars);
  }
  const size_t nnz = (size_t)ctx->nnz;
  // Everything that needs A only comes first; the caller may still be transposing on other threads (A^T is not
  // touched before transpose_ready returns).
  if (an) {
    // adopted: both matrices are the analysis' device arrays (allocated with the 8 spare entries the stream kernel may over-read)
    ctx->a_off = an->A.off, ctx->a_idx = an->A.idx, ctx->a_val = an->A.val;
    ctx->at_off = an->At.off, ctx->at_idx = an->At.idx, ctx->at_val = an->At.val;
    for (void* p : {(void*)an->A.off, (void*)an->A.idx, (void*)an->A.val, (void*)an->At.off, (void*)an->At.idx, (void*)an->At.val}) {
      an->owned.erase(std::remove(an->owned.begin(), an->owned.end(), p), an->owned.end());
      ctx->allocs.push_back(p);
    }
    an->adopted = true;
    ctx->bytes += (int64_t)(2 * (nnz + 8) * 12 + ((size_t)m + n + 2) * 4);
  } else {
    TRY(upload_i32(ctx, &ctx->a_off, a_offsets, (size_t)m + 1));
    TRY(upload_i32(ctx, &ctx->a_idx, a_indices, nnz, 8));  // +8: the vector loads of the stream kernel may over-read
    TRY(upload_f64(ctx, &ctx->a_val, a_values, nnz, 8));
  }
  lap("alloc + upload A");
  auto long_rows = [](int32_t rows, const int32_t* off) {
    constexpr int kParts = 16;
    std::vector<int32_t> part[kParts], v;
    cuopt_amd::parallel_tasks(kParts, [&](int t) {
      const int32_t a = (int32_t)((int64_t)rows * t / kParts), b = (int32_t)((int64_t)rows * (t + 1) / kParts);
      for (int32_t r = a; r < b; ++r)
        if (off[r + 1] - off[r] > kLongRow) part[t].push_back(r);
    }, (int64_t)rows * 8);
    for (int t = 0; t < kParts; ++t) v.insert(v.end(), part[t].begin(), part[t].end());
    return v;
  };
  std::vector<int32_t> la = long_rows(m, a_offsets), lat;  // alive until the stream is synchronised at the end
  ctx->a_nlong = (int)la.size();
  if (ctx->a_nlong) TRY(upload_i32(ctx, &ctx->a_long, la.data(), la.size()));
  // dense row segments leave the hot loop's copy of the matrix (single-GPU solves)
  const bool one_gpu = !g_create_sharded;
  g_create_sharded   = 0;
  DenseHost DH;
  std::vector<int32_t> hA_off, hA_idx, hA_perm, hT_off, hT_idx, hT_perm;  // the hot CSRs where they differ from the full ones
  lap("long rows A");
  auto a_idx_host = [&]() -> const int32_t* { return a_indices ? a_indices : (an ? analysis_host_idx(an) : nullptr); };
  if (one_gpu) {
    // (the scan reads the indices of rows with at least kDenseMin entries only: none of them, no indices needed)
    std::vector<int32_t> candidates;
    for (int32_t r : la)
      if (a_offsets[r + 1] - a_offsets[r] >= kDenseMin) candidates.push_back(r);
    // (no candidate: no segment can exist, and the pass over 1e6 rows was 0.26 ms.  A permuted matrix's indices live on the device:
    //  a few candidate rows -- the linking rows of a block-angular LP -- come over on their own instead of the whole 40 MB array)
    if (!candidates.empty()) {
      const int32_t* scan_idx = a_indices ? a_indices : (an && candidates.size() <= 64 ? analysis_host_idx_rows(an, a_offsets, candidates) : a_idx_host());
      find_dense_segments(m, n, a_offsets, scan_idx, &DH);
      if (DH.on && !a_indices) {  // (segments found after all: the host constructions behind them read every row)
        DenseHost again;
        find_dense_segments(m, n, a_offsets, a_idx_host(), &again);
        DH = std::move(again);
      }
    }
  }
  lap("dense scan");
  if (DH.on) hA_off.swap(DH.s_off), hA_idx.swap(DH.s_idx), hA_perm.swap(DH.s_perm);
  const bool hot_a     = !hA_off.empty();
  const int32_t* A_off = hot_a ? hA_off.data() : a_offsets;
  const int32_t* A_idx = hot_a ? hA_idx.data() : a_indices;  // (null: an analysed, permuted matrix whose indices stayed on the device)
  auto A_idx_host = [&]() -> const int32_t* { return A_idx ? A_idx : a_idx_host(); };
  ctx->ha_off = ctx->a_off, ctx->ha_idx = ctx->a_idx, ctx->ha_val = ctx->a_val;
  ctx->dense.hot_nnz = (int64_t)A_off[m];
  if (hot_a) {
    TRY(upload_i32(ctx, &ctx->ha_off, A_off, (size_t)m + 1));
    TRY(upload_i32(ctx, &ctx->ha_idx, A_idx, (size_t)ctx->dense.hot_nnz, 8));
    TRY(dev_alloc(ctx, &ctx->ha_val, (size_t)ctx->dense.hot_nnz + 8));
    TRY(upload_i32(ctx, &ctx->dense.s_perm_a, hA_perm.data(), hA_perm.size()));
  }
  if (DH.on) {
    pdlpdev_ctx::Dense& D = ctx->dense;
    D.nrows = (int)DH.row.size(), D.nseg = (int)DH.seg_row.size(), D.ntiles = (int)DH.tile_id.size(), D.nent = DH.nent;
    TRY(upload_i32(ctx, &D.row, DH.row.data(), DH.row.size()));
    TRY(upload_i32(ctx, &D.row_seg, DH.row_seg.data(), DH.row_seg.size()));
    TRY(upload_i32(ctx, &D.seg_row, DH.seg_row.data(), DH.seg_row.size()));
    TRY(upload_i32(ctx, &D.seg_c0, DH.seg_c0.data(), DH.seg_c0.size()));
    TRY(upload_i32(ctx, &D.seg_len, DH.seg_len.data(), DH.seg_len.size()));
    TRY(upload_i32(ctx, &D.seg_ptr, DH.seg_ptr.data(), DH.seg_ptr.size()));
    TRY(upload_i32(ctx, &D.tile_id, DH.tile_id.data(), DH.tile_id.size()));
    TRY(upload_i32(ctx, &D.tile_ptr, DH.tile_ptr.data(), DH.tile_ptr.size()));
    TRY(upload_i32(ctx, &D.tile_seg, DH.tile_seg.data(), DH.tile_seg.size()));
    TRY(upload_i32(ctx, &D.perm, DH.perm.data(), DH.perm.size()));
    D.nchunks = (int)DH.ch_seg.size();
    TRY(upload_i32(ctx, &D.ch_seg, DH.ch_seg.data(), DH.ch_seg.size()));
    TRY(upload_i32(ctx, &D.ch_k0, DH.ch_k0.data(), DH.ch_k0.size()));
    TRY(upload_i32(ctx, &D.row_ch, DH.row_ch.data(), DH.row_ch.size()));
    TRY(dev_alloc(ctx, &D.ch_part, (size_t)D.nchunks + 8));
    TRY(dev_alloc(ctx, &D.val, (size_t)D.nent + 8));
    D.on = true;
    if (timing) fprintf(stderr, "[cuopt_amd setup]   dense: %d segments in %d rows, %lld of %lld nonzeros stored index-free\n", D.nseg, D.nrows, (long long)D.nent, (long long)ctx->nnz);
  }
  if (DH.on) TRY(dev_alloc(ctx, &ctx->dense.add_m, (size_t)m));
  std::vector<int32_t> rba = build_row_blocks(m, A_off);
  lap("row blocks A");
  ctx->a_nb = (int)rba.size() / 2 - 1;
  TRY(upload_i32(ctx, &ctx->a_rb, rba.data(), rba.size()));
  if ((int64_t)m + n >= 262144) {
    // the vectors below (24 of n (+ pad), 15 of m entries) out of one zero-filled allocation
    const size_t bytes = (24 * ((size_t)n + kSlicePad + 32) + 15 * ((size_t)m + 32)) * sizeof(double);
    if (hipMalloc((void**)&ctx->slab, bytes) == hipSuccess) {
      ctx->allocs.push_back(ctx->slab);
      ctx->slab_cap = bytes, ctx->slab_used = 0;
      HIP_TRY(hipMemsetAsync(ctx->slab, 0, bytes, ctx->stream));
    } else {
      ctx->slab = nullptr;
      (void)hipGetLastError();
    }
  }
  // every problem vector crosses PCIe once: the unscaled copy is made on the device, and a bound vector that is one value
  // throughout (all lower bounds 0, all upper bounds +inf: most LPs) is not uploaded at all
  lap("slab");
  ctx->note_uniform_bounds(lb, ub);
  lap("uniform bounds");
  auto upload_pair = [&](double** work, double** keep, const double* src, size_t count, bool uniform, double value) -> int {
    TRY(dev_alloc(ctx, work, count));
    TRY(dev_alloc(ctx, keep, count));
    if (count == 0) return 0;
    if (uniform) {
      k_fill<<<grid_for((int64_t)count), kBlock, 0, ctx->stream>>>((int64_t)count, *work, value);
    } else if (const double* on_device = an ? an->prefetched(src) : nullptr) {  // (sent ahead while the analysis ran: kernels_setup.hip)
      HIP_TRY(hipMemcpyAsync(*work, on_device, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(*work, src, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(*keep, *work, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  };
  TRY(upload_pair(&ctx->c, &ctx->c_u, c, (size_t)n, false, 0.0));
  TRY(upload_pair(&ctx->lb, &ctx->lb_u, lb, (size_t)n, ctx->ubd.lb_same != 0, ctx->ubd.lb));
  TRY(upload_pair(&ctx->ub, &ctx->ub_u, ub, (size_t)n, ctx->ubd.ub_same != 0, ctx->ubd.ub));
  TRY(upload_pair(&ctx->lo, &ctx->lo_u, lo, (size_t)m, false, 0.0));
  TRY(upload_pair(&ctx->hi, &ctx->hi_u, hi, (size_t)m, false, 0.0));
  lap("vector uploads");
  TRY(dev_alloc(ctx, &ctx->dr, m)); TRY(dev_alloc(ctx, &ctx->dc, n));
  // x, A^T y, xbar, sum_x carry kSlicePad spare entries: the sliced-primal dataflow of a sharded solve all-gathers them in
  // equal slices of a multiple of 16 entries per rank (slice * world may exceed n by up to 16 * 16 - 1)
  for (int i = 0; i < 2; ++i) {
    TRY(dev_alloc(ctx, &ctx->x[i], (size_t)n + kSlicePad)); TRY(dev_alloc(ctx, &ctx->y[i], m));
    TRY(dev_alloc(ctx, &ctx->aty[i], (size_t)n + kSlicePad)); TRY(dev_alloc(ctx, &ctx->rc[i], n));
  }
  TRY(dev_alloc(ctx, &ctx->xbar, (size_t)n + kSlicePad)); TRY(dev_alloc(ctx, &ctx->sumx, (size_t)n + kSlicePad)); TRY(dev_alloc(ctx, &ctx->sumy, m));
  TRY(dev_alloc(ctx, &ctx->avgx, n)); TRY(dev_alloc(ctx, &ctx->avgy, m));
  TRY(dev_alloc(ctx, &ctx->lrx, n)); TRY(dev_alloc(ctx, &ctx->lry, m));
  TRY(dev_alloc(ctx, &ctx->tmp_n, n)); TRY(dev_alloc(ctx, &ctx->tmp_m, m));
  for (int i = 0; i < 3; ++i) {
    TRY(dev_alloc(ctx, &ctx->ax_u[i], m));
    TRY(dev_alloc(ctx, &ctx->aty_u[i], n));
  }
  TRY(dev_alloc(ctx, &ctx->rc_scratch, n));
  const size_t slab_rest = ctx->slab_cap - ctx->slab_used;  // (kept for the n-sized buffers allocated after the layouts)
  ctx->slab_cap = ctx->slab_used;
  {
    // layout choice: CUOPT_AMD_SPMV_LAYOUT = auto (default) | stream | panel | jag ; CUOPT_AMD_TUNE=slab_bytes=...
    // auto is structural (reproducible): the jagged layout when filling its LDS column sets costs at most half of the gathers
    // they serve (build_jag), else slab-major panels iff the stream kernel's live gather set exceeds an XCD's L2
    // (gather_working_set), else the CSR stream.  "timed" times panels against the stream on the device (pick_layout).
    const char* mode_env = getenv("CUOPT_AMD_SPMV_LAYOUT");
    const std::string mode = mode_env ? mode_env : "auto";
    // 1.33 MiB of the gathered vector per slab: measured optimum on the 1e6 x 1e6 random LP (6 slabs: 71 us per
    // SpMV; 8 slabs of 1 MiB: 74 us; 4 slabs of 2 MiB: 75 us) -- fewer tiles per panel against L2 capacity
    const int64_t slab_bytes = std::max<int64_t>(64, cuopt_amd::tune_int("slab_bytes", 1398102));
    if (mode != "auto" && mode != "stream" && mode != "panel" && mode != "jag" && mode != "timed" && mode != "pb")
      return fail(-1, "CUOPT_AMD_SPMV_LAYOUT must be auto, stream, panel, jag, pb or timed");
    const bool force = mode == "panel";
    const bool timed = mode == "timed";
    // gather-free layout: on request, or (auto) where the panels would need more than their 16 slabs to keep a slab in L2
    auto want_pb = [&](int32_t cols) { return mode == "pb" || (mode == "auto" && (int64_t)cols * 8 > 16 * slab_bytes); };
    const bool try_jag = mode == "auto" || mode == "jag" || timed;
    // auto, not jagged: panels when the CSR stream kernel's live gather set overflows what an XCD's L2 keeps of it
    const int64_t ws_limit = cuopt_amd::tune_int("panel_ws_bytes", kPanelWorkingSetBytes);
    int64_t ws_at_device = 0, ws_a_device = -1;  // (counted by the main thread: the A^T side's worker must not allocate from the context)
    bool ws_at_on_host = false;
    if (an && (mode == "auto" || timed) && (int64_t)m * 8 > ws_limit) {
      const int rc = gather_working_set_device(ctx, ctx->at_idx, ctx->nnz, m, &ws_at_device);
      if (rc < 0) return rc;
      ws_at_on_host = rc == 1;  // (beyond ~2e7 rows the bitmap leaves the LDS: the four windows come to the host)
    }
    if (an && !DH.on && (mode == "auto" || timed) && (int64_t)n * 8 > ws_limit) {
      const int rc = gather_working_set_device(ctx, ctx->a_idx, ctx->nnz, n, &ws_a_device);
      if (rc < 0) return rc;
      if (rc == 1) ws_a_device = -1;
    }
    auto want_panels = [&](int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, const char* name) {
      if (force) return true;
      if (!timed && (mode != "auto" || (int64_t)cols * 8 <= ws_limit)) return false;
      if (timed && !getenv("CUOPT_AMD_TIMING")) return true;
      int64_t ws = 0;
      const bool a_side = name[1] == '\0';
      if (a_side && ws_a_device >= 0) {
        ws = ws_a_device;  // (same windows, counted on the device)
      } else if (a_side && !idx) {
        ws = gather_working_set(rows, cols, off, A_idx_host());
      } else if (idx) {
        ws = gather_working_set(rows, cols, off, idx);
      } else if (!ws_at_on_host) {  // (A^T of an analysed matrix: its indices live on the device; the same four windows were counted there)
        ws = ws_at_device;
      } else {
        std::vector<int32_t> sparse;
        std::vector<std::pair<int64_t, int64_t>> windows;
        if (analysis_fetch_idx_windows(an, 1, (int64_t)off[rows], &sparse, &windows) != 0) return false;
        ws = gather_working_set_windows(cols, sparse.data(), windows);
      }
      if (getenv("CUOPT_AMD_TIMING"))
        fprintf(stderr, "[cuopt_amd setup]   layout %-3s: live gather set of the stream kernel %.2f MiB (limit %.2f) -> %s\n", name,
                ws / 1048576.0, ws_limit / 1048576.0, ws > ws_limit ? "panels" : "stream");
      return timed || ws > ws_limit;
    };
    lap("row blocks + vectors");
    // The A^T side's HOST work (waiting for the caller's transposition, the hot CSR, the layouts' construction) runs on a thread of
    // its own next to the A side's construction and uploads; its uploads follow below, in the order they always had.
    struct TSide {
      std::thread worker;
      std::vector<int32_t> rbt;
      JagHost jat;
      PbHost hbt;
      PanelHost hat;
      bool want_pb_layout = false, want_dev_panels = false, want_dev_pb = false, panels_pending = false, want_dev_jag = false;
      ~TSide() { if (worker.joinable()) worker.join(); }
    } ts;
    const int32_t* T_off = at_offsets;
    const int32_t* T_idx = at_indices;
    // An analysed matrix's A^T lives on the device: its index array comes to the host only for the constructions that still run there
    // (jagged, gather-free, dense segments); the analysis' sampled estimate already says whether the jagged layout is worth a look.
    auto t_idx_host = [&]() -> const int32_t* { return an ? analysis_host_t_idx(an) : at_indices; };
    const bool skip_jag_a  = an && an->estimated && !an->permuted && mode != "jag" && an->saving_natural[0] < 0.35;
    const bool skip_jag_at = an && an->estimated && !an->permuted && mode != "jag" && an->saving_natural[1] < 0.35;
    // the gather-free layout is built on the device when the matrices are there (CUOPT_AMD_TUNE=pb_device=0: the host construction,
    // the tests' reference)
    const bool pb_on_device = an && !DH.on && cuopt_amd::tune_int("pb_device", 1) != 0;
    // ... and so is the jagged layout (CUOPT_AMD_TUNE=jag_device=0: on the host)
    const bool jag_on_device = an && !DH.on && cuopt_amd::tune_int("jag_device", 1) != 0;
    const auto w0 = std::chrono::steady_clock::now();
    auto wlap = [&](const char* what) {
      if (timing) fprintf(stderr, "[cuopt_amd setup]   A^T thread: %-22s at %6.2f ms\n", what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count());
    };
    // stage 2 of the A^T side: which layout when it is not the jagged one (the gather-free layout / the panels, built below)
    auto at_side_rest = [&] {
      const bool jat_ok = ts.jat.ok || ctx->jat.on;
      if (!jat_ok && want_pb(m) && (mode == "pb" || want_panels(n, m, T_off, T_idx, "A^T"))) {
        ts.want_pb_layout = true;
        if (pb_on_device) {
          ts.want_dev_pb = true;  // (built on the device by the main thread, below; whether panels are wanted instead is known after that)
          ts.panels_pending = mode != "stream" && mode != "jag" && mode != "pb";
          wlap("layouts");
          return;
        }
        if (!T_idx) T_idx = t_idx_host();
        ts.hbt            = build_pb(n, m, T_off, T_idx, ctx->cus, mode == "pb");
      }
      if (mode != "stream" && mode != "jag" && mode != "pb" && !jat_ok && !ts.hbt.ok && want_panels(n, m, T_off, T_idx, "A^T")) {
        if (an && !DH.on) ts.want_dev_panels = true;  // (built on the device by the main thread, below)
        else ts.hat = build_panels(n, m, T_off, T_idx, slab_bytes, force || !timed);
      }
      wlap("layouts");
    };
    auto at_side = [&] {
      if (transpose_ready) transpose_ready(user);
      if ((int64_t)at_offsets[n] != ctx->nnz) return;  // (reported below)
      lat = long_rows(n, at_offsets);
      wlap("long rows");
      if (DH.on) {
        strip_transpose(DH, &DH, n, at_offsets, t_idx_host());
        hT_off.swap(DH.st_off), hT_idx.swap(DH.st_idx), hT_perm.swap(DH.st_perm);
      }
      if (!hT_off.empty()) T_off = hT_off.data(), T_idx = hT_idx.data();
      ts.rbt = build_row_blocks(n, T_off);
      wlap("row blocks");
      if (try_jag && !skip_jag_at) {
        if (jag_on_device) {
          ts.want_dev_jag = true;  // (built on the device by the main thread, below; the rest of this side's decisions follow it)
          return;
        }
        if (!T_idx) T_idx = t_idx_host();
        ts.jat = build_jag(n, m, T_off, T_idx, mode == "jag" ? 1 : 0, ctx->cus);
      } else if (an) {
        ts.jat.saving = an->saving_natural[1];
      }
      at_side_rest();
    };
    // (an analysed matrix: nothing to wait for and little left to do on the host -- the A^T side runs inline, behind the A side;
    // a thread of its own took 3 ms to do 0.5 ms of work next to the main thread's HIP calls)
    const bool at_thread = !an || (try_jag && !skip_jag_at && !jag_on_device) || DH.on || (want_pb(m) && !pb_on_device);  // (host constructions worth a thread)
    if (at_thread) ts.worker = std::thread(at_side);
    if (try_jag) {
      JagHost ja;
      int on_device = 1;
      if (skip_jag_a) {
        ja.saving = an->saving_natural[0];
      } else if (jag_on_device) {
        on_device = build_jag_device(ctx, &ctx->ja, m, n, A_off, ctx->ha_off, ctx->ha_idx, ctx->ha_val, mode == "jag" ? 1 : 0, ctx->cus);
        if (on_device < 0) return on_device;
        lap("jag A on the device");
      }
      if (on_device == 1) {
        if (!skip_jag_a) ja = build_jag(m, n, A_off, A_idx_host(), mode == "jag" ? 1 : 0, ctx->cus);
        lap("build_jag A");
        TRY(upload_jag(ctx, &ctx->ja, ja, ctx->ha_off, ctx->ha_idx, ctx->ha_val));
        lap("upload jag A");
      }
    }
    if (!ctx->ja.on && want_pb(n) && (mode == "pb" || want_panels(m, n, A_off, A_idx, "A"))) {
      int on_device = 1;
      if (pb_on_device) {
        std::string why;
        on_device = build_pb_device(ctx, &ctx->pba, m, n, A_off, ctx->ha_off, ctx->ha_idx, ctx->cus, mode == "pb", &why);
        if (on_device < 0) return on_device;
        lap("pb A on the device");
        if (on_device == 0 && !ctx->pba.on && mode == "pb") return fail(-1, "CUOPT_AMD_SPMV_LAYOUT=pb: A does not fit the gather-free layout (%s)", why.c_str());
      }
      if (on_device == 1) {
        PbHost hb = build_pb(m, n, A_off, A_idx_host(), ctx->cus, mode == "pb");
        lap("build_pb A");
        if (!hb.ok && mode == "pb") return fail(-1, "CUOPT_AMD_SPMV_LAYOUT=pb: A does not fit the gather-free layout (%s)", hb.why.c_str());
        TRY(upload_pb(ctx, &ctx->pba, hb));
        lap("upload pb A");
      }
    }
    if (mode != "stream" && mode != "jag" && mode != "pb" && !ctx->ja.on && !ctx->pba.on && want_panels(m, n, A_off, A_idx, "A")) {
      PanelHost ha;
      int on_device = an && !DH.on ? build_panels_device(ctx, &ctx->pa, m, n, A_off, ctx->ha_off, ctx->ha_idx, ctx->ha_val, slab_bytes, force || !timed) : 1;
      if (on_device < 0) return on_device;
      if (on_device == 1) {
        ha = build_panels(m, n, A_off, A_idx_host(), slab_bytes, force || !timed, DH.on ? &DH.first_seg : nullptr);
        lap("build_panels A");
        TRY(upload_panels(ctx, &ctx->pa, ha, ctx->ha_off, ctx->ha_idx, ctx->ha_val));
      }
      if (ctx->pa.on && DH.on) {
        // segments of the own rows, in own-row order (the rows that own segments are a subset of the own rows and both lists ascend)
        std::vector<int32_t> own_seg(ha.own_row.size() + 1, 0);
        for (size_t i = 0; i < ha.own_row.size(); ++i) {
          const int32_t f = DH.first_seg[ha.own_row[i]];
          int32_t cnt     = 0;
          for (int32_t q = f; f >= 0 && q < (int32_t)DH.seg_row.size() && DH.seg_row[q] == ha.own_row[i]; ++q) ++cnt;
          own_seg[i + 1] = own_seg[i] + cnt;
        }
        int32_t* d_own_seg = nullptr;
        TRY(upload_i32(ctx, &d_own_seg, own_seg.data(), own_seg.size()));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        PanelView& v = ctx->pa.v;
        v.dn_own_seg = d_own_seg, v.dn_seg_c0 = ctx->dense.seg_c0, v.dn_seg_len = ctx->dense.seg_len, v.dn_seg_ptr = ctx->dense.seg_ptr;
        v.dn_seg_row = ctx->dense.seg_row, v.dn_val = ctx->dense.val;
      }
      lap("upload panels A");
      HIP_TRY(hipStreamSynchronize(ctx->stream));  // the staged copies have left the host arrays (back to the pool)
    }
    if (!at_thread) at_side();
    else ts.worker.join();
    lap("wait for the A^T side");
    if ((int64_t)at_offsets[n] != ctx->nnz) return fail(-1, "pdlpdev_create: A and A^T disagree on nnz");
    if (!an) {
      TRY(upload_i32(ctx, &ctx->at_off, at_offsets, (size_t)n + 1));
      TRY(upload_i32(ctx, &ctx->at_idx, at_indices, nnz, 8));
      TRY(upload_f64(ctx, &ctx->at_val, at_values, nnz, 8));
    }
    ctx->at_nlong = (int)lat.size();
    if (ctx->at_nlong) TRY(upload_i32(ctx, &ctx->at_long, lat.data(), lat.size()));
    const bool hot_t = !hT_off.empty();
    ctx->hat_off = ctx->at_off, ctx->hat_idx = ctx->at_idx, ctx->hat_val = ctx->at_val;
    ctx->hot_nnz_at = (int64_t)T_off[n];
    if (hot_t) {
      TRY(upload_i32(ctx, &ctx->hat_off, T_off, (size_t)n + 1));
      TRY(upload_i32(ctx, &ctx->hat_idx, T_idx, (size_t)ctx->hot_nnz_at, 8));
      TRY(dev_alloc(ctx, &ctx->hat_val, (size_t)ctx->hot_nnz_at + 8));
      TRY(upload_i32(ctx, &ctx->dense.s_perm_at, hT_perm.data(), hT_perm.size()));
    }
    if (DH.on) TRY(dev_alloc(ctx, &ctx->dense.add_n, (size_t)n));
    ctx->at_nb = (int)ts.rbt.size() / 2 - 1;
    TRY(upload_i32(ctx, &ctx->at_rb, ts.rbt.data(), ts.rbt.size()));
    lap("upload A^T");
    bool jat_on_device = false;
    if (ts.want_dev_jag) {
      int on_device = build_jag_device(ctx, &ctx->jat, n, m, T_off, ctx->hat_off, ctx->hat_idx, ctx->hat_val, mode == "jag" ? 1 : 0, ctx->cus);
      if (on_device < 0) return on_device;
      if (on_device == 1) {
        if (!T_idx) T_idx = t_idx_host();
        ts.jat = build_jag(n, m, T_off, T_idx, mode == "jag" ? 1 : 0, ctx->cus);
      } else {
        jat_on_device = true;
      }
      lap("jag At on the device");
    }
    if (try_jag && !jat_on_device) {
      TRY(upload_jag(ctx, &ctx->jat, ts.jat, ctx->hat_off, ctx->hat_idx, ctx->hat_val));
      lap("upload jag At");
    }
    if (ts.want_dev_jag) at_side_rest();  // (the decisions that waited for the jagged layout's verdict)
    if (ts.want_dev_pb) {
      std::string why;
      int on_device = build_pb_device(ctx, &ctx->pbat, n, m, T_off, ctx->hat_off, ctx->hat_idx, ctx->cus, mode == "pb", &why);
      if (on_device < 0) return on_device;
      if (on_device == 1) {
        if (!T_idx) T_idx = t_idx_host();
        ts.hbt = build_pb(n, m, T_off, T_idx, ctx->cus, mode == "pb");
      } else {
        ts.want_pb_layout = false;  // (nothing to upload)
        if (!ctx->pbat.on && mode == "pb") return fail(-1, "CUOPT_AMD_SPMV_LAYOUT=pb: A^T does not fit the gather-free layout (%s)", why.c_str());
      }
      lap("pb At on the device");
      if (ts.panels_pending && !ctx->pbat.on && !ts.hbt.ok && want_panels(n, m, T_off, T_idx, "A^T")) ts.want_dev_panels = true;
    }
    if (ts.want_pb_layout) {
      if (!ts.hbt.ok && mode == "pb") return fail(-1, "CUOPT_AMD_SPMV_LAYOUT=pb: A^T does not fit the gather-free layout (%s)", ts.hbt.why.c_str());
      TRY(upload_pb(ctx, &ctx->pbat, ts.hbt));
      lap("upload pb At");
    }
    if (ts.want_dev_panels) {
      const int on_device = build_panels_device(ctx, &ctx->pat, n, m, T_off, ctx->hat_off, ctx->hat_idx, ctx->hat_val, slab_bytes, force || !timed);
      if (on_device < 0) return on_device;
      if (on_device == 1) ts.hat = build_panels(n, m, T_off, t_idx_host(), slab_bytes, force || !timed);
      lap("panels At on the device");
    }
    if (ts.hat.ok) {
      TRY(upload_panels(ctx, &ctx->pat, ts.hat, ctx->hat_off, ctx->hat_idx, ctx->hat_val));
      if (ctx->pat.on && DH.on) {
        // per panel of the column side: the segments that reach into its column range, ascending rows (= segment numbers)
        const std::vector<int32_t>& row0 = ts.hat.row0;
        std::vector<int32_t> pan_ptr(row0.size(), 0), pan_seg;
        bool fits = true;
        for (size_t w = 0; w + 1 < row0.size(); ++w) {
          for (size_t q = 0; q < DH.seg_row.size(); ++q)
            if (DH.seg_c0[q] < row0[w + 1] && DH.seg_c0[q] + DH.seg_len[q] > row0[w]) pan_seg.push_back((int32_t)q);
          pan_ptr[w + 1] = (int32_t)pan_seg.size();
          fits           = fits && pan_ptr[w + 1] - pan_ptr[w] <= kPanelDenseSegs;
        }
        if (fits && ts.hat.own_row.empty()) {  // (else: k_dense_cols in front of the panels, as for the other layouts)
          int32_t *d_ptr = nullptr, *d_seg = nullptr;
          TRY(upload_i32(ctx, &d_ptr, pan_ptr.data(), pan_ptr.size()));
          TRY(upload_i32(ctx, &d_seg, pan_seg.data(), pan_seg.size()));
          HIP_TRY(hipStreamSynchronize(ctx->stream));
          PanelView& v = ctx->pat.v;
          v.dn_pan_ptr = d_ptr, v.dn_pan_seg = d_seg;
          v.dn_seg_c0 = ctx->dense.seg_c0, v.dn_seg_len = ctx->dense.seg_len, v.dn_seg_ptr = ctx->dense.seg_ptr;
          v.dn_seg_row = ctx->dense.seg_row, v.dn_val = ctx->dense.val;
        }
      }
      lap("upload panels At");
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the staged copies have left the worker's host arrays
  }
  {
    // small LPs: a whole batch of attempts inside one workgroup (CUOPT_AMD_SMALL=0 switches it off)
    const char* small_env = getenv("CUOPT_AMD_SMALL");
    const int tier        = resident_tier(m, n, ctx->nnz);
    ctx->small_resident   = tier >= 0 && !(small_env && atoi(small_env) == 0) && !ctx->dense.add_m && !ctx->dense.add_n;
    if (small_env && atoi(small_env) != 0 && tier < 0)
      return fail(-1, "CUOPT_AMD_SMALL=1: the LP does not fit the resident kernel (m, n <= 2048, nnz <= 4096 ...)");
  }
  // every layout adds what the dense segments / the extracted long rows contribute ahead of its epilogue (null: nothing to add)
  ctx->pa.v.dense_add = ctx->ja.v.dense_add = ctx->pba.v.dense_add = ctx->dense.add_m;
  ctx->pat.v.dense_add = ctx->jat.v.dense_add = ctx->pbat.v.dense_add = ctx->dense.add_n;
  // ... except the panels, whose kernels add the segments themselves (own-row workgroups / the column epilogue): no launch in front
  if (ctx->pa.v.dn_own_seg) ctx->pa.v.dense_add = nullptr;
  if (ctx->pat.v.dn_pan_ptr) ctx->pat.v.dense_add = nullptr;
  ctx->slab_cap += slab_rest;
  TRY(dev_alloc(ctx, &ctx->part_a, (size_t)8 * std::max({ctx->a_nb, ctx->pba.on ? ctx->pba.v.B : 0, ctx->pa.on ? ctx->pa.v.W : 0, ctx->ja.on ? ctx->ja.v.nblk + ctx->ja.v.nlong : 0, 1})));
  TRY(dev_alloc(ctx, &ctx->part_at, (size_t)8 * std::max({ctx->at_nb, ctx->pbat.on ? ctx->pbat.v.B : 0, ctx->pat.on ? ctx->pat.v.W : 0, ctx->jat.on ? ctx->jat.v.nblk + ctx->jat.v.nlong : 0, 1})));
  TRY(dev_alloc(ctx, &ctx->part_g, (size_t)8 * 2048));
  TRY(dev_alloc(ctx, &ctx->scal, kScalars));
  TRY(dev_alloc(ctx, &ctx->ctl, 1));
  TRY(dev_alloc(ctx, &ctx->ar_buf, (size_t)n + kSlicePad));
  lap("partial buffers");
  k_fill<<<grid_for(m), kBlock, 0, ctx->stream>>>(m, ctx->dr, 1.0);
  k_fill<<<grid_for(n), kBlock, 0, ctx->stream>>>(n, ctx->dc, 1.0);
  HIP_TRY(hipGetLastError());
  lap("fill D");
  TRY(sync_panel_values(ctx));
  lap("panel values (permute)");
  {
    const char* mode_env = getenv("CUOPT_AMD_SPMV_LAYOUT");
    if (mode_env && std::string(mode_env) == "timed") {
      TRY(pick_layout(ctx, &ctx->pa, m, ctx->a_nb, ctx->a_rb, ctx->ha_off, ctx->ha_idx, ctx->ha_val, ctx->tmp_n, ctx->tmp_m, "A"));
      lap("layout autotune A");
      TRY(pick_layout(ctx, &ctx->pat, n, ctx->at_nb, ctx->at_rb, ctx->hat_off, ctx->hat_idx, ctx->hat_val, ctx->tmp_m, ctx->tmp_n, "A^T"));
      lap("layout autotune");
    }
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  scratch_free(ctx);  // (the layout constructions' temporaries: both sides are built)
  return 0;
}

extern "C" {

void pdlpdev_destroy(pdlpdev_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  scratch_free(ctx);
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
  if (ctx->shared_with_parent && ctx->parent) ctx->parent->clones_alive -= 1;
  if (ctx->batches_alive > 0)  // (a batch holds plain pointers to its members: destroy it first)
    fprintf(stderr, "[cuopt_amd] pdlpdev_destroy: %d batch(es) still hold this context -- destroy the batch before its members\n", ctx->batches_alive);
  if (ctx->clones_alive > 0)  // (a contract of pdlpdev_clone_shared; said aloud, since what follows frees the clones' matrices)
    fprintf(stderr, "[cuopt_amd] pdlpdev_destroy: %d clone(s) of this context are still alive -- they alias its matrices and must be destroyed first\n", ctx->clones_alive);
  for (void* mapped : ctx->p2p.opened) (void)hipIpcCloseMemHandle(mapped);
  if (ctx->p2p.base) (void)hipFree(ctx->p2p.base);
  if (ctx->comm && !ctx->soft) comm_cache::release(ctx->comm_key);
  for (void* p : ctx->allocs) (void)hipFree(p);
  const bool whole = ctx->stream && ctx->scal_h && ctx->first_chunk && !ctx->shared_with_parent && !ctx->stream_borrowed;
  if (!(whole && give_recycled(Recycled{ctx->device, ctx->stream, ctx->scal_h, ctx->first_chunk}))) {
    if (ctx->first_chunk) (void)hipFree(ctx->first_chunk);
    if (ctx->scal_h) (void)hipHostFree(ctx->scal_h);  // ctl_h lives in the same block
    if (ctx->stream && !ctx->shared_with_parent && !ctx->stream_borrowed) (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

}  // extern "C"
