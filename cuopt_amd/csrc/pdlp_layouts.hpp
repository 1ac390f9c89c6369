// Host side of the SpMV layouts: what a layout's construction hands to the core (structure only; values are permuted on the device)
// and the entry points of each layout's translation unit (kernels_<layout>.hip holds its kernels AND its host-side construction).
#pragma once
#include "pdlp_ctx.hpp"

// ---- slab-major row panels: host-side construction (structure only; values are permuted on the device)
struct PanelHost {
  bool ok = false, any_long = false;
  bool seg = false;                        // long-tail variant: packed (row, column) entries, no row pointers (panel_seg_block)
  int32_t slab_w = 0;
  int W = 0, S = 0;                        // W: panels only
  std::vector<int32_t> own_row, own_ptr;   // rows of more than kPanelOwnRow nonzeros (a workgroup each, behind the panels); per panel
  std::vector<int32_t> row0, tile_ptr;
  std::vector<int64_t> rp_base;
  // the three big arrays are deliberately NOT zero-filled (every entry is written by pass 2)
  cuopt_amd::PoolArray<int32_t> perm, col;
  cuopt_amd::PoolArray<uint16_t> rowptr;
  size_t nnz = 0, rowptr_size = 0;
};

// Bytes of the gathered vector that the CSR stream kernel keeps live in ONE XCD's L2: an XCD owns a contiguous range of
// row blocks and has 32 CUs x 8 workgroups x 2048 nonzeros = 512 K nonzeros of consecutive rows in flight, so what it
// re-reads from L2 is the set of 128-byte lines those rows touch.  Estimated on up to four evenly spaced windows of that
// size (exact when the matrix is smaller), mean over the windows.  Structural and reproducible: this, not a timing, is
// what 'auto' decides on.
constexpr int64_t kPanelWorkingSetBytes = 4 * (int64_t)1048576;  // an XCD's L2; calibration: profiles/r02_layout_rule.txt

// ---- sorted jagged rows: host-side construction (structure only; values are permuted on the device) -------------
struct JagHost {
  bool ok = false;
  int rows = 0, waves = 8, ngroups = 0, nblk = 0;
  std::vector<int32_t> row0, tile_e, tile_sr, win, set_ptr, set_col, lr_ptr, lr_row;
  cuopt_amd::PoolArray<uint32_t> sr;
  cuopt_amd::PoolArray<uint16_t> slot;
  cuopt_amd::PoolArray<int32_t> perm;
  size_t nsr = 0, nent = 0;
  double saving = 0.0;  // share of the global gathers the LDS column sets save: 1 - (cost of filling the sets) / nonzeros
};

// ---- gather-free layout: host-side construction (structure only; values are permuted on the device) -----------------
// Parallel over bins on the host pool's threads; every pass is O(nnz).  Geometry: panels of 8192 columns (16384 when the
// gathered vector has more than 2 M entries: fewer, longer chunks), pieces of 8 entries unless the chunks (nnz / (panels x bins))
// are shorter than 24 entries (then 4: at 1e8 nonzeros -- chunks of 13 -- 8-entry pieces pad by 33 % and run 20 % slower), bins filled to kPbCap padded entries (found by iterating on the per-bin nonzero target:
// the padding of a bin depends on how its entries spread over the panels).
struct PbHost {
  bool ok = false;
  std::string why;
  int rows = 0, cols = 0, S = 0, B = 0, gshift = 3, panel_shift = 13, p_threads = 512;
  int64_t np = 0, nnz = 0;
  std::vector<int32_t> bin_row0, bin_e0, wg_e0, wg_panel, bin_grp, grp_pos;
  cuopt_amd::PoolArray<int32_t> perm, piece_dst;
  cuopt_amd::PoolArray<uint16_t> lidx, pos;
  cuopt_amd::PoolArray<uint32_t> sr;
  // wide bins (build_pb_wide): bins of kPbwRows rows, 16-entry pieces, a slot word per image slot and a level per step instead of
  // sr / bin_grp / grp_pos / pos
  bool wide = false;
  cuopt_amd::PoolArray<uint16_t> rib;
  std::vector<uint8_t> step_lv;
  std::vector<int32_t> ser_ptr, ser_row, ser_eptr, ser_slot;  // serial rows (PbView)
};

// ---- dense row segments: detection and the sparse remainder (host) -----------------------------------------------------------
constexpr int kDenseMin = 256;  // consecutive columns of one row from which index-free storage is used

struct DenseHost {
  bool on = false;
  std::vector<int32_t> row, row_seg, seg_row, seg_c0, seg_len, seg_ptr, tile_id, tile_ptr, tile_seg, tile_slot, perm;
  std::vector<int32_t> s_off, s_idx, s_perm;     // A without the segments' entries (+ where each entry sits in the full CSR)
  std::vector<int32_t> st_off, st_idx, st_perm;  // A^T likewise
  std::vector<int32_t> first_seg;                // per row of A: first segment (seg_row ascending), -1 none
  std::vector<int32_t> ch_seg, ch_k0, row_ch;    // chunks of <= kDenseChunk entries, per owning row
  int64_t nent = 0;
};

inline int upload_i32(pdlpdev_ctx* c, int32_t** dst, const int32_t* src, size_t count, size_t pad = 0)
{
  TRY(dev_alloc(c, dst, count + pad));
  if (count) HIP_TRY(hipMemcpyAsync(*dst, src, count * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  return 0;
}

inline int upload_f64(pdlpdev_ctx* c, double** dst, const double* src, size_t count, size_t pad = 0)
{
  TRY(dev_alloc(c, dst, count + pad));
  if (count && src)
    HIP_TRY(hipMemcpyAsync(*dst, src, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}

// ---- defined in kernels_<layout>.hip -----------------------------------------------------------------------------------------
std::vector<int32_t> build_row_blocks(int32_t rows, const int32_t* off);
int64_t gather_working_set(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx);
int64_t gather_working_set_windows(int32_t cols, const int32_t* sparse, const std::vector<std::pair<int64_t, int64_t>>& windows);
PanelHost build_panels(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx,
                              int64_t slab_bytes, bool force, const std::vector<int32_t>* dense_first_seg = nullptr);
int upload_panels(pdlpdev_ctx* c, pdlpdev_ctx::Panels* dst, const PanelHost& h, const int32_t* d_off, const int32_t* d_idx,
                         const double* d_val);
int pick_layout(pdlpdev_ctx* c, pdlpdev_ctx::Panels* pn, int rows, int nb, const int32_t* rb, const int32_t* off,
                       const int32_t* idx, const double* val, const double* vec, double* out, const char* name);
JagHost build_jag(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int mode, int cus);
int upload_jag(pdlpdev_ctx* c, pdlpdev_ctx::Jag* dst, const JagHost& h, const int32_t* d_off, const int32_t* d_idx,
                      const double* d_val);
PbHost build_pb(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int cus, bool forced);
PbHost build_pb_wide(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, int cus, bool forced);
int pbw_piece_shift(int64_t nnz, int S, int B);  // 4 (16-entry pieces) or 3
bool pb_wants_wide(int32_t cols);  // the geometry 'auto' takes for this many gathered columns (CUOPT_AMD_TUNE=pb_wide=0/1 overrides)
int upload_pb(pdlpdev_ctx* c, pdlpdev_ctx::Pb* dst, const PbHost& h);
void find_dense_segments(int32_t m, int32_t n, const int32_t* off, const int32_t* idx, DenseHost* D);
void strip_transpose(const DenseHost& Din, DenseHost* D, int32_t n, const int32_t* t_off, const int32_t* t_idx);

// ---- the resident small-LP path (kernels_resident.hip) ----------------------------------------------------------------------------
int resident_tier(int m, int n, int64_t nnz);
int resident_run(pdlpdev_ctx* ctx, int32_t target_steps);  // attempts inside one workgroup until the target (pdlpdev_run's small branch)
int resident_major_eval(pdlpdev_ctx* ctx, int average_mode, int rc_rule_finite_bounds, int want_linf, double eps_rel_primal, double eps_rel_dual);
// the nine scalars an evaluation leaves (device or pinned) -> the PDLPDEV_EV_* array
inline void read_eval(const double* h, bool want_linf, double out[PDLPDEV_EV_COUNT])
{
  out[PDLPDEV_EV_PRES2]         = h[0];
  out[PDLPDEV_EV_DUAL_SUM]      = h[1] + h[5];
  out[PDLPDEV_EV_Y2]            = h[2];
  out[PDLPDEV_EV_LINF_PRES_REL] = want_linf ? h[3] : 0.0;
  out[PDLPDEV_EV_DRES2]         = h[4];
  out[PDLPDEV_EV_CX]            = h[6];
  out[PDLPDEV_EV_X2]            = h[7];
  out[PDLPDEV_EV_LINF_DRES_REL] = want_linf ? h[8] : 0.0;
}

// A^T y of the current iterate for SEVERAL stream-layout contexts in one launch (kernels_stream.hip): block i of the grid is row block
// blk[i].y of context blk[i].x -- k_spmv_at_cur's body, so the sums are the single launch's
struct StreamAtCurArgs {
  int nb;
  const int32_t *rb, *off, *idx;
  const double* val;
  const pdlpdev_ctl* ctl;
  const double *y0, *y1;
  double *aty0, *aty1;
};
int launch_stream_at_cur_batch(hipStream_t s, const StreamAtCurArgs* args, const int2* blk, int blocks);
