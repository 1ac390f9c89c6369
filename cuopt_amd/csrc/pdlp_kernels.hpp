// gfx950 device code of the PDLP hot path: wave64 reductions, the LDS-staged CSR "stream" SpMV
// skeleton and its fused epilogues.  Included only by pdlp_device.hip.
//
// Rounding contract (shared with oracle/pdlp_oracle.c so that element-wise results and short-row
// SpMV sums are bit-identical): compiled with -ffp-contract=off (no FMA contraction); a row sum is
// accumulated left-to-right in CSR order starting from 0.0 whenever the row has at most
// LONG_ROW nonzeros; longer rows and all dot-product style reductions use a fixed (launch-geometry
// determined, run-to-run reproducible) tree and are compared with a tolerance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>


#include "cuopt_amd/pdlp_device.h"

namespace pdlp {

typedef double vec4d __attribute__((ext_vector_type(4)));
typedef int vec4i __attribute__((ext_vector_type(4)));

constexpr int kBlock    = 256;   // threads per workgroup = 4 wave64
constexpr int kNnzTile  = 2048;  // LDS product slots per workgroup (16 KiB)
constexpr int kNnzBlock = kNnzTile - 4;  // nonzeros per row block: the tile starts at k0 rounded down to 4
constexpr int kLongRow  = 128;   // rows above this are reduced cooperatively, not by one lane
constexpr int kMaxRowsPerBlock = 1024;

// ------------------------------------------------------------------------------------------------
// wave64 / workgroup reductions.  Intra-32 butterflies use ds_swizzle (bit-mask mode: no LDS
// memory traffic, no address VGPR); the 32<->32 exchange is one ds_bpermute.
// ------------------------------------------------------------------------------------------------
template <int XOR_MASK>
__device__ __forceinline__ double swizzle_xor(double v)
{
  static_assert(XOR_MASK >= 1 && XOR_MASK <= 16, "ds_swizzle works inside 32 lanes");
  constexpr int pattern = (XOR_MASK << 10) | 0x1F;  // and=0x1f, or=0, xor=XOR_MASK
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, pattern);
  hi = __builtin_amdgcn_ds_swizzle(hi, pattern);
  return __hiloint2double(hi, lo);
}

struct SumOp {
  static __device__ __forceinline__ double identity() { return 0.0; }
  static __device__ __forceinline__ double apply(double a, double b) { return a + b; }
};
struct MaxOp {
  static __device__ __forceinline__ double identity() { return 0.0; }  // all our maxima are >= 0
  static __device__ __forceinline__ double apply(double a, double b) { return a > b ? a : b; }
};

// every lane ends with the reduction over the 64 lanes; the combination tree is fixed
template <class Op>
__device__ __forceinline__ double wave_reduce(double v)
{
  v = Op::apply(v, swizzle_xor<1>(v));
  v = Op::apply(v, swizzle_xor<2>(v));
  v = Op::apply(v, swizzle_xor<4>(v));
  v = Op::apply(v, swizzle_xor<8>(v));
  v = Op::apply(v, swizzle_xor<16>(v));
  v = Op::apply(v, __shfl_xor(v, 32, 64));
  return v;
}

// result valid in thread 0 (and broadcast through `scratch[0..NQ)` after the trailing barrier)
template <class Op, int NQ, int WAVES = kBlock / 64>
__device__ __forceinline__ void block_reduce(double (&v)[NQ], double* scratch /* >= WAVES*NQ */)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v[q] = wave_reduce<Op>(v[q]);
    if (lane == 0) scratch[wave * NQ + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double acc = scratch[q];
      for (int w = 1; w < WAVES; ++w) acc = Op::apply(acc, scratch[w * NQ + q]);
      v[q] = acc;
    }
  }
}

// ---- latency-optimised sum for single-workgroup loops (resident small-LP path) ---------------------
// DPP lane permutes run at VALU latency (a few cycles) where ds_swizzle / ds_bpermute pay an LDS round trip.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// every lane ends with the sum over its 16-lane DPP row; fixed tree: xor 1, xor 2, half mirror, mirror
__device__ __forceinline__ double row16_sum(double v)
{
  v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v = v + dpp_move<0x141>(v);  // row_half_mirror: the other quad of the 8
  v = v + dpp_move<0x140>(v);  // row_mirror: the other 8 of the 16
  return v;
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v)  // wave-uniform result
{
  v = row16_sum(v);
  return ((read_lane(v, 0) + read_lane(v, 16)) + read_lane(v, 32)) + read_lane(v, 48);
}
// Sum of NQ (<= 4) quantities over WAVES (<= 16) waves; EVERY lane of every wave ends with the totals (each wave
// repeats the cheap cross-wave step), so a uniform decision can follow without a second barrier.
// One barrier; the cross-wave step is one 16-lane row reduction per quantity instead of a serial loop.
template <int NQ, int WAVES>
__device__ __forceinline__ void block_sum_fast(double (&v)[NQ], double* scratch /* >= 16*NQ */)
{
  static_assert(NQ <= 4 && WAVES <= 16, "one DPP row per quantity");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double w = wave_sum_dpp(v[q]);
    if (lane == 0) scratch[q * 16 + wave] = w;
  }
  __syncthreads();
  const int q = lane >> 4, i = lane & 15;
  double part = (q < NQ && i < WAVES) ? scratch[q * 16 + i] : 0.0;
  part        = row16_sum(part);
#pragma unroll
  for (int k = 0; k < NQ; ++k) v[k] = read_lane(part, 16 * k);
}

// Dense row segments (index-free storage, pdlp_device.hip "dense"): the layouts below multiply the SPARSE remainder of the matrix;
// what the dense segments contribute to row r was computed just before and is added here, ahead of the fused epilogue.
__device__ __forceinline__ double dense_plus(const double* __restrict__ add, int r, double v) { return add ? v + add[r] : v; }

__device__ __forceinline__ bool loop_active(const pdlpdev_ctl* ctl)
{
  return ctl->error == 0 && ctl->steps_taken < ctl->target_steps;
}

// block b executes on XCD (b % 8) (observed dispatch order, speed only): give each XCD one
// contiguous range of row blocks so that neighbouring rows -- which in real LPs touch neighbouring
// columns -- share that XCD's private 4 MiB L2 for the gathered vector.
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
  const int per = (nb + 7) >> 3;
  const int v   = (b & 7) * per + (b >> 3);
  return v;  // may be >= nb for the ragged tail: caller skips
}

// (the stream skeleton: spmv_stream.hpp)


// ------------------------------------------------------------------------------------------------
// Slab-major row panels: the SpMV used when the gathered vector does not fit an XCD's 4 MiB L2.
// Measured on the 1e6 x 1e6 random LP (profiles/r01_spmv_tuning_table.txt): the CSR stream kernel is
// bound by the gather (48 % L2 hit rate, ~4.3x the algorithmic bytes fetched from the fabric); walking
// the columns in L2-sized slabs takes the SpMV from 105 us to ~75 us.
//   * workgroup w owns the contiguous row panel [row0[w], row0[w+1]) (balanced by nonzeros,
//     <= kPanelMaxRows rows so that one partial sum per row lives in LDS);
//   * the panel's nonzeros are stored slab by slab (slab = column range of ~1 MiB of the gathered
//     vector), rows ascending inside a slab, CSR order inside a (row, slab) cell -> with sorted CSR
//     columns every row is still summed strictly left to right (bit-identical to the oracle);
//   * every workgroup walks the slabs in the same order at the same pace (equal work per panel), so at
//     any moment an XCD's gathers fall into one slab that its L2 holds; no synchronisation is needed
//     for correctness;
//   * rowptr is 16-bit, relative to the tile start (tiles hold < 65536 nonzeros by construction).
// ------------------------------------------------------------------------------------------------
constexpr int kPanelThreads = 512;
constexpr int kPanelChunk   = 4096;   // nonzeros staged per pass (32 KiB of products)
constexpr int kPanelMaxRows = 3584;   // 28 KiB of per-row partial sums
constexpr int kPanelWaves   = kPanelThreads / 64;

struct PanelView {
  int W, S;
  int any_long;  // some row has more than kLongRow nonzeros: the row-sum phase takes the variant that shares such rows
  const int32_t* __restrict__ row0;      // W+1
  const int32_t* __restrict__ tile_ptr;  // W*S+1, positions in the permuted nonzero arrays
  const uint16_t* __restrict__ rowptr;   // per tile: (rows_w + 1) offsets relative to the tile start
  const int64_t* __restrict__ rp_base;   // W*S: where each tile's rowptr starts
  const int32_t* __restrict__ col;       // permuted column indices
  const double* __restrict__ val;        // permuted values
  const double* __restrict__ dense_add = nullptr;  // per row: what the dense segments contribute (see dense_plus)
  // Long-tail variant (panel_seg_block below): `col` holds (row within the panel) << kSegColBits | (column - slab * slab_w), the entries of a
  // chunk are stored lane-major (lane t's consecutive entries one per round), and there are no row pointers
  int seg = 0, slab_w = 0;
  // Rows of more than kPanelOwnRow nonzeros are not in the panels: each gets a workgroup of its own BEHIND the panels in the same
  // grid (W = NP panels + the own rows; a 20 000-entry row inside a panel is one wave adding up 3 000-entry segments while seven
  // waves wait, and that panel sets the kernel time).  They are read from the CSR arrays the layout was built on.
  int NP = 0;                                   // panels proper (0: same as W)
  const int32_t* __restrict__ own_row = nullptr;   // W - NP rows
  const int32_t* __restrict__ own_ptr = nullptr;   // NP + 1: the own rows inside each panel's row range (marked: skipped by its epilogue)
  const int32_t* __restrict__ csr_off = nullptr;
  const int32_t* __restrict__ csr_idx = nullptr;
  const double* __restrict__ csr_val  = nullptr;
  // Dense row segments folded into the panel kernels (round 4; until then two launches of their own in front of every product):
  //  * rows of A that own segments are "own rows": their workgroup, behind the panels of the same grid, adds the sparse remainder of
  //    the row AND its segments (values index-free, the vector read coalesced) and runs the row's epilogue -- dn_own_seg[i] ...
  //    dn_own_seg[i + 1] are the segments of own row i (the rows that own segments are exactly a subset of the own rows, in order);
  //  * columns (rows of the A^T panels): the epilogue adds what the segments that cover the column contribute: the (at most
  //    kPanelDenseSegs) segments that reach into the panel's column range are listed per panel (dn_pan_ptr / dn_pan_seg, ascending
  //    rows = the order k_dense_cols adds them in) and copied to LDS with their vector entries before the chunks start.
  const int32_t* __restrict__ dn_own_seg   = nullptr;
  const int32_t* __restrict__ dn_pan_ptr   = nullptr;
  const int32_t* __restrict__ dn_pan_seg   = nullptr;
  const int32_t* __restrict__ dn_seg_row   = nullptr;
  const int32_t* __restrict__ dn_seg_c0    = nullptr;
  const int32_t* __restrict__ dn_seg_len   = nullptr;
  const int32_t* __restrict__ dn_seg_ptr   = nullptr;
  const double* __restrict__ dn_val        = nullptr;
};
constexpr int kPanelOwnRow = 4096;
constexpr int kPanelDenseSegs = 32;  // dense segments per panel of the column side that the epilogue handles itself (more: k_dense_cols)
constexpr long long kPanelNotMine = 0x7FF8C0DEC0DEC0DELL;  // a NaN no arithmetic produces: "this row is summed elsewhere"

// (the panel skeleton: spmv_panel.hpp)
constexpr int kPanelPer     = kPanelChunk / kPanelThreads;                          // rounds per full chunk
constexpr int kPanelRowsPer = (kPanelMaxRows + kPanelThreads - 1) / kPanelThreads;  // rows per lane
constexpr int kSegColBits = 20;                     // slab width < 2^20 columns (1.33 MiB slabs: 174 763)
constexpr unsigned kSegColMask = (1u << kSegColBits) - 1u;

// ------------------------------------------------------------------------------------------------
// Sorted jagged rows ("jag"): the SpMV for STRUCTURED matrices, whose rows re-use a limited set of columns.
// Measured (the round-2 harness: profiles/r02_spmv_tune2.txt, index in profiles/r05_harness_variants.txt): an 8-byte gather through the vector
// memory path costs ~1.85 clocks of the CU's texture-address unit per lane even when it hits L1
// (1e7 gathers never finish under 30 us), and every L1 miss occupies one of a CU's limited miss slots
// for an L2 round trip.  A workgroup of this layout therefore copies EVERY column its rows use into LDS
// once and gathers from LDS only:
//   * a workgroup owns consecutive rows, at most waves * kJagMaxGroup of them and as many as keep the set of
//     distinct columns they touch within the LDS window (jag_window entries); the set is either one contiguous
//     column range (banded matrices: copied coalesced, no list) or a sorted list of columns (several bands,
//     linking rows / columns, block structure: one 4-byte index per slot).  The matrix entries carry the
//     16-bit LDS slot of their column, not the column: 10 bytes per nonzero instead of 12, and no entry ever
//     falls back to a global gather;
//   * its rows with 1..kLongRow nonzeros are sorted by length (descending, stable), cut into passes of 64 and
//     dealt to the waves in snake order (equal work, and every pass holds rows of nearly equal length: 97 % of
//     the lanes of a jagged diagonal are live on Poisson row lengths, 80 % when each wave sorted only its own
//     256 rows).  A pass is stored as jagged diagonals: entry k of every row of the pass that has one,
//     contiguous -- lane <-> row, coalesced, no padding, no LDS staging of products and no barrier in
//     the loop;
//   * a lane adds up ITS row left to right in a register -> bit-identical to a sequential CSR sum;
//   * rows longer than kLongRow get a workgroup each (appended to the grid): 512 strided chains + the
//     fixed tree, read from the CSR arrays, like every long row of the other layouts;
//   * the row sums go through a 16 KiB LDS strip of the workgroup so that the fused epilogue runs in
//     natural row order (coalesced streams whatever the sort did to the rows).
// ------------------------------------------------------------------------------------------------
// Two geometries: 8 waves + a window of 8192 entries (80 KiB of LDS, two workgroups per CU), or 16 waves + 16384 entries
// (160 KiB, one workgroup per CU: the same 16 waves per CU, twice the rows sharing a window twice as wide) -- chosen per
// matrix at set-up (8 waves unless CUOPT_AMD_TUNE=jag_waves=16: see build_jag).
constexpr int kJagMaxGroup = 256;   // rows per wave (2 KiB of row sums)
#ifndef CUOPT_AMD_JAG_U
#define CUOPT_AMD_JAG_U 8
#endif
constexpr int kJagU        = CUOPT_AMD_JAG_U;  // jagged diagonals requested per round (tuning builds: -DCUOPT_AMD_JAG_U=...)
constexpr int jag_window(int waves) { return waves == 16 ? 16384 : 8192; }  // entries of the gathered vector per workgroup
constexpr size_t jag_lds_bytes(int waves) { return sizeof(double) * (size_t)(jag_window(waves) + waves * kJagMaxGroup); }
constexpr long long kJagNotMine = 0x7FF8C0DEC0DEC0DELL;  // a NaN no arithmetic produces: "this row is summed elsewhere"

struct JagView {
  int rows, waves, ngroups, nblk, nlong;  // workgroups: nblk of `waves` groups, then one per long row
  const int32_t* __restrict__ row0;     // nblk + 1: first row of each workgroup
  const int32_t* __restrict__ tile_e;   // ngroups + 1: first entry of each group
  const int32_t* __restrict__ tile_sr;  // ngroups + 1: first row descriptor of each group
  const uint32_t* __restrict__ sr;      // (length - 1) << 16 | row within the WORKGROUP's rows, sorted by length
  const uint16_t* __restrict__ slot;    // jagged-diagonal order: LDS slot of the entry's column
  const double* __restrict__ val;
  const int32_t* __restrict__ win;      // 2 * nblk: (first column, length) of a contiguous column set; length 0: a list
  const int32_t* __restrict__ set_ptr;  // nblk + 1: the workgroup's column list ...
  const int32_t* __restrict__ set_col;  // ... sorted columns, slot s holds vec[set_col[set_ptr[blk] + s]]
  const int32_t* __restrict__ lr_ptr;   // nblk + 1: the workgroup's rows longer than kLongRow ...
  const int32_t* __restrict__ lr_row;   // ... as global row numbers, read from the CSR arrays below
  const int32_t* __restrict__ off;
  const int32_t* __restrict__ idx;
  const double* __restrict__ csr_val;
  const double* __restrict__ dense_add = nullptr;  // per row: what the dense segments contribute (see dense_plus)
};

// (the jag skeleton: spmv_jag.hpp)

// ------------------------------------------------------------------------------------------------
// Gather-free layout ("pb": products, then rows) for UNSTRUCTURED matrices whose gathered vector is far beyond the caches.
// Measurements of the round-3 harness: profiles/r03_pb_* (index: profiles/r05_harness_variants.txt).  out = M v in two pure streams, no global gather at all:
//   phase P: the columns are cut into SOURCE PANELS (8192 or 16384 columns); a workgroup copies its panel's slice of v into
//     LDS and streams the panel's nonzeros -- value + 16-bit column within the panel -- in (panel, bin, row, column) order;
//     the products go, in pieces of G entries (aligned 8 G-byte blocks, piece table: 4 B per piece), to where phase R reads
//     them;
//   phase R: consecutive rows form a BIN whose products (one chunk per source panel, padded to whole pieces) fit in LDS and
//     lie contiguously in the product buffer; the bin's workgroup copies that image into LDS, then lane <-> row adds up
//     the row's products in COLUMN order (16-bit positions in jagged-diagonal order, rows sorted by length inside the bin)
//     -- a row is summed strictly left to right from 0.0, bit-identical to the sequential CSR sum, whatever its length --
//     and the fused epilogue runs in natural row order.
// 28 B per nonzero instead of 12, all of it coalesced: 5.5 / 4.6 TB/s on MI355X, the same time as the slab-major gather
// kernels at 1e7 nonzeros and 1.4x faster per nonzero at 1e8, where 16 slabs no longer fit the L2.
// ------------------------------------------------------------------------------------------------
constexpr int kPbThreads = 512;    // phase R workgroup (two per CU)
constexpr int kPbCap     = 9088;   // padded products per bin (71 KiB of LDS)
constexpr int kPbMaxRows = 1024;   // rows per bin (two groups of 64 per wave)
constexpr int kPbKU      = 16;     // jagged diagonals of positions requested before the bin's image has landed
constexpr size_t kPbLdsBytes = sizeof(double) * (size_t)(kPbCap + kPbMaxRows + 128);
typedef double pb_vec2d __attribute__((ext_vector_type(2)));
// WIDE bins (round 6, profiles/r06_gather_free_at_scale.txt): beyond ~2 M columns the (panel, bin) chunks of the image-in-LDS geometry
// shrink to a handful of entries -- panels x bins grows with the square of the size -- and phase P degenerates into 32-byte
// scattered stores.  A wide bin is kPbwRows consecutive rows whose ACCUMULATORS live in LDS (64 KB: two workgroups per CU); its image
// (chunks in panel order, padded to 16-entry pieces = whole 128-byte lines for phase P, the bin padded to whole steps) is STREAMED by
// phase R in steps of kPbwStep products, lane <-> slot, with a static 16-bit word per slot: row in the bin (13 bits) and LEVEL (3 bits:
// how many earlier slots of the same step belong to the same row; 7 = padding).  A step adds level 0, barrier, level 1, barrier, ...
// up to the step's highest level (a byte per step): every accumulator gets one addition per phase, by one lane, in image order =
// column order -- a row is still summed strictly left to right from 0.0, bit-identical to the sequential CSR sum.
constexpr int kPbwRows     = 8192;
constexpr int kPbwStep     = 1024;
constexpr int kPbwThreads  = 1024;
constexpr int kPbwMaxLevel = 6;
constexpr int kPbwAhead    = 4;     // steps of products requested ahead (the arrays carry kPbwAhead steps of slack behind the last bin)
constexpr int kPbwMaxSteps = 4096;  // steps of one bin (their levels sit in LDS as bytes)
constexpr size_t kPbwLdsBytes = sizeof(double) * (size_t)(kPbwRows + 256) + kPbwMaxSteps;

struct PbView {
  int rows, cols, S, B, gshift, panel_shift, nwg;  // G = 1 << gshift entries per piece, panels of 1 << panel_shift columns
  const double* __restrict__ val;        // padded entries, P order
  const uint16_t* __restrict__ lidx;     // column within the source panel
  const int32_t* __restrict__ piece_dst; // per piece of P order: its piece in the product buffer
  const int32_t* __restrict__ wg_e0;     // nwg + 1: entry range of each P workgroup
  const int32_t* __restrict__ wg_panel;  // nwg
  const int32_t* __restrict__ bin_row0;  // B + 1
  const int32_t* __restrict__ bin_e0;    // B + 1: the bin's image in the product buffer
  const uint32_t* __restrict__ sr;       // rows: length << 16 | row within the bin, sorted by length inside the bin
  const int32_t* __restrict__ bin_grp;   // B + 1 -> grp_pos
  const int32_t* __restrict__ grp_pos;   // first position of every 64-row group
  const uint16_t* __restrict__ pos;      // nnz (+ pad): position inside the bin's image, jagged-diagonal order
  double* __restrict__ prod;             // padded entries (+ pad)
  const double* __restrict__ dense_add = nullptr;  // per row: what the dense segments contribute (see dense_plus)
  // wide bins: sr / bin_grp / grp_pos / pos are unused (null)
  int wide = 0;
  const uint16_t* __restrict__ rib     = nullptr;  // per image slot: row in the bin | level << 13
  const uint8_t* __restrict__ step_lv  = nullptr;  // per step of kPbwStep slots: its highest level
  // SERIAL rows of the wide bins: a row with more than kPbwMaxLevel + 1 entries inside one step (a long or clustered row among short
  // ones) leaves the steps altogether -- its slots read as padding -- and ONE lane adds its products up, left to right, behind the
  // bin's last step
  int nser = 0;
  const int32_t* __restrict__ ser_ptr  = nullptr;  // B + 1 -> ser_row / ser_eptr
  const int32_t* __restrict__ ser_row  = nullptr;  // nser rows, ascending
  const int32_t* __restrict__ ser_eptr = nullptr;  // nser + 1 -> ser_slot
  const int32_t* __restrict__ ser_slot = nullptr;  // the image slots of a serial row's entries, in column order
};

// phase P of one workgroup.  xs: LDS, 1 << panel_shift doubles.
// (the pb skeleton: spmv_pb.hpp)

// ---- element-wise rules of the reference (LP/utils.cuh) -----------------------------------------
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ bool dfinite(double v) { return fabs(v) <= 1.7976931348623157e308; }
// combine_finite_abs_bounds, utils.cuh:139-148
__device__ __forceinline__ double combine_bounds(double lower, double upper)
{
  double val = 0.0;
  if (dfinite(upper)) val = dmax(val, fabs(upper));
  if (dfinite(lower)) val = dmax(val, fabs(lower));
  return val;
}
// violation, utils.cuh:165-178
__device__ __forceinline__ double violation(double value, double lower, double upper)
{
  if (value < lower) return lower - value;
  if (value > upper) return value - upper;
  return 0.0;
}
// bound_value_reduced_cost_product, utils.cuh:204-219
__device__ __forceinline__ double bound_value_product(double value, double lower, double upper)
{
  double bound = 0.0;
  if (value > 0.0)
    bound = lower;
  else if (value < 0.0)
    bound = upper;
  return dfinite(bound) ? value * bound : 0.0;
}

// ---- scalar logic shared by the multi-launch loop, the resident small-LP loop and the batches -------------------------------------
// The scalar rule of compute_step_sizes_from_movement_and_interaction (adaptive_step_size_strategy.cu:91-188) +
// the accept/flip of update_solution (pdhg.cu:237-250) + add_weight_sums (weighted_average_solution.cu:63-71).
// One thread.
// `pw` (optional): pow(k + 2, -reduction_exponent), pow(k + 2, -growth_exponent) for the k the attempt
// started with, computed off the critical path by the caller.
__device__ __forceinline__ void apply_step_decision(pdlpdev_ctl* ctl, double dy2, double interaction, double dx2,
                                                    const pdlpdev_step_params& sp, const double* pw = nullptr)
{
  const double w = ctl->primal_weight;
  double step    = ctl->step_size;
  const double movement = sp.primal_distance_smoothing * w * dx2 + (sp.dual_distance_smoothing / w) * dy2;
  ctl->last_interaction = interaction;
  ctl->last_movement    = movement;
  ctl->last_dx2         = dx2;
  ctl->last_dy2         = dy2;
  ctl->attempts += 1;
  bool accepted;
  // pdlp_constants.hpp:39-47 (movement <= 0 or >= 1e100), written so that a NaN -- which the reference's comparisons let through
  // into an endless series of rejected steps -- takes the same "invalid step size" exit (-> NumericalError at the next check)
  if (!(movement > 0.0) || !(movement < 1.0e100) || interaction != interaction) {
    // reference: flag -1, k and eta untouched; take_step still averages and swaps
    // (pdlp.cu:1193-1221) and the next loop trip is forced to be a major iteration.
    ctl->error = 1;
    accepted   = true;
  } else {
    const double inter = fabs(interaction);
    ctl->k += 1;
    const double kc    = (double)ctl->k;
    const double limit = inter > 0.0 ? movement / inter : __builtin_huge_val();
    accepted           = step <= limit;
    const double s1    = (1.0 - (pw ? pw[0] : pow(kc + 1.0, -sp.reduction_exponent))) * limit;
    const double s2    = (1.0 + (pw ? pw[1] : pow(kc + 1.0, -sp.growth_exponent))) * step;
    step               = dmin(s1, s2);
    ctl->step_size     = step;
    ctl->tau           = step / w;
    ctl->sigma         = step * w;
  }
  if (accepted) {
    ctl->cur ^= 1;
    ctl->pending_avg = 1;
    ctl->sum_weights += step;  // the ALREADY UPDATED step size (pdlp.cu:1216-1220)
    ctl->steps_taken += 1;
    ctl->its_since_restart += 1;
  } else {
    ctl->pending_avg = 0;
  }
}


// out[q] = reduce(part[q * nb .. q * nb + nb)) ; op_mask bit q set => max.  One workgroup of kBlock threads (k_finalize; the
// small-LP batch runs it once per LP inside one launch: same tree, same bits).
__device__ __forceinline__ void finalize_rows(const double* __restrict__ part, int nb, int nq, unsigned op_mask, double* __restrict__ out, double* red /* >= 8 */)
{
  for (int q = 0; q < nq; ++q) {
    const bool is_max = (op_mask >> q) & 1u;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < nb; i += kBlock) {
      const double v = part[(size_t)q * nb + i];
      acc[0] = is_max ? (v > acc[0] ? v : acc[0]) : acc[0] + v;
    }
    if (is_max)
      block_reduce<MaxOp, 1>(acc, red);
    else
      block_reduce<SumOp, 1>(acc, red);
    if (threadIdx.x == 0) out[q] = acc[0];
    __syncthreads();
  }
}

// restart: squared distances to the last-restart anchors, candidate -> iterate / anchors, sums <- 0
// (pdlp_restart_strategy.cu:593-623,752-839).  Workgroup `bid` of `nblocks` (k_restart: blockIdx.x of gridDim.x).
struct RestartView {
  int n, m, which, unscaled;
  const double *dc, *dr;
  const pdlpdev_ctl* ctl;
  double *x0, *x1, *y0, *y1;
  const double *avgx, *avgy;
  double *lrx, *lry, *sumx, *sumy, *part;
};
__device__ __forceinline__ void restart_block(const RestartView& R, int bid, int nblocks, double* red /* >= 12 */)
{
  const int cur = R.ctl->cur;
  double* __restrict__ x = cur ? R.x1 : R.x0;
  double* __restrict__ y = cur ? R.y1 : R.y0;
  double acc[2] = {0.0, 0.0};
  const int tot = R.n > R.m ? R.n : R.m;
  for (int i = bid * kBlock + threadIdx.x; i < tot; i += nblocks * kBlock) {
    if (i < R.n) {
      const double cand = R.which == PDLPDEV_AVERAGE ? R.avgx[i] : x[i];
      double d          = R.lrx[i] - 1.0 * cand;
      if (R.unscaled) d *= R.dc[i];
      acc[0] += d * d;
      if (R.which == PDLPDEV_AVERAGE) x[i] = cand;
      R.lrx[i]  = cand;
      R.sumx[i] = 0.0;
    }
    if (i < R.m) {
      const double cand = R.which == PDLPDEV_AVERAGE ? R.avgy[i] : y[i];
      double d          = R.lry[i] - 1.0 * cand;
      if (R.unscaled) d *= R.dr[i];
      acc[1] += d * d;
      if (R.which == PDLPDEV_AVERAGE) y[i] = cand;
      R.lry[i]  = cand;
      R.sumy[i] = 0.0;
    }
  }
  block_reduce<SumOp, 2>(acc, red);
  if (threadIdx.x == 0) {
    R.part[bid]           = acc[0];
    R.part[nblocks + bid] = acc[1];
  }
}

}  // namespace pdlp
