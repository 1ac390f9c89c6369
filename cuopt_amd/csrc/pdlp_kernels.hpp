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

#include <type_traits>

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

// An epilogue may split its row step into load(row) -> Ops and apply(row, sum, Ops): the operands of a lane's first row are then
// requested together with the matrix stream instead of behind the row sums (one dependent round trip less per workgroup -- on a
// cache-resident LP the kernels are chains of round trips, nothing else).  Same arithmetic, same bits.
template <class E, class = void>
struct epilogue_has_ops : std::false_type {};
template <class E>
struct epilogue_has_ops<E, std::void_t<typename E::Ops>> : std::true_type {};
struct NoOps {};
template <class E, bool = epilogue_has_ops<E>::value>
struct epilogue_ops { using type = NoOps; };
template <class E>
struct epilogue_ops<E, true> { using type = typename E::Ops; };

// block b executes on XCD (b % 8) (observed dispatch order, speed only): give each XCD one
// contiguous range of row blocks so that neighbouring rows -- which in real LPs touch neighbouring
// columns -- share that XCD's private 4 MiB L2 for the gathered vector.
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
  const int per = (nb + 7) >> 3;
  const int v   = (b & 7) * per + (b >> 3);
  return v;  // may be >= nb for the ragged tail: caller skips
}

// ------------------------------------------------------------------------------------------------
// CSR "stream" SpMV skeleton.  One workgroup owns the contiguous row range
// [row_blocks[b], row_blocks[b+1]) whose nonzeros (<= kNnzBlock unless it is a single long row)
// are loaded fully coalesced (lane i <-> nonzero k0+i), multiplied by the gathered vector entry
// and parked in LDS; then lane r adds up row r's products in CSR order and hands (row, sum) to the
// epilogue.  Epilogue concept:
//   struct E { static constexpr int NQ; using Op; __device__ void row(int r, double sum, double (&acc)[NQ]); }
// After the rows, acc[] is reduced over the workgroup and written to partials[q * nb + b].
// ------------------------------------------------------------------------------------------------
template <class Epi>
__device__ __forceinline__ void csr_stream_block(int nb, const int32_t* __restrict__ row_blocks,
                                                 const int32_t* __restrict__ offsets,
                                                 const int32_t* __restrict__ indices,
                                                 const double* __restrict__ values,
                                                 const double* __restrict__ vec, Epi& epi,
                                                 double* __restrict__ partials, const double* __restrict__ dense_add = nullptr)
{
  __shared__ __attribute__((aligned(32))) double prod[kNnzTile];
  __shared__ double red[4 * (Epi::NQ > 0 ? Epi::NQ : 1) + 4];
  const int b = xcd_remap(blockIdx.x, nb);
  if (b >= nb) return;
  const int r0 = row_blocks[b], r1 = row_blocks[b + 1];
  const int k0 = row_blocks[nb + 1 + b], k1 = row_blocks[nb + 2 + b];  // = offsets[r0], offsets[r1] (build_row_blocks)
  const int cnt = k1 - k0;
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();

  // The tile starts at k0 rounded DOWN to a multiple of 4 nonzeros, so every lane's 4 values / 4 indices are one
  // aligned 32-byte / 16-byte vector load (the arrays are padded by 8 entries; the up-to-3 foreign entries at
  // either end are multiplied like the others and never summed).  All loads of the workgroup are issued
  // before the first LDS write.  Measured on a banded 1e7-nnz matrix: 39.9 -> 29.5 us (59 % of the HBM roofline).
  const int base = k0 & ~3;
  if (cnt <= kNnzBlock) {
    constexpr int kPasses = kNnzTile / (4 * kBlock);
    // extents of the lane's first rows, requested together with the matrix stream: their latency would otherwise sit
    // between the barrier and the row sums (every workgroup of a mid-size LP is one dependent chain of round trips)
    constexpr int kPre = 4;
    int ext0[kPre], ext1[kPre];
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
      int r   = r0 + threadIdx.x + q * kBlock;
      r       = r < r1 ? r : r1 - 1;
      ext0[q] = offsets[r];
      ext1[q] = offsets[r + 1];
    }
    typename epilogue_ops<Epi>::type pre{};  // the epilogue's operands of the lane's first row, in flight with the matrix stream
    if constexpr (epilogue_has_ops<Epi>::value) {
      const int r = r0 + (int)threadIdx.x;
      pre         = epi.load(r < r1 ? r : r1 - 1);
    }
    vec4d a[kPasses];
    vec4i j[kPasses];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int k = base + 4 * (p * kBlock + threadIdx.x);
      if (k < k1) {
        a[p] = __builtin_nontemporal_load(reinterpret_cast<const vec4d*>(values + k));
        j[p] = __builtin_nontemporal_load(reinterpret_cast<const vec4i*>(indices + k));
      } else {
        a[p] = (vec4d)(0.0);
        j[p] = (vec4i)(0);
      }
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const vec4d g = {vec[j[p].x], vec[j[p].y], vec[j[p].z], vec[j[p].w]};
      *reinterpret_cast<vec4d*>(&prod[4 * (p * kBlock + threadIdx.x)]) = a[p] * g;
    }
    __syncthreads();
    int q = 0;
    for (int r = r0 + threadIdx.x; r < r1; r += kBlock, ++q) {
      int s, e;
      switch (q) {  // register arrays want static indices
        case 0: s = ext0[0], e = ext1[0]; break;
        case 1: s = ext0[1], e = ext1[1]; break;
        case 2: s = ext0[2], e = ext1[2]; break;
        case 3: s = ext0[3], e = ext1[3]; break;
        default: s = offsets[r], e = offsets[r + 1]; break;
      }
      s -= base, e -= base;
      double sum = 0.0;
      if (e - s <= kLongRow) {
        for (int k = s; k < e; ++k) sum = sum + prod[k];
      } else {  // 4 interleaved chains: a medium-long row must not serialise the workgroup
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = s;
        for (; k + 3 < e; k += 4) {
          s0 += prod[k], s1 += prod[k + 1], s2 += prod[k + 2], s3 += prod[k + 3];
        }
        for (; k < e; ++k) s0 += prod[k];
        sum = (s0 + s1) + (s2 + s3);
      }
      if constexpr (epilogue_has_ops<Epi>::value) {
        if (q == 0) {
          epi.apply(r, dense_plus(dense_add, r, sum), pre, acc);
          continue;
        }
      }
      epi.row(r, dense_plus(dense_add, r, sum), acc);
    }
  } else {
    // a single row longer than the LDS tile: strided partial sums + workgroup tree
    double part[1] = {0.0};
    for (int k = k0 + threadIdx.x; k < k1; k += kBlock) {
      const double a = __builtin_nontemporal_load(values + k);
      const int j    = __builtin_nontemporal_load(indices + k);
      part[0] += a * vec[j];
    }
    block_reduce<SumOp, 1>(part, red);
    if (threadIdx.x == 0) epi.row(r0, dense_plus(dense_add, r0, part[0]), acc);
    __syncthreads();
  }
  if constexpr (Epi::NQ > 0) {
    block_reduce<typename Epi::Op, Epi::NQ>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nb + b] = acc[q];
    }
  }
}


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
};
constexpr int kPanelOwnRow = 4096;
constexpr long long kPanelNotMine = 0x7FF8C0DEC0DEC0DELL;  // a NaN no arithmetic produces: "this row is summed elsewhere"

// ---- the chunks of a panel, one stage ahead ------------------------------------------------------------------------
// A chunk is <= kPanelChunk consecutive nonzeros of one tile, handled in R = ceil(len / 512) rounds (lane <-> nonzero;
// only the last round has idle lanes: they re-read the chunk's last nonzero and never store).  The value / column
// loads and the packed 16-bit row extents of chunk i+1 are requested before the row sums of chunk i, so one of the two
// dependent memory round trips of a chunk (matrix stream -> gather) always overlaps LDS work of the same workgroup.
// Stages are straight-line code selected by a switch on R (exact load accounting, no wasted gathers).
constexpr int kPanelPer     = kPanelChunk / kPanelThreads;                          // rounds per full chunk
constexpr int kPanelRowsPer = (kPanelMaxRows + kPanelThreads - 1) / kPanelThreads;  // rows per lane
struct PanelChunk {
  int s, c0, c1, t0;  // tile (slab) index, nonzero range of the chunk, start of its tile
  bool valid;
};
template <int R>
__device__ __forceinline__ void panel_load(const PanelView& P, double (&va)[kPanelPer], int (&ja)[kPanelPer], int c0, int c1)
{
#pragma unroll
  for (int u = 0; u < R; ++u) {
    int k = c0 + threadIdx.x + u * kPanelThreads;
    if (u == R - 1) k = k < c1 ? k : c1 - 1;
    va[u] = __builtin_nontemporal_load(P.val + k);
    ja[u] = __builtin_nontemporal_load(P.col + k);
  }
}
template <int R>
__device__ __forceinline__ void panel_products(const double* __restrict__ vec, double* prod, const double (&va)[kPanelPer],
                                               const int (&ja)[kPanelPer], int len)
{
  double x[R];
#pragma unroll
  for (int u = 0; u < R; ++u) x[u] = vec[ja[u]];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int i = threadIdx.x + u * kPanelThreads;
    if (u < R - 1 || i < len) prod[i] = va[u] * x[u];
  }
}
#define PANEL_DISPATCH(rounds, CALL) \
  switch (rounds) {                   \
    case 1: CALL(1); break;           \
    case 2: CALL(2); break;           \
    case 3: CALL(3); break;           \
    case 4: CALL(4); break;           \
    case 5: CALL(5); break;           \
    case 6: CALL(6); break;           \
    case 7: CALL(7); break;           \
    default: CALL(8); break;          \
  }

// ---- a row of its own (more than kPanelOwnRow nonzeros): the whole workgroup strides over it, 16 entries per thread in flight
// (fixed tree, compared with a tolerance like every long row)
template <class Epi>
__device__ __forceinline__ void panel_own_row(const PanelView& P, const double* __restrict__ vec, Epi& epi, double* __restrict__ partials,
                                              int w, int NP, double* scratch /* >= kPanelWaves doubles of LDS */)
{
  const int r  = P.own_row[w - NP];
  const int k0 = P.csr_off[r], k1 = P.csr_off[r + 1];
  double part[1] = {0.0};
  constexpr int kLongU = 16;
  for (int k = k0 + (int)threadIdx.x; k < k1; k += kLongU * kPanelThreads) {
    double a[kLongU];
    int j[kLongU];
#pragma unroll
    for (int u = 0; u < kLongU; ++u) {
      a[u] = 0.0, j[u] = 0;
      if (k + u * kPanelThreads < k1) {
        a[u] = __builtin_nontemporal_load(P.csr_val + k + u * kPanelThreads);
        j[u] = __builtin_nontemporal_load(P.csr_idx + k + u * kPanelThreads);
      }
    }
    double xv[kLongU];
#pragma unroll
    for (int u = 0; u < kLongU; ++u) xv[u] = k + u * kPanelThreads < k1 ? vec[j[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kLongU; ++u) part[0] = part[0] + a[u] * xv[u];
  }
  block_reduce<SumOp, 1, kPanelWaves>(part, scratch);
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  if (threadIdx.x == 0) {
    epi.row(r, dense_plus(P.dense_add, r, part[0]), acc);
    if constexpr (Epi::NQ > 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * P.W + w] = acc[q];
    }
  }
}
// the fused epilogue over a panel's rows, natural order, from the row sums in LDS
template <class Epi>
__device__ __forceinline__ void panel_epilogue(const PanelView& P, Epi& epi, double* __restrict__ partials, const double* psum, double* red,
                                               int w, int r0, int nr, const double* psum2 = nullptr /* long-tail variant: the edge runs' share */)
{
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  for (int r = threadIdx.x; r < nr; r += kPanelThreads)
    if (__double_as_longlong(psum[r]) != kPanelNotMine) epi.row(r0 + r, dense_plus(P.dense_add, r0 + r, psum2 ? psum[r] + psum2[r] : psum[r]), acc);
  if constexpr (Epi::NQ > 0) {
    block_reduce<typename Epi::Op, Epi::NQ, kPanelWaves>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * P.W + w] = acc[q];
    }
  }
}

template <class Epi>
__device__ __forceinline__ void panel_spmv_block(const PanelView& P, const double* __restrict__ vec,
                                                 Epi& epi, double* __restrict__ partials)
{
  static_assert(kPanelPer == 8, "PANEL_DISPATCH enumerates 1..8 rounds");
  __shared__ double prod[kPanelChunk];
  __shared__ double psum[kPanelMaxRows];
  __shared__ double red[kPanelWaves * (Epi::NQ > 0 ? Epi::NQ : 1)];
  __shared__ int tile_s[17];        // this panel's tile boundaries (S <= 16) and row-pointer bases, fetched once
  __shared__ long long base_s[16];
  const int w  = blockIdx.x;
  const int NP = P.NP ? P.NP : P.W;
  if (w >= NP) {
    panel_own_row(P, vec, epi, partials, w, NP, prod);
    return;
  }
  const int r0 = P.row0[w], nr = P.row0[w + 1] - r0;
  if (threadIdx.x <= P.S) tile_s[threadIdx.x] = P.tile_ptr[w * P.S + threadIdx.x];
  if (threadIdx.x < P.S) base_s[threadIdx.x] = P.rp_base[w * P.S + threadIdx.x];
  for (int r = threadIdx.x; r < nr; r += kPanelThreads) psum[r] = 0.0;
  __syncthreads();
  if (P.own_ptr)  // rows that have a workgroup of their own: no segment of theirs is in the tiles, the epilogue below skips them
    for (int q = P.own_ptr[w] + (int)threadIdx.x; q < P.own_ptr[w + 1]; q += kPanelThreads) psum[P.own_row[q] - r0] = __longlong_as_double(kPanelNotMine);
  auto advance = [&](PanelChunk c) -> PanelChunk {
    if (c.valid && c.c1 < tile_s[c.s + 1]) {  // next chunk of the same tile
      c.c0 = c.c1;
      c.c1 = c.c0 + kPanelChunk < tile_s[c.s + 1] ? c.c0 + kPanelChunk : tile_s[c.s + 1];
      return c;
    }
    int s = c.s + 1;
    while (s < P.S && tile_s[s + 1] == tile_s[s]) ++s;  // empty tiles contribute nothing
    c.valid = s < P.S;
    c.s     = s;
    if (c.valid) {
      c.t0 = c.c0 = tile_s[s];
      c.c1 = c.c0 + kPanelChunk < tile_s[s + 1] ? c.c0 + kPanelChunk : tile_s[s + 1];
    }
    return c;
  };
  double va[kPanelPer];
  int ja[kPanelPer];
  unsigned ext_next[kPanelRowsPer], ext[kPanelRowsPer];  // (begin | end << 16) of the lane's rows in the chunk's tile
  // (macros, not lambdas: the register arrays must stay visible to scalar replacement)
#define PANEL_ROUNDS(c) (((c).c1 - (c).c0 + kPanelThreads - 1) / kPanelThreads)
#define PANEL_CALL_LOAD(R) panel_load<R>(P, va, ja, nxt.c0, nxt.c1)
#define PANEL_CALL_PRODUCTS(R) panel_products<R>(vec, prod, va, ja, cur.c1 - cur.c0)
#define PANEL_REQUEST(c)                                                         \
  if ((c).valid) {                                                               \
    PANEL_DISPATCH(PANEL_ROUNDS(c), PANEL_CALL_LOAD)                             \
    const uint16_t* __restrict__ rp_ = P.rowptr + base_s[(c).s];                 \
    _Pragma("unroll") for (int q = 0; q < kPanelRowsPer; ++q) {                  \
      int r_      = threadIdx.x + q * kPanelThreads;                             \
      r_          = r_ < nr ? r_ : 0;                                            \
      ext_next[q] = (unsigned)rp_[r_] | ((unsigned)rp_[r_ + 1] << 16);           \
    }                                                                            \
  }
  PanelChunk none{-1, 0, 0, 0, false};
  PanelChunk nxt = advance(none);
  PANEL_REQUEST(nxt)
  PanelChunk cur = nxt;
  while (cur.valid) {
    __syncthreads();  // the previous chunk's row sums are done with prod
    PANEL_DISPATCH(PANEL_ROUNDS(cur), PANEL_CALL_PRODUCTS)
#pragma unroll
    for (int q = 0; q < kPanelRowsPer; ++q) ext[q] = ext_next[q];
    const int lo = cur.c0 - cur.t0, hi = cur.c1 - cur.t0;
    nxt = advance(cur);
    PANEL_REQUEST(nxt)  // in flight during the row sums below
    __syncthreads();
    if (!P.any_long) {  // (uniform) the common case: no extra instruction in the loop
#pragma unroll
      for (int q = 0; q < kPanelRowsPer; ++q) {
        const int r = threadIdx.x + q * kPanelThreads;
        if (r < nr) {
          int a = (int)(ext[q] & 0xFFFFu), b = (int)(ext[q] >> 16);
          a = a > lo ? a : lo;
          b = b < hi ? b : hi;
          if (a < b) {
            double sum = psum[r];
            for (int k = a; k < b; ++k) sum = sum + prod[k - lo];
            psum[r] = sum;
          }
        }
      }
      cur = nxt;
      continue;
    }
    bool any_long = false;
#pragma unroll
    for (int q = 0; q < kPanelRowsPer; ++q) {
      const int r = threadIdx.x + q * kPanelThreads;
      if (r < nr) {
        int a = (int)(ext[q] & 0xFFFFu), b = (int)(ext[q] >> 16);
        a = a > lo ? a : lo;
        b = b < hi ? b : hi;
        if (b - a > kLongRow) {
          any_long = true;  // handled below, by the whole wave
        } else if (a < b) {  // left to right by the row's lane: bit-identical to a sequential CSR sum
          double sum = psum[r];
          for (int k = a; k < b; ++k) sum = sum + prod[k - lo];
          psum[r] = sum;
        }
      }
    }
    // A segment longer than kLongRow would keep ONE lane busy for thousands of dependent LDS reads: its wave sums it
    // together instead (64 strided chains + the fixed butterfly; compared with a tolerance like every long row).
    // One ballot per chunk on the common path.
    if (__ballot(any_long)) {
#pragma unroll
      for (int q = 0; q < kPanelRowsPer; ++q) {
        const int r = threadIdx.x + q * kPanelThreads;
        int a = 0, b = 0;
        if (r < nr) {
          a = (int)(ext[q] & 0xFFFFu), b = (int)(ext[q] >> 16);
          a = a > lo ? a : lo;
          b = b < hi ? b : hi;
        }
        unsigned long long todo = __ballot(b - a > kLongRow);
        while (todo) {
          const int l  = __builtin_ctzll(todo);
          todo &= todo - 1;
          const int la = __builtin_amdgcn_readlane(a, l), lb = __builtin_amdgcn_readlane(b, l);
          double part  = 0.0;
          for (int k = la + (int)(threadIdx.x & 63); k < lb; k += 64) part = part + prod[k - lo];
          part = wave_reduce<SumOp>(part);
          if ((int)(threadIdx.x & 63) == l) psum[r] = psum[r] + part;
        }
      }
    }
    cur = nxt;
  }
#undef PANEL_REQUEST
#undef PANEL_CALL_PRODUCTS
#undef PANEL_CALL_LOAD
#undef PANEL_ROUNDS
  __syncthreads();
  panel_epilogue(P, epi, partials, psum, red, w, r0, nr);
}
// ------------------------------------------------------------------------------------------------
// Long-tail variant of the panels: row sums dealt by NONZERO, not by row.
// The panel kernel above gives every ROW of the panel a lane, which walks the row's segment of the staged chunk while its wave
// waits for the longest one: on a matrix whose row lengths have a heavy tail (power law: 0.26 of the HBM roofline in round 3,
// 21.6 M issue cycles against 14.0 M on the uniform matrix of the same size) that walk is what the kernel waits for.  Here
//   * the storage is the panels' (chunk entry i <-> lane i % 512, round i / 512: neighbouring lanes hold neighbouring nonzeros, which
//     is what lets the texture path merge the gathers of neighbouring columns -- a first version that gave each lane CONSECUTIVE
//     entries lost 8 % on matrices with a diagonal for exactly that reason); each entry carries its row within the panel (12 bits)
//     next to its column relative to the slab (20 bits): 12 bytes per nonzero and no row pointers at all; products stay in registers;
//   * per round, a wave holds 64 consecutive entries: a segmented scan over the wave (DPP, fixed tree) sums the runs of equal rows;
//     a run that lies inside the wave-round is added to its row's LDS sum by the lane that ends it (LDS atomic as a fire-and-forget
//     add: one emission per run and chunk, so never two lanes at one row between two barriers);
//   * the runs that touch the edges of a wave-round (at most two per wave and round) go to a table of 64 records per chunk; after the
//     chunk's ONE barrier wave 0 joins neighbouring records of equal rows -- a second, 64-lane segmented scan in logical order --
//     and adds the joined sums to a second LDS strip that only it writes (so no emission of the next chunk can race with it).
// The work of a chunk is the same whatever the row lengths are; rows of any length need no special path.  The additions of a row are
// no longer left to right: this layout is compared with the oracle at rtol 1e-12 for EVERY row (the contract of rows > kLongRow in
// the other layouts), which is why `auto` takes it only for long-tailed matrices (build_panels) and the uniform ones keep their
// bit-exact kernels.
// ------------------------------------------------------------------------------------------------
constexpr int kSegColBits = 20;                     // slab width < 2^20 columns (1.33 MiB slabs: 174 763)
constexpr unsigned kSegColMask = (1u << kSegColBits) - 1u;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double seg_dpp(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// inclusive sums over the wave's lanes, restarting at every lane >= seg0 (seg0: the first lane of this lane's segment): four row_shr
// steps inside the 16-lane DPP rows, then the row totals through row_bcast:15 / row_bcast:31 -- VALU only, a fixed tree
__device__ __forceinline__ double seg_scan(double S, int lane, int seg0)
{
  double u;
  u = seg_dpp<0x111, 0xf>(S); if ((lane & 15) >= 1 && lane - 1 >= seg0) S = S + u;
  u = seg_dpp<0x112, 0xf>(S); if ((lane & 15) >= 2 && lane - 2 >= seg0) S = S + u;
  u = seg_dpp<0x114, 0xf>(S); if ((lane & 15) >= 4 && lane - 4 >= seg0) S = S + u;
  u = seg_dpp<0x118, 0xf>(S); if ((lane & 15) >= 8 && lane - 8 >= seg0) S = S + u;
  u = seg_dpp<0x142, 0xa>(S); if ((lane & 16) && (lane | 15) - 16 >= seg0) S = S + u;  // rows 1, 3 <- the last lane of rows 0, 2
  u = seg_dpp<0x143, 0xc>(S); if (lane >= 32 && 31 >= seg0) S = S + u;                 // rows 2, 3 <- lane 31
  return S;
}
__device__ __forceinline__ void seg_emit(double* strip, int row, double v)
{
  __hip_atomic_fetch_add(strip + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the edge runs of one wave-round: the run that starts at lane 0 and -- unless that run fills the whole wave-round -- the run that
// ends at lane 63
struct SegRecord {
  double lsum, rsum;
  int lkey, rkey;  // rkey < 0: one run from edge to edge (lsum is its sum)
};
template <int R>
__device__ __forceinline__ void seg_load(const PanelView& P, double (&va)[kPanelPer], int (&pk)[kPanelPer], int c0, int c1)
{
#pragma unroll
  for (int u = 0; u < R; ++u) {
    int k = c0 + threadIdx.x + u * kPanelThreads;
    if (u == R - 1) k = k < c1 ? k : c1 - 1;
    va[u] = __builtin_nontemporal_load(P.val + k);
    pk[u] = __builtin_nontemporal_load(P.col + k);
  }
}
template <int R>
__device__ __forceinline__ void seg_rounds(const double* __restrict__ vec, int slab_base, const double (&va)[kPanelPer], const int (&pk)[kPanelPer], int len,
                                           double* psum, SegRecord* rec /* this wave's records of the chunk, one per round */, int lane)
{
  double x[R];
#pragma unroll
  for (int u = 0; u < R; ++u) x[u] = vec[slab_base + (int)((unsigned)pk[u] & kSegColMask)];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int i   = threadIdx.x + u * kPanelThreads;
    const int key = (int)((unsigned)pk[u] >> kSegColBits);
    const double v = (u < R - 1 || i < len) ? va[u] * x[u] : 0.0;  // lanes behind the chunk's end repeat its last entry's row with nothing to add
    const int prev = __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 0: -1, no row)
    const unsigned long long starts = __ballot(key != prev);                       // bit t: a run starts at lane t (bit 0 always)
    const unsigned long long below  = starts & ((2ull << lane) - 1ull);
    const int seg0                  = 63 - __builtin_clzll(below);
    const double S                  = seg_scan(v, lane, seg0);
    const bool ends                 = lane == 63 || ((starts >> (lane + 1)) & 1ull);
    if (ends) {
      if (seg0 == 0) {  // the run that started at the wave-round's left edge
        rec[u].lsum = S, rec[u].lkey = key;
        if (lane == 63) rec[u].rkey = -1;
      } else if (lane == 63) {
        rec[u].rsum = S, rec[u].rkey = key;
      } else {
        seg_emit(psum, key, S);
      }
    }
  }
}
template <class Epi>
__device__ __forceinline__ void panel_seg_block(const PanelView& P, const double* __restrict__ vec, Epi& epi, double* __restrict__ partials)
{
  static_assert(kPanelPer == 8 && kPanelWaves == 8, "64 wave-rounds per chunk: one lane of wave 0 each");
  static_assert(kPanelMaxRows <= (1 << (32 - kSegColBits)), "row within the panel must fit beside the column");
  __shared__ double psum[kPanelMaxRows];   // runs inside a wave-round (every wave emits)
  __shared__ double psum2[kPanelMaxRows];  // runs that touch a wave-round's edge, joined (wave 0 alone emits)
  __shared__ double red[kPanelWaves * (Epi::NQ > 0 ? Epi::NQ : 1)];
  __shared__ SegRecord rec[2][kPanelWaves * kPanelPer];  // [chunk parity][wave * 8 + round]
  __shared__ int tile_s[17];
  const int w  = blockIdx.x;
  const int NP = P.NP ? P.NP : P.W;
  if (w >= NP) {
    panel_own_row(P, vec, epi, partials, w, NP, psum);
    return;
  }
  const int r0 = P.row0[w], nr = P.row0[w + 1] - r0;
  if (threadIdx.x <= P.S) tile_s[threadIdx.x] = P.tile_ptr[w * P.S + threadIdx.x];
  for (int r = threadIdx.x; r < nr; r += kPanelThreads) psum[r] = 0.0, psum2[r] = 0.0;
  __syncthreads();
  if (P.own_ptr)
    for (int q = P.own_ptr[w] + (int)threadIdx.x; q < P.own_ptr[w + 1]; q += kPanelThreads) psum[P.own_row[q] - r0] = __longlong_as_double(kPanelNotMine);
  auto advance = [&](PanelChunk c) -> PanelChunk {
    if (c.valid && c.c1 < tile_s[c.s + 1]) {
      c.c0 = c.c1;
      c.c1 = c.c0 + kPanelChunk < tile_s[c.s + 1] ? c.c0 + kPanelChunk : tile_s[c.s + 1];
      return c;
    }
    int s = c.s + 1;
    while (s < P.S && tile_s[s + 1] == tile_s[s]) ++s;
    c.valid = s < P.S;
    c.s     = s;
    if (c.valid) {
      c.t0 = c.c0 = tile_s[s];
      c.c1 = c.c0 + kPanelChunk < tile_s[s + 1] ? c.c0 + kPanelChunk : tile_s[s + 1];
    }
    return c;
  };
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double va[kPanelPer];
  int pk[kPanelPer];
#define SEG_ROUNDS(c) (((c).c1 - (c).c0 + kPanelThreads - 1) / kPanelThreads)
#define SEG_CALL_LOAD(R) seg_load<R>(P, va, pk, nxt.c0, nxt.c1)
#define SEG_CALL_ROUNDS(R) seg_rounds<R>(vec, cur.s * P.slab_w, va, pk, cur.c1 - cur.c0, psum, &rec[parity][wave * kPanelPer], lane)
  PanelChunk none{-1, 0, 0, 0, false};
  PanelChunk nxt = advance(none);
  if (nxt.valid) { PANEL_DISPATCH(SEG_ROUNDS(nxt), SEG_CALL_LOAD) }
  PanelChunk cur = nxt;
  int parity = 0;
  while (cur.valid) {
    const int R = SEG_ROUNDS(cur);
    PANEL_DISPATCH(R, SEG_CALL_ROUNDS)   // gathers, products, the runs inside the wave-rounds, the edge records
    nxt = advance(cur);
    if (nxt.valid) { PANEL_DISPATCH(SEG_ROUNDS(nxt), SEG_CALL_LOAD) }  // the next chunk's stream, in flight over the barrier and the join
    __syncthreads();  // the chunk's records are complete; every emission of the chunk before is done
    if (wave == 0) {
      // join the edge runs in logical order: lane l <-> wave-round (round l / 8, wave l % 8), two pieces each (the run from the
      // left edge, the run to the right edge; one piece when a single run fills the wave-round)
      const int u = lane >> 3, q = lane & 7;
      const bool on = u < R;
      SegRecord g{0.0, 0.0, 0, -1};
      if (on) g = rec[parity][q * kPanelPer + u];
      const bool single = g.rkey < 0;
      const int my_first = g.lkey, my_last = single ? g.lkey : g.rkey;
      const double open  = on ? (single ? g.lsum : g.rsum) : 0.0;  // the run still open at this lane's right edge
      int prev_last      = __builtin_amdgcn_update_dpp(0, my_last, 0x138, 0xf, 0xf, false);
      if (lane == 0) prev_last = my_first;
      const bool flag = on && (!single || my_first != prev_last);  // a run ends inside this lane's pieces or at its left edge
      const unsigned long long fm = __ballot(flag);
      const unsigned long long at_or_below = fm & ((2ull << lane) - 1ull);
      const int seg0  = at_or_below ? 63 - __builtin_clzll(at_or_below) : 0;
      const double S  = seg_scan(open, lane, seg0);
      double carry_in = seg_dpp<0x138, 0xf>(S);
      if (lane == 0) carry_in = 0.0;
      if (on) {
        int key    = prev_last;
        double sum = carry_in;
        if (my_first != key) {
          seg_emit(psum2, key, sum);
          key = my_first, sum = 0.0;
        }
        sum = sum + g.lsum;
        if (!single) {
          seg_emit(psum2, key, sum);
          key = g.rkey, sum = g.rsum;
        }
        if (lane == R * kPanelWaves - 1) seg_emit(psum2, key, sum);  // the chunk's last wave-round closes what is still open
      }
    }
    parity ^= 1;
    cur = nxt;
  }
#undef SEG_CALL_ROUNDS
#undef SEG_CALL_LOAD
#undef SEG_ROUNDS
  __syncthreads();
  panel_epilogue(P, epi, partials, psum, red, w, r0, nr, psum2);
}
#undef PANEL_DISPATCH
template <bool SEG, class Epi>
__device__ __forceinline__ void panel_block(const PanelView& P, const double* __restrict__ vec, Epi& epi, double* __restrict__ partials)
{
  if constexpr (SEG) panel_seg_block(P, vec, epi, partials);
  else panel_spmv_block(P, vec, epi, partials);
}

// ------------------------------------------------------------------------------------------------
// Sorted jagged rows ("jag"): the SpMV for STRUCTURED matrices, whose rows re-use a limited set of columns.
// Measured (tools/spmv_tune2.hip, profiles/r02_spmv_tune2.txt): an 8-byte gather through the vector
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
// matrix at set-up (8 waves unless CUOPT_AMD_JAG_WAVES=16: see build_jag).
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

template <class Epi, int WAVES>
__device__ __forceinline__ void jag_block(const JagView& J, const double* __restrict__ vec, Epi& epi,
                                          double* __restrict__ partials)
{
  extern __shared__ __attribute__((aligned(16))) double jag_lds[];
  double* xwin   = jag_lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // grid: the row blocks (blockIdx & 7 = XCD -> a contiguous range of row blocks per XCD), padded to a multiple of 8, then the
  // long rows round-robin over the XCDs (inside one xcd_remap over both, the last XCDs would get long rows only);
  // partials keep the order [row blocks..., long rows...]
  const int nparts = J.nblk + J.nlong;
  const int nb8    = (J.nblk + 7) & ~7;
  int blk;
  if ((int)blockIdx.x < nb8) {
    blk = xcd_remap((int)blockIdx.x, J.nblk);
    if (blk >= J.nblk) return;
  } else {
    blk = J.nblk + ((int)blockIdx.x - nb8);
    if (blk >= nparts) return;
  }
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  if (blk >= J.nblk) {
    // ---- a row longer than kLongRow: the whole workgroup strides over it (fixed tree, compared with a tolerance like
    // every long row of the other layouts); such rows are spread over the grid instead of queueing up in one wave ----
    const int r  = J.lr_row[blk - J.nblk];
    const int k0 = J.off[r], k1 = J.off[r + 1];
    double part[1] = {0.0};
    // 16 entries per thread in flight: a row of up to 8192 nonzeros costs two dependent round trips (entries, then gathers),
    // not two per 2048 -- these workgroups run behind the row blocks and their latency is the kernel's tail
    constexpr int kLongU = 16;
    for (int k = k0 + (int)threadIdx.x; k < k1; k += kLongU * (WAVES * 64)) {
      double a[kLongU];
      int j[kLongU];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) {
        a[u] = 0.0, j[u] = 0;
        if (k + u * (WAVES * 64) < k1) {
          a[u] = __builtin_nontemporal_load(J.csr_val + k + u * (WAVES * 64));
          j[u] = __builtin_nontemporal_load(J.idx + k + u * (WAVES * 64));
        }
      }
      double xv[kLongU];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) xv[u] = k + u * (WAVES * 64) < k1 ? vec[j[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < kLongU; ++u) part[0] = part[0] + a[u] * xv[u];
    }
    block_reduce<SumOp, 1, WAVES>(part, xwin);
    if (threadIdx.x == 0) epi.row(r, dense_plus(J.dense_add, r, part[0]), acc);
    if constexpr (Epi::NQ > 0) {
      if (threadIdx.x == 0) {  // one row: thread 0's accumulators are the workgroup's
#pragma unroll
        for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nparts + blk] = acc[q];
      }
    }
    return;
  }
  double* psum        = jag_lds + jag_window(WAVES);  // row sums of the workgroup's rows, natural order
  const int g         = blk * WAVES + wave;
  const int row0      = J.row0[blk];
  const int brows     = J.row0[blk + 1] - row0;
  const int wbase     = J.win[2 * blk];
  const unsigned wlen = (unsigned)J.win[2 * blk + 1];
  // fill the LDS column set, eight requests per thread in flight (a plain loop pays one round trip per element)
  constexpr int kFill = 8, T = WAVES * 64;
  if (wlen) {
    for (unsigned b0 = 0; b0 < wlen; b0 += kFill * T) {
      double v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const unsigned i = b0 + u * T + threadIdx.x;
        v[u] = i < wlen ? vec[wbase + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const unsigned i = b0 + u * T + threadIdx.x;
        if (i < wlen) xwin[i] = v[u];
      }
    }
  } else {
    const int s0 = J.set_ptr[blk], ns = J.set_ptr[blk + 1] - s0;
    for (int b0 = 0; b0 < ns; b0 += kFill * T) {
      int c[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int i = b0 + u * T + (int)threadIdx.x;
        c[u] = i < ns ? __builtin_nontemporal_load(J.set_col + s0 + i) : -1;
      }
      double v[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) v[u] = c[u] >= 0 ? vec[c[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int i = b0 + u * T + (int)threadIdx.x;
        if (i < ns) xwin[i] = v[u];
      }
    }
  }
  for (int i = threadIdx.x; i < brows; i += (WAVES * 64)) psum[i] = 0.0;  // rows without nonzeros
  __syncthreads();
  // rows longer than kLongRow belong to their own workgroups (above): mark them so that the epilogue skips them.  AFTER the
  // barrier: the zero fill above touches the same strip entries from other waves; the passes below never write these entries
  // (they hold rows of <= kLongRow nonzeros only) and the barrier before the epilogue publishes the marks.
  for (int q = J.lr_ptr[blk] + (int)threadIdx.x; q < J.lr_ptr[blk + 1]; q += (WAVES * 64))
    psum[J.lr_row[q] - row0] = __longlong_as_double(kJagNotMine);
  {
    int e         = __builtin_amdgcn_readfirstlane(J.tile_e[g]);
    const int sr0 = __builtin_amdgcn_readfirstlane(J.tile_sr[g]);
    const int ns  = __builtin_amdgcn_readfirstlane(J.tile_sr[g + 1]) - sr0;
    for (int p0 = 0; p0 < ns; p0 += 64) {
      const int i      = p0 + lane;
      const bool have  = i < ns;
      const unsigned d = have ? J.sr[sr0 + i] : 0u;
      const int cnt    = have ? (int)(d >> 16) + 1 : 0;
      const int lrow   = (int)(d & 0xFFFFu);
      double sum       = 0.0;
      const int kmax   = __builtin_amdgcn_readfirstlane(cnt);  // sorted: lane 0 holds the longest row of the pass
      for (int k0 = 0; k0 < kmax; k0 += kJagU) {
        int at[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) {  // diagonal k holds one entry per row longer than k: a prefix of the lanes
          at[u] = e;
          e += __builtin_popcountll(__ballot(cnt > k0 + u));
        }
        double a[kJagU];
        unsigned j[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) {
          a[u] = 0.0, j[u] = 0;
          if (cnt > k0 + u) {
            a[u] = __builtin_nontemporal_load(J.val + at[u] + lane);
            j[u] = __builtin_nontemporal_load(J.slot + at[u] + lane);
          }
        }
        // every gather is an LDS read.  Lanes past their row's end add +0.0 * 0.0: a sum that started at +0.0 is never
        // -0.0, so this changes no bit
        double xv[kJagU];
#pragma unroll
        for (int u = 0; u < kJagU; ++u) xv[u] = cnt > k0 + u ? xwin[j[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < kJagU; ++u) sum = sum + a[u] * xv[u];
      }
      if (have) psum[lrow] = sum;
    }
  }
  __syncthreads();  // the strip is complete: the fused epilogue streams the workgroup's rows in natural order
  for (int i = threadIdx.x; i < brows; i += (WAVES * 64)) {
    const int row = row0 + i;
    if (row < J.rows && __double_as_longlong(psum[i]) != kJagNotMine) epi.row(row, dense_plus(J.dense_add, row, psum[i]), acc);
  }
  if constexpr (Epi::NQ > 0) {
    __syncthreads();  // every wave is done with the window: its first bytes become the reduction scratch
    block_reduce<typename Epi::Op, Epi::NQ, WAVES>(acc, xwin);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * nparts + blk] = acc[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Gather-free layout ("pb": products, then rows) for UNSTRUCTURED matrices whose gathered vector is far beyond the caches.
// Harness and measurements: tools/spmv_pb.hip, profiles/r03_pb_*.  out = M v in two pure streams, no global gather at all:
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
};

// phase P of one workgroup.  xs: LDS, 1 << panel_shift doubles.
template <int THREADS>
__device__ __forceinline__ void pb_products_block(const PbView& V, const double* __restrict__ vec, double* xs)
{
  constexpr int U = 4;
  const int w     = xcd_remap((int)blockIdx.x, V.nwg);
  if (w >= V.nwg) return;
  const int panel = V.wg_panel[w];
  const int c0    = panel << V.panel_shift;
  const int len   = min(1 << V.panel_shift, V.cols - c0);
  constexpr int kFill = 8;
  for (int b0 = 0; b0 < len; b0 += kFill * THREADS) {
    double v[kFill];
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int i = b0 + u * THREADS + (int)threadIdx.x;
      v[u]        = i < len ? vec[c0 + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int i = b0 + u * THREADS + (int)threadIdx.x;
      if (i < len) xs[i] = v[u];
    }
  }
  __syncthreads();
  const int e0 = V.wg_e0[w], e1 = V.wg_e0[w + 1];
  const int gmask = (1 << V.gshift) - 1;
  for (int e = e0 + 2 * (int)threadIdx.x; e < e1; e += 2 * THREADS * U) {
    pb_vec2d a[U];
    uint32_t j[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      a[u] = (pb_vec2d)(0.0), j[u] = 0, dst[u] = 0;
      if (ee < e1) {
        a[u]   = __builtin_nontemporal_load(reinterpret_cast<const pb_vec2d*>(V.val + ee));
        j[u]   = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(V.lidx + ee));
        dst[u] = __builtin_nontemporal_load(V.piece_dst + (ee >> V.gshift));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      if (ee < e1) {
        pb_vec2d p;
        p.x = a[u].x * xs[j[u] & 0xFFFFu];
        p.y = a[u].y * xs[j[u] >> 16];
        *reinterpret_cast<pb_vec2d*>(V.prod + (((int64_t)dst[u] << V.gshift) + (ee & gmask))) = p;
      }
    }
  }
}

// phase R of one workgroup (kPbThreads threads, kPbLdsBytes of dynamic LDS at `lds`)
template <class Epi>
__device__ __forceinline__ void pb_rows_block(const PbView& V, Epi& epi, double* __restrict__ partials, double* lds)
{
  constexpr int THREADS = kPbThreads, WAVES = THREADS / 64, GR = kPbMaxRows / 64 / WAVES, U = (kPbCap / 2 + THREADS - 1) / THREADS;
  static_assert(GR * WAVES * 64 == kPbMaxRows, "whole groups per wave");
  double* lp       = lds;
  double* strip    = lds + kPbCap + 128;
  const int lane   = threadIdx.x & 63;
  const int wave   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b      = xcd_remap((int)blockIdx.x, V.B);
  if (b >= V.B) return;
  const int e0     = V.bin_e0[b];
  const int nunits = (V.bin_e0[b + 1] - e0) >> 1;  // 16-byte units of the image
  const double* __restrict__ prod = V.prod;
  pb_vec2d stage[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int un = threadIdx.x + u * THREADS;
    stage[u]     = (pb_vec2d)(0.0);
    if (un < nunits) stage[u] = __builtin_nontemporal_load(reinterpret_cast<const pb_vec2d*>(prod + e0 + 2 * un));
  }
  const int row0  = V.bin_row0[b];
  const int brows = V.bin_row0[b + 1] - row0;
  const int ng    = (brows + 63) >> 6;
  uint32_t d[GR];
  int eg[GR];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int g = wave + q * WAVES;
    const int i = g * 64 + lane;
    d[q]        = i < brows ? V.sr[row0 + i] : 0u;
    eg[q]       = g < ng ? V.grp_pos[V.bin_grp[b] + g] : 0;
  }
  uint32_t p[GR][kPbKU];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int cnt = (int)(d[q] >> 16);
    int e         = __builtin_amdgcn_readfirstlane(eg[q]);
#pragma unroll
    for (int u = 0; u < kPbKU; ++u) {
      const int at = e;
      e += __builtin_popcountll(__ballot(cnt > u));
      p[q][u] = 0;
      if (cnt > u) p[q][u] = __builtin_nontemporal_load(V.pos + at + lane);
    }
    eg[q] = e;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int un = threadIdx.x + u * THREADS;
    if (un < nunits) *reinterpret_cast<pb_vec2d*>(lp + 2 * un) = stage[u];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int g = wave + q * WAVES;
    if (g < ng) {
      const int cnt  = (int)(d[q] >> 16);
      const int lrow = (int)(d[q] & 0xFFFFu);
      double sum     = 0.0;
#pragma unroll
      for (int u = 0; u < kPbKU; ++u)
        if (cnt > u) sum = sum + lp[p[q][u]];
      const int kmax = __builtin_amdgcn_readfirstlane(cnt);  // sorted: lane 0 holds the longest row of the group
      int e          = eg[q];
      for (int k = kPbKU; k < kmax; ++k) {  // rows longer than the prefetched diagonals
        const int at = e;
        e += __builtin_popcountll(__ballot(cnt > k));
        if (cnt > k) sum = sum + lp[V.pos[at + lane]];
      }
      if (g * 64 + lane < brows) strip[lrow] = sum;
    }
  }
  __syncthreads();
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  for (int i = threadIdx.x; i < brows; i += THREADS) epi.row(row0 + i, dense_plus(V.dense_add, row0 + i, strip[i]), acc);
  if constexpr (Epi::NQ > 0) {
    __syncthreads();  // every wave is done with the image: its first bytes become the reduction scratch
    block_reduce<typename Epi::Op, Epi::NQ, WAVES>(acc, lp);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * V.B + b] = acc[q];
    }
  }
}

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

}  // namespace pdlp
