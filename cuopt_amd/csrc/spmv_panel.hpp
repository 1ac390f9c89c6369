// The panel layout's SpMV skeleton (device code).  Included by kernels_panel.hip only; the layout's view struct and constants, which the
// host side needs too, stay in pdlp_kernels.hpp.
#pragma once
#include "pdlp_kernels.hpp"

namespace pdlp {

// ---- the chunks of a panel, one stage ahead ------------------------------------------------------------------------
// A chunk is <= kPanelChunk consecutive nonzeros of one tile, handled in R = ceil(len / 512) rounds (lane <-> nonzero;
// only the last round has idle lanes: they re-read the chunk's last nonzero and never store).  The value / column
// loads and the packed 16-bit row extents of chunk i+1 are requested before the row sums of chunk i, so one of the two
// dependent memory round trips of a chunk (matrix stream -> gather) always overlaps LDS work of the same workgroup.
// Stages are straight-line code selected by a switch on R (exact load accounting, no wasted gathers).
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
  if (P.dn_own_seg) {  // the row's dense segments: values and vector are coalesced streams, 8 entries per thread in flight
    for (int sg = P.dn_own_seg[w - NP]; sg < P.dn_own_seg[w - NP + 1]; ++sg) {
      const int len = P.dn_seg_len[sg];
      const double* __restrict__ a = P.dn_val + P.dn_seg_ptr[sg];
      const double* __restrict__ x = vec + P.dn_seg_c0[sg];
      constexpr int kSegU = 8;
      for (int k = (int)threadIdx.x; k < len; k += kSegU * kPanelThreads) {
        double av[kSegU], xv[kSegU];
#pragma unroll
        for (int u = 0; u < kSegU; ++u) {
          const int i = k + u * kPanelThreads;
          av[u] = i < len ? __builtin_nontemporal_load(a + i) : 0.0;
          xv[u] = i < len ? x[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kSegU; ++u) part[0] = part[0] + av[u] * xv[u];
      }
    }
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
// the dense segments that reach into a panel of the column side, in LDS: first column, length, where the values start, and the
// gathered vector's entry of the segment's row
struct PanelDenseSegs {
  int n;
  int c0[kPanelDenseSegs], len[kPanelDenseSegs], ptr[kPanelDenseSegs];
  double y[kPanelDenseSegs];
};
__device__ __forceinline__ void panel_dense_prepare(const PanelView& P, int w, const double* __restrict__ vec, PanelDenseSegs* D)
{
  if (!P.dn_pan_ptr) return;  // (uniform)
  const int q0 = P.dn_pan_ptr[w], n = P.dn_pan_ptr[w + 1] - q0;
  if (threadIdx.x == 0) D->n = n;
  if ((int)threadIdx.x < n) {
    const int sg = P.dn_pan_seg[q0 + threadIdx.x];
    D->c0[threadIdx.x] = P.dn_seg_c0[sg], D->len[threadIdx.x] = P.dn_seg_len[sg], D->ptr[threadIdx.x] = P.dn_seg_ptr[sg];
    D->y[threadIdx.x]  = vec[P.dn_seg_row[sg]];
  }
}
// what the dense segments contribute to column j of the panel, ascending rows: the sum k_dense_cols formed.  Added in the epilogue
// (as cheap as the kernel it replaces, one launch less; starting the rows' LDS sums from it at the head of the kernel instead was
// measured 5 us SLOWER: one more barrier in front of the first chunk)
__device__ __forceinline__ double panel_dense_col(const PanelView& P, const PanelDenseSegs* D, int j)
{
  double add   = 0.0;
  const int ns = D->n;
  for (int q = 0; q < ns; ++q) {
    const int k = j - D->c0[q];
    if (k >= 0 && k < D->len[q]) add += __builtin_nontemporal_load(P.dn_val + D->ptr[q] + k) * D->y[q];
  }
  return add;
}
// the fused epilogue over a panel's rows, natural order, from the row sums in LDS
template <class Epi>
__device__ __forceinline__ void panel_epilogue(const PanelView& P, Epi& epi, double* __restrict__ partials, const double* psum, double* red,
                                               int w, int r0, int nr, const PanelDenseSegs* D,
                                               const double* psum2 = nullptr /* long-tail variant: the edge runs' share */)
{
  double acc[Epi::NQ > 0 ? Epi::NQ : 1];
#pragma unroll
  for (int q = 0; q < (Epi::NQ > 0 ? Epi::NQ : 1); ++q) acc[q] = Epi::Op::identity();
  for (int r = threadIdx.x; r < nr; r += kPanelThreads)
    if (__double_as_longlong(psum[r]) != kPanelNotMine) {
      double v = psum2 ? psum[r] + psum2[r] : psum[r];
      if (P.dn_pan_ptr) v = v + panel_dense_col(P, D, r0 + r);
      epi.row(r0 + r, dense_plus(P.dense_add, r0 + r, v), acc);
    }
  if constexpr (Epi::NQ > 0) {
    block_reduce<typename Epi::Op, Epi::NQ, kPanelWaves>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < Epi::NQ; ++q) partials[(size_t)q * P.W + w] = acc[q];
    }
  }
}

// LDS hazards of panel_spmv_block (round-6 audit; "reader -> next writer: the barrier between them"):
//   prod[]  (ONE buffer; the chunk "one stage ahead" lives in REGISTERS: va / ja / ext_next)
//            written by panel_products of chunk i between barrier T(i) [loop top] and barrier M(i) [after the next chunk's requests];
//            read by the row sums of chunk i after M(i); next written by chunk i + 1 after T(i + 1) -- every row sum of chunk i ends
//            before its thread arrives at T(i + 1).
//   psum[]  zeroed before the first barrier; row r is read and written by lane r % 512 ONLY (also in the long-segment path: the lane
//            that owns the row adds the wave's partial), the "not mine" marks are written after the first barrier to rows that have no
//            segment in any tile; the epilogue reads behind the barrier after the loop.
//   tile_s / base_s  written once before the first barrier, read-only afterwards.
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
  __shared__ PanelDenseSegs dseg;
  const int r0 = P.row0[w], nr = P.row0[w + 1] - r0;
  panel_dense_prepare(P, w, vec, &dseg);  // (visible after the barriers of the chunk loop / before the epilogue)
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
  panel_epilogue(P, epi, partials, psum, red, w, r0, nr, &dseg);
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
// LDS hazards of panel_seg_block (round-6 audit):
//   rec[2][]  the edge-run records, double-buffered by chunk parity.  rec[p] is written by every wave during the rounds of chunk i
//             (parity p), read by wave 0's join after the chunk's barrier B(i), and next written during the rounds of chunk i + 2,
//             i.e. after B(i + 1) -- wave 0 finishes its join of chunk i, in program order, before it arrives at B(i + 1).
//   psum[]    LDS atomics (ds_add_f64) from every wave, at most ONE emission per row and chunk (a row's interior run lies inside one
//             wave-round; its edge runs go through the records), so two chunks' emissions to one row are separated by a barrier and
//             the order of a row's additions is fixed.
//   psum2[]   emitted to by wave 0 alone (the joined edge runs), in program order.
//   the epilogue reads psum + psum2 behind the barrier after the loop.
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
  __shared__ PanelDenseSegs dseg;
  const int r0 = P.row0[w], nr = P.row0[w + 1] - r0;
  panel_dense_prepare(P, w, vec, &dseg);
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
  panel_epilogue(P, epi, partials, psum, red, w, r0, nr, &dseg, psum2);
}
#undef PANEL_DISPATCH
template <bool SEG, class Epi>
__device__ __forceinline__ void panel_block(const PanelView& P, const double* __restrict__ vec, Epi& epi, double* __restrict__ partials)
{
  if constexpr (SEG) panel_seg_block(P, vec, epi, partials);
  else panel_spmv_block(P, vec, epi, partials);
}

}  // namespace pdlp
