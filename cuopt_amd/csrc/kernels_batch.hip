// Shared-matrix batched PDHG (round 5; BASELINE config 5's pattern: the MIP heuristics re-solve the SAME A and c under different bounds,
// cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-127; the reference builds a solver per call, its batch entry point -- cython_solve.cu:264-296 --
// is a thread pool of independent solves).  K = 2, 4, 8 or 16 LPs that share the matrix advance together:
//   * every LP keeps a full context of its own (pdlpdev_clone_shared: the matrices, their layouts, D_r, D_c and c are the parent's,
//     read-only; iterates, bounds, sums, control block, partials are the clone's): the major iterations -- KKT evaluation, restarts,
//     the primal weight -- run through the single-LP code, untouched, LP by LP;
//   * an attempt is FOUR launches for all K LPs: kb_primal (the K primal steps, xbar written interleaved), kb_a_dual and kb_at_step
//     (the two products, fused with the K dual steps / step-size sums), k_step_decision_batch (K workgroups, each k_step_decision);
//   * the two products serve all K LPs from ONE pass over the matrix.  The K gathered vectors are INTERLEAVED (v[j * K + l]): a
//     nonzero's gather is one 64-byte request (K = 8) for all LPs instead of K requests of 8 bytes -- the request rate is what bounds
//     the unstructured single-LP product (profiles/r05_gather_calibration.txt);
//   * the trajectories are BIT-IDENTICAL to the single solves': rows are summed left to right (the single kernels' order for rows
//     of <= 128 entries), the fused epilogues are the single kernels' expressions, and the per-workgroup partial sums reproduce the
//     panel kernels' grouping exactly -- workgroup <-> row panel (the single layout's own boundaries), lane t of an LP's wave
//     accumulates rows t, t + 512, ... in ascending order, the same wave tree (ds_swizzle butterflies) and wave-by-wave sum as
//     block_reduce (CSR stream layout: workgroup <-> row block, 256 "threads", four waves).  Hence the restriction: each matrix in
//     the row-sum variant of the panels or in the CSR stream layout, no row beyond 128 entries, no dense segments, columns ascending
//     within rows (else: not eligible, the caller keeps its independent solves).
// What it buys (C3, 1e6 x 1e6, 1e7 nonzeros; profiles/r05_bench_lines.jsonl, c3_batch16 / c3_batch8): 14.7 k iterations/s aggregate
// over 16 LPs, 10.7 k over 8, against 6.0 k for one -- 2.43x / 1.77x; K = 4: 1.24x; K = 2: 0.80x (two single solves are faster).
// Why not more: only the MATRIX is shared.  A lockstep iteration of 8 LPs moves 1.85 GB at the fused floor (0.24 GB of matrix once,
// 8 x 0.18 GB of vectors, the interleaved copies) against 0.42 GB for one LP: at EQUAL fractions of the HBM roofline the ceiling is
// 8 x 0.42 / 1.85 = 1.80x (16 LPs: 1.92x); the single solve and the batch of 8 run at 0.31 of their floors, the batch of 16 at 0.40.
// The K = 8 products sit at ~300 us whatever their internal structure (row walk / LDS-staged chunks / autonomous waves, 1 to 32
// gathers in flight, 2 or 4 workgroups per CU: tools/batch_spmv_probe.hip, profiles/r05_batch_spmv_probe.txt): 1e7 gathered
// 128-byte lines from beyond L2 (the interleaved vector is 64 MB; an XCD's L2 holds 4) + 0.5 GB of streams ~ 1.8 GB at ~6 TB/s --
// K = 16 uses the whole line a miss fetches.  The single-LP panels avoid those line fills by sweeping 1.33 MB column slabs in step
// across the chip; eight interleaved vectors would need 46 slabs and a sweep synchronised to +-3 %: with the epilogue phases in
// between it does not hold (window-major orders in the probe: no gain once the epilogue is in).
#include <cstring>
#include <hip/hip_runtime.h>

#include "pdlp_ctx.hpp"

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// single-LP kernels of the core (pdlp_device.hip), launched per LP
__global__ void k_step_decision_batch(const pdlpdev_decision_args* __restrict__ args);
__global__ void k_set_target(pdlpdev_ctl* ctl, int target);
__global__ void k_set_error(pdlpdev_ctl* ctl);

namespace {

constexpr int kBT = kPanelThreads;  // threads per workgroup = the panel kernels' (the partial sums reproduce their tree)
static_assert(kBT == 512, "8 waves of 64 lanes: the reduction below mirrors block_reduce<.., kPanelWaves>");
constexpr int kBatchMax = 16;       // LPs per batch

struct BatchLp {  // what the batched kernels need of one LP, in device memory
  pdlpdev_ctl* ctl;
  double *y0, *y1, *sumy;
  const double *lo, *hi;
  double *x0, *x1, *aty0, *aty1, *sumx;
  const double *c, *lb, *ub;
  pdlpdev_ctx::UniformBounds ubd;
  double *part_a, *part_at;
};
static_assert(sizeof(BatchLp) == 16 * sizeof(void*) + 2 * sizeof(int) + 2 * sizeof(double), "no padding: batch_refresh_table compares entries with memcmp");

// wave <-> LP in the element-wise phases: K <= 8: 8 / K waves share an LP (wave w: LP w % K, every (8 / K)-th piece of 64 rows from
// piece w / K on); K = 16: a wave serves two LPs one after the other (w and w + 8)
template <int K>
struct WaveLps {
  static constexpr int PASSES = K > 8 ? K / 8 : 1;  // LPs per wave
  static constexpr int NSUB   = K > 8 ? 1 : 8 / K;  // waves per LP
  static __device__ __forceinline__ int lp(int wave, int pass) { return K > 8 ? wave + 8 * pass : wave % K; }
  static __device__ __forceinline__ int sub(int wave) { return K > 8 ? 0 : wave / K; }
};

// ---- (1) the primal step of K LPs (k_primal's expressions, LP by LP) with xbar written INTERLEAVED ------------------------------
// wave <-> LP, lane <-> column: every per-LP stream is read and written in 512-byte pieces; the tile of xbar goes through LDS (one
// padded row per column) and leaves as whole entries of the interleaved vector.
template <int K>
__global__ void __launch_bounds__(kBT) kb_primal(const BatchLp* __restrict__ lp, int n, double* __restrict__ xK)
{
  __shared__ double tile[kBT][K + 1];
  using WL = WaveLps<K>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = WL::sub(wave);
  for (int t0 = blockIdx.x * kBT; t0 < n; t0 += gridDim.x * kBT) {
#pragma unroll
    for (int pass = 0; pass < WL::PASSES; ++pass) {
      const int l       = WL::lp(wave, pass);
      const BatchLp L   = lp[l];
      const bool active = loop_active(L.ctl);
      const int cur       = L.ctl->cur;
      const double tau    = L.ctl->tau;
      const double weight = L.ctl->step_size;
      const bool pend     = L.ctl->pending_avg != 0;
      const double* __restrict__ x   = cur ? L.x1 : L.x0;
      double* __restrict__ xn        = cur ? L.x0 : L.x1;
      const double* __restrict__ aty = cur ? L.aty1 : L.aty0;
#pragma unroll
      for (int q = sub; q < 8; q += WL::NSUB) {
        const int jl = lane + 64 * q, j = t0 + jl;
        double xb    = 0.0;
        if (active && j < n) {
          const double xj       = x[j];
          const double gradient = L.c[j] - aty[j];
          double next           = xj - (tau * gradient);
          next                  = dmax(dmin(next, L.ubd.ub_same ? L.ubd.ub : L.ub[j]), L.ubd.lb_same ? L.ubd.lb : L.lb[j]);
          xn[j]                 = next;
          xb                    = next - xj + next;
          if (pend) L.sumx[j] = L.sumx[j] + weight * xj;
        }
        tile[jl][l] = xb;
      }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < kBT * K; f += kBT) {
      const int jl = f / K;
      if (t0 + jl < n) xK[(size_t)t0 * K + f] = tile[jl][f % K];
    }
    __syncthreads();
  }
}

// ---- (2), (3) the two products for K LPs ----------------------------------------------------------------------------------------
// The panel kernels' structure, K wide.  A workgroup owns the rows of ONE panel of the single-LP layout (same boundaries: the
// partial sums below are then the panel kernels' own) and walks them in blocks of 512 rows.  Per block the matrix entries -- one
// contiguous CSR range -- are read coalesced, a chunk at a time, and handed round through LDS; a GROUP of K / 2 lanes fetches one
// entry's K vector values (lane h: the LPs 2h and 2h + 1, one 16-byte load; K = 8: one 64-byte request per entry, K = 16: the whole
// 128-byte line a miss fetches anyway) and leaves the K products in LDS; lane (g, h) then adds the products of ITS rows (g, g + G,
// ...) left to right in registers -- the CSR order, the order every single-LP kernel uses for rows of up to 128 entries.  One
// barrier per chunk, two chunks of gathers in flight.  The fused epilogue runs wave <-> LP, lane <-> row (the row sums cross over
// through LDS): every per-LP stream is read and written in 512-byte pieces, and lane t of an LP's wave holds exactly the panel
// kernels' "thread t" accumulators (rows t, t + 512, ... of the panel in ascending order), so the wave tree and the wave-by-wave
// sum of block_reduce apply unchanged.
template <int K>
struct BatchGeometry {
  static constexpr int KL    = K / 2;              // lanes per entry
  static constexpr int G     = kBT / KL;           // groups per workgroup
  static constexpr int CHUNK = K > 8 ? 256 : 512;  // matrix entries staged per pass (64 KB of products in two buffers)
  static constexpr int PER   = CHUNK / G;          // entries per lane and chunk
  static constexpr int RU    = kBT / G;            // rows per lane and block of 512 rows
};
template <int K>
struct alignas(16) BatchShared {
  using Geo = BatchGeometry<K>;
  union {
    struct {
      double prod[2][Geo::CHUNK][K];
      int scol[2][Geo::CHUNK];
      double sval[2][Geo::CHUNK];
    } p;
    double sums[kBT][K + 1];  // the epilogue's view of a block: row sums / new iterates, one padded row per matrix row
  } u;
  double red[2][K][8];
};
static_assert(sizeof(BatchShared<8>) <= 80 * 1024 && sizeof(BatchShared<16>) <= 80 * 1024, "two workgroups per CU");

// row sums of the block [b0, b0 + 512) of panel rows [r0, r0 + nr): lane (g, h) -- group g of K / 2 lanes, lane h of it = the LPs 2h and
// 2h + 1 -- ends with s[u][0..1] = the sums of row b0 + g + G * u for its two LPs
// LDS hazards of batch_block_sums (round-6 audit; the stage round 5's contention run had caught one barrier short):
//   prod[2][], scol[2][], sval[2][]  double-buffered by chunk parity.  Trip c (ends in barrier E(c)): reads scol / sval[(c + 1) & 1]
//   (chunk c + 1's entries, written in trip c - 1), reads prod[(c - 1) & 1] (chunk c - 1's products, written in trip c - 1), writes
//   prod[c & 1] (last read by the row sums of chunk c - 2 in trip c - 1, before E(c - 1)) and scol / sval[c & 1] with chunk c + 2's
//   entries (last read by trip c - 1's requests for chunk c, before E(c - 1)).  Every write is separated from the last read of its slot
//   by E(c - 1), every read from the write it depends on by E(c - 1) as well; the barrier in front of trip 0 covers the two staged chunks.
template <int K>
__device__ __forceinline__ void batch_block_sums(BatchShared<K>& S, int r0, int nr, int b0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                 const double* __restrict__ val, const double* __restrict__ vK, double (&s)[BatchGeometry<K>::RU][2])
{
  using Geo = BatchGeometry<K>;
  constexpr int KL = Geo::KL, G = Geo::G, PER = Geo::PER, RU = Geo::RU, CH = Geo::CHUNK;
  const int tid = threadIdx.x, h = tid % KL, g = tid / KL;
  int k0[RU], k1[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) {
    const int r = b0 + g + G * u;
    const int i = r0 + (r < nr ? r : 0);
    k0[u] = off[i];
    k1[u] = r < nr ? off[i + 1] : k0[u];
    s[u][0] = 0.0, s[u][1] = 0.0;
  }
  const int eb0 = off[r0 + b0], eb1 = off[r0 + (b0 + kBT < nr ? b0 + kBT : nr)];
  const int nch = (eb1 - eb0 + CH - 1) / CH;
  const bool stager = CH == kBT || tid < CH;  // (chunks of 256: the first four waves fetch and stage)
  // chunks 0 and 1 staged (past the block's last entry: column 0 with value 0 -- gathered, multiplied, never added);
  // chunk 0's gathers on their way
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int e   = eb0 + c * CH + tid;
    const bool in = stager && e < eb1;
    const int cl  = idx[in ? e : eb0];
    const double vl = val[in ? e : eb0];
    if (stager) S.u.p.scol[c][tid] = in ? cl : 0, S.u.p.sval[c][tid] = in ? vl : 0.0;
  }
  __syncthreads();
  auto rowsum = [&](int cc) {
    const int c0 = eb0 + cc * CH, c1 = c0 + CH < eb1 ? c0 + CH : eb1, pb = cc & 1;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int a = k0[u] > c0 ? k0[u] : c0, e = k1[u] < c1 ? k1[u] : c1;
      for (int k = a; k < e; ++k) {
        const double2 p = *(const double2*)&S.u.p.prod[pb][k - c0][2 * h];
        s[u][0] = s[u][0] + p.x, s[u][1] = s[u][1] + p.y;
      }
    }
  };
  double2 pv[PER], pvn[PER];
  double sv[PER], svn[PER];
  auto request = [&](int cc, double2 (&p)[PER], double (&v)[PER]) {  // chunk cc: its entries' values, its gathers
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i] = S.u.p.sval[cc & 1][g + G * i];
      p[i] = *(const double2*)(vK + ((unsigned)S.u.p.scol[cc & 1][g + G * i] * (unsigned)K + 2u * h));
    }
  };
  // the entries of chunk c + 2 wait in registers for one trip before they go to LDS: every wait below is for loads issued a whole
  // trip earlier (the vector memory counter completes in order: a load consumed in the trip that issued it would drag the trip's
  // gathers along)
  auto fetch = [&](int cc, int& col, double& v) {
    const int e   = eb0 + cc * CH + tid;
    const bool in = stager && e < eb1;
    const int cl  = idx[in ? e : eb0];
    const double vl = val[in ? e : eb0];
    col = in ? cl : 0, v = in ? vl : 0.0;
  };
  int colA = 0, colB = 0;
  double valA = 0.0, valB = 0.0;
  // one trip: chunk c's products (its gathers were issued a trip ago into `cur`), chunk c + 1's gathers into `nxt`, chunk c + 2's
  // entries from their registers to LDS, chunk c + 3's entries requested.  (Two register sets that swap roles, the loop unrolled by
  // two: a copy "cur = nxt" at the end of a trip would wait for the gathers it has just issued.)
  auto trip = [&](int c, double2 (&cur)[PER], double (&curv)[PER], double2 (&nxt)[PER], double (&nxtv)[PER], int& col_st, double& val_st, int& col_ld,
                  double& val_ld) {
    request(c + 1, nxt, nxtv);  // (past the last chunk: staged zeros -- column 0, value 0; no branch around the loads: the counter
    fetch(c + 3, col_ld, val_ld);  //  bookkeeping of the compiler stays exact only in straight-line code)
    rowsum(c - 1);                // (c = 0: an empty range)
#pragma unroll
    for (int i = 0; i < PER; ++i) *(double2*)&S.u.p.prod[c & 1][g + G * i][2 * h] = double2{curv[i] * cur[i].x, curv[i] * cur[i].y};
    if (stager) S.u.p.scol[c & 1][tid] = col_st, S.u.p.sval[c & 1][tid] = val_st;  // (chunk c + 2 takes chunk c's place: read one barrier ago)
    __syncthreads();
  };
  request(0, pv, sv);
  fetch(2, colA, valA);
  __syncthreads();  // (trip 0 puts chunk 2 where chunk 0's entries are: every lane has read them first)
  for (int c = 0; c < nch; c += 2) {
    trip(c, pv, sv, pvn, svn, colA, valA, colB, valB);
    if (c + 1 < nch) trip(c + 1, pvn, svn, pv, sv, colB, valB, colA, valA);
  }
  if (nch > 0) rowsum(nch - 1);
}

// the row sums of a block cross over: lane (g, h) -> sums[row][LP] (padded rows), for the epilogue's wave <-> LP, lane <-> row
template <int K>
__device__ __forceinline__ void batch_cross_over(BatchShared<K>& S, const double (&s)[BatchGeometry<K>::RU][2])
{
  using Geo = BatchGeometry<K>;
  const int h = threadIdx.x % Geo::KL, g = threadIdx.x / Geo::KL;
  __syncthreads();  // (the last chunk's products are read)
#pragma unroll
  for (int u = 0; u < Geo::RU; ++u) S.u.sums[g + Geo::G * u][2 * h] = s[u][0], S.u.sums[g + Geo::G * u][2 * h + 1] = s[u][1];
  __syncthreads();
}

// block_reduce<SumOp, NQ, VW> of the single-LP kernels for every LP: acc[pass][q][v] = the sums of virtual threads lane + 64 v of LP
// (wave, pass).  VW = 8: the panel kernels' 512 threads; VW = 4: the CSR stream kernels' 256 (row t of a block belongs to thread t mod 256).
template <int K, int NQ, int VW>
__device__ __forceinline__ void batch_block_partials(BatchShared<K>& S, const double (&acc)[WaveLps<K>::PASSES][NQ][8], const BatchLp* __restrict__ lp, bool a_side,
                                                     int W, int w)
{
  using WL = WaveLps<K>;
  static_assert(WL::NSUB <= VW, "a wave owns whole virtual waves");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = WL::sub(wave);
#pragma unroll
  for (int pass = 0; pass < WL::PASSES; ++pass)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int v = 0; v < VW; ++v)
        if (v % WL::NSUB == sub) {  // (rows lane + 64 j, j = sub mod NSUB, are this wave's: virtual waves j mod VW)
          const double r = wave_reduce<SumOp>(acc[pass][q][v]);
          if (lane == 0) S.red[q][WL::lp(wave, pass)][v] = r;
        }
  __syncthreads();
  if (threadIdx.x < K * NQ) {
    const int q = threadIdx.x / K, ll = threadIdx.x % K;
    if (loop_active(lp[ll].ctl)) {
      double total = S.red[q][ll][0];
      for (int vw = 1; vw < VW; ++vw) total = total + S.red[q][ll][vw];
      (a_side ? lp[ll].part_a : lp[ll].part_at)[(size_t)q * W + w] = total;
    }
  }
}

// rows of A for K LPs: y' = proj(y - sigma A xbar), ||dy||^2 partials, the deferred dual averaging (DualEpilogue, pdlp_epilogues.hpp);
// y' also goes, interleaved, to the vector the column side gathers from
template <int K, int VW>
__global__ void __launch_bounds__(kBT) kb_a_dual(int W, const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                 const double* __restrict__ val, const BatchLp* __restrict__ lp, const double* __restrict__ xK,
                                                 double* __restrict__ yK)
{
  __shared__ BatchShared<K> S;
  using WL = WaveLps<K>;
  const int w = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = WL::sub(wave);
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  double acc[WL::PASSES][1][8];
#pragma unroll
  for (int pass = 0; pass < WL::PASSES; ++pass)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[pass][0][j] = 0.0;
  for (int b0 = 0; b0 < nr; b0 += kBT) {
    double s[BatchGeometry<K>::RU][2];
    batch_block_sums<K>(S, r0, nr, b0, off, idx, val, xK, s);
    batch_cross_over<K>(S, s);
#pragma unroll
    for (int pass = 0; pass < WL::PASSES; ++pass) {
      const int l      = WL::lp(wave, pass);
      const BatchLp L  = lp[l];
      if (!loop_active(L.ctl)) continue;
      const int cur      = L.ctl->cur;
      const double sigma = L.ctl->sigma, weight = L.ctl->step_size;
      const bool pend    = L.ctl->pending_avg != 0;
      const double* __restrict__ y = cur ? L.y1 : L.y0;
      double* __restrict__ yn      = cur ? L.y0 : L.y1;
      double yv[8], lov[8], hiv[8], sy[8];
#pragma unroll
      for (int j = sub; j < 8; j += WL::NSUB) {
        const int r = b0 + lane + 64 * j, i = r0 + (r < nr ? r : 0);
        yv[j] = y[i], lov[j] = L.lo[i], hiv[j] = L.hi[i], sy[j] = pend ? L.sumy[i] : 0.0;
      }
#pragma unroll
      for (int j = sub; j < 8; j += WL::NSUB) {
        const int r = b0 + lane + 64 * j;
        if (r < nr) {
          const int i      = r0 + r;
          const double yi  = yv[j];
          double next      = yi - (sigma * S.u.sums[lane + 64 * j][l]);
          const double low = next + sigma * lov[j];
          const double up  = next + sigma * hiv[j];
          next             = dmax(low, dmin(up, 0.0));
          yn[i]            = next;
          S.u.sums[lane + 64 * j][l] = next;
          const double dy = next - yi;
          acc[pass][0][j % VW] += dy * dy;
          if (pend) L.sumy[i] = sy[j] + weight * yi;
        }
      }
    }
    __syncthreads();
    const int rows = nr - b0 < kBT ? nr - b0 : kBT;
    for (int f = threadIdx.x; f < rows * K; f += kBT) yK[(size_t)(r0 + b0) * K + f] = S.u.sums[f / K][f % K];
    __syncthreads();
  }
  batch_block_partials<K, 1, VW>(S, acc, lp, true, W, w);
}

// rows of A^T for K LPs: AtY' = A^T y', interaction and ||dx||^2 partials (StepEpilogue)
template <int K, int VW>
__global__ void __launch_bounds__(kBT) kb_at_step(int W, const int32_t* __restrict__ row0, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                                  const double* __restrict__ val, const BatchLp* __restrict__ lp, const double* __restrict__ yK)
{
  __shared__ BatchShared<K> S;
  using WL = WaveLps<K>;
  const int w = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = WL::sub(wave);
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  double acc[WL::PASSES][2][8];
#pragma unroll
  for (int pass = 0; pass < WL::PASSES; ++pass)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[pass][0][j] = 0.0, acc[pass][1][j] = 0.0;
  for (int b0 = 0; b0 < nr; b0 += kBT) {
    double s[BatchGeometry<K>::RU][2];
    batch_block_sums<K>(S, r0, nr, b0, off, idx, val, yK, s);
    batch_cross_over<K>(S, s);
#pragma unroll
    for (int pass = 0; pass < WL::PASSES; ++pass) {
      const int l     = WL::lp(wave, pass);
      const BatchLp L = lp[l];
      if (!loop_active(L.ctl)) continue;
      const int cur = L.ctl->cur;
      const double* __restrict__ x   = cur ? L.x1 : L.x0;
      const double* __restrict__ xn  = cur ? L.x0 : L.x1;
      const double* __restrict__ aty = cur ? L.aty1 : L.aty0;
      double* __restrict__ atyn      = cur ? L.aty0 : L.aty1;
      double xv[8], xnv[8], av[8];
#pragma unroll
      for (int j = sub; j < 8; j += WL::NSUB) {
        const int r = b0 + lane + 64 * j, i = r0 + (r < nr ? r : 0);
        xv[j] = x[i], xnv[j] = xn[i], av[j] = aty[i];
      }
#pragma unroll
      for (int j = sub; j < 8; j += WL::NSUB) {
        const int r = b0 + lane + 64 * j;
        if (r < nr) {
          const double v  = S.u.sums[lane + 64 * j][l];
          atyn[r0 + r]    = v;
          const double dx = xnv[j] - xv[j];
          const double t  = v - av[j];
          acc[pass][0][j % VW] += t * dx;
          acc[pass][1][j % VW] += dx * dx;
        }
      }
    }
    __syncthreads();
  }
  batch_block_partials<K, 2, VW>(S, acc, lp, false, W, w);
}

// columns ascending within every row?  (the panels add a row's products slab by slab = by ascending column; the batched products
// walk the CSR row: the same order only then)
__global__ void __launch_bounds__(256) kb_check_sorted(int rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, int* __restrict__ bad)
{
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256)
    for (int k = off[i] + 1; k < off[i + 1]; ++k)
      if (idx[k] <= idx[k - 1]) *bad = 1;
}

}  // namespace

struct pdlpdev_batch {
  int K = 0, device = 0;
  hipStream_t stream = nullptr;
  pdlpdev_ctx* ctx[kBatchMax] = {nullptr};
  BatchLp* lp_dev = nullptr;
  pdlpdev_decision_args* dargs_dev = nullptr;
  std::vector<pdlpdev_decision_args> dargs_host;  // (refreshed with the table: pdlpdev_set_step_params of a member after the batch was made)
  double *xK = nullptr, *yK = nullptr;
  // the row blocks whose partial sums the products reproduce: the panels of the single-LP layout (512 "threads"), or the row blocks
  // of the CSR stream kernels (256)
  struct Side {
    int W = 0;
    const int32_t* row0 = nullptr;
    bool panel = false;
  } a_side, t_side;
  std::vector<BatchLp> lp_host;  // what lp_dev holds
  std::map<int, hipGraphExec_t> graphs;
};

static BatchLp batch_lp_of(const pdlpdev_ctx* c)
{
  return BatchLp{c->ctl, c->y[0], c->y[1], c->sumy, c->lo, c->hi, c->x[0], c->x[1], c->aty[0], c->aty[1], c->sumx, c->c, c->lb, c->ub, c->ubd, c->part_a, c->part_at};
}

// A reset that gives a member row bounds of its own moves its lo / hi (pdlpdev_reset: copy on change): the table the kernels read
// follows (the kernels -- and the captured graphs -- take the table's address, not its contents).
static int batch_refresh_table(pdlpdev_batch* b)
{
  bool changed = false;
  for (int l = 0; l < b->K; ++l) {
    const BatchLp now = batch_lp_of(b->ctx[l]);
    if (memcmp(&now, &b->lp_host[l], sizeof(BatchLp)) != 0) b->lp_host[l] = now, changed = true;
  }
  bool sp_changed = false;  // (the step-size exponents travel by value in the decision kernel's arguments)
  for (int l = 0; l < b->K; ++l)
    if (memcmp(&b->dargs_host[l].sp, &b->ctx[l]->sp, sizeof(pdlpdev_step_params)) != 0) b->dargs_host[l].sp = b->ctx[l]->sp, sp_changed = true;
  if (changed) HIP_TRY(hipMemcpyAsync(b->lp_dev, b->lp_host.data(), b->K * sizeof(BatchLp), hipMemcpyHostToDevice, b->stream));
  if (sp_changed)  // (the attempt graphs carry the POINTER to this table, not its contents: no re-capture)
    HIP_TRY(hipMemcpyAsync(b->dargs_dev, b->dargs_host.data(), b->K * sizeof(pdlpdev_decision_args), hipMemcpyHostToDevice, b->stream));
  if (changed || sp_changed) HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

static int batch_fetch_ctl(pdlpdev_batch* b)
{
  for (int l = 0; l < b->K; ++l)
    HIP_TRY(hipMemcpyAsync(b->ctx[l]->ctl_h, b->ctx[l]->ctl, sizeof(pdlpdev_ctl), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

// (ev != nullptr: the four dispatches carry start / stop events -- hipExtLaunchKernel's own timestamps, no records between them)
template <int K>
static int batch_enqueue_attempt(pdlpdev_batch* b, hipEvent_t* ev = nullptr)
{
  hipStream_t s   = b->stream;
  pdlpdev_ctx* c0 = b->ctx[0];
  int n           = c0->n;
  int aw = b->a_side.W, tw = b->t_side.W;
  const int32_t *arow0 = b->a_side.row0, *trow0 = b->t_side.row0;
  const bool ap = b->a_side.panel, tp = b->t_side.panel;
  const int pgrid = std::min((n + kBT - 1) / kBT, 4096);
  if (ev) {
    void* a0[] = {&b->lp_dev, &n, &b->xK};
    void* a1[] = {&aw, &arow0, &c0->ha_off, &c0->ha_idx, &c0->ha_val, &b->lp_dev, &b->xK, &b->yK};
    void* a2[] = {&tw, &trow0, &c0->hat_off, &c0->hat_idx, &c0->hat_val, &b->lp_dev, &b->yK};
    void* a3[] = {&b->dargs_dev};
    HIP_TRY(hipExtLaunchKernel((const void*)kb_primal<K>, dim3(pgrid), dim3(kBT), a0, 0, s, ev[0], ev[1], 0));
    HIP_TRY(hipExtLaunchKernel(ap ? (const void*)kb_a_dual<K, 8> : (const void*)kb_a_dual<K, 4>, dim3(aw), dim3(kBT), a1, 0, s, ev[2], ev[3], 0));
    HIP_TRY(hipExtLaunchKernel(tp ? (const void*)kb_at_step<K, 8> : (const void*)kb_at_step<K, 4>, dim3(tw), dim3(kBT), a2, 0, s, ev[4], ev[5], 0));
    HIP_TRY(hipExtLaunchKernel((const void*)k_step_decision_batch, dim3(K), dim3(1024), a3, 0, s, ev[6], ev[7], 0));
    return 0;
  }
  kb_primal<K><<<pgrid, kBT, 0, s>>>(b->lp_dev, n, b->xK);
  if (ap) kb_a_dual<K, 8><<<aw, kBT, 0, s>>>(aw, arow0, c0->ha_off, c0->ha_idx, c0->ha_val, b->lp_dev, b->xK, b->yK);
  else kb_a_dual<K, 4><<<aw, kBT, 0, s>>>(aw, arow0, c0->ha_off, c0->ha_idx, c0->ha_val, b->lp_dev, b->xK, b->yK);
  if (tp) kb_at_step<K, 8><<<tw, kBT, 0, s>>>(tw, trow0, c0->hat_off, c0->hat_idx, c0->hat_val, b->lp_dev, b->yK);
  else kb_at_step<K, 4><<<tw, kBT, 0, s>>>(tw, trow0, c0->hat_off, c0->hat_idx, c0->hat_val, b->lp_dev, b->yK);
  k_step_decision_batch<<<K, 1024, 0, s>>>(b->dargs_dev);
  LAUNCH_CHECK();
  return 0;
}

static int batch_enqueue(pdlpdev_batch* b, hipEvent_t* ev = nullptr)
{
  return b->K == 16  ? batch_enqueue_attempt<16>(b, ev)
         : b->K == 8 ? batch_enqueue_attempt<8>(b, ev)
         : b->K == 4 ? batch_enqueue_attempt<4>(b, ev)
                     : batch_enqueue_attempt<2>(b, ev);
}

static int batch_graph(pdlpdev_batch* b, int attempts, hipGraphExec_t* out)
{
  auto it = b->graphs.find(attempts);
  if (it != b->graphs.end()) {
    *out = it->second;
    return 0;
  }
  hipGraph_t graph;
  HIP_TRY(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < attempts && rc == 0; ++i) rc = batch_enqueue(b);
  hipError_t e = hipStreamEndCapture(b->stream, &graph);
  if (rc != 0) return rc;
  HIP_TRY(e);
  hipGraphExec_t exec;
  HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIP_TRY(hipGraphDestroy(graph));
  b->graphs[attempts] = exec;
  *out                = exec;
  return 0;
}

extern "C" {

// A context for ANOTHER LP over the same matrix and objective: matrices, layouts, scaling vectors and c are the parent's (which must
// outlive the clone; resets of either side are fine: row bounds are shared until a reset changes them); the iterate, variable bounds, sums,
// control block and partial buffers are the
// clone's own.  The clone starts with the parent's (scaled) bounds: pdlpdev_reset(clone, lb, ub, lo, hi) gives it its own, bit for
// bit the state of a freshly created context of that LP (tests/test_persistent_resolve_gpu.py pins reset against a fresh solver).
int pdlpdev_clone_shared(pdlpdev_ctx** out, pdlpdev_ctx* parent)
{
  if (!out || !parent) return fail(-1, "pdlpdev_clone_shared: null argument");
  if (!parent->scaled) return fail(-1, "pdlpdev_clone_shared: the parent has not been scaled yet");
  if (parent->comm || parent->small_resident || parent->dense.on) return fail(-7, "pdlpdev_clone_shared: not for sharded, resident or dense-segment contexts");
  HIP_TRY(hipSetDevice(parent->device));
  HIP_TRY(hipStreamSynchronize(parent->stream));
  pdlpdev_ctx* c = new pdlpdev_ctx(*parent);  // (pointers to the shared arrays, geometry, parameters: by value)
  *out           = c;
  c->allocs.clear(), c->graphs.clear(), c->scratch.clear();
  c->shared_with_parent = true;
  c->parent = nullptr;  // (set once the clone is complete: a failed clone is destroyed without touching the parent's count)
  c->scal_h = nullptr, c->ctl_h = nullptr;  // (the parent's pinned block until the clone has its own: a clone that fails below must not free it)
  c->clones_alive = 0, c->batches_alive = 0, c->rows_private = false;
  c->bytes = 0, c->slab = nullptr, c->slab_cap = c->slab_used = 0, c->arena = nullptr, c->arena_used = 0, c->first_chunk = nullptr;
  c->bestx = c->besty = c->bestrc = nullptr;
  c->prof_armed = false, c->prof_used = 0, c->rejected_in_a_row = 0;
  for (hipEvent_t& e : c->prof_ev) e = nullptr;
  HIP_TRY(hipHostMalloc((void**)&c->scal_h, kScalars * sizeof(double) + sizeof(pdlpdev_ctl)));
  c->ctl_h = (pdlpdev_ctl*)(c->scal_h + kScalars);
  HIP_TRY(hipMalloc((void**)&c->arena, kArenaChunk));
  HIP_TRY(hipMemsetAsync(c->arena, 0, kArenaChunk, c->stream));
  c->first_chunk = c->arena;
  const int m = c->m, n = c->n;
  {
    const size_t bytes = (24 * ((size_t)n + kSlicePad + 32) + 15 * ((size_t)m + 32)) * sizeof(double);
    if ((int64_t)m + n >= 262144 && hipMalloc((void**)&c->slab, bytes) == hipSuccess) {
      c->allocs.push_back(c->slab);
      c->slab_cap = bytes;
      HIP_TRY(hipMemsetAsync(c->slab, 0, bytes, c->stream));
    } else {
      c->slab = nullptr;
      (void)hipGetLastError();
    }
  }
  auto own_copy = [&](double** p, size_t count) -> int {  // an own buffer holding what the parent's holds
    const double* src = *p;
    TRY(dev_alloc(c, p, count));
    HIP_TRY(hipMemcpyAsync(*p, src, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
  };
  TRY(own_copy(&c->lb, n)); TRY(own_copy(&c->lb_u, n)); TRY(own_copy(&c->ub, n)); TRY(own_copy(&c->ub_u, n));
  // (lo, hi and their unscaled twins stay the parent's until a reset brings other row bounds: pdlpdev_reset makes the copies then)
  c->rows_aliased = true, c->parent = parent, c->clones_alive = 0;
  parent->clones_alive += 1;
  parent->rows_private = false;  // (this clone aliases the parent's CURRENT row bounds)
  for (int i = 0; i < 2; ++i) {
    TRY(dev_alloc(c, &c->x[i], (size_t)n + kSlicePad)); TRY(dev_alloc(c, &c->y[i], m));
    TRY(dev_alloc(c, &c->aty[i], (size_t)n + kSlicePad)); TRY(dev_alloc(c, &c->rc[i], n));
  }
  TRY(dev_alloc(c, &c->xbar, (size_t)n + kSlicePad)); TRY(dev_alloc(c, &c->sumx, (size_t)n + kSlicePad)); TRY(dev_alloc(c, &c->sumy, m));
  TRY(dev_alloc(c, &c->avgx, n)); TRY(dev_alloc(c, &c->avgy, m));
  TRY(dev_alloc(c, &c->lrx, n)); TRY(dev_alloc(c, &c->lry, m));
  TRY(dev_alloc(c, &c->tmp_n, n)); TRY(dev_alloc(c, &c->tmp_m, m));
  for (int i = 0; i < 3; ++i) {
    TRY(dev_alloc(c, &c->ax_u[i], m));
    TRY(dev_alloc(c, &c->aty_u[i], n));
  }
  TRY(dev_alloc(c, &c->rc_scratch, n));
  const int pa_w = std::max({c->a_nb, c->pa.on ? c->pa.v.W : 0, c->ja.on ? c->ja.v.nblk + c->ja.v.nlong : 0, c->pba.on ? c->pba.v.B : 0, 1});
  const int pt_w = std::max({c->at_nb, c->pat.on ? c->pat.v.W : 0, c->jat.on ? c->jat.v.nblk + c->jat.v.nlong : 0, c->pbat.on ? c->pbat.v.B : 0, 1});
  TRY(dev_alloc(c, &c->part_a, (size_t)8 * pa_w));
  TRY(dev_alloc(c, &c->part_at, (size_t)8 * pt_w));
  TRY(dev_alloc(c, &c->part_g, (size_t)8 * 2048));
  TRY(dev_alloc(c, &c->scal, kScalars));
  TRY(dev_alloc(c, &c->ctl, 1));
  TRY(dev_alloc(c, &c->ar_buf, (size_t)n + kSlicePad));
  HIP_TRY(hipMemcpyAsync(c->ctl, parent->ctl, sizeof(pdlpdev_ctl), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// K = 2, 4, 8 or 16 contexts over ONE matrix (ctx[0] and its clones, or contexts that alias the same arrays) advance together.  -7: the
// layouts are not the ones whose reduction trees the batched products reproduce (the caller keeps its independent solves).
int pdlpdev_batch_create(pdlpdev_batch** out, pdlpdev_ctx** ctx, int K)
{
  if (!out || !ctx || (K != 2 && K != 4 && K != 8 && K != 16)) return fail(-1, "pdlpdev_batch_create: K must be 2, 4, 8 or 16");
  pdlpdev_ctx* c0 = ctx[0];
  for (int l = 0; l < K; ++l)
    for (int q = 0; q < l; ++q)
      if (ctx[q] == ctx[l]) return fail(-1, "pdlpdev_batch_create: LP %d and LP %d are the same context (two lanes would share one set of iterates)", q, l);
  for (int l = 0; l < K; ++l) {
    pdlpdev_ctx* c = ctx[l];
    if (!c || c->ha_off != c0->ha_off || c->hat_off != c0->hat_off || c->pa.v.row0 != c0->pa.v.row0 || c->stream != c0->stream)
      return fail(-1, "pdlpdev_batch_create: the contexts do not share one matrix (pdlpdev_clone_shared)");
  }
  if ((int64_t)std::max(c0->m, c0->n) * K >= ((int64_t)1 << 32))
    return fail(-7, "pdlpdev_batch_create: not eligible (the interleaved vectors are addressed with 32-bit element offsets: max(m, n) * K < 2^32)");
  // per side: the row-sum variant of the panels, or the CSR stream layout -- the two whose rows are summed left to right by one lane and
  // whose per-block reduction the batched products reproduce
  pdlpdev_batch::Side side[2];
  for (int t = 0; t < 2; ++t) {
    const pdlpdev_ctx::Panels& P = t ? c0->pat : c0->pa;
    const bool jag = t ? c0->jat.on : c0->ja.on, pb = t ? c0->pbat.on : c0->pba.on;
    const int nlong = t ? c0->at_nlong : c0->a_nlong;
    bool ok = !jag && !pb && nlong == 0;
    if (ok && P.on) {
      ok      = !P.v.seg && !P.v.own_row && !P.v.any_long && !P.v.dense_add;
      side[t] = pdlpdev_batch::Side{P.v.W, P.v.row0, true};
    } else if (ok) {
      side[t] = pdlpdev_batch::Side{t ? c0->at_nb : c0->a_nb, t ? c0->at_rb : c0->a_rb, false};
      ok      = side[t].W > 0 && side[t].row0 != nullptr;
    }
    if (!ok || c0->dense.on || c0->comm || c0->small_resident)
      return fail(-7, "pdlpdev_batch_create: not eligible (the batched products reproduce the reductions of the row-sum panels and of the CSR stream "
                      "kernels: both matrices in one of these layouts, no row of more than %d entries, no dense segments, one GPU, not the resident "
                      "small-LP loop)", kLongRow);
  }
  HIP_TRY(hipSetDevice(c0->device));
  {
    struct Flag {  // (released on every way out)
      int* p = nullptr;
      ~Flag() { if (p) (void)hipFree(p); }
    } flag;
    HIP_TRY(hipMalloc((void**)&flag.p, sizeof(int)));
    int* bad = flag.p;
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), c0->stream));
    kb_check_sorted<<<2048, 256, 0, c0->stream>>>(c0->m, c0->ha_off, c0->ha_idx, bad);
    kb_check_sorted<<<2048, 256, 0, c0->stream>>>(c0->n, c0->hat_off, c0->hat_idx, bad);
    LAUNCH_CHECK();
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    if (h) return fail(-7, "pdlpdev_batch_create: not eligible (column indices are not ascending within the rows)");
  }
  pdlpdev_batch* b = new pdlpdev_batch();
  *out      = b;
  b->K = K, b->device = c0->device, b->stream = c0->stream;
  b->a_side = side[0], b->t_side = side[1];
  std::vector<BatchLp>& h = b->lp_host;
  h.resize(K);
  std::vector<pdlpdev_decision_args>& dargs = b->dargs_host;
  dargs.resize(K);
  for (int l = 0; l < K; ++l) {
    pdlpdev_ctx* c = ctx[l];
    b->ctx[l]      = c;
    h[l]     = batch_lp_of(c);
    dargs[l] = pdlpdev_decision_args{c->ctl, c->part_a, side[0].W, c->part_at, side[1].W, c->sp};
    c->batches_alive += 1;
  }
  HIP_TRY(hipMalloc((void**)&b->lp_dev, K * sizeof(BatchLp)));
  HIP_TRY(hipMalloc((void**)&b->xK, ((size_t)c0->n * K + 64) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&b->yK, ((size_t)c0->m * K + 64) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&b->dargs_dev, K * sizeof(pdlpdev_decision_args)));
  // everything on the batch's OWN stream (a non-blocking one: work of the null stream -- hipMemset, hipMemcpy -- is not ordered
  // with it, and a memset that the queue scheduler lets wait can land attempts later, in the middle of a product's vectors)
  HIP_TRY(hipMemcpyAsync(b->lp_dev, h.data(), K * sizeof(BatchLp), hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(b->dargs_dev, dargs.data(), K * sizeof(pdlpdev_decision_args), hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipMemsetAsync(b->xK, 0, ((size_t)c0->n * K + 64) * sizeof(double), b->stream));
  HIP_TRY(hipMemsetAsync(b->yK, 0, ((size_t)c0->m * K + 64) * sizeof(double), b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

void pdlpdev_batch_destroy(pdlpdev_batch* b)
{
  if (!b) return;
  for (int l = 0; l < b->K; ++l)
    if (b->ctx[l]) b->ctx[l]->batches_alive -= 1;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (auto& kv : b->graphs) (void)hipGraphExecDestroy(kv.second);
  if (b->lp_dev) (void)hipFree(b->lp_dev);
  if (b->dargs_dev) (void)hipFree(b->dargs_dev);
  if (b->xK) (void)hipFree(b->xK);
  if (b->yK) (void)hipFree(b->yK);
  delete b;
}

// Attempts for all K LPs until LP l holds targets[l] accepted steps (targets[l] <= 0: LP l rests; an LP that reaches its target
// rests too: its lanes and its per-LP kernels turn into no-ops).  ctl[l] receives LP l's control block.
int pdlpdev_batch_run(pdlpdev_batch* b, const int32_t* targets, pdlpdev_ctl* ctl)
{
  if (!b || !targets) return fail(-1, "pdlpdev_batch_run: null argument");
  roctx::Range range("pdlp: batched PDHG attempts");
  HIP_TRY(hipSetDevice(b->device));
  TRY(batch_refresh_table(b));
  const int K = b->K;
  for (int l = 0; l < K; ++l)
    if (targets[l] > 0) k_set_target<<<1, 1, 0, b->stream>>>(b->ctx[l]->ctl, targets[l]);
  LAUNCH_CHECK();
  TRY(batch_fetch_ctl(b));
  auto wants = [&](int l) { return targets[l] > 0 && b->ctx[l]->ctl_h->error == 0 && b->ctx[l]->ctl_h->steps_taken < targets[l]; };
  int guard = 0;
  for (;;) {
    int remaining = 0;
    int before[kBatchMax], asked[kBatchMax];
    for (int l = 0; l < K; ++l) {
      before[l] = b->ctx[l]->ctl_h->steps_taken;
      asked[l]  = wants(l) ? targets[l] - before[l] : 0;
      remaining = std::max(remaining, asked[l]);
    }
    if (remaining == 0) break;
    const bool use_graph = b->ctx[0]->use_graph != 0;
    while (remaining > 0) {
      int chunk = 1;
      while (chunk * 2 <= remaining && chunk < 64) chunk *= 2;
      if (use_graph) {
        hipGraphExec_t g;
        TRY(batch_graph(b, chunk, &g));
        HIP_TRY(hipGraphLaunch(g, b->stream));
      } else {
        for (int i = 0; i < chunk; ++i) TRY(batch_enqueue(b));
      }
      remaining -= chunk;
    }
    TRY(batch_fetch_ctl(b));
    for (int l = 0; l < K; ++l) {  // (64 rejections in a row leave nothing of a step size: the single loop's rule, per LP)
      if (asked[l] == 0) continue;
      pdlpdev_ctx* c = b->ctx[l];
      c->rejected_in_a_row = c->ctl_h->steps_taken == before[l] ? c->rejected_in_a_row + asked[l] : 0;
      if (c->ctl_h->error == 0 && c->rejected_in_a_row >= 64) {
        c->rejected_in_a_row = 0;
        k_set_error<<<1, 1, 0, b->stream>>>(c->ctl);
        LAUNCH_CHECK();
      }
    }
    TRY(batch_fetch_ctl(b));
    if (++guard > 100000) return fail(-6, "pdlpdev_batch_run: no progress");
  }
  if (ctl)
    for (int l = 0; l < K; ++l) ctl[l] = *b->ctx[l]->ctl_h;
  return 0;
}

// Average dispatch durations (ms) of the four kernels of a batched attempt: {primal, A / dual, A^T / step, decisions}, measured as
// pdlpdev_time_kernel measures the single-LP kernels -- whole attempts in the loop's order (each kernel meets the caches as the loop
// leaves them), every LP forced active with its averaging traffic, the dispatches' own timestamps; control blocks and running sums
// are put back afterwards (the "other" iterate buffers are scratch between attempts).
int pdlpdev_batch_time_kernels(pdlpdev_batch* b, int reps, double avg_ms[4])
{
  if (!b || !avg_ms) return fail(-1, "pdlpdev_batch_time_kernels: null argument");
  HIP_TRY(hipSetDevice(b->device));
  TRY(batch_refresh_table(b));
  hipStream_t s = b->stream;
  const int K   = b->K;
  if (reps < 1) reps = 1;
  TRY(batch_fetch_ctl(b));
  pdlpdev_ctl saved[kBatchMax], forced[kBatchMax];
  struct Scratch {  // (copies of the running sums, the dispatches' events: released on every way out)
    double *sx[kBatchMax] = {nullptr}, *sy[kBatchMax] = {nullptr};
    hipEvent_t ev[8] = {nullptr};
    ~Scratch()
    {
      for (double* p : sx) if (p) (void)hipFree(p);
      for (double* p : sy) if (p) (void)hipFree(p);
      for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
  } scratch;
  double **sx = scratch.sx, **sy = scratch.sy;
  hipEvent_t* ev = scratch.ev;
  for (int l = 0; l < K; ++l) {
    pdlpdev_ctx* c = b->ctx[l];
    saved[l] = forced[l] = *c->ctl_h;
    forced[l].pending_avg = 1, forced[l].target_steps = saved[l].steps_taken + 1, forced[l].error = 0;
    HIP_TRY(hipMalloc((void**)&sx[l], std::max<size_t>(c->n, 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&sy[l], std::max<size_t>(c->m, 1) * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(sx[l], c->sumx, (size_t)c->n * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(sy[l], c->sumy, (size_t)c->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  auto arm = [&]() -> int {
    for (int l = 0; l < K; ++l) HIP_TRY(hipMemcpyAsync(b->ctx[l]->ctl, &forced[l], sizeof(pdlpdev_ctl), hipMemcpyHostToDevice, s));
    return 0;
  };
  for (int q = 0; q < 8; ++q) HIP_TRY(hipEventCreate(&ev[q]));
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  TRY(arm());
  TRY(batch_enqueue(b));  // warm
  for (int i = 0; i < reps; ++i) {
    TRY(arm());
    TRY(batch_enqueue(b, ev));
    HIP_TRY(hipStreamSynchronize(s));
    for (int q = 0; q < 4; ++q) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ev[2 * q], ev[2 * q + 1]));
      sum[q] += ms;
    }
  }
  for (int q = 0; q < 4; ++q) avg_ms[q] = sum[q] / reps;
  for (int l = 0; l < K; ++l) {
    pdlpdev_ctx* c = b->ctx[l];
    HIP_TRY(hipMemcpyAsync(c->ctl, &saved[l], sizeof(pdlpdev_ctl), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->sumx, sx[l], (size_t)c->n * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->sumy, sy[l], (size_t)c->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
