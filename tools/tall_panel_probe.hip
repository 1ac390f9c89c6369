// Round-6 probe (review item 1): the one design class for the unstructured product that rounds 1-5 had not tried --
// NO global gather and NO product round trip through HBM.
//
//   * a workgroup owns a TALL row panel (R rows, their fp64 accumulators live in LDS: R = 15 625 -> 125 KB) times a COLUMN RANGE
//     (n / S columns); NP * S workgroups, one per CU;
//   * it walks the range in SLABS of SW columns: the slab of the gathered vector is read COALESCED (16 bytes per lane, no divergent
//     address, so the 1.85-clocks-per-lane texture path is not involved) by LOADER waves, D slabs ahead in registers, and parked
//     in a double-buffered LDS stage;
//   * the panel's entries of the slab (the "cell": 8-byte value + 32-bit {row in panel 14 b, column in slab 11 b, level 7 b}) are
//     streamed by CONSUMER waves in 64-entry chunks of ONE contiguous stream per workgroup, sorted (slab, row, column); an entry is
//     one ds_read_b64 of the slab + one ds_add_f64 into its row's accumulator;
//   * order of the additions of a row: slabs are separated by the step's barrier (columns ascending); inside a cell the entries of a
//     row are adjacent in the stream, never straddle a chunk (the builder pads: a few thousand pads in 1e7 entries) and carry their
//     position in the run as a LEVEL: the chunk's wave issues level 0, then level 1, ... -- LDS operations of one wave execute in
//     order, so every row is summed left to right inside its column range, deterministically;
//   * the S partial vectors are combined in a fixed order by a second, streaming kernel that also carries the fused epilogue:
//     ((p0 + p1) + p2) + p3 is not the oracle's left-to-right sum -> rtol 1e-12 per row, the contract of the long-tail panels.
//
// Workgroups of one XCD take the same column range (blockIdx % 8 -> XCD), so its slice of the vector sits in that XCD's L2.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tall_panel_probe.hip -o /tmp/tall_panel_probe
//   /tmp/tall_panel_probe <dir from scripts/dump_csr.py> [NP S SW LW CW D [reps]]      (no arguments after dir: the sweep)
// Never linked into the product: a harness, like the rest of tools/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <class T>
static bool slurp(const std::string& path, std::vector<T>* out)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out->resize((size_t)bytes / sizeof(T));
  const size_t got = std::fread(out->data(), sizeof(T), out->size(), f);
  std::fclose(f);
  return got == out->size();
}

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int kRowBits = 14, kColBits = 11, kLvlShift = kRowBits + kColBits;
constexpr unsigned kPadLevel = 127u;

struct TallView {
  int rows, cols;       // of the matrix
  int NP, S, SW, NS;    // row panels, column ranges, slab width (columns), STEPS per workgroup (the largest count, a multiple of the prefetch depths)
  int R, CWID;          // rows per panel, columns per range (NS * SW)
  int G;                // workgroups in the grid (NP * S rounded up to 8)
  const double* val;    // entries of all workgroups, each workgroup's stream 64-aligned
  const uint32_t* pk;
  const long long* gbase;  // [G] first entry of the workgroup's stream
  const int2* cq;          // [G * (NS + 1)] per step: its first entry inside the stream, the length q <= 64 of a consumer wave's share (the step holds CW * q slots)
  const int* sl;           // [G * (NS + 1)] per step: its slab (a cell of more than 64 * CW entries takes several steps over the same slab)
  int rows_pad;            // stride of the partial vectors
};

// blockIdx -> (panel, range): the workgroups of one XCD share a column range
__device__ __forceinline__ bool tall_where(const TallView& V, int b, int* p, int* q)
{
  const int xcd = b & 7, j = b >> 3;
  if (V.S <= 8) {
    const int per = 8 / V.S;  // panels per group of 8 workgroups
    *q = xcd % V.S;
    *p = j * per + xcd / V.S;
  } else {
    *q = b % V.S, *p = b / V.S;
  }
  return *p < V.NP;
}

__device__ __forceinline__ void lds_add(double* a, double v) { __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// (a barrier that leaves global loads in flight: LDS traffic drained, vector-memory counter untouched)
__device__ __forceinline__ void step_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One entry of a consumer's share: one ds_read_b64 of the slab, one addition into the row's accumulator.  ATOMIC: ds_add_f64 (fire and
// forget); otherwise read - add - write by the lane (no other wave touches the row during the step, and the LDS executes one wave's
// operations in order).
template <bool ATOMIC, int DBG>
__device__ __forceinline__ void tall_consume(double cv, uint32_t cpk, bool in_share, const double* xs, double* acc)
{
  const unsigned lvl = cpk >> kLvlShift;
  const bool act     = in_share && lvl != kPadLevel;
  const int row      = (int)(cpk & ((1u << kRowBits) - 1u));
  const int col      = (int)((cpk >> kRowBits) & ((1u << kColBits) - 1u));
  const double pr    = cv * ((DBG & 4) ? (double)col : (act ? xs[col] : 0.0));
  if (DBG & 8) { if (pr == 1.2345) acc[row] = pr; return; }
  if (!__ballot(act && lvl > 0u)) {  // (uniform) no row has two entries in this part of the share
    if (act) {
      if (ATOMIC) lds_add(acc + row, pr);
      else acc[row] = acc[row] + pr;
    }
  } else {
    for (unsigned d = 0; __ballot(act && lvl >= d); ++d)
      if (act && lvl == d) {
        if (ATOMIC) lds_add(acc + row, pr);
        else acc[row] = acc[row] + pr;
      }
  }
}

template <int LW, int CW, int D, int XL /* 16-byte loads per loader lane and slab */, int P /* steps the entry stream runs ahead */, bool ATOMIC, int DBG = 0 /* timing experiments: 1 no slab writes, 2 no slab loads, 4 no slab reads, 8 no additions, 16 no entry loads */>
__global__ __launch_bounds__((LW + CW) * 64) void k_tall(TallView V, const double* __restrict__ x, double* __restrict__ partial)
{
  extern __shared__ double lds[];
  constexpr int T   = (LW + CW) * 64;
  constexpr int TAB = (P > D ? P : D) + 2;      // table entries read beyond the last step
  int p, q;
  if (!tall_where(V, blockIdx.x, &p, &q)) return;
  const int g     = blockIdx.x;
  double* acc     = lds;                        // [R]
  double* xb      = lds + ((V.R + 1) & ~1);     // [2][SW]
  int2* cqs       = (int2*)(xb + 2 * V.SW);     // [NS + TAB]
  int* sls        = (int*)(cqs + V.NS + TAB);   // [NS + TAB]
  const int r0    = p * V.R;
  const int nr    = min(V.R, V.rows - r0);
  const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane  = threadIdx.x & 63;
  for (int i = threadIdx.x; i < V.R; i += T) acc[i] = 0.0;
  for (int i = threadIdx.x; i < V.NS + TAB; i += T) {
    cqs[i] = i <= V.NS ? V.cq[(size_t)g * (V.NS + 1) + i] : make_int2(0, 0);
    sls[i] = V.sl[(size_t)g * (V.NS + 1) + (i <= V.NS ? i : V.NS)];
  }
  const double* xr = x + (size_t)q * V.CWID;
  __syncthreads();                              // the tables
  if (wave < LW) {
    // ---- loader: the slab of step s + 1 is written to LDS behind barrier s; D slabs in flight in registers
    d2 st[D][XL];
    const int t = wave * 64 + lane;             // 0 .. LW * 64: 16-byte index inside a slab, stride LW * 64
    {
      const double* src = xr + (size_t)__builtin_amdgcn_readfirstlane(sls[0]) * V.SW;
#pragma unroll
      for (int u = 0; u < XL; ++u) {            // step 0's slab straight into its buffer
        d2 v = *(const d2*)(src + 2 * (t + u * LW * 64));
        *(d2*)(xb + 2 * (t + u * LW * 64)) = v;
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double* src = xr + (size_t)__builtin_amdgcn_readfirstlane(sls[d + 1]) * V.SW;
#pragma unroll
      for (int u = 0; u < XL; ++u) st[d][u] = *(const d2*)(src + 2 * (t + u * LW * 64));
    }
    // (table look-ups out of the step's dependent chain: lane l keeps entry w + l of a 64-entry window, v_readlane picks)
    int tsl = sls[min(lane, V.NS + TAB - 1)];
    step_barrier();                             // barrier 0: step 0's slab and the zeroed accumulators are visible
    for (int s0 = 0; s0 < V.NS; s0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int s = s0 + d;                   // step s: park the slab of step s + 1 (registers d), request the one of step s + 1 + D
        if (((s + 1 + D) & 63) == 0) tsl = sls[min(s + 1 + D + lane, V.NS + TAB - 1)];
        double* dst = xb + ((s + 1) & 1) * V.SW;
#pragma unroll
        for (int u = 0; u < XL; ++u) {
          if (DBG & 1) { if (st[d][u].x == 1.2345) *(d2*)(dst + 2 * (t + u * LW * 64)) = st[d][u]; }
          else *(d2*)(dst + 2 * (t + u * LW * 64)) = st[d][u];
        }
        const double* src = xr + (size_t)__builtin_amdgcn_readlane(tsl, (s + 1 + D) & 63) * V.SW;
#pragma unroll
        for (int u = 0; u < XL; ++u) {
          if (DBG & 2) st[d][u].x = st[d][u].x + 1.0;
          else st[d][u] = *(const d2*)(src + 2 * (t + u * LW * 64));
        }
        step_barrier();                         // barrier s + 1
      }
    }
  } else {
    // ---- consumer c: of every step the share [c * q, (c + 1) * q) of its CW * q slots (q <= 64: lane <-> entry), requested P steps
    // ahead into a ring of registers.  Straight-line code (every lane always loads, lanes beyond the share re-read the step's first
    // entry; no branch contains a load), so that the compiler's own counted vmcnt waits keep P - 1 steps of loads in flight.
    const int c = wave - LW;
    const double* __restrict__ val = V.val + V.gbase[g];
    const uint32_t* __restrict__ pk = V.pk + V.gbase[g];
    double rv[P];
    uint32_t rp[P];
#pragma unroll
    for (int d = 0; d < P; ++d) {
      const int base = __builtin_amdgcn_readfirstlane(cqs[d].x), qn = __builtin_amdgcn_readfirstlane(cqs[d].y);
      const int i    = lane < qn ? base + c * qn + lane : base;
      rv[d] = __builtin_nontemporal_load(val + i);
      rp[d] = __builtin_nontemporal_load(pk + i);
    }
    int tq = 0, nb = cqs[min(lane, V.NS + TAB - 1)].x, nq = cqs[min(lane, V.NS + TAB - 1)].y;
    step_barrier();                             // barrier 0
    for (int s0 = 0; s0 < V.NS; s0 += P) {
#pragma unroll
      for (int d = 0; d < P; ++d) {
        const int s      = s0 + d;
        const double* xs = xb + (s & 1) * V.SW;
        if ((s & 63) == 0) tq = cqs[min(s + lane, V.NS + TAB - 1)].y;
        if (((s + P) & 63) == 0) {
          const int2 e = cqs[min(s + P + lane, V.NS + TAB - 1)];
          nb = e.x, nq = e.y;
        }
        const int qs     = __builtin_amdgcn_readlane(tq, s & 63);
        tall_consume<ATOMIC, DBG>(rv[d], rp[d], lane < qs, xs, acc);
        const int nbase = __builtin_amdgcn_readlane(nb, (s + P) & 63), qn = __builtin_amdgcn_readlane(nq, (s + P) & 63);
        const int i     = lane < qn ? nbase + c * qn + lane : nbase;
        if (!(DBG & 16)) {
          rv[d] = __builtin_nontemporal_load(val + i);
          rp[d] = __builtin_nontemporal_load(pk + i);
        } else {
          rp[d] = (rp[d] & ~((1u << kRowBits) - 1u)) | ((rp[d] + 77u) & 8191u);
        }
        step_barrier();                         // barrier s + 1
      }
    }
  }
  // (the last barrier of either branch: every addition is done)
  double* out = partial + (size_t)q * V.rows_pad + r0;
  for (int i = threadIdx.x; i < nr; i += T) out[i] = acc[i];
}

// combine the S partial vectors in a fixed order + a stand-in for the fused dual-side epilogue's traffic (reads y, lo, hi, the
// pending average; writes y', the average: 48 bytes per row like k_panel_a_dual's)
template <bool EPI>
__global__ __launch_bounds__(256) void k_combine(int rows, int rows_pad, int S, const double* __restrict__ partial, double* __restrict__ y_out,
                                                 const double* __restrict__ e0, const double* __restrict__ e1, const double* __restrict__ e2,
                                                 double* __restrict__ e3, double sigma)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  double v = __builtin_nontemporal_load(partial + i);
  for (int q = 1; q < S; ++q) v = v + __builtin_nontemporal_load(partial + (size_t)q * rows_pad + i);
  if (EPI) {
    const double y = e0[i], lo = e1[i], hi = e2[i];
    double t = y - sigma * v;
    t = t < lo ? lo : (t > hi ? hi : t);
    e3[i] = e3[i] + sigma * y;
    v = t;
  }
  y_out[i] = v;
}

// stand-in for k_primal between the products: 72 bytes per column, so that the products evict each other exactly like in the loop
__global__ __launch_bounds__(256) void k_touch(int n, const double* a, const double* b, const double* c, const double* d, const double* e, double* f, double* g, double* h)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = a[i] + b[i] + c[i] + d[i] + e[i];
  f[i] = v, g[i] = 2 * v;
  h[i] = h[i] + v;
}

struct TallHost {
  TallView V{};
  std::vector<double> val;
  std::vector<uint32_t> pk;
  std::vector<long long> gbase;
  std::vector<int2> cq;
  std::vector<int> sl;
  long long pads = 0, entries = 0, max_cell = 0, dup_entries = 0, split_steps = 0;
  int slabs = 0, min_steps = 0;
  double build_s = 0;
};

static int lcm_int(int a, int b) { int x = a, y = b; while (y) { int t = x % y; x = y; y = t; } return a / x * b; }

static bool build_tall(int rows, int cols, const std::vector<int>& off, const std::vector<int>& idx, const std::vector<double>& a, int NP, int S, int SW, int D, int P,
                       int CW, int rot, TallHost* H)
{
  const auto t0 = std::chrono::steady_clock::now();
  TallView& V = H->V;
  V.rows = rows, V.cols = cols, V.NP = NP, V.S = S, V.SW = SW;
  V.R  = (rows + NP - 1) / NP;
  if (V.R > (1 << kRowBits) || SW > (1 << kColBits)) return false;
  const int cw    = (cols + S - 1) / S;
  const int mult  = lcm_int(D, P);
  const int slabs = (cw + SW - 1) / SW;
  H->slabs = slabs;
  V.CWID   = slabs * SW;
  // (ranges of slabs * SW columns: the last range may be partly beyond the matrix -- the vector is padded)
  V.G        = S <= 8 ? ((NP + 8 / S - 1) / (8 / S)) * 8 : NP * S;
  V.rows_pad = (rows + 63) & ~63;
  H->gbase.assign(V.G, 0);
  H->val.clear(), H->pk.clear();
  H->val.reserve(idx.size() + idx.size() / 8), H->pk.reserve(idx.size() + idx.size() / 8);
  std::vector<std::vector<std::pair<uint32_t, double>>> cell((size_t)slabs);  // (row << 11 | col in slab, value), pushed row by row = sorted
  std::vector<std::vector<int2>> wg_cq(V.G);
  std::vector<std::vector<int>> wg_sl(V.G);
  std::vector<int> run_len;
  const uint32_t kPad = kPadLevel << kLvlShift;
  for (int b = 0; b < V.G; ++b) {
    int p, q;
    {
      const int xcd = b & 7, j = b >> 3;
      if (S <= 8) { q = xcd % S; p = j * (8 / S) + xcd / S; } else { q = b % S; p = b / S; }
    }
    H->gbase[b] = (long long)H->val.size();
    if (p >= NP) continue;
    for (auto& c : cell) c.clear();
    const int r0 = p * V.R, r1 = std::min(rows, r0 + V.R);
    const int c0 = q * V.CWID, c1 = c0 + V.CWID;
    for (int r = r0; r < r1; ++r) {
      const int* jb = idx.data() + off[r];
      const int* je = idx.data() + off[r + 1];
      const int* lo = std::lower_bound(jb, je, c0);
      for (const int* j = lo; j < je && *j < c1; ++j) {
        const int s = (*j - c0) / SW;
        cell[s].push_back({(uint32_t)(r - r0) << kColBits | (uint32_t)(*j - c0 - s * SW), a[(size_t)(j - idx.data())]});
      }
    }
    long long pos = 0;  // inside the workgroup's stream
    // rot: the workgroups of one XCD (they share the column range) start their walk at different slabs, so that at any moment
    // they read different lines of the range instead of all the same 16 KB (a fixed order per workgroup: still deterministic)
    const int first_slab = rot ? (int)(((long long)(b >> 3) * slabs) / std::max(1, V.G / 8)) % slabs : 0;
    for (int sv = 0; sv < slabs; ++sv) {
      const int s   = (sv + first_slab) % slabs;
      const auto& c = cell[s];
      H->max_cell = std::max<long long>(H->max_cell, (long long)c.size());
      H->entries += (long long)c.size();
      // the runs of equal rows (an entry alone is a run of one); a run never leaves its wave's share
      run_len.clear();
      for (size_t i = 0; i < c.size();) {
        size_t e = i + 1;
        while (e < c.size() && (c[e].first >> kColBits) == (c[i].first >> kColBits)) ++e;
        run_len.push_back((int)(e - i));
        if (e - i > 64) return false;  // (a row with more than 64 entries inside one slab: not this layout's matrix)
        if (e - i > 1) H->dup_entries += (long long)(e - i);
        i = e;
      }
      // steps over this slab: consecutive groups of runs, each group dealt to the CW consumer waves in shares of q <= 64 slots
      size_t ri = 0, ei = 0;
      bool first = true;
      while (first || ri < run_len.size()) {
        first = false;
        // the longest prefix of the remaining runs that fits CW shares of 64, then the smallest q that still holds it
        size_t rj = ri;
        {
          int w = 0, fill = 0;
          for (; rj < run_len.size(); ++rj) {
            const int L = run_len[rj];
            if (fill + L > 64) { ++w, fill = 0; }
            if (w >= CW) break;
            fill += L;
          }
        }
        int cnt = 0, longest = 0;
        for (size_t k = ri; k < rj; ++k) cnt += run_len[k], longest = std::max(longest, run_len[k]);
        int qs = std::max((cnt + CW - 1) / CW, longest);
        for (;; ++qs) {
          int w = 0, fill = 0;
          bool fits = true;
          for (size_t k = ri; k < rj; ++k) {
            const int L = run_len[k];
            if (fill + L > qs) { ++w, fill = 0; }
            if (w >= CW) { fits = false; break; }
            fill += L;
          }
          if (fits) break;
        }
        wg_cq[b].push_back(make_int2((int)pos, qs));
        wg_sl[b].push_back(s);
        if (ri > 0) ++H->split_steps;
        int w = 0, fill = 0;
        for (size_t k = ri; k < rj; ++k) {
          const int L = run_len[k];
          if (fill + L > qs) {
            for (; fill < qs; ++fill) { H->val.push_back(0.0), H->pk.push_back(kPad), ++H->pads; }
            ++w, fill = 0;
          }
          for (int u = 0; u < L; ++u, ++ei) {
            const uint32_t row = c[ei].first >> kColBits, col = c[ei].first & ((1u << kColBits) - 1u);
            H->val.push_back(c[ei].second);
            H->pk.push_back(row | col << kRowBits | (uint32_t)u << kLvlShift);
          }
          fill += L;
        }
        for (long long k = (long long)w * qs + fill; k < (long long)CW * qs; ++k) { H->val.push_back(0.0), H->pk.push_back(kPad), ++H->pads; }
        pos += (long long)CW * qs;
        ri = rj;
      }
    }
    wg_cq[b].push_back(make_int2((int)pos, 0));  // (an empty step: the requests beyond the last step read its first slot -- one pad)
    H->val.push_back(0.0), H->pk.push_back(kPad);
  }
  int ns = 0, ns_min = 1 << 30;
  for (int b = 0; b < V.G; ++b) ns = std::max(ns, (int)wg_sl[b].size()), ns_min = wg_sl[b].empty() ? ns_min : std::min(ns_min, (int)wg_sl[b].size());
  H->min_steps = ns_min;
  V.NS = (ns + mult - 1) / mult * mult;
  H->cq.assign((size_t)V.G * (V.NS + 1), make_int2(0, 0));
  H->sl.assign((size_t)V.G * (V.NS + 1), 0);
  for (int b = 0; b < V.G; ++b) {
    if (wg_cq[b].empty()) continue;
    const int2 last = wg_cq[b].back();
    for (int s = 0; s <= V.NS; ++s) {
      H->cq[(size_t)b * (V.NS + 1) + s] = s < (int)wg_sl[b].size() ? wg_cq[b][s] : last;
      H->sl[(size_t)b * (V.NS + 1) + s] = s < (int)wg_sl[b].size() ? wg_sl[b][s] : slabs - 1;
    }
  }
  for (int i = 0; i < 256; ++i) H->val.push_back(0.0), H->pk.push_back(kPad);
  H->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return true;
}

struct TallDev {
  TallView V{};
  double* val = nullptr;
  uint32_t* pk = nullptr;
  long long* gbase = nullptr;
  int2* cq = nullptr;
  int* sl = nullptr;
  double* partial = nullptr;
  size_t lds_bytes = 0;
};
static void upload(const TallHost& H, TallDev* Dv)
{
  Dv->V = H.V;
  OK(hipMalloc((void**)&Dv->val, H.val.size() * 8)); OK(hipMalloc((void**)&Dv->pk, H.pk.size() * 4));
  OK(hipMalloc((void**)&Dv->gbase, H.gbase.size() * 8)); OK(hipMalloc((void**)&Dv->cq, H.cq.size() * 8)); OK(hipMalloc((void**)&Dv->sl, H.sl.size() * 4));
  OK(hipMalloc((void**)&Dv->partial, (size_t)H.V.S * H.V.rows_pad * 8));
  OK(hipMemcpy(Dv->val, H.val.data(), H.val.size() * 8, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->pk, H.pk.data(), H.pk.size() * 4, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->gbase, H.gbase.data(), H.gbase.size() * 8, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->cq, H.cq.data(), H.cq.size() * 8, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->sl, H.sl.data(), H.sl.size() * 4, hipMemcpyHostToDevice));
  Dv->V.val = Dv->val, Dv->V.pk = Dv->pk, Dv->V.gbase = Dv->gbase, Dv->V.cq = Dv->cq, Dv->V.sl = Dv->sl;
  Dv->lds_bytes = (size_t)((H.V.R + 1) & ~1) * 8 + (size_t)2 * H.V.SW * 8 + (size_t)(H.V.NS + 1 + 16) * 12 + 16;
}
static void release(TallDev* Dv)
{
  (void)hipFree(Dv->val); (void)hipFree(Dv->pk); (void)hipFree(Dv->gbase); (void)hipFree(Dv->cq); (void)hipFree(Dv->sl); (void)hipFree(Dv->partial);
}

typedef void (*tall_fn)(TallView, const double*, double*);
struct Variant { int LW, CW, D, SW, P, XM, atomic; tall_fn fn; };
#define VARIANT(LW, CW, D, SW, P, AT) Variant{LW, CW, D, SW, P, 1, AT, k_tall<LW, CW, D, (SW) / (2 * 64 * (LW)), P, (AT) != 0>}
#define VARIANT_DBG(LW, CW, D, SW, P, DBG) Variant{LW, CW, D, SW, P, 1, 100 + DBG, k_tall<LW, CW, D, (SW) / (2 * 64 * (LW)), P, true, DBG>}
static const Variant kVariants[] = {
  VARIANT(4, 8, 4, 2048, 8, 1),  VARIANT(4, 8, 4, 2048, 8, 0),  VARIANT(8, 8, 8, 2048, 8, 1),  VARIANT(8, 8, 8, 2048, 8, 0),  VARIANT(4, 12, 4, 2048, 8, 1), VARIANT(4, 12, 6, 2048, 12, 1),
  VARIANT(4, 6, 4, 2048, 8, 1),  VARIANT(4, 8, 6, 2048, 12, 1), VARIANT(4, 8, 2, 2048, 2, 1),  VARIANT(4, 8, 4, 2048, 4, 1),  VARIANT(8, 8, 12, 2048, 12, 1), VARIANT(8, 8, 16, 2048, 16, 1),
  VARIANT(4, 4, 4, 1024, 8, 1),  VARIANT(2, 6, 8, 1024, 8, 1),  VARIANT(4, 8, 8, 1024, 8, 1),  VARIANT(4, 12, 8, 1024, 8, 1),
  // timing experiments (wrong results by construction): what each part of a step costs
  VARIANT_DBG(4, 8, 4, 2048, 8, 1), VARIANT_DBG(4, 8, 4, 2048, 8, 2), VARIANT_DBG(4, 8, 4, 2048, 8, 3), VARIANT_DBG(4, 8, 4, 2048, 8, 4), VARIANT_DBG(4, 8, 4, 2048, 8, 8),
  VARIANT_DBG(4, 8, 4, 2048, 8, 12), VARIANT_DBG(4, 8, 4, 2048, 8, 16), VARIANT_DBG(4, 8, 4, 2048, 8, 28), VARIANT_DBG(4, 8, 4, 2048, 8, 31), VARIANT_DBG(4, 8, 4, 2048, 8, 15),
};
static const Variant* find_variant(int LW, int CW, int D, int SW, int P, int XM, int AT)
{
  for (const auto& v : kVariants)
    if (v.LW == LW && v.CW == CW && v.D == D && v.SW == SW && v.P == P && v.XM == XM && v.atomic == AT) return &v;
  return nullptr;
}

int main(int argc, char** argv)
{
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc < 2) { std::printf("usage: %s <dir> [NP S SW LW CW D P XM atomic [reps]]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  int m = 0, n = 0;
  {
    FILE* f = std::fopen((dir + "/dims.txt").c_str(), "r");
    if (!f || std::fscanf(f, "%d %d", &m, &n) != 2) { std::printf("dims.txt missing\n"); return 2; }
    std::fclose(f);
  }
  std::vector<int> off[2], idx[2];
  std::vector<double> val[2];
  const char* names[2] = {"a", "at"};
  for (int t = 0; t < 2; ++t)
    if (!slurp(dir + "/" + names[t] + "_off.i32", &off[t]) || !slurp(dir + "/" + names[t] + "_idx.i32", &idx[t]) || !slurp(dir + "/" + names[t] + "_val.f64", &val[t])) {
      std::printf("cannot read %s\n", names[t]);
      return 2;
    }
  const int rows[2] = {m, n}, cols[2] = {n, m};
  const int big = std::max(m, n);
  // vectors: padded far enough for any (S, SW, D) of the sweep
  const size_t vec_pad = (size_t)big + (size_t)big / 2 + 64 * 4096;
  std::vector<double> hx(vec_pad, 0.0);
  for (int i = 0; i < big; ++i) hx[i] = 1.0 + ((i * 2654435761u) % 1000) * 1e-3 - (i % 3) * 0.7;
  double *x[2], *yout[2], *e[4], *tv[8];
  for (int t = 0; t < 2; ++t) { OK(hipMalloc((void**)&x[t], vec_pad * 8)); OK(hipMemcpy(x[t], hx.data(), vec_pad * 8, hipMemcpyHostToDevice)); OK(hipMalloc((void**)&yout[t], vec_pad * 8)); }
  for (auto& p : e) { OK(hipMalloc((void**)&p, vec_pad * 8)); OK(hipMemcpy(p, hx.data(), vec_pad * 8, hipMemcpyHostToDevice)); }
  for (auto& p : tv) { OK(hipMalloc((void**)&p, vec_pad * 8)); OK(hipMemset(p, 0, vec_pad * 8)); }
  // reference: sequential CSR sums on the host
  std::vector<double> ref[2], mag[2];
  for (int t = 0; t < 2; ++t) {
    ref[t].assign(rows[t], 0.0), mag[t].assign(rows[t], 0.0);
    for (int r = 0; r < rows[t]; ++r) {
      double s = 0, g = 0;
      for (int k = off[t][r]; k < off[t][r + 1]; ++k) { s = s + val[t][k] * hx[idx[t][k]]; g += std::fabs(val[t][k] * hx[idx[t][k]]); }
      ref[t][r] = s, mag[t][r] = g;
    }
  }
  hipStream_t st;
  OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t ev[6];
  for (auto& v : ev) OK(hipEventCreate(&v));
  std::printf("tall row panels with LDS accumulators x column ranges, x slabs streamed coalesced into LDS: %d x %d, %zu nonzeros\n", m, n, idx[0].size());
  std::printf("%4s %2s %5s %2s %2s %2s %2s %2s %2s | %6s %7s %8s %8s | %9s %9s %9s %9s | %9s | %s\n", "NP", "S", "SW", "LW", "CW", "D", "P", "XM", "at", "R", "LDS KB", "pads", "dup ent", "A us",
              "comb us", "AT us", "comb us", "pair us", "max rel err (A, AT)");
  struct Cfg { int NP, S, SW, LW, CW, D, P, XM, AT; };
  std::vector<Cfg> cfgs;
  int reps = 20;
  const int rot = std::getenv("TALL_ROT") ? std::atoi(std::getenv("TALL_ROT")) : 0;
  std::printf("slab walk of the workgroups of an XCD: %s\n", rot ? "staggered (TALL_ROT=1)" : "in lockstep from slab 0");
  if (argc >= 11) {
    cfgs.push_back({std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8]), std::atoi(argv[9]), std::atoi(argv[10])});
    if (argc >= 12) reps = std::atoi(argv[11]);
  } else {
    const bool dbg = std::getenv("TALL_DBG") != nullptr;
    for (const auto& v : kVariants) {
      if ((v.atomic >= 100) != dbg) continue;
      if (v.SW == 2048) cfgs.push_back({64, 4, v.SW, v.LW, v.CW, v.D, v.P, v.XM, v.atomic});
      else cfgs.push_back({128, 4, v.SW, v.LW, v.CW, v.D, v.P, v.XM, v.atomic}), cfgs.push_back({64, 4, v.SW, v.LW, v.CW, v.D, v.P, v.XM, v.atomic});
    }
  }
  for (const Cfg& c : cfgs) {
    const Variant* var = find_variant(c.LW, c.CW, c.D, c.SW, c.P, c.XM, c.AT);
    if (!var) { std::printf("no kernel instance for LW %d CW %d D %d SW %d P %d XM %d atomic %d\n", c.LW, c.CW, c.D, c.SW, c.P, c.XM, c.AT); continue; }
    TallHost H[2];
    TallDev Dv[2];
    bool ok = true;
    for (int t = 0; t < 2 && ok; ++t) {
      ok = build_tall(rows[t], cols[t], off[t], idx[t], val[t], c.NP, c.S, c.SW, c.D, c.P, c.CW, rot, &H[t]);
      if (ok) upload(H[t], &Dv[t]);
      if (ok && ((size_t)c.S * H[t].V.CWID + (size_t)c.SW > vec_pad)) ok = false;
    }
    if (!ok || Dv[0].lds_bytes > 163840 || Dv[1].lds_bytes > 163840) {
      std::printf("%4d %2d %5d %2d %2d %2d %2d %2d %2d | not representable (R %d, LDS %zu B)\n", c.NP, c.S, c.SW, c.LW, c.CW, c.D, c.P, c.XM, c.AT, H[0].V.R, Dv[0].lds_bytes);
      for (int t = 0; t < 2; ++t) release(&Dv[t]);
      continue;
    }
    const size_t lds = std::max(Dv[0].lds_bytes, Dv[1].lds_bytes);
    OK(hipFuncSetAttribute((const void*)var->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int T = (c.LW + c.CW) * 64;
    auto launch_pair = [&](int t, bool epi) {
      hipLaunchKernelGGL(var->fn, dim3(Dv[t].V.G), dim3(T), Dv[t].lds_bytes, st, Dv[t].V, (const double*)x[t], Dv[t].partial);
      if (epi)
        hipLaunchKernelGGL(k_combine<true>, dim3((rows[t] + 255) / 256), dim3(256), 0, st, rows[t], Dv[t].V.rows_pad, c.S, (const double*)Dv[t].partial, yout[t],
                           (const double*)e[0], (const double*)e[1], (const double*)e[2], e[3], 0.37);
      else
        hipLaunchKernelGGL(k_combine<false>, dim3((rows[t] + 255) / 256), dim3(256), 0, st, rows[t], Dv[t].V.rows_pad, c.S, (const double*)Dv[t].partial, yout[t],
                           (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, 0.0);
    };
    // correctness: plain products against the host's sequential sums
    double err[2] = {0, 0};
    for (int t = 0; t < 2; ++t) {
      launch_pair(t, false);
      OK(hipStreamSynchronize(st));
      OK(hipGetLastError());
      std::vector<double> got(rows[t]);
      OK(hipMemcpy(got.data(), yout[t], (size_t)rows[t] * 8, hipMemcpyDeviceToHost));
      for (int r = 0; r < rows[t]; ++r) {
        const double d = std::fabs(got[r] - ref[t][r]) / (mag[t][r] > 0 ? mag[t][r] : 1.0);
        if (!(d <= err[t])) err[t] = d;  // (NaN sticks)
      }
    }
    // run-to-run reproducibility (the additions' order is fixed by construction: two launches must agree bit for bit)
    bool same = true;
    {
      std::vector<double> g1(rows[0]), g2(rows[0]);
      launch_pair(0, false); OK(hipStreamSynchronize(st)); OK(hipMemcpy(g1.data(), yout[0], (size_t)rows[0] * 8, hipMemcpyDeviceToHost));
      launch_pair(0, false); OK(hipStreamSynchronize(st)); OK(hipMemcpy(g2.data(), yout[0], (size_t)rows[0] * 8, hipMemcpyDeviceToHost));
      same = std::memcmp(g1.data(), g2.data(), (size_t)rows[0] * 8) == 0;
    }
    // timing: touch (72 MB) -> A product -> combine + epilogue -> A^T product -> combine + epilogue, as the loop orders them
    double us[5] = {0, 0, 0, 0, 0};
    for (int r = -3; r < reps; ++r) {
      hipLaunchKernelGGL(k_touch, dim3((big + 255) / 256), dim3(256), 0, st, big, (const double*)tv[0], (const double*)tv[1], (const double*)tv[2], (const double*)tv[3],
                         (const double*)tv[4], tv[5], tv[6], tv[7]);
      OK(hipEventRecord(ev[0], st));
      hipLaunchKernelGGL(var->fn, dim3(Dv[0].V.G), dim3(T), Dv[0].lds_bytes, st, Dv[0].V, (const double*)x[0], Dv[0].partial);
      OK(hipEventRecord(ev[1], st));
      hipLaunchKernelGGL(k_combine<true>, dim3((rows[0] + 255) / 256), dim3(256), 0, st, rows[0], Dv[0].V.rows_pad, c.S, (const double*)Dv[0].partial, yout[0], (const double*)e[0],
                         (const double*)e[1], (const double*)e[2], e[3], 0.37);
      OK(hipEventRecord(ev[2], st));
      hipLaunchKernelGGL(var->fn, dim3(Dv[1].V.G), dim3(T), Dv[1].lds_bytes, st, Dv[1].V, (const double*)x[1], Dv[1].partial);
      OK(hipEventRecord(ev[3], st));
      hipLaunchKernelGGL(k_combine<true>, dim3((rows[1] + 255) / 256), dim3(256), 0, st, rows[1], Dv[1].V.rows_pad, c.S, (const double*)Dv[1].partial, yout[1], (const double*)e[0],
                         (const double*)e[1], (const double*)e[2], e[3], 0.37);
      OK(hipEventRecord(ev[4], st));
      OK(hipEventSynchronize(ev[4]));
      if (r >= 0)
        for (int i = 0; i < 4; ++i) {
          float ms = 0;
          OK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
          us[i] += 1e3 * ms;
        }
      if (r >= 0) { float ms = 0; OK(hipEventElapsedTime(&ms, ev[0], ev[4])); us[4] += 1e3 * ms; }
    }
    std::printf("%4d %2d %5d %2d %2d %2d %2d %2d %2d | %6d %7.1f %8lld %8lld | %9.1f %9.1f %9.1f %9.1f | %9.1f | %.2e %.2e %s  (build %.1f s, max cell %lld, split steps %lld, NS %d, G %d)\n", c.NP, c.S, c.SW, c.LW,
                c.CW, c.D, c.P, c.XM, c.AT, H[0].V.R, lds / 1024.0, H[0].pads + H[1].pads, H[0].dup_entries + H[1].dup_entries, us[0] / reps, us[1] / reps, us[2] / reps, us[3] / reps,
                us[4] / reps / 2, err[0], err[1], same ? "repro" : "NOT REPRODUCIBLE", H[0].build_s + H[1].build_s, std::max(H[0].max_cell, H[1].max_cell), H[0].split_steps + H[1].split_steps, H[0].V.NS, H[0].V.G);
    for (int t = 0; t < 2; ++t) release(&Dv[t]);
  }
  std::printf("(times: hipEvent pairs around single launches inside the touch -> A -> combine -> A^T -> combine sequence; 'pair' = (A + comb + AT + comb) / 2;\n"
              " the panels' fused a_dual / at_step are 72.6 / 71.1 us on the same matrices, plain products 66.7 us)\n");
  return 0;
}
