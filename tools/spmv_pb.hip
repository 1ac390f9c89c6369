// Round-3 tuning harness: a GATHER-FREE SpMV for unstructured matrices ("propagation blocking" with statically
// precomputed runs).  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/spmv_pb.hip -o tools/bin/spmv_pb
//   ./spmv_pb [rows=1000000] [nnz_per_row=10] [reps=20] [mode=sweep|selftest|prof] [config index for prof]
//
// out = M v is computed in two streaming phases, neither of which issues a single global-memory gather:
//   phase P ("products"): the columns of M are cut into SOURCE PANELS of SP columns; a workgroup copies its panel's
//     slice of v into LDS (a coalesced copy) and streams the panel's nonzeros -- value (8 B) + 16-bit column within the
//     panel -- multiplies, and writes the product to the SAME index of a product buffer (a pure stream: 10 B read +
//     8 B written per nonzero, every gather is an LDS read);
//   phase R ("rows"): the rows of M are cut into BINS whose nonzeros fit in LDS.  The nonzeros of a panel are stored
//     bin by bin (a CHUNK = panel x bin, sorted by (row, column), padded to pieces of G entries so that every piece is
//     an aligned 8*G-byte block), so a bin's products are S chunks, one per source panel.  The bin's workgroup copies
//     them into LDS (piece table: 4 B per piece), then lane <-> row adds up the row's products in column order, reading
//     16-bit LDS positions in jagged-diagonal order (rows sorted by length inside the bin: coalesced 2-byte stream).
//     A row is summed strictly left to right from 0.0: BIT-IDENTICAL to the sequential CSR sum of the oracle.
// 28 B per nonzero instead of 12, but all of it coalesced.  The sweep prices P and R separately for A and A^T inside a
// loop shaped like a PDHG iteration (element-wise kernel, P_A, R_A with the dual epilogue's streams, P_At, R_At with the
// step epilogue's streams) so that cache state between the kernels is what the solver would see.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

struct Csr {
  int rows = 0, cols = 0;
  std::vector<int> off, idx;
  std::vector<double> val;
  int64_t nnz() const { return (int64_t)idx.size(); }
};

static Csr transpose(const Csr& a)
{
  Csr t;
  t.rows = a.cols, t.cols = a.rows;
  t.off.assign(t.rows + 1, 0);
  for (int j : a.idx) t.off[j + 1]++;
  for (int i = 0; i < t.rows; ++i) t.off[i + 1] += t.off[i];
  t.idx.resize(a.idx.size());
  t.val.resize(a.idx.size());
  std::vector<int> cur(t.off.begin(), t.off.end() - 1);
  for (int r = 0; r < a.rows; ++r)
    for (int k = a.off[r]; k < a.off[r + 1]; ++k) {
      const int p = cur[a.idx[k]]++;
      t.idx[p] = r, t.val[p] = a.val[k];
    }
  return t;
}

static Csr make_matrix(int m, int n, int k, uint64_t seed)
{
  Csr a;
  a.rows = m, a.cols = n;
  a.off.resize(m + 1);
  a.idx.resize((size_t)m * k);
  a.val.resize((size_t)m * k);
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> nd;
  std::vector<int> c(k);
  for (int r = 0; r < m; ++r) {
    a.off[r] = r * k;
    for (;;) {
      for (int q = 0; q < k; ++q) c[q] = (int)(rng() % (uint64_t)n);
      std::sort(c.begin(), c.end());
      if (std::adjacent_find(c.begin(), c.end()) == c.end()) break;
    }
    for (int q = 0; q < k; ++q) a.idx[(size_t)r * k + q] = c[q], a.val[(size_t)r * k + q] = nd(rng);
  }
  a.off[m] = m * k;
  return a;
}

static void cpu_spmv(const Csr& a, const std::vector<double>& x, std::vector<double>& y)
{
  y.assign(a.rows, 0.0);
  for (int r = 0; r < a.rows; ++r) {
    double s = 0.0;
    for (int k = a.off[r]; k < a.off[r + 1]; ++k) s = s + a.val[k] * x[a.idx[k]];
    y[r] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side construction
// ---------------------------------------------------------------------------------------------------------------------
struct PbHost {
  int rows = 0, cols = 0, S = 0, B = 0, G = 0, cap = 0, max_panel = 0;
  int64_t np = 0;                    // padded entries (same count in P order and in the bins' images)
  std::vector<int> panel_start;      // S + 1: the source panels (column ranges of M)
  std::vector<double> val;           // np, P order (panel-major, then bin, then (row, column))
  std::vector<uint16_t> lidx;        // np: column within the source panel
  std::vector<int> piece_dst;        // np / G: where each piece of G products goes in the product buffer (bin-major)
  std::vector<int> wg_e0, wg_panel;  // P workgroups: entry range (multiples of G) and panel
  std::vector<int> bin_row0;         // B + 1
  std::vector<int> bin_e0;           // B + 1: the bin's image in the product buffer (contiguous)
  std::vector<uint32_t> sr;          // per bin (same extents as the rows): length << 16 | row within the bin, sorted by length
  std::vector<int> bin_grp;          // B + 1 into grp_pos
  std::vector<int> grp_pos;          // per 64-row group: first entry in pos
  std::vector<uint16_t> pos;         // nnz: position inside the bin's image, jagged-diagonal order
  int max_bin_entries = 0, max_bin_rows = 0;
  double avg_chunk = 0, pad = 0;
};

static std::vector<int> uniform_panels(int cols, int SP)
{
  std::vector<int> p;
  for (int c = 0; c < cols; c += SP) p.push_back(c);
  p.push_back(cols);
  return p;
}

static PbHost build_pb(const Csr& a, const std::vector<int>& panel_start, int cap, int G, int Q, int max_rows)
{
  PbHost h;
  h.rows = a.rows, h.cols = a.cols, h.G = G, h.cap = cap;
  h.panel_start = panel_start;
  h.S = (int)panel_start.size() - 1;
  const int S = h.S;
  std::vector<int> panel_of(a.cols);
  for (int s = 0; s < S; ++s) {
    for (int c = panel_start[s]; c < panel_start[s + 1]; ++c) panel_of[c] = s;
    h.max_panel = std::max(h.max_panel, panel_start[s + 1] - panel_start[s]);
  }
  if (h.max_panel > 65536) { printf("source panel too wide\n"); exit(1); }
  // 1. bins: consecutive rows while the padded entry count stays within cap
  {
    std::vector<int> cnt(S, 0), touched;
    int padded = 0, r0 = 0;
    h.bin_row0.push_back(0);
    auto add_row = [&](int r) {
      int add = 0;
      for (int k = a.off[r]; k < a.off[r + 1]; ++k) {
        const int s = panel_of[a.idx[k]];
        if (cnt[s] % G == 0) add += G;
        if (cnt[s] == 0) touched.push_back(s);
        cnt[s]++;
      }
      return add;
    };
    for (int r = 0; r < a.rows; ++r) {
      const int add = add_row(r);
      if (padded + add > cap || r - r0 >= max_rows) {  // close the bin before this row and start over with it
        for (int s : touched) cnt[s] = 0;
        touched.clear();
        h.bin_row0.push_back(r);
        r0     = r;
        padded = add_row(r);
        if (padded > cap) { printf("row %d alone overflows a bin: long-row path not in the harness\n", r); exit(1); }
      } else {
        padded += add;
      }
    }
    h.bin_row0.push_back(a.rows);
  }
  h.B = (int)h.bin_row0.size() - 1;
  const int B = h.B;
  std::vector<int> bin_of(a.rows);
  for (int b = 0; b < B; ++b) {
    for (int r = h.bin_row0[b]; r < h.bin_row0[b + 1]; ++r) bin_of[r] = b;
    h.max_bin_rows = std::max(h.max_bin_rows, h.bin_row0[b + 1] - h.bin_row0[b]);
  }
  // 2. chunk sizes
  std::vector<int> cnt((size_t)S * B, 0);
  for (int r = 0; r < a.rows; ++r)
    for (int k = a.off[r]; k < a.off[r + 1]; ++k) cnt[(size_t)panel_of[a.idx[k]] * B + bin_of[r]]++;
  // 3. padded starts in P order (panel-major) and in the product buffer (bin-major)
  std::vector<int64_t> pstart((size_t)S * B + 1, 0);
  int64_t nchunks = 0;
  for (size_t c = 0; c < (size_t)S * B; ++c) {
    pstart[c + 1] = pstart[c] + (cnt[c] + G - 1) / G * G;
    nchunks += cnt[c] > 0;
  }
  h.np = pstart[(size_t)S * B];
  if (h.np >= ((int64_t)1 << 31) - 4096) { printf("too many padded entries\n"); exit(1); }
  h.avg_chunk = (double)a.nnz() / (double)std::max<int64_t>(nchunks, 1);
  h.pad       = (double)h.np / (double)a.nnz();
  std::vector<int> lstart((size_t)B * S, 0);
  h.bin_e0.assign(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    int at = 0;
    for (int s = 0; s < S; ++s) {
      lstart[(size_t)b * S + s] = at;
      at += (cnt[(size_t)s * B + b] + G - 1) / G * G;
    }
    h.max_bin_entries = std::max(h.max_bin_entries, at);
    h.bin_e0[b + 1]   = h.bin_e0[b] + at;
  }
  if (h.max_bin_entries > cap || h.max_bin_entries > 65536) { printf("bin overflow %d > %d\n", h.max_bin_entries, cap); exit(1); }
  h.piece_dst.resize(h.np / G);
  for (int s = 0; s < S; ++s)
    for (int b = 0; b < B; ++b) {
      const size_t c = (size_t)s * B + b;
      const int np_  = (cnt[c] + G - 1) / G;
      for (int i = 0; i < np_; ++i) h.piece_dst[pstart[c] / G + i] = (h.bin_e0[b] + lstart[(size_t)b * S + s]) / G + i;
    }
  // 4. values / local columns in P order; position of every entry (CSR order) inside its bin's image
  h.val.assign(h.np, 0.0);
  h.lidx.assign(h.np, 0);
  std::vector<int> cursor((size_t)S * B, 0);
  std::vector<uint16_t> epos(a.nnz());
  for (int r = 0; r < a.rows; ++r) {
    const int b = bin_of[r];
    for (int k = a.off[r]; k < a.off[r + 1]; ++k) {
      const int s    = panel_of[a.idx[k]];
      const size_t c = (size_t)s * B + b;
      const int rank = cursor[c]++;
      h.val[pstart[c] + rank]  = a.val[k];
      h.lidx[pstart[c] + rank] = (uint16_t)(a.idx[k] - panel_start[s]);
      epos[k]                  = (uint16_t)(lstart[(size_t)b * S + s] + rank);
    }
  }
  // 5. row descriptors and positions: per bin rows sorted by length (descending, stable), groups of 64, jagged diagonals
  h.sr.resize(a.rows);
  h.pos.reserve(a.nnz() + 64);
  h.bin_grp.assign(B + 1, 0);
  std::vector<int> order;
  for (int b = 0; b < B; ++b) {
    const int r0 = h.bin_row0[b], nr = h.bin_row0[b + 1] - r0;
    order.resize(nr);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a.off[r0 + x + 1] - a.off[r0 + x] > a.off[r0 + y + 1] - a.off[r0 + y]; });
    for (int i = 0; i < nr; ++i) {
      const int len = a.off[r0 + order[i] + 1] - a.off[r0 + order[i]];
      h.sr[r0 + i]  = ((uint32_t)len << 16) | (uint32_t)order[i];
    }
    for (int g0 = 0; g0 < nr; g0 += 64) {
      h.grp_pos.push_back((int)h.pos.size());
      const int g1   = std::min(nr, g0 + 64);
      const int kmax = a.off[r0 + order[g0] + 1] - a.off[r0 + order[g0]];
      for (int k = 0; k < kmax; ++k)
        for (int i = g0; i < g1; ++i) {
          const int r = r0 + order[i];
          if (a.off[r + 1] - a.off[r] <= k) break;
          h.pos.push_back(epos[a.off[r] + k]);
        }
    }
    h.bin_grp[b + 1] = (int)h.grp_pos.size();
  }
  h.grp_pos.push_back((int)h.pos.size());
  h.pos.resize(h.pos.size() + 64, 0);
  // 6. P workgroups: every panel's entry range in Q parts
  for (int s = 0; s < S; ++s) {
    const int64_t e0 = pstart[(size_t)s * B], e1 = pstart[(size_t)(s + 1) * B];
    const int64_t per = std::max<int64_t>(G, ((e1 - e0 + Q - 1) / Q + G - 1) / G * G);
    if (e0 == e1) {  // keep one (empty) workgroup per panel: a fused producer indexes workgroups by panel
      h.wg_e0.push_back((int)e0);
      h.wg_panel.push_back(s);
    }
    for (int64_t e = e0; e < e1; e += per) {
      h.wg_e0.push_back((int)e);
      h.wg_panel.push_back(s);
    }
  }
  h.wg_e0.push_back((int)h.np);
  return h;
}

// CPU model of the two phases (checks the construction without a GPU)
static void cpu_pb(const PbHost& h, const std::vector<double>& x, std::vector<double>& y)
{
  std::vector<double> prod(h.np);
  const int nwg = (int)h.wg_panel.size();
  for (int w = 0; w < nwg; ++w)
    for (int e = h.wg_e0[w]; e < h.wg_e0[w + 1]; ++e) {
      const int c = h.panel_start[h.wg_panel[w]] + h.lidx[e];
      prod[(size_t)h.piece_dst[e / h.G] * h.G + e % h.G] = h.val[e] * (c < h.cols ? x[c] : 0.0);
    }
  y.assign(h.rows, 0.0);
  for (int b = 0; b < h.B; ++b) {
    const double* lds = prod.data() + h.bin_e0[b];
    const int r0 = h.bin_row0[b], nr = h.bin_row0[b + 1] - r0;
    for (int g = 0; g * 64 < nr; ++g) {
      int e        = h.grp_pos[h.bin_grp[b] + g];
      const int i0 = g * 64, i1 = std::min(nr, i0 + 64);
      std::vector<double> sum(64, 0.0);
      const int kmax = (int)(h.sr[r0 + i0] >> 16);
      for (int k = 0; k < kmax; ++k)
        for (int i = i0; i < i1 && (int)(h.sr[r0 + i] >> 16) > k; ++i) sum[i - i0] = sum[i - i0] + lds[h.pos[e++]];
      for (int i = i0; i < i1; ++i) y[r0 + (h.sr[r0 + i] & 0xFFFF)] = sum[i - i0];
    }
  }
}

// slab-major row panels (the gather layout of the product, in its simplest form): W panels balanced by nonzeros, S slabs
struct SlabHost {
  int W = 0, S = 0;
  std::vector<int> row0;      // W + 1
  std::vector<int> tile_ptr;  // W * S + 1
  std::vector<int> rowptr;    // per tile: rows_w + 1 absolute positions
  std::vector<int64_t> rp_base;
  std::vector<int> col;
  std::vector<double> val;
};
static SlabHost build_slabs(const Csr& a, int W, int S)
{
  SlabHost h;
  h.W = W, h.S = S;
  const int64_t per = (a.nnz() + W - 1) / W;
  h.row0.push_back(0);
  for (int w = 1; w < W; ++w) {
    const int64_t target = per * w;
    int r = (int)(std::lower_bound(a.off.begin(), a.off.end(), (int)target) - a.off.begin());
    r     = std::max(r, h.row0.back());
    h.row0.push_back(std::min(r, a.rows));
  }
  h.row0.push_back(a.rows);
  const int slab_w = (a.cols + S - 1) / S;
  h.col.resize(a.nnz() + 64, 0);
  h.val.resize(a.nnz() + 64, 0.0);
  h.tile_ptr.assign((size_t)W * S + 1, 0);
  int64_t at = 0;
  for (int w = 0; w < W; ++w) {
    const int r0 = h.row0[w], r1 = h.row0[w + 1];
    std::vector<int> cur(a.off.begin() + r0, a.off.begin() + r1);
    for (int s = 0; s < S; ++s) {
      h.rp_base.push_back((int64_t)h.rowptr.size());
      const int cend = (s + 1) * slab_w;
      for (int r = r0; r < r1; ++r) {
        h.rowptr.push_back((int)at);
        int k = cur[r - r0];
        while (k < a.off[r + 1] && a.idx[k] < cend) {
          h.col[at] = a.idx[k], h.val[at] = a.val[k];
          ++at, ++k;
        }
        cur[r - r0] = k;
      }
      h.rowptr.push_back((int)at);
      h.tile_ptr[(size_t)w * S + s + 1] = (int)at;
    }
  }
  return h;
}

// ---------------------------------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------------------------------
typedef double vec2d __attribute__((ext_vector_type(2)));

struct PbView {
  int rows, cols, S, B, cap, nwg, gshift, max_panel;  // gshift = log2(G)
  const int* __restrict__ panel_start;
  const double* __restrict__ val;
  const uint16_t* __restrict__ lidx;
  const int* __restrict__ piece_dst;
  const int* __restrict__ wg_e0;
  const int* __restrict__ wg_panel;
  const int* __restrict__ bin_row0;
  const int* __restrict__ bin_e0;
  const uint32_t* __restrict__ sr;
  const int* __restrict__ bin_grp;
  const int* __restrict__ grp_pos;
  const uint16_t* __restrict__ pos;
};

struct Streams {
  const double *a, *b, *c;  // read per row
  double* d;                // read + written per row (dual epilogue)
  double* out;
  double* part;
  double sigma, weight;
};

__device__ __forceinline__ int xcd_remap(int b, int nb)
{
  const int per = (nb + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// the product stream of one source panel: prod[dst(e)] = val[e] * xs[lidx[e]] with the panel's slice of the vector in LDS.
// Two entries per lane and request (16-byte loads and stores); a piece of G entries goes to one aligned 8 G-byte block.
template <int THREADS, int U, bool NTS>
__device__ __forceinline__ void emit_products(const PbView& V, const double* xs, int e0, int e1, double* __restrict__ prod)
{
  const int gmask = (1 << V.gshift) - 1;
  for (int e = e0 + 2 * (int)threadIdx.x; e < e1; e += 2 * THREADS * U) {
    vec2d a[U];
    uint32_t j[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      a[u] = (vec2d)(0.0), j[u] = 0, dst[u] = 0;
      if (ee < e1) {
        a[u]   = __builtin_nontemporal_load(reinterpret_cast<const vec2d*>(V.val + ee));
        j[u]   = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(V.lidx + ee));
        dst[u] = __builtin_nontemporal_load(V.piece_dst + (ee >> V.gshift));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u * 2 * THREADS;
      if (ee < e1) {
        vec2d p;
        p.x = a[u].x * xs[j[u] & 0xFFFFu];
        p.y = a[u].y * xs[j[u] >> 16];
        vec2d* out = reinterpret_cast<vec2d*>(prod + (((int64_t)dst[u] << V.gshift) + (ee & gmask)));
        if constexpr (NTS) __builtin_nontemporal_store(p, out);
        else *out = p;
      }
    }
  }
}

// phase P as its own kernel
template <int THREADS, int U, bool NTS>
__global__ void __launch_bounds__(THREADS) k_pb_p(PbView V, const double* __restrict__ vec, double* __restrict__ prod)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  double* xs  = pb_lds;
  const int w = xcd_remap((int)blockIdx.x, V.nwg);
  if (w >= V.nwg) return;
  const int panel = V.wg_panel[w];
  const int c0 = V.panel_start[panel], len = V.panel_start[panel + 1] - c0;
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
  emit_products<THREADS, U, NTS>(V, xs, V.wg_e0[w], V.wg_e0[w + 1], prod);
}

template <int WAVES, int NQ>
__device__ __forceinline__ void block_sum(double (&v)[NQ], double* scratch)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double x = v[q];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (lane == 0) scratch[wave * NQ + q] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double acc = scratch[q];
      for (int w = 1; w < WAVES; ++w) acc += scratch[w * NQ + q];
      v[q] = acc;
    }
  }
}

// phase R: the bin's image (contiguous) -> LDS by LDS-DMA, row sums in column order, epilogue in natural row order.
// Everything the workgroup will need (row descriptors, the first KU jagged diagonals of positions, the epilogue's
// operands) is requested before the one barrier that waits for the image.
// EPI 0: out = sum.  1: the dual update's streams (3 read, 1 read+written, 1 written, one partial).  2: the step
// statistics' streams (3 read, 1 written, two partials).
template <int THREADS, int EPI, int KU, bool DMA>
__global__ void __launch_bounds__(THREADS) k_pb_r(PbView V, const double* __restrict__ prod, Streams S)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  double* lp          = pb_lds;
  double* strip       = pb_lds + V.cap;
  constexpr int WAVES = THREADS / 64;
  constexpr int GR    = 2;  // groups of 64 rows per wave: bins hold at most 2 * THREADS rows
  const int lane      = threadIdx.x & 63;
  const int wave      = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b         = xcd_remap((int)blockIdx.x, V.B);
  if (b >= V.B) return;
  const int e0     = V.bin_e0[b];
  const int nunits = (V.bin_e0[b + 1] - e0) >> 1;  // 16-byte units
  if constexpr (DMA) {
    for (int u0 = wave * 64; u0 < nunits; u0 += WAVES * 64)  // the tail reads past the image: buffer and LDS are padded
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prod + e0 + 2 * (u0 + lane)),
                                       (__attribute__((address_space(3))) void*)(lp + 2 * u0), 16, 0, 0);
  }
  vec2d stage[DMA ? 1 : 10];
  if constexpr (!DMA) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int un = threadIdx.x + u * THREADS;
      stage[u]     = (vec2d)(0.0);
      if (un < nunits) stage[u] = __builtin_nontemporal_load(reinterpret_cast<const vec2d*>(prod + e0 + 2 * un));
    }
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
  uint32_t p[GR][KU];
#pragma unroll
  for (int q = 0; q < GR; ++q) {
    const int cnt = (int)(d[q] >> 16);
    int e         = __builtin_amdgcn_readfirstlane(eg[q]);
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int at = e;
      e += __builtin_popcountll(__ballot(cnt > u));
      p[q][u] = 0;
      if (cnt > u) p[q][u] = __builtin_nontemporal_load(V.pos + at + lane);
    }
    eg[q] = e;
  }
  double oa[2], ob[2], oc[2], od[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = threadIdx.x + q * THREADS;
    oa[q] = ob[q] = oc[q] = od[q] = 0.0;
    if (EPI != 0 && i < brows) {
      oa[q] = S.a[row0 + i], ob[q] = S.b[row0 + i], oc[q] = S.c[row0 + i];
      if (EPI == 1) od[q] = S.d[row0 + i];
    }
  }
  if constexpr (!DMA) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int un = threadIdx.x + u * THREADS;
      if (un < nunits) *reinterpret_cast<vec2d*>(lp + 2 * un) = stage[u];
    }
    for (int un = threadIdx.x + 10 * THREADS; un < nunits; un += THREADS)
      *reinterpret_cast<vec2d*>(lp + 2 * un) = *reinterpret_cast<const vec2d*>(prod + e0 + 2 * un);
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
      for (int u = 0; u < KU; ++u)
        if (cnt > u) sum = sum + lp[p[q][u]];
      const int kmax = __builtin_amdgcn_readfirstlane(cnt);  // sorted: lane 0 holds the longest row of the group
      int e          = eg[q];
      for (int k = KU; k < kmax; ++k) {  // rows longer than the prefetched diagonals
        const int at = e;
        e += __builtin_popcountll(__ballot(cnt > k));
        if (cnt > k) sum = sum + lp[V.pos[at + lane]];
      }
      if (g * 64 + lane < brows) strip[lrow] = sum;
    }
  }
  __syncthreads();
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = threadIdx.x + q * THREADS;
    if (i < brows) {
      const int row  = row0 + i;
      const double v = strip[i];
      if constexpr (EPI == 0) {
        S.out[row] = v;
      } else if constexpr (EPI == 1) {
        const double yi = oa[q];
        double next     = yi - (S.sigma * v);
        const double lo = next + S.sigma * ob[q];
        const double up = next + S.sigma * oc[q];
        next            = fmax(lo, fmin(up, 0.0));
        S.out[row]      = next;
        const double dy = next - yi;
        acc[0] += dy * dy;
        S.d[row] = od[q] + S.weight * yi;
      } else {
        S.out[row]      = v;
        const double dx = ob[q] - oa[q];
        const double t  = v - oc[q];
        acc[0] += t * dx;
        acc[1] += dx * dx;
      }
    }
  }
  if constexpr (EPI != 0) {
    __syncthreads();
    block_sum<WAVES, 2>(acc, lp);
    if (threadIdx.x == 0) {
      S.part[b]       = acc[0];
      S.part[V.B + b] = acc[1];
    }
  }
}

// the gather side of the fused variant: slab-major row panels (simplest form of the product's panel kernel), y = A x with the
// plain store epilogue; with EMIT the workgroup then streams its panel's nonzeros a second time (emission order: by column
// bin of A^T's phase R) and writes a_ij * y_i into the product buffer -- the next kernel (phase R of A^T) never gathers.
template <int T, int CH, bool EMIT>
__global__ void __launch_bounds__(T) k_slab(int S, const int* __restrict__ panel_row0, const int* __restrict__ tile_ptr,
                                            const int* __restrict__ rowptr, const int64_t* __restrict__ rp_base,
                                            const int* __restrict__ col, const double* __restrict__ val,
                                            const double* __restrict__ x, double* __restrict__ y, PbView V, double* __restrict__ prod)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  double* pr   = pb_lds;
  double* psum = pb_lds + CH;
  const int w = blockIdx.x, r0 = panel_row0[w], nr = panel_row0[w + 1] - r0;
  for (int r = threadIdx.x; r < nr; r += T) psum[r] = 0.0;
  for (int s = 0; s < S; ++s) {
    const int t0 = tile_ptr[w * S + s], t1 = tile_ptr[w * S + s + 1];
    const int* __restrict__ rp = rowptr + rp_base[w * S + s];
    for (int c0 = t0; c0 < t1; c0 += CH) {
      const int c1 = c0 + CH < t1 ? c0 + CH : t1;
      __syncthreads();
#pragma unroll 4
      for (int k = c0 + threadIdx.x; k < c1; k += T) {
        const double a = __builtin_nontemporal_load(val + k);
        const int j    = __builtin_nontemporal_load(col + k);
        pr[k - c0]     = a * x[j];
      }
      __syncthreads();
      for (int r = threadIdx.x; r < nr; r += T) {
        int a = rp[r], b = rp[r + 1];
        a = a > c0 ? a : c0;
        b = b < c1 ? b : c1;
        if (a < b) {
          double sum = psum[r];
          for (int k = a; k < b; ++k) sum = sum + pr[k - c0];
          psum[r] = sum;
        }
      }
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < nr; r += T) y[r0 + r] = psum[r];
  if constexpr (EMIT) emit_products<T, 4, false>(V, psum, V.wg_e0[w], V.wg_e0[w + 1], prod);
}

// the element-wise kernel of an iteration: 5 streams read, 4 written (k_primal's traffic)
__global__ void __launch_bounds__(256) k_elementwise(int n, const double* __restrict__ x, const double* __restrict__ aty,
                                                     const double* __restrict__ c, const double* __restrict__ lb,
                                                     double* __restrict__ xn, double* __restrict__ xbar, double* __restrict__ sumx)
{
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const double xj = x[j];
    double next     = xj - 0.001 * (c[j] - aty[j]);
    next            = fmax(next, lb[j]);
    xn[j]           = next;
    xbar[j]         = next - xj + next;
    sumx[j]         = sumx[j] + 0.5 * xj;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
template <class T>
static T* upload(const std::vector<T>& v, size_t pad = 256)
{
  T* d;
  CK(hipMalloc(&d, (v.size() + pad) * sizeof(T)));
  CK(hipMemset(d, 0, (v.size() + pad) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

struct PbDev {
  PbView v;
  double* prod;
  std::vector<void*> owned;
  int nwg;
};

static PbDev to_device(const PbHost& h)
{
  PbDev d;
  auto keep = [&](auto* p) { d.owned.push_back((void*)p); return p; };
  int gs = 0;
  while ((1 << gs) < h.G) ++gs;
  d.v = PbView{h.rows, h.cols, h.S, h.B, h.cap, (int)h.wg_panel.size(), gs, h.max_panel,
               keep(upload(h.panel_start)), keep(upload(h.val)), keep(upload(h.lidx)), keep(upload(h.piece_dst)),
               keep(upload(h.wg_e0)), keep(upload(h.wg_panel)), keep(upload(h.bin_row0)), keep(upload(h.bin_e0)),
               keep(upload(h.sr)), keep(upload(h.bin_grp)), keep(upload(h.grp_pos)), keep(upload(h.pos))};
  CK(hipMalloc(&d.prod, (h.np + 256) * 8));
  CK(hipMemset(d.prod, 0, (h.np + 256) * 8));
  d.nwg = (int)h.wg_panel.size();
  return d;
}
static void release(PbDev& d)
{
  for (void* p : d.owned) CK(hipFree(p));
  CK(hipFree(d.prod));
}

struct Side {
  const Csr* m;
  std::vector<double> hx, ref;
  double *x, *y, *a, *b, *c, *d, *part;
};
static Side make_side(const Csr& m, uint64_t seed)
{
  Side s;
  s.m = &m;
  s.hx.resize(m.cols);
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> nd;
  for (auto& v : s.hx) v = nd(rng);
  cpu_spmv(m, s.hx, s.ref);
  s.x = upload(s.hx);
  std::vector<double> z(m.rows, 0.25);
  s.y = upload(z), s.a = upload(z), s.b = upload(z), s.c = upload(z), s.d = upload(z);
  CK(hipMalloc(&s.part, 2 * 16384 * 8));
  return s;
}
static void restore(Side& s) { CK(hipMemcpy(s.x, s.hx.data(), s.hx.size() * 8, hipMemcpyHostToDevice)); }

struct Config {
  int SP, Q, cap, G, threadsR, threadsP;
  bool dma, nts;
  const char* note;
};

template <int THREADS, bool NTS>
static void launch_p(const PbDev& d, const double* vec)
{
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)k_pb_p<THREADS, 4, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  const int grid = (d.nwg + 7) / 8 * 8;
  k_pb_p<THREADS, 4, NTS><<<grid, THREADS, (size_t)d.v.max_panel * 8>>>(d.v, vec, d.prod);
}
template <int THREADS, int EPI, bool DMA>
static void launch_r(const PbDev& d, const Streams& st, int max_rows)
{
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)k_pb_r<THREADS, EPI, 16, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  const int grid = (d.v.B + 7) / 8 * 8;
  k_pb_r<THREADS, EPI, 16, DMA><<<grid, THREADS, (size_t)(d.v.cap + max_rows + 128) * 8>>>(d.v, d.prod, st);
}

static void run_p(const Config& c, const PbDev& d, const double* vec)
{
  if (c.threadsP == 1024) {
    if (c.nts) launch_p<1024, true>(d, vec);
    else launch_p<1024, false>(d, vec);
  } else {
    if (c.nts) launch_p<512, true>(d, vec);
    else launch_p<512, false>(d, vec);
  }
}
template <int EPI>
static void run_r(const Config& c, const PbDev& d, const Streams& st, int max_rows)
{
  if (c.threadsR == 1024) {
    if (c.dma) launch_r<1024, EPI, true>(d, st, max_rows);
    else launch_r<1024, EPI, false>(d, st, max_rows);
  } else if (c.threadsR == 256) {
    if (c.dma) launch_r<256, EPI, true>(d, st, max_rows);
    else launch_r<256, EPI, false>(d, st, max_rows);
  } else {
    if (c.dma) launch_r<512, EPI, true>(d, st, max_rows);
    else launch_r<512, EPI, false>(d, st, max_rows);
  }
}

static bool check(const double* dev, const std::vector<double>& ref, const char* tag)
{
  std::vector<double> got(ref.size());
  CK(hipMemcpy(got.data(), dev, got.size() * 8, hipMemcpyDeviceToHost));
  if (memcmp(got.data(), ref.data(), got.size() * 8) == 0) return true;
  int64_t bad = 0;
  int first   = -1;
  for (size_t r = 0; r < ref.size(); ++r)
    if (memcmp(&got[r], &ref[r], 8) != 0) {
      if (first < 0) first = (int)r;
      ++bad;
    }
  printf("    !! %s: %lld rows differ (first %d: %.17g vs %.17g)\n", tag, (long long)bad, first, got[first], ref[first]);
  return false;
}

static int max_rows_of(const Config& c) { return 2 * c.threadsR; }

// the unfused variant: P and R for both products
static void run_config(const Config& c, const Csr& a, const Csr& at, Side& A, Side& At, int reps, bool prof_only)
{
  PbHost ha = build_pb(a, uniform_panels(a.cols, c.SP), c.cap, c.G, c.Q, max_rows_of(c));
  PbHost hat = build_pb(at, uniform_panels(at.cols, c.SP), c.cap, c.G, c.Q, max_rows_of(c));
  PbDev da = to_device(ha), dat = to_device(hat);
  const int n = a.cols;
  restore(A), restore(At);
  // correctness: plain epilogue, bit-exact against the sequential CSR sum
  Streams sa{A.a, A.b, A.c, A.d, A.y, A.part, 0.01, 0.5}, sat{At.a, At.b, At.c, At.d, At.y, At.part, 0.01, 0.5};
  run_p(c, da, A.x);
  run_r<0>(c, da, sa, ha.max_bin_rows);
  run_p(c, dat, At.x);
  run_r<0>(c, dat, sat, hat.max_bin_rows);
  CK(hipDeviceSynchronize());
  const bool ok = check(A.y, A.ref, "A") & check(At.y, At.ref, "At");
  // the loop of an iteration: elementwise, P_A, R_A (dual streams), P_At, R_At (step streams)
  hipEvent_t ev[6];
  for (auto& e : ev) CK(hipEventCreate(&e));
  double t[5] = {0, 0, 0, 0, 0};
  auto iteration = [&](bool timed) {
    if (timed) CK(hipEventRecord(ev[0]));
    k_elementwise<<<2048, 256>>>(n, At.a, At.b, At.c, At.d, At.y, A.x, At.x /* stand-ins: 5 read, 4 written */);
    if (timed) CK(hipEventRecord(ev[1]));
    run_p(c, da, A.x);
    if (timed) CK(hipEventRecord(ev[2]));
    run_r<1>(c, da, sa, ha.max_bin_rows);
    if (timed) CK(hipEventRecord(ev[3]));
    run_p(c, dat, A.y);  // y' -> products of A^T
    if (timed) CK(hipEventRecord(ev[4]));
    run_r<2>(c, dat, sat, hat.max_bin_rows);
    if (timed) CK(hipEventRecord(ev[5]));
  };
  for (int w = 0; w < 3; ++w) iteration(false);
  CK(hipDeviceSynchronize());
  if (prof_only) {
    for (int r = 0; r < reps; ++r) iteration(false);
    CK(hipDeviceSynchronize());
  } else {
    for (int r = 0; r < reps; ++r) {
      iteration(true);
      CK(hipEventSynchronize(ev[5]));
      for (int q = 0; q < 5; ++q) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[q], ev[q + 1]));
        t[q] += ms * 1e3 / reps;
      }
    }
    CK(hipEventRecord(ev[0]));
    for (int r = 0; r < reps; ++r) iteration(false);
    CK(hipEventRecord(ev[1]));
    CK(hipEventSynchronize(ev[1]));
    float ms;
    CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    const double loop = ms * 1e3 / reps;
    const double mb_p = (ha.np * 18.0 + ha.np / c.G * 4.0 + n * 8.0) * 1e-6, mb_ra = (ha.np * 8.0 + a.nnz() * 2.0 + a.rows * 52.0) * 1e-6,
                 mb_rat = (hat.np * 8.0 + a.nnz() * 2.0 + at.rows * 36.0) * 1e-6;
    printf("PR   SP %5d Q %d cap %5d G %2d thrR %4d dma %d nts %d | bins %4d/%4d chunk %5.1f pad %.3f | elem %5.1f  P_A %5.1f (%.2f TB/s)  R_A %5.1f (%.2f)  "
           "P_At %5.1f  R_At %5.1f (%.2f) | 4 kernels %6.1f us, loop %6.1f us %s %s\n",
           c.SP, c.Q, c.cap, c.G, c.threadsR, (int)c.dma, (int)c.nts, ha.B, hat.B, ha.avg_chunk, ha.pad, t[0], t[1], mb_p / t[1], t[2],
           mb_ra / t[2], t[3], t[4], mb_rat / t[4], t[1] + t[2] + t[3] + t[4], loop, ok ? "ok" : "WRONG", c.note);
    fflush(stdout);
  }
  for (auto& e : ev) CK(hipEventDestroy(e));
  release(da);
  release(dat);
}

// the fused variant: A x by the gather kernel (slab-major row panels) which also EMITS a_ij * y_i; A^T y by phase R alone
struct FusedConfig {
  int W, threads, S, cap, G, threadsR;
  bool dma;
  const char* note;
};

template <int T, bool EMIT>
static void launch_slab(const SlabHost& h, const int* row0, const int* tile_ptr, const int* rowptr, const int64_t* rp_base, const int* col,
                        const double* val, const double* x, double* y, const PbDev& d, int max_rows)
{
  constexpr int CH = 4096;
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)k_slab<T, CH, EMIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  k_slab<T, CH, EMIT><<<h.W, T, (size_t)(CH + max_rows) * 8>>>(h.S, row0, tile_ptr, rowptr, rp_base, col, val, x, y, d.v, d.prod);
}

static void run_fused(const FusedConfig& c, const Csr& a, const Csr& at, Side& A, Side& At, int reps, bool prof_only)
{
  SlabHost sh = build_slabs(a, c.W, c.S);
  int max_rows = 0;
  for (int w = 0; w < sh.W; ++w) max_rows = std::max(max_rows, sh.row0[w + 1] - sh.row0[w]);
  // A^T's phase R consumes products emitted per ROW PANEL of A: the source panels of A^T's layout are those row ranges
  Config rc{0, 1, c.cap, c.G, c.threadsR, c.threads, c.dma, false, ""};
  PbHost hat = build_pb(at, sh.row0, c.cap, c.G, 1, max_rows_of(rc));
  if ((int)hat.wg_panel.size() != sh.W) { printf("workgroup / panel mismatch\n"); exit(1); }
  PbDev dat = to_device(hat);
  int* d_row0 = upload(sh.row0); int* d_tp = upload(sh.tile_ptr); int* d_rp = upload(sh.rowptr);
  int64_t* d_rb = upload(sh.rp_base); int* d_col = upload(sh.col); double* d_val = upload(sh.val);
  restore(A), restore(At);
  Streams sat0{At.a, At.b, At.c, At.d, At.y, At.part, 0.01, 0.5};
  auto slab = [&](bool emit) {
    if (c.threads == 1024) {
      if (emit) launch_slab<1024, true>(sh, d_row0, d_tp, d_rp, d_rb, d_col, d_val, A.x, A.y, dat, max_rows);
      else launch_slab<1024, false>(sh, d_row0, d_tp, d_rp, d_rb, d_col, d_val, A.x, A.y, dat, max_rows);
    } else {
      if (emit) launch_slab<512, true>(sh, d_row0, d_tp, d_rp, d_rb, d_col, d_val, A.x, A.y, dat, max_rows);
      else launch_slab<512, false>(sh, d_row0, d_tp, d_rp, d_rb, d_col, d_val, A.x, A.y, dat, max_rows);
    }
  };
  // correctness: y = A x bit-exact, then A^T y from the emitted products bit-exact against the CPU's A^T (A x)
  slab(true);
  run_r<0>(rc, dat, sat0, hat.max_bin_rows);
  CK(hipDeviceSynchronize());
  std::vector<double> ref2;
  cpu_spmv(at, A.ref, ref2);
  const bool ok = check(A.y, A.ref, "A x") & check(At.y, ref2, "At (A x)");
  hipEvent_t ev[4];
  for (auto& e : ev) CK(hipEventCreate(&e));
  const int n = a.cols;
  double t[2][3] = {{0, 0, 0}, {0, 0, 0}}, loop[2] = {0, 0};
  for (int emit = 0; emit < 2; ++emit) {
    auto iteration = [&](bool timed) {
      if (timed) CK(hipEventRecord(ev[0]));
      k_elementwise<<<2048, 256>>>(n, At.a, At.b, At.c, At.d, At.d, A.x, At.x);
      if (timed) CK(hipEventRecord(ev[1]));
      slab(emit != 0);
      if (timed) CK(hipEventRecord(ev[2]));
      run_r<2>(rc, dat, sat0, hat.max_bin_rows);
      if (timed) CK(hipEventRecord(ev[3]));
    };
    for (int w = 0; w < 3; ++w) iteration(false);
    CK(hipDeviceSynchronize());
    if (prof_only) {
      for (int r = 0; r < reps; ++r) iteration(false);
      CK(hipDeviceSynchronize());
      continue;
    }
    for (int r = 0; r < reps; ++r) {
      iteration(true);
      CK(hipEventSynchronize(ev[3]));
      for (int q = 0; q < 3; ++q) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[q], ev[q + 1]));
        t[emit][q] += ms * 1e3 / reps;
      }
    }
    CK(hipEventRecord(ev[0]));
    for (int r = 0; r < reps; ++r) iteration(false);
    CK(hipEventRecord(ev[1]));
    CK(hipEventSynchronize(ev[1]));
    float ms;
    CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    loop[emit] = ms * 1e3 / reps;
  }
  if (!prof_only) {
    const double mb_rat = (hat.np * 8.0 + a.nnz() * 2.0 + at.rows * 36.0) * 1e-6;
    printf("FUSE W %4d thr %4d S %d cap %5d G %2d thrR %4d dma %d | bins %4d chunk %5.1f pad %.3f | gather kernel %5.1f us, with emission %5.1f us (+%.1f)  "
           "R_At %5.1f (%.2f TB/s) | a_dual+emit + R_At = %6.1f us %s %s\n",
           c.W, c.threads, c.S, c.cap, c.G, c.threadsR, (int)c.dma, hat.B, hat.avg_chunk, hat.pad, t[0][1], t[1][1], t[1][1] - t[0][1], t[1][2],
           mb_rat / t[1][2], t[1][1] + t[1][2], ok ? "ok" : "WRONG", c.note);
    fflush(stdout);
  }
  for (auto& e : ev) CK(hipEventDestroy(e));
  release(dat);
  for (void* p : {(void*)d_row0, (void*)d_tp, (void*)d_rp, (void*)d_rb, (void*)d_col, (void*)d_val}) CK(hipFree(p));
}

// rows [r0, r1) of a as a matrix of its own
static Csr sub_rows(const Csr& a, int r0, int r1)
{
  Csr s;
  s.rows = r1 - r0, s.cols = a.cols;
  s.off.resize(s.rows + 1);
  for (int i = 0; i <= s.rows; ++i) s.off[i] = a.off[r0 + i] - a.off[r0];
  s.idx.assign(a.idx.begin() + a.off[r0], a.idx.begin() + a.off[r1]);
  s.val.assign(a.val.begin() + a.off[r0], a.val.begin() + a.off[r1]);
  return s;
}

// HYBRID: the two designs are bound by different resources (the gather kernel by the texture-address path and the L2 miss
// slots, the gather-free pair by HBM bandwidth), so the first `share` of the rows goes through the gather kernel on one stream
// while the rest goes through P + R on a second one.  Rows are independent: every row is still summed left to right.
static void run_hybrid(double share, int W, const Csr& a, Side& A, int reps)
{
  const int rs = (int)((double)a.rows * share);
  Csr top = sub_rows(a, 0, rs), bot = sub_rows(a, rs, a.rows);
  SlabHost sh = build_slabs(top, W, 6);
  int max_rows = 0;
  for (int w = 0; w < sh.W; ++w) max_rows = std::max(max_rows, sh.row0[w + 1] - sh.row0[w]);
  Config pc{8192, 4, 9088, 8, 512, 512, false, false, ""};
  PbHost hb = build_pb(bot, uniform_panels(bot.cols, pc.SP), pc.cap, pc.G, pc.Q, max_rows_of(pc));
  PbDev db = to_device(hb);
  int* d_row0 = upload(sh.row0); int* d_tp = upload(sh.tile_ptr); int* d_rp = upload(sh.rowptr);
  int64_t* d_rb = upload(sh.rp_base); int* d_col = upload(sh.col); double* d_val = upload(sh.val);
  restore(A);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t fork, join, e0, e1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Streams st{A.a, A.b, A.c, A.d, A.y + rs, A.part, 0.01, 0.5};
  constexpr int CH = 4096;
  CK(hipFuncSetAttribute((const void*)k_slab<512, CH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_pb_p<512, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_pb_r<512, 0, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto gather_part = [&](hipStream_t s) {
    k_slab<512, CH, false><<<sh.W, 512, (size_t)(CH + max_rows) * 8, s>>>(sh.S, d_row0, d_tp, d_rp, d_rb, d_col, d_val, A.x, A.y, db.v, db.prod);
  };
  auto pb_part = [&](hipStream_t s) {
    k_pb_p<512, 4, false><<<(db.nwg + 7) / 8 * 8, 512, (size_t)db.v.max_panel * 8, s>>>(db.v, A.x, db.prod);
    k_pb_r<512, 0, 16, false><<<(db.v.B + 7) / 8 * 8, 512, (size_t)(db.v.cap + hb.max_bin_rows + 128) * 8, s>>>(db.v, db.prod, st);
  };
  auto timeit = [&](int mode) {  // 0: both parts one after the other on one stream, 1: concurrently, gather first, 2: concurrently, P first
    for (int rep = -3; rep < reps; ++rep) {
      if (rep == 0) CK(hipEventRecord(e0, s1));
      k_elementwise<<<2048, 256, 0, s1>>>(a.cols, A.a, A.b, A.c, A.d, A.d, A.c, A.b);
      if (mode == 0) {
        gather_part(s1);
        pb_part(s1);
      } else {
        CK(hipEventRecord(fork, s1));
        CK(hipStreamWaitEvent(s2, fork, 0));
        if (mode == 1) { gather_part(s1); pb_part(s2); }
        else { pb_part(s2); gather_part(s1); }
        CK(hipEventRecord(join, s2));
        CK(hipStreamWaitEvent(s1, join, 0));
      }
    }
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
  };
  // correctness first (sequential mode)
  k_elementwise<<<1, 1, 0, s1>>>(0, A.a, A.b, A.c, A.d, A.d, A.c, A.b);
  gather_part(s1);
  pb_part(s1);
  CK(hipStreamSynchronize(s1));
  const bool ok = check(A.y, A.ref, "hybrid A x");
  const double t_seq = timeit(0), t_con = timeit(1), t_con2 = timeit(2);
  // the element-wise kernel alone, to subtract
  CK(hipEventRecord(e0, s1));
  for (int rep = 0; rep < reps; ++rep) k_elementwise<<<2048, 256, 0, s1>>>(a.cols, A.a, A.b, A.c, A.d, A.d, A.c, A.b);
  CK(hipEventRecord(e1, s1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double t_el = ms * 1e3 / reps;
  printf("HYBRID gather share %.2f (W %4d panels) + gather-free %.2f | one after the other %6.1f us, concurrent %6.1f us (gather launched first) / %6.1f us (P first)  [element-wise kernel %.1f us subtracted] %s\n",
         share, W, 1.0 - share, t_seq - t_el, t_con - t_el, t_con2 - t_el, t_el, ok ? "ok" : "WRONG");
  fflush(stdout);
  release(db);
  for (void* q : {(void*)d_row0, (void*)d_tp, (void*)d_rp, (void*)d_rb, (void*)d_col, (void*)d_val}) CK(hipFree(q));
  CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
}

int main(int argc, char** argv)
{
  const int m    = argc > 1 ? atoi(argv[1]) : 1000000;
  const int k    = argc > 2 ? atoi(argv[2]) : 10;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const std::string mode = argc > 4 ? argv[4] : "sweep";
  const int which = argc > 5 ? atoi(argv[5]) : 0;

  const std::vector<Config> configs = {
    {8192, 4, 9088, 8, 512, 512, true, false, "base"},
    {8192, 4, 9088, 8, 512, 512, false, false, "register-staged image"},
    {8192, 4, 9088, 8, 512, 512, true, true, "nt stores"},
    {8192, 4, 5504, 8, 512, 512, true, false, "3 R workgroups per CU"},
    {8192, 4, 4096, 8, 256, 512, true, false, "4 R workgroups per CU, 256 threads"},
    {8192, 4, 18304, 8, 1024, 512, true, false, "1024-thread R"},
    {16384, 8, 9088, 8, 512, 1024, true, false, "wide panels"},
    {4096, 2, 9088, 8, 512, 512, true, false, "narrow panels"},
    {2048, 1, 9088, 8, 512, 512, true, false, "one workgroup per panel"},
    {2048, 1, 9088, 4, 512, 512, true, false, "one workgroup per panel, 32-byte pieces"},
    {8192, 4, 9088, 4, 512, 512, true, false, "32-byte pieces"},
    {8192, 4, 9088, 16, 512, 512, true, false, "128-byte pieces"},
    {8192, 8, 9088, 8, 512, 512, true, false, "more P workgroups"},
  };
  const std::vector<FusedConfig> fused = {
    {512, 512, 6, 9088, 8, 512, true, "product geometry"},
    {512, 512, 6, 9088, 4, 512, true, "32-byte pieces"},
    {512, 512, 6, 5504, 4, 512, true, "3 R workgroups per CU"},
    {256, 1024, 6, 9088, 8, 512, true, "256 wide panels"},
    {256, 1024, 6, 9088, 4, 512, true, "256 wide panels, 32-byte pieces"},
    {1024, 512, 6, 9088, 4, 512, true, "1024 panels"},
  };

  if (mode == "selftest") {  // CPU only: the construction reproduces the sequential CSR sums bit for bit
    Csr a = make_matrix(m, m, k, 99), at = transpose(a);
    for (const Csr* mm : {&a, &at})
      for (const Config& c : configs) {
        std::vector<double> x(mm->cols), ref, got;
        std::mt19937_64 rng(5);
        std::normal_distribution<double> nd;
        for (auto& v : x) v = nd(rng);
        cpu_spmv(*mm, x, ref);
        PbHost h = build_pb(*mm, uniform_panels(mm->cols, c.SP), c.cap, c.G, c.Q, max_rows_of(c));
        cpu_pb(h, x, got);
        if (memcmp(ref.data(), got.data(), ref.size() * 8) != 0) { printf("selftest FAILED SP %d cap %d G %d\n", c.SP, c.cap, c.G); return 1; }
        printf("SP %d Q %d cap %d G %d: bins %d, chunk %.1f entries, padding %.3f, P workgroups %d\n", c.SP, c.Q, c.cap, c.G, h.B, h.avg_chunk, h.pad, (int)h.wg_panel.size());
      }
    printf("selftest ok\n");
    return 0;
  }

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; rows = cols = %d, %d nonzeros per row of A, uniformly random columns\n"
         "bytes counted: P = 18/padded nnz + 4/piece + 8/col;  R_A = 8/padded nnz + 2/nnz + 52/row;  R_At = 8/padded nnz + 2/nnz + 36/row\n",
         prop.name, prop.multiProcessorCount, m, k);
  Csr a = make_matrix(m, m, k, 1234), at = transpose(a);
  Side A = make_side(a, 77), At = make_side(at, 78);
  if (mode == "prof") {
    run_config(configs[which], a, at, A, At, reps, true);
    return 0;
  }
  if (mode == "hybrid") {
    run_hybrid(1.0, 512, a, A, reps);
    for (double share : {0.8, 0.7, 0.6, 0.5, 0.4})
      for (int W : {512, 384, 256}) run_hybrid(share, W, a, A, reps);
    return 0;
  }
  if (mode == "proffused") {
    run_fused(fused[which], a, at, A, At, reps, true);
    return 0;
  }
  if (mode.rfind("pr:", 0) == 0) {  // a chosen list of configurations, e.g. pr:1,10,6
    for (size_t at_ = 3; at_ < mode.size();) {
      const size_t comma = mode.find(',', at_);
      run_config(configs[std::stoi(mode.substr(at_, comma - at_))], a, at, A, At, reps, false);
      if (comma == std::string::npos) break;
      at_ = comma + 1;
    }
    return 0;
  }
  if (mode != "fused")
    for (const Config& c : configs) run_config(c, a, at, A, At, reps, false);
  if (mode != "pr")
    for (const FusedConfig& c : fused) run_fused(c, a, at, A, At, reps, false);
  return 0;
}
