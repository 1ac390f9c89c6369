// Design probe for the shared-matrix batch (kernels_batch.hip): the fused dual-side product for K = 8 LPs over one CSR matrix,
// K gather vectors interleaved.  Variants of the ROW WALK only (same epilogue, same reduction): how many rows a lane keeps in flight.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/batch_spmv_probe.hip -o /tmp/batch_spmv_probe && /tmp/batch_spmv_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int K = 8, T = 512, G = T / K;

struct Lp {
  const double *y, *lo, *hi;
  double *yn, *part;
};
struct Lps {
  Lp lp[K];
};

__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }

__device__ __forceinline__ void epilogue(const Lp& L, int i, int l, double s, double sigma, double* yK, double& acc)
{
  const double yi  = L.y[i];
  double next      = yi - (sigma * s);
  const double low = next + sigma * L.lo[i];
  const double up  = next + sigma * L.hi[i];
  next             = dmax(low, dmin(up, 0.0));
  L.yn[i]          = next;
  yK[(size_t)i * K + l] = next;
  const double dy = next - yi;
  acc += dy * dy;
}

__device__ __forceinline__ void reduce_store(double (*accs)[T], const double* acc, const Lps& P, int w, int l, int g)
{
#pragma unroll
  for (int u = 0; u < K; ++u) accs[l][g + G * u] = acc[u];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double total = 0.0;
  for (int vw = 0; vw < 8; ++vw) {
    double v = accs[wave][vw * 64 + lane];
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
    total = vw == 0 ? v : total + v;
  }
  if (lane == 0) P.lp[wave].part[w] = total;
}

// V0: one row after the other (what kernels_batch.hip did first)
__global__ void __launch_bounds__(T) k_v0(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  __shared__ double accs[K][T];
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  const Lp L = P.lp[l];
  double acc[K];
  for (int u = 0; u < K; ++u) acc[u] = 0.0;
  for (int it0 = 0; it0 * G < nr; it0 += K) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int r = g + G * (it0 + u);
      if (r < nr) {
        const int i = r0 + r;
        double s    = 0.0;
        for (int k = off[i]; k < off[i + 1]; ++k) s = s + val[k] * xK[(size_t)idx[k] * K + l];
        epilogue(L, i, l, s, 0.37, yK, acc[u]);
      }
    }
  }
  reduce_store(accs, acc, P, w, l, g);
}

// E1: v0 without the epilogue's streams (sums straight to the interleaved output): what the matrix + the gathers cost alone
__global__ void __launch_bounds__(T) k_v0_noepi(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  for (int r = g; r < nr; r += G) {
    const int i = r0 + r;
    double s    = 0.0;
    for (int k = off[i]; k < off[i + 1]; ++k) s = s + val[k] * xK[(size_t)idx[k] * K + l];
    yK[(size_t)i * K + l] = s;
  }
}

// V1: U sub-blocks of K rows each in flight per lane (U * K independent chains), positions in lockstep, loads unconditional
template <int U>
__global__ void __launch_bounds__(T) k_v1(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  __shared__ double accs[K][T];
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  const Lp L = P.lp[l];
  double acc[K];
#pragma unroll
  for (int u = 0; u < K; ++u) acc[u] = 0.0;
  constexpr int R = U * K;  // rows in flight per lane
  for (int it0 = 0; it0 * G < nr; it0 += R) {
    int k[R], len[R];
    double s[R];
    int maxlen = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int r = g + G * (it0 + q);
      const int i = r0 + (r < nr ? r : 0);
      const int a = off[i], b = off[i + 1];
      k[q]   = a;
      len[q] = r < nr ? b - a : 0;
      s[q]   = 0.0;
      maxlen = len[q] > maxlen ? len[q] : maxlen;
    }
    for (int p = 0; p < maxlen; ++p) {
      int c[R];
      double v[R], x[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int kk = p < len[q] ? k[q] + p : k[q];  // (a safe address when the row is used up; off[m] itself is never read: see host)
        c[q] = idx[kk];
        v[q] = val[kk];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) x[q] = xK[(size_t)c[q] * K + l];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const double t = s[q] + v[q] * x[q];
        s[q]           = p < len[q] ? t : s[q];
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int r = g + G * (it0 + q);
      if (r < nr) epilogue(L, r0 + r, l, s[q], 0.37, yK, acc[q % K]);
    }
  }
  reduce_store(accs, acc, P, w, l, g);
}

// V5: the panel kernel's structure, K wide.  The matrix is read coalesced, a chunk of CH entries at a time, and handed round through
// LDS; the K lanes of a group gather one entry's 64 bytes; the products go to LDS; lane (g, l) adds its rows' products left to right
// from there (row sums in registers).  One barrier per chunk: the next chunk's gathers fly while the previous chunk's row sums run.
template <bool EPI, int CH = 512>
__global__ void __launch_bounds__(T) k_v5(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  constexpr int PER = CH / G;
  __shared__ double prod[2][CH][K];  // (>= 32 KB: the final reduction's accs[K][T] lives here)
  __shared__ int scol[2][CH];
  __shared__ double sval[2][CH];
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K, tid = threadIdx.x;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  const Lp L = P.lp[l];
  double acc[K];
#pragma unroll
  for (int u = 0; u < K; ++u) acc[u] = 0.0;
  for (int b0 = 0; b0 < nr; b0 += T) {
    int k0[K], k1[K];
    double s[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int r = b0 + g + G * u;
      const int i = r0 + (r < nr ? r : 0);
      k0[u] = off[i];
      k1[u] = r < nr ? off[i + 1] : k0[u];
      s[u]  = 0.0;
    }
    const int eb0 = off[r0 + b0], eb1 = off[r0 + (b0 + T < nr ? b0 + T : nr)];
    const int nch = (eb1 - eb0 + CH - 1) / CH;
    if (tid < CH && eb0 + tid < eb1) scol[0][tid] = idx[eb0 + tid], sval[0][tid] = val[eb0 + tid];
    __syncthreads();
    auto rowsum = [&](int cc) {
      const int c0 = eb0 + cc * CH, c1 = c0 + CH < eb1 ? c0 + CH : eb1, pb = cc & 1;
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int a = k0[u] > c0 ? k0[u] : c0, e = k1[u] < c1 ? k1[u] : c1;
        for (int k = a; k < e; ++k) s[u] = s[u] + prod[pb][k - c0][l];
      }
    };
    for (int c = 0; c < nch; ++c) {
      const int c0 = eb0 + c * CH, cnt = eb1 - c0 < CH ? eb1 - c0 : CH, buf = c & 1;
      double pv[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ee = g + G * i < cnt ? g + G * i : 0;
        pv[i]        = xK[(size_t)scol[buf][ee] * K + l];
      }
      const int en = tid < CH && c0 + CH + tid < eb1 ? c0 + CH + tid : c0;
      const int ncol = idx[en];
      const double nval = val[en];
      if (c > 0) rowsum(c - 1);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int ee = g + G * i < cnt ? g + G * i : 0;
        prod[buf][g + G * i][l] = sval[buf][ee] * pv[i];
      }
      if (tid < CH) scol[buf ^ 1][tid] = ncol, sval[buf ^ 1][tid] = nval;
      __syncthreads();
    }
    if (nch > 0) rowsum(nch - 1);
    if (EPI) {
      double yv[K], lov[K], hiv[K];
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int r = b0 + g + G * u;
        const int i = r0 + (r < nr ? r : 0);
        yv[u] = L.y[i], lov[u] = L.lo[i], hiv[u] = L.hi[i];
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int r = b0 + g + G * u;
        if (r < nr) {
          const int i = r0 + r;
          const double sigma = 0.37;
          double next      = yv[u] - (sigma * s[u]);
          const double low = next + sigma * lov[u];
          const double up  = next + sigma * hiv[u];
          next             = dmax(low, dmin(up, 0.0));
          L.yn[i]          = next;
          yK[(size_t)i * K + l] = next;
          const double dy = next - yv[u];
          acc[u] += dy * dy;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int r = b0 + g + G * u;
        if (r < nr) yK[(size_t)(r0 + r) * K + l] = s[u];
      }
    }
  }
  if (EPI) {
    __syncthreads();
    reduce_store((double (*)[T]) & prod[0][0][0], acc, P, w, l, g);
  }
}

// the epilogue's streams alone (no product): MODE 1 all, 2 the three loads only, 3 the two stores only, 4 = 1 in the wave <-> LP, lane <-> row mapping
template <int MODE>
__global__ void __launch_bounds__(T) k_epi_only(const int* row0, Lps P, double* yK)
{
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  const Lp L = MODE == 4 ? P.lp[wave] : P.lp[l];
  double acc = 0.0;
  for (int b0 = 0; b0 < nr; b0 += T) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int r = MODE == 4 ? b0 + lane + 64 * u : b0 + g + G * u;
      if (r < nr) {
        const int i = r0 + r;
        double next = 0.25 * r;
        if (MODE != 3) {
          const double yi = L.y[i];
          next            = dmax(yi + 0.37 * L.lo[i], dmin(yi + 0.37 * L.hi[i], 0.0));
          acc += next;
        }
        if (MODE != 2) {
          L.yn[i] = next;
          if (MODE != 4) yK[(size_t)i * K + l] = next;
        }
      }
    }
  }
  if (MODE == 2 && acc == 1.2345) yK[0] = acc;
}

// V7: every WAVE on its own (no workgroup barrier until the final reduction).  A wave's 8 lane groups own 8 consecutive rows (a "run":
// ~80 consecutive entries); the run's columns and values are read coalesced into a wave-private LDS window, each group then walks
// ITS row left to right: entry address from LDS, one 64-byte gather per group and step, the sum stays in the lane.  NR runs are
// walked side by side (independent chains), the steps unrolled by 2.
template <int NR, bool EPI, int PP = 2>
__global__ void __launch_bounds__(T) k_v7(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  constexpr int CAP = 128;  // entries of a run the window holds (longer runs: straight from global memory)
  __shared__ int scol[T / 64][NR][CAP];
  __shared__ double sval[T / 64][NR][CAP];
  __shared__ double accs[K][T];
  const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane % K, gg = lane / K, g = tid / K;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  const Lp L = P.lp[l];
  double acc[K];
#pragma unroll
  for (int u = 0; u < K; ++u) acc[u] = 0.0;
  const int nruns = (nr + T - 1) / T * K;  // runs of this wave: (sub-block b, u) -> rows b * 512 + 8 * wave + 64 * u + (0..7)
  for (int q0 = 0; q0 < nruns; q0 += NR) {
    int k0[NR], len[NR], e0[NR], i_row[NR];
    bool valid[NR], windowed[NR];
    double s[NR], yv[NR], lov[NR], hiv[NR];
    int maxlen = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int q = q0 + j, b = q / K, u = q % K;
      const int R = b * T + 8 * wave + G * u;  // first row of the run (panel-relative)
      const int r = R + gg;
      valid[j]    = q < nruns && r < nr;
      i_row[j]    = r0 + (valid[j] ? r : 0);
      const int a = off[i_row[j]], e = off[i_row[j] + 1];
      k0[j]  = a;
      len[j] = valid[j] ? e - a : 0;
      s[j]   = 0.0;
      if (EPI) yv[j] = L.y[i_row[j]], lov[j] = L.lo[i_row[j]], hiv[j] = L.hi[i_row[j]];
      // the run's entries [e0, e1): from its first row's start to its last valid row's end (wave-uniform)
      const int Rl   = R < nr ? R : 0;
      const int Rend = R + 8 < nr ? R + 8 : nr;
      e0[j]          = q < nruns && R < nr ? off[r0 + Rl] : 0;
      const int e1   = q < nruns && R < nr ? off[r0 + Rend] : 0;
      windowed[j]    = e1 - e0[j] <= CAP;
      if (windowed[j]) {
        for (int t = lane; t < e1 - e0[j]; t += 64) scol[wave][j][t] = idx[e0[j] + t], sval[wave][j][t] = val[e0[j] + t];
      }
      maxlen = len[j] > maxlen ? len[j] : maxlen;
    }
    // longest row over the wave
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_xor(maxlen, d, 64);
      maxlen      = o > maxlen ? o : maxlen;
    }
    __builtin_amdgcn_wave_barrier();
    for (int p = 0; p < maxlen; p += PP) {
      int c[NR][PP];
      double v[NR][PP], x[NR][PP];
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int d = 0; d < PP; ++d) {
          const bool act = p + d < len[j];
          if (windowed[j]) {
            const int a = act ? k0[j] - e0[j] + p + d : 0;
            c[j][d] = scol[wave][j][a], v[j][d] = sval[wave][j][a];
          } else {
            const int a = act ? k0[j] + p + d : k0[j];
            c[j][d] = idx[a], v[j][d] = val[a];
          }
        }
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int d = 0; d < PP; ++d) x[j][d] = xK[(size_t)c[j][d] * K + l];
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int d = 0; d < PP; ++d) {
          const double t = s[j] + v[j][d] * x[j][d];
          s[j]           = p + d < len[j] ? t : s[j];
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int u = (q0 + j) % K;
      if (EPI) {
        const double sigma = 0.37;
        double next      = yv[j] - (sigma * s[j]);
        const double low = next + sigma * lov[j];
        const double up  = next + sigma * hiv[j];
        next             = dmax(low, dmin(up, 0.0));
        const double dy  = next - yv[j];
        const double add = dy * dy;
        if (valid[j]) {
          L.yn[i_row[j]]                 = next;
          yK[(size_t)i_row[j] * K + l] = next;
        }
#pragma unroll
        for (int uu = 0; uu < K; ++uu) acc[uu] = (valid[j] && uu == u) ? acc[uu] + add : acc[uu];
      } else if (valid[j]) {
        yK[(size_t)i_row[j] * K + l] = s[j];
      }
    }
  }
  if (EPI) reduce_store(accs, acc, P, w, l, g);
}

template <int U>
__global__ void __launch_bounds__(T) k_v1_noepi(const int* row0, const int* off, const int* idx, const double* val, Lps P, const double* xK, double* yK)
{
  const int w = blockIdx.x, l = threadIdx.x % K, g = threadIdx.x / K;
  const int r0 = row0[w], nr = row0[w + 1] - r0;
  constexpr int R = U * K;
  for (int it0 = 0; it0 * G < nr; it0 += R) {
    int k[R], len[R];
    double s[R];
    int maxlen = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int r = g + G * (it0 + q);
      const int i = r0 + (r < nr ? r : 0);
      const int a = off[i], b = off[i + 1];
      k[q]   = a;
      len[q] = r < nr ? b - a : 0;
      s[q]   = 0.0;
      maxlen = len[q] > maxlen ? len[q] : maxlen;
    }
    for (int p = 0; p < maxlen; ++p) {
      int c[R];
      double v[R], x[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int kk = p < len[q] ? k[q] + p : k[q];
        c[q] = idx[kk];
        v[q] = val[kk];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) x[q] = xK[(size_t)c[q] * K + l];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const double t = s[q] + v[q] * x[q];
        s[q]           = p < len[q] ? t : s[q];
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int r = g + G * (it0 + q);
      if (r < nr) yK[(size_t)(r0 + r) * K + l] = s[q];
    }
  }
}

int main(int argc, char** argv)
{
  const int m = 1000000, n = 1000000, W = argc > 1 ? atoi(argv[1]) : 510;
  std::mt19937_64 rng(5);
  std::vector<int> off(m + 1, 0), idx;
  std::vector<double> val;
  idx.reserve((size_t)m * 11), val.reserve((size_t)m * 11);
  std::poisson_distribution<int> plen(10.0);
  for (int i = 0; i < m; ++i) {
    int len = std::min(std::max(plen(rng), 1), 40);
    std::vector<int> cols(len);
    for (int& c : cols) c = (int)(rng() % n);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (int c : cols) idx.push_back(c), val.push_back((double)(rng() % 2000) / 1000.0 - 1.0);
    off[i + 1] = (int)idx.size();
  }
  const size_t nnz = idx.size();
  idx.push_back(0), val.push_back(0.0);  // (one spare entry: the clamped address of an empty last row)
  std::vector<int> row0(W + 1);
  for (int w = 0; w <= W; ++w) row0[w] = (int)((int64_t)m * w / W);
  printf("m %d n %d nnz %zu W %d\n", m, n, nnz, W);
  int *d_row0, *d_off, *d_idx;
  double *d_val, *d_xK, *d_yK;
  CK(hipMalloc(&d_row0, (W + 1) * 4)); CK(hipMalloc(&d_off, (m + 1) * 4)); CK(hipMalloc(&d_idx, (nnz + 1) * 4)); CK(hipMalloc(&d_val, (nnz + 1) * 8));
  CK(hipMalloc(&d_xK, (size_t)n * K * 8)); CK(hipMalloc(&d_yK, (size_t)m * K * 8));
  CK(hipMemcpy(d_row0, row0.data(), (W + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_off, off.data(), (m + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), (nnz + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), (nnz + 1) * 8, hipMemcpyHostToDevice));
  std::vector<double> xK((size_t)n * K);
  for (double& v : xK) v = (double)(rng() % 4001) / 2000.0 - 1.0;
  CK(hipMemcpy(d_xK, xK.data(), xK.size() * 8, hipMemcpyHostToDevice));
  Lps P;
  std::vector<double> tmp(m);
  for (int l = 0; l < K; ++l) {
    double *y, *lo, *hi, *yn, *part;
    CK(hipMalloc(&y, m * 8)); CK(hipMalloc(&lo, m * 8)); CK(hipMalloc(&hi, m * 8)); CK(hipMalloc(&yn, m * 8)); CK(hipMalloc(&part, W * 8));
    for (double& v : tmp) v = (double)(rng() % 2001) / 1000.0 - 1.0;
    CK(hipMemcpy(y, tmp.data(), m * 8, hipMemcpyHostToDevice));
    for (double& v : tmp) v = -0.5;
    CK(hipMemcpy(lo, tmp.data(), m * 8, hipMemcpyHostToDevice));
    for (double& v : tmp) v = 0.75;
    CK(hipMemcpy(hi, tmp.data(), m * 8, hipMemcpyHostToDevice));
    P.lp[l] = Lp{y, lo, hi, yn, part};
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> ref_y((size_t)m * K), got_y((size_t)m * K), ref_p(W), got_p(W);
  auto run = [&](const char* name, auto launch, bool is_ref) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got_y.data(), d_yK, got_y.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(got_p.data(), P.lp[3].part, W * 8, hipMemcpyDeviceToHost));
    if (is_ref) ref_y = got_y, ref_p = got_p;
    const bool same = ref_y == got_y && ref_p == got_p;
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.1f us   %s\n", name, 1e3 * ms / reps, same ? "bit-identical" : "DIFFERS");
    fflush(stdout);
  };
  run("v0 row after row", [&] { k_v0<<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, true);
  run("v0 without epilogue", [&] { k_v0_noepi<<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  {  // the gathers confined to a window of the interleaved vector: which level of the hierarchy serves them at what rate
    std::vector<int> folded(nnz + 1);
    int* d_fold;
    CK(hipMalloc(&d_fold, (nnz + 1) * 4));
    for (int window : {250000, 62500, 2000}) {
      for (size_t k = 0; k <= nnz; ++k) folded[k] = idx[k] % window;
      CK(hipMemcpy(d_fold, folded.data(), (nnz + 1) * 4, hipMemcpyHostToDevice));
      char name[96];
      snprintf(name, sizeof name, "v0, columns mod %d (%.1f MB)", window, window * 64e-6);
      run(name, [&] { k_v0<<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      snprintf(name, sizeof name, "  ... without epilogue");
      run(name, [&] { k_v0_noepi<<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 1 run, 16 gathers in flight", [&] { k_v7<1, true, 16><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 1 run, 16 in flight, no epilogue", [&] { k_v7<1, false, 16><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 2 runs, 8 in flight, no epilogue", [&] { k_v7<2, false, 8><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 2 runs", [&] { k_v7<2, true><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 2 runs without epilogue", [&] { k_v7<2, false><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v7 4 runs without epilogue", [&] { k_v7<4, false><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
      run("  v5 without epilogue", [&] { k_v5<false><<<W, T>>>(d_row0, d_off, d_fold, d_val, P, d_xK, d_yK); }, false);
    }
  }
  run("v5 staged through LDS", [&] { k_v5<true><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v5 without epilogue", [&] { k_v5<false><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v5, chunks of 256", [&] { k_v5<true, 256><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v5, chunks of 256, no epilogue", [&] { k_v5<false, 256><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  {  // feasibility of a column-window sweep: every panel's entries re-ordered window-major (the row sums are then meaningless: timing only)
    std::vector<int> widx(nnz + 1);
    std::vector<double> wval(nnz + 1);
    int* d_widx;
    double* d_wval;
    CK(hipMalloc(&d_widx, (nnz + 1) * 4)); CK(hipMalloc(&d_wval, (nnz + 1) * 8));
    for (int windows : {6, 16, 32, -16, -32}) {  // (negative: window-major inside every block of 512 rows instead of inside the panel)
      const bool per_block = windows < 0;
      windows = windows < 0 ? -windows : windows;
      const int wc = (n + windows - 1) / windows;
      std::vector<int> order;
      std::vector<int> bounds;
      for (int w = 0; w < W; ++w)
        for (int r = row0[w]; r < row0[w + 1]; r += per_block ? 512 : (1 << 30)) bounds.push_back(r);
      bounds.push_back(m);
      std::vector<int> starts;
      for (int w = 0; w < W; ++w) starts.push_back(row0[w]);
      for (size_t q = 0; q + 1 < bounds.size(); ++q) {
        int rb = bounds[q];
        int re = bounds[q + 1];
        if (!per_block) re = *std::upper_bound(row0.begin(), row0.end(), rb);
        else re = std::min(rb + 512, *std::upper_bound(row0.begin(), row0.end(), rb));
        const int a = off[rb], b = off[re];
        order.resize(b - a);
        for (int k = a; k < b; ++k) order[k - a] = k;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return idx[x] / wc < idx[y] / wc; });
        for (int k = a; k < b; ++k) widx[k] = idx[order[k - a]], wval[k] = val[order[k - a]];
      }
      CK(hipMemcpy(d_widx, widx.data(), (nnz + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wval, wval.data(), (nnz + 1) * 8, hipMemcpyHostToDevice));
      char name[96];
      snprintf(name, sizeof name, "v5 no epi, %d windows (%.1f MB)%s", windows, wc * 64e-6, per_block ? " per 512-row block" : "");
      run(name, [&] { k_v5<false><<<W, T>>>(d_row0, d_off, d_widx, d_wval, P, d_xK, d_yK); }, false);
      snprintf(name, sizeof name, "v5 with epi, %d windows", windows);
      run(name, [&] { k_v5<true><<<W, T>>>(d_row0, d_off, d_widx, d_wval, P, d_xK, d_yK); }, false);
    }
  }
  run("epilogue streams alone", [&] { k_epi_only<1><<<W, T>>>(d_row0, P, d_yK); }, false);
  run("  loads only", [&] { k_epi_only<2><<<W, T>>>(d_row0, P, d_yK); }, false);
  run("  stores only", [&] { k_epi_only<3><<<W, T>>>(d_row0, P, d_yK); }, false);
  run("  wave <-> LP mapping (no yK)", [&] { k_epi_only<4><<<W, T>>>(d_row0, P, d_yK); }, false);
  run("v7 waves on their own, 1 run", [&] { k_v7<1, true><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 1 run, 16 gathers in flight", [&] { k_v7<1, true, 16><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 1 run, 8 gathers in flight", [&] { k_v7<1, true, 8><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 2 runs, 8 gathers in flight", [&] { k_v7<2, true, 8><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 1 run, 16 in flight, no epilogue", [&] { k_v7<1, false, 16><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 2 runs side by side", [&] { k_v7<2, true><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 4 runs side by side", [&] { k_v7<4, true><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 2 runs, without epilogue", [&] { k_v7<2, false><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v7 4 runs, without epilogue", [&] { k_v7<4, false><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v1 8 rows in flight, no epi", [&] { k_v1_noepi<1><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v1 16 rows in flight, no epi", [&] { k_v1_noepi<2><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v1 8 rows in flight", [&] { k_v1<1><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  run("v1 16 rows in flight", [&] { k_v1<2><<<W, T>>>(d_row0, d_off, d_idx, d_val, P, d_xK, d_yK); }, false);
  return 0;
}
