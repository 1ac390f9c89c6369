// Stand-alone tuning harness for the CSR stream SpMV (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/spmv_tune.hip -o spmv_tune
//   ./spmv_tune [rows=1000000] [cols=1000000] [nnz_per_row=10] [reps=30]
// Times kernel variants on a random CSR matrix and prints one line per variant with the achieved
// algorithmic bandwidth (12 nnz + 4(r+1) + 8c + 8r bytes per SpMV) so that one GPU call yields the
// whole design table.  Variants:
//   base        the skeleton as first written (scalar strided loads, 2048 nnz / 256 threads)
//   vec<B,N>    aligned window, 4 nonzeros per lane per pass (dwordx4 values x2 + dwordx4 indices),
//               all loads of a workgroup issued before the first LDS write; B threads, N nnz tile
//   *_local     same kernel, gather indices folded into a 1 MiB window (upper bound if the gathered
//               vector were L2 resident)
//   stream      matrix stream only (no gather): what the 12 B/nnz alone costs
//   sub8/sub16  no LDS: 8 / 16 lanes per row, swizzle reduction (comparison point)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                            \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int b, int nb)
{
  const int per = (nb + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// ---------------------------------------------------------------------------------------------------
template <int BLOCK, int NNZB>
__global__ void __launch_bounds__(BLOCK) k_base(int nb, const int* __restrict__ rb, const int* __restrict__ off,
                                                 const int* __restrict__ idx, const double* __restrict__ val,
                                                 const double* __restrict__ x, double* __restrict__ y)
{
  __shared__ double prod[NNZB];
  const int b = xcd_remap(blockIdx.x, nb);
  if (b >= nb) return;
  const int r0 = rb[b], r1 = rb[b + 1];
  const int k0 = off[r0], cnt = off[r1] - k0;
  for (int k = threadIdx.x; k < cnt; k += BLOCK) {
    const double a = __builtin_nontemporal_load(val + k0 + k);
    const int j    = __builtin_nontemporal_load(idx + k0 + k);
    prod[k]        = a * x[j];
  }
  __syncthreads();
  for (int r = r0 + threadIdx.x; r < r1; r += BLOCK) {
    const int s = off[r] - k0, e = off[r + 1] - k0;
    double sum = 0.0;
    for (int k = s; k < e; ++k) sum = sum + prod[k];
    y[r] = sum;
  }
}

template <int MODE>
__device__ __forceinline__ double gload(const double* p)
{
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  if (MODE == 3) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  return *p;
}

// aligned-window vector loads.  The tile starts at k0 rounded DOWN to a multiple of 4 nonzeros so that
// every lane's 4 values / 4 indices are one 32-byte / 16-byte aligned vector load.
template <int BLOCK, int NNZB, bool LOCAL, bool NOGATHER, bool XCD, int GMODE = 0, bool ACCUM = false>
__global__ void __launch_bounds__(BLOCK) k_vec(int nb, const int* __restrict__ rb, const int* __restrict__ off,
                                                const int* __restrict__ idx, const double* __restrict__ val,
                                                const double* __restrict__ x, double* __restrict__ y, int nnz_total)
{
  constexpr int PASSES = (NNZB + 4 + 4 * BLOCK - 1) / (4 * BLOCK);
  __shared__ double prod[PASSES * 4 * BLOCK];
  const int b = XCD ? xcd_remap(blockIdx.x, nb) : blockIdx.x;
  if (b >= nb) return;
  const int r0 = rb[b], r1 = rb[b + 1];
  const int k0 = off[r0], k1 = off[r1];
  const int base = k0 & ~3;
  d4 a[PASSES];
  i4 j[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int k = base + 4 * (p * BLOCK + threadIdx.x);
    if (k < k1) {  // whole vectors may over-read up to 3 entries past k1: the arrays are padded
      a[p] = __builtin_nontemporal_load(reinterpret_cast<const d4*>(val + k));
      j[p] = __builtin_nontemporal_load(reinterpret_cast<const i4*>(idx + k));
    } else {
      a[p] = (d4)(0.0);
      j[p] = (i4)(0);
    }
  }
  d4 g[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    if (NOGATHER) {
      g[p] = (d4){(double)j[p].x, (double)j[p].y, (double)j[p].z, (double)j[p].w};
    } else if (LOCAL) {
      g[p] = (d4){x[j[p].x & 0x1FFFF], x[j[p].y & 0x1FFFF], x[j[p].z & 0x1FFFF], x[j[p].w & 0x1FFFF]};
    } else {
      g[p] = (d4){gload<GMODE>(x + j[p].x), gload<GMODE>(x + j[p].y), gload<GMODE>(x + j[p].z), gload<GMODE>(x + j[p].w)};
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int l = 4 * (p * BLOCK + threadIdx.x);
    *reinterpret_cast<d4*>(&prod[l]) = a[p] * g[p];
  }
  __syncthreads();
  for (int r = r0 + threadIdx.x; r < r1; r += BLOCK) {
    const int s = off[r] - base, e = off[r + 1] - base;
    double sum = ACCUM ? y[r] : 0.0;
    for (int k = s; k < e; ++k) sum = sum + prod[k];
    y[r] = sum;
  }
}

template <int XM>
__device__ __forceinline__ double swz(double v)
{
  constexpr int pattern = (XM << 10) | 0x1F;
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, pattern);
  hi = __builtin_amdgcn_ds_swizzle(hi, pattern);
  return __hiloint2double(hi, lo);
}
// G lanes per row, no LDS
template <int G>
__global__ void __launch_bounds__(256) k_sub(int rows, const int* __restrict__ off, const int* __restrict__ idx,
                                             const double* __restrict__ val, const double* __restrict__ x,
                                             double* __restrict__ y)
{
  const int gid = (blockIdx.x * 256 + threadIdx.x) / G, lane = threadIdx.x % G;
  if (gid >= rows) return;
  const int s = off[gid], e = off[gid + 1];
  double acc = 0.0;
  for (int k = s + lane; k < e; k += G) acc += __builtin_nontemporal_load(val + k) * x[__builtin_nontemporal_load(idx + k)];
  if (G > 1) acc += swz<1>(acc);
  if (G > 2) acc += swz<2>(acc);
  if (G > 4) acc += swz<4>(acc);
  if (G > 8) acc += swz<8>(acc);
  if (lane == 0) y[gid] = acc;
}


// ---------------------------------------------------------------------------------------------------
// slab-major row panels: workgroup w owns a contiguous row panel; its nonzeros are stored slab by slab
// (column ranges of the gathered vector sized for the 4 MiB per-XCD L2); all workgroups walk the slabs
// in the same order at the same pace, so at any time the gathers of an XCD fall into one slab.
template <int T, int CH>
__global__ void __launch_bounds__(T) k_slab(int S, const int* __restrict__ panel_row0, const int* __restrict__ tile_ptr,
                                            const int* __restrict__ rowptr, const long* __restrict__ rp_base,
                                            const int* __restrict__ col, const double* __restrict__ val,
                                            const double* __restrict__ x, double* __restrict__ y)
{
  extern __shared__ double lds[];
  double* prod = lds;
  double* psum = lds + CH;
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
        prod[k - c0]   = a * x[j];
      }
      __syncthreads();
      for (int r = threadIdx.x; r < nr; r += T) {
        int a = rp[r], b = rp[r + 1];
        a = a > c0 ? a : c0;
        b = b < c1 ? b : c1;
        if (a < b) {
          double sum = psum[r];
          for (int k = a; k < b; ++k) sum = sum + prod[k - c0];
          psum[r] = sum;
        }
      }
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < nr; r += T) y[r0 + r] = psum[r];
}

// ---------------------------------------------------------------------------------------------------
static std::vector<int> row_blocks(int rows, const std::vector<int>& off, int nnzb, int max_rows)
{
  std::vector<int> rb{0};
  int start = 0;
  while (start < rows) {
    int end = start;
    long cnt = 0;
    while (end < rows && end - start < max_rows) {
      long len = off[end + 1] - off[end];
      if (cnt + len > nnzb) break;
      cnt += len, ++end;
    }
    if (end == start) end = start + 1;
    rb.push_back(end);
    start = end;
  }
  return rb;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <class F>
  double run(F&& f, int reps)
  {
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / reps;
  }
};

int main(int argc, char** argv)
{
  const int rows = argc > 1 ? atoi(argv[1]) : 1000000, cols = argc > 2 ? atoi(argv[2]) : 1000000;
  const int k = argc > 3 ? atoi(argv[3]) : 10, reps = argc > 4 ? atoi(argv[4]) : 30;
  const long band = argc > 5 ? atol(argv[5]) : 0;  // > 0: columns within +-band of the (scaled) diagonal
  const long nnz = (long)rows * k;
  std::vector<int> off(rows + 1), idx(nnz + 8, 0);
  std::vector<double> val(nnz + 8, 0.0), x(cols), yref(rows);
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> nd;
  for (int i = 0; i <= rows; ++i) off[i] = i * k;
  for (int i = 0; i < rows; ++i) {
    int* p = &idx[(long)i * k];
    for (int t = 0; t < k; ++t) {
      if (band > 0) {
        long c0 = (long)((double)i * cols / rows) + (long)(rng() % (uint64_t)(2 * band + 1)) - band;
        p[t]    = (int)std::min<long>(std::max<long>(c0, 0), cols - 1);
      } else {
        p[t] = (int)(rng() % (uint64_t)cols);
      }
    }
    std::sort(p, p + k);
    for (int t = 0; t < k; ++t) val[(long)i * k + t] = nd(rng);
  }
  for (auto& v : x) v = nd(rng);
  for (int i = 0; i < rows; ++i) {
    double s = 0;
    for (int t = off[i]; t < off[i + 1]; ++t) s = s + val[t] * x[idx[t]];
    yref[i] = s;
  }
  int *d_off, *d_idx;
  double *d_val, *d_x, *d_y;
  CK(hipMalloc(&d_off, (rows + 1) * 4)); CK(hipMalloc(&d_idx, (nnz + 8) * 4)); CK(hipMalloc(&d_val, (nnz + 8) * 8));
  CK(hipMalloc(&d_x, cols * 8)); CK(hipMalloc(&d_y, rows * 8));
  CK(hipMemcpy(d_off, off.data(), (rows + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), (nnz + 8) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), (nnz + 8) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), cols * 8, hipMemcpyHostToDevice));
  const double bytes = 12.0 * nnz + 4.0 * (rows + 1) + 8.0 * cols + 8.0 * rows;
  Timer T;
  Timer& T_ = T;
  std::vector<double> y(rows);
  auto report = [&](const char* name, double us, bool check) {
    long bad = 0;
    if (check) {
      CK(hipMemcpy(y.data(), d_y, rows * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < rows; ++i) bad += (y[i] != yref[i]);
    }
    printf("%-28s %9.2f us  %8.1f GB/s  %5.1f %% of 8 TB/s  %s\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0,
           check ? (bad ? "MISMATCH" : "bit-exact") : "-");
    fflush(stdout);
  };
  printf("rows %d cols %d nnz %ld  algorithmic bytes %.1f MB  (floor %.1f us at 8 TB/s)\n", rows, cols, nnz, bytes / 1e6, bytes / 8e6);

  auto run_tiles = [&](int nnzb, auto launcher, const char* name, bool check) {
    std::vector<int> rb = row_blocks(rows, off, nnzb, 1024);
    int nb = (int)rb.size() - 1;
    int* d_rb;
    CK(hipMalloc(&d_rb, rb.size() * 4));
    CK(hipMemcpy(d_rb, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_y, 0, rows * 8));
    double us = T.run([&] { launcher(nb, d_rb); }, reps);
    report(name, us, check);
    CK(hipFree(d_rb));
  };
  auto grid8 = [](int nb) { return ((nb + 7) / 8) * 8; };

  run_tiles(2048, [&](int nb, int* rb) { k_base<256, 2048><<<grid8(nb), 256>>>(nb, rb, d_off, d_idx, d_val, d_x, d_y); }, "base<256,2048>", true);
#define VEC(B, N, LOC, NOG, XCD, NAME, CHECK) \
  run_tiles(N, [&](int nb, int* rb) { k_vec<B, N, LOC, NOG, XCD><<<grid8(nb), B>>>(nb, rb, d_off, d_idx, d_val, d_x, d_y, (int)nnz); }, NAME, CHECK)
  VEC(256, 1020, false, false, true, "vec<256,1020>", true);
  VEC(256, 2044, false, false, true, "vec<256,2044>", true);
  VEC(256, 4092, false, false, true, "vec<256,4092>", true);
  VEC(512, 2044, false, false, true, "vec<512,2044>", true);
  VEC(512, 4092, false, false, true, "vec<512,4092>", true);
  VEC(128, 1020, false, false, true, "vec<128,1020>", true);
  VEC(1024, 4092, false, false, true, "vec<1024,4092>", true);
  VEC(256, 2044, false, false, false, "vec<256,2044> no xcd remap", true);
  VEC(256, 2044, true, false, true, "vec<256,2044> local gather", false);
  VEC(256, 2044, false, true, true, "vec<256,2044> stream only", false);
  VEC(512, 4092, true, false, true, "vec<512,4092> local gather", false);
  VEC(512, 4092, false, true, true, "vec<512,4092> stream only", false);
  run_tiles(2044, [&](int nb, int* rb) { k_vec<256, 2044, false, false, true, 1><<<grid8(nb), 256>>>(nb, rb, d_off, d_idx, d_val, d_x, d_y, (int)nnz); }, "vec<256,2044> gather nt", true);
  run_tiles(2044, [&](int nb, int* rb) { k_vec<256, 2044, false, false, true, 2><<<grid8(nb), 256>>>(nb, rb, d_off, d_idx, d_val, d_x, d_y, (int)nnz); }, "vec<256,2044> gather sc1(serial)", true);
  run_tiles(2044, [&](int nb, int* rb) { k_vec<256, 2044, true, false, true, 1><<<grid8(nb), 256>>>(nb, rb, d_off, d_idx, d_val, d_x, d_y, (int)nnz); }, "vec<256,2044> local gather nt", false);
  // ---- column slabs: S sub-matrices (columns split into S equal ranges), S launches, y accumulated in order
  for (int S : {2, 4, 8}) {
    std::vector<std::vector<int>> soff(S), sidx(S);
    std::vector<std::vector<double>> sval(S);
    const int W = (cols + S - 1) / S;
    for (int s2 = 0; s2 < S; ++s2) soff[s2].push_back(0);
    for (int i = 0; i < rows; ++i) {
      for (int t = off[i]; t < off[i + 1]; ++t) {
        int s2 = idx[t] / W;
        sidx[s2].push_back(idx[t]), sval[s2].push_back(val[t]);
      }
      for (int s2 = 0; s2 < S; ++s2) soff[s2].push_back((int)sidx[s2].size());
    }
    std::vector<int*> doff(S), didx(S), drb(S);
    std::vector<double*> dval(S);
    std::vector<int> nbs(S);
    for (int s2 = 0; s2 < S; ++s2) {
      sidx[s2].resize(sidx[s2].size() + 8, 0), sval[s2].resize(sval[s2].size() + 8, 0.0);
      std::vector<int> rb = row_blocks(rows, soff[s2], 2044, 1024);
      nbs[s2] = (int)rb.size() - 1;
      CK(hipMalloc(&doff[s2], soff[s2].size() * 4)); CK(hipMalloc(&didx[s2], sidx[s2].size() * 4));
      CK(hipMalloc(&dval[s2], sval[s2].size() * 8)); CK(hipMalloc(&drb[s2], rb.size() * 4));
      CK(hipMemcpy(doff[s2], soff[s2].data(), soff[s2].size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(didx[s2], sidx[s2].data(), sidx[s2].size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dval[s2], sval[s2].data(), sval[s2].size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(drb[s2], rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
    }
    double us = T.run([&] {
      k_vec<256, 2044, false, false, true, 0, false><<<grid8(nbs[0]), 256>>>(nbs[0], drb[0], doff[0], didx[0], dval[0], d_x, d_y, 0);
      for (int s2 = 1; s2 < S; ++s2)
        k_vec<256, 2044, false, false, true, 0, true><<<grid8(nbs[s2]), 256>>>(nbs[s2], drb[s2], doff[s2], didx[s2], dval[s2], d_x, d_y, 0);
    }, reps);
    char nm[64];
    snprintf(nm, sizeof nm, "column slabs x%d (rows<=1024/blk)", S);
    report(nm, us, true);
    for (int s2 = 0; s2 < S; ++s2) { CK(hipFree(doff[s2])); CK(hipFree(didx[s2])); CK(hipFree(dval[s2])); CK(hipFree(drb[s2])); }
  }

  // ---- slab-major row panels -----------------------------------------------------------------------------
  for (int S : {4, 8, 12, 16, 24, 32}) for (int W : {512, 1024, 2048}) {
    const int SW = (cols + S - 1) / S;
    std::vector<int> prow0(W + 1);
    for (int w = 0; w <= W; ++w) {
      long want = nnz * w / W;
      prow0[w] = (int)(std::lower_bound(off.begin(), off.end(), (int)want) - off.begin());
    }
    prow0[W] = rows;
    std::vector<int> tile_ptr((size_t)W * S + 1), pcol(nnz + 8, 0);
    std::vector<double> pval(nnz + 8, 0.0);
    std::vector<long> rp_base((size_t)W * S);
    std::vector<int> rowptr;
    rowptr.reserve((size_t)S * (rows + W));
    long pos = 0;
    int max_rows = 0;
    for (int w = 0; w < W; ++w) {
      const int a = prow0[w], b = prow0[w + 1];
      max_rows = std::max(max_rows, b - a);
      for (int s2 = 0; s2 < S; ++s2) {
        tile_ptr[(size_t)w * S + s2] = (int)pos;
        rp_base[(size_t)w * S + s2]  = (long)rowptr.size();
        for (int i = a; i < b; ++i) {
          rowptr.push_back((int)pos);
          for (int t = off[i]; t < off[i + 1]; ++t)
            if (idx[t] / SW == s2) pcol[pos] = idx[t], pval[pos] = val[t], ++pos;
        }
        rowptr.push_back((int)pos);
      }
    }
    tile_ptr[(size_t)W * S] = (int)pos;
    int *d_p0, *d_tp, *d_rp, *d_pc; long* d_rb; double* d_pv;
    CK(hipMalloc(&d_p0, prow0.size() * 4)); CK(hipMalloc(&d_tp, tile_ptr.size() * 4)); CK(hipMalloc(&d_rp, rowptr.size() * 4));
    CK(hipMalloc(&d_pc, pcol.size() * 4)); CK(hipMalloc(&d_rb, rp_base.size() * 8)); CK(hipMalloc(&d_pv, pval.size() * 8));
    CK(hipMemcpy(d_p0, prow0.data(), prow0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tp, tile_ptr.data(), tile_ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rp, rowptr.data(), rowptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pc, pcol.data(), pcol.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rb, rp_base.data(), rp_base.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pv, pval.data(), pval.size() * 8, hipMemcpyHostToDevice));
    for (int T : {256, 512}) {
      constexpr int CH = 4096;
      const size_t lds = (size_t)(CH + max_rows) * 8;
      if (lds > 160 * 1024) continue;
      CK(hipMemset(d_y, 0, rows * 8));
      double us;
      if (T == 512) {
        CK(hipFuncSetAttribute((const void*)k_slab<512, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        us = T_.run([&] { k_slab<512, CH><<<W, 512, lds>>>(S, d_p0, d_tp, d_rp, d_rb, d_pc, d_pv, d_x, d_y); }, reps);
      } else {
        CK(hipFuncSetAttribute((const void*)k_slab<256, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        us = T_.run([&] { k_slab<256, CH><<<W, 256, lds>>>(S, d_p0, d_tp, d_rp, d_rb, d_pc, d_pv, d_x, d_y); }, reps);
      }
      char nm[96];
      snprintf(nm, sizeof nm, "slab-major S=%d W=%d T=%d lds=%zuK", S, W, T, lds / 1024);
      report(nm, us, true);
    }
    CK(hipFree(d_p0)); CK(hipFree(d_tp)); CK(hipFree(d_rp)); CK(hipFree(d_pc)); CK(hipFree(d_rb)); CK(hipFree(d_pv));
  }
  {
    CK(hipMemset(d_y, 0, rows * 8));
    double us = T.run([&] { k_sub<8><<<(rows * 8 + 255) / 256, 256>>>(rows, d_off, d_idx, d_val, d_x, d_y); }, reps);
    report("sub8 (no LDS)", us, false);
    us = T.run([&] { k_sub<16><<<(rows * 16 + 255) / 256, 256>>>(rows, d_off, d_idx, d_val, d_x, d_y); }, reps);
    report("sub16 (no LDS)", us, false);
    us = T.run([&] { k_sub<1><<<(rows + 255) / 256, 256>>>(rows, d_off, d_idx, d_val, d_x, d_y); }, reps);
    report("thread per row", us, true);
    us = T.run([&] { k_sub<4><<<(rows * 4 + 255) / 256, 256>>>(rows, d_off, d_idx, d_val, d_x, d_y); }, reps);
    report("sub4 (no LDS)", us, false);
  }
  return 0;
}
