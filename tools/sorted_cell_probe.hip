// Round-6 probe, second design of the class "no far gather, no product round trip" (the first, tools/tall_panel_probe.hip, staged
// slabs of the vector in LDS and was bound by the LDS pipeline: profiles/r06_tall_probe.txt).
//
//   * a workgroup (1024 threads, one per CU) owns a TALL row panel (R <= 16 383 rows, fp64 accumulators in LDS) times a COLUMN
//     RANGE (the panel's entries cut into S ranges of EQUAL COUNT: ~n / S columns each); NP * S workgroups; the workgroups of one
//     XCD take the same range number, so the range (~2 MB at C3) sits in that XCD's L2;
//   * the cell's entries are stored sorted by COLUMN, thread t of a step taking entry t: a wave's 64 gathers of the vector fall on
//     ~64 * (n / S) / (R * nnz_per_row / S) consecutive columns -- ~26 lines of 128 bytes instead of 64 -- and the workgroup as a
//     whole reads every line of its range about once, front to back.  No slab staging: nothing but the accumulators is in LDS;
//   * a step is 1024 * E entries with pairwise DIFFERENT rows (the builder defers an entry whose row is already in the step to the
//     next one), steps are separated by one barrier: every accumulator gets at most one addition per step, by one lane, so the
//     additions of a row happen in step order = column order, deterministically, with plain ds_read / add / ds_write;
//   * entries are requested PE steps ahead, the gathers PG steps ahead (the barrier leaves vector-memory loads in flight);
//   * S partial vectors, combined in a fixed order by a streaming kernel that carries the epilogue (rtol 1e-12 per row).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sorted_cell_probe.hip -o /tmp/sorted_cell_probe
//   /tmp/sorted_cell_probe <dir from scripts/dump_csr.py> [NP S E PE PG atomic dbg [reps]]     (no configuration: the sweep)
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

constexpr int kRowBits = 14, kT = 1024;
constexpr int kAcc = 1 << kRowBits;  // accumulators in LDS (the pads add 0.0 to slot R..)

struct CellView {
  int rows, cols, NP, S, R, CWID, G, rows_pad;
  const double* val;      // every workgroup's stream: steps of kT * E entries, entry e of thread t of step s at ((s * E + e) * kT + t)
  const uint32_t* pk;     // row in panel | column in range << 14
  const long long* gbase; // [G]
  const int* gsteps;      // [G] steps of the workgroup
  const int* sbase;       // [G * NSMAX] first column of a step: the column field is relative to it (a step spans a few thousand columns)
  int NSMAX;              // table stride (>= the longest workgroup's steps + 2 * PE)
  long long pad_at;       // one step of pads (0.0 into accumulator R), read by every request beyond a workgroup's last step
};

__device__ __forceinline__ bool cell_where(const CellView& V, int b, int* p, int* q)
{
  const int xcd = b & 7, j = b >> 3;
  if (V.S <= 8) {
    const int per = 8 / V.S;
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
  __builtin_amdgcn_sched_barrier(0);  // (nothing moves across a step: the scheduler otherwise hoists the products of the whole ring to the loop's head, behind a vmcnt(0))
}

// DBG: 1 no gathers, 2 no additions, 4 no entry loads (timing experiments, results wrong by construction)
template <int E, int PE, int PG, bool ATOMIC, int DBG>
__global__ __launch_bounds__(kT) void k_cell(CellView V, const double* __restrict__ x, double* __restrict__ partial)
{
  static_assert(PG < PE, "the gather of step s + PG reads the entry words of a slot that is rewritten at step s + PE");
  extern __shared__ double acc[];
  int p, q;
  if (!cell_where(V, blockIdx.x, &p, &q)) return;
  const int g  = blockIdx.x;
  const int t  = threadIdx.x;
  const int r0 = p * V.R;
  const int nr = min(V.R, V.rows - r0);
  for (int i = t; i < kAcc; i += kT) acc[i] = 0.0;
  const double* __restrict__ val  = V.val + t;
  const uint32_t* __restrict__ pk = V.pk + t;
  const int* __restrict__ sb      = V.sbase + (size_t)g * V.NSMAX;
  const long long wbase = V.gbase[g];
  const int ns = V.gsteps[g];
  // entries of step s: the workgroup's own stream, or the shared step of pads beyond its end
  auto at = [&](int s, int e) -> size_t { return (size_t)(s < ns ? wbase + (long long)(s * E + e) * kT : V.pad_at + (long long)e * kT); };
  double rv[PE][E], xg[PG][E];
  uint32_t rp[PE][E];
  // (the prologue issues its requests in the loop's own order -- gathers of step k + PG, entries of step k + PE for k = -PE .. -1 --
  //  so that the counted vmcnt waits of the loop's first step are the steady state's, not a drain)
#pragma unroll
  for (int k = -PE; k < 0; ++k) {
    if (k + PG >= 0) {
      const double* xs = x + sb[k + PG];
#pragma unroll
      for (int e = 0; e < E; ++e) xg[(k + PG) % PG][e] = (DBG & 1) ? (double)(rp[k + PG][e] >> kRowBits) : xs[rp[k + PG][e] >> kRowBits];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      rv[k + PE][e] = __builtin_nontemporal_load(val + at(k + PE, e));
      rp[k + PE][e] = __builtin_nontemporal_load(pk + at(k + PE, e));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  step_barrier();  // the zeroed accumulators
  // (ns >= 1: a do-while -- behind a loop guard the compiler sinks the prologue's gathers into the pre-header, after the entries.  The
  //  trip count is ns rounded up to PE: steps beyond ns read the shared pads.  An exit from the middle of the ring was tried: the
  //  structurizer's flow blocks bring vmcnt(0) back.)
  int s0 = 0;
  do {
#pragma unroll
    for (int d = 0; d < PE; ++d) {
      // step s0 + d: additions of its entries
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int row   = (int)(rp[d][e] & (kAcc - 1));
        const double pr = rv[d][e] * xg[d % PG][e];
        if (DBG & 2) { if (pr == 1.2345) acc[row] = pr; }
        else if (ATOMIC) lds_add(acc + row, pr);
        else acc[row] = acc[row] + pr;
      }
      // the gathers of step s0 + d + PG, the entries of step s0 + d + PE
      {
        const double* xs = x + sb[s0 + d + PG];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint32_t w = rp[(d + PG) % PE][e];
          xg[d % PG][e]    = (DBG & 1) ? (double)(w >> kRowBits) : xs[w >> kRowBits];
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (DBG & 4) {}  // (the ring keeps its first PE steps: rows of a step stay pairwise different)
        else {
          const size_t i = at(s0 + d + PE, e);
          rv[d][e] = __builtin_nontemporal_load(val + i);
          rp[d][e] = __builtin_nontemporal_load(pk + i);
        }
      }
      step_barrier();
    }
    s0 += PE;
  } while (s0 < ns);
  double* out = partial + (size_t)q * V.rows_pad + r0;
  for (int i = t; i < nr; i += kT) out[i] = acc[i];
}

// combine the S partial vectors in a fixed order + a stand-in for the fused dual-side epilogue's traffic (48 bytes per row)
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
    double u = y - sigma * v;
    u = u < lo ? lo : (u > hi ? hi : u);
    e3[i] = e3[i] + sigma * y;
    v = u;
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

struct CellHost {
  CellView V{};
  std::vector<double> val;
  std::vector<uint32_t> pk;
  std::vector<long long> gbase;
  std::vector<int> gsteps, sbase;
  long long pads = 0, entries = 0, deferred = 0;
  int max_steps = 0, min_steps = 1 << 30, max_span = 0;
  double build_s = 0, lines_per_wave = 0;
};

static bool build_cells(int rows, int cols, const std::vector<int>& off, const std::vector<int>& idx, const std::vector<double>& a, int NP, int S, int E, int PE, CellHost* H)
{
  const auto t0 = std::chrono::steady_clock::now();
  CellView& V = H->V;
  V.rows = rows, V.cols = cols, V.NP = NP, V.S = S;
  V.R    = (rows + NP - 1) / NP;
  V.CWID = (cols + S - 1) / S;
  if (V.R >= kAcc) return false;
  V.G        = S <= 8 ? ((NP + 8 / S - 1) / (8 / S)) * 8 : NP * S;
  V.rows_pad = (rows + 63) & ~63;
  H->gbase.assign(V.G, 0), H->gsteps.assign(V.G, 1);
  H->val.clear(), H->pk.clear();
  H->val.reserve(idx.size() + idx.size() / 8), H->pk.reserve(idx.size() + idx.size() / 8);
  const int C = kT * E;
  struct Ent { uint32_t col, row; double v; };
  std::vector<Ent> panel;
  std::vector<std::vector<Ent>> steps;
  std::vector<std::vector<int>> wg_sbase(V.G);
  std::vector<int> last(V.R + 1);
  const uint32_t pad_pk = (uint32_t)V.R;  // row R: an accumulator nobody reads; the step's first column
  long long line_touches = 0, wave_instr = 0;
  for (int p = 0; p < NP; ++p) {
    // the panel's entries by column, cut into S cells of equal count
    panel.clear();
    const int r0 = p * V.R, r1 = std::min(rows, r0 + V.R);
    for (int r = r0; r < r1; ++r)
      for (int k = off[r]; k < off[r + 1]; ++k) panel.push_back({(uint32_t)idx[k], (uint32_t)(r - r0), a[(size_t)k]});
    std::stable_sort(panel.begin(), panel.end(), [](const Ent& u, const Ent& w) { return u.col < w.col; });
    H->entries += (long long)panel.size();
    for (int q = 0; q < S; ++q) {
      const int b = S <= 8 ? (p / (8 / S)) * 8 + (p % (8 / S)) * S + q : p * S + q;
      const size_t e0 = panel.size() * (size_t)q / S, e1 = panel.size() * (size_t)(q + 1) / S;
      // steps of C entries with pairwise different rows; an entry goes to the first step behind its row's previous entry that has room
      steps.clear();
      std::fill(last.begin(), last.end(), -1);
      int cur = 0;
      for (size_t k = e0; k < e1; ++k) {
        const Ent& en = panel[k];
        int st = std::max(cur, last[en.row] + 1);
        for (;; ++st) {
          if ((int)steps.size() <= st) steps.resize(st + 1);
          if ((int)steps[st].size() < C) break;
        }
        if (st != cur) ++H->deferred;
        steps[st].push_back(en);
        last[en.row] = st;
        while (cur < (int)steps.size() && (int)steps[cur].size() == C) ++cur;
      }
      if (steps.empty()) steps.resize(1);
      const int ns = (int)steps.size();
      H->gbase[b]  = (long long)H->val.size();
      H->gsteps[b] = ns;
      H->max_steps = std::max(H->max_steps, ns);
      H->min_steps = std::min(H->min_steps, ns);
      for (auto& st : steps) {
        std::stable_sort(st.begin(), st.end(), [](const Ent& u, const Ent& w) { return u.col < w.col; });
        const uint32_t base = st.empty() ? 0u : st.front().col;
        const uint32_t span = st.empty() ? 0u : st.back().col - base;
        H->max_span = std::max(H->max_span, (int)span);
        if (span >= (1u << (32 - kRowBits))) return false;
        wg_sbase[b].push_back((int)base);
        const size_t real = st.size();
        H->pads += (long long)(C - (int)real);
        for (int k = 0; k < C; ++k) {
          if ((size_t)k < real) { H->val.push_back(st[k].v), H->pk.push_back(st[k].row | (st[k].col - base) << kRowBits); }
          else { H->val.push_back(0.0), H->pk.push_back(pad_pk | span << kRowBits); }
        }
        for (size_t k = 0; k < real; k += 64) {
          ++wave_instr;
          uint32_t prev = ~0u;
          for (size_t u = k; u < std::min(real, k + 64); ++u) { if ((st[u].col >> 4) != prev) ++line_touches; prev = st[u].col >> 4; }
        }
      }
    }
  }
  V.pad_at = (long long)H->val.size();
  for (int i = 0; i < C + 4096; ++i) H->val.push_back(0.0), H->pk.push_back(pad_pk);
  V.NSMAX = H->max_steps + 2 * PE + 2;
  H->sbase.assign((size_t)V.G * V.NSMAX, 0);
  for (int b = 0; b < V.G; ++b)
    for (size_t k = 0; k < wg_sbase[b].size(); ++k) H->sbase[(size_t)b * V.NSMAX + k] = wg_sbase[b][k];
  H->lines_per_wave = wave_instr ? (double)line_touches / (double)wave_instr : 0.0;
  H->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return true;
}

struct CellDev {
  CellView V{};
  double* val = nullptr;
  uint32_t* pk = nullptr;
  long long* gbase = nullptr;
  int* gsteps = nullptr;
  int* sbase = nullptr;
  double* partial = nullptr;
};
static void upload(const CellHost& H, CellDev* Dv)
{
  Dv->V = H.V;
  OK(hipMalloc((void**)&Dv->val, H.val.size() * 8)); OK(hipMalloc((void**)&Dv->pk, H.pk.size() * 4));
  OK(hipMalloc((void**)&Dv->gbase, H.gbase.size() * 8)); OK(hipMalloc((void**)&Dv->gsteps, H.gsteps.size() * 4));
  OK(hipMalloc((void**)&Dv->partial, (size_t)H.V.S * H.V.rows_pad * 8));
  OK(hipMemcpy(Dv->val, H.val.data(), H.val.size() * 8, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->pk, H.pk.data(), H.pk.size() * 4, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->gbase, H.gbase.data(), H.gbase.size() * 8, hipMemcpyHostToDevice));
  OK(hipMemcpy(Dv->gsteps, H.gsteps.data(), H.gsteps.size() * 4, hipMemcpyHostToDevice));
  OK(hipMalloc((void**)&Dv->sbase, H.sbase.size() * 4));
  OK(hipMemcpy(Dv->sbase, H.sbase.data(), H.sbase.size() * 4, hipMemcpyHostToDevice));
  Dv->V.val = Dv->val, Dv->V.pk = Dv->pk, Dv->V.gbase = Dv->gbase, Dv->V.gsteps = Dv->gsteps, Dv->V.sbase = Dv->sbase;
}
static void release(CellDev* Dv) { (void)hipFree(Dv->val); (void)hipFree(Dv->pk); (void)hipFree(Dv->gbase); (void)hipFree(Dv->gsteps); (void)hipFree(Dv->sbase); (void)hipFree(Dv->partial); }

typedef void (*cell_fn)(CellView, const double*, double*);
struct Variant { int E, PE, PG, atomic, dbg; cell_fn fn; };
#define VARIANT(E, PE, PG, AT, DBG) Variant{E, PE, PG, AT, DBG, k_cell<E, PE, PG, (AT) != 0, DBG>}
static const Variant kVariants[] = {
  VARIANT(1, 4, 2, 0, 0), VARIANT(1, 4, 2, 1, 0), VARIANT(1, 4, 3, 0, 0), VARIANT(1, 8, 4, 0, 0), VARIANT(1, 8, 6, 0, 0), VARIANT(1, 6, 3, 0, 0), VARIANT(1, 3, 2, 0, 0), VARIANT(1, 2, 1, 0, 0),
  VARIANT(2, 4, 2, 0, 0), VARIANT(2, 2, 1, 0, 0), VARIANT(2, 3, 2, 0, 0),
  // timing experiments
  VARIANT(1, 4, 2, 0, 1), VARIANT(1, 4, 2, 0, 2), VARIANT(1, 4, 2, 0, 3), VARIANT(1, 4, 2, 0, 4), VARIANT(1, 4, 2, 0, 6), VARIANT(1, 4, 2, 0, 7), VARIANT(1, 4, 2, 0, 5),
};

int main(int argc, char** argv)
{
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc < 2) { std::printf("usage: %s <dir> [NP S E PE PG atomic dbg [reps]]\n", argv[0]); return 2; }
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
  if (std::getenv("CELL_BUILD_ONLY")) {  // the builder's statistics, no GPU needed
    for (int t = 0; t < 2; ++t) {
      CellHost H;
      const bool ok = build_cells(rows[t], cols[t], off[t], idx[t], val[t], 64, 4, 1, 8, &H);
      std::printf("%s: ok %d entries %lld pads %lld deferred %lld steps %d-%d max span %d lines/wave %.1f build %.1f s\n", names[t], (int)ok, H.entries, H.pads, H.deferred, H.min_steps, H.max_steps, H.max_span,
                  H.lines_per_wave, H.build_s);
    }
    return 0;
  }
  const int big = std::max(m, n);
  const size_t vec_pad = (size_t)big + (size_t)big / 2 + 64 * 4096;
  std::vector<double> hx(vec_pad, 0.0);
  for (int i = 0; i < big; ++i) hx[i] = 1.0 + ((i * 2654435761u) % 1000) * 1e-3 - (i % 3) * 0.7;
  double *x[2], *yout[2], *e[4], *tv[8];
  for (int t = 0; t < 2; ++t) { OK(hipMalloc((void**)&x[t], vec_pad * 8)); OK(hipMemcpy(x[t], hx.data(), vec_pad * 8, hipMemcpyHostToDevice)); OK(hipMalloc((void**)&yout[t], vec_pad * 8)); }
  for (auto& p : e) { OK(hipMalloc((void**)&p, vec_pad * 8)); OK(hipMemcpy(p, hx.data(), vec_pad * 8, hipMemcpyHostToDevice)); }
  for (auto& p : tv) { OK(hipMalloc((void**)&p, vec_pad * 8)); OK(hipMemset(p, 0, vec_pad * 8)); }
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
  std::printf("tall row panels with LDS accumulators x column ranges, entries sorted by column, the vector gathered from L2 (no LDS slabs): %d x %d, %zu nonzeros\n", m, n, idx[0].size());
  std::printf("%4s %2s %2s %2s %2s %2s %3s | %6s %8s %8s %6s %6s | %9s %9s %9s %9s | %9s | %s\n", "NP", "S", "E", "PE", "PG", "at", "dbg", "R", "pads", "deferred", "steps", "ln/wv", "A us", "comb us", "AT us",
              "comb us", "pair us", "max rel err (A, AT)");
  struct Cfg { int NP, S, E, PE, PG, AT, DBG; };
  std::vector<Cfg> cfgs;
  int reps = 20;
  if (argc >= 9) {
    cfgs.push_back({std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8])});
    if (argc >= 10) reps = std::atoi(argv[9]);
  } else {
    for (const auto& v : kVariants) cfgs.push_back({64, 4, v.E, v.PE, v.PG, v.atomic, v.dbg});
    cfgs.push_back({62, 4, 1, 4, 2, 0, 0});   // R = 16 130: 248 workgroups
  }
  for (const Cfg& c : cfgs) {
    const Variant* var = nullptr;
    for (const auto& v : kVariants)
      if (v.E == c.E && v.PE == c.PE && v.PG == c.PG && v.atomic == c.AT && v.dbg == c.DBG) var = &v;
    if (!var) { std::printf("no kernel instance for E %d PE %d PG %d atomic %d dbg %d\n", c.E, c.PE, c.PG, c.AT, c.DBG); continue; }
    CellHost H[2];
    CellDev Dv[2];
    bool ok = true;
    for (int t = 0; t < 2 && ok; ++t) {
      ok = build_cells(rows[t], cols[t], off[t], idx[t], val[t], c.NP, c.S, c.E, c.PE, &H[t]);
      if (ok) upload(H[t], &Dv[t]);
    }
    if (!ok) {
      std::printf("%4d %2d %2d %2d %2d %2d %3d | not representable (R %d, columns per range %d)\n", c.NP, c.S, c.E, c.PE, c.PG, c.AT, c.DBG, H[0].V.R, H[0].V.CWID);
      for (int t = 0; t < 2; ++t) release(&Dv[t]);
      continue;
    }
    const size_t lds = (size_t)kAcc * 8;
    OK(hipFuncSetAttribute((const void*)var->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch_pair = [&](int t, bool epi) {
      hipLaunchKernelGGL(var->fn, dim3(Dv[t].V.G), dim3(kT), lds, st, Dv[t].V, (const double*)x[t], Dv[t].partial);
      if (epi)
        hipLaunchKernelGGL(k_combine<true>, dim3((rows[t] + 255) / 256), dim3(256), 0, st, rows[t], Dv[t].V.rows_pad, c.S, (const double*)Dv[t].partial, yout[t],
                           (const double*)e[0], (const double*)e[1], (const double*)e[2], e[3], 0.37);
      else
        hipLaunchKernelGGL(k_combine<false>, dim3((rows[t] + 255) / 256), dim3(256), 0, st, rows[t], Dv[t].V.rows_pad, c.S, (const double*)Dv[t].partial, yout[t],
                           (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, 0.0);
    };
    double err[2] = {0, 0};
    for (int t = 0; t < 2; ++t) {
      launch_pair(t, false);
      OK(hipStreamSynchronize(st));
      OK(hipGetLastError());
      std::vector<double> got(rows[t]);
      OK(hipMemcpy(got.data(), yout[t], (size_t)rows[t] * 8, hipMemcpyDeviceToHost));
      for (int r = 0; r < rows[t]; ++r) {
        const double d = std::fabs(got[r] - ref[t][r]) / (mag[t][r] > 0 ? mag[t][r] : 1.0);
        if (!(d <= err[t])) err[t] = d;
      }
    }
    bool same = true;
    {
      std::vector<double> g1(rows[0]), g2(rows[0]);
      launch_pair(0, false); OK(hipStreamSynchronize(st)); OK(hipMemcpy(g1.data(), yout[0], (size_t)rows[0] * 8, hipMemcpyDeviceToHost));
      launch_pair(0, false); OK(hipStreamSynchronize(st)); OK(hipMemcpy(g2.data(), yout[0], (size_t)rows[0] * 8, hipMemcpyDeviceToHost));
      same = std::memcmp(g1.data(), g2.data(), (size_t)rows[0] * 8) == 0;
    }
    double us[5] = {0, 0, 0, 0, 0};
    for (int r = -3; r < reps; ++r) {
      hipLaunchKernelGGL(k_touch, dim3((big + 255) / 256), dim3(256), 0, st, big, (const double*)tv[0], (const double*)tv[1], (const double*)tv[2], (const double*)tv[3],
                         (const double*)tv[4], tv[5], tv[6], tv[7]);
      OK(hipEventRecord(ev[0], st));
      hipLaunchKernelGGL(var->fn, dim3(Dv[0].V.G), dim3(kT), lds, st, Dv[0].V, (const double*)x[0], Dv[0].partial);
      OK(hipEventRecord(ev[1], st));
      hipLaunchKernelGGL(k_combine<true>, dim3((rows[0] + 255) / 256), dim3(256), 0, st, rows[0], Dv[0].V.rows_pad, c.S, (const double*)Dv[0].partial, yout[0], (const double*)e[0],
                         (const double*)e[1], (const double*)e[2], e[3], 0.37);
      OK(hipEventRecord(ev[2], st));
      hipLaunchKernelGGL(var->fn, dim3(Dv[1].V.G), dim3(kT), lds, st, Dv[1].V, (const double*)x[1], Dv[1].partial);
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
    std::printf("%4d %2d %2d %2d %2d %2d %3d | %6d %8lld %8lld %3d-%-3d %5.1f | %9.1f %9.1f %9.1f %9.1f | %9.1f | %.2e %.2e %s  (build %.1f s, G %d)\n", c.NP, c.S, c.E, c.PE, c.PG, c.AT, c.DBG, H[0].V.R,
                H[0].pads + H[1].pads, H[0].deferred + H[1].deferred, H[0].min_steps, H[0].max_steps, H[0].lines_per_wave, us[0] / reps, us[1] / reps, us[2] / reps, us[3] / reps, us[4] / reps / 2, err[0], err[1],
                same ? "repro" : "NOT REPRODUCIBLE", H[0].build_s + H[1].build_s, H[0].V.G);
    for (int t = 0; t < 2; ++t) release(&Dv[t]);
  }
  std::printf("(times: hipEvent pairs around single launches inside the touch -> A -> combine -> A^T -> combine sequence; 'pair' = (A + comb + AT + comb) / 2;\n"
              " the panels' fused a_dual / at_step are 72.6 / 71.1 us on the same matrices, plain products 66.7 us; kill criterion: pair > 55 us)\n");
  return 0;
}
