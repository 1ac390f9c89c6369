// An OUTSIDE yardstick for the "0.31 is this part's floor for a uniformly random matrix" claim (round-4 review, task 4): rocSPARSE's CSR
// SpMV (adaptive, LRB, row split, nonzero split and the library's default, each with its analysis / preprocess step) on the SAME
// matrices the solver multiplies -- C3's A and A^T, dumped as raw arrays by scripts/dump_csr.py -- inside the same kind of loop
// pdlpdev_time_kernel uses: the two products alternate, with a 512 MB memset between them so that neither the 120 MB matrix nor the
// vectors survive in the 256 MiB Infinity Cache from one launch to the next (the solver's four kernels evict each other the same
// way).  Never linked into the product: a harness, like the rest of tools/.
//   hipcc -O3 --offload-arch=gfx950 -Wno-deprecated-declarations tools/rocsparse_yardstick.cpp -lrocsparse -o /tmp/rocsparse_yardstick
//   /tmp/rocsparse_yardstick <dir with a_off.i32 a_idx.i32 a_val.f64 at_off.i32 at_idx.i32 at_val.f64 dims.txt>
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RS(e) do { rocsparse_status s_ = (e); if (s_ != rocsparse_status_success) { std::printf("rocsparse status %d line %d\n", (int)s_, __LINE__); return 1; } } while (0)

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

struct Mat {
  int m = 0, n = 0;
  long long nnz = 0;
  int *off = nullptr, *idx = nullptr;
  double* val = nullptr;
  rocsparse_spmat_descr descr = nullptr;
};

int main(int argc, char** argv)
{
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  const bool verbose = argc > 3;
  const std::string only = argc > 2 ? argv[2] : "";  // ONE algorithm per process (the library's analysis data does not survive a change
  if (argc < 2) { std::printf("usage: %s <dir> [algorithm [verbose]]\n", argv[0]); return 2; }  // of algorithm in one process: see profiles/r05_rocsparse_yardstick.txt)
  const std::string dir = argv[1];
  int m = 0, n = 0;
  {
    FILE* f = std::fopen((dir + "/dims.txt").c_str(), "r");
    if (!f || std::fscanf(f, "%d %d", &m, &n) != 2) { std::printf("dims.txt missing\n"); return 2; }
    std::fclose(f);
  }
  Mat M[2];
  const char* names[2] = {"a", "at"};
  for (int t = 0; t < 2; ++t) {
    std::vector<int> off, idx;
    std::vector<double> val;
    if (!slurp(dir + "/" + names[t] + "_off.i32", &off) || !slurp(dir + "/" + names[t] + "_idx.i32", &idx) || !slurp(dir + "/" + names[t] + "_val.f64", &val)) {
      std::printf("cannot read %s\n", names[t]);
      return 2;
    }
    M[t].m = t == 0 ? m : n, M[t].n = t == 0 ? n : m, M[t].nnz = (long long)idx.size();
    OK(hipMalloc((void**)&M[t].off, off.size() * 4)); OK(hipMalloc((void**)&M[t].idx, idx.size() * 4)); OK(hipMalloc((void**)&M[t].val, val.size() * 8));
    OK(hipMemcpy(M[t].off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(M[t].idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(M[t].val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
  }
  const int big = std::max(m, n);
  double *x = nullptr, *y = nullptr;
  char* flush = nullptr;
  const size_t flush_bytes = (size_t)512 << 20;
  OK(hipMalloc((void**)&x, (size_t)big * 8)); OK(hipMalloc((void**)&y, (size_t)big * 8)); OK(hipMalloc((void**)&flush, flush_bytes));
  {
    std::vector<double> h((size_t)big);
    for (int i = 0; i < big; ++i) h[i] = 1.0 + (i % 7) * 0.25;
    OK(hipMemcpy(x, h.data(), (size_t)big * 8, hipMemcpyHostToDevice));
    OK(hipMemcpy(y, h.data(), (size_t)big * 8, hipMemcpyHostToDevice));
  }
  hipStream_t s;
  OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  rocsparse_handle handle;
  RS(rocsparse_create_handle(&handle));
  RS(rocsparse_set_stream(handle, s));
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  struct { const char* name; rocsparse_spmv_alg alg; } algs[] = {{"default", rocsparse_spmv_alg_default}, {"csr_adaptive", rocsparse_spmv_alg_csr_adaptive},
                                                               {"csr_rowsplit", rocsparse_spmv_alg_csr_rowsplit}, {"csr_lrb", rocsparse_spmv_alg_csr_lrb},
                                                               {"csr_nnzsplit", rocsparse_spmv_alg_csr_nnzsplit}};
  std::printf("rocSPARSE CSR SpMV, fp64, %d x %d, %lld nonzeros; A then A^T alternate, 512 MB memset between launches (Infinity Cache flushed)\n", m, n, M[0].nnz);
  std::printf("%-24s %14s %14s %16s %16s %16s\n", "algorithm", "A us", "A^T us", "A us (no flush)", "analysis A ms", "analysis A^T ms");
  const double alpha = 1.0, beta = 0.0;
  for (auto& a : algs) {
    if (!only.empty() && only != a.name) continue;
    rocsparse_dnvec_descr vx[2], vy[2];
    void* buf[2] = {nullptr, nullptr};
    size_t bytes[2] = {0, 0};
    float prep_ms[2] = {0, 0};
    bool ok = true;
    for (int t = 0; t < 2 && ok; ++t) {
      // (a matrix descriptor keeps the analysis of ONE algorithm: a fresh one per algorithm)
      RS(rocsparse_create_csr_descr(&M[t].descr, M[t].m, M[t].n, M[t].nnz, M[t].off, M[t].idx, M[t].val, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                    rocsparse_index_base_zero, rocsparse_datatype_f64_r));
      RS(rocsparse_create_dnvec_descr(&vx[t], M[t].n, t == 0 ? x : y, rocsparse_datatype_f64_r));
      RS(rocsparse_create_dnvec_descr(&vy[t], M[t].m, t == 0 ? y : x, rocsparse_datatype_f64_r));
      if (verbose) std::printf("[%s %s] buffer size\n", a.name, names[t]);
      rocsparse_status st = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, M[t].descr, vx[t], &beta, vy[t], rocsparse_datatype_f64_r, a.alg,
                                           rocsparse_spmv_stage_buffer_size, &bytes[t], nullptr);
      if (verbose) std::printf("[%s %s] %zu bytes, status %d; preprocess\n", a.name, names[t], bytes[t], (int)st);
      if (st != rocsparse_status_success) { ok = false; break; }
      OK(hipMalloc(&buf[t], bytes[t] ? bytes[t] : 16));
      OK(hipEventRecord(e0, s));
      st = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, M[t].descr, vx[t], &beta, vy[t], rocsparse_datatype_f64_r, a.alg, rocsparse_spmv_stage_preprocess,
                          &bytes[t], buf[t]);
      OK(hipEventRecord(e1, s));
      OK(hipEventSynchronize(e1));
      OK(hipEventElapsedTime(&prep_ms[t], e0, e1));
      if (st != rocsparse_status_success) ok = false;
    }
    if (!ok) { std::printf("%-24s not available for this format / build\n", a.name); continue; }
    OK(hipStreamSynchronize(s));
    if (verbose) std::printf("[%s] compute\n", a.name);
    double us[2] = {0, 0}, us_noflush = 0;
    const int reps = 30;
    for (int mode = 0; mode < 2; ++mode) {  // 0: flushed, 1: A alone back to back
      for (int r = -3; r < reps; ++r)
        for (int t = 0; t < (mode == 0 ? 2 : 1); ++t) {
          if (mode == 0) OK(hipMemsetAsync(flush, r & 1, flush_bytes, s));
          OK(hipEventRecord(e0, s));
          RS(rocsparse_spmv(handle, rocsparse_operation_none, &alpha, M[t].descr, vx[t], &beta, vy[t], rocsparse_datatype_f64_r, a.alg, rocsparse_spmv_stage_compute, &bytes[t], buf[t]));
          OK(hipEventRecord(e1, s));
          OK(hipEventSynchronize(e1));
          float ms = 0;
          OK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 0) (mode == 0 ? us[t] : us_noflush) += 1e3 * ms;
        }
    }
    std::printf("%-24s %14.1f %14.1f %16.1f %16.2f %16.2f\n", a.name, us[0] / reps, us[1] / reps, us_noflush / reps, prep_ms[0], prep_ms[1]);
    OK(hipStreamSynchronize(s));
    for (int t = 0; t < 2; ++t) { (void)hipFree(buf[t]); (void)rocsparse_destroy_dnvec_descr(vx[t]); (void)rocsparse_destroy_dnvec_descr(vy[t]); (void)rocsparse_destroy_spmat_descr(M[t].descr); }
  }
  std::printf("(the solver's own plain products on the same matrices: bench.py per_kernel_ms SPMV_A_PLAIN / SPMV_AT_PLAIN, timed inside full attempts)\n");
  return 0;
}
