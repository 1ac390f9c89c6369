// Two questions the round-4 review left open about the gather that bounds the unstructured SpMV, in one harness:
//
// (1) CALIBRATION of rocprofv3's FETCH_SIZE for scattered 8-byte reads.  The guide's rule (FETCH_SIZE x 2) is calibrated on wide
//     coalesced streams; the 1.47x "wasted traffic" of k_panel_a_dual rests on applying it to gather misses.  Kernels with a KNOWN
//     miss count: G gathers of 8 bytes, one per `stride` bytes of a buffer far larger than L2 + Infinity Cache (every line is touched
//     once per launch and cannot survive to the next), next to a coalesced stream of known size.  Run under
//       rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- /tmp/gather_probe calibrate
//     and read FETCH_SIZE per gather: 64 B at strides >= 128 and 32 B at stride 64 means the fabric moves whole 128-byte lines that the
//     counter tallies at 64 (the stream's factor holds for gathers); 64 B at stride 64 too would mean 64-byte sectors are fetched on
//     their own and the x2 overstates the gather share.
//
// (2) The vector-side idea DESIGN 9.2 listed last: the gathered fp64 vector as two fp32 PLANES (hi = (float)x, lo = (float)(x - hi)),
//     so that an L2-resident slab holds twice the columns.  (The sum hi + lo carries 48 of the 53 mantissa bits: NOT exact -- it
//     would break the bit-exact contract of the short-row sums -- so it would have to win clearly.)  `planes` times, over the same
//     1e7 random indices: one 8-byte gather per nonzero inside windows of W bytes of an fp64 vector, against TWO 4-byte gathers (hi
//     and lo planes, windows of W / 2 bytes each: the same columns in half the bytes per plane).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_gather_stride(const double* __restrict__ v, long long gathers, long long stride_doubles, double* __restrict__ out)
{
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < gathers; i += (long long)gridDim.x * 256) acc += v[i * stride_doubles];
  if (acc == 123.456) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const double4* __restrict__ v, long long quads, double* __restrict__ out)
{
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long long)gridDim.x * 256) {
    const double4 q = v[i];
    acc += q.x + q.y + q.z + q.w;
  }
  if (acc == 123.456) out[0] = acc;
}
// one 8-byte gather per index / two 4-byte gathers per index (hi and lo planes); idx[k] is a column inside the window of its slab
__global__ void __launch_bounds__(256) k_gather64(const int* __restrict__ idx, long long n, const double* __restrict__ v, double* __restrict__ out)
{
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) acc += v[__builtin_nontemporal_load(idx + k)];
  if (acc == 123.456) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather32x2(const int* __restrict__ idx, long long n, const float* __restrict__ hi, const float* __restrict__ lo, double* __restrict__ out)
{
  double acc = 0.0;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
    const int j = __builtin_nontemporal_load(idx + k);
    acc += (double)hi[j] + (double)lo[j];
  }
  if (acc == 123.456) out[0] = acc;
}

// (3) What the L2 -> L1 fabric delivers (round 6): every workgroup re-reads a window that stays in its XCD's L2 -- WIDE: 16 bytes per lane,
//     whole lines, the most the path can carry; LINE: 8 bytes out of every 128-byte line (what a gather that misses the L1 uses of the line
//     it pulls).  `l2`: prints both rates; the unstructured SpMV's gathers are priced against them (profiles/r06_cell_probe.txt).
__global__ void __launch_bounds__(256) k_l2_wide(const double2* __restrict__ v, int nchunks /* of 16 KB */, int reps, double* __restrict__ out)
{
  // (blockIdx & 7 = XCD: one window per XCD; workgroup j of the XCD reads chunk (j + 257 r) mod nchunks at repetition r: the CU's eight
  //  workgroups never touch a chunk another of them read less than ~1 MB of traffic ago -- nothing comes out of the 32 KB L1)
  const double2* w = v + (size_t)(blockIdx.x & 7) * nchunks * 1024;
  const int j = blockIdx.x >> 3;
  double acc = 0.0;
  for (int r = 0; r < reps; ++r) {
    const double2* c = w + (size_t)((j + 257 * r) % nchunks) * 1024 + threadIdx.x;
    const double2 q0 = c[0], q1 = c[256], q2 = c[512], q3 = c[768];
    acc += (q0.x + q1.y) + (q2.x + q3.y);
  }
  if (acc == 123.456) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_l2_line(const double* __restrict__ v, int nchunks /* of 256 lines = 32 KB */, int reps, double* __restrict__ out)
{
  const double* w = v + (size_t)(blockIdx.x & 7) * nchunks * 4096;
  const int j = blockIdx.x >> 3;
  double acc = 0.0;
  for (int r = 0; r < reps; r += 4) {  // (four independent lines per lane in flight)
    double q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = w[(size_t)((j + 129 * (r + u)) % nchunks) * 4096 + threadIdx.x * 16 + (threadIdx.x & 15)];
    acc += (q[0] + q[1]) + (q[2] + q[3]);
  }
  if (acc == 123.456) out[0] = acc;
}

// (3b) the same, with RANDOM lines of the window: mode 0 any line; mode 1 line = 16 q + (lane mod 16) with q random (every 16 lanes of a
//      request cover the 16 residues of the line number: balanced if a channel is a residue); mode 2 the same with 32; mode 3 with 64
__global__ void __launch_bounds__(256) k_l2_random(const double* __restrict__ v, int window_lines, int reps, int mode, double* __restrict__ out)
{
  // (mode >= 8: ONE window for the whole chip -- the gathered vector of an SpMV that is not walked in slabs)
  const double* w = v + (mode >= 8 ? (size_t)0 : (size_t)(blockIdx.x & 7) * window_lines * 16);
  mode &= 7;
  unsigned long long state = 88172645463325252ull ^ ((unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull);
  const int g = mode == 1 ? 16 : mode == 2 ? 32 : 64;
  double acc = 0.0;
  for (int r = 0; r < reps; r += 4) {
    double q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      state ^= state << 13, state ^= state >> 7, state ^= state << 17;
      const unsigned rnd = (unsigned)(state >> 20);
      const int line     = mode == 0 ? (int)(rnd % (unsigned)window_lines) : (int)(rnd % (unsigned)(window_lines / g)) * g + (int)(threadIdx.x % g);
      q[u] = w[(size_t)line * 16 + (threadIdx.x & 15)];
    }
    acc += (q[0] + q[1]) + (q[2] + q[3]);
  }
  if (acc == 123.456) out[0] = acc;
}

// (3c) the gather fed from MEMORY: indices (4 B) and values (8 B) streamed coalesced, one gather per index out of the XCD's window, U
//      independent (index -> gather) chains per lane in flight: what an SpMV does, without rows and without LDS
template <int U>
__global__ void __launch_bounds__(256) k_stream_gather(const int* __restrict__ idx, const double* __restrict__ val, long long per_xcd, const double* __restrict__ v,
                                                       int window_lines, double* __restrict__ out)
{
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
  const double* w = v + (size_t)xcd * window_lines * 16;
  const int* ix = idx + (size_t)xcd * per_xcd;
  const double* vl = val + (size_t)xcd * per_xcd;
  double acc = 0.0;
  for (long long k = (long long)j * 256 * U + threadIdx.x; k + (U - 1) * 256 < per_xcd; k += (long long)nb * 256 * U) {
    int c[U];
    double a[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = __builtin_nontemporal_load(ix + k + u * 256), a[u] = __builtin_nontemporal_load(vl + k + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = w[c[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += a[u] * x[u];
  }
  if (acc == 123.456) out[0] = acc;
}

static float time_ms(hipStream_t s, hipEvent_t e0, hipEvent_t e1) { float ms = 0; (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); (void)s; return ms; }

int main(int argc, char** argv)
{
  const bool calibrate = argc > 1 && !std::strcmp(argv[1], "calibrate");
  if (argc > 1 && !std::strcmp(argv[1], "l2")) {
    hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    double* out; OK(hipMalloc((void**)&out, 64));
    std::printf("L2 -> L1: 2048 workgroups of 256 threads re-read a window that sits in their XCD's L2 (one window per XCD)\n");
    std::printf("%-18s %16s %16s %22s\n", "window per XCD", "wide 16 B/lane", "8 B per line", "lines/s (8 B per line)");
    for (int window_kb : {512, 1024, 2048, 3072}) {
      const size_t bytes = (size_t)window_kb * 1024;
      double* buf; OK(hipMalloc((void**)&buf, bytes * 8)); OK(hipMemset(buf, 0, bytes * 8));
      const int cw = (int)(bytes / 16384), cl = (int)(bytes / 32768), reps = 256;
      float tw = 0, tl = 0;
      for (int rep = -2; rep < 5; ++rep) {
        OK(hipEventRecord(e0, s)); k_l2_wide<<<2048, 256, 0, s>>>((const double2*)buf, cw, reps, out); OK(hipEventRecord(e1, s));
        const float a = time_ms(s, e0, e1);
        OK(hipEventRecord(e0, s)); k_l2_line<<<2048, 256, 0, s>>>(buf, cl, reps, out); OK(hipEventRecord(e1, s));
        const float b = time_ms(s, e0, e1);
        if (rep >= 0) tw += a, tl += b;
      }
      const double bw = 2048.0 * reps * 16384.0, lines = 2048.0 * reps * 256.0;
      std::printf("%6d KiB         %10.1f TB/s %10.1f TB/s (as 128-B lines) %12.1f G lines/s", window_kb, bw / (tw / 5 * 1e-3) / 1e12, lines * 128 / (tl / 5 * 1e-3) / 1e12,
                  lines / (tl / 5 * 1e-3) / 1e9);
      std::printf("   random lines:");
      for (int mode = 0; mode < 4; ++mode) {
        float tr = 0;
        for (int rep = -2; rep < 5; ++rep) {
          OK(hipEventRecord(e0, s)); k_l2_random<<<2048, 256, 0, s>>>(buf, (int)(bytes / 128), reps, mode, out); OK(hipEventRecord(e1, s));
          const float a = time_ms(s, e0, e1);
          if (rep >= 0) tr += a;
        }
        std::printf(" %s %.1f", mode == 0 ? "any" : mode == 1 ? "| 16-residue" : mode == 2 ? "| 32-residue" : "| 64-residue", lines / (tr / 5 * 1e-3) / 1e9);
      }
      std::printf(" G lines/s\n");
      if (window_kb == 3072) {
        for (int shared_kb : {4096, 8192, 16384, 65536}) {  // one window for all eight XCDs
          double* big; OK(hipMalloc((void**)&big, (size_t)shared_kb * 1024)); OK(hipMemset(big, 0, (size_t)shared_kb * 1024));
          float tr = 0;
          for (int rep = -2; rep < 5; ++rep) {
            OK(hipEventRecord(e0, s)); k_l2_random<<<2048, 256, 0, s>>>(big, shared_kb * 8, reps, 8, out); OK(hipEventRecord(e1, s));
            const float a = time_ms(s, e0, e1);
            if (rep >= 0) tr += a;
          }
          std::printf("  ONE window of %6d KiB for the whole chip, random lines: %.1f G lines/s\n", shared_kb, lines / (tr / 5 * 1e-3) / 1e9);
          (void)hipFree(big);
        }
      }
      (void)hipFree(buf);
    }
    {
      // streamed indices: 1.25e6 gathers per XCD (1e7 in all) from a 1.33 MiB window per XCD, indices random doubles of the window
      const long long per = 1250000;
      const int wl = 10922;  // lines of a 1.33 MiB window
      std::vector<int> h((size_t)per * 8);
      unsigned long long st = 88172645463325252ull;
      for (auto& e : h) { st ^= st << 13, st ^= st >> 7, st ^= st << 17; e = (int)((st >> 20) % (unsigned long long)(wl * 16)); }
      int* di; double *dv, *dw;
      OK(hipMalloc((void**)&di, h.size() * 4)); OK(hipMalloc((void**)&dv, h.size() * 8)); OK(hipMalloc((void**)&dw, (size_t)wl * 128 * 8));
      OK(hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice)); OK(hipMemset(dv, 0, h.size() * 8)); OK(hipMemset(dw, 0, (size_t)wl * 128 * 8));
      std::printf("gathers fed from memory (1e7 indices + values streamed, 1.33 MiB window per XCD), G gathers/s by chains per lane and grid:\n");
      for (int grid : {1024, 2048, 4096, 8192}) {
        std::printf("  grid %5d:", grid);
        for (int U : {1, 2, 4, 8}) {
          float t = 0;
          for (int rep = -2; rep < 5; ++rep) {
            OK(hipEventRecord(e0, s));
            if (U == 1) k_stream_gather<1><<<grid, 256, 0, s>>>(di, dv, per, dw, wl, out);
            else if (U == 2) k_stream_gather<2><<<grid, 256, 0, s>>>(di, dv, per, dw, wl, out);
            else if (U == 4) k_stream_gather<4><<<grid, 256, 0, s>>>(di, dv, per, dw, wl, out);
            else k_stream_gather<8><<<grid, 256, 0, s>>>(di, dv, per, dw, wl, out);
            OK(hipEventRecord(e1, s));
            const float a = time_ms(s, e0, e1);
            if (rep >= 0) t += a;
          }
          std::printf("  U=%d %6.1f (%5.1f us)", U, 1e7 / (t / 5 * 1e-3) / 1e9, t / 5 * 1e3);
        }
        std::printf("\n");
      }
    }
    return 0;
  }
  hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  double* out; OK(hipMalloc((void**)&out, 64));
  if (calibrate) {
    const long long G = 8000000;             // gathers per launch
    const size_t bytes = (size_t)G * 256 + 4096;  // 2 GB: stride up to 256 B, far beyond L2 (32 MiB) + Infinity Cache (256 MiB)
    double* buf; OK(hipMalloc((void**)&buf, bytes)); OK(hipMemset(buf, 0, bytes));
    std::printf("calibration kernels (read their FETCH_SIZE from the rocprofv3 counter table):\n");
    for (long long stride : {64LL, 128LL, 256LL}) {
      for (int rep = 0; rep < 3; ++rep) {
        OK(hipEventRecord(e0, s));
        k_gather_stride<<<4096, 256, 0, s>>>(buf, G, stride / 8, out);
        OK(hipEventRecord(e1, s));
        const float ms = time_ms(s, e0, e1);
        if (rep == 2) std::printf("  k_gather_stride  stride %4lld B: %lld gathers of 8 B, %.1f us  (expected: one missed line each)\n", stride, G, 1e3 * ms);
      }
    }
    const long long quads = (long long)(512ull << 20) / 32;  // a 512 MB coalesced stream: FETCH_SIZE should read 256 MB
    for (int rep = 0; rep < 3; ++rep) {
      OK(hipEventRecord(e0, s));
      k_stream<<<4096, 256, 0, s>>>((const double4*)buf, quads, out);
      OK(hipEventRecord(e1, s));
      const float ms = time_ms(s, e0, e1);
      if (rep == 2) std::printf("  k_stream         512 MB coalesced (32 B per lane), %.1f us = %.2f TB/s\n", 1e3 * ms, 512.0 / 1024 / 1024 * 1.048576 / (ms * 1e-3) / 1e6 * 1e6 / 1e6);
    }
    return 0;
  }
  // ---- planes: 1e7 random gathers, slab by slab as the panel layout orders them
  const long long nnz = 10000000;
  const int n = 1000000;
  std::printf("%-34s %12s %12s %12s\n", "window of the gathered vector", "fp64 8 B us", "2 x fp32 us", "ratio");
  for (int window_cols : {4096, 65536, 174763, 349526, 1000000}) {  // 32 KiB (L1), 512 KiB, 1.33 MiB (the slab), 2.67 MiB, the whole vector
    std::vector<int> h((size_t)nnz);
    unsigned long long state = 88172645463325252ull;
    const int slabs = (n + window_cols - 1) / window_cols;
    for (long long k = 0; k < nnz; ++k) {
      state ^= state << 13, state ^= state >> 7, state ^= state << 17;
      const int slab = (int)(k * slabs / nnz);  // entries walk the slabs in order, random inside a slab
      const int lo = slab * window_cols, width = std::min(window_cols, n - lo);
      h[(size_t)k] = lo + (int)(state % (unsigned long long)width);
    }
    int* idx; double* v; float *hi, *lo;
    OK(hipMalloc((void**)&idx, (size_t)nnz * 4)); OK(hipMalloc((void**)&v, (size_t)n * 8)); OK(hipMalloc((void**)&hi, (size_t)n * 4)); OK(hipMalloc((void**)&lo, (size_t)n * 4));
    OK(hipMemcpy(idx, h.data(), (size_t)nnz * 4, hipMemcpyHostToDevice)); OK(hipMemset(v, 0, (size_t)n * 8)); OK(hipMemset(hi, 0, (size_t)n * 4)); OK(hipMemset(lo, 0, (size_t)n * 4));
    float t64 = 0, t32 = 0;
    for (int rep = -2; rep < 10; ++rep) {
      OK(hipEventRecord(e0, s)); k_gather64<<<2048, 256, 0, s>>>(idx, nnz, v, out); OK(hipEventRecord(e1, s));
      float a = time_ms(s, e0, e1);
      OK(hipEventRecord(e0, s)); k_gather32x2<<<2048, 256, 0, s>>>(idx, nnz, hi, lo, out); OK(hipEventRecord(e1, s));
      float b = time_ms(s, e0, e1);
      if (rep >= 0) t64 += a, t32 += b;
    }
    char name[64];
    std::snprintf(name, sizeof(name), "%d columns (%.2f MiB of fp64)", window_cols, window_cols * 8.0 / 1048576.0);
    std::printf("%-34s %12.1f %12.1f %12.2f\n", name, 100.0 * t64, 100.0 * t32, t32 / t64);
    (void)hipFree(idx); (void)hipFree(v); (void)hipFree(hi); (void)hipFree(lo);
  }
  return 0;
}
