// Launch helpers shared by the translation units of the core (pdlp_device.hip: set-up, the attempt, results; pdlp_eval.hip: the major
// iteration): argument packing for hipExtLaunchKernel-style launches with the context's timing hooks, and the per-layout launch wrappers
// (the two geometries of the jagged kernels, the two launches of a gather-free product).  Static state (the "attribute already set"
// lists) is per translation unit: setting a kernel's attribute twice is harmless.
#pragma once
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"

template <size_t... I, typename Tuple>
static void arg_pointers(Tuple& t, void** out, std::index_sequence<I...>)
{
  ((out[I] = (void*)&std::get<I>(t)), ...);
}
template <typename... KArgs, typename... Args>
static void launch_k(pdlpdev_ctx* c, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, Args... args)
{
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count");
  if (c->prof_armed && c->prof_used < pdlpdev_ctx::kProfPairs) {
    hipEvent_t e0_ = c->prof_ev[2 * c->prof_used], e1_ = c->prof_ev[2 * c->prof_used + 1];
    c->prof_used += 1;
    std::tuple<std::remove_cv_t<KArgs>...> vals{static_cast<KArgs>(args)...};
    void* ptrs[sizeof...(KArgs)];
    arg_pointers(vals, ptrs, std::index_sequence_for<KArgs...>{});
    (void)hipExtLaunchKernel((const void*)kernel, grid, block, ptrs, lds, c->stream, e0_, e1_, 0);
    return;
  }
  kernel<<<grid, block, lds, c->stream>>>(static_cast<KArgs>(args)...);
}
// Launch of a jagged-layout kernel: 80 or 160 KiB of dynamic LDS (the attribute is per kernel and device, set once)
template <typename... KArgs, typename... Args>
static int jag_launch(pdlpdev_ctx* c, void (*kernel)(JagView, KArgs...), const JagView& v, Args... args)
{
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  const size_t lds = jag_lds_bytes(v.waves);
  {
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key((const void*)kernel, c->device);
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done.push_back(key);
    }
  }
  launch_k(c, kernel, ((v.nblk + 7) & ~7) + v.nlong, v.waves * 64, lds, v, args...);
  return 0;
}
// the two geometries are two instantiations of every jagged kernel
#define JAG_LAUNCH(ctx, KERNEL, VIEW, ...) \
  ((VIEW).waves == 16 ? jag_launch(ctx, KERNEL<16>, VIEW, __VA_ARGS__) : jag_launch(ctx, KERNEL<8>, VIEW, __VA_ARGS__))


// the two launches of a gather-free SpMV: phase P with the gathered vector picked on the device (mode: see k_pb_products), ...
static int pb_products(pdlpdev_ctx* c, const pdlpdev_ctx::Pb& L, const double* v0, const double* v1, int mode, int in_loop)
{
  static std::mutex mu;
  static std::vector<std::pair<int, int>> done;
  {
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, int> key(L.p_threads, c->device);
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      if (L.p_threads == 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_pb_products<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      else HIP_TRY(hipFuncSetAttribute((const void*)k_pb_products<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done.push_back(key);
    }
  }
  const int grid   = (L.v.nwg + 7) & ~7;
  const size_t lds = sizeof(double) << L.v.panel_shift;
  if (L.p_threads == 1024) launch_k(c, k_pb_products<1024>, grid, 1024, lds, L.v, c->ctl, v0, v1, mode, in_loop);
  else launch_k(c, k_pb_products<512>, grid, 512, lds, L.v, c->ctl, v0, v1, mode, in_loop);
  return 0;
}
// ... and phase R with the epilogue of the call site (two skeletons: the image in LDS, or -- wide bins -- the accumulators in LDS)
template <typename... KArgs, typename... Args>
static int pb_rows_launch(pdlpdev_ctx* c, void (*kernel)(PbView, KArgs...), const pdlpdev_ctx::Pb& L, Args... args)
{
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  const size_t lds  = L.v.wide ? kPbwLdsBytes : kPbLdsBytes;
  {
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key((const void*)kernel, c->device);
    if (std::find(done.begin(), done.end(), key) == done.end()) {
      HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done.push_back(key);
    }
  }
  launch_k(c, kernel, (L.v.B + 7) & ~7, L.v.wide ? kPbwThreads : kPbThreads, lds, L.v, args...);
  return 0;
}
#define pb_rows(ctx, KERNEL, L, ...) ((L).v.wide ? pb_rows_launch(ctx, KERNEL<true>, L, __VA_ARGS__) : pb_rows_launch(ctx, KERNEL<false>, L, __VA_ARGS__))

