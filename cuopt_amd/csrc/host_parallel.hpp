// Tiny host-side fork/join helper for the O(nnz) setup passes (transpose, panel construction).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace cuopt_amd {

// threads the process may actually run on (cgroup / affinity aware), capped
inline int host_threads(int cap = 16)
{
  cpu_set_t set;
  int avail = (int)std::thread::hardware_concurrency();
  if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = CPU_COUNT(&set);
  return std::max(1, std::min(avail, cap));
}

// fn(task) for task in [0, tasks), statically chunked over the threads
template <class F>
inline void parallel_tasks(int tasks, F&& fn, int64_t work_hint = 1 << 30)
{
  int nt = std::min(host_threads(), tasks);
  if (work_hint < (1 << 18)) nt = 1;  // not worth a thread launch
  if (nt <= 1) {
    for (int t = 0; t < tasks; ++t) fn(t);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (int w = 0; w < nt; ++w)
    pool.emplace_back([&, w] {
      for (int t = w; t < tasks; t += nt) fn(t);
    });
  for (auto& th : pool) th.join();
}

}  // namespace cuopt_amd
