// Tiny host-side fork/join helper for the O(nnz) setup passes (transpose, panel construction).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace cuopt_amd {

// threads the process may actually run on (cgroup / affinity aware), capped
inline int host_threads(int cap = 16)
{
  static const int env_cap = [] {
    const char* e = getenv("CUOPT_AMD_HOST_THREADS");  // (the set-up passes are memory bound: 16 is where they stop scaling on the boxes measured)
    return e && atoi(e) > 0 ? atoi(e) : 0;
  }();
  if (env_cap) cap = env_cap;
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

// Large host temporaries of the set-up (transposed copy, panel permutations: ~400 MB at 1e7 nonzeros).  Giving them
// back to the OS costs 10-30 ms of address-space lock per solve (munmap), which stalls hipMalloc, page faults and kernel
// launches of whichever thread runs next to it, and the next solve pays the page faults again.  They come from a small
// process-wide pool instead: blocks are reused by size, the pool keeps at most kPoolCap bytes.
class HostPool {
 public:
  static HostPool& instance()
  {
    static HostPool* pool = new HostPool();  // never destroyed: no ordering issues at exit
    return *pool;
  }
  void* take(size_t bytes, size_t* capacity)
  {
    {
      std::lock_guard<std::mutex> lock(m_);
      size_t best = free_.size();
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].first >= bytes && free_[i].first <= 2 * bytes + 4096 && (best == free_.size() || free_[i].first < free_[best].first))
          best = i;
      if (best != free_.size()) {
        void* p   = free_[best].second;
        *capacity = free_[best].first;
        held_ -= free_[best].first;
        free_.erase(free_.begin() + best);
        return p;
      }
    }
    *capacity = bytes;
    return ::operator new(bytes);
  }
  void give(void* p, size_t capacity)
  {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lock(m_);
      if (capacity >= kPoolMinBlock && held_ + capacity <= kPoolCap) {
        free_.emplace_back(capacity, p);
        held_ += capacity;
        return;
      }
    }
    ::operator delete(p);
  }

 private:
  static constexpr size_t kPoolCap = (size_t)1 << 30, kPoolMinBlock = (size_t)1 << 20;
  std::mutex m_;
  std::vector<std::pair<size_t, void*>> free_;
  size_t held_ = 0;
};
// uninitialised array of trivially-copyable T from the pool (like std::unique_ptr<T[]>(new T[n]), minus the munmap)
template <class T>
class PoolArray {
 public:
  PoolArray() = default;
  explicit PoolArray(size_t n) { reset(n); }
  PoolArray(PoolArray&& o) noexcept : p_(o.p_), cap_(o.cap_) { o.p_ = nullptr, o.cap_ = 0; }
  PoolArray& operator=(PoolArray&& o) noexcept
  {
    if (this != &o) {
      HostPool::instance().give(p_, cap_);
      p_ = o.p_, cap_ = o.cap_;
      o.p_ = nullptr, o.cap_ = 0;
    }
    return *this;
  }
  PoolArray(const PoolArray&)            = delete;
  PoolArray& operator=(const PoolArray&) = delete;
  ~PoolArray() { HostPool::instance().give(p_, cap_); }
  void reset(size_t n)
  {
    HostPool::instance().give(p_, cap_);
    p_ = static_cast<T*>(HostPool::instance().take(std::max<size_t>(n, 1) * sizeof(T), &cap_));
  }
  T* get() const { return p_; }
  T& operator[](size_t i) const { return p_[i]; }

 private:
  T* p_       = nullptr;
  size_t cap_ = 0;
};

}  // namespace cuopt_amd
