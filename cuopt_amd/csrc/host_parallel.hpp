// Tiny host-side fork/join helper for the O(nnz) setup passes (transpose, panel construction).
#pragma once
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace cuopt_amd {

// Knobs that only tests, harnesses and tuning sweeps turn live in ONE environment string, read at every query (set-up time only):
//   CUOPT_AMD_TUNE="slab_bytes=65536,panel_nnz=4096,dense=0"
// keys: slab_bytes, panel_nnz, panel_ws_bytes, panel_seg, jag_waves, dense, roctx, transpose_direct, fault_inject, soft_communicator,
// simplex_grade, simplex_pricing, simplex_solves, simplex_debug (each documented where it is read).
inline bool tune_get(const char* key, std::string* out)
{
  const char* e = getenv("CUOPT_AMD_TUNE");
  if (!e) return false;
  const std::string all(e), k(key);
  size_t pos = 0;
  while (pos <= all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    const std::string item = all.substr(pos, end - pos);
    const size_t eq        = item.find('=');
    if (item.substr(0, eq) == k) {
      if (out) *out = eq == std::string::npos ? std::string("1") : item.substr(eq + 1);
      return true;
    }
    pos = end + 1;
  }
  return false;
}
inline long long tune_int(const char* key, long long fallback)
{
  std::string v;
  return tune_get(key, &v) ? atoll(v.c_str()) : fallback;
}

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

// A small persistent pool for the set-up's fork/join passes.  Creating and joining 16 std::threads costs 0.3-0.5 ms per parallel
// region, and a set-up at 1e6 x 1e6 runs a dozen of them (19 ms in all): the workers are created once and sleep on a condition
// variable between regions.  One region at a time (a second caller -- the A^T side's thread next to the A side's -- runs its tasks
// on threads of its own, as before the pool existed); the caller takes part in its region.  The workers are never joined (no
// ordering issues at exit); a forked child starts with a fresh pool.
class TaskPool {
 public:
  static TaskPool& instance()
  {
    static TaskPool* pool = new TaskPool();
    return *pool;
  }
  // false: the pool is busy with another region (the caller falls back to threads of its own)
  template <class F>
  bool run(int workers_wanted, int tasks, F& fn)
  {
    if (in_region()) return false;  // a task of a running region asks for a region of its own: never try_lock a mutex this thread may hold
    std::unique_lock<std::mutex> region(region_, std::try_to_lock);
    if (!region.owns_lock()) return false;
    struct Mark {
      Mark() { in_region() = true; }
      ~Mark() { in_region() = false; }
    } mark;
    ensure_workers(workers_wanted - 1);
    std::exception_ptr failure;
    std::mutex failure_m;
    // (a task that throws -- bad_alloc in a set-up pass -- must not end a detached worker, i.e. the process: the first exception is
    //  kept and rethrown by the caller of the region once every task has been accounted for)
    std::function<void(int)> call = [&fn, &failure, &failure_m](int t) {
      try {
        fn(t);
      } catch (...) {
        std::lock_guard<std::mutex> lk(failure_m);
        if (!failure) failure = std::current_exception();
      }
    };
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &call, tasks_ = tasks, next_ = 0, done_ = 0, active_ = std::min<int>((int)threads_.size(), workers_wanted - 1);
      ++generation_;
    }
    cv_work_.notify_all();
    work(call, tasks);
    std::unique_lock<std::mutex> lk(m_);
    job_ = nullptr;  // (no worker joins from here on; those inside hold `call` until they leave)
    cv_done_.wait(lk, [&] { return done_ == tasks_ && inflight_ == 0; });
    lk.unlock();
    if (failure) std::rethrow_exception(failure);
    return true;
  }

 private:
  static bool& in_region()
  {
    static thread_local bool flag = false;
    return flag;
  }
  void work(std::function<void(int)>& call, int tasks)
  {
    for (;;) {
      int t;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (next_ >= tasks) return;
        t = next_++;
      }
      call(t);
      std::lock_guard<std::mutex> lk(m_);
      if (++done_ == tasks_) cv_done_.notify_all();
    }
  }
  void ensure_workers(int count)
  {
    std::lock_guard<std::mutex> lk(m_);
    while ((int)threads_.size() < count) {
      const int id = (int)threads_.size();
      threads_.emplace_back([this, id] {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(int)>* call = nullptr;
          int tasks = 0;
          {
            std::unique_lock<std::mutex> lk(m_);
            cv_work_.wait(lk, [&] { return generation_ != seen; });
            seen = generation_;
            if (id >= active_ || !job_) continue;
            call = job_, tasks = tasks_;
            ++inflight_;
          }
          work(*call, tasks);
          {
            std::lock_guard<std::mutex> lk(m_);
            if (--inflight_ == 0) cv_done_.notify_all();
          }
        }
      });
      threads_.back().detach();
    }
  }
  TaskPool()
  {
    pthread_atfork(nullptr, nullptr, [] {
      // the child has none of the parent's threads: a fresh pool object (the old one's mutexes may be held by threads that do not exist)
      new (&instance()) TaskPool(0);
    });
  }
  explicit TaskPool(int) {}
  std::mutex region_, m_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::thread> threads_;
  std::function<void(int)>* job_ = nullptr;
  int tasks_ = 0, next_ = 0, done_ = 0, active_ = 0, inflight_ = 0;
  uint64_t generation_ = 0;
};

// fn(task) for task in [0, tasks): dealt to the pool's threads one task at a time (the caller works too)
template <class F>
inline void parallel_tasks(int tasks, F&& fn, int64_t work_hint = 1 << 30)
{
  int nt = std::min(host_threads(), tasks);
  if (work_hint < (1 << 18)) nt = 1;  // not worth waking a thread
  if (nt <= 1) {
    for (int t = 0; t < tasks; ++t) fn(t);
    return;
  }
  if (TaskPool::instance().run(nt, tasks, fn)) return;
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
