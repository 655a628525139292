// Host-side copy engine of the boundary: a small PERSISTENT pool of threads that moves weight vectors between the
// caller's pageable arrays and the library's pinned staging buffers (si_construct_push: host -> pinned -> HBM;
// si_reconstruct / si_sample_rwmh_weights: HBM -> pinned -> host).  One host thread copies ~10 GB/s; the PCIe link
// moves ~55 GB/s, so the staging copy, not the DMA, would set the pace of a push without it.  Threads are created once
// (creating them per call costs more than a 4 MB copy); between jobs they spin for a short while (a loop of pushes hands
// over the next vector within ~100 us: a futex wake-up per worker and push would cost as much as its slice of the copy)
// and then sleep on a condition variable.
#include <sched.h>

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "si_internal.h"

#ifndef SI_COPY_SPIN_US
#define SI_COPY_SPIN_US 50    // measured on the GPU box (tools/host_copy_bench.cpp, profiles/r03_host_copy_pool.log): 81 GB/s back to back, 62 GB/s with 3 ms between copies
#endif

namespace si {

namespace {

int pool_threads() {
  int avail = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = CPU_COUNT(&set);
  if (const char* e = getenv("SI_HOST_COPY_THREADS")) return std::max(1, atoi(e));
  return std::max(1, std::min(8, avail / 2));
}

class CopyPool {
 public:
  CopyPool() {
    const int want = pool_threads() - 1;  // the calling thread copies a slice too
    for (int i = 0; i < want; ++i) {
      try {
        workers_.emplace_back([this, i] { run(i); });
      } catch (const std::system_error&) {  // no more threads to be had (process limit): fewer workers
        break;
      }
    }
    slices_.resize(workers_.size());
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void copy(char* dst, const char* src, size_t bytes) {
    const size_t nw = workers_.size();
    if (nw == 0 || bytes < ((size_t)1 << 20)) {
      std::memcpy(dst, src, bytes);
      return;
    }
    std::lock_guard<std::mutex> use(use_);  // one copy at a time (contexts on several host threads share the pool)
    const size_t parts = nw + 1;
    const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;  // 4 KiB-granular slices
    int sleepers;
    {
      std::lock_guard<std::mutex> lk(m_);
      for (size_t i = 0; i < nw; ++i) {
        const size_t lo = std::min(bytes, per * (i + 1)), hi = std::min(bytes, per * (i + 2));
        slices_[i] = {dst + lo, src + lo, hi - lo};
      }
      pending_.store((int)nw, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
      sleepers = sleeping_;
    }
    if (sleepers > 0) cv_.notify_all();   // spinning workers see the new generation by themselves
    std::memcpy(dst, src, std::min(bytes, per));
    while (pending_.load(std::memory_order_acquire) != 0) _mm_pause();   // the slices are equal: the others finish within us
  }
  int threads() const { return (int)workers_.size() + 1; }

 private:
  struct Slice {
    char* dst;
    const char* src;
    size_t bytes;
  };
  void run(int i) {
    uint64_t seen = 0;
    for (;;) {
      // spin SI_COPY_SPIN_US for the next job, then sleep
      const auto t0 = std::chrono::steady_clock::now();
      int polls = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        _mm_pause();
        if ((++polls & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SI_COPY_SPIN_US)) {
          std::unique_lock<std::mutex> lk(m_);
          ++sleeping_;
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
          --sleeping_;
          break;
        }
      }
      Slice s;
      {
        std::lock_guard<std::mutex> lk(m_);   // orders the read of the slice after the writer's update
        seen = gen_.load(std::memory_order_acquire);
        if (stop_) return;
        s = slices_[(size_t)i];
      }
      if (s.bytes) std::memcpy(s.dst, s.src, s.bytes);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> workers_;
  std::vector<Slice> slices_;
  std::mutex m_, use_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> pending_{0};
  int sleeping_ = 0;   // workers blocked on cv_ (guarded by m_)
  bool stop_ = false;
};

CopyPool& pool() {
  static CopyPool p;
  return p;
}

}  // namespace

void host_copy(void* dst, const void* src, size_t bytes) {
  pool().copy(static_cast<char*>(dst), static_cast<const char*>(src), bytes);
}

int host_copy_threads() { return pool().threads(); }

}  // namespace si
