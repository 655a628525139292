// Host-side copy engine of the boundary: a small PERSISTENT pool of threads that moves weight vectors between the
// caller's pageable arrays and the library's pinned staging buffers (si_construct_push: host -> pinned -> HBM;
// si_reconstruct / si_sample_rwmh_weights: HBM -> pinned -> host).  One host thread copies ~10 GB/s; the PCIe link
// moves ~55 GB/s, so the staging copy, not the DMA, would set the pace of a push without it.  Threads are created once
// (creating them per call costs more than a 4 MB copy); between jobs they spin for a short while (a loop of pushes hands
// over the next vector within ~100 us: a futex wake-up per worker and push would cost as much as its slice of the copy)
// and then sleep on a condition variable.
#include <sched.h>
#include <unistd.h>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define SI_CPU_RELAX() _mm_pause()
#else
#define SI_CPU_RELAX() std::this_thread::yield()
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
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

// "<quota> <period>" of a cgroup-v2 cpu.max file ("max 100000" = unlimited) -> CPUs granted, 0 = no limit / unparsable
double parse_cpu_max(const char* text) {
  if (!text) return 0.0;
  long long quota = 0, period = 0;
  if (std::sscanf(text, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) return (double)quota / (double)period;
  return 0.0;
}

static double read_quota_file(const char* path) {
  FILE* f = std::fopen(path, "r");
  if (!f) return -1.0;
  char buf[128] = {0};
  const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
  std::fclose(f);
  buf[n] = 0;
  return parse_cpu_max(buf);
}

// CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (cgroup v2 cpu.max, else v1
// cpu.cfs_quota_us / cpu.cfs_period_us).  The GPU boxes show 256 logical CPUs and grant 16: threads sized by the mask alone
// get the whole process throttled ~80 ms at a time.
int host_cpu_budget() {
  int avail = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = std::max(1, CPU_COUNT(&set));
  double q = read_quota_file("/sys/fs/cgroup/cpu.max");
  if (q < 0.0) {   // cgroup v1
    FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    long long quota = -1, period = 0;
    if (fq && fp && std::fscanf(fq, "%lld", &quota) == 1 && std::fscanf(fp, "%lld", &period) == 1 && quota > 0 && period > 0)
      q = (double)quota / (double)period;
    if (fq) std::fclose(fq);
    if (fp) std::fclose(fp);
  }
  if (q > 0.0) avail = std::max(1, std::min(avail, (int)q));
  return avail;
}

// threads of the copy pool (the caller included) for `nproc` processes of this library sharing the host (one per GPU):
// SI_HOST_COPY_THREADS wins; else half of this process's share of the CPU budget, at most 8
int host_copy_plan(int budget, int nproc, const char* env) {
  if (env && *env) return std::max(1, std::min(64, atoi(env)));
  return std::max(1, std::min(8, budget / (2 * std::max(1, nproc))));
}

namespace {

int pool_threads() { return host_copy_plan(host_cpu_budget(), 1, getenv("SI_HOST_COPY_THREADS")); }

class CopyPool {
 public:
  CopyPool() : pid_(getpid()) {
    const int want = pool_threads() - 1;  // the calling thread copies a slice too
    for (int i = 0; i < want; ++i) {
      try {
        workers_.emplace_back([this, i] { run(i); });
      } catch (const std::system_error&) {  // no more threads to be had (process limit): fewer workers
        break;
      }
    }
    slices_.resize(workers_.size());
    claimed_ = std::vector<std::atomic<int>>(workers_.size());
    limit_.store((int)workers_.size(), std::memory_order_relaxed);
  }
  // several processes of the library share the host (si_comm_init_rank tells): use fewer workers per copy
  void set_share(int nproc) {
    if (getenv("SI_HOST_COPY_THREADS")) return;   // an explicit size stays
    const int want = host_copy_plan(host_cpu_budget(), nproc, nullptr) - 1;
    limit_.store(std::max(0, std::min((int)workers_.size(), want)), std::memory_order_relaxed);
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
    // (after a fork() the child owns the pool object but none of its threads: plain memcpy there)
    const size_t nw = getpid() == pid_ ? (size_t)limit_.load(std::memory_order_relaxed) : 0;
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
      for (size_t i = 0; i < workers_.size(); ++i) {
        const size_t lo = std::min(bytes, per * (i + 1)), hi = std::min(bytes, per * (i + 2));
        slices_[i] = i < nw ? Slice{dst + lo, src + lo, hi - lo} : Slice{nullptr, nullptr, 0};
        claimed_[i].store(i < nw ? 0 : 1, std::memory_order_relaxed);
      }
      pending_.store((int)nw, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
      sleepers = sleeping_;
    }
    if (sleepers > 0) cv_.notify_all();   // spinning workers see the new generation by themselves
    std::memcpy(dst, src, std::min(bytes, per));
    // the slices are equal: the others finish within microseconds.  A worker that has not even STARTED after a few
    // milliseconds (descheduled under a CPU quota, or gone) loses its slice to the caller; a slice in progress is waited for.
    const auto t0 = std::chrono::steady_clock::now();
    int polls = 0;
    bool rescued = false;
    while (pending_.load(std::memory_order_acquire) != 0) {
      SI_CPU_RELAX();
      if (!rescued && (++polls & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
        rescued = true;
        for (size_t i = 0; i < nw; ++i)
          if (claimed_[i].exchange(1, std::memory_order_acq_rel) == 0) {
            if (slices_[i].bytes) std::memcpy(slices_[i].dst, slices_[i].src, slices_[i].bytes);
            pending_.fetch_sub(1, std::memory_order_release);
          }
      }
    }
  }
  int threads() const { return limit_.load(std::memory_order_relaxed) + 1; }

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
        SI_CPU_RELAX();
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
        // The slice is read AND claimed inside the critical section in which copy() publishes a generation (slices, claim
        // words, pending count): a worker descheduled between the two could otherwise claim the NEXT copy's slice with the
        // previous copy's (stale, possibly freed) pointers in hand and take one off the next copy's pending count.
        // A claim taken here belongs to generation `seen`; its copy() cannot return -- and no later generation begin --
        // before this worker has taken its slice off pending_.
        std::lock_guard<std::mutex> lk(m_);
        seen = gen_.load(std::memory_order_acquire);
        if (stop_) return;
        s = slices_[(size_t)i];
        if (claimed_[(size_t)i].exchange(1, std::memory_order_acq_rel) != 0) continue;   // not part of this copy, or taken over by the caller
      }
      if (s.bytes) std::memcpy(s.dst, s.src, s.bytes);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  const pid_t pid_;
  std::vector<std::thread> workers_;
  std::vector<Slice> slices_;
  std::vector<std::atomic<int>> claimed_;   // per worker slice: 0 = open, 1 = taken (by its worker or, late, by the caller)
  std::atomic<int> limit_{0};               // workers used per copy (<= workers_.size())
  std::mutex m_, use_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> pending_{0};
  int sleeping_ = 0;   // workers blocked on cv_ (guarded by m_)
  bool stop_ = false;
};

CopyPool& pool() {
  // never destroyed: a static destructor joining the workers at exit could stall behind a copy on another thread
  static CopyPool* p = new CopyPool;
  return *p;
}

}  // namespace

void host_copy(void* dst, const void* src, size_t bytes) {
  pool().copy(static_cast<char*>(dst), static_cast<const char*>(src), bytes);
}

int host_copy_threads() { return pool().threads(); }
void host_copy_set_share(int nproc) { pool().set_share(nproc); }

}  // namespace si
