// Development harness for the host copy pool (csrc/host_copy.cpp): pageable -> pinned-sized buffer, cfg2 snapshot size,
// back-to-back like a loop of si_construct_push calls.  Built in variants of the spin time:
//   g++ -O2 -std=c++17 -pthread -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DSI_COPY_SPIN_US=200 tools/host_copy_bench.cpp \
//       subspaceinference.jl_amd/csrc/host_copy.cpp -o tools/bin/host_copy_bench_spin200
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
namespace si {
void host_copy(void*, const void*, size_t);
int host_copy_threads();
}
int main(int argc, char** argv) {
  const size_t n = 4189444;
  const int gap_us = argc > 1 ? atoi(argv[1]) : 0;   // idle time between copies (the caller's training step)
  std::vector<char> a(n * 100), b(n);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (char)(i * 131);
  for (int rep = 0; rep < 3; ++rep) {
    double busy = 0.0;
    for (int j = 0; j < 100; ++j) {
      auto t0 = std::chrono::steady_clock::now();
      si::host_copy(b.data(), a.data() + n * j, n);
      busy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (memcmp(b.data(), a.data() + n * j, n)) return 1;
      if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
    }
    printf("spin %d us, threads %d, gap %d us: 100 copies of %zu B in %.2f ms = %.1f GB/s\n", SI_COPY_SPIN_US, si::host_copy_threads(),
           gap_us, n, busy, 100.0 * n / busy / 1e6);
  }
  return 0;
}
