// The host copy pool (csrc/host_copy.cpp) under ThreadSanitizer / AddressSanitizer on the CPU: back-to-back copies (workers
// spinning), copies after an idle gap (workers asleep on the condition variable), odd sizes and offsets, two caller
// threads sharing the pool, and process exit with the pool alive.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
namespace si {
void host_copy(void*, const void*, size_t);
int host_copy_threads();
}
static bool run(size_t n, int reps, int gap_us, unsigned seed) {
  std::vector<unsigned char> a(n + 64), b(n + 64, 0);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (unsigned char)(i * 131 + seed);
  for (int r = 0; r < reps; ++r) {
    const size_t off = (size_t)(r % 7), len = n - (size_t)(r % 13);
    std::memset(b.data(), 0, b.size());
    si::host_copy(b.data() + off, a.data() + off + 1, len);
    if (std::memcmp(b.data() + off, a.data() + off + 1, len) != 0) return false;
    if (b[off + len] != 0 || (off > 0 && b[off - 1] != 0)) return false;   // nothing outside the range was written
    if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
  }
  return true;
}
int main() {
  bool ok = true;
  ok = ok && run(4189444, 20, 0, 1);        // cfg2 snapshot size, back to back
  ok = ok && run(4189444, 6, 2000, 2);      // with gaps: the workers go to sleep in between
  ok = ok && run(1 << 20, 10, 0, 3);        // the smallest size that is split
  ok = ok && run(1000, 10, 0, 4);           // below the threshold: plain memcpy
  ok = ok && run((1 << 20) + 4097, 10, 300, 5);
  bool ok2[2] = {false, false};
  std::thread t0([&] { ok2[0] = run(3 << 20, 15, 0, 6); }), t1([&] { ok2[1] = run(2 << 20, 15, 100, 7); });
  t0.join();
  t1.join();
  ok = ok && ok2[0] && ok2[1];
  std::printf("threads %d -> %s\n", si::host_copy_threads(), ok ? "HOST_COPY_OK" : "HOST_COPY_FAIL");
  return ok ? 0 : 1;
}
