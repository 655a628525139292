// DEVELOPMENT BUILD ONLY (python subspaceinference.jl_amd/build.py --dev; not part of libsubspace_hip.so).
// Guard-page device allocator for the library's own buffers: GPU AddressSanitizer is not available on this pool, and an
// out-of-bounds access of a kernel only faults when the page it lands on happens to be unmapped (round 3: a staging read
// in front of the weight vector passed every test for two rounds and faulted once the allocation order changed).
// With SI_GUARD_ALLOC=end|begin every dev_alloc reserves its own virtual range with an UNMAPPED granule on either side
// (hipMemAddressReserve / hipMemCreate / hipMemMap) and places the buffer flush against the end (or the start) of the
// mapped part, so that any access past the end (in front of the start) faults at once, whatever else is allocated.
// `end` leaves up to 8 bytes of slack (the kernels need 16-byte aligned bases).  Costs >= one granule (2 MiB) per
// buffer: for the test suite only.  tests: SI_TEST_LIB=tools/bin/libsubspace_hip_dev.so SI_GUARD_ALLOC=end pytest -m gpu
#ifdef SI_DEV_KNOBS
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace si {

namespace {
struct Guarded {
  void* base;
  size_t va, mapped, gran;
  hipMemGenericAllocationHandle_t handle;
};
std::mutex g_m;
std::map<void*, Guarded> g_live;

int guard_mode() {  // 0 off, 1 end, 2 begin
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SI_GUARD_ALLOC");
    mode = !e ? 0 : !strcmp(e, "end") ? 1 : !strcmp(e, "begin") ? 2 : 0;
  }
  return mode;
}
}  // namespace

hipError_t guard_malloc(void** out, size_t bytes) {
  const int mode = guard_mode();
  if (mode == 0) return hipMalloc(out, bytes);
  *out = nullptr;
  if (bytes == 0) bytes = 8;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  Guarded g{};
  g.gran = gran;
  g.mapped = (bytes + gran - 1) / gran * gran;
  g.va = g.mapped + 2 * gran;
  if ((e = hipMemAddressReserve(&g.base, g.va, gran, nullptr, 0)) != hipSuccess) return e;
  if ((e = hipMemCreate(&g.handle, g.mapped, &prop, 0)) != hipSuccess) {
    (void)hipMemAddressFree(g.base, g.va);
    return e;
  }
  char* lo = static_cast<char*>(g.base) + gran;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemMap(lo, g.mapped, 0, g.handle, 0)) != hipSuccess || (e = hipMemSetAccess(lo, g.mapped, &acc, 1)) != hipSuccess) {
    (void)hipMemRelease(g.handle);
    (void)hipMemAddressFree(g.base, g.va);
    return e;
  }
  // SI_GUARD_FILL=nan|zero: poison (all-ones bytes = a NaN in every double) or clear the fresh memory, to tell a read of
  // never-written memory (results turn NaN / depend on the fill) from everything else
  if (const char* f = getenv("SI_GUARD_FILL")) {
    (void)hipMemset(lo, !strcmp(f, "zero") ? 0 : 0xFF, g.mapped);
    (void)hipDeviceSynchronize();
  }
  char* p = lo;
  if (mode == 1) p = lo + ((g.mapped - bytes) & ~(size_t)15);
  std::lock_guard<std::mutex> lk(g_m);
  g_live[p] = g;
  *out = p;
  return hipSuccess;
}

hipError_t guard_free(void* p) {
  if (!p) return hipSuccess;
  Guarded g{};
  {
    std::lock_guard<std::mutex> lk(g_m);
    auto it = g_live.find(p);
    if (it == g_live.end()) return hipFree(p);
    g = it->second;
    g_live.erase(it);
  }
  (void)hipDeviceSynchronize();  // hipFree's semantics: nothing in flight touches the buffer any more
  char* lo = static_cast<char*>(g.base) + g.gran;
  hipError_t e = hipMemUnmap(lo, g.mapped);
  (void)hipMemRelease(g.handle);
  // The virtual range stays reserved for the life of the process.  Measured (tools/guard_probe.py): after
  // hipMemUnmap + hipMemAddressFree a later reservation gets the same addresses back and kernels on some XCDs still
  // translate them to the RELEASED pages (a second kernel read zeros where the first had written); never handing an
  // address out twice avoids that, and makes a stale pointer fault too.
  return e;
}

}  // namespace si
#endif  // SI_DEV_KNOBS
