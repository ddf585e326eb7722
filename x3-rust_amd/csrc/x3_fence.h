// x3_fence.h -- device allocations with guard pages (a debugging aid; off unless X3HIP_FENCE is set).
//
// The GPU sanitizer is not available on the pool this library is tested on, and hipMalloc pads what it hands out to its
// page granularity: a kernel that reads or writes a few bytes behind a buffer goes unnoticed until, once in a long soak,
// the buffer happens to end where the mapping does.  With X3HIP_FENCE=<align> (a power of two >= 16; 1 means 16) every
// device buffer the library allocates -- its own (ensure(), the tables) and the ones it hands out (x3_dev_alloc) -- is
// mapped through the virtual memory API into a reservation of its own with an UNMAPPED granule in front of and behind
// it, and placed so that it ENDS (rounded up to <align> bytes) at the end of the mapping: the first access behind it is
// a memory fault at once, in every run.  tools/fuzz_parity.py and the GPU tests run under it (profiles/r5/fence.txt).
//
// Kernels of this library read their input in aligned 16-byte chunks, so up to 15 bytes behind the last byte of a buffer
// are read by design (never across a 16-byte line, hence never across a page): <align> = 16 is the tightest fence that
// design passes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

struct X3FenceRec { void* va; size_t reserved; void* mapped; size_t map_size; hipMemGenericAllocationHandle_t h; };

inline size_t x3_fence_align() {
  static const size_t a = [] {
    const char* e = getenv("X3HIP_FENCE");
    if (!e || !*e) return (size_t)0;
    size_t v = (size_t)strtoull(e, nullptr, 10);
    if (v == 0) return (size_t)0;
    if (v < 16) v = 16;
    while (v & (v - 1)) v += v & (0 - v);   // (up to a power of two)
    return v;
  }();
  return a;
}
inline std::mutex& x3_fence_mutex() { static std::mutex m; return m; }
inline std::unordered_map<void*, X3FenceRec>& x3_fence_map() { static std::unordered_map<void*, X3FenceRec> m; return m; }

inline hipError_t x3_dmalloc_impl(void** p, size_t bytes) {
  const size_t align = x3_fence_align();
  if (!align) return hipMalloc(p, bytes);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  if (gran < 4096) gran = 4096;
  const size_t need = ((bytes ? bytes : 1) + align - 1) & ~(align - 1);
  const size_t map_size = (need + gran - 1) / gran * gran;
  // (map with the device idle: a buffer mapped from one thread while kernels launched by another were in flight was read
  // with stale contents once in a few thousand calls -- X3HIP_FENCE_NOSYNC=1 to see that again)
  static const bool nosync = getenv("X3HIP_FENCE_NOSYNC") != nullptr;
  static const bool log = getenv("X3HIP_FENCE_LOG") != nullptr;
  if (!nosync && (e = hipDeviceSynchronize()) != hipSuccess) return e;
  X3FenceRec r{};
  r.reserved = map_size + 2 * gran;
  r.map_size = map_size;
  if ((e = hipMemAddressReserve(&r.va, r.reserved, gran, nullptr, 0)) != hipSuccess) return e;
  if ((e = hipMemCreate(&r.h, map_size, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, r.reserved); return e; }
  r.mapped = static_cast<char*>(r.va) + gran;
  if ((e = hipMemMap(r.mapped, map_size, 0, r.h, 0)) != hipSuccess) {
    (void)hipMemRelease(r.h);
    (void)hipMemAddressFree(r.va, r.reserved);
    return e;
  }
  hipMemAccessDesc ad{};
  ad.location.type = hipMemLocationTypeDevice;
  ad.location.id = dev;
  ad.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(r.mapped, map_size, &ad, 1)) != hipSuccess) {
    (void)hipMemUnmap(r.mapped, map_size);
    (void)hipMemRelease(r.h);
    (void)hipMemAddressFree(r.va, r.reserved);
    return e;
  }
  *p = static_cast<char*>(r.mapped) + (map_size - need);
  // X3HIP_FENCE_FILL=<byte>: what a fresh buffer holds (hipMalloc promises nothing; fresh pages happen to be zero, recycled
  // ones are not -- 165 finds code that counts on zeros)
  auto undo = [&](hipError_t err) {   // (nothing of a reservation that is not handed out stays behind: ADVICE r5)
    (void)hipMemUnmap(r.mapped, map_size);
    (void)hipMemRelease(r.h);
    (void)hipMemAddressFree(r.va, r.reserved);
    *p = nullptr;
    return err;
  };
  if (const char* fill = getenv("X3HIP_FENCE_FILL")) {
    if ((e = hipMemset(r.mapped, (int)strtol(fill, nullptr, 10) & 255, map_size)) != hipSuccess) return undo(e);
  }
  if (!nosync && (e = hipDeviceSynchronize()) != hipSuccess) return undo(e);
  if (log) fprintf(stderr, "x3_fence: %p + %zu (mapped %p + %zu)\n", *p, bytes, r.mapped, map_size);
  std::lock_guard<std::mutex> lk(x3_fence_mutex());
  x3_fence_map()[*p] = r;
  return hipSuccess;
}

inline hipError_t x3_dfree_impl(void* p) {
  if (!p) return hipSuccess;
  if (x3_fence_align()) {
    X3FenceRec r{};
    bool mine = false;
    {
      std::lock_guard<std::mutex> lk(x3_fence_mutex());
      auto it = x3_fence_map().find(p);
      if (it != x3_fence_map().end()) { r = it->second; x3_fence_map().erase(it); mine = true; }
    }
    if (mine) {
      hipError_t e = hipDeviceSynchronize();
      hipError_t e2 = hipMemUnmap(r.mapped, r.map_size);
      if (e == hipSuccess) e = e2;
      e2 = hipMemRelease(r.h);
      if (e == hipSuccess) e = e2;
      // The address range stays RESERVED (and unmapped) for the life of the process: an access through a stale pointer is a
      // fault too, and no later buffer is ever mapped where an earlier one was (with the range handed back, the next
      // reservation got the same addresses and kernels read what looked like the previous mapping's pages).
      return e;
    }
  }
  return hipFree(p);
}

template <typename T> inline hipError_t x3_dmalloc(T** p, size_t bytes) {
  void* q = nullptr;
  const hipError_t e = x3_dmalloc_impl(&q, bytes);
  if (e == hipSuccess) *p = static_cast<T*>(q);
  return e;
}
inline hipError_t x3_dfree(void* p) { return x3_dfree_impl(p); }
