// pcie_copy.hip -- how fast can host buffers the CALLER owns (pageable memory) reach HBM and come back?
// Measures the candidates for the host-buffer entry points (x3_encode / x3_decode_stream):
//   a) hipMemcpy straight from / to pageable memory (the runtime stages it)
//   b) hipHostRegister the caller's pages, DMA, hipHostUnregister
//   c) memcpy into a pinned staging buffer with 1..T threads, then DMA
// usage: pcie_copy [MiB] [threads]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t n, int threads) {
  if (threads <= 1) { std::memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = ((n / threads) + 4095) & ~(size_t)4095;
  for (int t = 0; t < threads; ++t) {
    const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    if (hi > lo) th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? std::atol(argv[1]) : 1024;
  const int maxthr = argc > 2 ? std::atoi(argv[2]) : 8;
  const size_t n = mib << 20;
  char* host = (char*)std::malloc(n);
  char* back = (char*)std::malloc(n);
  std::memset(host, 1, n);
  std::memset(back, 2, n);
  void* dev; CK(hipMalloc(&dev, n));
  char* pin; CK(hipHostMalloc((void**)&pin, n));
  std::memset(pin, 3, n);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMemcpy(dev, pin, n, hipMemcpyHostToDevice));
  auto gbs = [&](double t) { return n / t / 1e9; };
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); CK(hipMemcpy(dev, host, n, hipMemcpyHostToDevice)); double t1 = now();
    std::printf("a) H2D pageable hipMemcpy        %7.2f GB/s\n", gbs(t1 - t0));
    t0 = now(); CK(hipMemcpy(back, dev, n, hipMemcpyDeviceToHost)); t1 = now();
    std::printf("a) D2H pageable hipMemcpy        %7.2f GB/s\n", gbs(t1 - t0));
  }
  {
    double t0 = now(); CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t1 = now();
    std::printf("   H2D pinned                    %7.2f GB/s\n", gbs(t1 - t0));
    t0 = now(); CK(hipMemcpyAsync(pin, dev, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t1 = now();
    std::printf("   D2H pinned                    %7.2f GB/s\n", gbs(t1 - t0));
  }
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); CK(hipHostRegister(host, n, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
    CK(hipHostUnregister(host)); double t3 = now();
    std::printf("b) register %7.2f GB/s, H2D %7.2f GB/s, unregister %7.2f GB/s, all %7.2f GB/s\n", gbs(t1 - t0),
                gbs(t2 - t1), gbs(t3 - t2), gbs(t3 - t0));
    t0 = now(); CK(hipHostRegister(back, n, hipHostRegisterDefault)); t1 = now();
    CK(hipMemcpyAsync(back, dev, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t2 = now();
    CK(hipHostUnregister(back)); t3 = now();
    std::printf("b) register %7.2f GB/s, D2H %7.2f GB/s, unregister %7.2f GB/s, all %7.2f GB/s\n", gbs(t1 - t0),
                gbs(t2 - t1), gbs(t3 - t2), gbs(t3 - t0));
  }
  for (int thr = 1; thr <= maxthr; thr *= 2) {
    double t0 = now(); par_copy(pin, host, n, thr); double t1 = now();
    std::printf("c) memcpy pageable->pinned, %2d threads  %7.2f GB/s\n", thr, gbs(t1 - t0));
    t0 = now(); par_copy(back, pin, n, thr); t1 = now();
    std::printf("c) memcpy pinned->pageable, %2d threads  %7.2f GB/s\n", thr, gbs(t1 - t0));
  }
  {  // chunked pipeline, one thread: memcpy chunk k+1 into pinned while chunk k is on the wire
    const size_t chunk = 16u << 20;
    for (int thr = 1; thr <= maxthr; thr *= 4) {
      double t0 = now();
      hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
      size_t k = 0;
      for (size_t off = 0; off < n; off += chunk, ++k) {
        const size_t len = std::min(chunk, n - off);
        char* slot = pin + (k & 1) * chunk;
        if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
        par_copy(slot, host + off, len, thr);
        CK(hipMemcpyAsync((char*)dev + off, slot, len, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[k & 1], s));
      }
      CK(hipStreamSynchronize(s));
      double t1 = now();
      std::printf("c) chunked 16 MiB staging H2D, %2d threads  %7.2f GB/s\n", thr, gbs(t1 - t0));
    }
  }
  {  // d) is hipMemcpyAsync on pageable memory asynchronous to the host, and do H2D and D2H overlap?
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void* dev2; CK(hipMalloc(&dev2, n));
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now(); CK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, s)); double t1 = now();
      CK(hipStreamSynchronize(s)); double t2 = now();
      std::printf("d) pageable H2D async: call returns after %7.2f ms, done after %7.2f ms\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3);
      t0 = now(); CK(hipMemcpyAsync(back, dev2, n, hipMemcpyDeviceToHost, s2)); t1 = now();
      CK(hipStreamSynchronize(s2)); t2 = now();
      std::printf("d) pageable D2H async: call returns after %7.2f ms, done after %7.2f ms\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3);
      t0 = now();
      CK(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, s)); t1 = now();
      CK(hipMemcpyAsync(back, dev2, n, hipMemcpyDeviceToHost, s2)); t2 = now();
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); double t3 = now();
      std::printf("d) pageable H2D + D2H on two streams, one thread: issued after %7.2f / %7.2f ms, both done after %7.2f ms (%.1f GB/s each way)\n",
                  (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3, gbs(t3 - t0));
      t0 = now();
      std::thread th([&] { (void)hipMemcpy(back, dev2, n, hipMemcpyDeviceToHost); });
      CK(hipMemcpy(dev, host, n, hipMemcpyHostToDevice));
      th.join(); t3 = now();
      std::printf("d) pageable H2D + D2H from two threads: both done after %7.2f ms (%.1f GB/s each way)\n", (t3 - t0) * 1e3, gbs(t3 - t0));
      t0 = now();
      CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s));
      CK(hipMemcpyAsync(pin, dev2, n, hipMemcpyDeviceToHost, s2));   // (overwrites pin while it is read: bandwidth only)
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); t3 = now();
      std::printf("d) pinned H2D + D2H on two streams: both done after %7.2f ms (%.1f GB/s each way)\n", (t3 - t0) * 1e3, gbs(t3 - t0));
    }
  }
  return 0;
}
