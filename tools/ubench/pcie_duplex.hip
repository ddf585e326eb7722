// pcie_duplex.hip -- do an upload and a download of pageable host memory run side by side when two host threads issue
// them in chunks on two streams?  (The host-buffer entry points move 2N bytes one way and P the other: x3_encode
// 1.38 GB up + 0.36 GB down on config 3.)   usage: pcie_duplex [up MiB] [down MiB] [chunk MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t up = (size_t)(argc > 1 ? std::atol(argv[1]) : 1318) << 20;
  const size_t down = (size_t)(argc > 2 ? std::atol(argv[2]) : 346) << 20;
  const size_t chunk = (size_t)(argc > 3 ? std::atol(argv[3]) : 64) << 20;
  char* hu = (char*)std::malloc(up);
  char* hd = (char*)std::malloc(down);
  std::memset(hu, 1, up);
  std::memset(hd, 2, down);
  char *du, *dd;
  CK(hipMalloc((void**)&du, up));
  CK(hipMalloc((void**)&dd, down));
  CK(hipMemset(dd, 7, down));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto upload = [&](size_t c) {
    for (size_t o = 0; o < up; o += c) {
      CK(hipMemcpyAsync(du + o, hu + o, std::min(c, up - o), hipMemcpyHostToDevice, s1));
      CK(hipStreamSynchronize(s1));
    }
  };
  auto download = [&](size_t c) {
    for (size_t o = 0; o < down; o += c) {
      CK(hipMemcpyAsync(hd + o, dd + o, std::min(c, down - o), hipMemcpyDeviceToHost, s2));
      CK(hipStreamSynchronize(s2));
    }
  };
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); upload(up); double t1 = now(); download(down); double t2 = now();
    std::printf("whole, one after the other : up %6.2f ms (%5.1f GB/s) down %6.2f ms (%5.1f GB/s) sum %6.2f ms\n", (t1 - t0) * 1e3,
                up / (t1 - t0) / 1e9, (t2 - t1) * 1e3, down / (t2 - t1) / 1e9, (t2 - t0) * 1e3);
    t0 = now(); upload(chunk); t1 = now(); download(chunk * down / up + 4096); t2 = now();
    std::printf("chunks, one after the other: up %6.2f ms down %6.2f ms sum %6.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3);
    t0 = now();
    {
      std::thread th([&] { CK(hipSetDevice(0)); download(chunk * down / up + 4096); });
      upload(chunk);
      th.join();
    }
    t1 = now();
    std::printf("chunks, two threads        : %6.2f ms (%5.1f GB/s both ways)\n", (t1 - t0) * 1e3, (up + down) / (t1 - t0) / 1e9);
    t0 = now();
    {
      std::thread th([&] { CK(hipSetDevice(0)); download(down); });
      upload(up);
      th.join();
    }
    t1 = now();
    std::printf("whole, two threads         : %6.2f ms\n", (t1 - t0) * 1e3);
  }
  return 0;
}
