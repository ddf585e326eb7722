// read_rate.hip -- what a kernel that ONLY READS gets out of HBM on this box: 1.38 GB (config 3's samples) summed by
// 16-byte loads, U independent loads in flight per lane, W workgroups of 256 threads per CU (grid-stride over the
// buffer, consecutive lanes consecutive 16-byte units).  The encoder reads its samples at 3.2 TB/s (0.40 of the 8 TB/s
// HBM peak) and the frame walk's candidate scan its stream at 3.2 TB/s: is that the kernels or the box?
// Build: hipcc --offload-arch=gfx950 -O3 -o read_rate read_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int U>
__global__ void __launch_bounds__(256) rd(const uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = p[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n16; i += stride) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// the same bytes, every workgroup ONE contiguous span of the buffer (consecutive trips consecutive 4 KB): the frame walk's
// candidate scan reads its stream this way (its candidates then come out in stream order), 4 096 spans at once
__global__ void __launch_bounds__(256) rd_span(const uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ out) {
  const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  uint32_t acc = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
static void run_span(const uint4* d, size_t n16, uint32_t* out, int wgs_per_cu, int cus) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(rd_span, dim3(grid), dim3(256), 0, 0, d, n16, out);
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(rd_span, dim3(grid), dim3(256), 0, 0, d, n16, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("  one span per workgroup, %2d workgroups per CU: %.3f ms  %.2f TB/s\n", wgs_per_cu, ms / 10, n16 * 16.0 / (ms / 10 * 1e-3) * 1e-12);
}

template <int U>
static void run(const uint4* d, size_t n16, uint32_t* out, int wgs_per_cu, int cus) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(rd<U>, dim3(grid), dim3(256), 0, 0, d, n16, out);
  (void)hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(rd<U>, dim3(grid), dim3(256), 0, 0, d, n16, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  printf("  %d loads in flight per lane, %2d workgroups per CU: %.3f ms  %.2f TB/s\n", U, wgs_per_cu, ms / 10, n16 * 16.0 / (ms / 10 * 1e-3) * 1e-12);
}

int main() {
  const size_t bytes = 1382400000ull, n16 = bytes / 16;
  uint4* d; uint32_t* out;
  (void)hipMalloc(&d, bytes); (void)hipMalloc(&out, 1 << 20);
  (void)hipMemset(d, 1, bytes);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("read-only kernel over %.2f GB, %d CUs\n", bytes * 1e-9, cus);
  for (int w : {2, 4, 8, 16}) { run<1>(d, n16, out, w, cus); run<2>(d, n16, out, w, cus); run<4>(d, n16, out, w, cus); run<8>(d, n16, out, w, cus); }
  for (int w : {2, 4, 8, 16, 32}) run_span(d, n16, out, w, cus);
  const size_t n16s = 363376758ull / 16;   // config 3's stream
  printf("the same on 363 MB:\n");
  for (int w : {8, 16}) { run<1>(d, n16s, out, w, cus); run_span(d, n16s, out, w, cus); }
  return 0;
}
