// write_pattern.hip -- how does HBM write traffic (rocprofv3 WRITE_SIZE) depend on the shape of a store instruction?
//   hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ./write_pattern
// Every kernel writes the same 1 GiB exactly once, 16 bytes per lane and store:
//   contig      a wave's 64 lanes write 1 KB contiguous
//   chunk<C>    a wave writes 1024/C chunks of C bytes (C/16 neighbouring lanes each), chunk j of row r at r*ROW + ..., rows
//               ROW = 20000 bytes apart (the decoder's output: 64 frames side by side), walking along the rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ROW 20000u
__global__ void contig(uint4* out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// group g owns 64 rows of ROW bytes; step s: lane l writes piece (l % P) of chunk (l / P) ... rows (l / P) + 16/.. see below
template <int C>
__global__ void chunked(uint8_t* out, uint32_t groups, uint32_t off) {
  constexpr uint32_t P = C / 16;          // lanes per chunk
  constexpr uint32_t RPI = 64 / P;        // rows per instruction
  const uint32_t lane = threadIdx.x;
  for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
    uint8_t* base = out + (size_t)g * 64 * ROW + off;
    const uint32_t nch = (ROW - 64) / C;    // whole chunks per row (the ragged ends are left out)
    for (uint32_t k = 0; k < nch; ++k)
      for (uint32_t r0 = 0; r0 < 64; r0 += RPI) {
        const uint32_t r = r0 + lane / P, p = lane % P;
        *reinterpret_cast<uint4*>(base + (size_t)r * ROW + (size_t)k * C + 16 * p) = make_uint4(k, r, p, 7);
      }
  }
}
// the two 64-byte halves of every 128-byte line written far apart in time (aligned chunks: off picks the aligned rows'
// phase per row): all even chunks of a group first, then all odd ones
__global__ void halves(uint8_t* out, uint32_t groups) {
  const uint32_t lane = threadIdx.x;
  for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
    uint8_t* base = out + (size_t)g * 64 * ROW;
    const uint32_t nch = (ROW - 128) / 64;
    for (uint32_t par = 0; par < 2; ++par)
      for (uint32_t k = par; k < nch; k += 2)
        for (uint32_t r0 = 0; r0 < 64; r0 += 16) {
          const uint32_t r = r0 + lane / 4, p = lane % 4;
          const size_t row = (size_t)r * ROW;
          const size_t al = (64 - ((uintptr_t)(base + row) & 63)) & 63;   // first 64-byte boundary of the row
          *reinterpret_cast<uint4*>(base + row + al + (size_t)k * 64 + 16 * p) = make_uint4(k, r, p, 7);
        }
  }
}
int main() {
  const uint32_t groups = 800;                    // 800 * 64 * 20000 = 1.024 GB
  uint8_t* d;
  hipMalloc(&d, (size_t)groups * 64 * ROW + 4096);
  hipMemset(d, 0, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    contig<<<4096, 256>>>((uint4*)d, (size_t)groups * 64 * ROW / 16);
    chunked<64><<<groups, 64>>>(d, groups, 0);      // rows start at 20000*r: 32 mod 64 on odd rows -> misaligned there
    chunked<64><<<groups, 64>>>(d, groups, 32);     // (the other rows misaligned)
    chunked<128><<<groups, 64>>>(d, groups, 0);
    chunked<32><<<groups, 64>>>(d, groups, 0);
    chunked<160><<<groups, 64>>>(d, groups, 0);     // the decoder's 160-byte runs (10 lanes per run, 6.4 runs per instruction)
    halves<<<groups, 64>>>(d, groups);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: contig %zu, chunked %zu\n", (size_t)groups * 64 * ROW, (size_t)groups * 64 * ((ROW - 64) / 64) * 64);
  return 0;
}
