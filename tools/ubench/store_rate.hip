// store_rate.hip -- how fast can a CU retire 16-byte-per-lane stores, by the shape of the store instruction?
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
// The decoder's output: 64 frames (rows, ROW = 20000 bytes apart) per wave, every store instruction writes 1 KB as
// 1024/C runs of C contiguous bytes (C/16 neighbouring lanes each) in 1024/C different rows.  Every kernel writes the
// same ~1 GB once; `split` waves share a group's rows (each takes every split-th step), so that the number of waves per
// CU can be varied with the bytes fixed.  NT: non-temporal stores.  `off`: byte offset of the runs in their rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ROW 20000u
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int C, bool NT>
__global__ void runs(uint8_t* out, uint32_t groups, uint32_t split, uint32_t off) {
  constexpr uint32_t P = C / 16, RPI = 64 / P;
  const uint32_t lane = threadIdx.x;
  const uint32_t g = blockIdx.x / split, part = blockIdx.x % split;
  uint8_t* base = out + (size_t)g * 64 * ROW + off;
  const uint32_t nch = (ROW - 1024) / C;
  for (uint32_t k = part; k < nch; k += split)
    for (uint32_t r0 = 0; r0 < 64; r0 += RPI) {
      const uint32_t r = r0 + lane / P, p = lane % P;
      u32x4 v = {k, r, p, 7u};
      u32x4* dst = reinterpret_cast<u32x4*>(base + (size_t)r * ROW + (size_t)k * C + 16 * p);
      if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}
// any run length (a multiple of 16): the decoder's flush -- piece t = 64*it + lane of row t / P; cache bits by MODE
#define ST(mod) asm volatile("global_store_dwordx4 %0, %1, off " mod ::"v"(dst), "v"(v) : "memory")
template <int MODE>
__global__ void window(uint8_t* out, uint32_t groups, uint32_t C, uint32_t off, uint32_t pitch) {
#undef ROW
#define ROW pitch
  const uint32_t P = C / 16, lane = threadIdx.x;
  uint8_t* base = out + (size_t)blockIdx.x * 64 * ROW + off;
  const uint32_t nch = (ROW - 1024) / C;
  for (uint32_t k = 0; k < nch; ++k)
    for (uint32_t t = lane; t < 64 * P; t += 64) {
      const uint32_t r = t / P, p = t - r * P;
      u32x4 v = {k, r, p, 7u};
      u32x4* dst = reinterpret_cast<u32x4*>(base + (size_t)r * ROW + (size_t)k * C + 16 * p);
      if (MODE == 0) ST("");
      if (MODE == 1) ST("nt");
      if (MODE == 2) ST("sc0");
      if (MODE == 3) ST("sc1");
      if (MODE == 4) ST("sc0 sc1");
      if (MODE == 5) ST("sc0 nt");
      if (MODE == 6) ST("sc1 nt");
      if (MODE == 7) ST("sc0 sc1 nt");
    }
}
#undef ROW
#define ROW 20000u
template <int MODE>
static void gow(uint8_t* d, uint32_t groups, uint32_t C, uint32_t off, uint32_t pitch = 20000u) {
  static const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    window<MODE><<<groups, 64>>>(d, groups, C, off, pitch);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)groups * 64 * ((ROW - 1024) / C) * C;
  printf("window C=%4u off=%2u pitch=%u %-10s : %.3f ms  %.2f TB/s\n", C, off, pitch, names[MODE], best, bytes / best * 1e-9);
}
template <int C, bool NT>
static void go(uint8_t* d, uint32_t groups, uint32_t split, uint32_t off, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    runs<C, NT><<<groups * split, 64>>>(d, groups, split, off);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)groups * 64 * ((ROW - 1024) / C) * C;
  const double instr_per_cu = bytes / 1024.0 / 256.0;
  printf("%-10s C=%4d nt=%d split=%u off=%2u : %.3f ms  %.2f TB/s  %.0f clocks(2.1GHz)/store instr/CU\n", name, C, (int)NT, split, off,
         best, bytes / best * 1e-9, best * 1e-3 * 2.1e9 / instr_per_cu);
}
int main() {
  const uint32_t groups = 1080;   // the decoder's config 3
  uint8_t* d;
  hipMalloc(&d, (size_t)groups * 64 * 20096u + 65536);
  for (uint32_t pitch : {20000u, 20032u, 20096u})
    for (uint32_t C : {64u, 128u, 160u, 256u, 320u, 640u}) {
      gow<0>(d, groups, C, 0, pitch);
      gow<1>(d, groups, C, 0, pitch);
    }
  return 0;
}
