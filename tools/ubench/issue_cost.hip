// issue_cost.hip -- cycles per instruction for ONE wave64 per SIMD (and for two), dependent chains.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip ; run on the GPU box.
// Informs the decode kernel design (x3_decode_fast_kernel runs ~1 wave per SIMD on config 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 2000

template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned long long* out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = seed * 3 + 1, c = seed ^ 0x55, d = threadIdx.x * 7 + 1;
  uint32_t e = threadIdx.x * 3 + 5, f = threadIdx.x ^ 0x33;
  __shared__ uint32_t lds[64 * 33];
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lds[threadIdx.x * 33];
  asm volatile("s_mov_b64 s[40:41], exec" ::: "s40", "s41");
  const long long t0 = clock64();
  for (int i = 0; i < ITER; ++i) {
    if (KIND == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 1) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 2) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 3) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 4) { REP16(asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));) }
    if (KIND == 5) { REP16(asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 6) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));) }
    if (KIND == 7) { REP16(asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));) }
    if (KIND == 8) { REP16(asm volatile("v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(a));) }
    if (KIND == 9) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 10) { REP16(asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(*(unsigned long long*)&a) : "v"(b));) }
    if (KIND == 11) {  // independent pair of chains
      REP16(asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
    }
    if (KIND == 12) { REP16(asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 s40, s40, 1" : "+v"(a) : "v"(b) : "s40", "scc");) }
    if (KIND == 13) { REP16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    if (KIND == 16) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[40:41]" : "+v"(a) : "v"(b));) }
    if (KIND == 17) { REP16(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 18) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 19) { REP16(asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 20) { REP16(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(a), "+v"(d));) }
    if (KIND == 21) {
      REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));)
    }
    if (KIND == 22) {
      REP16(asm volatile("v_cmp_gt_i32 vcc, 0, %0\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : : "vcc");)
    }
    if (KIND == 23) {
      REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n\tv_perm_b32 %1, %1, %4, %4\n\tv_pk_add_u16 %2, %2, %4\n\tv_perm_b32 %3, %3, %4, %4"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));)
    }
    if (KIND == 24) { REP16(asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(la));) }
    if (KIND == 25) { REP16(asm volatile("ds_write_b32 %1, %0" : : "v"(a), "v"(la));) }
    if (KIND == 26) { REP16(asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(a) : "v"(b));) }
    if (KIND == 14) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 15) { REP16(asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0) + ((unsigned long long)(a + d + c + e + f) & 1ull);
}

template <int KIND>
static void run(const char* name, int per_rep, int grid) {
  unsigned long long* d;
  (void)hipMalloc(&d, grid * 8);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 1u);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 2u);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-34s waves=%5d  cycles/instr = %.2f\n", name, grid, s / grid / ((double)ITER * 16 * per_rep));
  (void)hipFree(d);
}

int main() {
  for (int grid : {1024, 2048}) {
    run<0>("v_add_u32 (dependent)", 1, grid);
    run<1>("v_pk_add_u16", 1, grid);
    run<2>("v_perm_b32", 1, grid);
    run<3>("v_alignbit_b32", 1, grid);
    run<4>("v_ffbh_u32", 1, grid);
    run<5>("v_lshl_add_u32", 1, grid);
    run<6>("v_cndmask_b32 (vcc)", 1, grid);
    run<7>("v_lshlrev_b32", 1, grid);
    run<8>("v_pk_ashrrev_i16", 1, grid);
    run<9>("v_add3_u32", 1, grid);
    run<10>("v_lshlrev_b64", 1, grid);
    run<11>("2 independent v_add_u32", 2, grid);
    run<12>("v_add_u32 + s_add_u32", 2, grid);
    run<13>("v_cmp + v_cndmask", 2, grid);
    run<14>("v_and_b32", 1, grid);
    run<15>("v_bfe_u32", 1, grid);
    run<16>("v_cndmask_b32_e64 (sgpr pair)", 1, grid);
    run<17>("v_bfi_b32", 1, grid);
    run<18>("v_and_or_b32", 1, grid);
    run<19>("v_max_u32", 1, grid);
    run<20>("2 x v_mov_b32 (swap chain)", 2, grid);
    run<21>("4 independent v_add_u32", 4, grid);
    run<22>("v_cmp + 3 v_cndmask (vcc)", 4, grid);
    run<23>("4 independent pk_add/perm", 4, grid);
    run<24>("ds_read_b32 + wait", 1, grid);
    run<25>("ds_write_b32", 1, grid);
    run<26>("v_add_u32 + s_nop 0", 2, grid);
  }
  return 0;
}
