// valu_rate.hip -- what a VALU instruction costs a SATURATED SIMD on gfx950, by kind: W single-wave workgroups per
// SIMD, every wave runs four independent chains of the instruction under test; reported as ns of SIMD time per
// instruction and relative to v_add_u32.  (tools/ubench/issue_cost.hip measures the other end: what ONE wave can issue.)
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2000
#define R4(x) x x x x

#define CHAIN4(ins)                                                                                  \
  asm volatile(ins "\n" : "+v"(a0), "+v"(b0) : "v"(c) : "vcc");                                        \
  asm volatile(ins "\n" : "+v"(a1), "+v"(b1) : "v"(c) : "vcc");                                        \
  asm volatile(ins "\n" : "+v"(a2), "+v"(b2) : "v"(c) : "vcc");                                        \
  asm volatile(ins "\n" : "+v"(a3), "+v"(b3) : "v"(c) : "vcc");

template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned* out, unsigned seed) {
  unsigned a0 = threadIdx.x * seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  unsigned b0 = seed + 5, b1 = seed + 6, b2 = seed + 7, b3 = seed + 8;
  unsigned c = (seed & 3) + 1;
  unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;
  for (int i = 0; i < ITER; ++i) {
    if (KIND == 0) { R4(CHAIN4("v_add_u32 %0, %0, %2")) }
    if (KIND == 1) { R4(CHAIN4("v_xor_b32 %0, %0, %2")) }
    if (KIND == 2) { R4(CHAIN4("v_pk_add_u16 %0, %0, %2")) }
    if (KIND == 3) { R4(CHAIN4("v_pk_lshrrev_b16 %0, %2, %0")) }
    if (KIND == 4) { R4(CHAIN4("v_pk_min_i16 %0, %0, %2")) }
    if (KIND == 5) { R4(CHAIN4("v_pk_sub_i16 %0, %0, %2 clamp")) }
    if (KIND == 6) { R4(CHAIN4("v_add_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")) }
    if (KIND == 7) { R4(CHAIN4("v_lshlrev_b32_sdwa %0, %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")) }
    if (KIND == 8) { R4(CHAIN4("v_or_b32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")) }
    if (KIND == 9) {
      R4(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3"
                      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c));)
    }
    if (KIND == 10) {
      R4(asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3"
                      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c));)
    }
    if (KIND == 11) { R4(CHAIN4("v_perm_b32 %0, %0, %1, %2")) }
    if (KIND == 12) { R4(CHAIN4("v_alignbit_b32 %0, %0, %1, %2")) }
    if (KIND == 13) { R4(CHAIN4("v_and_or_b32 %0, %0, %1, %2")) }
    if (KIND == 14) { R4(CHAIN4("v_cmp_lt_u32 vcc, %0, %2")) }
    if (KIND == 15) { R4(CHAIN4("v_readlane_b32 s40, %0, 3")) }
    if (KIND == 16) { R4(CHAIN4("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")) }
    if (KIND == 17) { R4(CHAIN4("v_bfe_i32 %0, %0, %2, 16")) }
    if (KIND == 18) { R4(CHAIN4("v_mov_b32 %0, %1")) }
    if (KIND == 19) { R4(CHAIN4("v_nop")) }
    if (KIND == 20) { R4(CHAIN4("v_mad_u32_u24 %0, %0, %2, %1")) }
    if (KIND == 21) { R4(CHAIN4("v_lshl_add_u32 %0, %0, %2, %1")) }
    if (KIND == 22) { R4(CHAIN4("v_cndmask_b32_e64 %0, %0, %1, s[42:43]")) }
    if (KIND == 23) { R4(CHAIN4("v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc")) }
    if (KIND == 24) { R4(CHAIN4("s_nop 0")) }
    if (KIND == 25) { R4(CHAIN4("v_bfi_b32 %0, %2, %0, %1")) }
    if (KIND == 26) { R4(CHAIN4("v_mul_lo_u32 %0, %0, %2")) }
    if (KIND == 27) { R4(CHAIN4("v_pk_lshlrev_b16 %0, %2, %0")) }
    if (KIND == 28) { R4(CHAIN4("v_xor_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")) }
    if (KIND == 29) { R4(CHAIN4("v_add_u32 %0, %0, %2\n s_add_u32 s44, s44, 1")) }
  }
  if (((a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3) + (unsigned)(q0 + q1 + q2 + q3)) == 0x1234567u) out[0] = 1;
}

static double base_ns[16];

template <int KIND>
static void run(const char* name, int per_rep, int w, int wi) {
  unsigned* d;
  (void)hipMalloc(&d, 64);
  const int grid = 1024 * w;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 3u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, d, 5u);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)ITER * 16 * per_rep * w;
  const double ns = ms * 1e6 / instr_per_simd;
  if (KIND == 0) base_ns[wi] = ns;
  printf("%-34s waves/SIMD %d  %.3f ns per instruction and SIMD  (x%.2f of v_add_u32)\n", name, w, ns, ns / base_ns[wi]);
  (void)hipFree(d);
}

int main() {
  int wi = 0;
  for (int w : {1, 2, 6}) {
    run<0>("v_add_u32", 1, w, wi);
    run<1>("v_xor_b32", 1, w, wi);
    run<18>("v_mov_b32", 1, w, wi);
    run<19>("v_nop", 1, w, wi);
    run<24>("s_nop 0", 1, w, wi);
    run<29>("v_add_u32 + s_add_u32 (per pair)", 1, w, wi);
    run<2>("v_pk_add_u16", 1, w, wi);
    run<3>("v_pk_lshrrev_b16", 1, w, wi);
    run<27>("v_pk_lshlrev_b16", 1, w, wi);
    run<4>("v_pk_min_i16", 1, w, wi);
    run<5>("v_pk_sub_i16 clamp", 1, w, wi);
    run<6>("v_add_u32_sdwa", 1, w, wi);
    run<7>("v_lshlrev_b32_sdwa", 1, w, wi);
    run<8>("v_or_b32_sdwa", 1, w, wi);
    run<9>("v_lshlrev_b64", 1, w, wi);
    run<10>("v_lshrrev_b64", 1, w, wi);
    run<11>("v_perm_b32", 1, w, wi);
    run<12>("v_alignbit_b32", 1, w, wi);
    run<13>("v_and_or_b32", 1, w, wi);
    run<25>("v_bfi_b32", 1, w, wi);
    run<21>("v_lshl_add_u32", 1, w, wi);
    run<20>("v_mad_u32_u24", 1, w, wi);
    run<26>("v_mul_lo_u32", 1, w, wi);
    run<17>("v_bfe_i32", 1, w, wi);
    run<14>("v_cmp_lt_u32 vcc", 1, w, wi);
    run<22>("v_cndmask_b32_e64 (sgpr pair)", 1, w, wi);
    run<23>("v_cmp + v_cndmask vcc (per pair)", 1, w, wi);
    run<15>("v_readlane_b32", 1, w, wi);
    run<16>("v_add_u32_dpp row_shr:1", 1, w, wi);
    run<28>("v_xor_b32_dpp row_shr:1", 1, w, wi);
    ++wi;
  }
  return 0;
}
