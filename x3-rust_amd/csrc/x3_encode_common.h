// x3_encode_common.h -- what the single-pass encoders share: packed 16-bit and SDWA helpers, the LDS OR, compile-time
// GF(2) constants of the CRC-16 polynomial, the {epoch:12 | bytes:20} size-word format and the bounded-wait protocol
// constants.  (Until round 4 these lived in the first-generation kernel's header, x3_encode_stream_kernel.h; that kernel
// -- nine waves per frame, a sample tile in LDS -- was superseded in round 2 and is gone: its arithmetic, steps B-F, is
// described in x3_encode_stream2_kernel.h, which took it over.)
//
// The (code, len) recipe of the emission loops, shared by both kernels: per lane a packed source (zigzag / difference /
// raw), an AND mask, an OR constant, a shift and a base length turn a sample into (code, len) -- for both samples of a
// pair at once in packed 16-bit arithmetic, the halves recombined with SDWA operand selects.
#pragma once
#include "x3_encode_kernel.h"


typedef short x3_short2 __attribute__((ext_vector_type(2)));
typedef unsigned short x3_ushort2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t x3_pk_sub_sat(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_min_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_max_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_shl_b16(uint32_t a, uint32_t sh) {  // a << sh, per half
  uint32_t r;
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_shr_u16(uint32_t a, uint32_t sh) {  // logical
  uint32_t r;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_sar_i16(uint32_t a, uint32_t sh) {  // arithmetic
  uint32_t r;
  asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}

// SDWA operand selects (gfx9 encoding, available on gfx950): halves of a dword as operands of a 32-bit op
__device__ __forceinline__ uint32_t x3_sdwa_add_w0_w1(uint32_t a) {  // a.lo16 + a.hi16
  uint32_t r;
  asm("v_add_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_sdwa_shl_w0_by_w1(uint32_t v, uint32_t sh) {  // v.lo16 << sh.hi16
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"
      : "=v"(r) : "v"(sh), "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t x3_sdwa_or_w1(uint32_t a, uint32_t b) {  // a | b.hi16
  uint32_t r;
  asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
      : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// ds_or_b32 on an LDS byte address (no return value)
__device__ __forceinline__ void x3_lds_or_b32(uint32_t addr, uint32_t v) {
  __hip_atomic_fetch_or(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)addr), v,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// vmcnt(0) as the BUILTIN, not as asm text: hipcc's wait-count pass sees it and clears its scoreboard, so it
// does not add conservative vmcnt(0) waits of its own later (those would also wait for the hidden LDS-DMA).
__device__ __forceinline__ void x3_dma_wait() {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  asm volatile("" ::: "memory");
}

// a(x) * C(x) mod 0x11021 for a compile-time constant C: the sixteen C*x^b are immediates
constexpr uint32_t x3_gf_xtime(uint32_t k) { return ((k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u)) & 0xFFFFu; }
constexpr uint32_t x3_gf_mul_c(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 0; i < 16; ++i) {
    if ((a >> i) & 1u) r ^= b;
    b = x3_gf_xtime(b);
  }
  return r;
}
constexpr uint32_t x3_gf_pow_c(uint32_t base, int e) {
  uint32_t r = 1;
  for (int i = 0; i < e; ++i) r = x3_gf_mul_c(r, base);
  return r;
}
constexpr uint32_t x3_crc16_step_c(uint32_t crc, uint32_t byte) {
  uint32_t t = ((crc >> 8) ^ byte) & 0xFFu;
  t ^= t >> 4;
  return ((crc << 8) ^ (t << 12) ^ (t << 5) ^ t) & 0xFFFFu;
}
constexpr uint32_t x3_crc16_const4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
  return x3_crc16_step_c(x3_crc16_step_c(x3_crc16_step_c(x3_crc16_step_c(0xFFFFu, b0), b1), b2), b3);
}
// x^-1 = x^15 + x^11 + x^4 (x * that = x^16 + x^12 + x^5 = P + 1); x^-16 = (x^-1)^16
constexpr uint32_t X3_XINV16_C = x3_gf_pow_c(0x8810u, 16);
template <uint32_t C>
__device__ __forceinline__ uint32_t x3_gf_mul_const(uint32_t a) {
  uint32_t r = 0, k = C;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    r ^= (0u - ((a >> b) & 1u)) & k;
    k = x3_gf_xtime(k);
  }
  return r;
}

#define X3_DESC_BYTES_BITS 20u                       // frame bytes <= 20 + 65535 < 2^20
#define X3_DESC_BYTES_MASK ((1u << X3_DESC_BYTES_BITS) - 1u)
