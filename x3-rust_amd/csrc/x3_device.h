// x3_device.h -- types and small device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define X3_WAVE 64

// Codec parameters in the form the kernels want (derived on the host from x3_params).
// Rice table geometry follows src/x3.rs:207-252: (nsubs, offset, len, inv_len) =
// (0,6,14,16) (1,11,22,26) (2,20,40,44) (3,28,56,60).
struct X3DevParams {
  uint32_t block_len;
  uint32_t blocks_per_frame;
  uint32_t spf;          // samples per frame = block_len * blocks_per_frame
  uint32_t thr[3];       // thresholds
  uint32_t k[3];         // nsubs of rice_codes[ftype]
  int32_t dmin[3];       // smallest diff the reference's table for rice_codes[ftype] can index
  int32_t dmax[3];       // largest  (outside -> the reference panics -> X3_ERR_BAD_ARG)
  uint32_t inv_len[3];   // decoder bound for rice_codes[ftype]
};

// Where the frames of a uniform batch live (single stream: n_clips = 1).
struct X3Geom {
  uint64_t n_per_clip;   // samples in each clip
  uint64_t clip_stride;  // samples between clip starts
  uint32_t fpc;          // frames per clip
  uint64_t n_frames;     // fpc * n_clips
  // x3_encode_frames_dev: frame f is the src_n[f] samples at wav + src_off[f] (a table instead of the uniform layout above;
  // n_frames entries; nullptr otherwise).  Encoders only: the decoders place frames by wav_off.
  const uint64_t* src_off = nullptr;
  const uint32_t* src_n = nullptr;
};

// status codes used on the device (values of enum x3_status in include/x3hip.h)
#define X3D_OK 0
#define X3D_IO 1   // a read past the real end of the data (X3Error::Io)
#define X3D_OUT_OF_BOUNDS_INVERSE 5
#define X3D_MORE_THAN_ONE_CHANNEL 6
#define X3D_FRAME_LENGTH 10
#define X3D_FRAME_HEADER_INVALID_KEY 11
#define X3D_FRAME_HEADER_INVALID_PAYLOAD_LEN 12
#define X3D_FRAME_HEADER_INVALID_HEADER_CRC 13
#define X3D_FRAME_HEADER_INVALID_PAYLOAD_CRC 14
#define X3D_FRAME_DECODE_INVALID_BPF 20
#define X3D_BYTE_WRITER_INSUFFICIENT_MEMORY 22
#define X3D_BAD_ARG 24

__device__ __forceinline__ uint32_t x3_bswap32(uint32_t v) { return __builtin_bswap32(v); }

// ---------------------------------------------------------------------------------------------
// CRC-16/CCITT-FALSE (src/crc.rs): polynomial 0x1021, MSB first, no reflection, no final xor.
// Table-free byte step: for t = (crc>>8) ^ byte, t ^= t>>4 ; crc' = (crc<<8) ^ (t<<12) ^ (t<<5) ^ t.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t x3_crc_byte(uint32_t crc, uint32_t byte) {
  uint32_t t = ((crc >> 8) ^ byte) & 0xFFu;
  t ^= t >> 4;
  return ((crc << 8) ^ (t << 12) ^ (t << 5) ^ t) & 0xFFFFu;
}

// four stream bytes held big-endian in `be` (first byte in bits 31..24)
__device__ __forceinline__ uint32_t x3_crc_be32(uint32_t crc, uint32_t be) {
  crc = x3_crc_byte(crc, be >> 24);
  crc = x3_crc_byte(crc, (be >> 16) & 0xFFu);
  crc = x3_crc_byte(crc, (be >> 8) & 0xFFu);
  crc = x3_crc_byte(crc, be & 0xFFu);
  return crc;
}

// a(x) * k(x) mod 0x11021 for 16-bit a, k.  k is wave-uniform in every caller, so the
// k*x^b chain runs on the scalar unit and the vector cost is 3 ops per bit.
__device__ __forceinline__ uint32_t x3_gf_mul(uint32_t a, uint32_t k) {
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    r ^= (0u - ((a >> b) & 1u)) & k;
    k = (k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u);
  }
  return r;
}

// xpow table (built on the host, x3_ctx.hip): XP[j][m] = x^(32*m*2^j) mod P, j < X3_XP_LEVELS, m <= X3_XP_M
#define X3_XP_LEVELS 10
#define X3_XP_M 128
#define X3_XINV16_INDEX (X3_XP_LEVELS * (X3_XP_M + 1))  // x^(-16) mod P stored after the table
#define X3_XP_SIZE (X3_XINV16_INDEX + 5)               // + x^(-8t), t = 1..3 (x3_decode_kernel.h)

__device__ __forceinline__ uint32_t x3_xp(const uint16_t* __restrict__ xpow, int level, uint32_t m) {
  return xpow[level * (X3_XP_M + 1) + m];
}

// ---------------------------------------------------------------------------------------------
// wave / workgroup scans
// ---------------------------------------------------------------------------------------------
// DPP form (no LDS round trips): Hillis-Steele inside each row of 16 lanes with row_shr, then
// row_bcast:15 / row_bcast:31 carry the row totals across rows (gfx9 DPP controls).
__device__ __forceinline__ uint32_t x3_wave_incl_scan_dpp(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// XOR of all 64 lanes, valid in lane 63 (same DPP ladder; a __shfl_xor butterfly costs six ds_bpermute
// round trips and ~36 VALU)
__device__ __forceinline__ uint32_t x3_wave_xor_to_lane63_dpp(uint32_t v) {
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ uint32_t x3_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < X3_WAVE; d <<= 1) {
    uint32_t t = __shfl_up(v, d, X3_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}

// packed 16-bit helpers (VOP3P: both halves of a dword at once)
__device__ __forceinline__ uint32_t x3_pk_add_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_sub_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_max_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#define X3_XINV8_INDEX(t) (X3_XINV16_INDEX + 1 + (t))   // x^(-8t), t = 1..3, behind x^(-16) in the xpow table
// stamp builds only (-DX3_DBG_STAMPS, tools/scratch/dbg_stamps_*.py): per-phase clock sums of a wave.  One copy of the
// array per translation unit; x3_dbg_read / x3_dbg_read_enc fetch the decoders' / the encoders'.
#ifdef X3_DBG_STAMPS
static __device__ unsigned long long x3_dbg[8 * 8192];
#define X3_STAMP(k) do { unsigned long long t_ = clock64(); dbg_acc[k] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#endif
// The context's pace / log words (x3_ctx::d_pace): [0..3] decoder pace by launch parity, [4..7] second-generation
// encoder pace, [8..9] the decoder's launch shape (groups) by launch parity; from X3_LOG_BASE the decoder's launch log, X3_LOG_ENTRIES entries of X3_LOG_WORDS words indexed by the
// launch epoch: {tag | slowest group's ticks per 16 blocks, tag | target, group 0's shader ticks, its 10 ns ticks};
// behind it the wave encoder's: {tag | 0, 0, workgroup 0's shader ticks, its 10 ns ticks}.
#define X3_LOG_BASE 16u
#define X3_LOG_ENTRIES 256u
#define X3_LOG_WORDS 4u
#define X3_LOG_ENC_BASE (X3_LOG_BASE + X3_LOG_ENTRIES * X3_LOG_WORDS)
#define X3_PACE_WORDS (X3_LOG_ENC_BASE + X3_LOG_ENTRIES * X3_LOG_WORDS)
// packed 16-bit logical shift right, per-half amounts in sh2
__device__ __forceinline__ uint32_t x3_pk_lshr_b16(uint32_t a, uint32_t sh2) {
  uint32_t r;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh2), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_lshr_b16_1(uint32_t a) {
  uint32_t r;
  asm("v_pk_lshrrev_b16 %0, 1, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_ashr_i16_15(uint32_t a) {
  uint32_t r;
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
// (a.lo + b.hi, a.hi + b.hi): add the high half of b to both halves of a
__device__ __forceinline__ uint32_t x3_pk_add_u16_bhi(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a.lo * k.lo + c.lo, a.lo * k.hi + c.hi): the low half of a times each half of k, plus c
__device__ __forceinline__ uint32_t x3_pk_mad_u16_alo(uint32_t a, uint32_t k, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(k), "v"(c));
  return r;
}
// (a & 0xFFFF) | (b << 16)
__device__ __forceinline__ uint32_t x3_pack_lo16(uint32_t a, uint32_t b) {
  return __builtin_amdgcn_perm(b, a, 0x05040100u);
}
// (m & a) | (~m & b)
__device__ __forceinline__ uint32_t x3_bfi(uint32_t m, uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}
// (a & b) | c
__device__ __forceinline__ uint32_t x3_and_or(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// a * b + c on the signed low 24 bits of a and b (full rate)
__device__ __forceinline__ uint32_t x3_mad_i24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// one dword from an LDS byte address
__device__ __forceinline__ uint32_t x3_lds_read_b32(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(addr);
}
// LDS byte address of an object in LDS
__device__ __forceinline__ uint32_t x3_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)p;
}
// byte k of a dword, times two (a uint16 table offset): one SDWA shift (the shift count from a scalar register: a
// vector register holding the constant 1 for the life of a kernel is one register too many in the tightest ones)
__device__ __forceinline__ uint32_t x3_sdwa_byte_x2(uint32_t v, int k) {
  uint32_t r;
  switch (k) {
    case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(1u), "v"(v)); break;
    case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(1u), "v"(v)); break;
    case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(1u), "v"(v)); break;
    default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(1u), "v"(v)); break;
  }
  return r;
}
// a << 1 per half, the count as an inline constant
__device__ __forceinline__ uint32_t x3_pk_shl_b16_1(uint32_t a) {
  uint32_t r;
  asm("v_pk_lshlrev_b16 %0, 1, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
// one uint16 from an LDS byte address + constant
__device__ __forceinline__ uint32_t x3_lds_read_u16(uint32_t addr, uint32_t const_off) {
  return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>(addr + const_off);
}
// sixteen bytes to a 16-byte aligned LDS byte address
typedef uint32_t x3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x3_lds_write_b128(uint32_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  x3_u32x4 v = {a, b, c, d};
  *reinterpret_cast<__attribute__((address_space(3))) x3_u32x4*>(addr) = v;
}
__device__ __forceinline__ x3_u32x4 x3_lds_read_b128(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) x3_u32x4*>(addr);
}
typedef uint32_t x3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_lds_write_b64(uint32_t addr, uint32_t a, uint32_t b) {
  x3_u32x2 v = {a, b};
  *reinterpret_cast<__attribute__((address_space(3))) x3_u32x2*>(addr) = v;
}
__device__ __forceinline__ void x3_lds_write_u16(uint32_t addr, uint32_t v) {
  *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>(addr) = (uint16_t)v;
}
// sixteen bytes to global memory as a streaming (non-temporal) store: output that is written once and not read
// again by this kernel must not displace what the kernel is still reading from L2
__device__ __forceinline__ x3_u32x2 x3_lds_read_b64(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) x3_u32x2*>(addr);
}
__device__ __forceinline__ void x3_store_stream8(void* p, x3_u32x2 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<x3_u32x2*>(p));
}
__device__ __forceinline__ void x3_store_stream16(void* p, x3_u32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<x3_u32x4*>(p));
}
// v_ffbh_u32 without __clz's clamp: -1 (not 32) for 0
__device__ __forceinline__ uint32_t x3_ffbh(uint32_t a) {
  uint32_t r;
  asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
// all ones when j == 0 (j >= 0), else 0 -- two VALU, no compare
__device__ __forceinline__ uint32_t x3_mask_if_zero(int32_t j) {
  uint32_t r;
  asm("v_add_u32 %0, -1, %1\n\tv_ashrrev_i32 %0, 31, %0" : "=&v"(r) : "v"(j));
  return r;
}
