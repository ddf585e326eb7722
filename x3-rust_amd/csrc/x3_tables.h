// x3_tables.h -- sizes and layouts of the constant tables the kernels take from the context (built on the host at context
// creation, x3_ctx.hip), and the two small result structs that cross the host / device border.  Included by the kernel
// headers that use them and by x3_ctx.hip, which includes no kernel header of another unit (x3_internal.h).
#pragma once
#include <cstdint>

// ---- x3_encode_stream2_kernel.h: CRC multipliers per chunk size (see there for KN / KA)
#ifndef X3E_CRC_WAVES
#define X3E_CRC_WAVES 8u
#endif
#define X3E_CRC_LANES (64u * X3E_CRC_WAVES)
// the longest payload of this path: 512 blocks of 20 literals, 20 828 bytes
#define X3_STREAM2_MAX_PAYLOAD_DWORDS 5248u
#define X3_K2_MAXC ((X3_STREAM2_MAX_PAYLOAD_DWORDS + X3E_CRC_LANES - 1u) / X3E_CRC_LANES)
#define X3_K2_ROW 33u
#define X3_K2_KA (64u * X3_K2_ROW)
#define X3_K2_DWORDS (X3_K2_KA + 8u * 16u)

// ---- x3_encode_wave_kernel.h: M0..M3 "byte k of a 32-bit state times x^4096", T4/T5 "16-bit state times x^2048"
// (6 x 256 x u16), lane weights 64 x 16 x u16, x^(-16k) k < 128
#define X3W_TAB_BYTES 5376u

// ---- x3_frame_check_kernel (x3_decode_kernel.h): twelve rows of 256 uint16 (T0, M2, M4) and x^(-8k), k < 1024
// X3_CHECK_STEP_ASM (round 5): the check kernel's two-row step as one asm block (x3_decode_kernel.h).
// (Tried and dropped: the rows of bytes 1 and 3 loaded into the HIGH half of the register that holds the look-up of byte
// 0 / 2, ds_read_u16_d16_hi, with those rows holding x^16 less -- four registers to XOR instead of eight.  The arithmetic is
// right and the instruction exists, but with SRAM ECC -- gfx950 -- a d16 load does not preserve the other half: payload CRC
// errors on every frame.)
// Measured (profiles/r5/check_kernel_step_asm.txt): 12 instead of 18 vector instructions per step and the kernel is no
// faster -- white noise 1.02 against 1.03 ms beside the decoder, config 3 0.666 against 0.614 (80 registers and a spill
// instead of 92): it is not bound by what it issues.  Off.
#ifndef X3_CHECK_STEP_ASM
#define X3_CHECK_STEP_ASM 0
#endif
#define X3_CHECK_TAB_U16 (12u * 256u)
#define X3_CHECK_TAB_DW (X3_CHECK_TAB_U16 / 2u)
#define X3_CHECK_XINV_N 1024u  // x^(-8k), k < 1024: undoes the zero bytes the row grid adds behind a payload

#define X3_MAX_CHANNELS 8u   // the multi-channel extension (x3_mc.h, x3_decode_replay.h)

// ---- what x3_decode_result / x3_index_dev read back
struct X3DecodeSummary {
  unsigned long long first_bad;
  unsigned long long samples_before;  // valid when first_bad == n_frames (else see x3_decode_prefix_kernel)
  int first_bad_status;
  int pad;
};
struct X3IndexSummary {
  unsigned long long n_frames;
  unsigned long long n_samples;
  int terminal;
  uint32_t last_node;   // scratch: candidate index of the last frame of the chain
  unsigned long long first_over;  // scratch: first frame that does not fit wav_cap
  unsigned long long n_chain;     // scratch: frames reachable from the start node
  uint32_t start;                 // scratch: candidate at offset 0 (X3I_NONE: the walk cannot step onto it)
  uint32_t pad;                   // 1: more frames than the caller's arrays hold
  uint32_t unaligned;             // 1: some frame's sample offset is not a multiple of four: rows off the 8-byte grid (picks the decoder kernel)
  uint32_t pad2;
};
