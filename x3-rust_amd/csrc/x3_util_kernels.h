// x3_util_kernels.h -- frame-offset scan, buffer CRC-16 (segmented reduction), synthetic input.
#pragma once
#include "x3_device.h"
#include "x3_synth_core.h"

// (x3_scan_frame_offsets_kernel: x3_encode_kernel.h, with the two-pass encoder it serves)

// ---------------------------------------------------------------------------------------------
// CRC-16 of an arbitrary device buffer as a segmented reduction (crc::crc16, src/crc.rs:49-58).
// The 4-byte-aligned prefix is cut into right-aligned segments of X3_CRC_SEG_DW dwords, one wave
// per segment, 128 dwords per lane; partial CRCs (init 0) are combined with x^(8*len) mod P.
// ---------------------------------------------------------------------------------------------
#define X3_CRC_LANE_DW 128u
#define X3_CRC_SEG_DW (X3_CRC_LANE_DW * 64u)

__global__ void __launch_bounds__(64)
x3_crc_segments_kernel(const uint32_t* __restrict__ data, uint64_t n_dw, uint64_t n_seg,
                       const uint16_t* __restrict__ xpow, uint16_t* __restrict__ seg_crc) {
  const uint64_t seg = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const int64_t shift = (int64_t)(n_seg * X3_CRC_SEG_DW - n_dw);  // virtual leading zero dwords
  const int64_t j0 = (int64_t)(seg * X3_CRC_SEG_DW + (uint64_t)lane * X3_CRC_LANE_DW) - shift;
  uint32_t crc = 0;
  for (uint32_t i = 0; i < X3_CRC_LANE_DW; ++i) {
    const int64_t j = j0 + i;
    if (j >= 0) {
      uint32_t be = x3_bswap32(data[j]);
      if (j == 0) be ^= 0xFFFF0000u;
      crc = x3_crc_be32(crc, be);
    }
  }
#pragma unroll
  for (int lvl = 0; lvl < 6; ++lvl) {
    const uint32_t kx = x3_xp(xpow, lvl, X3_CRC_LANE_DW);
    const uint32_t t = __shfl_up(crc, 1 << lvl, X3_WAVE);
    if (lane >= (1u << lvl)) crc = x3_gf_mul(t, kx) ^ crc;
  }
  if (lane == 63) seg_crc[seg] = (uint16_t)crc;
}

// Horner over the segment CRCs, then the <= 3 tail bytes and the short-buffer cases.
__global__ void x3_crc_combine_kernel(const uint8_t* __restrict__ bytes, uint64_t n, uint64_t n_seg,
                                      const uint16_t* __restrict__ seg_crc, const uint16_t* __restrict__ xpow,
                                      uint16_t* __restrict__ result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint64_t n4 = n & ~3ull;
  uint32_t crc;
  if (n4 == 0) {
    crc = 0xFFFFu;  // nothing went through the segment pass
  } else {
    const uint32_t kseg = x3_xp(xpow, 6, X3_CRC_LANE_DW);  // x^(32 * 128 * 64)
    crc = 0;
    for (uint64_t s = 0; s < n_seg; ++s) crc = x3_gf_mul(crc, kseg) ^ seg_crc[s];
  }
  for (uint64_t i = n4; i < n; ++i) crc = x3_crc_byte(crc, bytes[i]);
  *result = (uint16_t)crc;
}

// ---------------------------------------------------------------------------------------------
// synthetic input: one thread per X3_SYNTH_SEG-sample segment
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
x3_synth_kernel(int kind, uint64_t seed, uint64_t start, uint64_t n, int16_t* __restrict__ out) {
  const uint64_t first_seg = start / X3_SYNTH_SEG;
  const uint64_t seg = first_seg + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t seg_lo = seg * X3_SYNTH_SEG, seg_hi = seg_lo + X3_SYNTH_SEG;
  const uint64_t end = start + n;
  if (seg_lo >= end) return;
  const uint32_t lo = start > seg_lo ? (uint32_t)(start - seg_lo) : 0u;
  const uint32_t hi = end < seg_hi ? (uint32_t)(end - seg_lo) : X3_SYNTH_SEG;
  int16_t* o = out + (seg_lo + lo - start);
  x3_synth_segment(kind, seed, seg, lo, hi, [o](uint32_t i, int16_t v) { o[i] = v; });
}
