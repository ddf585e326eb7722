// x3_decode_kernel.h -- frame decoder for gfx950.
//
// Decoding a frame is sequential (variable-length codes; a block's start is known only once the
// previous block is decoded) but frames are independent (each re-seeds the predictor with a raw
// sample, decoder.rs:42-46), so the parallel axis is the frame.  Two kernels:
//
//  x3_frame_check_kernel  -- one WAVE per frame.  decoder::read_frame_header (decoder.rs:69-118),
//      the walk's length checks (decodefile.rs:107-121) and the payload CRC
//      (X3aReader::read_frame_payload, decodefile.rs:93-103).  The CRC is a segmented reduction:
//      every lane CRCs a right-aligned chunk of payload dwords with init 0 (the 0xFFFF init is
//      folded into the first two payload bytes), then a 6-step wavefront tree multiplies by
//      x^(8*len) mod 0x11021.  Writes status[f] and meta[f] = {payload_len, samples}.
//
//  x3_decode_lanes_kernel -- one frame per LANE, 64 consecutive frames per wave, all lanes in
//      lock-step on the sample index.  decode_frame / decode_block (decoder.rs:36-58,132-235)
//      with BitReader (bitreader.rs:29-176) replaced by a per-lane 64-bit MSB-first window fed
//      from a per-lane LDS ring.  HBM traffic is kept wide on both sides:
//        in : every lane streams its payload with 16-byte global loads issued on a wave-uniform
//             schedule one chunk (<= 20 samples) ahead of use and parked in its LDS ring;
//        out: samples are staged in LDS as [lane][sample] and flushed every 160 samples by the
//             whole wave as 16-byte stores of contiguous 320-byte runs per frame.
//
// Bit-window semantics: the payload is an MSB-first bit string that is zero beyond its last byte
// (bitreader.rs:34-48,157-161).  Zero runs are counted exactly; the reference caps a run at the
// end of the next 32-bit word (bitreader.rs:129-139), which only differs for runs >= 32 bits -- an
// OutOfBoundsInverse error either way with the default parameters ("Known divergences").
#pragma once
#include <type_traits>

#include "x3_device.h"
#include "x3_decode_replay.h"

#define X3D_STREAM_ENDS_IN_FRAME (-1)  // quiet stop of the walk (decodefile.rs:107-116)

struct X3FrameMeta {
  uint32_t payload_len;
  uint32_t samples;
};

// 4 stream bytes at byte offset `o` of the 4-byte-aligned buffer xw, as a big-endian value;
// dwords at or beyond n_dw read as zero
__device__ __forceinline__ uint32_t x3_be32_at(const uint32_t* __restrict__ xw, uint64_t n_dw, uint64_t o) {
  const uint64_t j = o >> 2;
  const uint32_t sh = (uint32_t)(o & 3u) * 8u;
  const uint32_t a = j < n_dw ? x3_bswap32(xw[j]) : 0u;
  if (sh == 0) return a;
  const uint32_t b = (j + 1) < n_dw ? x3_bswap32(xw[j + 1]) : 0u;
  return (a << sh) | (b >> (32u - sh));
}

// x^(-8t) mod P for t = 1..3 lives behind the x^n table
// (X3_XINV8_INDEX: x3_device.h)

// decoder::read_frame_header (decoder.rs:69-118) + the walk's length checks (decodefile.rs:107-121) for
// the frame at byte offset `off`; same check order as the reference.
// ... on the five big-endian words of the header
// hc = CRC-16 of the first 16 header bytes, computed by the caller (table-free or from LDS tables)
__device__ __forceinline__ int32_t x3_frame_header_check_words(uint32_t h0, uint32_t h1, uint32_t h4, uint32_t hc,
                                                               uint64_t x3_len, uint64_t off, uint32_t& plen,
                                                               uint32_t& samples, uint32_t& pcrc, uint32_t n_ch = 1u) {
  samples = h1 >> 16;
  plen = h1 & 0xFFFFu;
  pcrc = h4 & 0xFFFFu;
  if ((h4 >> 16) != hc) return X3D_FRAME_HEADER_INVALID_HEADER_CRC;
  if ((h0 >> 16) != 0x7833u) return X3D_FRAME_HEADER_INVALID_KEY;
  // (n_ch > 1: the multi-channel extension -- the frame must say exactly n_ch; else the reference's test)
  if (n_ch == 1u ? (h0 & 0xFFu) > 1u : (h0 & 0xFFu) != n_ch) return X3D_MORE_THAN_ONE_CHANNEL;
  if (plen >= 0x7fe0u) return X3D_FRAME_LENGTH;
  if (off + 20 + plen > x3_len) return X3D_STREAM_ENDS_IN_FRAME;   // decodefile.rs:114-116
  if (plen > 24576u) return X3D_FRAME_HEADER_INVALID_PAYLOAD_LEN;  // decodefile.rs:118-121
  return X3D_OK;
}

__device__ __forceinline__ int32_t x3_frame_header_check(const uint32_t* __restrict__ xw, uint64_t n_dw,
                                                         uint64_t x3_len, uint64_t off, uint32_t& plen,
                                                         uint32_t& samples, uint32_t& pcrc) {
  plen = 0;
  samples = 0;
  pcrc = 0;
  if (off + 20 > x3_len) return X3D_STREAM_ENDS_IN_FRAME;
  const uint32_t h0 = x3_be32_at(xw, n_dw, off), h1 = x3_be32_at(xw, n_dw, off + 4);
  const uint32_t h2 = x3_be32_at(xw, n_dw, off + 8), h3 = x3_be32_at(xw, n_dw, off + 12);
  const uint32_t h4 = x3_be32_at(xw, n_dw, off + 16);
  uint32_t hc = 0xFFFFu;
  hc = x3_crc_be32(hc, h0);
  hc = x3_crc_be32(hc, h1);
  hc = x3_crc_be32(hc, h2);
  hc = x3_crc_be32(hc, h3);
  return x3_frame_header_check_words(h0, h1, h4, hc, x3_len, off, plen, samples, pcrc);
}

// Payload-CRC tables in LDS: T[0..3][v] = crc0 of byte v followed by 0..3 zero bytes (slicing by 4),
// T[4][v] = (v << 8) * x^2048, T[5][v] = v * x^2048 (a 16-bit state times x^(32*64) is T[4][hi] ^ T[5][lo]).
// LDS tables of x3_frame_check_kernel (uint16): T[s][k][v] = v * x^(8k + 16) * x^(2048 s), s, k = 0..3 -- the
// contribution of the byte that k bytes follow in its dword, in a dword that s rows of 64 dwords follow --, then
// the two rows of "times x^8192" (for v << 8 and for v)
#ifndef X3_CHECK_SCHED_FENCE
#define X3_CHECK_SCHED_FENCE 1
#endif
#include "x3_tables.h"   // X3_CHECK_STEP_ASM, X3_CHECK_TAB_U16 / _DW, X3_CHECK_XINV_N
#ifndef X3_CHECK_MIN_WGS
// (workgroups per CU the register allocator plans for: with the step as an asm block it took 151 registers at 3 -- nothing
// holds the allocator back below the bound -- against 92 before; 6 = 80 registers, three of its waves fit a SIMD beside the
// decoder's as before)
#define X3_CHECK_MIN_WGS (X3_CHECK_STEP_ASM ? 6 : 3)
#endif
// LDS tables (uint16, twelve rows of 256): T0[k][v] = v * x^(8k + 16) (crc0 of byte k of a big-endian dword: the header
// CRC and the final reduction), M2[k][v] = v * x^(8k + 2048) and M4[k][v] = v * x^(8k + 4096) (a dword one / two rows of 64
// dwords further from the end): 6 KB (round 2: sixteen row tables + two, 9 KB)

// One WAVE per frame, waves walk the frames grid-stride (the tables are loaded once per workgroup).
// Lane t takes the payload dwords t, t + 64, t + 128, ... (every load is one contiguous 256-byte run of the
// wave), folds them Horner-style with x^2048 between consecutive ones, and multiplies its partial by
// x^(32*m), m = the number of dwords behind its last one (0..63; sixteen pre-shifted words per m in LDS); the
// products are XOR-reduced.  The kernel is LATENCY-bound per wave (frame offset -> header -> payload are
// dependent loads), so everything a frame needs is requested one frame ahead: its offset two frames ahead,
// its header words and -- speculatively, the mapping does not depend on the payload length -- its first
// X3_CHECK_AHEAD x 64 payload dwords one frame ahead.
#ifndef X3_CHECK_SETPRIO
#define X3_CHECK_SETPRIO 1   // (round 3, with the leaner kernel: 3 -> it is done in 0.38 ms and the decoder needs 0.73; 1 with 4 workgroups per CU: 0.54 beside a 0.65 ms decoder; 0: 0.53 / 0.65 with 8)
#endif
#ifndef X3_CHECK_AHEAD
#define X3_CHECK_AHEAD 24u  // dwords per lane requested ahead: 6 KB of payload (a default frame is ~5.3 KB)
#endif

__global__ void __launch_bounds__(256, X3_CHECK_MIN_WGS)
x3_frame_check_kernel(const uint32_t* __restrict__ xw, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                      uint64_t n_frames_arg, const uint16_t* __restrict__ xinv8, const uint16_t* __restrict__ tab_g,
                      const uint32_t* __restrict__ kx64, int32_t* __restrict__ status,
                      unsigned long long* __restrict__ summary, uint32_t n_ch,
                      const unsigned long long* __restrict__ d_nf = nullptr) {
  const uint64_t n_frames = d_nf ? (*d_nf < n_frames_arg ? (uint64_t)*d_nf : n_frames_arg) : n_frames_arg;   // (the count from device memory: x3_decode_split_kernel.h)
  __shared__ __attribute__((aligned(16))) uint16_t tab[X3_CHECK_TAB_U16];
  // short and latency-bound: ahead of the decoder's waves it shares the SIMDs with, so that it is out of the way
  // early instead of being stretched to the decoder's whole duration
  __builtin_amdgcn_s_setprio(X3_CHECK_SETPRIO);
  // the summary x3_decode_merge_kernel reduces into starts as {first_bad = n_frames, samples_before = 0,
  // status 0} (X3DecodeSummary, 24 bytes); this kernel is joined in front of the merge, so it can set that up
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    summary[0] = n_frames;
    summary[1] = 0;
    summary[2] = 0;
  }
  for (uint32_t i = threadIdx.x; i < X3_CHECK_TAB_DW; i += blockDim.x)
    reinterpret_cast<uint32_t*>(tab)[i] = reinterpret_cast<const uint32_t*>(tab_g)[i];
  const uint32_t lane = threadIdx.x & 63u;
  __syncthreads();
  const uint32_t tab_base = x3_lds_addr(tab);
  // sum over the four bytes of a big-endian dword of T[s][k][byte]: the byte selects are SDWA operands of the
  // shift that turns the byte into a table offset, the row of the table is the immediate offset of the read
  auto rowsum = [&](uint32_t be, uint32_t s_row) -> uint32_t {  // s_row: compile-time after unrolling
    const uint32_t o = s_row * 2048u;
    const uint32_t a3 = x3_sdwa_byte_x2(be, 3), a2 = x3_sdwa_byte_x2(be, 2), a1 = x3_sdwa_byte_x2(be, 1),
                   a0 = x3_sdwa_byte_x2(be, 0);
    return x3_lds_read_u16(tab_base + a3, o + 1536u) ^ x3_lds_read_u16(tab_base + a2, o + 1024u) ^
           x3_lds_read_u16(tab_base + a1, o + 512u) ^ x3_lds_read_u16(tab_base + a0, o);
  };
  auto crc0_be32 = [&](uint32_t m) -> uint32_t { return rowsum(m, 0u); };  // crc0 of four bytes held big-endian
  // a 32-bit state (big-endian polynomial) times x^4096 mod P: byte k in the table of its weight
  auto m4096 = [&](uint32_t v) -> uint32_t {
    const uint32_t a3 = x3_sdwa_byte_x2(v, 3), a2 = x3_sdwa_byte_x2(v, 2), a1 = x3_sdwa_byte_x2(v, 1), a0 = x3_sdwa_byte_x2(v, 0);
    return x3_lds_read_u16(tab_base + a3, 4096u + 1536u) ^ x3_lds_read_u16(tab_base + a2, 4096u + 1024u) ^
           x3_lds_read_u16(tab_base + a1, 4096u + 512u) ^ x3_lds_read_u16(tab_base + a0, 4096u);
  };
  // a dword AS LOADED (little-endian: stream byte j is byte j) times x^2048 mod P: stream byte j has the weight of byte
  // 3 - j of the big-endian value -- the byte swap is in the choice of the table
  auto m2048_le = [&](uint32_t raw) -> uint32_t {
    const uint32_t a3 = x3_sdwa_byte_x2(raw, 3), a2 = x3_sdwa_byte_x2(raw, 2), a1 = x3_sdwa_byte_x2(raw, 1), a0 = x3_sdwa_byte_x2(raw, 0);
    return x3_lds_read_u16(tab_base + a0, 2048u + 1536u) ^ x3_lds_read_u16(tab_base + a1, 2048u + 1024u) ^
           x3_lds_read_u16(tab_base + a2, 2048u + 512u) ^ x3_lds_read_u16(tab_base + a3, 2048u);
  };
#if X3_CHECK_STEP_ASM
  // A two-row step in one asm block: A' = A * x^4096 ^ d0 * x^2048 ^ d1sw (d0 as loaded, d1sw byte-swapped): eight shifts
  // that turn a byte into a table offset, eight look-ups, four three-input XORs on whole registers (the look-ups come back
  // zero-extended) -- 12 vector instructions where the compiler's version has 14 and an s_nop or two.  (The block waits for
  // its own reads: the compiler does not count LDS operations issued by an asm.)  `tab` is the kernel's only LDS variable, at
  // address 0: the rows are immediate offsets.  A look-up's destination is its own address register (the address is read
  // at issue; the compiler's code does the same).
#define X3C_SH(dst, src, b) "v_lshlrev_b32_sdwa %[" dst "], %[one], %[" src "] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" b "\n\t"
  auto step_first = [&](uint32_t d0, uint32_t d1sw) -> uint32_t {
    uint32_t t0, t1, t2, t3, out;
    asm(X3C_SH("t0", "d0", "3") X3C_SH("t1", "d0", "2") X3C_SH("t2", "d0", "1") X3C_SH("t3", "d0", "0")
        "ds_read_u16 %[t0], %[t0] offset:2048\n\t"
        "ds_read_u16 %[t1], %[t1] offset:2560\n\t"
        "ds_read_u16 %[t2], %[t2] offset:3072\n\t"
        "ds_read_u16 %[t3], %[t3] offset:3584\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bitop3_b32 %[t0], %[t0], %[t1], %[t2] bitop3:0x96\n\t"
        "v_bitop3_b32 %[out], %[t0], %[t3], %[d1] bitop3:0x96"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [out] "=&v"(out)
        : [d0] "v"(d0), [d1] "v"(d1sw), [one] "s"(1u));
    return out;
  };
  auto step_next = [&](uint32_t A, uint32_t d0, uint32_t d1sw) -> uint32_t {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    asm(X3C_SH("t0", "d0", "3") X3C_SH("t1", "d0", "2") X3C_SH("t2", "d0", "1") X3C_SH("t3", "d0", "0")
        "ds_read_u16 %[t0], %[t0] offset:2048\n\t"
        "ds_read_u16 %[t1], %[t1] offset:2560\n\t"
        X3C_SH("t4", "A", "0") X3C_SH("t5", "A", "1")
        "ds_read_u16 %[t2], %[t2] offset:3072\n\t"
        "ds_read_u16 %[t3], %[t3] offset:3584\n\t"
        X3C_SH("t6", "A", "2") X3C_SH("t7", "A", "3")
        "ds_read_u16 %[t4], %[t4] offset:4096\n\t"
        "ds_read_u16 %[t5], %[t5] offset:4608\n\t"
        "ds_read_u16 %[t6], %[t6] offset:5120\n\t"
        "ds_read_u16 %[t7], %[t7] offset:5632\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_bitop3_b32 %[t0], %[t0], %[t1], %[t2] bitop3:0x96\n\t"
        "v_bitop3_b32 %[t0], %[t0], %[t3], %[d1] bitop3:0x96\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bitop3_b32 %[t0], %[t0], %[t4], %[t5] bitop3:0x96\n\t"
        "v_bitop3_b32 %[A], %[t0], %[t6], %[t7] bitop3:0x96"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
          [t7] "=&v"(t7), [A] "+v"(A)
        : [d0] "v"(d0), [d1] "v"(d1sw), [one] "s"(1u));
    return A;
  };
#undef X3C_SH
#endif
  const uint64_t n_dw = (x3_len + 3) >> 2;
  const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  // (the wave index through readfirstlane: the frame number is then provably uniform, and the frame offsets, spans,
  // descriptors and the walk's checks are scalar loads and scalar arithmetic instead of 64-bit vector arithmetic)
#ifndef X3_CHECK_UNIFORM
#define X3_CHECK_UNIFORM 1
#endif
#if X3_CHECK_UNIFORM
  const uint64_t f0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
  const uint64_t f0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#endif
  if (f0 >= n_frames) return;  // whole wave

  // what is in flight for a frame: header dwords (6 aligned dwords cover 20 bytes at an even offset) and the
  // first X3_CHECK_AHEAD payload dwords of this lane
  // Range-checked buffer loads: the frame is the wave's, so its address is uniform -- one descriptor per frame in
  // SGPRs, ONE offset register (4 * lane) for all rows, the row in the instruction's immediate (and 4096 in the scalar
  // offset from row 16 on); dwords behind the end of the stream read as zero.  (With a 64-bit address and a bounds
  // select per load the kernel needed 167 VGPRs, and beside the decoder's groups a SIMD had room for ONE of its waves.)
  const uint32_t lane4 = 4u * lane;
  // the stream from dword `dw` on (uniform), `cap` bytes of it at most
  auto rsrc_at = [&](uint64_t dw, uint64_t cap = ~0ull) -> __amdgpu_buffer_rsrc_t {
    uint64_t left = dw < n_dw ? (n_dw - dw) * 4u : 0u;
    if (left > cap) left = cap;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dw);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dw >> 32));
    const uint64_t d = ((uint64_t)hi << 32) | lo;
    const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(left > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : left));
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(xw + d), 0, (int)bytes, 0x00020000);
  };
  auto load_rows = [&](const __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t (&pd)[X3_CHECK_AHEAD]) {
#pragma unroll
    for (uint32_t u = 0; u < X3_CHECK_AHEAD; ++u)
      pd[u] = u < 16u ? __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voff + 256u * u), 0, 0)
                      : __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voff + 256u * (u - 16u)), 4096, 0);
  };
  // `span`: bytes from this frame's header to the next frame's, where the frame index says so (the descriptor then ends
  // with the frame: rows behind it cost no memory traffic -- fetching 24 rows = 6 KB of every 5.3 KB frame was 1.23 x the
  // stream in HBM reads); ~0: not known, the payload is asked for speculatively up to the end of the stream
  auto request = [&](uint64_t off, uint64_t span, uint32_t (&hd)[6], uint32_t (&pd)[X3_CHECK_AHEAD]) {
    const __amdgpu_buffer_rsrc_t rh = rsrc_at(off >> 2);
#pragma unroll
    for (uint32_t i = 0; i < 6u; ++i) hd[i] = __builtin_amdgcn_raw_buffer_load_b32(rh, (int)(4u * i), 0, 0);
    // (whole dwords from the aligned dword of payload byte 0 to the one of the frame's last byte)
    const uint64_t cap = span == ~0ull || span < 20u ? ~0ull : ((((off + span + 3u) >> 2) - ((off + 20u) >> 2)) << 2);
    load_rows(rsrc_at((off + 20) >> 2, cap), lane4, pd);
  };
  auto span_of = [&](uint64_t f_) -> uint64_t {  // (frame_off holds n_frames entries: the last frame's end is not known)
    if (f_ + 1u >= n_frames) return ~0ull;
    const uint64_t a_ = frame_off[f_], b_ = frame_off[f_ + 1u];
    return b_ > a_ ? b_ - a_ : ~0ull;
  };
  uint64_t off_cur = frame_off[f0];
  uint64_t off_next = f0 + waves < n_frames ? frame_off[f0 + waves] : 0;
  uint64_t span_cur = span_of(f0), span_next = f0 + waves < n_frames ? span_of(f0 + waves) : ~0ull;
  uint32_t hd[6], pd[X3_CHECK_AHEAD];
  request(off_cur, span_cur, hd, pd);

  for (uint64_t f = f0; f < n_frames; f += waves) {
    // ---- requests for the frames behind this one
    const uint64_t off_next2 = f + 2 * waves < n_frames ? frame_off[f + 2 * waves] : 0;
    const uint64_t span_next2 = f + 2 * waves < n_frames ? span_of(f + 2 * waves) : ~0ull;
    uint32_t hn[6], pn[X3_CHECK_AHEAD];
    if (f + waves < n_frames) request(off_next, span_next, hn, pn);
    // ---- this frame: header (decoder.rs:69-118 + the walk's checks)
    const uint64_t off = off_cur;
    uint32_t plen = 0, samples = 0, pcrc = 0;
    int32_t st;
    if (off + 20 > x3_len) {
      st = X3D_STREAM_ENDS_IN_FRAME;
    } else {
      uint32_t h[5];
      const uint32_t sh = 8u * (uint32_t)(off & 3u);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const uint32_t x = x3_bswap32(hd[i]), y = x3_bswap32(hd[i + 1]);
        h[i] = sh ? (x << sh) | (y >> (32u - sh)) : x;
      }
      // header CRC from the LDS tables (init 0xFFFF folded into the first word)
      uint32_t hc = crc0_be32(h[0] ^ 0xFFFF0000u);
      hc = crc0_be32(h[1] ^ (hc << 16));
      hc = crc0_be32(h[2] ^ (hc << 16));
      hc = crc0_be32(h[3] ^ (hc << 16));
      st = x3_frame_header_check_words(h[0], h[1], h[4], hc, x3_len, off, plen, samples, pcrc, n_ch);
    }
    // ---- payload CRC (decodefile.rs:96-100)
    if (st == X3D_OK) {
      const uint64_t p0 = off + 20;
      uint32_t crc;
      if (plen < 4) {
        crc = 0xFFFFu;
        for (uint32_t i = 0; i < plen; ++i) crc = x3_crc_byte(crc, x3_be32_at(xw, n_dw, p0 + i) >> 24);
      } else {
        // The payload as rows of 64 aligned dwords, lane t on dword t of every row, folded Horner-style two rows a step on a
        // 32-bit state that is only CONGRUENT to the sum so far (the wave encoder's trick, round 3):
        //     A' = (A * x^4096 mod P)  ^  (row r * x^2048 mod P)  ^  row r+1
        // four look-ups for the state, four for the first row of the pair, none for the second: 4 per dword (round 2:
        // 4.5, in groups of four rows on a 16-bit sum) and 17 instead of 47 / 2 vector instructions per two rows; a
        // frame of 21 rows pays for 22 (then: 24).  The rows are taken as loaded (little-endian); only the second row of
        // a pair is byte-swapped.  The grid ends on a whole pair: what it holds beyond the payload is masked to zero,
        // which adds nothing to the sum but counts as trailing zero bytes, undone by one multiplication with x^(-8k) at
        // the end.  The CRC's init value is XORed into payload bytes 0, 1.
        const uint32_t lead = (uint32_t)(p0 & 3u);       // header bytes in front, inside dword 0
        const uint32_t nd = (lead + plen + 3u) >> 2;     // aligned dwords covering the payload
        const uint32_t tpad = 4u * nd - lead - plen;     // bytes behind the payload in the last dword
        const uint32_t R = (nd + 63u) >> 6;              // rows that hold payload
        const uint32_t last_lane = (nd - 1u) & 63u;
        // masks on the dwords AS LOADED (stream byte j = byte j).  Row 0: header bytes in front of the payload off, CRC
        // init in; row R-1: nothing behind the payload
        uint32_t and0 = 0xFFFFFFFFu, xor0 = 0u;
        if (lane == 0) {
          and0 = 0xFFFFFFFFu << (8u * lead);
          xor0 = lead <= 2 ? (0x0000FFFFu << (8u * lead)) : 0xFF000000u;
        }
        if (lane == 1 && lead == 3) xor0 = 0x000000FFu;
        const uint32_t last_mask = lane < last_lane ? 0xFFFFFFFFu : (lane == last_lane ? 0xFFFFFFFFu >> (8u * tpad) : 0u);
        uint32_t A = 0;
        uint32_t rows_done = 0;
        // (a frame index that disagrees with the header -- the caller's offsets are closer together than the frames are
        // long: the speculative fetch was cut short; the payload again, unbounded)
        if (span_cur != ~0ull && 20ull + plen > span_cur) load_rows(rsrc_at(p0 >> 2), lane4, pd);
        for (uint32_t rbase = 0; rbase < R; rbase += X3_CHECK_AHEAD) {
          if (rbase) {
            // payloads longer than the look-ahead (high-entropy data: up to 20 KB per frame): the same registers,
            // X3_CHECK_AHEAD rows per round trip (rows behind the payload are masked below)
            load_rows(rsrc_at(p0 >> 2), lane4 + 256u * rbase, pd);
          }
#pragma unroll
          for (uint32_t g2 = 0; g2 < X3_CHECK_AHEAD; g2 += 2u) {
            if (rbase + g2 < R) {  // (whole wave)
              const uint32_t row = rbase + g2;
              uint32_t d0 = pd[g2], d1 = pd[g2 + 1u];
              if (g2 == 0) {
                if (rbase == 0) d0 = (d0 & and0) ^ xor0;
              }
#if X3_CHECK_STEP_ASM
              if (row + 2u >= R) {   // (whole wave, the frame's last pair of rows: a branch, not selects in every step)
                __builtin_amdgcn_sched_barrier(0);
                if (row + 1u >= R) d0 &= last_mask;                        // row is the last one
                d1 &= row + 2u == R ? last_mask : 0u;                      // row + 1 is the last one, or behind it
              }
              if (tab_base != 0u) __builtin_trap();                        // (the asm's offsets are absolute; folded away)
              A = row ? step_next(A, d0, x3_bswap32(d1)) : step_first(d0, x3_bswap32(d1));
#else
              if (row + 1u >= R) d0 &= last_mask;                          // (whole wave) row is the last one
              if (row + 2u >= R) d1 &= row + 2u == R ? last_mask : 0u;     // (whole wave) row + 1 is the last one, or behind it
              uint32_t a2 = m2048_le(d0) ^ x3_bswap32(d1);
              if (row) a2 ^= m4096(A);
              A = a2;
#endif
              rows_done = row + 2u;
            }
#if X3_CHECK_SCHED_FENCE
            // keep the scheduler from hoisting every pair's table look-ups to the front: their results in flight were
            // most of the kernel's 167 VGPRs, and beside the decoder's groups a SIMD has room for ONE such wave
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        }
        const uint32_t acc = crc0_be32(A);   // the lane's column as a CRC-0 value (times x^16, reduced)
        // this lane's dwords are followed by 63 - lane dwords in their rows: times x^(32 * (63 - lane)) -- sixteen
        // pre-shifted words per lane, fetched here (4 KB in all, L1-resident) rather than held in registers all along
        {
          uint32_t kk[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4 k4 = reinterpret_cast<const uint4*>(kx64 + lane * 16u)[q];
            kk[4 * q] = k4.x; kk[4 * q + 1] = k4.y; kk[4 * q + 2] = k4.z; kk[4 * q + 3] = k4.w;
          }
          uint32_t r = 0;
#pragma unroll
          for (int bit = 0; bit < 16; ++bit) r ^= (0u - ((acc >> bit) & 1u)) & kk[bit];
          crc = r;
        }
        crc = (uint32_t)__builtin_amdgcn_readlane((int)x3_wave_xor_to_lane63_dpp(crc), 63);
        // the grid counted 64 * rows_done dwords, the payload ends tpad bytes inside dword nd - 1
        const uint32_t zeros = 4u * (64u * rows_done - nd) + tpad;
        if (zeros) crc = x3_gf_mul(crc, xinv8[zeros]);
      }
      if (crc != pcrc) st = X3D_FRAME_HEADER_INVALID_PAYLOAD_CRC;
    }
    if (lane == 0) status[f] = st;
    // ---- rotate
    off_cur = off_next;
    off_next = off_next2;
    span_cur = span_next;
    span_next = span_next2;
#pragma unroll
    for (uint32_t i = 0; i < 6u; ++i) hd[i] = hn[i];
#pragma unroll
    for (uint32_t u = 0; u < X3_CHECK_AHEAD; ++u) pd[u] = pn[u];
  }
}

// inverse Rice map (x3.rs:200-204): 0,-1,1,-2,2,...
__device__ __forceinline__ int32_t x3_inv_rice(uint32_t i) { return (i & 1u) ? -(int32_t)((i + 1u) >> 1) : (int32_t)(i >> 1); }

#define X3_DEC_RING_DW 32u       // per-lane input ring: 32 dwords = 128 bytes
#define X3_DEC_RING_STRIDE 36u   // row stride in dwords (144 B: 16-byte aligned rows, spread over banks)
#define X3_DEC_WIN 160u          // samples staged per lane between flushes (320 bytes)
#define X3_DEC_OUT_STRIDE 82u    // row stride in dwords (80 + 2: 8-byte aligned rows)
#define X3_DEC_CHUNK 20u         // samples decoded between two services of the input ring

__device__ __forceinline__ uint32_t x3_wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t t = __shfl_xor(v, d, X3_WAVE);
    v = t > v ? t : v;
  }
  return v;
}

// LDS traffic between lanes of ONE wave needs no s_barrier: a wave's DS instructions execute in
// issue order.  This only stops the compiler from moving LDS accesses across the point.
// (x3_dbg, X3_STAMP: x3_device.h)

#define X3_WAVE_LDS_ORDER()                                   \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

// FAST: every valid Rice codeword (zero run + terminator + sub-code) is at most 33 bits, so one
// window refill per sample suffices and a zero run of >= 32 bits is an error outright.  True for
// the default parameters; the host picks the general instantiation otherwise.
// LANES: frames decoded per wave (the first LANES lanes decode, all 64 lanes flush).  The decode
// loop is one long dependent instruction chain per wave, so with few frames (config 3: 69 120)
// it is better to spread them over MORE waves than to fill every lane: 16 frames per wave gives
// ~4 resident waves per SIMD whose chains interleave ("Decode occupancy").
template <bool FAST, int LANES>
__global__ void __launch_bounds__(64)
x3_decode_lanes_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                       uint64_t n_frames, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                       int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                       X3FrameMeta* __restrict__ meta) {
  __shared__ __attribute__((aligned(16))) uint32_t ring[LANES * X3_DEC_RING_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t outs[LANES * X3_DEC_OUT_STRIDE];
  __shared__ unsigned long long s_wo[LANES];  // sample offset of each lane's frame in wav
  __shared__ uint32_t s_ns[LANES];            // samples of each lane's frame (0 = not flushed cooperatively)

  const uint32_t lane = threadIdx.x;
  const bool decoder = lane < (uint32_t)LANES;
  const uint32_t dl = decoder ? lane : 0u;  // row used by this lane (idle lanes alias row 0, never write)
  const uint64_t f = (uint64_t)blockIdx.x * LANES + lane;
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  uint32_t* const row = ring + dl * X3_DEC_RING_STRIDE;
  uint32_t* const orow = outs + dl * X3_DEC_OUT_STRIDE;

  // ---- per-lane frame setup
  bool active = decoder && f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 2;
  uint64_t p0 = 0, wo = 0;
  if (active) {
    // header validation is repeated here (cheap, once per frame) so that this kernel does not depend
    // on x3_frame_check_kernel: the payload-CRC pass runs CONCURRENTLY on a second stream and the
    // two status arrays are merged afterwards (x3_decode_merge_kernel)
    uint32_t pcrc_unused;
    st = x3_frame_header_check(reinterpret_cast<const uint32_t*>(x3 - (reinterpret_cast<uintptr_t>(x3) & 3u)),
                               (x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u) + 3) >> 2,
                               x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u),
                               frame_off[f] + (reinterpret_cast<uintptr_t>(x3) & 3u), plen, samples, pcrc_unused);
    meta[f].payload_len = plen;
    meta[f].samples = samples;
    p0 = frame_off[f] + 20;
    if (st != X3D_OK) {
      active = false;
    } else if (samples == 0 || plen < 2) {
      st = X3D_BAD_ARG;  // the reference panics (decoder.rs:42,47)
      active = false;
    } else {
      if (wav_off) {
        wo = wav_off[f];
      } else {
        const uint64_t clip = f / g.fpc;
        const uint64_t idx = f - clip * g.fpc;
        wo = clip * g.clip_stride + idx * (uint64_t)p.spf;
      }
      if (wo + samples > wav_cap) {
        st = X3D_BAD_ARG;  // slice index panic
        active = false;
      }
    }
  }
  if (!active) { p0 = 0; plen = 2; wo = 0; }  // harmless addresses for idle lanes
  int16_t* __restrict__ const o = wav + wo;
  // frames whose output is 16-byte aligned are flushed cooperatively, the others store directly
  const bool coop = active && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0);
  if (decoder) {
    s_wo[lane] = wo;
    s_ns[lane] = coop ? samples : 0u;
  }

  // ---- input ring.  Offsets are "virtual": v = byte offset from x3b, the 16-byte-aligned address
  // at or below x3, so that 16-byte chunks are aligned in memory whatever x3's own alignment is.
  const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
  const uint8_t* __restrict__ const x3b = x3 - adj;
  const uint64_t v_end = adj + p0 + plen;              // first byte that must read as zero
  const uint64_t v_bits = adj + p0 + 2;                // the bit stream starts behind the raw first sample
  const uint64_t v_last = (v_end - 1) & ~15ull;        // last chunk that holds payload bytes
  uint64_t v_next = v_bits & ~15ull;                   // next 16-byte chunk the ring needs
  uint32_t wr_abs = 0;                                 // dwords written to the ring so far
  uint32_t rd_abs = 0;                                 // dwords taken out of the ring so far

  // request a chunk (never touches memory behind the payload's last chunk)
  auto request = [&](uint64_t v) -> uint4 {
    const uint64_t a = v < v_last ? v : v_last;
    return *reinterpret_cast<const uint4*>(x3b + a);
  };
  // park a chunk requested for virtual offset v in the ring, zeroing the bytes past the payload
  auto park = [&](uint4 c, uint64_t v) {
    const int64_t left = (int64_t)(v_end - v);  // payload bytes from the chunk start on (may be <= 0)
    if (left < 16) {
      uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int64_t r = left - 4 * d;
        w[d] = r >= 4 ? w[d] : (r <= 0 ? 0u : (w[d] & ((1u << (8u * (uint32_t)r)) - 1u)));
      }
      c = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (decoder) *reinterpret_cast<uint4*>(row + (wr_abs & (X3_DEC_RING_DW - 1u))) = c;
    wr_abs += 4;
  };

  uint64_t win = 0;          // next bits, MSB first
  uint32_t have = 0;         // valid bits in win
  bool deferred = false;     // a zero run the reference's reader counts differently (x3_decode_replay.h)
  uint32_t nextw_raw = 0;    // ring dword behind the window, in memory byte order (swapped on use)
  int32_t last = 0;
  uint32_t remaining = 0;    // samples of this lane's frame still to decode
  {
    uint4 c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);  // fill the ring: 128 bytes
#pragma unroll
    for (int k = 0; k < 8; ++k) park(c[k], v_next + 16u * k);
    v_next += 128;
    const uint32_t skip = (uint32_t)(v_bits & 15u);  // bytes of the first chunk in front of the bit stream
    rd_abs = skip >> 2;
    const uint32_t a = skip & 3u;
    const uint32_t w0 = x3_bswap32(row[rd_abs & (X3_DEC_RING_DW - 1u)]);
    ++rd_abs;
    win = (uint64_t)w0 << (32u + 8u * a);
    have = 32u - 8u * a;
    nextw_raw = row[rd_abs & (X3_DEC_RING_DW - 1u)];
  }
  if (active) {
    last = (int16_t)(uint16_t)((uint32_t)x3[p0] << 8 | x3[p0 + 1]);  // <Audio State> (decoder.rs:42)
    remaining = samples - 1u;
  }
  // the next three chunks are always requested one service ahead of use; what fits the ring when
  // they are parked is kept, the rest is simply requested again (it stays in L1/L2)
  uint4 ld0 = request(v_next), ld1 = request(v_next + 16), ld2 = request(v_next + 32);
  uint64_t v_req = v_next;

  auto refill = [&]() {
    if (have <= 32u) {
      win |= (uint64_t)x3_bswap32(nextw_raw) << (32u - have);
      have += 32u;
      ++rd_abs;
      nextw_raw = row[rd_abs & (X3_DEC_RING_DW - 1u)];
    }
  };
  // keep the ring ahead of the reader (48 B per service >= the 41 B a 20-sample chunk of literals
  // consumes; the ring starts full, so it never runs dry -- see DESIGN.md)
  auto service = [&]() {
    const uint32_t free_dw = X3_DEC_RING_DW - (wr_abs - rd_abs) - 1u;  // keep nextw's slot
    const uint32_t fit = free_dw >> 2;
    if (fit > 0) park(ld0, v_req);
    if (fit > 1) park(ld1, v_req + 16);
    if (fit > 2) park(ld2, v_req + 32);
    v_next += 16u * (fit > 3u ? 3u : fit);
    v_req = v_next;
#ifndef X3_DBG_NOLOAD
    ld0 = request(v_req);
    ld1 = request(v_req + 16);
    ld2 = request(v_req + 32);
#endif
  };

  // ---- output staging
  uint32_t wbase = 0;   // first sample index of the staged window (uniform, multiple of X3_DEC_WIN)
  int32_t carry = 0;    // even-indexed sample waiting for its odd partner
  auto flush = [&](uint32_t upto) {  // stage holds samples [wbase, upto) of every lane's frame
    // A frame that ends inside this window: the cooperative flush stores only 16-byte pieces that lie fully inside
    // the frame, so the owner stores the < 8 samples behind its last full piece -- HERE, while the window still
    // holds them (the frame may end in the middle of a block of the longer frames beside it, and the window
    // moves on before that block is over).
    if (coop && st == X3D_OK && samples > wbase && samples <= upto) {
      if (samples & 1u) orow[(samples - 1u - wbase) >> 1] = (uint32_t)carry & 0xFFFFu;
      const uint32_t done = samples & ~7u;
      const uint16_t* h = reinterpret_cast<const uint16_t*>(orow);
      for (uint32_t s = done > wbase ? done : wbase; s < samples; ++s) o[s] = (int16_t)h[s - wbase];
    }
    X3_WAVE_LDS_ORDER();
    const uint32_t pieces = (upto - wbase + 7u) >> 3;  // 16-byte pieces per frame in this window
    const uint32_t total = pieces * (uint32_t)LANES;
    for (uint32_t t = lane; t < total; t += 64u) {
      const uint32_t r = t / pieces, q = t - r * pieces;
      const uint32_t ns = s_ns[r];
      if (wbase + 8u * q + 8u <= ns) {
        const uint2* src = reinterpret_cast<const uint2*>(outs + r * X3_DEC_OUT_STRIDE + 4u * q);
        const uint2 lo = src[0], hi = src[1];
#ifndef X3_DBG_NOSTORE
        { const x3_u32x4 v = {lo.x, lo.y, hi.x, hi.y}; x3_store_stream16(wav + s_wo[r] + wbase + 8u * q, v); }
#else
        if (lo.x == 0x12345678u && hi.y == 0x9abcdef0u) wav[0] = 1;
#endif
      }
    }
    X3_WAVE_LDS_ORDER();
  };

  X3_WAVE_LDS_ORDER();  // s_wo / s_ns visible to the wave
  X3_STAMP(0);
  if (active) {
    carry = last;  // sample 0
    if (!coop || samples == 1u) o[0] = (int16_t)last;
  }

  // ---- lock-step decode: block index and in-block sample index are wave-uniform
  const uint32_t bl = p.block_len;
  uint32_t i = 1;  // uniform index of the sample being produced
  for (;;) {
    uint32_t cnt = remaining < bl ? remaining : bl;
    const uint32_t maxcnt = __any(cnt == bl) ? bl : x3_wave_max_u32(cnt);
    if (maxcnt == 0) break;
    X3_STAMP(1);
    service();
    X3_STAMP(2);
    // block header (decoder.rs:138-144).  Both block families are decoded by one branch-free
    // body: [z = leading zeros, Rice only] then a fixed-width field of `width` bits.
    //   Rice: r1 counts zeros and skips the 1 (decoder.rs:156-163) == width 1, level 1;
    //         r2r3: width hard-wired 2 / 4, level = 1 << nsubs (decoder.rs:180-186)
    //   BFP / literal: width = E (decoder.rs:209-235)
    uint32_t zmask = 0, width = 1, bound = 0xFFFFFFFFu, level = 0, lit = 0, neg_thresh = 0xFFFFFFFFu, neg2 = 0;
    if (cnt) {
      refill();
      const uint32_t hdr = (uint32_t)(win >> 58);  // 6 header bits
      const uint32_t ftype = hdr >> 4;
      if (ftype == 0) {
        const uint32_t E = (hdr & 15u) + 1u;
        win <<= 6;
        have -= 6;
        width = E;
        lit = E == 16u ? 1u : 0u;
        neg_thresh = 1u << (E - 1u);
        neg2 = lit ? 0u : (neg_thresh << 1);
        if (E <= 5u) {
          st = X3D_FRAME_DECODE_INVALID_BPF;
          cnt = 0;
          remaining = 0;
        }
      } else {
        win <<= 2;
        have -= 2;
        zmask = 0xFFFFFFFFu;
        width = ftype == 1u ? 1u : (ftype == 2u ? 2u : 4u);
        level = ftype == 1u ? 1u : (1u << (ftype == 2u ? p.k[1] : p.k[2]));
        bound = ftype == 1u ? p.inv_len[0] : (ftype == 2u ? p.inv_len[1] : p.inv_len[2]);
      }
    }
    const uint32_t rsh = 32u - width;
    X3_STAMP(3);
    for (uint32_t j = 0; j < maxcnt; ++j) {
      if (j && (j % X3_DEC_CHUNK) == 0) service();
      if (j < cnt) {
        refill();
        uint32_t top = (uint32_t)(win >> 32);
        uint32_t z = (uint32_t)__clz(top) & zmask;  // __clz(0) = 32
        uint32_t zextra = 0;
        if (!FAST) {
          if (zmask && top == 0) {  // zero run of >= 32 bits: keep counting (general parameters only)
            deferred = true;        // ... and let the reference's reader have the last word (x3_decode_replay.h)
            do {
              win <<= 32;
              have -= 32;
              zextra += 32;
              refill();
              top = (uint32_t)(win >> 32);
            } while (top == 0 && zextra < 128);
            z = top ? (uint32_t)__clz(top) : 0u;
          }
        }
        win <<= z;
        have -= z;
        if (!FAST) refill();
        const uint32_t v = (uint32_t)(win >> 32) >> rsh;
        win <<= width;
        have -= width;
        z += zextra;
        // Rice: i = r + level*(n-1) (decoder.rs:186), inverse table = zigzag (x3.rs:200-204)
        const uint32_t ii = v + level * z - level;
        const int32_t d_rice = (int32_t)(ii >> 1) ^ -(int32_t)(ii & 1u);
        // BFP: unsigned_to_i16 (decoder.rs:198-207)
        const int32_t d_bfp = (int32_t)v - (int32_t)(v > neg_thresh ? neg2 : 0u);
        const int32_t d = zmask ? d_rice : d_bfp;
        const int32_t nl = lit ? (int32_t)v : last + d;
        if (zmask && ii >= bound) {  // OutOfBoundsInverse (decoder.rs:160,187)
          st = X3D_OUT_OF_BOUNDS_INVERSE;
          cnt = 0;
          remaining = 0;
        } else {
          last = (int16_t)(uint16_t)nl;
        }
      }
      if (j < cnt) {  // still alive after this sample
        if (coop) {
          if (i & 1u) orow[(i - wbase) >> 1] = ((uint32_t)carry & 0xFFFFu) | ((uint32_t)last << 16);
          else carry = last;
        } else {
          o[i] = (int16_t)last;
        }
      }
      ++i;
      if (i - wbase == X3_DEC_WIN) {
        X3_STAMP(4);
        flush(i);
        wbase = i;
        X3_STAMP(5);
      }
    }
    X3_STAMP(4);
    if (remaining) remaining -= cnt;
  }
  if (i > wbase) flush(i);  // the partial last window
  if (active) {
    // bits taken from the ring (from its first chunk on) against the bits the payload holds there
    const uint64_t taken = 32ull * rd_abs - have, held = 8ull * (v_end - (v_bits & ~15ull));
    if (st == X3D_OUT_OF_BOUNDS_INVERSE || st == X3D_FRAME_DECODE_INVALID_BPF || deferred || taken > held)
      st = X3D_REPLAY;
  }
  if (decoder && f < n_frames) status[f] = st;
#ifdef X3_DBG_STAMPS
  X3_STAMP(6);
  if (lane == 0 && blockIdx.x < 4096)
    for (int k = 0; k < 8; ++k) x3_dbg[blockIdx.x * 8 + k] = dbg_acc[k];
#endif
}

// ---------------------------------------------------------------------------------------------
// x3_decode_fast_kernel -- the same lane-per-frame decoder with a branch-free sample body for
// parameter sets where every valid codeword is at most 32 bits (the default parameters:
// Rice0 <= 16, Rice1 <= 15, Rice3 <= 12 bits, BFP/literal <= 16).
//
// A wave64 VALU instruction costs a 4-cycle SIMD slot whatever the exec mask holds, and with only
// 69 120 frames (config 3) there is about one wave per SIMD, so the decoder is bound by the
// LENGTH of its per-sample instruction chain.  Hence:
//   * bit window = three big-endian words w0,w1,w2 + s = unconsumed bits of w0 in [0,31];
//     peek32 = v_alignbit_b32(w0, w1, s); consuming n <= 32 bits is s -= n and, if negative,
//     one word shift (3 v_cndmask) -- no 64-bit shifts, no refill branch;
//   * the ring word behind w2 is re-read from LDS every sample (ds_read_b32, branch-free) and
//     used one sample later, so its latency hides behind the value arithmetic;
//   * Rice and BFP/literal blocks share one body: z = clz(peek) & zmask, field = next `width`
//     bits; both deltas are computed and selected with v_bfi;
//   * ring words are byte-swapped once when parked, not when consumed.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
x3_decode_fast_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                      uint64_t n_frames, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                      int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                      X3FrameMeta* __restrict__ meta) {
  __shared__ __attribute__((aligned(16))) uint32_t ring[64 * X3_DEC_RING_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t outs[64 * X3_DEC_OUT_STRIDE];
  __shared__ unsigned long long s_wo[64];
  __shared__ uint32_t s_ns[64];

  const uint32_t lane = threadIdx.x;
  const uint64_t f = (uint64_t)blockIdx.x * 64 + lane;
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  uint32_t* const row = ring + lane * X3_DEC_RING_STRIDE;
  uint32_t* const orow = outs + lane * X3_DEC_OUT_STRIDE;

  // ---- per-lane frame setup (same checks as the general kernel)
  bool active = f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 2;
  uint64_t p0 = 0, wo = 0;
  if (active) {
    // header validation is repeated here (cheap, once per frame) so that this kernel does not depend
    // on x3_frame_check_kernel: the payload-CRC pass runs CONCURRENTLY on a second stream and the
    // two status arrays are merged afterwards (x3_decode_merge_kernel)
    uint32_t pcrc_unused;
    st = x3_frame_header_check(reinterpret_cast<const uint32_t*>(x3 - (reinterpret_cast<uintptr_t>(x3) & 3u)),
                               (x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u) + 3) >> 2,
                               x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u),
                               frame_off[f] + (reinterpret_cast<uintptr_t>(x3) & 3u), plen, samples, pcrc_unused);
    meta[f].payload_len = plen;
    meta[f].samples = samples;
    p0 = frame_off[f] + 20;
    if (st != X3D_OK) {
      active = false;
    } else if (samples == 0 || plen < 2) {
      st = X3D_BAD_ARG;
      active = false;
    } else {
      if (wav_off) {
        wo = wav_off[f];
      } else {
        const uint64_t clip = f / g.fpc;
        const uint64_t idx = f - clip * g.fpc;
        wo = clip * g.clip_stride + idx * (uint64_t)p.spf;
      }
      if (wo + samples > wav_cap) {
        st = X3D_BAD_ARG;
        active = false;
      }
    }
  }
  if (!active) { p0 = 0; plen = 2; wo = 0; }
  int16_t* __restrict__ const o = wav + wo;
  const bool coop = active && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0);
  s_wo[lane] = wo;
  s_ns[lane] = coop ? samples : 0u;

  // ---- input ring; words are parked BIG-ENDIAN
  const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
  // offsets are relative to this lane's first 16-byte chunk (a frame is < 64 KB): a 64-bit pointer per lane,
  // 32-bit arithmetic on everything else, streams of any length
  const uint64_t abs_bits = (uint64_t)adj + p0 + 2u;
  // (the first chunk: the one with the first block header, or -- a payload of two bytes that ends on a 16-byte boundary --
  // the one with the payload's last byte: the chunk behind it may be the first one behind the stream; x3_decode_split_kernel.h)
  const uint64_t abs_last = (uint64_t)adj + p0 + plen - 1u;
  const uint64_t abs_base = (abs_bits < abs_last ? abs_bits : abs_last) & ~15ull;
  const uint8_t* __restrict__ const x3b = (x3 - adj) + abs_base;
  const uint32_t v_bits = (uint32_t)(abs_bits - abs_base);   // first block header (0..16)
  const uint32_t v_end = v_bits - 2u + plen;               // end of the payload
  const uint32_t v_last = (v_end - 1u) & ~15u;             // last 16-byte chunk that holds payload
  uint32_t v_next = 0;
  uint32_t wr_abs = 0;

  auto request = [&](uint32_t v) -> uint4 {
    const uint32_t a = v < v_last ? v : v_last;
    return *reinterpret_cast<const uint4*>(x3b + a);
  };
  auto park = [&](uint4 c, uint32_t v) {
    const int32_t left = (int32_t)(v_end - v);
    uint32_t w[4] = {c.x, c.y, c.z, c.w};
    if (__any(left < 16)) {  // some lane is at (or past) the end of its payload: rare until the last blocks
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int32_t r = left - 4 * d;
        w[d] = r >= 4 ? w[d] : (r <= 0 ? 0u : (w[d] & ((1u << (8u * (uint32_t)r)) - 1u)));
      }
    }
    *reinterpret_cast<uint4*>(row + (wr_abs & (X3_DEC_RING_DW - 1u))) =
        make_uint4(x3_bswap32(w[0]), x3_bswap32(w[1]), x3_bswap32(w[2]), x3_bswap32(w[3]));
    wr_abs += 4;
  };

  {
    uint4 c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) park(c[k], v_next + 16u * k);
    v_next += 128;
  }
  // window: w0 holds `s` unconsumed bits (its low s bits), then w1, w2; widx = ring index of w0
  const uint32_t skip = v_bits;                        // (0..16)
  const uint32_t a0 = skip & 3u;
  uint32_t widx = (skip >> 2) - (a0 == 0 ? 1u : 0u);   // a0 == 0: start with a fully consumed w0
  uint32_t s = (32u - 8u * a0) & 31u;
  uint32_t w0 = row[widx & 31u], w1 = row[(widx + 1) & 31u], w2 = row[(widx + 2) & 31u];
  uint32_t w3 = row[(widx + 3) & 31u];                 // look-ahead word (re-read every sample)
  int32_t last = 0;
  uint32_t remaining = 0;
  if (active) {
    last = (int16_t)(uint16_t)((uint32_t)x3[p0] << 8 | x3[p0 + 1]);
    remaining = samples - 1u;
  }
  uint4 ld0 = request(v_next), ld1 = request(v_next + 16), ld2 = request(v_next + 32);
  uint32_t v_req = v_next;

  // consume n (<= 32) bits
  // The word shift is a v_bfi_b32 with a VGPR mask, NOT a compare + v_cndmask: on gfx950 a VOP2
  // v_cndmask that reads VCC without directly following its v_cmp costs ~19 cycles of a wave's issue
  // time instead of ~8 and is a SIMD-wide bottleneck when two waves share a SIMD (tools/ubench/issue_cost.hip).
  auto consume = [&](uint32_t n) {
    const int32_t s2 = (int32_t)s - (int32_t)n;
    const uint32_t m = (uint32_t)(s2 >> 31);  // all ones: the window moves on by one word
    s = (uint32_t)s2 & 31u;
    w0 = x3_bfi(m, w1, w0);
    w1 = x3_bfi(m, w2, w1);
    w2 = x3_bfi(m, w3, w2);
    widx -= m;
    w3 = row[(widx + 3u) & 31u];
  };
  auto service = [&]() {
    const uint32_t used = wr_abs - widx;                 // dwords from w0 on that the ring still needs
    const uint32_t fit = used >= X3_DEC_RING_DW ? 0u : (X3_DEC_RING_DW - used) >> 2;  // (widx may be -1)
    if (fit > 0) park(ld0, v_req);
    if (fit > 1) park(ld1, v_req + 16);
    if (fit > 2) park(ld2, v_req + 32);
    v_next += 16u * (fit > 3u ? 3u : fit);
    v_req = v_next;
    ld0 = request(v_req);
    ld1 = request(v_req + 16);
    ld2 = request(v_req + 32);
  };

  uint32_t wbase = 0;
  int32_t carry = 0;
  // the usual wave: 64 frames of the same size, one behind the other in wav, size a multiple of 8 samples
  const uint32_t S0 = __builtin_amdgcn_readfirstlane(samples);
  // (the builtin returns int: without the casts a low word with bit 31 set sign-extends over the high one, and every
  // group whose sample offset has that bit stops being "regular" -- half of all groups beyond 2^31 samples)
  const uint64_t wo0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wo >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wo);
  const bool regular = __all(coop && samples == S0 && wo == wo0 + (uint64_t)lane * S0) && (S0 & 7u) == 0;
  auto flush = [&](uint32_t upto) {
    // the < 8 samples behind the last full 16-byte piece of a frame that ends inside this window: stored by the
    // owner, here, while the window still holds them (see x3_decode_lanes_kernel)
    if (coop && st == X3D_OK && samples > wbase && samples <= upto) {
      if (samples & 1u) orow[(samples - 1u - wbase) >> 1] = (uint32_t)carry & 0xFFFFu;
      const uint32_t done = samples & ~7u;
      const uint16_t* h = reinterpret_cast<const uint16_t*>(orow);
      for (uint32_t sx = done > wbase ? done : wbase; sx < samples; ++sx) o[sx] = (int16_t)h[sx - wbase];
    }
    X3_WAVE_LDS_ORDER();
    const uint32_t pieces = (upto - wbase + 7u) >> 3;
    if (pieces == X3_DEC_WIN / 8u && regular) {
      // 64 equal, consecutive frames: row r goes to wav + wo0 + r*S0, nothing to look up
      uint32_t r = lane / 20u, q = lane - r * 20u;   // 64 = 3*20 + 4
      const uint4* __restrict__ const w4 = reinterpret_cast<const uint4*>(wav + wo0 + wbase);  // 16-byte aligned
      const bool inside = wbase + X3_DEC_WIN <= S0;  // whole window inside the frames (all but the last one)
#pragma unroll 5
      for (uint32_t it = 0; it < 20u; ++it) {
        if (inside || wbase + 8u * q + 8u <= S0) {
          const uint2* src = reinterpret_cast<const uint2*>(outs + r * X3_DEC_OUT_STRIDE + 4u * q);
          const uint2 lo = src[0], hi = src[1];
          { const x3_u32x4 v = {lo.x, lo.y, hi.x, hi.y}; x3_store_stream16(const_cast<uint4*>(w4) + (r * (S0 >> 3) + q), v); }
        }
        q += 4u;
        r += 3u;
        if (q >= 20u) { q -= 20u; r += 1u; }
      }
    } else if (pieces == X3_DEC_WIN / 8u) {
      uint32_t r = lane / 20u, q = lane - r * 20u;   // 64 = 3*20 + 4
#pragma unroll 4
      for (uint32_t it = 0; it < 20u; ++it) {
        const uint32_t ns = s_ns[r];
        if (wbase + 8u * q + 8u <= ns) {
          const uint2* src = reinterpret_cast<const uint2*>(outs + r * X3_DEC_OUT_STRIDE + 4u * q);
          const uint2 lo = src[0], hi = src[1];
          { const x3_u32x4 v = {lo.x, lo.y, hi.x, hi.y}; x3_store_stream16(wav + s_wo[r] + wbase + 8u * q, v); }
        }
        q += 4u;
        r += 3u;
        if (q >= 20u) { q -= 20u; r += 1u; }
      }
    } else {
      const uint32_t total = pieces * 64u;
      for (uint32_t t = lane; t < total; t += 64u) {
        const uint32_t r = t / pieces, q = t - r * pieces;
        const uint32_t ns = s_ns[r];
        if (wbase + 8u * q + 8u <= ns) {
          const uint2* src = reinterpret_cast<const uint2*>(outs + r * X3_DEC_OUT_STRIDE + 4u * q);
          const uint2 lo = src[0], hi = src[1];
          { const x3_u32x4 v = {lo.x, lo.y, hi.x, hi.y}; x3_store_stream16(wav + s_wo[r] + wbase + 8u * q, v); }
        }
      }
    }
    X3_WAVE_LDS_ORDER();
  };

  X3_WAVE_LDS_ORDER();
  if (active) {
    carry = last;
    if (!coop || samples == 1u) o[0] = (int16_t)last;
  }
  // true when every decoding lane stages through LDS (the normal case: 16-byte aligned frames)
  const bool all_coop = !__any(active && !coop);

  const uint32_t bl = p.block_len;
  uint32_t i = 1;  // index of the sample being produced; wave-uniform (kept in an SGPR)
  X3_STAMP(0);
  for (;;) {
    uint32_t cnt = remaining < bl ? remaining : bl;
    uint32_t maxcnt = bl;
    if (!__any(cnt == bl)) {
      maxcnt = x3_wave_max_u32(cnt);
      if (maxcnt == 0) break;
    }
    maxcnt = __builtin_amdgcn_readfirstlane(maxcnt);
    X3_STAMP(1);
    service();
    X3_STAMP(2);
    // block header: 2 bits ftype; ftype 0 -> 4 more bits E-1 (decoder.rs:138-144, 209-216)
    uint32_t zmask = 0, width = 1, bound = 0xFFFFFFFFu, level = 0, lit = 0, neg_thresh = 0xFFFFFFFFu, neg2 = 0;
    {
      const uint32_t hdr = __builtin_amdgcn_alignbit(w0, w1, s) >> 26;  // 6 header bits
      const uint32_t ftype = hdr >> 4;
      const uint32_t E = (hdr & 15u) + 1u;
      const bool bfp = ftype == 0;
      if (cnt) {
        consume(bfp ? 6u : 2u);
        if (bfp) {
          width = E;
          lit = E == 16u ? 1u : 0u;
          neg_thresh = 1u << (E - 1u);
          neg2 = lit ? 0u : (neg_thresh << 1);
          if (E <= 5u) {
            st = X3D_FRAME_DECODE_INVALID_BPF;
            cnt = 0;
            remaining = 0;
          }
        } else {
          zmask = 0xFFFFFFFFu;
          width = ftype == 1u ? 1u : (ftype == 2u ? 2u : 4u);
          level = ftype == 1u ? 1u : (1u << (ftype == 2u ? p.k[1] : p.k[2]));
          bound = ftype == 1u ? p.inv_len[0] : (ftype == 2u ? p.inv_len[1] : p.inv_len[2]);
        }
      }
    }
    const uint32_t rsh = 32u - width;
    const uint32_t litmask = lit ? 0xFFFFFFFFu : 0u;
    const uint32_t nlevel = 0u - level;
    const uint32_t lsh = 31u - (uint32_t)__clz(level | 1u);  // level is a power of two (or 0 for BFP)
    uint32_t maxii = 0;  // largest inverse-table index seen in this block (Rice lanes only)
    // the same per-block constants in both halves of a dword, for the two-sample body (packed 16-bit math)
    const uint32_t nlevel2 = (nlevel & 0xFFFFu) * 0x10001u;
    const uint32_t nt2 = neg_thresh * 0x10001u;            // BFP lanes: <= 0x8000 (0xFFFFFFFF on Rice lanes: unused)
    const uint32_t neg22 = (neg2 & 0xFFFFu) * 0x10001u;
    uint32_t maxii2 = 0;  // packed running maximum of the two-sample body

    // one sample, branch-free.  A zero run of 32 (t == 0) gives ii >= 31*level, beyond every bound
    // the fast path admits, so it needs no test of its own; an over-long codeword (z + width > 32)
    // only happens together with such an error, so `consume` need not clamp it.
    // `last` is kept modulo 2^32 (only its low 16 bits are ever stored or compared).
    auto sample = [&](uint32_t idx, auto direct_tag) {
      constexpr bool DIRECT = decltype(direct_tag)::value;
      const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);        // next 32 bits
      const uint32_t z = (uint32_t)__clz(t) & zmask;                   // __clz(0) = 32
      const uint32_t v = (t << (z & 31u)) >> rsh;                      // the field behind the zero run
      consume(z + width);
      // Rice: i = r + level*(n-1) (decoder.rs:186); inverse table = zigzag (x3.rs:200-204)
      const uint32_t ii = (z << lsh) + (v + nlevel);
      const uint32_t d_rice = (ii >> 1) ^ (0u - (ii & 1u));
      // BFP: unsigned_to_i16 (decoder.rs:198-207)
      const uint32_t d_bfp = v - (v > neg_thresh ? neg2 : 0u);
      const uint32_t d = (d_rice & zmask) | (d_bfp & ~zmask);
      const uint32_t nl = (((uint32_t)last + d) & ~litmask) | (v & litmask);  // literal: field = sample
      const uint32_t iim = ii & zmask;
      maxii = iim > maxii ? iim : maxii;
      last = (int32_t)nl;
      if (DIRECT) {
        if (coop) {
          if (idx & 1u) orow[(idx - wbase) >> 1] = ((uint32_t)carry & 0xFFFFu) | ((uint32_t)last << 16);
          else carry = last;
        } else {
          o[idx] = (int16_t)last;
        }
      } else {
        if (idx & 1u) orow[(idx - wbase) >> 1] = ((uint32_t)carry & 0xFFFFu) | ((uint32_t)last << 16);
        else carry = last;
      }
    };
    // two samples from ONE 32-bit peek and ONE window update (two valid codewords are <= 32 bits on
    // this path); staged output only.  The values are computed for both samples at once in packed 16-bit
    // arithmetic (everything here is modulo 2^16 anyway): P = (la, lb) is the pair of output samples,
    // which is also the staged dword when the pair starts on an even sample index; when it starts on an
    // odd one (the usual case: blocks start at sample 1 and have an even length) the dword is
    // (previous pair's lb, la) = v_alignbit(P, prevP, 16).
    // A zero run of 32 makes v_ffbh return -1 (shifts use its low 5 bits): the index is far beyond every
    // bound in 16 bits as well, so the error is still recorded by maxii2 and the frame stops at this block.
    // BFP lanes have bound = 0xFFFFFFFF, which no 16-bit maximum reaches.
    uint32_t LL = 0, prevP = 0;  // (last, last); the previous pair (its high half is the pending sample)
    auto sample2 = [&](uint32_t* dst, auto odd_tag) {
      constexpr bool ODD = decltype(odd_tag)::value;
      const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
      const uint32_t z1 = x3_ffbh(t) & zmask;
      const uint32_t v1 = (t << (z1 & 31u)) >> rsh;
      const uint32_t n1 = z1 + width;
      const uint32_t t2 = t << (n1 & 31u);
      const uint32_t z2 = x3_ffbh(t2) & zmask;
      const uint32_t v2 = (t2 << (z2 & 31u)) >> rsh;
      consume(n1 + z2 + width);
      // Rice: i = r + level*(n-1) (decoder.rs:186); inverse table = zigzag (x3.rs:200-204)
      const uint32_t I = x3_pk_add_u16(x3_pack_lo16((z1 << lsh) + v1, (z2 << lsh) + v2), nlevel2);
      maxii2 = x3_pk_max_u16(maxii2, I);
      const uint32_t R = x3_pk_lshr_b16_1(I) ^ x3_pk_sub_u16(0u, I & 0x00010001u);
      // BFP: unsigned_to_i16 (decoder.rs:198-207): v - (v > thresh ? 2*thresh : 0), strict compare
      const uint32_t V = x3_pack_lo16(v1, v2);
      const uint32_t M = x3_pk_ashr_i16_15(x3_pk_sub_u16(nt2, V));  // 0xFFFF where v > thresh
      const uint32_t B = x3_pk_sub_u16(V, M & neg22);
      const uint32_t D = (R & zmask) | (B & ~zmask);                 // (d1, d2)
      const uint32_t Q = x3_pk_add_u16(D, D << 16);                  // (d1, d1 + d2)
      uint32_t P = x3_pk_add_u16(Q, LL);                             // (last + d1, last + d1 + d2)
      P = (P & ~litmask) | (V & litmask);                            // literal: field = sample
      LL = __builtin_amdgcn_perm(P, P, 0x07060706u);                 // (lb, lb)
      *dst = ODD ? __builtin_amdgcn_alignbit(P, prevP, 16) : P;
      prevP = P;
    };

    X3_STAMP(3);
    uint32_t j = 0;
    while (j < maxcnt) {  // uniform: segments end at flush points and at ring services
      uint32_t seg_end = j + (X3_DEC_WIN - (i - wbase));
      const uint32_t next_service = (j / X3_DEC_CHUNK + 1u) * X3_DEC_CHUNK;
      seg_end = seg_end < maxcnt ? seg_end : maxcnt;
      seg_end = seg_end < next_service ? seg_end : next_service;
      seg_end = __builtin_amdgcn_readfirstlane(seg_end);
      const uint32_t i0 = __builtin_amdgcn_readfirstlane(i);
      if (cnt >= seg_end) {
        // every sample of the segment belongs to this lane's block
        if (all_coop) {
          uint32_t jj = j;
          if (jj + 1 < seg_end) {
            LL = ((uint32_t)last & 0xFFFFu) * 0x10001u;
            uint32_t* dst = orow + ((i0 - wbase) >> 1);  // dword of sample i0 (odd i0: its high half)
            if (i0 & 1u) {
              prevP = (uint32_t)carry << 16;
              for (; jj + 1 < seg_end; jj += 2) sample2(dst++, std::true_type{});
              carry = (int32_t)(prevP >> 16);
            } else {
              for (; jj + 1 < seg_end; jj += 2) sample2(dst++, std::false_type{});
            }
            last = (int32_t)(LL & 0xFFFFu);
          }
          if (jj < seg_end) sample(i0 + (jj - j), std::false_type{});
        } else {
          for (uint32_t jj = j; jj < seg_end; ++jj) sample(i0 + (jj - j), std::true_type{});
        }
      } else if (cnt > j) {
        // tail: the block ends inside the segment
        for (uint32_t jj = j; jj < cnt; ++jj) sample(i + (jj - j), std::true_type{});
      }
      i = i0 + (seg_end - j);
      j = seg_end;
      X3_STAMP(4);
      if (i - wbase == X3_DEC_WIN) {
        flush(i);
        wbase = i;
        X3_STAMP(5);
      }
      if (j < maxcnt && (j % X3_DEC_CHUNK) == 0) service();
    }
    {
      const uint32_t m2 = max(maxii2 & 0xFFFFu, maxii2 >> 16);
      maxii = m2 > maxii ? m2 : maxii;
    }
    const uint32_t errflag = maxii >= bound ? 1u : 0u;
    if (errflag) {  // OutOfBoundsInverse (decoder.rs:160,187): the frame stops here
      st = X3D_OUT_OF_BOUNDS_INVERSE;
      cnt = 0;
      remaining = 0;
    }
    if (remaining) remaining -= cnt;
  }
  if (i > wbase) flush(i);
  if (active) {
    // read position (bits from the ring's first chunk on) against the end of the payload
    const int32_t taken = 32 * (int32_t)widx + 32 - (int32_t)s;
    if (st == X3D_OUT_OF_BOUNDS_INVERSE || st == X3D_FRAME_DECODE_INVALID_BPF || taken > (int32_t)(8u * v_end))
      st = X3D_REPLAY;  // x3_decode_replay.h
  }
  if (f < n_frames) status[f] = st;
#ifdef X3_DBG_STAMPS
  X3_STAMP(6);
  if (lane == 0 && blockIdx.x < 4096)
    for (int k = 0; k < 8; ++k) x3_dbg[blockIdx.x * 8 + k] = dbg_acc[k];
#endif
}

// first frame with a non-zero status, and the samples of the good frames before it
// (struct X3DecodeSummary: x3_tables.h)

// Merge the two concurrent passes: a frame's status is the check pass's (header, then payload CRC --
// the reference tests those first, decodefile.rs:112-121,96-100) if that is non-zero, else the
// decoder's.  A frame the decoder deferred (X3D_REPLAY: decode error, zero run >= 32 bits or a read behind
// the payload's end -- see x3_decode_replay.h) is decoded again here, by this thread alone, through the
// reference's own reader semantics.  Also finds the first bad frame and the total sample count.
// summary must be pre-set to {n_frames, 0, 0}.
__global__ void __launch_bounds__(256)
x3_decode_merge_kernel(const int32_t* __restrict__ cstatus, int32_t* __restrict__ status,
                       const X3FrameMeta* __restrict__ meta, uint64_t n_frames_arg, X3DecodeSummary* __restrict__ out,
                       const uint8_t* __restrict__ x3, const uint64_t* __restrict__ frame_off, X3Geom g,
                       const uint64_t* __restrict__ wav_off, X3DevParams p, int16_t* __restrict__ wav, uint32_t bl0,
                       const unsigned long long* __restrict__ d_nf = nullptr) {
  const uint64_t n_frames = d_nf ? (*d_nf < n_frames_arg ? (uint64_t)*d_nf : n_frames_arg) : n_frames_arg;
  // grid-stride, one atomic per workgroup: a thousand waves adding to ONE address took 11 of this kernel's 16 us
  __shared__ unsigned long long s_ns[4];
  unsigned long long ns = 0;
  for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n_frames; f += (uint64_t)gridDim.x * blockDim.x) {
    const int32_t cs = cstatus[f];
    int32_t st = status[f];
    if (cs != 0) {
      st = cs;
      status[f] = cs;
    } else if (bl0 && meta[f].samples > 1u && st != X3D_BAD_ARG) {
      // Parameters.block_len == 0 (the kernels ran with 1): what the reference makes of such a frame is decided by its
      // block TYPE bits alone (x3_replay_frame) -- FrameDecodeInvalidBPF or its panic; nothing is written behind out[0]
      const X3FrameMeta m = meta[f];
      X3DevParams q = p;
      q.block_len = 0;
      int16_t first_sample;
      st = x3_replay_frame(x3 + frame_off[f] + 20, m.payload_len, m.samples, q, &first_sample);
      status[f] = st;
    } else if (st == X3D_REPLAY) {
      // (the decoder has validated the header, the sample count and the output range of this frame)
      const X3FrameMeta m = meta[f];
      uint64_t wo;
      if (wav_off) {
        wo = wav_off[f];
      } else {
        const uint64_t clip = f / g.fpc;
        wo = clip * g.clip_stride + (f - clip * g.fpc) * (uint64_t)p.spf;
      }
      st = x3_replay_frame(x3 + frame_off[f] + 20, m.payload_len, m.samples, p, wav + wo);
      status[f] = st;
    }
    if (st != 0) atomicMin(&out->first_bad, (unsigned long long)f);
    else ns += meta[f].samples;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ns += __shfl_xor(ns, o, X3_WAVE);
  if ((threadIdx.x & 63u) == 0) s_ns[threadIdx.x >> 6] = ns;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_ns[0] + s_ns[1] + s_ns[2] + s_ns[3];
    if (t) atomicAdd(&out->samples_before, t);
  }
}

// only when a frame is bad: status of the first bad frame and the samples of the good frames before it
__global__ void __launch_bounds__(1024)
x3_decode_prefix_kernel(const int32_t* __restrict__ status, const X3FrameMeta* __restrict__ meta, uint64_t n_frames,
                        X3DecodeSummary* __restrict__ out) {
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const unsigned long long first = out->first_bad;
  unsigned long long sum = 0;
  for (uint64_t f = threadIdx.x; f < first && f < n_frames; f += blockDim.x) sum += meta[f].samples;
  if (sum) atomicAdd(&s_sum, sum);
  __syncthreads();
  if (threadIdx.x == 0) {
    out->samples_before = s_sum;
    out->first_bad_status = first < n_frames ? status[first] : 0;
  }
}
