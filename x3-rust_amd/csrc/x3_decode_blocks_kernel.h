// x3_decode_blocks_kernel.h -- the decoder of round 6: one WALKER wave finds where the blocks begin, three DECODER waves
// decode them a BLOCK per lane.
//
// Why.  A frame is one serial bit stream (a block begins where the one before it ends), so the lane-per-frame decoders
// (x3_decode_split_kernel.h and the single-wave kernels behind it) are one dependent instruction chain per frame: config
// 3 is 1 080 waves on 1 024 SIMDs, a lone wave issues a dependent vector instruction every ~8.5 clocks, and the kernel
// sat at 0.65 ms -- 30 % of the SIMDs' issue rate, 54 % of the wave-cycles waiting -- through five rounds of
// rearrangements (DESIGN_HISTORY.md).  But only the block BOUNDARIES are serial.  Once a block's first bit is known it
// depends on nothing but the sample in front of it (decoder.rs:36-58), and that dependency is an addition: a prefix sum.
//
// So the group (up to 64 frames) is split by what is serial and what is not:
//
//   wave 0, the WALKER (a frame per lane, as before): header checks, the per-lane input ring, and the codeword LENGTHS
//     only -- per pair of codewords one peek, two v_ffbh / v_mad, the window update: no values, no hand-over of
//     indices.  For every block it leaves the bit position the block begins at (16 bits, relative to the batch) in
//     LDS.  It works in BATCHES of 32 blocks per frame, one s_barrier per batch (the three-wave kernel: one per block).
//   waves 1-3, the DECODERS: a batch behind the walker, a BLOCK per lane -- 32 consecutive blocks of one frame in
//     lanes 0-31, of another in lanes 32-63.  The frame pieces' bytes are copied into LDS (coalesced 16-byte loads:
//     no per-lane ring, no service), each lane walks ITS 20 codewords from its block's first bit (decoder.rs:132-235)
//     and sums its differences from zero; a 32-lane segmented scan of the blocks' totals (literal blocks reset the
//     sum) gives every block the sample in front of it, ten packed additions put it on the block's samples.  The 64
//     lanes' samples are consecutive in wav: they go through LDS once and leave as WHOLE 128-byte lines, 32 lanes x
//     16 bytes per frame piece -- no staging rings, no flusher wave.
//
// Every lane's first batch is cut so that it ends on a line of the destination (n0 blocks, 17..32: 40 * n0 + the
// row's phase is a multiple of 128); all later batches are 32 blocks = 1 280 bytes = ten lines.  Partial lines are
// only written at the two ends of a frame's row.
//
// The group count is chosen by the host so that every CU gets the same number of groups (config 3: 1 280 groups of 54
// frames, five per CU) -- the walker's idle lanes cost nothing, it is a latency chain; the decoders' lanes are blocks.
// LDS (28.4 KB) admits exactly five groups per CU.
//
// Anything irregular -- a decode error (BFP exponent, table bound), a zero run of 32 bits, a read behind the payload,
// a batch longer than valid codewords can make it -- marks the FRAME for the reference's own reader
// (X3D_REPLAY, x3_decode_replay.h; x3_decode_merge_kernel), exactly as in the lane-per-frame kernels; what the
// decoders wrote for such a frame is overwritten there.  Every read is bounded whatever the bits say.
//
// Geometry: block_len = 20, the default Rice codes (the hard-wired decoder.rs:180), output rows on 8-byte boundaries
// (the host sends everything else to the older kernels).  Same results as x3_decode_split_kernel.
#pragma once
#include "x3_decode_kernel.h"

// A lane of a decoder wave takes a UNIT: UNIT samples (20, or 10) of one block.  Blocks of 20 or 10 samples are one unit, blocks
// of 40 are two (UPB, units per block: the second unit of a block reads the block's header where the first unit begins and
// walks on from its own first bit) -- so that what a lane holds, what a piece is and what the staging takes do not depend on
// the block length: instantiations <20, 1> (the default geometry), <20, 2> (block_len 40), <10, 1> (block_len 10).
#define X3B_NB 32u                       // units per batch and frame
#ifndef X3B_D
#define X3B_D 3u
#endif
// (decoder waves)                         // decoder waves
#define X3B_WAVES (1u + X3B_D)
#define X3B_SPAN_MAX (X3B_NB * 326u)     // bits 32 valid units can take (a literal block of 20: 6 + 20 * 16)
#define X3B_IN_PITCH 1392u               // input staging per frame piece: 86 chunks of 16 bytes + 16
#define X3B_IN_CHUNKS 85u
#define X3B_OUT_PITCH 1408u              // output staging per frame piece: up to 120 bytes of phase + 1 280 + a pending sample
#define X3B_SCRATCH (2u * X3B_OUT_PITCH) // per decoder wave (input and output staging share it)
#define X3B_DESC_PITCH 34u               // uint16 per frame and batch buffer: 32 + 2 (17 dwords: rows spread over the banks)
#define X3B_PERIOD 2u                    // the walker's ring is topped up every second block
#define X3B_AHEAD 3u
#ifndef X3B_WALKER_PRIO
#define X3B_WALKER_PRIO 3
#endif

// TIMING experiments (results are wrong; -DX3_EXPERIMENT builds only): 1 = the decoders do nothing, 2 = the walker skips its
// codeword walk, 4 = no global stores from the decoders, 8 = no staging copy, 16 = the decoders skip their pair loops,
// 32 = the walker skips its ring service
#ifndef X3B_KO
#define X3B_KO 0
#endif
#if X3B_KO && !defined(X3_EXPERIMENT)
#error "X3B_KO builds give wrong results: experiment builds only (-DX3_EXPERIMENT)"
#endif

#define X3B_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// one pair of codewords from the 32-bit peek t (x3_decode_split_kernel.h, the parser's pair): a codeword = z zeros +
// `width` bits (z counts in Rice blocks only: zmask), n = z + width bits in all; nn = -n serves as shift count for
// "drop n bits" and "the fw bits that end n bits in".  Returns nn1 + nn2.
#define X3B_PAIR_FIELDS(t, zmask, nwidth, fw, z1, v1, z2, v2, nsum)                                         \
  do {                                                                                                      \
    uint32_t t2_, nn1_, nn2_;                                                                               \
    asm("v_ffbh_u32 %[z1], %[tt]\n\t"                                                                       \
        "v_mad_i32_i24 %[nn1], %[z1], %[zm], %[nw]\n\t"                                                     \
        "v_alignbit_b32 %[t2], %[tt], 0, %[nn1]\n\t"                                                        \
        "v_bfe_u32 %[v1], %[tt], %[nn1], %[fw_]\n\t"                                                        \
        "v_ffbh_u32 %[z2], %[t2]\n\t"                                                                       \
        "v_mad_i32_i24 %[nn2], %[z2], %[zm], %[nw]\n\t"                                                     \
        "v_bfe_u32 %[v2], %[t2], %[nn2], %[fw_]\n\t"                                                        \
        "v_add_u32 %[ns], %[nn1], %[nn2]"                                                                   \
        : [z1] "=&v"(z1), [v1] "=&v"(v1), [z2] "=&v"(z2), [v2] "=&v"(v2), [nn1] "=&v"(nn1_), [nn2] "=&v"(nn2_), \
          [t2] "=&v"(t2_), [ns] "=&v"(nsum)                                                                 \
        : [tt] "v"(t), [zm] "v"(zmask), [nw] "v"(nwidth), [fw_] "v"(fw));                                   \
  } while (0)

// ---- the WALKER's pair: the lengths only.  14 vector instructions + 1 LDS read; the word behind the window is read into
// WR while WU (read by the pair before) moves into the window -- a read has a whole pair's time to arrive.
// in/out: w0 w1 s qb; temps t z n1; constants zm nw c124 rowb
#define X3B_WPAIR(WR, WU)                                          \
  "v_alignbit_b32 %[t], %[w0], %[w1], %[s]\n\t"                    \
  "v_ffbh_u32 %[z], %[t]\n\t"                                      \
  "v_mad_i32_i24 %[n1], %[z], %[zm], %[nw]\n\t"                    \
  "v_alignbit_b32 %[t], %[t], 0, %[n1]\n\t"                        \
  "v_ffbh_u32 %[z], %[t]\n\t"                                      \
  "v_mad_i32_i24 %[z], %[z], %[zm], %[nw]\n\t"                     \
  "v_add3_u32 %[s], %[s], %[n1], %[z]\n\t"                         \
  "v_ashrrev_i32 %[n1], 31, %[s]\n\t"                              \
  "v_lshl_add_u32 %[qb], %[n1], 2, %[qb]\n\t"                      \
  "v_and_or_b32 %[t], %[qb], %[c124], %[rowb]\n\t"                 \
  "ds_read_b32 %[" WR "], %[t]\n\t"                                \
  "v_and_b32 %[s], 31, %[s]\n\t"                                   \
  "v_bfi_b32 %[w0], %[n1], %[w1], %[w0]\n\t"                       \
  "s_waitcnt lgkmcnt(1)\n\t"                                       \
  "v_bfi_b32 %[w1], %[n1], %[" WU "], %[w1]\n\t"

// ---- a DECODER's pair: codewords -> indices -> differences -> samples counted from zero, 29 vector instructions + 1 LDS
// read (the parser's pair and the valuer's of x3_decode_split_kernel.h in one lane).  The Rice / BFP choice is in the
// constants (a Rice lane's tm12 / neg22 make the BFP step the identity, a BFP lane's zsh2 = 0 the zigzag step); no
// literal blocks here (the caller takes the general path when a lane of the wave has one).
// PI: the pair before (its high half = the last sample so far), PO: this pair; WOUT = (the sample in front, this pair's first)
#define X3B_DPAIR(WR, WU, PI, PO, WOUT)                                                     \
  "v_alignbit_b32 %[t], %[w0], %[w1], %[s]\n\t"                                             \
  "v_ffbh_u32 %[z1], %[t]\n\t"                                                              \
  "v_mad_i32_i24 %[n1], %[z1], %[zm], %[nw]\n\t"                                            \
  "v_alignbit_b32 %[t2], %[t], 0, %[n1]\n\t"                                                \
  "v_bfe_u32 %[v1], %[t], %[n1], %[fw]\n\t"                                                 \
  "v_ffbh_u32 %[z2], %[t2]\n\t"                                                             \
  "v_mad_i32_i24 %[n2], %[z2], %[zm], %[nw]\n\t"                                            \
  "v_bfe_u32 %[v2], %[t2], %[n2], %[fw]\n\t"                                                \
  "v_add3_u32 %[s], %[s], %[n1], %[n2]\n\t"                                                 \
  "v_ashrrev_i32 %[n1], 31, %[s]\n\t"                                                       \
  "v_lshl_add_u32 %[qa], %[n1], 2, %[qa]\n\t"                                               \
  "ds_read_b32 %[" WR "], %[qa]\n\t"                                                        \
  "v_and_b32 %[s], 31, %[s]\n\t"                                                            \
  "v_bfi_b32 %[w0], %[n1], %[w1], %[w0]\n\t"                                                \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                \
  "v_bfi_b32 %[w1], %[n1], %[" WU "], %[w1]\n\t"                                            \
  "v_lshl_add_u32 %[z1], %[z1], %[lsh], %[v1]\n\t"                                          \
  "v_lshl_add_u32 %[z2], %[z2], %[lsh], %[v2]\n\t"                                          \
  "v_perm_b32 %[v1], %[z2], %[z1], %[sel]\n\t"                                              \
  "v_pk_max_u16 %[mx], %[mx], %[v1]\n\t"                                                    \
  "v_pk_add_u16 %[v2], %[v1], %[tm12]\n\t"                                                  \
  "v_and_b32 %[v2], %[v2], %[neg22]\n\t"                                                    \
  "v_pk_sub_u16 %[t], %[v1], %[v2]\n\t"                                                     \
  "v_and_b32 %[t2], %[t], %[zsh2]\n\t"                                                      \
  "v_pk_lshrrev_b16 %[n2], %[zsh2], %[t]\n\t"                                               \
  "v_pk_sub_u16 %[t2], 0, %[t2]\n\t"                                                        \
  "v_xor_b32 %[n2], %[n2], %[t2]\n\t"                                                       \
  "v_pk_add_u16 %[z1], %[n2], %[" PI "] op_sel:[0,1] op_sel_hi:[1,1]\n\t"                   \
  "v_pk_mad_u16 %[" PO "], %[n2], %[c1], %[z1] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"        \
  "v_alignbit_b32 %[" WOUT "], %[" PO "], %[" PI "], 16\n\t"

__device__ __forceinline__ void x3b_lds_write_b32(uint32_t addr, uint32_t v) {
  *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(addr) = v;
}
// sixteen bytes to GLOBAL memory, non-temporal (an address that comes out of LDS as an integer is "generic" to the compiler: a
// flat_store, which counts in lgkmcnt too and is slower than the global_store this makes of it)
__device__ __forceinline__ void x3b_global_store16_nt(uint8_t* p, x3_u32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((address_space(1))) x3_u32x4*>(reinterpret_cast<uintptr_t>(p)));
}
__device__ __forceinline__ void x3b_global_store2(uint8_t* p, uint32_t v) {
  *reinterpret_cast<__attribute__((address_space(1))) uint16_t*>(reinterpret_cast<uintptr_t>(p)) = (uint16_t)v;
}
__device__ __forceinline__ uint4 x3b_global_load16(uint64_t addr) {
  const x3_u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) x3_u32x4*>(addr);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// units of the first batch of a row that begins at byte address `row`: (row + 2 * UNIT * n0) is a multiple of 128, and
// whole blocks.  UNIT 20: rows on 8-byte boundaries (16-byte for two units a block: n0 is then even), n0 = 17..32;
// UNIT 10: rows on 4-byte boundaries, n0 = 1..32
template <uint32_t UNIT, uint32_t UPB>
__device__ __forceinline__ uint32_t x3b_first_batch(uint64_t row) {
  if (UNIT == 20u) {
    const uint32_t phi8 = ((uint32_t)row & 127u) >> 3;      // 40 n0 = -8 phi8 (mod 128): n0 = 3 phi8 (mod 16)
    const uint32_t n0 = 17u + ((3u * phi8 + 15u) & 15u);
    // (two units a block: whole blocks come first -- a row that begins 8 bytes into a 16-byte unit then has pieces that do
    // not end on lines, and its lines are written in parts: slower, the same bytes)
    return UPB == 2u ? (n0 & ~1u) : n0;
  }
  const uint32_t phi4 = ((uint32_t)row & 127u) >> 2;        // 20 n0 = -4 phi4 (mod 128): n0 = 19 phi4 (mod 32)
  return 1u + ((19u * phi4 + 31u) & 31u);
}

// What the walker leaves for the decoders per frame and batch (LDS).  A: {first 16-byte chunk of the piece's bytes (a
// global address), first 128-byte line of its samples}; B.x = blocks | samples of the last of them << 8 | first bit of the
// first block in chunk 0 << 16 | chunks << 24; B.y = bytes of that line in front of the piece | end of the piece << 8 |
// (the piece ends with a full block that is the frame's last: a pending sample) << 20
struct X3BRecA { unsigned long long ga, gl; };

template <uint32_t UNIT, uint32_t UPB>
__global__ void __launch_bounds__(64 * X3B_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5)))
x3_decode_blocks_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                        uint64_t n_frames_arg, uint32_t fpg, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                        int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                        X3FrameMeta* __restrict__ meta, uint32_t* __restrict__ pace, uint32_t pace_epoch,
                        const unsigned long long* __restrict__ d_nf) {
  const uint64_t n_frames = d_nf ? (*d_nf < n_frames_arg ? (uint64_t)*d_nf : n_frames_arg) : n_frames_arg;
  const uint64_t f0 = (uint64_t)blockIdx.x * fpg;   // the group's first frame
  if (f0 >= n_frames) return;
  const uint32_t nfr = (uint32_t)(n_frames - f0 < fpg ? n_frames - f0 : fpg);   // its frames (uniform)

  __shared__ __attribute__((aligned(128))) uint32_t ring[64 * X3_DEC_RING_DW];             // the walker's input ring
  __shared__ __attribute__((aligned(16))) uint16_t desc[2][64 * X3B_DESC_PITCH];            // where the blocks begin
  __shared__ __attribute__((aligned(16))) X3BRecA recA[2][64];
  __shared__ __attribute__((aligned(8))) uint2 recB[2][64];
  __shared__ uint32_t fr_last[64], fr_bad[64];
  __shared__ uint32_t s_nbatch;
  __shared__ uint32_t s_tick[2];   // the decoders' iteration tickets of a batch buffer (the walker clears them)
  __shared__ __attribute__((aligned(128))) uint8_t scratch[X3B_D * X3B_SCRATCH];

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const unsigned long long wall_t0 = wall_clock64();
  const unsigned long long clk_t0 = clock64();
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_start = wall_clock64();
#endif

  if (wave == 0u) {
    // ================================================================================ the walker
    __builtin_amdgcn_s_setprio(X3B_WALKER_PRIO);
    const uint64_t f = f0 + lane;
    bool active = lane < nfr;
    int32_t st = X3D_OK;
    uint32_t samples = 0, plen = 2;
    uint64_t p0 = 0, wo = 0;
    if (active) {
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
          wo = clip * g.clip_stride + (f - clip * g.fpc) * (uint64_t)p.spf;
        }
        if (wo + samples > wav_cap) {
          st = X3D_BAD_ARG;
          active = false;
        }
      }
    }
    if (!active) { p0 = 0; plen = 2; wo = 0; samples = 0; }
    const uint32_t nblk = samples ? (samples - 1u + UNIT - 1u) / UNIT : 0u;   // UNITS of the frame
    const uint64_t rowb = (uint64_t)(uintptr_t)(wav + wo);
    const uint32_t n0 = x3b_first_batch<UNIT, UPB>(rowb);
    const uint32_t nbatch = nblk == 0u ? 0u : (nblk <= n0 ? 1u : 1u + (nblk - n0 + X3B_NB - 1u) / X3B_NB);
    const uint32_t nbatch_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)x3_wave_max_u32(nbatch));
    uint32_t first = 0;
    if (active) {
      first = ((uint32_t)x3[p0] << 8) | x3[p0 + 1];
      if (samples == 1u) wav[wo] = (int16_t)first;
    }
    fr_last[lane] = first;
    fr_bad[lane] = 0u;
    if (lane == 0u) s_nbatch = nbatch_max;
    const uint64_t pay = (uint64_t)(uintptr_t)(x3 + p0);                                   // the payload's first byte
    const uint64_t x3_lastc = ((uint64_t)(uintptr_t)x3 + x3_len - 1u) & ~15ull;          // the last 16-byte chunk that holds stream

    // ---- the input ring (x3_decode_split_kernel.h): 32 dwords per lane in a row of 128 bytes, stream word j in slot
    // ~j & 31 (descending), parked big-endian
    uint32_t* const row = ring + lane * X3_DEC_RING_DW;
    const uint32_t row_base = x3_lds_addr(row);
#define X3B_RING_WORD(j) row[~(j) & 31u]
    const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
    const uint64_t abs_bits = (uint64_t)adj + p0 + 2u;              // the byte behind the first sample
    const uint64_t abs_last = (uint64_t)adj + p0 + plen - 1u;       // the payload's last byte
    const uint64_t abs_base = (abs_bits < abs_last ? abs_bits : abs_last) & ~15ull;
    const uint8_t* __restrict__ const x3b = (x3 - adj) + abs_base;
    const uint32_t v_bits = (uint32_t)(abs_bits - abs_base);        // first block header (0..16), in bytes from the ring's first chunk
    const int32_t v_rel = 16 - 8 * (int32_t)v_bits;                 // payload bit = ring bit + v_rel
    // The ring takes the STREAM as it comes -- behind the payload the next frame's bytes, up to the stream's last 16-byte
    // chunk, which repeats from there on -- and the decoders stage the very same bytes (below): a codeword is then parsed
    // alike by walker and decoders wherever it stands.  (Until the soak of round 6 the walker stopped at the PAYLOAD's last
    // chunk, as the lane-per-frame kernels do; there parser and valuer share one view.  Here a codeword whose zero run
    // began in the payload's last bits was parsed on different bits by the two sides, and the frame was not flagged:
    // tools/r6/repro_overread.py.)  What is read behind the payload still sends the frame to the reference's reader.
    uint32_t v_last;
    {
      const uint64_t lastc_abs = ((uint64_t)adj + x3_len - 1u) & ~15ull;          // (in the coordinates of abs_base)
      const uint64_t rel = lastc_abs > abs_base ? lastc_abs - abs_base : 0u;
      v_last = rel > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)rel;
    }
    uint32_t v_next = 0, wr_abs = 0;
    constexpr uint32_t SVC_MAX = 3u * X3B_PERIOD;
    constexpr uint32_t SVC_AHEAD = X3B_AHEAD;
    auto request = [&](uint32_t v) -> uint4 {
      const uint32_t a = v < v_last ? v : v_last;
      return *reinterpret_cast<const uint4*>(x3b + a);
    };
    auto park = [&](uint4 c) {
      x3_lds_write_b128(x3_and_or(0u - 4u * wr_abs - 16u, 112u, row_base), x3_bswap32(c.w), x3_bswap32(c.z),
                        x3_bswap32(c.y), x3_bswap32(c.x));
      wr_abs += 4;
    };
    const bool dense_grp = __any(active && plen > 9728u);
    uint4 ld[SVC_MAX];
    uint32_t v_req = 0;
    {
      uint4 c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);
#pragma unroll
      for (int k = 0; k < 8; ++k) park(c[k]);
      v_next += 128;
#pragma unroll
      for (int k = 0; k < (int)SVC_AHEAD; ++k) ld[k] = request(v_next + 16u * k);
      if (dense_grp) {
#pragma unroll
        for (int k = (int)SVC_AHEAD; k < (int)SVC_MAX; ++k) ld[k] = request(v_next + 16u * k);
      }
      v_req = v_next;
    }
    auto service = [&](uint32_t widx) {
      const uint32_t used = wr_abs - widx;
      const uint32_t fit = used >= X3_DEC_RING_DW ? 0u : (X3_DEC_RING_DW - used) >> 2;
#pragma unroll
      for (uint32_t k = 0; k < SVC_AHEAD; ++k) {
        if (fit > k) park(ld[k]);
      }
      if (__any(fit > SVC_AHEAD)) {
        if (!dense_grp) {
#pragma unroll
          for (uint32_t k = SVC_AHEAD; k < SVC_MAX; ++k) ld[k] = request(v_req + 16u * k);
        }
#pragma unroll
        for (uint32_t k = SVC_AHEAD; k < SVC_MAX; ++k) {
          if (fit > k) park(ld[k]);
        }
      }
      v_next += 16u * (fit > SVC_MAX ? SVC_MAX : fit);
      v_req = v_next;
#pragma unroll
      for (int k = 0; k < (int)SVC_AHEAD; ++k) ld[k] = request(v_req + 16u * k);
      if (dense_grp) {
#pragma unroll
        for (int k = (int)SVC_AHEAD; k < (int)SVC_MAX; ++k) ld[k] = request(v_req + 16u * k);
      }
    };
    X3_WAVE_LDS_ORDER();
    // window: w0 holds `s` unconsumed bits (its low s bits), then w1; wn is the word behind w1
    const uint32_t a0 = 8u * (v_bits & 3u);
    const uint32_t widx0 = (v_bits >> 2) - (a0 == 0 ? 1u : 0u);
    uint32_t s = (32u - a0) & 31u;
    uint32_t w0 = X3B_RING_WORD(widx0), w1 = X3B_RING_WORD(widx0 + 1u), wn = X3B_RING_WORD(widx0 + 2u);
    uint32_t qb = 4u * ~(widx0 + 2u);   // (a multiple of four: the byte counter of wn's slot before masking)
    auto consume_to = [&](int32_t s2) {
      const uint32_t m = (uint32_t)(s2 >> 31);
      s = (uint32_t)s2 & 31u;
      w0 = x3_bfi(m, w1, w0);
      w1 = x3_bfi(m, wn, w1);
      uint32_t addr;
      asm("v_lshl_add_u32 %0, %2, 2, %0\n\t"
          "v_and_or_b32 %1, %0, %3, %4"
          : "+v"(qb), "=v"(addr) : "v"(m), "v"(124u), "v"(row_base));
      wn = x3_lds_read_b32(addr);
      __builtin_amdgcn_sched_barrier(0);
    };
    // ring index of w0: -qb / 4 - 3; the window's position in payload bits: 32 * index + 32 - s + v_rel
    auto ring_index = [&]() -> uint32_t { return ~((uint32_t)((int32_t)qb >> 2)) - 2u; };
    const uint32_t pos_c = (uint32_t)(v_rel - 64);
    auto position = [&]() -> uint32_t { return pos_c - ((qb << 3) + s); };

    X3B_BARRIER();   // the frames' records are there (and the decoders have read nothing yet)

    const uint32_t cnt_last = nblk ? samples - 1u - UNIT * (nblk - 1u) : 0u;   // samples of the frame's last unit (1..UNIT)
    uint32_t blocks_left = nblk;
    uint32_t bytes_left = 2u * samples;                  // of the row, from the current batch's first block on
    uint64_t G = rowb;                                   // where the current batch's samples go
    bool bad = false;
    uint32_t it = 0;    // blocks walked by the wave (the service's clock)
    for (uint32_t k = 0; k < nbatch_max; ++k) {
      const uint32_t buf = k & 1u;
      const uint32_t quota = k ? X3B_NB : n0;
      const uint32_t nbk = blocks_left < quota ? blocks_left : quota;
      blocks_left -= nbk;
      const bool is_last = blocks_left == 0u;           // the batch holds the frame's last block (or the frame is through)
      const uint32_t lastcnt = is_last ? cnt_last : UNIT;     // samples of the batch's last unit
      const uint32_t maxb = (uint32_t)__builtin_amdgcn_readfirstlane((int)x3_wave_max_u32(nbk));
      // (only a frame's last block can be short: batches without one take the ten-pair path without looking)
      const bool shorts = __any(nbk != 0u && lastcnt != UNIT);
      const uint32_t base = position();
      const uint32_t rel_c = pos_c - base;
      uint16_t* const dq = &desc[buf][lane * X3B_DESC_PITCH];
      if (lane == 0u) s_tick[buf] = 0u;
      uint32_t zmask0 = 0, width = 0;   // of the block the current unit belongs to
      for (uint32_t b = 0; b < maxb; ++b, ++it) {
        const uint32_t cnt = b < nbk ? (b + 1u == nbk ? lastcnt : UNIT) : 0u;
        dq[b] = (uint16_t)(rel_c - ((qb << 3) + s));
        X3_STAMP(0);
        // (every 40 samples walked, whatever a unit is: units of 10 are served every fourth)
        if (!(X3B_KO & 32) && (it % (X3B_PERIOD * (20u / UNIT))) == 0u) service(ring_index());
        X3_STAMP(1);
        const uint32_t live = cnt ? 0xFFFFFFFFu : 0u;
        if (UPB == 1u || (b % UPB) == 0u) {   // (batches are whole blocks in every lane: the header's turn is the wave's)
          const uint32_t hdr = __builtin_amdgcn_alignbit(w0, w1, s) >> 26;
          const uint32_t ftype = hdr >> 4;
          zmask0 = (uint32_t)((int32_t)(15u - hdr) >> 31);                      // all ones for Rice
          consume_to((int32_t)(s + (live & ((zmask0 & 4u) - 6u))));             // 6 header bits for BFP, 2 for Rice
          width = x3_bfi(zmask0, (1u << ftype) >> 1, (hdr & 15u) + 1u);         // Rice 1, 2, 4; BFP E
        }
        const uint32_t zmask = zmask0 & live;
        const uint32_t nwidth = (0u - width) & live;
        X3_STAMP(2);
        if (X3B_KO & 2) {
        } else if (!shorts || __all(cnt == UNIT || cnt == 0u)) {
          uint32_t t_, z_, n1_, wnb_;
          if (UNIT == 20u) {
            asm volatile(
                X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb") X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb")
                X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb") X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb")
                X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb")
                "s_waitcnt lgkmcnt(0)"
                : [w0] "+v"(w0), [w1] "+v"(w1), [wn] "+v"(wn), [wnb] "=&v"(wnb_), [s] "+v"(s), [qb] "+v"(qb),
                  [t] "=&v"(t_), [z] "=&v"(z_), [n1] "=&v"(n1_)
                : [zm] "v"(zmask), [nw] "v"(nwidth), [c124] "v"(124u), [rowb] "v"(row_base)
                : "memory");
          } else {
            // (five pairs: the word behind the window ends up in wnb)
            asm volatile(
                X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb") X3B_WPAIR("wnb", "wn") X3B_WPAIR("wn", "wnb")
                X3B_WPAIR("wnb", "wn")
                "s_waitcnt lgkmcnt(0)"
                : [w0] "+v"(w0), [w1] "+v"(w1), [wn] "+v"(wn), [wnb] "=&v"(wnb_), [s] "+v"(s), [qb] "+v"(qb),
                  [t] "=&v"(t_), [z] "=&v"(z_), [n1] "=&v"(n1_)
                : [zm] "v"(zmask), [nw] "v"(nwidth), [c124] "v"(124u), [rowb] "v"(row_base)
                : "memory");
            wn = wnb_;
          }
        } else {
          for (uint32_t j = 0; j < UNIT; ++j) {
            const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
            const uint32_t z = x3_ffbh(t) & zmask;
            const uint32_t nn = nwidth - z;
            consume_to((int32_t)(s + (j < cnt ? nn : 0u)));
          }
        }
        X3_STAMP(3);
      }
      // ---- the batch's record
      {
        const uint32_t end = position();
        // the frame ends here: was it read beyond its payload?  (x3_decode_replay.h: the reference's reader knows about
        // the zeros there; a lane whose frame is through consumes nothing more)
        if (nbk && is_last && (int32_t)end > (int32_t)(8u * plen)) bad = true;
        const uint32_t span = end - base;
        uint32_t nb_ok = nbk;
        if (nbk && span > X3B_SPAN_MAX) {   // not what 32 valid blocks can take: the reference's reader decides
          bad = true;
          nb_ok = 0u;
        }
        const uint64_t gp = pay + (base >> 3);
        const uint32_t off16 = (uint32_t)gp & 15u;
        const uint64_t ga = gp - off16;
        const uint32_t c0 = 8u * off16 + (base & 7u);
        uint32_t nchunk = (((c0 + span + 7u) >> 3) + 15u) / 16u + 1u;
        // (a frame that was walked beyond its payload -- a header that asks for more samples than the payload holds -- can
        // stand behind the END OF THE STREAM: the ring repeats its last chunk there, the decoders must not ask for anything)
        if (ga > x3_lastc) {
          if (nbk) bad = true;
          nb_ok = 0u;
        }
        if (nchunk > X3B_IN_CHUNKS) nchunk = X3B_IN_CHUNKS;
        // (chunks behind the stream's last one repeat it, as in the walker's ring: the decoders take such a piece -- the last
        // frames of a stream -- chunk by chunk with clamped addresses)
        const bool clamped = ga <= x3_lastc && (uint64_t)nchunk > ((x3_lastc - ga) >> 4) + 1u;
        const uint32_t PH = (uint32_t)G & 127u;
        const uint32_t bytes = is_last ? bytes_left : 2u * UNIT * nbk;
        X3BRecA ra;
        ra.ga = ga;
        ra.gl = G - PH;
        uint2 rb = make_uint2(0u, 0u);
        if (nb_ok) {
          rb.x = nb_ok | (lastcnt << 8) | (c0 << 16) | (nchunk << 24);
          rb.y = PH | ((PH + bytes) << 8) | ((is_last && lastcnt == UNIT) ? (1u << 20) : 0u) | (clamped ? (1u << 21) : 0u);
        }
        recA[buf][lane] = ra;
        recB[buf][lane] = rb;
        G += (uint64_t)(2u * UNIT) * nbk;
        bytes_left -= 2u * UNIT * nbk < bytes_left ? 2u * UNIT * nbk : bytes_left;
      }
      X3_STAMP(5);
      X3B_BARRIER();
      X3_STAMP(4);
    }
    X3B_BARRIER();   // the decoders are through the last batch
    if (lane < nfr) {
      if (active && (bad || fr_bad[lane] != 0u)) st = X3D_REPLAY;
      status[f] = st;
    }
    if (lane == 0u && blockIdx.x == 0u) {
      // the launch log (x3_ctx_launch_log): group 0's shader ticks against the 100 MHz clock
      uint32_t* const lg = pace + X3_LOG_BASE + X3_LOG_WORDS * (pace_epoch & (X3_LOG_ENTRIES - 1u));
      const unsigned long long dt = clock64() - clk_t0;
      uint64_t t16 = nblk ? (dt * 16u) / nblk : 0u;
      if (t16 >= (1u << 20)) t16 = (1u << 20) - 1u;
      lg[0] = ((pace_epoch & 0xFFFu) << 20) | (uint32_t)t16;
      lg[1] = ((pace_epoch & 0xFFFu) << 20);
      lg[2] = (uint32_t)dt;
      lg[3] = (uint32_t)(wall_clock64() - wall_t0);
    }
#undef X3B_RING_WORD
  } else {
    // ================================================================================ the decoders
    constexpr uint32_t PAIRS = UNIT / 2u;
    const uint32_t dw = wave - 1u;
    const uint32_t h = lane >> 5, j = lane & 31u;
    uint8_t* const my = scratch + dw * X3B_SCRATCH;
    const uint32_t in_base = x3_lds_addr(my) + h * X3B_IN_PITCH;      // (2 * 1392 <= 2 * 1408)
    const uint32_t in_top = in_base + X3B_IN_PITCH - 20u;             // LDS byte address of stream word 0 (words descend)
    const uint32_t out_base = x3_lds_addr(my) + h * X3B_OUT_PITCH;
    const uint64_t x3_lastc = ((uint64_t)(uintptr_t)x3 + x3_len - 1u) & ~15ull;   // the last 16-byte chunk that holds stream
    const uint32_t k_tab = (p.k[1] << 16) | (p.k[2] << 24);           // log2(level) by ftype
    const uint32_t bound_tab = (p.inv_len[0] << 8) | (p.inv_len[1] << 16) | (p.inv_len[2] << 24);
    X3B_BARRIER();
    const uint32_t nbatch_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_nbatch);
    const uint32_t niter = (nfr + 1u) >> 1;
    for (uint32_t k = 0; k < nbatch_max; ++k) {
      X3_STAMP(0);
      X3B_BARRIER();   // the walker is through batch k
      X3_STAMP(4);
      const uint32_t buf = k & 1u;
      // the first chunk of a piece's bytes per lane, requested an iteration ahead
      auto fetch0 = [&](uint32_t i) -> uint4 {
        const uint32_t fl = 2u * i + h;
        const uint2 b = recB[buf][fl];
        const uint32_t nchunk = b.x >> 24;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (j < nchunk && !(b.y & (1u << 21)) && !(X3B_KO & 8)) q = x3b_global_load16(recA[buf][fl].ga + 16u * j);
        return q;
      };
      // iterations go to whichever decoder wave is free (the waves of a group sit on different SIMDs, whose other
      // tenants differ: a fixed share would make the group as slow as its slowest wave)
      auto ticket = [&]() -> uint32_t {
        uint32_t t = 0;
        if (lane == 0u) t = atomicAdd(&s_tick[buf], 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      };
      uint32_t i = (X3B_KO & 1) ? niter : ticket();
      uint4 q0 = make_uint4(0u, 0u, 0u, 0u);
      if (i < niter) q0 = fetch0(i);
      const uint32_t desc_lane = x3_lds_addr(&desc[buf][0]) + 2u * (h * X3B_DESC_PITCH + j);
      while (i < niter) {
        const uint32_t inext = ticket();
        const uint32_t fl = 2u * i + h;    // (<= 63; a frame that is not there has an empty record)
        const X3BRecA ra = recA[buf][fl];
        const uint2 rb = recB[buf][fl];
        const uint32_t nbk = rb.x & 0xFFu, lastcnt = (rb.x >> 8) & 0xFFu, c0 = (rb.x >> 16) & 0xFFu, nchunk = rb.x >> 24;
        const uint32_t PH = rb.y & 0xFFu, EB = (rb.y >> 8) & 0xFFFu;
        const bool has = j < nbk;
        const uint32_t cnt = has ? (j + 1u == nbk ? lastcnt : UNIT) : 0u;

        // ---- the piece's bytes into LDS: 16-byte chunks from the aligned address in front of its first bit; words descend:
        // chunk c's words 4c .. 4c+3 at in_top - 4 * (4c + i)
        X3_WAVE_LDS_ORDER();
        const bool clamped = (rb.y & (1u << 21)) != 0u;
        if (j < nchunk && !clamped)
          x3_lds_write_b128(in_top - 16u * j - 12u, x3_bswap32(q0.w), x3_bswap32(q0.z), x3_bswap32(q0.y), x3_bswap32(q0.x));
        if (__any(clamped)) {   // a piece that reaches behind the stream's last chunk (that one repeats): chunk by chunk
#pragma unroll
          for (uint32_t r = 0; r < 3u; ++r) {
            const uint32_t c = j + 32u * r;
            if (clamped && c < nchunk) {
              uint64_t a = ra.ga + 16u * c;
              if (a > x3_lastc) a = x3_lastc;
              const uint4 q = x3b_global_load16(a);
              x3_lds_write_b128(in_top - 16u * c - 12u, x3_bswap32(q.w), x3_bswap32(q.z), x3_bswap32(q.y), x3_bswap32(q.x));
            }
          }
        }
        if (__any(nchunk > 32u && !clamped) && !(X3B_KO & 8)) {   // long pieces (BFP / literal blocks): the rest at once
#pragma unroll
          for (uint32_t r = 1; r < 3u; ++r) {
            const uint32_t c = j + 32u * r;
            if (c < nchunk && !clamped) {
              const uint4 q = x3b_global_load16(ra.ga + 16u * c);
              x3_lds_write_b128(in_top - 16u * c - 12u, x3_bswap32(q.w), x3_bswap32(q.z), x3_bswap32(q.y), x3_bswap32(q.x));
            }
          }
        }
        X3_WAVE_LDS_ORDER();
        if (inext < niter) q0 = fetch0(inext);
        X3_STAMP(0);

        // ---- this lane's unit: its first bit, counted from chunk 0 (the walker has checked the batch's span: a unit
        // begins at most X3B_SPAN_MAX bits in).  The unit's BLOCK begins ub units earlier, in the same piece: its header is there.
        const uint32_t ub = UPB == 1u ? 0u : j % UPB;
        const uint32_t rel_u = has ? x3_lds_read_u16(desc_lane + 2u * (2u * X3B_DESC_PITCH) * i, 0u) : 0u;
        const uint32_t rel = UPB == 1u ? rel_u : (has ? x3_lds_read_u16(desc_lane + 2u * (2u * X3B_DESC_PITCH) * i - 2u * ub, 0u) : 0u);
        uint32_t s, qa, w0, w1, wn;
        auto window_at = [&](uint32_t bbit) {
          s = (0u - bbit) & 31u;
          const int32_t jdx = (int32_t)(bbit - 1u) >> 5;           // the word w0 is in (-1: a word that is read and not used)
          qa = in_top - 8u - 4u * (uint32_t)jdx;                   // LDS address of wn, two words on
          w0 = x3_lds_read_b32(qa + 8u); w1 = x3_lds_read_b32(qa + 4u); wn = x3_lds_read_b32(qa);
        };
        window_at(c0 + rel);
        auto consume_to = [&](int32_t s2) {
          const uint32_t m = (uint32_t)(s2 >> 31);
          s = (uint32_t)s2 & 31u;
          w0 = x3_bfi(m, w1, w0);
          w1 = x3_bfi(m, wn, w1);
          asm("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(qa) : "v"(m));   // a shift: the next word is four bytes down
          wn = x3_lds_read_b32(qa);
          __builtin_amdgcn_sched_barrier(0);
        };
        // block header and parameters (decoder.rs:138-144, 209-216), as arithmetic on the six bits
        const uint32_t hdr = __builtin_amdgcn_alignbit(w0, w1, s) >> 26;
        const uint32_t ftype = hdr >> 4;
        const uint32_t E = (hdr & 15u) + 1u;
        const uint32_t zmask = (uint32_t)((int32_t)(15u - hdr) >> 31);     // all ones for Rice
        if (UPB == 1u) {
          consume_to((int32_t)(s + ((zmask & 4u) - 6u)));
        } else {
          // (the block's first unit goes on behind the header; a later one begins where the walker left its mark)
          const uint32_t behind = (c0 + rel) + 6u - (zmask & 4u);
          window_at(ub ? c0 + rel_u : behind);
        }
        const uint32_t width = x3_bfi(zmask, (1u << ftype) >> 1, E);
        const uint32_t kk = (k_tab >> (8u * ftype)) & 0xFFu;
        const uint32_t fw = x3_bfi(zmask, kk, width);
        const uint32_t lsh = x3_bfi(zmask, kk, 31u);
        const uint32_t nwidth = 0u - width;
        const bool bfp = zmask == 0u;
        const uint32_t litmask = cnt ? (~zmask & (uint32_t)((int32_t)(14u - (hdr & 15u)) >> 31)) : 0u;  // E == 16
        const uint32_t neg_thresh = ~zmask & (1u << (E - 1u));
        const uint32_t neg2 = (neg_thresh << 1) & ~litmask;
        const uint32_t bound = x3_bfi(zmask, (bound_tab >> (8u * ftype)) & 0xFFu, 0xFFFFFFFFu);
        const uint32_t tm12 = ((neg_thresh - 1u) & 0xFFFFu) * 0x10001u;
        const uint32_t neg22 = (neg2 & 0xFFFFu) * 0x10001u;
        const bool full = __all(cnt == UNIT || cnt == 0u);
        const bool any_lit = __any(litmask != 0u);
        X3_STAMP(1);
        uint32_t maxii2 = 0, prevP = 0;
        uint32_t W[PAIRS];
        if (X3B_KO & 16) {
#pragma unroll
          for (uint32_t r = 0; r < PAIRS; ++r) W[r] = w0 + r;
        } else if (full && !any_lit) {
          const uint32_t zsh2 = zmask & 0x00010001u;
          uint32_t wnb_, pb_, t_, t2_, z1_, z2_, n1_, n2_, v1_, v2_;
          asm volatile(
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W0") X3B_DPAIR("wn", "wnb", "pb", "pa", "W1")
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W2") X3B_DPAIR("wn", "wnb", "pb", "pa", "W3")
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W4")
              "s_waitcnt lgkmcnt(0)"
              : [w0] "+v"(w0), [w1] "+v"(w1), [wn] "+v"(wn), [wnb] "=&v"(wnb_), [s] "+v"(s), [qa] "+v"(qa), [pa] "+v"(prevP),
                [pb] "=&v"(pb_), [mx] "+v"(maxii2), [W0] "=&v"(W[0]), [W1] "=&v"(W[1]), [W2] "=&v"(W[2]), [W3] "=&v"(W[3]),
                [W4] "=&v"(W[4]), [t] "=&v"(t_), [t2] "=&v"(t2_), [z1] "=&v"(z1_), [z2] "=&v"(z2_), [n1] "=&v"(n1_),
                [n2] "=&v"(n2_), [v1] "=&v"(v1_), [v2] "=&v"(v2_)
              : [zm] "v"(zmask), [nw] "v"(nwidth), [fw] "v"(fw), [lsh] "v"(lsh), [tm12] "v"(tm12), [neg22] "v"(neg22),
                [zsh2] "v"(zsh2), [sel] "s"(0x05040100u), [c1] "s"(0x00010000u)
              : "memory");
          // (five pairs: the window is wnb / wn swapped an odd number of times -- the word behind it is in wnb, the last pair in pb)
          wn = wnb_;
          prevP = pb_;
          if (UNIT == 20u) asm volatile(
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W0") X3B_DPAIR("wn", "wnb", "pb", "pa", "W1")
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W2") X3B_DPAIR("wn", "wnb", "pb", "pa", "W3")
              X3B_DPAIR("wnb", "wn", "pa", "pb", "W4")
              "s_waitcnt lgkmcnt(0)"
              : [w0] "+v"(w0), [w1] "+v"(w1), [wn] "+v"(wn), [wnb] "=&v"(wnb_), [s] "+v"(s), [qa] "+v"(qa), [pa] "+v"(prevP),
                [pb] "=&v"(pb_), [mx] "+v"(maxii2), [W0] "=&v"(W[PAIRS - 5]), [W1] "=&v"(W[PAIRS - 4]), [W2] "=&v"(W[PAIRS - 3]), [W3] "=&v"(W[PAIRS - 2]),
                [W4] "=&v"(W[PAIRS - 1]), [t] "=&v"(t_), [t2] "=&v"(t2_), [z1] "=&v"(z1_), [z2] "=&v"(z2_), [n1] "=&v"(n1_),
                [n2] "=&v"(n2_), [v1] "=&v"(v1_), [v2] "=&v"(v2_)
              : [zm] "v"(zmask), [nw] "v"(nwidth), [fw] "v"(fw), [lsh] "v"(lsh), [tm12] "v"(tm12), [neg22] "v"(neg22),
                [zsh2] "v"(zsh2), [sel] "s"(0x05040100u), [c1] "s"(0x00010000u)
              : "memory");
          if (UNIT == 20u) prevP = pb_;
        } else {
          // a frame's last unit, or a literal block in some lane: pair by pair (as x3_decode_split_kernel's valuer)
#pragma unroll
          for (uint32_t r = 0; r < PAIRS; ++r) {
            const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
            uint32_t z1, v1, z2, v2, ns;
            X3B_PAIR_FIELDS(t, zmask, nwidth, fw, z1, v1, z2, v2, ns);
            consume_to((int32_t)(s + ns));
            uint32_t X = x3_pack_lo16((z1 << lsh) + v1, (z2 << lsh) + v2);
            // (the samples behind a short block's end count for nothing)
            X &= (2u * r < cnt ? 0xFFFFu : 0u) | (2u * r + 1u < cnt ? 0xFFFF0000u : 0u);
            // Rice: X = i, the index into the inverse table (decoder.rs:186), a zigzag (x3.rs:200-204); BFP:
            // unsigned_to_i16 (decoder.rs:198-207), strict compare; literal: the field is the sample
            maxii2 = x3_pk_max_u16(maxii2, X);
            const uint32_t R = x3_pk_lshr_b16_1(X) ^ x3_pk_sub_u16(0u, X & 0x00010001u);
            const uint32_t B = x3_pk_sub_u16(X, x3_pk_add_u16(X, tm12) & neg22);
            const uint32_t D = x3_bfi(zmask, R, B);
            uint32_t P = x3_pk_mad_u16_alo(D, 0x00010000u, x3_pk_add_u16_bhi(D, prevP));   // (last + d1, last + d1 + d2)
            P = x3_bfi(litmask, X, P);
            W[r] = __builtin_amdgcn_alignbit(P, prevP, 16);   // (the sample in front, this pair's first)
            prevP = P;
          }
        }
        X3_STAMP(2);
        if (cnt && ((bfp && E <= 5u) || max(maxii2 & 0xFFFFu, maxii2 >> 16) >= bound)) fr_bad[fl] = 1u;

        // ---- the sample in front of every block: a scan over the 32 blocks of the piece.  A block is the map
        // x -> lit ? T : x + T (T = its last sample counted from zero); maps compose as (L, T) after (L', T') =
        // (L | L', L ? T : T' + T).  Identity for lanes without a block.
        uint32_t sT = cnt ? (prevP >> 16) : 0u;   // (a short last block's is not its last sample; nothing follows it)
        const uint32_t last_in = fr_last[fl];
        uint32_t front, last_out;
        if (!any_lit) {
#define X3B_ADD_STEP(ctrl, rmask) sT += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sT, ctrl, rmask, 0xF, false);
          X3B_ADD_STEP(0x111, 0xF) X3B_ADD_STEP(0x112, 0xF) X3B_ADD_STEP(0x114, 0xF) X3B_ADD_STEP(0x118, 0xF)
          X3B_ADD_STEP(0x142, 0xA)   // row_bcast:15 -> rows 1, 3: the halves' second rows take their first row's total
#undef X3B_ADD_STEP
          uint32_t xT = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sT, 0x138, 0xF, 0xF, false);   // wave_shr:1
          if (j == 0u) xT = 0u;
          front = (last_in + xT) & 0xFFFFu;
          last_out = (last_in + sT) & 0xFFFFu;
        } else {
          uint32_t sL = litmask;
#define X3B_SCAN_STEP(ctrl, rmask)                                                                   \
          {                                                                                          \
            const uint32_t pT = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sT, ctrl, rmask, 0xF, false); \
            const uint32_t pL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sL, ctrl, rmask, 0xF, false); \
            sT = (pT & ~sL) + sT;                                                                    \
            sL = sL | pL;                                                                            \
          }
          X3B_SCAN_STEP(0x111, 0xF) X3B_SCAN_STEP(0x112, 0xF) X3B_SCAN_STEP(0x114, 0xF) X3B_SCAN_STEP(0x118, 0xF)
          X3B_SCAN_STEP(0x142, 0xA)
#undef X3B_SCAN_STEP
          uint32_t xT = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sT, 0x138, 0xF, 0xF, false);   // wave_shr:1
          uint32_t xL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sL, 0x138, 0xF, 0xF, false);
          if (j == 0u) { xT = 0u; xL = 0u; }
          front = ((xL ? 0u : last_in) + xT) & 0xFFFFu;       // the sample in front of this lane's block
          last_out = ((sL ? 0u : last_in) + sT) & 0xFFFFu;
        }
        X3_WAVE_LDS_ORDER();
        if (j == 31u && nbk) fr_last[fl] = last_out;
        const uint32_t add2 = litmask ? 0u : front * 0x10001u;
        W[0] = x3_pk_add_u16(W[0], litmask ? front : add2);
#pragma unroll
        for (uint32_t r = 1; r < PAIRS; ++r) W[r] = x3_pk_add_u16(W[r], add2);

        // ---- out: the piece's samples through LDS, then whole lines.  LDS byte = destination byte modulo 128.
        X3_WAVE_LDS_ORDER();
        if (has) {
          const uint32_t oa = out_base + PH + 2u * UNIT * j;
          if (UNIT == 20u) {   // (rows on 8-byte boundaries)
#pragma unroll
            for (uint32_t r = 0; r + 1u < PAIRS; r += 2) x3_lds_write_b64(oa + 4u * r, W[r], W[r + 1u]);
          } else {             // (rows on 4-byte boundaries)
#pragma unroll
            for (uint32_t r = 0; r < PAIRS; ++r) x3b_lds_write_b32(oa + 4u * r, W[r]);
          }
          // a full last unit of the frame: its last sample is nobody's "sample in front"
          if ((rb.y >> 20) && j + 1u == nbk)
            x3b_lds_write_b32(oa + 2u * UNIT, ((litmask ? 0u : front) + (prevP >> 16)) & 0xFFFFu);
        }
        X3_WAVE_LDS_ORDER();
        X3_STAMP(3);
        // bytes [PH, EB) of the staging are the piece
        uint8_t* const GL = reinterpret_cast<uint8_t*>(ra.gl);
        if (UNIT == 20u && __all((rb.y & 0xFFFFFu) == (1280u << 8))) {
          // the usual piece: ten whole lines, in both halves
          const x3_u32x4 v0 = x3_lds_read_b128(out_base + 16u * j);
          const x3_u32x4 v1 = x3_lds_read_b128(out_base + 16u * j + 512u);
          if (!(X3B_KO & 4)) {
            x3b_global_store16_nt(GL + 16u * j, v0);
            x3b_global_store16_nt(GL + 16u * j + 512u, v1);
          }
          if (j < 16u) {
            const x3_u32x4 v2 = x3_lds_read_b128(out_base + 16u * j + 1024u);
            if (!(X3B_KO & 4)) x3b_global_store16_nt(GL + 16u * j + 1024u, v2);
          }
        } else if (UNIT == 10u && __all((rb.y & 0xFFFFFu) == (640u << 8))) {
          // ... five whole lines
          const x3_u32x4 v0 = x3_lds_read_b128(out_base + 16u * j);
          if (!(X3B_KO & 4)) x3b_global_store16_nt(GL + 16u * j, v0);
          if (j < 8u) {
            const x3_u32x4 v1 = x3_lds_read_b128(out_base + 16u * j + 512u);
            if (!(X3B_KO & 4)) x3b_global_store16_nt(GL + 16u * j + 512u, v1);
          }
        } else {
#pragma unroll
          for (uint32_t r = 0; r < 3u; ++r) {
            const uint32_t lo = 16u * (j + 32u * r);
            if (lo < EB && lo + 16u > PH) {
              if (lo >= PH && lo + 16u <= EB) {
                if (!(X3B_KO & 4)) x3b_global_store16_nt(GL + lo, x3_lds_read_b128(out_base + lo));
              } else {
                // the two ends of a row: sample by sample
                const uint32_t from = lo > PH ? lo : PH, to = lo + 16u < EB ? lo + 16u : EB;
                for (uint32_t bb = from; bb < to; bb += 2u)
                  x3b_global_store2(GL + bb, x3_lds_read_u16(out_base + bb, 0u));
              }
            }
          }
        }
        X3_WAVE_LDS_ORDER();
        X3_STAMP(5);
        i = inext;
      }
    }
    X3B_BARRIER();   // (the walker reads fr_bad behind this one)
  }
#ifdef X3_DBG_STAMPS
  dbg_acc[6] = dbg_start;
  dbg_acc[7] = wall_clock64();
  if (lane == 0 && blockIdx.x < 2048)
    for (int q = 0; q < 8; ++q) x3_dbg[(blockIdx.x * 4 + wave) * 8 + q] = dbg_acc[q];
#endif
}
