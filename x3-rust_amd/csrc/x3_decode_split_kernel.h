// x3_decode_split_kernel.h -- the lane-per-frame decoder of x3_decode_fast_kernel split over TWO waves.
//
// Why: a frame is one serial bit stream, so decode parallelism is frames (69 120 in config 3 = 1 080
// waves, about one per SIMD), and ONE wave can issue a VALU instruction only every ~5-8 cycles however
// idle its SIMD is (tools/ubench/issue_cost.hip: 8.5 cycles dependent, 5.3 with four independent
// chains; the SIMD itself sustains one every ~3).  The time of x3_decode_fast_kernel is therefore
// (instructions per sample) x (that latency), with more than half of every SIMD unused.  Here each group
// of 64 frames gets a workgroup of two waves on two SIMDs:
//
//   wave 0, the PARSER: owns the input ring and the bit window.  Per block it reads the 6 header bits and
//     walks the codewords: for every sample the zero run z and the field v behind it, two samples per
//     32-bit peek.  It does not compute a single sample value; it hands over i = (z << k) + r (the index
//     into the reference's inverse Rice table, or simply the field for BFP/literal blocks),
//     two 16-bit values per dword, through a double-buffered LDS block buffer, and the header bits.
//   wave 1, the VALUER: turns indices into differences (zigzag / unsigned_to_i16) and samples (running sum,
//     in packed 16-bit arithmetic), checks the table bounds, stages the samples in LDS and flushes them to
//     HBM as 16-byte pieces of contiguous runs, and owns status and metadata.
//
// One s_barrier per 20-sample block: behind barrier k the parser works on block k+1 while the valuer
// consumes block k.  Both waves derive the per-lane block sizes from the frame header alone, so their
// loop trip counts agree whatever the payload holds; a lane whose frame fails (BFP exponent, table
// bound) is only marked dead in the valuer -- the parser keeps walking its bits (lanes are independent,
// and every read is bounded by the ring), the valuer stops storing for it.
//
// Geometry: block_len = 20 (ten pairs per block), staging window = 4 blocks = 80 samples, output frames
// 16-byte aligned (the host launches x3_decode_fast_kernel otherwise).  Same results as the fast kernel.
//
// The samples go out with NON-TEMPORAL stores (x3_store_stream16).  With plain stores the kernel moved 1.20x its
// algorithmic bytes (profiles/r1: WRITE_SIZE 1.17x the samples, FETCH_SIZE 1.29x the stream): the 160-byte runs
// of a flush leave partially written lines in L2, which the streaming input evicts half-done and which evict the
// input's lines in turn.  Streaming stores do neither: 1.025x / 1.03x, same kernel time (profiles/r2).  (A flush
// that writes whole 64-byte-aligned chunks from a destination-indexed ring was built and measured as well: the
// same traffic once its stores were non-temporal, 5 % slower for its address arithmetic -- not kept.)
#pragma once
#include "x3_decode_kernel.h"

#define X3S_BL 20u             // block length served by this kernel
#define X3S_PAIRS 10u
#ifndef X3S_WBLK
#define X3S_WBLK 4u            // blocks staged per lane between flushes (a power of two)
#endif
#define X3S_WIN (X3S_BL * X3S_WBLK)   // samples per staging window
#define X3S_WPIECES (X3S_WIN / 8u)     // 16-byte pieces per row and window
#define X3S_OUT_STRIDE (X3S_WIN / 2u + 4u)  // dwords per staging row: 16-byte aligned rows, spread over the banks
#ifndef X3S_PERIOD
#define X3S_PERIOD 2u         // the ring is topped up every X3S_PERIOD blocks (1 or 2)
#endif
#ifndef X3S_AHEAD
#define X3S_AHEAD 3u          // 16-byte chunks per lane requested one service ahead (of up to 6 per service)
#endif
#define X3S_XROWS 11u          // transfer rows per block buffer: 10 pair dwords + the header word

// halfword index of sample j (0..19) of a block in its transfer buffer: pair j/2 is dword (j/2 & 1) of the
// 8-byte slot of this lane in row j/4
__device__ __forceinline__ uint32_t x3s_half_index(uint32_t j, uint32_t lane) {
  return 2u * (((j >> 2) * 64u + lane) * 2u + ((j >> 1) & 1u)) + (j & 1u);
}

// LDS barrier of the two waves: LDS operations retired, nothing else waited for
#define X3S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ void __launch_bounds__(128)
x3_decode_split_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                       uint64_t n_frames, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                       int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                       X3FrameMeta* __restrict__ meta) {
  // input ring, 32 dwords per lane in rows of exactly 128 bytes at 128-byte aligned addresses, stream word j in
  // slot ~j & 31 (descending): the address of a word is then ONE v_and_or_b32 on a byte counter that a shift of
  // the window decrements with one v_lshl_add_u32.  (Lanes are at different places in their rows, so the aligned
  // rows do not line the reads up on one bank.)
  __shared__ __attribute__((aligned(128))) uint32_t ring[64 * X3_DEC_RING_DW];
  __shared__ __attribute__((aligned(16))) uint32_t outs[64 * X3S_OUT_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t xfer[2 * X3S_XROWS * 64];
  __shared__ unsigned long long s_wo[64];
  __shared__ uint32_t s_ns[64];
  __shared__ uint32_t s_over[64];  // parser -> valuer: the frame was read beyond its payload (x3_decode_replay.h)

  const uint32_t lane = threadIdx.x & 63u;
  const bool parser = threadIdx.x < 64u;
  const uint64_t f = (uint64_t)blockIdx.x * 64 + lane;
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_start = wall_clock64();
#endif

  // ---- per-lane frame setup, done by both waves (same checks as the fast kernel)
  bool active = f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 2;
  uint64_t p0 = 0, wo = 0;
  if (active) {
    uint32_t pcrc_unused;
    st = x3_frame_header_check(reinterpret_cast<const uint32_t*>(x3 - (reinterpret_cast<uintptr_t>(x3) & 3u)),
                               (x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u) + 3) >> 2,
                               x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u),
                               frame_off[f] + (reinterpret_cast<uintptr_t>(x3) & 3u), plen, samples, pcrc_unused);
    if (!parser) {
      meta[f].payload_len = plen;
      meta[f].samples = samples;
    }
    p0 = frame_off[f] + 20;
    if (st != X3D_OK) {
      active = false;
    } else if (samples == 0 || plen < 2) {
      st = X3D_BAD_ARG;
      active = false;
    } else {
      if (wav_off) {  // (the caller vouches for multiples of eight samples: 16-byte aligned output rows)
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
  if (!active) { p0 = 0; plen = 2; wo = 0; samples = 0; }
  // blocks of this lane's frame and of the longest frame of the group: the loop both waves run
  const uint32_t nblk = samples ? (samples - 1u + X3S_BL - 1u) / X3S_BL : 0u;
  const uint32_t nblk_max = __builtin_amdgcn_readfirstlane(x3_wave_max_u32(nblk));
  uint32_t remaining = samples ? samples - 1u : 0u;

  if (parser) {
    // ================================================================= wave 0: parser
    uint32_t* const row = ring + lane * X3_DEC_RING_DW;
    const uint32_t row_base = (uint32_t)(uintptr_t)row;  // LDS byte address of the row (low 7 bits zero)
    // words are parked BIG-ENDIAN
    const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
    // offsets are relative to this lane's first 16-byte chunk (a frame is < 64 KB): a 64-bit pointer per lane,
    // 32-bit arithmetic on everything else, streams of any length
    const uint64_t abs_bits = (uint64_t)adj + p0 + 2u;
    const uint8_t* __restrict__ const x3b = (x3 - adj) + (abs_bits & ~15ull);
    const uint32_t v_bits = (uint32_t)(abs_bits & 15u);      // first block header
    const uint32_t v_end = v_bits - 2u + plen;                 // end of the payload
    const uint32_t v_last = (v_end - 1u) & ~15u;               // last 16-byte chunk that holds payload
    uint32_t v_next = 0;
    uint32_t wr_abs = 0;
    auto request = [&](uint32_t v) -> uint4 {
      const uint32_t a = v < v_last ? v : v_last;
      return *reinterpret_cast<const uint4*>(x3b + a);
    };
    // `near_end`: wave-uniform, some lane of the wave may be within the chunks of this service of the end of its
    // payload (bytes behind the payload are parked as zeros, bitreader.rs:34-48)
    auto park = [&](uint4 c, uint32_t v, bool near_end) {
      uint32_t w[4] = {c.x, c.y, c.z, c.w};
      if (near_end) {
        const int32_t left = (int32_t)(v_end - v);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int32_t r = left - 4 * d;
          w[d] = r >= 4 ? w[d] : (r <= 0 ? 0u : (w[d] & ((1u << (8u * (uint32_t)r)) - 1u)));
        }
      }
      // words wr_abs .. wr_abs+3 -> slots ~wr_abs & 31 downwards = the aligned 16-byte block at slot ~(wr_abs+3) & 31:
      // byte offset (-4 * wr_abs - 16) & 112 of the 128-byte aligned row
      x3_lds_write_b128(x3_and_or(0u - 4u * wr_abs - 16u, 112u, row_base), x3_bswap32(w[3]), x3_bswap32(w[2]),
                        x3_bswap32(w[1]), x3_bswap32(w[0]));
      wr_abs += 4;
    };
    {
      uint4 c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);
#pragma unroll
      for (int k = 0; k < 8; ++k) park(c[k], v_next + 16u * k, true);
      v_next += 128;
    }
    // window: w0 holds `s` unconsumed bits (its low s bits), then w1; widx = ring index of w0.  A pair of
    // codewords is at most 32 bits, so one peek never reaches beyond w1; wn is the word behind w1, re-read from
    // the ring after every consume (the read has a whole pair's time to arrive before the next shift needs it)
    const uint32_t skip = v_bits & 15u;
    const uint32_t a0 = skip & 3u;
    const uint32_t widx0 = (skip >> 2) - (a0 == 0 ? 1u : 0u);  // a0 == 0: start with a fully consumed w0
    uint32_t s = (32u - 8u * a0) & 31u;
    uint32_t w0 = row[~widx0 & 31u], w1 = row[~(widx0 + 1u) & 31u];
    uint32_t wn = row[~(widx0 + 2u) & 31u];
    // qb = 4 * ~(widx + 2): the byte offset of wn's slot before masking; widx itself is only needed by service()
    uint32_t qb = 4u * ~(widx0 + 2u);
    // the ring is topped up every SECOND block with up to 6 chunks (96 bytes >= the 80 bytes two blocks can
    // consume: at most one word per pair).  Three of them are requested one service ahead -- most lanes need
    // one or two (0.53 bytes per sample), and a scattered 16-byte-per-lane load costs ~64 cycles of issue --
    // the other three only when some lane does need them.
    uint4 ld[X3S_AHEAD];
#pragma unroll
    for (int k = 0; k < (int)X3S_AHEAD; ++k) ld[k] = request(v_next + 16u * k);
    uint32_t v_req = v_next;
    // consume -nn (<= 32) bits, given as the NEGATIVE count (that is what the codeword walk below has at hand);
    // the word shift is v_bfi with a VGPR mask (see x3_decode_fast_kernel)
    auto consume_to = [&](int32_t s2) {  // s2 = s - bits consumed (>= -32)
      const uint32_t m = (uint32_t)(s2 >> 31);
      s = (uint32_t)s2 & 31u;
      w0 = x3_bfi(m, w1, w0);
      w1 = x3_bfi(m, wn, w1);
      uint32_t addr;
      asm("v_lshl_add_u32 %0, %2, 2, %0\n\t"                 // widx += 1 on a shift: qb -= 4
          "v_and_or_b32 %1, %0, %3, %4"
          : "+v"(qb), "=v"(addr) : "v"(m), "v"(124u), "v"(row_base));
      wn = x3_lds_read_b32(addr);
      // keep the read HERE: left to itself the scheduler sinks it to just in front of the next shift, where its
      // whole LDS latency is waited for
      __builtin_amdgcn_sched_barrier(0);
    };
    auto consume_neg = [&](uint32_t nn) { consume_to((int32_t)(s + nn)); };
    auto service = [&]() {
      const uint32_t widx = ~((uint32_t)((int32_t)qb >> 2)) - 2u;
      const uint32_t used = wr_abs - widx;  // dwords from w0 on that the ring still needs
      const uint32_t fit = used >= X3_DEC_RING_DW ? 0u : (X3_DEC_RING_DW - used) >> 2;  // (widx may be -1)
      const bool near_end = __any((int32_t)(v_end - v_req) < (int32_t)(16u * 3u * X3S_PERIOD));
#pragma unroll
      for (uint32_t k = 0; k < X3S_AHEAD; ++k) {
        if (fit > k) park(ld[k], v_req + 16u * k, near_end);
      }
      if (__any(fit > X3S_AHEAD)) {  // a lane went through more than that in two blocks (BFP / literal blocks)
        uint4 more[3u * X3S_PERIOD - X3S_AHEAD];
#pragma unroll
        for (uint32_t k = 0; k < 3u * X3S_PERIOD - X3S_AHEAD; ++k) more[k] = request(v_req + 16u * (X3S_AHEAD + k));
#pragma unroll
        for (uint32_t k = 0; k < 3u * X3S_PERIOD - X3S_AHEAD; ++k) {
          if (fit > X3S_AHEAD + k) park(more[k], v_req + 16u * (X3S_AHEAD + k), near_end);
        }
      }
      v_next += 16u * (fit > 3u * X3S_PERIOD ? 3u * X3S_PERIOD : fit);
      v_req = v_next;
#pragma unroll
      for (int k = 0; k < (int)X3S_AHEAD; ++k) ld[k] = request(v_req + 16u * k);
    };

    const uint32_t k_tab = (p.k[1] << 16) | (p.k[2] << 24);  // log2(level) by ftype
    for (uint32_t b = 0; b < nblk_max; ++b) {
      const uint32_t cnt = remaining < X3S_BL ? remaining : X3S_BL;
      remaining -= cnt;
      X3_STAMP(0);
      if ((b % X3S_PERIOD) == 0) service();
      X3_STAMP(1);
      // block header: 2 bits ftype; ftype 0 -> 4 more bits E-1 (decoder.rs:138-144, 209-216)
      const uint32_t hdr = __builtin_amdgcn_alignbit(w0, w1, s) >> 26;
      // all of it as arithmetic on the 6 bits (no compares: selects on a stale VCC are slow on gfx950)
      const uint32_t ftype = hdr >> 4;
      const uint32_t zmask = (uint32_t)((int32_t)(15u - hdr) >> 31);     // all ones for Rice (hdr >= 16)
      consume_neg((cnt ? 0xFFFFFFFFu : 0u) & ((zmask & 4u) - 6u));       // 6 header bits for BFP, 2 for Rice
      const uint32_t width = x3_bfi(zmask, (1u << ftype) >> 1, (hdr & 15u) + 1u);  // Rice 1,2,4; BFP E
      const uint32_t kk = (k_tab >> (8u * ftype)) & 0xFFu;               // 0, 0, k1, k2
      // what is handed over is the index into the reference's inverse table, i = (z << k) + r with r the k bits
      // BEHIND the terminating one (decoder.rs:186 computes the same as r' + level * (n - 1) with the one
      // included in r'), and the whole field in BFP/literal blocks: the last fw bits of the codeword, and a
      // zero run shifted by 31 there, out of the 16 bits that go across
      const uint32_t fw = x3_bfi(zmask, kk, width);
      const uint32_t lsh = x3_bfi(zmask, kk, 31u);
      const uint32_t nwidth = 0u - width;
      // block buffer: five rows of 64 x 8 bytes (two pair dwords per lane per row), then the 64 header words
      uint32_t* const buf = xfer + (b & 1u) * (X3S_XROWS * 64u);
      buf[X3S_PAIRS * 64u + lane] = hdr;
      X3_STAMP(2);
      if (__all(cnt == X3S_BL || cnt == 0u)) {
        // two samples per 32-bit peek and per window update (two valid codewords are <= 32 bits)
        uint2* const b2 = reinterpret_cast<uint2*>(buf) + lane;
#pragma unroll
        for (uint32_t j = 0; j < X3S_PAIRS; j += 2) {
          uint32_t X[2];
#pragma unroll
          for (uint32_t e = 0; e < 2; ++e) {
            // a codeword = z zeros + `width` bits (z only counts in Rice blocks): n = z + width bits in all, and
            // its field v = the top n bits of the peek (the zeros in front do not change the value).  With
            // nn = -n = z * zmask - width (one v_mad_i32_i24, zmask being -1 or 0), both "drop n bits"
            // (alignbit by 32 - n) and "the fw bits that end n bits in" (v_bfe_u32 at 32 - n) take nn as their
            // shift count (the hardware uses its low five bits): 4 instructions per codeword.  One asm block, so that the
            // compiler neither pads the dependent chain with s_nop nor reorders it.
            const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
            uint32_t z1, z2, v1, v2, t2, nn1, nn2;
            int32_t s2;
            asm("v_ffbh_u32 %0, %8\n\t"
                "v_mad_i32_i24 %4, %0, %9, %10\n\t"
                "v_alignbit_b32 %6, %8, 0, %4\n\t"
                "v_bfe_u32 %1, %8, %4, %12\n\t"
                "v_ffbh_u32 %2, %6\n\t"
                "v_mad_i32_i24 %5, %2, %9, %10\n\t"
                "v_bfe_u32 %3, %6, %5, %12\n\t"
                "v_add3_u32 %7, %11, %4, %5"
                : "=&v"(z1), "=&v"(v1), "=&v"(z2), "=&v"(v2), "=&v"(nn1), "=&v"(nn2), "=&v"(t2), "=&v"(s2)
                : "v"(t), "v"(zmask), "v"(nwidth), "v"(s), "v"(fw));
            consume_to(s2);
            X[e] = x3_pack_lo16((z1 << lsh) + v1, (z2 << lsh) + v2);
          }
          b2[(j >> 1) * 64u] = make_uint2(X[0], X[1]);
        }
      } else {
        // a block that is short in some lane (the last block of a frame): one sample at a time
        uint16_t* const h = reinterpret_cast<uint16_t*>(buf);
        for (uint32_t j = 0; j < X3S_BL; ++j) {
          const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
          const uint32_t z = x3_ffbh(t) & zmask;
          const uint32_t nn = nwidth - z;
          const uint32_t v = (t >> (nn & 31u)) & ((1u << fw) - 1u);
          consume_neg(j < cnt ? nn : 0u);
          h[x3s_half_index(j, lane)] = (uint16_t)((z << lsh) + v);
        }
      }
      X3_STAMP(3);
      X3S_BARRIER();
      X3_STAMP(4);
    }
    {
      // read position (bits from the ring's first chunk on) against the end of the payload
      const int32_t widx = (int32_t)~((uint32_t)((int32_t)qb >> 2)) - 2;
      s_over[lane] = (32 * widx + 32 - (int32_t)s > (int32_t)(8u * v_end)) ? 1u : 0u;
    }
    X3S_BARRIER();
  } else {
    // ================================================================= wave 1: valuer
    uint32_t* const orow = outs + lane * X3S_OUT_STRIDE;
    int16_t* __restrict__ const o = wav + wo;
    s_wo[lane] = wo;
    s_ns[lane] = samples;
    bool alive = active;
    uint32_t prevP = 0;  // the previous pair; its high half is the last sample so far (pending: even index)
    if (active) {
      const uint32_t first = ((uint32_t)x3[p0] << 8) | x3[p0 + 1];
      prevP = first << 16;
      if (samples == 1u) o[0] = (int16_t)first;
    }
    // the usual group: 64 frames of the same size, one behind the other in wav
    const uint32_t S0 = __builtin_amdgcn_readfirstlane(samples);
    const uint64_t wo0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(wo >> 32)) << 32) |
                         __builtin_amdgcn_readfirstlane((uint32_t)wo);
    bool regular = __all(active && samples == S0 && wo == wo0 + (uint64_t)lane * S0) && (S0 & 7u) == 0;
    uint32_t wbase = 0;  // first sample index of the staging window (a multiple of X3S_WIN)
    X3_WAVE_LDS_ORDER();

    // window [wbase, wbase + X3S_WIN): X3S_WPIECES pieces of 16 bytes per row, as many per lane.
    // Which pieces a lane moves never changes: piece t = 64*it + lane of row r = t / 10.  For the regular
    // group the LDS offset and the offset in wav (relative to the window) are computed once.
    uint32_t f_src[X3S_WPIECES], f_dst[X3S_WPIECES];
#pragma unroll
    for (uint32_t it = 0; it < X3S_WPIECES; ++it) {
      const uint32_t t = it * 64u + lane;
      const uint32_t r = t / X3S_WPIECES, q = t - r * X3S_WPIECES;
      f_src[it] = r * X3S_OUT_STRIDE + 4u * q;   // dwords into outs
      f_dst[it] = r * S0 + 8u * q;               // samples behind wav + wo0 + wbase (< 2^32: 64 frames)
    }
    auto flush = [&]() {
      X3_WAVE_LDS_ORDER();
      if (regular && wbase + X3S_WIN <= S0) {
        int16_t* const base = wav + wo0 + wbase;
#pragma unroll
        for (uint32_t it = 0; it < X3S_WPIECES; ++it)
          x3_store_stream16(base + f_dst[it], *reinterpret_cast<const x3_u32x4*>(outs + f_src[it]));
      } else {
#pragma unroll 2
        for (uint32_t it = 0; it < X3S_WPIECES; ++it) {
          const uint32_t t = it * 64u + lane;
          const uint32_t r = t / X3S_WPIECES, q = t - r * X3S_WPIECES;
          const uint32_t ns = s_ns[r];
          if (wbase + 8u * q + 8u <= ns)
            x3_store_stream16(wav + s_wo[r] + wbase + 8u * q,
                              *reinterpret_cast<const x3_u32x4*>(outs + r * X3S_OUT_STRIDE + 4u * q));
        }
      }
      X3_WAVE_LDS_ORDER();
    };

    const uint32_t bound_tab = (p.inv_len[0] << 8) | (p.inv_len[1] << 16) | (p.inv_len[2] << 24);  // by ftype (< 256)
    for (uint32_t b = 0; b < nblk_max; ++b) {
      X3_STAMP(0);
      X3S_BARRIER();
      X3_STAMP(4);
      const uint32_t cnt = remaining < X3S_BL ? remaining : X3S_BL;
      remaining -= cnt;
      const uint32_t* const buf = xfer + (b & 1u) * (X3S_XROWS * 64u);
      const uint32_t hdr = buf[X3S_PAIRS * 64u + lane];
      // block parameters from the header bits, as arithmetic (see the parser)
      const uint32_t ftype = hdr >> 4;
      const uint32_t E = (hdr & 15u) + 1u;
      const uint32_t zmask = (uint32_t)((int32_t)(15u - hdr) >> 31);      // all ones for Rice
      const bool bfp = zmask == 0u;
      const uint32_t litmask = ~zmask & (uint32_t)((int32_t)(14u - (hdr & 15u)) >> 31);  // BFP with E == 16
      const uint32_t neg_thresh = ~zmask & (1u << (E - 1u));
      const uint32_t neg2 = (neg_thresh << 1) & ~litmask;
      const uint32_t bound = x3_bfi(zmask, (bound_tab >> (8u * ftype)) & 0xFFu, 0xFFFFFFFFu);
      if (cnt && alive && bfp && E <= 5u) {  // decoder.rs:209-216
        st = X3D_FRAME_DECODE_INVALID_BPF;
        alive = false;
      }
      const uint32_t tm12 = ((neg_thresh - 1u) & 0xFFFFu) * 0x10001u;  // thresh - 1 in each half
      const uint32_t neg22 = (neg2 & 0xFFFFu) * 0x10001u;              // 2 * thresh = 2^E (0 for a literal block)
      uint32_t maxii2 = 0;
      uint32_t* const dst = orow + X3S_PAIRS * (b & (X3S_WBLK - 1u));
      X3_STAMP(2);

      if (__all(cnt == X3S_BL || cnt == 0u)) {
        // the whole block's indices at once (five 8-byte reads in flight), then ten pairs from registers
        const uint2* const b2 = reinterpret_cast<const uint2*>(buf) + lane;
        uint2 XX[5];
#pragma unroll
        for (uint32_t r = 0; r < 5u; ++r) XX[r] = b2[r * 64u];
#pragma unroll
        for (uint32_t r = 0; r < 5u; ++r) {
          uint32_t W[2];
#pragma unroll
          for (uint32_t e = 0; e < 2; ++e) {
            const uint32_t X = e ? XX[r].y : XX[r].x;
            // Rice: X = i, the index into the inverse table (decoder.rs:186), which is a zigzag (x3.rs:200-204)
            maxii2 = x3_pk_max_u16(maxii2, X);
            const uint32_t R = x3_pk_lshr_b16_1(X) ^ x3_pk_sub_u16(0u, X & 0x00010001u);
            // BFP: unsigned_to_i16 (decoder.rs:198-207): v - (v > thresh ? 2*thresh : 0), strict compare.
            // v < 2^E and thresh = 2^(E-1), so bit E of v + thresh - 1 says v > thresh.
            const uint32_t B = x3_pk_sub_u16(X, x3_pk_add_u16(X, tm12) & neg22);
            const uint32_t D = x3_bfi(zmask, R, B);                        // (d1, d2)
            // (last + d1, last + d1 + d2), last = the high half of the previous pair: both halves of D plus last,
            // then d1 once more onto the high half
            uint32_t P = x3_pk_mad_u16_alo(D, 0x00010000u, x3_pk_add_u16_bhi(D, prevP));
            P = x3_bfi(litmask, X, P);                                     // literal: field = sample
            W[e] = __builtin_amdgcn_alignbit(P, prevP, 16);                // (pending sample, la)
            prevP = P;
          }
          *reinterpret_cast<uint2*>(dst + 2u * r) = make_uint2(W[0], W[1]);
        }
      } else {
        // short block somewhere in the group: one sample at a time, staged as halfwords
        const uint16_t* const h = reinterpret_cast<const uint16_t*>(buf);
        uint16_t* const oh = reinterpret_cast<uint16_t*>(orow);
        const uint32_t idx0 = 1u + X3S_BL * b - wbase;  // window-relative index of the block's first sample
        if (cnt) oh[idx0 - 1u] = (uint16_t)(prevP >> 16);
        uint32_t last = prevP >> 16, maxii = 0;
        for (uint32_t j = 0; j < X3S_BL; ++j) {
          if (j < cnt) {
            const uint32_t x = h[x3s_half_index(j, lane)];
            const uint32_t ii = x;
            const uint32_t d_rice = (ii >> 1) ^ (0u - (ii & 1u));
            const uint32_t d_bfp = x - (x > neg_thresh ? neg2 : 0u);
            const uint32_t d = bfp ? d_bfp : d_rice;
            last = litmask ? x : ((last + d) & 0xFFFFu);
            maxii = bfp ? maxii : (ii > maxii ? ii : maxii);
            oh[idx0 + j] = (uint16_t)last;
          }
        }
        maxii2 = maxii;
        prevP = last << 16;
      }
      X3_STAMP(3);
      // OutOfBoundsInverse (decoder.rs:160,187): the frame stops here
      if (cnt && alive && max(maxii2 & 0xFFFFu, maxii2 >> 16) >= bound) {
        st = X3D_OUT_OF_BOUNDS_INVERSE;
        alive = false;
      }
      if (__any(active && !alive)) {  // a frame of the group failed: no more stores for it
        regular = false;
        if (!alive) s_ns[lane] = 0;
      }
      if (cnt && remaining == 0 && alive) {
        // this lane's frame is complete: the flush stores only 16-byte pieces that lie inside the frame
        if (samples & 1u) orow[(samples - 1u - wbase) >> 1] = prevP >> 16;
        const uint32_t done = samples & ~7u;
        const uint32_t from = done > wbase ? done : wbase;
        const uint16_t* h = reinterpret_cast<const uint16_t*>(orow);
        for (uint32_t sx = from; sx < samples; ++sx) o[sx] = (int16_t)h[sx - wbase];
      }
      X3_STAMP(1);
      if ((b & (X3S_WBLK - 1u)) == X3S_WBLK - 1u) {
        flush();
        wbase += X3S_WIN;
        X3_STAMP(5);
      }
    }
    if (nblk_max & (X3S_WBLK - 1u)) flush();
    X3S_BARRIER();
    if (active && (st == X3D_OUT_OF_BOUNDS_INVERSE || st == X3D_FRAME_DECODE_INVALID_BPF || s_over[lane]))
      st = X3D_REPLAY;  // the reference's reader decides (x3_decode_replay.h; x3_decode_merge_kernel)
    if (f < n_frames) status[f] = st;
  }
#ifdef X3_DBG_STAMPS
  dbg_acc[6] = dbg_start;          // constant-rate clock: which groups were resident together
  dbg_acc[7] = wall_clock64();
  {
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const unsigned long long where = ((unsigned long long)xcc_id << 32) | hw_id;
    if (parser) dbg_acc[5] = where; else dbg_acc[6] = where;
  }
  if (lane == 0 && blockIdx.x < 2048)
    for (int k = 0; k < 8; ++k) x3_dbg[(blockIdx.x * 2 + (parser ? 0 : 1)) * 8 + k] = dbg_acc[k];
#endif
}
