// x3_encode_wave_kernel.h -- single-pass frame encoder for block_len = 20 (and, round 6, 10 and 40: template parameter BL),
// third generation: ONE WAVE PER FRAME.
//
// Replaces encoder::encode / encode_frame / x3_encode_block / encode_rice_block / encode_bfp_block / encode_literal
// (src/encoder.rs:51-315), BitPacker (src/bitpacker.rs:46-177) and the running crc16 (src/crc.rs:44-58).
//
// The second generation (x3_encode_stream2_kernel.h: eight waves per frame, one block of 20 samples per lane) is bound by
// VALU issue, and about half of its 565 vector instructions per wave and frame do not depend on the number of
// samples: bit-length scans, the size words, CRC multipliers and reductions, header, copy-out bookkeeping, three
// barriers (profiles/r2/pmc_sq.txt).  Here that part is paid once per FRAME instead of once per 1 280 samples:
//
//  * a frame (<= 10 240 samples) belongs to ONE wave: two half-frames of 5 120 samples, a lane takes FOUR consecutive
//    blocks (80 samples, 160 contiguous bytes) of each.  Consecutive blocks share one bit accumulator: one start
//    position, one final flush per lane and half.  No barrier anywhere in the frame loop -- everything between lanes is
//    DPP or the wave's own LDS image (a wave's DS instructions execute in order);
//  * sixteen such waves per CU (one workgroup of 1 024 threads, <= 128 VGPRs) share 5 KB of CRC tables; each has a
//    frame image of 38 rows x 256 B in LDS.  The samples of both halves live in registers (2 x 41 dwords) and are
//    turned IN PLACE into the emission source (zigzag / difference / raw per pair) by the analysis; the next frame's
//    halves are requested as soon as the emission has consumed a half and fly under the rest of the frame;
//  * the frame loop is skewed: a frame's size is published right behind the analysis of its two halves, at the top of
//    an iteration, and its offset is asked for one iteration LATER, behind the analysis of the next frame (the image
//    waits in LDS meanwhile).  A frame's offset needs the sizes of the up to 4 095 frames in front of it that are in
//    flight; with the size out at 20 % of a frame's work and the offset due at 120 %, a wave waits only for waves that
//    are a whole frame behind it (measured without the skew: a third of every wave's time, tools/scratch/dbg_stamps_wave.py);
//  * payload CRC-16: lane t folds dword t of every 64-dword row of the image Horner-style -- a 32-bit state that is
//    congruent to the sum so far, times x^4096 per step of one of two chains: four look-ups per dword, the incoming
//    dword is not reduced; the 0xFFFF init is folded into the first 16 payload bits -- multiplies by its fixed
//    x^(32*(63-t)), the lanes are XOR-reduced by DPP, and the zero bytes between the payload's end and the end of its
//    last row are undone by one multiplication with x^(-8 z): no per-size tables;
//  * stream offsets: the sixteen frames of a workgroup's generation are CONSECUTIVE frames; their sizes meet in LDS
//    (slot + arrival counter, the last arriver publishes the generation's total as one {epoch:12 | bytes:20} word),
//    and a generation's base is its predecessor's base plus the <= 256 totals in between (the second-generation kernel's
//    scheme, one level up).  A wave asks for those words when its emission is done and prefetches the next frame's
//    samples behind the request.
//
// A frame whose payload does not fit the image (> 9 728 bytes: dense content) is analysed like any other -- its size goes
// out, its statistics are counted, its offset is assigned and written to the frame index -- but it is not emitted here:
// it appends itself to a device list, and the launch behind this one (x3_encode_stream2_kernel<true>, one workgroup per
// listed frame, worst-case images) writes it at that offset.  Round 3 flagged the whole launch instead and the host
// encoded the call again: one loud frame in an hour of recording cost a second encode.
#pragma once
#include "x3_encode_stream2_kernel.h"

#ifndef X3W_CLAIM
#define X3W_CLAIM 1
#endif
#ifndef X3W_PRIO
#define X3W_PRIO 1  // 1: priorities from the arrival rank in the workgroup's generation; >= 2: this priority until the size is out; 0: none
#endif
#ifndef X3W_BARRIER
#define X3W_BARRIER 0
#endif
// Polls are far apart: a waiting wave that looks again every 128 clocks takes issue slots and LDS / L2 bandwidth from the
// waves it is waiting for.  s_sleep 24 (1 500 clocks) between looks at the workgroup's LDS words and 127 (8 000 clocks)
// between trips to the size words: 0.4225 ms on every box seen, against 0.427-0.439 with 2 and 8
// (sweeps of five pairs of values x four to six processes on three boxes, round 3).
#ifndef X3W_SLEEP_LDS
#define X3W_SLEEP_LDS 24
#endif
#ifndef X3W_SLEEP_DESC
#define X3W_SLEEP_DESC 127
#endif
// bounded waits of about 25 ms with those sleeps
#define X3W_SPINS_LDS (X3_SPIN_LIMIT << 2)
#define X3W_SPINS_DESC (X3_SPIN_LIMIT >> 1)
#ifndef X3W_SKIP_EMPTY_HALF
#define X3W_SKIP_EMPTY_HALF 1
#endif
#ifndef X3W_NOWAIT
#define X3W_NOWAIT 0
#endif
#if X3W_NOWAIT && !defined(X3_EXPERIMENT)
#error "X3W_NOWAIT builds give wrong results (frames land at wrong offsets): experiment builds only (-DX3_EXPERIMENT)"
#endif
#ifndef X3W_COPY_UNROLL
#define X3W_COPY_UNROLL 2
#endif
#define X3W_WAVES 16u
#define X3W_THREADS (64u * X3W_WAVES)
// (X3W_TAB_BYTES: x3_tables.h)
#define X3W_BOOK_BYTES 1024u
#define X3W_IMG_ROWS 38u
#define X3W_IMG_BYTES (X3W_IMG_ROWS * 256u)
#define X3W_SMEM (X3W_TAB_BYTES + X3W_BOOK_BYTES + X3W_WAVES * X3W_IMG_BYTES)  // 162 048 of 163 840
#define X3W_PART 5120u         // samples per part: 64 lanes x 4 blocks x 20
#define X3W_MAX_NWG 256u      // a generation's base sums at most this many totals: four words per lane
#define X3W_DESC_PAD 320u     // words in front of desc[0]: the windows of the first generation reach below 0
static_assert(X3W_IMG_BYTES == X3_DENSE_PAYLOAD_BYTES, "the dense pass takes what the image does not hold");

struct X3WaveArgs {
  const int16_t* wav;
  uint8_t* out;
  uint64_t* frame_off;
  uint32_t* desc;           // one word per workgroup generation, X3W_DESC_PAD words in front
  unsigned char* ctl;       // int status[8] | u64 stats[6] | u64 end_pos
  const uint32_t* tabs;     // X3W_TAB_BYTES
  uint32_t* log;            // the context's launch log (X3_LOG_ENC_BASE words in): workgroup 0's clocks, nothing reads them
  uint32_t log_epoch;
  uint32_t* dense_list;     // frames that did not fit the image, in the order their offsets became known (count: ctl + X3_CTL_DENSE_COUNT)
  uint64_t out_cap, start_pos, n_per_clip, clip_stride, n_frames;
  uint32_t fpc, spf, epoch;
  uint32_t m;               // frames per workgroup generation = active waves per workgroup (1..16)
  uint32_t nwg;             // workgroups (<= X3W_MAX_NWG)
  uint32_t n_wggen;         // ceil(n_frames / m)
  uint32_t step_clip, step_idx;  // (nwg * m) frames as clips + frames
  const uint64_t* src_off;       // x3_encode_frames_dev: frame f = src_n[f] samples at wav + src_off[f] (else nullptr)
  const uint32_t* src_n;
  uint32_t thr0, thr1, thr2, kpack;
  uint32_t drop_wgi;        // tests: the generation whose total is never published (a workgroup that is not resident); ~0: none
  // the segment index (x3_decode_split_kernel.h, "STRETCHES"; include/x3hip.h): where every 2^seg_log2 -th block of a frame
  // begins in its payload and the sample in front of it -- the prefix scan below has the one, the input is the other
  uint2* seg;               // nullptr: none asked for
  uint32_t seg_log2;        // blocks per entry = 1 << seg_log2 (>= 4)
  uint32_t seg_pitch;       // entries per frame
};

__device__ __forceinline__ uint32_t x3_pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the wave's LDS traffic between lanes needs no barrier (DS instructions of a wave execute in order); this only
// keeps the compiler from moving LDS accesses across the point
__device__ __forceinline__ void x3w_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); asm volatile("" ::: "memory"); }

// ---- analysis of block Q of a lane's four (x3_encode_block, encoder.rs:289-315).  X[10Q .. 10Q+10] hold samples
// 20Q .. 20Q+21 of the lane's run as (even, odd) pairs; the block is samples 20Q+1 .. 20Q+20, predicted from 20Q.
// cnt is 20, 19 (the last block of a full frame) or 0 (behind the frame): frames with any other block take the generic
// path below.  X[10Q .. 10Q+9] are REPLACED, pair by pair, by the emission source (zigzag / exact difference / raw samples;
// X[10Q+10], the next block's first word, stays) and meta (15 bits) says how to emit it:
// hdr value [0..5] | bits per field [6..10] | Rice [11] | statistics index [12..14]; 0: no block.
// No temporaries survive a pair (the registers are full: two halves of a frame); the rare literal block, whose raw
// samples the saturated differences no longer hold, reads its 44 bytes again (rs, vo: the lane's run in the frame).
template <int Q>
__device__ __forceinline__ void x3w_analyse(uint32_t (&X)[41], uint32_t cnt, uint32_t thr0, uint32_t thr1, uint32_t thr2,
                                            uint32_t kpack, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so,
                                            uint32_t& nbits, uint32_t& meta) {
  constexpr int B = 10 * Q;
  // one pass: the (saturated) differences in place, their maximum and minimum per half.  (Until round 3 this pass left
  // zigzag values and the BFP blocks turned them back: four instructions per pair that every wave paid, because some
  // lane of 64 nearly always has a BFP block -- 12.7 % of config 3's blocks.  Now the Rice blocks do the zigzag, in
  // the pass that sums their code lengths, and BFP blocks find their differences ready.)
  uint32_t mx = 0x80008000u, mn = 0x7FFF7FFFu;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const uint32_t Xj = __builtin_amdgcn_alignbit(X[B + j + 1], X[B + j], 16);  // (s[2j+1], s[2j+2])
    uint32_t d = x3_pk_sub_sat(Xj, X[B + j]);                                    // (d[2j+1], d[2j+2]), saturated
    if (j == 9) d &= cnt == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;                     // a 19-sample block has no sample 20
    X[B + j] = d;
    mx = x3_pk_max_i16(mx, d);
    mn = x3_pk_min_i16(mn, d);
  }
  // max |d|: max(mx, -mn) per half, then over the halves.  -(-32768) saturates to 32767 and a saturated difference stands
  // for anything beyond: both are literal blocks (nb >= 15) whatever the exact value (thresholds are < 32: stream_safe_thresholds)
  const uint32_t ab = x3_pk_max_i16(mx, x3_pk_sub_sat(0u, mn));
  const int32_t maxabs = (int32_t)max(ab & 0xFFFFu, ab >> 16);
  uint32_t nb_ = 0, mt = 0;
  if (maxabs <= (int32_t)thr2) {
    const uint32_t ft = (maxabs > (int32_t)thr0 ? 1u : 0u) + (maxabs > (int32_t)thr1 ? 1u : 0u);
    const uint32_t k = (kpack >> (8u * ft)) & 0xFFu;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const uint32_t d = X[B + j];
      const uint32_t z = x3_pk_shl_b16_1(d) ^ x3_pk_ashr_i16_15(d);              // zigzag, per half (0 stays 0)
      X[B + j] = z;
      sum = x3_pk_add_u16(sum, x3_pk_shr_u16(z, k));
    }
    nb_ = 2u + cnt * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
    mt = (ft + 1u) | ((k + 1u) << 6) | (1u << 11) | (k << 12);
  } else {
    const uint32_t nb = 32u - (uint32_t)__clz(maxabs);
    if (nb >= 15u) {
      nb_ = 6u + 16u * cnt;
      // raw samples: the block's dwords again (range-checked as the first time: zeros behind the frame)
      uint32_t prev = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * B), (int)so, 0);
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const uint32_t nxt = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * (B + j + 1)), (int)so, 0);
        X[B + j] = __builtin_amdgcn_alignbit(nxt, prev, 16);
        prev = nxt;
      }
      mt = 15u | (16u << 6) | (5u << 12);
    } else {
      // (X holds the exact differences: |d| < 16 384, nothing was saturated)
      nb_ = 6u + cnt * (nb + 1u);
      mt = nb | ((nb + 1u) << 6) | (4u << 12);
    }
  }
  nbits = cnt ? nb_ : 0u;
  meta = cnt ? mt : 0u;
}

// ---- block length 40 (round 6): a block is TWO of the lane's four runs of ten dwords -- one filter over both (their largest
// |difference| together), one header, the second run a continuation that x3w_emit<Q, true> writes without a header.  cntA, cntB:
// samples of the two runs, (20, 20), (20, 19), (20, 0), (19, 0) or (0, 0): everything else takes the generic path.  The
// arithmetic is x3w_analyse's; what is new are the masks of the second run (a block that ends in its first run must not
// see what the loads returned behind it).
template <int QQ>
__device__ __forceinline__ void x3w_analyse40(uint32_t (&X)[41], uint32_t cntA, uint32_t cntB, uint32_t thr0, uint32_t thr1,
                                              uint32_t thr2, uint32_t kpack, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so,
                                              uint32_t& nbits, uint32_t& metaA, uint32_t& metaB) {
  constexpr int B = 20 * QQ;
  uint32_t mx = 0x80008000u, mn = 0x7FFF7FFFu;
  const uint32_t onB = cntB ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int j = 0; j < 20; ++j) {
    const uint32_t Xj = __builtin_amdgcn_alignbit(X[B + j + 1], X[B + j], 16);
    uint32_t d = x3_pk_sub_sat(Xj, X[B + j]);
    if (j == 9) d &= cntA == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;
    if (j >= 10 && j < 19) d &= onB;
    if (j == 19) d &= cntB == 20u ? 0xFFFFFFFFu : (cntB ? 0x0000FFFFu : 0u);
    X[B + j] = d;
    mx = x3_pk_max_i16(mx, d);
    mn = x3_pk_min_i16(mn, d);
  }
  const uint32_t ab = x3_pk_max_i16(mx, x3_pk_sub_sat(0u, mn));
  const int32_t maxabs = (int32_t)max(ab & 0xFFFFu, ab >> 16);
  const uint32_t cnt = cntA + cntB;
  uint32_t nb_ = 0, mt = 0;
  if (maxabs <= (int32_t)thr2) {
    const uint32_t ft = (maxabs > (int32_t)thr0 ? 1u : 0u) + (maxabs > (int32_t)thr1 ? 1u : 0u);
    const uint32_t k = (kpack >> (8u * ft)) & 0xFFu;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const uint32_t d = X[B + j];
      const uint32_t z = x3_pk_shl_b16_1(d) ^ x3_pk_ashr_i16_15(d);
      X[B + j] = z;
      sum = x3_pk_add_u16(sum, x3_pk_shr_u16(z, k));
    }
    nb_ = 2u + cnt * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
    mt = (ft + 1u) | ((k + 1u) << 6) | (1u << 11) | (k << 12);
  } else {
    const uint32_t nb = 32u - (uint32_t)__clz(maxabs);
    if (nb >= 15u) {
      nb_ = 6u + 16u * cnt;
      uint32_t prev = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * B), (int)so, 0);
#pragma unroll
      for (int j = 0; j < 20; ++j) {
        const uint32_t nxt = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * (B + j + 1)), (int)so, 0);
        X[B + j] = __builtin_amdgcn_alignbit(nxt, prev, 16);
        prev = nxt;
      }
      mt = 15u | (16u << 6) | (5u << 12);
    } else {
      nb_ = 6u + cnt * (nb + 1u);
      mt = nb | ((nb + 1u) << 6) | (4u << 12);
    }
  }
  nbits = cntA ? nb_ : 0u;
  metaA = cntA ? mt : 0u;
  metaB = cntB ? mt : 0u;
}

// ---- block length 10 (round 6): a lane's run of ten dwords is TWO blocks, each with its own filter and header.  The lane
// has eight blocks per half and four 16-bit meta slots: a block's meta shrinks to 8 bits -- header value [0..3] | Rice [4] |
// present [5] -- and x3w_meta_of() makes the 15-bit form of x3w_analyse out of it again where a block is emitted or counted
// (field width, Rice parameter and statistics index all follow from the header value and the parameter set).
__device__ __forceinline__ uint32_t x3w_meta8(uint32_t meta) {   // 15-bit form -> 8-bit form
  return meta ? (meta & 15u) | (((meta >> 11) & 1u) << 4) | 32u : 0u;
}
__device__ __forceinline__ uint32_t x3w_meta_of(uint32_t m8, uint32_t kpack) {
  const uint32_t hv = m8 & 15u, rice = (m8 >> 4) & 1u;
  const uint32_t k = (kpack >> (8u * ((hv - 1u) & 3u))) & 0xFFu;
  const uint32_t lbase = rice ? k + 1u : (hv == 15u ? 16u : hv + 1u);
  const uint32_t sidx = rice ? k : (hv == 15u ? 5u : 4u);
  return (m8 & 32u) ? hv | (lbase << 6) | (rice << 11) | (sidx << 12) : 0u;
}
// cnt: samples of the RUN, 20, 19 or 0 (as x3w_analyse); meta: the two blocks' 8-bit forms, first block low
template <int Q>
__device__ __forceinline__ void x3w_analyse10(uint32_t (&X)[41], uint32_t cnt, uint32_t thr0, uint32_t thr1, uint32_t thr2,
                                              uint32_t kpack, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so,
                                              uint32_t& nbits, uint32_t& meta) {
  constexpr int B = 10 * Q;
  uint32_t mx[2] = {0x80008000u, 0x80008000u}, mn[2] = {0x7FFF7FFFu, 0x7FFF7FFFu};
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const uint32_t Xj = __builtin_amdgcn_alignbit(X[B + j + 1], X[B + j], 16);
    uint32_t d = x3_pk_sub_sat(Xj, X[B + j]);
    if (j == 9) d &= cnt == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;
    X[B + j] = d;
    mx[j / 5] = x3_pk_max_i16(mx[j / 5], d);
    mn[j / 5] = x3_pk_min_i16(mn[j / 5], d);
  }
  uint32_t nbt = 0, mt2 = 0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t cu = u ? cnt - 10u : 10u;   // (used when cnt != 0: 20 or 19)
    const uint32_t ab = x3_pk_max_i16(mx[u], x3_pk_sub_sat(0u, mn[u]));
    const int32_t maxabs = (int32_t)max(ab & 0xFFFFu, ab >> 16);
    uint32_t nb_ = 0, m8 = 0;
    if (maxabs <= (int32_t)thr2) {
      const uint32_t ft = (maxabs > (int32_t)thr0 ? 1u : 0u) + (maxabs > (int32_t)thr1 ? 1u : 0u);
      const uint32_t k = (kpack >> (8u * ft)) & 0xFFu;
      uint32_t sum = 0;
#pragma unroll
      for (int j = 5 * u; j < 5 * u + 5; ++j) {
        const uint32_t d = X[B + j];
        const uint32_t z = x3_pk_shl_b16_1(d) ^ x3_pk_ashr_i16_15(d);
        X[B + j] = z;
        sum = x3_pk_add_u16(sum, x3_pk_shr_u16(z, k));
      }
      nb_ = 2u + cu * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
      m8 = (ft + 1u) | 16u | 32u;
    } else {
      const uint32_t nb = 32u - (uint32_t)__clz(maxabs);
      if (nb >= 15u) {
        nb_ = 6u + 16u * cu;
        uint32_t prev = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * (B + 5 * u)), (int)so, 0);
#pragma unroll
        for (int j = 5 * u; j < 5 * u + 5; ++j) {
          const uint32_t nxt = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 4u * (B + j + 1)), (int)so, 0);
          X[B + j] = __builtin_amdgcn_alignbit(nxt, prev, 16);
          prev = nxt;
        }
        m8 = 15u | 32u;
      } else {
        nb_ = 6u + cu * (nb + 1u);
        m8 = nb | 32u;
      }
    }
    nbt += nb_;
    mt2 |= m8 << (8 * u);
  }
  nbits = cnt ? nbt : 0u;
  meta = cnt ? mt2 : 0u;
}

// ---- the generic path: frames whose last block has 1..18 samples (the ragged tail frame of a clip).  Sample by
// sample from memory as the reference does it (x3_encode_block, encoder.rs:289-315), in 32-bit arithmetic; it shares
// nothing with the register arrays of the fast path.  bi: block index in the frame, cnt: its samples (1..20).
__device__ __forceinline__ int32_t x3w_sample(__amdgpu_buffer_rsrc_t rs, uint32_t i) {
  return (int32_t)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)(2u * i), 0, 0);
}
template <uint32_t BL = 20u>
__device__ __forceinline__ void x3w_slow_analyse(__amdgpu_buffer_rsrc_t rs, uint32_t bi, uint32_t cnt, uint32_t thr0,
                                                 uint32_t thr1, uint32_t thr2, uint32_t kpack, uint32_t& nbits, uint32_t& meta) {
  nbits = 0;
  meta = 0;
  if (cnt == 0) return;
  int32_t prev = x3w_sample(rs, BL * bi), maxabs = 0;
  for (uint32_t i = 1; i <= cnt; ++i) {
    const int32_t sv = x3w_sample(rs, BL * bi + i), d = sv - prev;
    prev = sv;
    maxabs = max(maxabs, d < 0 ? -d : d);
  }
  if (maxabs <= (int32_t)thr2) {
    const uint32_t ft = (maxabs > (int32_t)thr0 ? 1u : 0u) + (maxabs > (int32_t)thr1 ? 1u : 0u);
    const uint32_t k = (kpack >> (8u * ft)) & 0xFFu;
    uint32_t sum = 0;
    prev = x3w_sample(rs, BL * bi);
    for (uint32_t i = 1; i <= cnt; ++i) {
      const int32_t sv = x3w_sample(rs, BL * bi + i), d = sv - prev;
      prev = sv;
      const uint32_t u = d >= 0 ? 2u * (uint32_t)d : 2u * (uint32_t)(-d) - 1u;
      sum += u >> k;
    }
    nbits = 2u + cnt * (k + 1u) + sum;
    meta = (ft + 1u) | ((k + 1u) << 6) | (1u << 11) | (k << 12);
  } else {
    const uint32_t nb = 32u - (uint32_t)__clz(maxabs);
    if (nb >= 15u) {
      nbits = 6u + 16u * cnt;
      meta = 15u | (16u << 6) | (5u << 12);
    } else {
      nbits = 6u + cnt * (nb + 1u);
      meta = nb | ((nb + 1u) << 6) | (4u << 12);
    }
  }
}
struct X3WEmit;

struct X3WEmit {
  uint64_t acc;    // low (q + 32) bits are waiting for their word
  uint32_t q;      // bits waiting MINUS 32 (mod 2^32): adding a length carries out exactly when a word is complete, and
                   // the sum is then the shift that brings the word down -- one v_add_co instead of add + compare
  uint32_t waddr;  // LDS byte address of the word the accumulator flushes to
  __device__ __forceinline__ void start(uint32_t bit_pos, uint32_t img_addr) {
    acc = 0;
    q = (bit_pos & 31u) - 32u;
    waddr = img_addr + 4u * (bit_pos >> 5);
  }
  __device__ __forceinline__ void put(uint32_t code, uint32_t len) {  // len <= 32, code < 2^len
    acc = (acc << len) | (unsigned long long)code;
    uint32_t q2;
    if (__builtin_add_overflow(q, len, &q2)) {
      x3_lds_or_b32(waddr, (uint32_t)(acc >> q2));
      waddr += 4u;
      q2 -= 32u;
    }
    q = q2;
  }
  __device__ __forceinline__ void finish() {
    if (q != 0u - 32u) x3_lds_or_b32(waddr, (uint32_t)acc << ((0u - q) & 31u));
  }
};

// ---- emission of block Q (encode_rice_block / encode_bfp_block / encode_literal, encoder.rs:233-285): the image
// holds the stream's bytes as big-endian dword VALUES (bit 31 of a word = the first bit of the stream in it).
// CONT (block length 40): this run continues the block of the run in front of it -- no header
// J0, J1 (block length 10): the pairs of the run that belong to this block
template <int Q, bool CONT = false, int J0 = 0, int J1 = 10>
__device__ __forceinline__ void x3w_emit(const uint32_t (&W)[41], uint32_t meta, uint32_t cnt, X3WEmit& e) {
  constexpr int B = 10 * Q;
  if (meta) {
    const uint32_t lbase = (meta >> 6) & 31u, rice = (meta >> 11) & 1u;
    const uint32_t kq = lbase - rice;
    if (!CONT) e.put(meta & 63u, rice ? 2u : 6u);
    // (code, len) of a sample v: ((v & amask) | orc, (v >> kq) * rice + lbase) -- both samples of a pair at once in
    // packed 16-bit arithmetic, the halves combined with SDWA operand selects (x3_encode_common.h)
    const uint32_t qsh2 = kq * 0x10001u, lbase2 = lbase * 0x10001u, qmul2 = rice * 0x10001u;
    const uint32_t amask2 = ((1u << kq) - 1u) * 0x10001u, orc2 = (rice << kq) * 0x10001u;
    const uint32_t last_on = cnt == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;
    uint64_t acc = e.acc;
    uint32_t q = e.q, waddr = e.waddr;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      uint32_t Lp = x3_pk_mad_u16(x3_pk_lshr_b16(W[B + j], qsh2), qmul2, lbase2);  // (la, lc)
      uint32_t Cp = x3_and_or(W[B + j], amask2, orc2);                              // (ca, cc)
      if (j == 9) { Lp &= last_on; Cp &= last_on; }
      const uint32_t tot = x3_sdwa_add_w0_w1(Lp);                                   // la + lc <= 32
      const uint32_t pair = x3_sdwa_or_w1(x3_sdwa_shl_w0_by_w1(Cp, Lp), Cp);        // (ca << lc) | cc
      acc = (acc << tot) | (unsigned long long)pair;
      uint32_t q2;
      if (__builtin_add_overflow(q, tot, &q2)) {
        x3_lds_or_b32(waddr, (uint32_t)(acc >> q2));
        waddr += 4u;
        q2 -= 32u;
      }
      q = q2;
    }
    e.acc = acc;
    e.q = q;
    e.waddr = waddr;
  }
}

template <uint32_t BL = 20u>
__device__ __forceinline__ void x3w_slow_emit(__amdgpu_buffer_rsrc_t rs, uint32_t bi, uint32_t cnt, uint32_t meta, X3WEmit& e) {
  if (meta == 0) return;
  const uint32_t lbase = (meta >> 6) & 31u, rice = (meta >> 11) & 1u;
  e.put(meta & 63u, rice ? 2u : 6u);
  int32_t prev = x3w_sample(rs, BL * bi);
  for (uint32_t i = 1; i <= cnt; ++i) {
    const int32_t sv = x3w_sample(rs, BL * bi + i), d = sv - prev;
    prev = sv;
    if (rice) {
      const uint32_t k = lbase - 1u;
      const uint32_t u = d >= 0 ? 2u * (uint32_t)d : 2u * (uint32_t)(-d) - 1u;
      e.put((1u << k) | (u & ((1u << k) - 1u)), (u >> k) + k + 1u);   // (u >> k) zeros, then 1 and the k low bits
    } else if (lbase == 16u) {
      e.put((uint32_t)sv & 0xFFFFu, 16u);                             // literal: the raw sample
    } else {
      e.put((uint32_t)d & ((1u << lbase) - 1u), lbase);               // BFP: the difference in nb + 1 bits
    }
  }
}

__device__ __forceinline__ uint32_t x3w_cnt_of(int32_t rem, int q) {
  const int32_t c = rem - 20 * q;
  return c <= 0 ? 0u : (c < 20 ? (uint32_t)c : 20u);
}

// TAB: the frames come from a table (x3_encode_frames_dev) instead of the uniform clip layout.  A template parameter, not a
// test of a.src_off: with the test in it the kernel of the uniform layout ran 48 % slower (0.64 ms against 0.43) -- it sits
// at 128 VGPRs, and hipcc's schedule does not survive the extra live values.
// BL (round 6): 20, or 40 -- a block is then two of a lane's four runs per half (x3w_analyse40); the BL = 20 instantiation
// is the code of rounds 3-5 (`if constexpr`).
template <bool TAB, uint32_t BL = 20u>
__global__ void __launch_bounds__(X3W_THREADS) x3_encode_wave_kernel(X3WaveArgs a) {
  static_assert(BL == 10u || BL == 20u || BL == 40u, "block lengths 10, 20 and 40");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* const tab = reinterpret_cast<uint16_t*>(smem);
  uint32_t* const book = reinterpret_cast<uint32_t*>(smem + X3W_TAB_BYTES);
  // book: [0..127] size slots {tag:12 | bytes:20} of the waves, eight generations (a wave is at most a few generations
  //       ahead of another one of its workgroup: it passes generation g+1 only behind the total of g, which every wave
  //       has contributed to); [128..135] arrival counters; [136] waves that have left; [192..255] statistics (eight copies of six counters);
  //       [160..167] / [168..175] / [176..183] tag, low and high word of the generations' bases; [184..191] who fetches them
  const uint32_t tid = threadIdx.x;
  uint32_t lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  uint32_t* const img = reinterpret_cast<uint32_t*>(smem + X3W_TAB_BYTES + X3W_BOOK_BYTES + w * X3W_IMG_BYTES);
  const uint32_t img_addr = x3_lds_addr(img);
  const uint32_t tab_base = x3_lds_addr(tab);
  int* const status = reinterpret_cast<int*>(a.ctl);
  unsigned long long* const stats = reinterpret_cast<unsigned long long*>(a.ctl + 32);
  unsigned long long* const end_pos = stats + 6;

  const unsigned long long log_c0 = clock64(), log_w0 = wall_clock64();
  // ---- prologue: tables, bookkeeping, clear images
  for (uint32_t i = tid; i < X3W_TAB_BYTES / 4u; i += X3W_THREADS) reinterpret_cast<uint32_t*>(smem)[i] = a.tabs[i];
  if (tid < X3W_BOOK_BYTES / 4u) book[tid] = 0;
  for (uint32_t i = lane; i < X3W_IMG_BYTES / 16u; i += 64u) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (a.seg && blockIdx.x == 0 && tid == 0) a.seg[0] = make_uint2(0x58335347u, 1u << a.seg_log2);   // the index's header word
  if (w >= a.m) return;  // (whole waves; nothing below is a workgroup barrier)

  const uint32_t b = blockIdx.x;
  const uint32_t ready_tag = a.epoch << X3_DESC_BYTES_BITS;

  // frame of generation 0: f = b * m + w as (clip, idx); advanced by nwg * m frames per generation
  uint64_t clip;
  uint32_t idx;
  {
    const uint64_t f0 = (uint64_t)b * a.m + w;
    clip = f0 / a.fpc;
    idx = (uint32_t)(f0 - clip * a.fpc);
    if (TAB) { clip = f0; idx = 0; }   // (a frame table: `clip` carries the frame number)
  }
  auto geom_at = [&](uint64_t clip_, uint32_t idx_, const int16_t*& src, uint32_t& n) __attribute__((always_inline)) {
    if (TAB) {
      n = a.src_n[clip_];
      src = a.wav + a.src_off[clip_];
      return;
    }
    const uint64_t left = a.n_per_clip - (uint64_t)idx_ * (uint64_t)a.spf;
    n = left < a.spf ? (uint32_t)left : a.spf;
    src = a.wav + clip_ * a.clip_stride + (uint64_t)idx_ * (uint64_t)a.spf;
  };

  // a lane's run of half h of a frame of n samples at src: samples 5120 h + 80 l .. + 81 as 41 (even, odd) pairs.
  // Range-checked buffer loads (descriptor = the frame): dwords behind the frame read as zero.
  uint32_t X0[41], X1[41];
  auto load_half = [&](uint32_t (&X)[41], const int16_t* src, uint32_t n, uint32_t h) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
    uint32_t vo = 160u * lane;
    // (opaque: hipcc otherwise hoists the sums vo + 16 i out of the frame loop into registers of their own; inside
    // the loop they fold into the instructions' immediate offsets)
    asm volatile("" : "+v"(vo));
    const uint32_t so = 2u * X3W_PART * h;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const x3_v4u32 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16u * i), (int)so, 0);
      X[4 * i] = q.x; X[4 * i + 1] = q.y; X[4 * i + 2] = q.z; X[4 * i + 3] = q.w;
    }
    X[40] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 160u), (int)so, 0);
  };

  uint64_t gen_base = 0;   // stream offset of the generation of the frame in waiting (its workgroup's base)
  uint32_t gen = 0;
  uint64_t wgi = b, f = 0;
  const int16_t* src = nullptr;
  uint32_t n = 0;
  bool lost = false;       // a size wait timed out: this wave no longer knows where its frames go
  uint32_t rank = 0;       // how many waves of the workgroup had published their size of the last generation before this one

  // the frame in waiting: emitted and summed, its image in LDS, its offset not asked for yet
  bool have_prev = false, prev_ovf = false;
  uint64_t prev_f = 0, prev_wgi = 0;
  uint32_t prev_n = 0, prev_L = 0, prev_crc = 0, prev_gen = 0;

#ifdef X3_DBG_STAMPS
  // phase times: 0 wait for samples, 1 analysis, 2 scan + size, 3 emission, 4 CRC, 5 offset waits, 6 copy-out,
  // 7 clearing + bookkeeping
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  unsigned long long dbg_cnt[4] = {0, 0, 0, 0};
  unsigned long long dbg_far[4] = {0, 0, 0, 0};  // totals missing at a trip's first look, by distance  // descriptor polls, trips to the descriptors, ticks there, ticks waiting for another wave's base
#define X3W_DBG(x) x
  const int dbg_tl = b == 0 ? 0 : b == 64 ? 1 : b == 128 ? 2 : b == 192 ? 3 : b == 255 ? 4 : -1;  // timeline of five workgroups
#define X3W_TL(g, k) do { if (dbg_tl >= 0 && (g) < 20u && lane == 0) x3_dbg[49152 + ((dbg_tl * 16 + w) * 20 + (g)) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define X3W_DBG(x)
#define X3W_TL(g, k) do { } while (0)
#endif

  // ---- F2 + F3 for the frame in waiting: where it goes (the generation's base needs the totals of the generations
  // between this workgroup's previous one, inclusive, and this one), then header and payload to their final place
  auto finish_prev = [&]() __attribute__((always_inline)) {
    const uint32_t par = prev_gen & 7u;
    const uint32_t gtag = ((prev_gen + 1u) & 0xFFFu) << X3_DESC_BYTES_BITS;
    const uint32_t L = prev_L, frame_bytes = 20u + prev_L, rtot = (prev_L + 255u) >> 8;
    uint32_t intra = 0;
#if X3W_NOWAIT
    // timing experiment only (the stream is NOT valid): every frame at a fixed stride, nobody waits for anybody
    gen_base = prev_f * 10240ull;
#endif
    X3W_TL(prev_gen, 1);
    if (!X3W_NOWAIT) {
      // the wave's predecessors in that generation (LDS)
      uint32_t spins = 0;
      for (;;) {
        const uint32_t v = lane < w ? __hip_atomic_load(&book[16u * par + (lane & 15u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                    : gtag;
        if (!__any((v & ~X3_DESC_BYTES_MASK) != gtag)) {
          intra = (uint32_t)__builtin_amdgcn_readlane((int)x3_wave_incl_scan_dpp(v & X3_DESC_BYTES_MASK), 63);
          break;
        }
        if (++spins > X3W_SPINS_LDS ||
            ((spins & 1023u) == 0u && __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT)) {
          lost = true;
          if (spins > X3W_SPINS_LDS && lane == 0 && atomicCAS(&status[2], 0, 1) == 0) {  // diagnosis of the first wait that gave up
            status[3] = (int)prev_wgi; status[4] = (int)(w | (prev_gen << 8)); status[5] = (int)__ballot((v & ~X3_DESC_BYTES_MASK) != gtag);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    // The generation's base is the same for the sixteen waves: whoever has summed it leaves it in LDS, the others
    // take it from there (about one wave in two pays the trip to the descriptors: they arrive in clusters).
    bool have_base = X3W_NOWAIT != 0;
    if (!lost && !X3W_NOWAIT) {
      const uint32_t t = __hip_atomic_load(&book[160u + par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (t == gtag) {
        const uint32_t lo = __hip_atomic_load(&book[168u + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t hi = __hip_atomic_load(&book[176u + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        gen_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        have_base = true;
      }
    }
#if X3W_CLAIM
    // one wave per generation makes the trip; the others wait for its result in LDS (the descriptors are polled by
    // 256 waves instead of 4 096)
    if (!lost && !have_base) {
      uint32_t mine = 0;
      if (lane == 0) mine = __hip_atomic_fetch_max(&book[184u + par], prev_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != prev_gen + 1u;
      if (!(uint32_t)__builtin_amdgcn_readfirstlane((int)mine)) {
        uint32_t spins = 0;
        X3W_DBG(const unsigned long long tw0 = clock64();)
        for (;;) {
          if (__hip_atomic_load(&book[160u + par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == gtag) {
            const uint32_t lo = __hip_atomic_load(&book[168u + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t hi = __hip_atomic_load(&book[176u + par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            gen_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
            have_base = true;
            break;
          }
          if (++spins > X3W_SPINS_LDS ||
              ((spins & 1023u) == 0u && __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT)) {
            lost = true;
            break;
          }
          __builtin_amdgcn_s_sleep(X3W_SLEEP_LDS);
        }
        X3W_DBG(dbg_cnt[3] += clock64() - tw0;)
      }
    }
#endif
    if (!lost && !have_base) {
      const uint32_t need = prev_gen == 0 ? b : a.nwg;  // totals in front of generation prev_wgi that count
      const bool in0 = lane < need, in1 = lane + 64u < need, in2 = lane + 128u < need, in3 = lane + 192u < need;
      const uint32_t* p0 = a.desc + prev_wgi - 1u - lane;
      uint32_t spins = 0;
      X3W_DBG(const unsigned long long td0 = clock64(); dbg_cnt[1] += 1;)
      for (;;) {
        X3W_DBG(dbg_cnt[0] += 1;)
        const uint32_t q0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t q1 = __hip_atomic_load(p0 - 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t q2 = __hip_atomic_load(p0 - 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t q3 = __hip_atomic_load(p0 - 192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t v0 = in0 ? q0 : ready_tag, v1 = in1 ? q1 : ready_tag, v2 = in2 ? q2 : ready_tag, v3 = in3 ? q3 : ready_tag;
        if (!__any(((v0 >> X3_DESC_BYTES_BITS) != a.epoch) || ((v1 >> X3_DESC_BYTES_BITS) != a.epoch) ||
                   ((v2 >> X3_DESC_BYTES_BITS) != a.epoch) || ((v3 >> X3_DESC_BYTES_BITS) != a.epoch))) {
          // 256 totals < 2^20 each: a 32-bit sum
          const uint32_t sum = x3_wave_incl_scan_dpp((v0 & X3_DESC_BYTES_MASK) + (v1 & X3_DESC_BYTES_MASK) +
                                                     (v2 & X3_DESC_BYTES_MASK) + (v3 & X3_DESC_BYTES_MASK));
          const uint64_t from = prev_gen == 0 ? ((a.start_pos + 1ull) & ~1ull) : gen_base;  // writer.align::<2>() (encoder.rs:182)
          gen_base = from + (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);
          if (lane == 0) {
            __hip_atomic_store(&book[168u + par], (uint32_t)gen_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&book[176u + par], (uint32_t)(gen_base >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&book[160u + par], gtag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          break;
        }
        // give up after the bounded spin -- or as soon as ANY wave has: the host encodes the call again
        if (++spins > X3W_SPINS_DESC ||
            __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT) {
          lost = true;
          if (spins > X3W_SPINS_DESC && lane == 0 && atomicCAS(&status[2], 0, 2) == 0) {
            const unsigned long long nr = __ballot((v0 >> X3_DESC_BYTES_BITS) != a.epoch);
            status[3] = (int)prev_wgi; status[4] = (int)(w | (prev_gen << 8)); status[5] = (int)nr; status[6] = (int)(nr >> 32);
            status[7] = (int)(__popcll(__ballot((v1 >> X3_DESC_BYTES_BITS) != a.epoch)) | (__popcll(__ballot((v2 >> X3_DESC_BYTES_BITS) != a.epoch)) << 8) |
                              (__popcll(__ballot((v3 >> X3_DESC_BYTES_BITS) != a.epoch)) << 16));
          }
          break;
        }
        __builtin_amdgcn_s_sleep(X3W_SLEEP_DESC);
      }
      X3W_DBG(dbg_cnt[2] += clock64() - td0;)
    }
    if (lost) {
      // This wave no longer knows where its frames go -- this one and, since each base builds on the last, every
      // later one.  Nothing of them may reach the output, the frame index or the end position.
      if (lane == 0) atomicMax(&status[1], X3D_SIZE_WAIT_TIMEOUT);
      return;
    }
    const uint64_t off = gen_base + intra;
    X3W_TL(prev_gen, 2);
    X3_STAMP(5);

    if (lane == 0) {
      a.frame_off[prev_f] = off;
      if (off + frame_bytes > a.out_cap) atomicMax(&status[0], X3D_BYTE_WRITER_INSUFFICIENT_MEMORY);
      if (prev_f == a.n_frames - 1) {
        a.frame_off[a.n_frames] = off + frame_bytes;
        *end_pos = off + frame_bytes;
      }
      if (prev_f == 0 && (a.start_pos & 1ull) && a.start_pos < a.out_cap) a.out[a.start_pos] = 0;  // align pad byte
      // a frame that is not in the image: to the dense pass, now that frame_off[prev_f] says where it goes
      if (prev_ovf) a.dense_list[atomicAdd(reinterpret_cast<uint32_t*>(a.ctl + X3_CTL_DENSE_COUNT), 1u)] = (uint32_t)prev_f;
    }
    if (!prev_ovf && off + frame_bytes <= a.out_cap) {
      uint8_t* const dst = a.out + off;
      // header (encoder.rs:122-162): "x3", id, id, samples, payload_len, 8 zero time bytes, header crc over bytes
      // 0..16, payload crc; audio frames use id 1.  Ten big-endian halfwords at an even address.  The header CRC
      // (encoder.rs:153-154): the state behind the constant bytes "x3", id, id is a constant; the (samples,
      // payload_len) word and the eight zero time bytes go through the slicing tables.
      if (lane < 10) {
        // (table-free byte steps on wave-uniform values: the scalar unit's work, not the vector unit's)
        uint32_t hc = x3_crc16_const4(0x78u, 0x33u, 0x01u, 0x01u);
        hc = x3_crc_byte(hc, (prev_n >> 8) & 0xFFu);
        hc = x3_crc_byte(hc, prev_n & 0xFFu);
        hc = x3_crc_byte(hc, (L >> 8) & 0xFFu);
        hc = x3_crc_byte(hc, L & 0xFFu);
#pragma unroll
        for (int zb8 = 0; zb8 < 8; ++zb8) hc = x3_crc_byte(hc, 0u);
        const uint32_t hw = lane == 0 ? 0x7833u : lane == 1 ? 0x0101u : lane == 2 ? (prev_n & 0xFFFFu) : lane == 3 ? (L & 0xFFFFu)
                          : lane == 8 ? hc : lane == 9 ? prev_crc : 0u;
        reinterpret_cast<uint16_t*>(dst)[lane] = (uint16_t)(((hw & 0xFFu) << 8) | ((hw >> 8) & 0xFFu));
      }
      // payload: image bytes [0, L) -> dst + 20 ...; sixteen bytes per lane and trip, aligned to the DESTINATION.
      // Both are at even addresses, so a destination dword is one image dword or the halves of two.  The pieces in
      // front of the first and behind the last whole unit go out as halfwords (at most seven each).
      uint8_t* const pdst = dst + 20;
      const uint32_t dmis = (uint32_t)(reinterpret_cast<uintptr_t>(pdst) & 15u);
      const uint32_t head = (16u - dmis) & 15u;                  // payload bytes in front of the first aligned unit
      if (L >= head + 16u) {
        const uint32_t nfull = (L - head) >> 4;
        uint8_t* const abase = pdst + head;                     // 16-byte aligned
        // memory order of a stored dword = stream order: bytes (31..24), (23..16), (15..8), (7..0) of the image's value
        if (head & 2u) {
#if X3W_COPY_UNROLL == 2
#pragma unroll 2
#elif X3W_COPY_UNROLL == 3
#pragma unroll 3
#endif
          for (uint32_t u = lane; u < nfull; u += 64u) {
            const uint32_t ia = img_addr + ((head + 16u * u) & ~3u);  // the unit starts in the low half of this dword
            const uint32_t v0 = x3_lds_read_b32(ia), v1 = x3_lds_read_b32(ia + 4u), v2 = x3_lds_read_b32(ia + 8u),
                           v3 = x3_lds_read_b32(ia + 12u), v4 = x3_lds_read_b32(ia + 16u);
            const x3_u32x4 vv = {__builtin_amdgcn_perm(v1, v0, 0x06070001u), __builtin_amdgcn_perm(v2, v1, 0x06070001u),
                                 __builtin_amdgcn_perm(v3, v2, 0x06070001u), __builtin_amdgcn_perm(v4, v3, 0x06070001u)};
            *reinterpret_cast<x3_u32x4*>(abase + 16u * u) = vv;
          }
        } else {
#if X3W_COPY_UNROLL == 2
#pragma unroll 2
#elif X3W_COPY_UNROLL == 3
#pragma unroll 3
#endif
          for (uint32_t u = lane; u < nfull; u += 64u) {
            const uint32_t ia = img_addr + head + 16u * u;
            const uint32_t v0 = x3_lds_read_b32(ia), v1 = x3_lds_read_b32(ia + 4u), v2 = x3_lds_read_b32(ia + 8u),
                           v3 = x3_lds_read_b32(ia + 12u);
            const x3_u32x4 vv = {__builtin_amdgcn_perm(0u, v0, 0x00010203u), __builtin_amdgcn_perm(0u, v1, 0x00010203u),
                                 __builtin_amdgcn_perm(0u, v2, 0x00010203u), __builtin_amdgcn_perm(0u, v3, 0x00010203u)};
            *reinterpret_cast<x3_u32x4*>(abase + 16u * u) = vv;
          }
        }
        // halfword h of the payload: bits 31..16 (h even) or 15..0 (h odd) of image dword h / 2, first byte on top
        const uint32_t tail0 = head + 16u * nfull;               // first byte behind the whole units
        const uint32_t nh = head >> 1, nt = (L - tail0) >> 1;   // halfwords in front / behind: < 8 each
        if (lane < nh + nt) {
          const uint32_t hb = lane < nh ? 2u * lane : tail0 + 2u * (lane - nh);
          const uint32_t v = x3_lds_read_b32(img_addr + (hb & ~3u));
          const uint32_t hv = (hb & 2u) ? (v & 0xFFFFu) : (v >> 16);
          *reinterpret_cast<uint16_t*>(pdst + hb) = (uint16_t)(((hv & 0xFFu) << 8) | (hv >> 8));
        }
      } else {
        // a payload shorter than the first aligned unit and one more: all of it by halfwords (L < 31)
        if (lane < (L >> 1)) {
          const uint32_t hb = 2u * lane;
          const uint32_t v = x3_lds_read_b32(img_addr + (hb & ~3u));
          const uint32_t hv = (hb & 2u) ? (v & 0xFFFFu) : (v >> 16);
          *reinterpret_cast<uint16_t*>(pdst + hb) = (uint16_t)(((hv & 0xFFu) << 8) | (hv >> 8));
        }
      }
    }
    X3_STAMP(6);
    // clear what the frame used (all lanes' reads of the image are done: same wave, in order)
    x3w_lds_fence();
    if (!prev_ovf) {
      const uint32_t nq = rtot * 16u;  // 16-byte pieces
      uint32_t z0 = 0;
      asm volatile("" : "+v"(z0));    // (made here: hipcc otherwise keeps four registers of zeros for the whole kernel -- and spills them)
      for (uint32_t i = lane; i < nq; i += 64u) reinterpret_cast<uint4*>(img)[i] = make_uint4(z0, z0, z0, z0);
    }
    x3w_lds_fence();
    X3_STAMP(7);
  };

  f = wgi * a.m + w;
  if (wgi < a.n_wggen && f < a.n_frames) {
    geom_at(clip, idx, src, n);
    load_half(X0, src, n, 0);
    load_half(X1, src, n, 1);

    for (;;) {
      // (opaque per frame: nothing derived from the lane index is kept in registers across the loop -- hipcc otherwise
      // hoists a dozen lane masks and offsets out of it and spills them)
      asm volatile("" : "+v"(lane));
#if X3W_PRIO == 1
      // Feedback priorities.  The SQ issues oldest-first, so left alone the first wave of every SIMD runs a generation
      // ahead and then sits at its offset wait for the sizes of the waves it has starved, and the SIMD runs on the
      // waves that are left.  A wave's arrival rank among the sixteen of its workgroup generation says where it
      // stands: the first arrivers yield, the last ones are served first.
      if (rank >= 12u) __builtin_amdgcn_s_setprio(3);
      else if (rank >= 8u) __builtin_amdgcn_s_setprio(2);
      else if (rank >= 4u) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#elif X3W_PRIO >= 2
      // a wave that still owes its frame's size goes first: everybody's offsets wait for it
      __builtin_amdgcn_s_setprio(X3W_PRIO);
#endif
      const uint32_t tail = (n - 1u) % 20u;
      const bool plain = tail == 0u || tail == 19u;  // every block has 20 samples, or 19 in the frame's last block
#ifdef X3_DBG_STAMPS
      x3_dma_wait();
      X3_STAMP(0);
#endif

      // ---- B: analysis.  rem = samples behind this lane's predecessor sample; block q has clamp(rem - 20 q, 0, 20).
      const int32_t rem0 = (int32_t)n - 1 - 80 * (int32_t)lane, rem1 = rem0 - (int32_t)X3W_PART;
      uint32_t nb0 = 0, nb1 = 0, mA, mB, mC, mD;  // metas: blocks 0, 1 | 2, 3 of half 0, of half 1 (16 bits each)
      {
        uint32_t t, m0, m1, m2, m3;
        const __amdgpu_buffer_rsrc_t rs_cur =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
        const uint32_t vo_cur = 160u * lane;
        auto stat = [&](uint32_t mt, uint32_t cnt) __attribute__((always_inline)) {
          // statistics (encoder.rs:199): stats[index] += block.len(), summed per workgroup in LDS
          // (eight copies of the six counters, by lane: sixty-four lanes adding to two or three addresses serialise in
          // the LDS -- 4 % of the kernel's time with one copy)
          if (mt) atomicAdd(&book[192u + 8u * (lane & 7u) + ((mt >> 12) & 7u)], cnt);
        };
        // (blocks of 10, generic path: eight blocks of a lane's half one by one, their metas in the 8-bit form)
        auto slow10 = [&](uint32_t bi0, int32_t rem, uint32_t& nbh, uint32_t& ma, uint32_t& mb, uint32_t& mc, uint32_t& md)
                          __attribute__((always_inline)) {
          uint32_t mm[4] = {0, 0, 0, 0};
          for (uint32_t i = 0; i < 8u; ++i) {
            const int32_t c = rem - 10 * (int32_t)i;
            const uint32_t cu = c <= 0 ? 0u : (c < 10 ? (uint32_t)c : 10u);
            uint32_t tt, m15;
            x3w_slow_analyse<10u>(rs_cur, bi0 + i, cu, a.thr0, a.thr1, a.thr2, a.kpack, tt, m15);
            nbh += tt;
            stat(m15, cu);
            mm[i >> 1] |= x3w_meta8(m15) << (8u * (i & 1u));
          }
          ma = mm[0]; mb = mm[1]; mc = mm[2]; md = mm[3];
        };
        auto stat10 = [&](uint32_t m, uint32_t cnt) __attribute__((always_inline)) {   // a run's two blocks
          stat(x3w_meta_of(m & 0xFFu, a.kpack), 10u);
          stat(x3w_meta_of(m >> 8, a.kpack), cnt - 10u);
        };
        if constexpr (BL == 10u) {
          if (plain) {
            x3w_analyse10<0>(X0, x3w_cnt_of(rem0, 0), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m0); nb0 += t;
            x3w_analyse10<1>(X0, x3w_cnt_of(rem0, 1), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m1); nb0 += t;
            x3w_analyse10<2>(X0, x3w_cnt_of(rem0, 2), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m2); nb0 += t;
            x3w_analyse10<3>(X0, x3w_cnt_of(rem0, 3), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m3); nb0 += t;
            stat10(m0, x3w_cnt_of(rem0, 0)); stat10(m1, x3w_cnt_of(rem0, 1)); stat10(m2, x3w_cnt_of(rem0, 2)); stat10(m3, x3w_cnt_of(rem0, 3));
          } else {
            slow10(8u * lane, rem0, nb0, m0, m1, m2, m3);
          }
        } else if constexpr (BL == 40u) {
          // (blocks of 40: runs 0+1 and 2+3; the generic path takes a whole block at the first run's meta)
          const uint32_t c0 = x3w_cnt_of(rem0, 0), c1 = x3w_cnt_of(rem0, 1), c2 = x3w_cnt_of(rem0, 2), c3 = x3w_cnt_of(rem0, 3);
          if (plain) {
            x3w_analyse40<0>(X0, c0, c1, a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m0, m1); nb0 += t;
            x3w_analyse40<1>(X0, c2, c3, a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m2, m3); nb0 += t;
            stat(m0, c0 + c1); stat(m2, c2 + c3);
          } else {
            x3w_slow_analyse<40u>(rs_cur, 2u * lane + 0u, c0 + c1, a.thr0, a.thr1, a.thr2, a.kpack, t, m0); nb0 += t;
            x3w_slow_analyse<40u>(rs_cur, 2u * lane + 1u, c2 + c3, a.thr0, a.thr1, a.thr2, a.kpack, t, m2); nb0 += t;
            m1 = m3 = 0;
            stat(m0, c0 + c1); stat(m2, c2 + c3);
          }
        } else {
          if (plain) {
            x3w_analyse<0>(X0, x3w_cnt_of(rem0, 0), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m0); nb0 += t;
            x3w_analyse<1>(X0, x3w_cnt_of(rem0, 1), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m1); nb0 += t;
            x3w_analyse<2>(X0, x3w_cnt_of(rem0, 2), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m2); nb0 += t;
            x3w_analyse<3>(X0, x3w_cnt_of(rem0, 3), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 0u, t, m3); nb0 += t;
          } else {
            x3w_slow_analyse(rs_cur, 4u * lane + 0u, x3w_cnt_of(rem0, 0), a.thr0, a.thr1, a.thr2, a.kpack, t, m0); nb0 += t;
            x3w_slow_analyse(rs_cur, 4u * lane + 1u, x3w_cnt_of(rem0, 1), a.thr0, a.thr1, a.thr2, a.kpack, t, m1); nb0 += t;
            x3w_slow_analyse(rs_cur, 4u * lane + 2u, x3w_cnt_of(rem0, 2), a.thr0, a.thr1, a.thr2, a.kpack, t, m2); nb0 += t;
            x3w_slow_analyse(rs_cur, 4u * lane + 3u, x3w_cnt_of(rem0, 3), a.thr0, a.thr1, a.thr2, a.kpack, t, m3); nb0 += t;
          }
          stat(m0, x3w_cnt_of(rem0, 0)); stat(m1, x3w_cnt_of(rem0, 1)); stat(m2, x3w_cnt_of(rem0, 2)); stat(m3, x3w_cnt_of(rem0, 3));
        }
        mA = m0 | (m1 << 16);
        mB = m2 | (m3 << 16);
        mC = 0;
        mD = 0;
        if (X3W_SKIP_EMPTY_HALF && n <= X3W_PART + 1u) {
          // (a frame of at most 5 121 samples has nothing in its second half: no analysis, no emission, no loads for it --
          // 256 blocks a frame 0.58 -> 0.48 ms, 100 blocks 1.34 -> 1.06; config 3 the same)
        } else {
        if constexpr (BL == 10u) {
          if (plain) {
            x3w_analyse10<0>(X1, x3w_cnt_of(rem1, 0), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m0); nb1 += t;
            x3w_analyse10<1>(X1, x3w_cnt_of(rem1, 1), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m1); nb1 += t;
            x3w_analyse10<2>(X1, x3w_cnt_of(rem1, 2), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m2); nb1 += t;
            x3w_analyse10<3>(X1, x3w_cnt_of(rem1, 3), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m3); nb1 += t;
            stat10(m0, x3w_cnt_of(rem1, 0)); stat10(m1, x3w_cnt_of(rem1, 1)); stat10(m2, x3w_cnt_of(rem1, 2)); stat10(m3, x3w_cnt_of(rem1, 3));
          } else {
            slow10(512u + 8u * lane, rem1, nb1, m0, m1, m2, m3);
          }
        } else if constexpr (BL == 40u) {
          const uint32_t c0 = x3w_cnt_of(rem1, 0), c1 = x3w_cnt_of(rem1, 1), c2 = x3w_cnt_of(rem1, 2), c3 = x3w_cnt_of(rem1, 3);
          if (plain) {
            x3w_analyse40<0>(X1, c0, c1, a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m0, m1); nb1 += t;
            x3w_analyse40<1>(X1, c2, c3, a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m2, m3); nb1 += t;
          } else {
            x3w_slow_analyse<40u>(rs_cur, 128u + 2u * lane + 0u, c0 + c1, a.thr0, a.thr1, a.thr2, a.kpack, t, m0); nb1 += t;
            x3w_slow_analyse<40u>(rs_cur, 128u + 2u * lane + 1u, c2 + c3, a.thr0, a.thr1, a.thr2, a.kpack, t, m2); nb1 += t;
            m1 = m3 = 0;
          }
          stat(m0, c0 + c1); stat(m2, c2 + c3);
        } else {
          if (plain) {
            x3w_analyse<0>(X1, x3w_cnt_of(rem1, 0), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m0); nb1 += t;
            x3w_analyse<1>(X1, x3w_cnt_of(rem1, 1), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m1); nb1 += t;
            x3w_analyse<2>(X1, x3w_cnt_of(rem1, 2), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m2); nb1 += t;
            x3w_analyse<3>(X1, x3w_cnt_of(rem1, 3), a.thr0, a.thr1, a.thr2, a.kpack, rs_cur, vo_cur, 2u * X3W_PART, t, m3); nb1 += t;
          } else {
            x3w_slow_analyse(rs_cur, 256u + 4u * lane + 0u, x3w_cnt_of(rem1, 0), a.thr0, a.thr1, a.thr2, a.kpack, t, m0); nb1 += t;
            x3w_slow_analyse(rs_cur, 256u + 4u * lane + 1u, x3w_cnt_of(rem1, 1), a.thr0, a.thr1, a.thr2, a.kpack, t, m1); nb1 += t;
            x3w_slow_analyse(rs_cur, 256u + 4u * lane + 2u, x3w_cnt_of(rem1, 2), a.thr0, a.thr1, a.thr2, a.kpack, t, m2); nb1 += t;
            x3w_slow_analyse(rs_cur, 256u + 4u * lane + 3u, x3w_cnt_of(rem1, 3), a.thr0, a.thr1, a.thr2, a.kpack, t, m3); nb1 += t;
          }
          stat(m0, x3w_cnt_of(rem1, 0)); stat(m1, x3w_cnt_of(rem1, 1)); stat(m2, x3w_cnt_of(rem1, 2)); stat(m3, x3w_cnt_of(rem1, 3));
        }
        mC = m0 | (m1 << 16);
        mD = m2 | (m3 << 16);
        }
      }
      X3_STAMP(1);
#if X3W_BARRIER
      // experiment: the sixteen waves of the workgroup meet behind their analyses (waves that have ended do not count)
      __builtin_amdgcn_s_barrier();
#endif

      // ---- C: bit offsets (the BitPacker's running position as two wave scans)
      uint32_t excl0, excl1, tot0, tot1;  // (only the exclusive sums stay: whether a lane has blocks is in its metas)
      {
        const uint32_t incl0 = x3_wave_incl_scan_dpp(nb0), incl1 = x3_wave_incl_scan_dpp(nb1);
        tot0 = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
        tot1 = (uint32_t)__builtin_amdgcn_readlane((int)incl1, 63);
        excl0 = incl0 - nb0;
        excl1 = incl1 - nb1;
      }
      if (a.seg) {
        // a lane's first block of half h is block 256 h + 4 lane of the frame; it begins 16 + (tot0 if h) + excl bits into the
        // payload.  Entry k (1 .. pitch) is for block k << seg_log2; blocks the frame does not have get no entry (zero).
        const uint32_t nbf = (n - 1u + 19u) / 20u, msk = (1u << a.seg_log2) - 1u;
        uint2* const row = a.seg + 1 + f * (uint64_t)a.seg_pitch - 1;   // (entry k at row[k])
#pragma unroll
        for (uint32_t h = 0; h < 2u; ++h) {
          const uint32_t bk = 256u * h + 4u * lane, k = bk >> a.seg_log2;
          if (bk && (bk & msk) == 0u && k <= a.seg_pitch) {
            uint2 e = make_uint2(0u, 0u);
            if (bk < nbf) e = make_uint2(16u + (h ? tot0 + excl1 : excl0), (uint32_t)(uint16_t)src[20u * bk] | 0x10000u);
            row[k] = e;
          }
        }
      }
      const uint32_t bits = 16u + tot0 + tot1;                 // <Audio State> + blocks (encoder.rs:189-200)
      const uint32_t L = (((bits + 7u) >> 3) + 1u) & ~1u;      // word_align (bitpacker.rs:124-132)
      const uint32_t rtot = (L + 255u) >> 8;                   // image rows that hold payload
      const bool ovf = L > X3W_IMG_BYTES;

      // ---- F1: this frame's size to the workgroup; the last arriver of the generation publishes its total
      // (the word is its own flag: cdna_hip_programming.md G16, form R2)
      uint32_t arrived = 0;
      if (lane == 0) {
        const uint64_t left_f = a.n_frames - wgi * a.m;
        const uint32_t m_eff = left_f < a.m ? (uint32_t)left_f : a.m;
        const uint32_t par = gen & 7u;
        const uint32_t gtag = ((gen + 1u) & 0xFFFu) << X3_DESC_BYTES_BITS;
        __hip_atomic_store(&book[16u * par + w], gtag | (20u + L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t old = __hip_atomic_fetch_add(&book[128u + par], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = old;
        if (old == m_eff - 1u) {
          uint32_t tb = 0;
          for (uint32_t i = 0; i < m_eff; ++i)
            tb += __hip_atomic_load(&book[16u * par + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & X3_DESC_BYTES_MASK;
          __hip_atomic_store(&book[128u + par], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (wgi != a.drop_wgi) __hip_atomic_store(&a.desc[wgi], ready_tag | tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
      X3W_TL(gen, 0);
#if X3W_PRIO >= 2
      __builtin_amdgcn_s_setprio(0);
#endif
      // the geometry of this wave's next frame
      bool have_next;
      uint32_t n_next = 0;
      const int16_t* src_next = nullptr;
      {
        const uint64_t wgi_n = wgi + a.nwg;
        have_next = wgi_n < a.n_wggen && wgi_n * a.m + w < a.n_frames;
        if (TAB) {
          clip += (uint64_t)a.nwg * a.m;
        } else {
          idx += a.step_idx;
          clip += a.step_clip;
          if (idx >= a.fpc) { idx -= a.fpc; ++clip; }
        }
        if (have_next) geom_at(clip, idx, src_next, n_next);
      }
      X3_STAMP(2);

      // ---- the frame in waiting leaves its image now (its size went out a whole iteration ago)
#ifndef X3W_SKEW2_TIMING
      if (have_prev) {
        finish_prev();
        if (lost) break;
      }
#endif

      // ---- D: emission, half 0 then half 1; the next frame's halves are requested as their registers fall free.
      // Lane 0 starts with the frame's first sample.  (The block sizes are worked out again rather than kept from the
      // analysis: registers.)
      int32_t rem0e, rem1e;
      {
        uint32_t lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        rem0e = (int32_t)n - 1 - 80 * (int32_t)lane_e;
        rem1e = rem0e - (int32_t)X3W_PART;
      }
      // (blocks of 10: a run's two blocks from its two 8-bit metas)
      auto emit10 = [&](const uint32_t (&X)[41], uint32_t mlo, uint32_t mhi, int32_t rem, X3WEmit& e) __attribute__((always_inline)) {
        x3w_emit<0, false, 0, 5>(X, x3w_meta_of(mlo & 0xFFu, a.kpack), x3w_cnt_of(rem, 0), e);
        x3w_emit<0, false, 5, 10>(X, x3w_meta_of((mlo >> 8) & 0xFFu, a.kpack), x3w_cnt_of(rem, 0), e);
        x3w_emit<1, false, 0, 5>(X, x3w_meta_of((mlo >> 16) & 0xFFu, a.kpack), x3w_cnt_of(rem, 1), e);
        x3w_emit<1, false, 5, 10>(X, x3w_meta_of(mlo >> 24, a.kpack), x3w_cnt_of(rem, 1), e);
        x3w_emit<2, false, 0, 5>(X, x3w_meta_of(mhi & 0xFFu, a.kpack), x3w_cnt_of(rem, 2), e);
        x3w_emit<2, false, 5, 10>(X, x3w_meta_of((mhi >> 8) & 0xFFu, a.kpack), x3w_cnt_of(rem, 2), e);
        x3w_emit<3, false, 0, 5>(X, x3w_meta_of((mhi >> 16) & 0xFFu, a.kpack), x3w_cnt_of(rem, 3), e);
        x3w_emit<3, false, 5, 10>(X, x3w_meta_of(mhi >> 24, a.kpack), x3w_cnt_of(rem, 3), e);
      };
      auto emit10_slow = [&](__amdgpu_buffer_rsrc_t rs_, uint32_t bi0, uint32_t mlo, uint32_t mhi, int32_t rem, X3WEmit& e)
                             __attribute__((always_inline)) {
        for (uint32_t i = 0; i < 8u; ++i) {
          const int32_t c = rem - 10 * (int32_t)i;
          const uint32_t cu = c <= 0 ? 0u : (c < 10 ? (uint32_t)c : 10u);
          const uint32_t m8 = ((i < 4u ? mlo : mhi) >> (8u * (i & 3u))) & 0xFFu;
          x3w_slow_emit<10u>(rs_, bi0 + i, cu, x3w_meta_of(m8, a.kpack), e);
        }
      };
      if (!ovf) {
        X3WEmit e;
        e.start(lane ? 16u + excl0 : 0u, img_addr);
        if (lane == 0) e.put(*reinterpret_cast<const uint32_t*>(src) & 0xFFFFu, 16u);  // the frame's first sample (frames are 16-byte aligned)
        if constexpr (BL == 10u) {
          if (plain) {
            emit10(X0, mA, mB, rem0e, e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            emit10_slow(rs_cur, 8u * lane, mA, mB, rem0e, e);
          }
        } else if constexpr (BL == 40u) {
          if (plain) {
            x3w_emit<0>(X0, mA & 0xFFFFu, x3w_cnt_of(rem0e, 0), e); x3w_emit<1, true>(X0, mA >> 16, x3w_cnt_of(rem0e, 1), e);
            x3w_emit<2>(X0, mB & 0xFFFFu, x3w_cnt_of(rem0e, 2), e); x3w_emit<3, true>(X0, mB >> 16, x3w_cnt_of(rem0e, 3), e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            x3w_slow_emit<40u>(rs_cur, 2u * lane + 0u, x3w_cnt_of(rem0e, 0) + x3w_cnt_of(rem0e, 1), mA & 0xFFFFu, e);
            x3w_slow_emit<40u>(rs_cur, 2u * lane + 1u, x3w_cnt_of(rem0e, 2) + x3w_cnt_of(rem0e, 3), mB & 0xFFFFu, e);
          }
        } else {
          if (plain) {
            x3w_emit<0>(X0, mA & 0xFFFFu, x3w_cnt_of(rem0e, 0), e); x3w_emit<1>(X0, mA >> 16, x3w_cnt_of(rem0e, 1), e);
            x3w_emit<2>(X0, mB & 0xFFFFu, x3w_cnt_of(rem0e, 2), e); x3w_emit<3>(X0, mB >> 16, x3w_cnt_of(rem0e, 3), e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            x3w_slow_emit(rs_cur, 4u * lane + 0u, x3w_cnt_of(rem0e, 0), mA & 0xFFFFu, e);
            x3w_slow_emit(rs_cur, 4u * lane + 1u, x3w_cnt_of(rem0e, 1), mA >> 16, e);
            x3w_slow_emit(rs_cur, 4u * lane + 2u, x3w_cnt_of(rem0e, 2), mB & 0xFFFFu, e);
            x3w_slow_emit(rs_cur, 4u * lane + 3u, x3w_cnt_of(rem0e, 3), mB >> 16, e);
          }
        }
        if ((mA | mB) || lane == 0) e.finish();
      }
      if (have_next) load_half(X0, src_next, n_next, 0);
      if (!ovf && !(X3W_SKIP_EMPTY_HALF && n <= X3W_PART + 1u)) {
        X3WEmit e;
        e.start(16u + tot0 + excl1, img_addr);
        if constexpr (BL == 10u) {
          if (plain) {
            emit10(X1, mC, mD, rem1e, e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            emit10_slow(rs_cur, 512u + 8u * lane, mC, mD, rem1e, e);
          }
        } else if constexpr (BL == 40u) {
          if (plain) {
            x3w_emit<0>(X1, mC & 0xFFFFu, x3w_cnt_of(rem1e, 0), e); x3w_emit<1, true>(X1, mC >> 16, x3w_cnt_of(rem1e, 1), e);
            x3w_emit<2>(X1, mD & 0xFFFFu, x3w_cnt_of(rem1e, 2), e); x3w_emit<3, true>(X1, mD >> 16, x3w_cnt_of(rem1e, 3), e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            x3w_slow_emit<40u>(rs_cur, 128u + 2u * lane + 0u, x3w_cnt_of(rem1e, 0) + x3w_cnt_of(rem1e, 1), mC & 0xFFFFu, e);
            x3w_slow_emit<40u>(rs_cur, 128u + 2u * lane + 1u, x3w_cnt_of(rem1e, 2) + x3w_cnt_of(rem1e, 3), mD & 0xFFFFu, e);
          }
        } else {
          if (plain) {
            x3w_emit<0>(X1, mC & 0xFFFFu, x3w_cnt_of(rem1e, 0), e); x3w_emit<1>(X1, mC >> 16, x3w_cnt_of(rem1e, 1), e);
            x3w_emit<2>(X1, mD & 0xFFFFu, x3w_cnt_of(rem1e, 2), e); x3w_emit<3>(X1, mD >> 16, x3w_cnt_of(rem1e, 3), e);
          } else {
            const __amdgpu_buffer_rsrc_t rs_cur =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
            x3w_slow_emit(rs_cur, 256u + 4u * lane + 0u, x3w_cnt_of(rem1e, 0), mC & 0xFFFFu, e);
            x3w_slow_emit(rs_cur, 256u + 4u * lane + 1u, x3w_cnt_of(rem1e, 1), mC >> 16, e);
            x3w_slow_emit(rs_cur, 256u + 4u * lane + 2u, x3w_cnt_of(rem1e, 2), mD & 0xFFFFu, e);
            x3w_slow_emit(rs_cur, 256u + 4u * lane + 3u, x3w_cnt_of(rem1e, 3), mD >> 16, e);
          }
        }
        if (mC | mD) e.finish();
      }
      X3_STAMP(3);
      x3w_lds_fence();

      // ---- E: payload CRC-16 (crc.rs:44-58 as a segmented reduction, see the file header)
      uint32_t crc = 0;
      if (!ovf) {
        // Two independent Horner chains, over the even and the odd rows counted from the END of the payload (times
        // x^4096 per step): the look-ups of one chain fly under those of the other.  An odd number of rows starts with a
        // row of zeros in front, which adds nothing.  A chain's state is a 32-bit polynomial A that is only CONGRUENT to
        // the sum so far:  A' = (A * x^4096 mod P) ^ dword  -- four look-ups, one per byte of A in the table of its
        // weight, and no reduction of the incoming dword at all (until round 3: four look-ups to reduce the dword to
        // crc0 form and two more to move the 16-bit state on).
        auto m4096 = [&](uint32_t v) __attribute__((always_inline)) -> uint32_t {
          const uint32_t a3 = x3_sdwa_byte_x2(v, 3), a2 = x3_sdwa_byte_x2(v, 2), a1 = x3_sdwa_byte_x2(v, 1), a0 = x3_sdwa_byte_x2(v, 0);
          return (uint32_t)x3_lds_read_u16(tab_base + a3, 1536u) ^ (uint32_t)x3_lds_read_u16(tab_base + a2, 1024u) ^
                 (uint32_t)x3_lds_read_u16(tab_base + a1, 512u) ^ (uint32_t)x3_lds_read_u16(tab_base + a0, 0u);
        };
        uint32_t sa = 0, sb = 0;
        const uint32_t odd = rtot & 1u;
        uint32_t ra = img_addr + 4u * lane;
        {
          // the first step: row 0 is in chain b if the count is odd (chain a sees the row of zeros), else in chain a
          uint32_t d0 = x3_lds_read_b32(ra);
          if (lane == 0) d0 ^= 0xFFFF0000u;  // CRC init 0xFFFF folded into the first 16 message bits
          if (odd) {
            sb = d0;
            ra += 256u;
          } else {
            sa = d0;
            sb = x3_lds_read_b32(ra + 256u);
            ra += 512u;
          }
        }
        // (four rows a trip: their words are in flight together.  Preparing the next trip's words and look-ups under
        // this trip's chain steps as well -- software pipelining by hand -- made the pass slower: with sixteen waves the
        // LDS pipe is bound by the look-ups' bank conflicts, not by their latency)
        uint32_t r = 2u - odd;
        for (; r + 3u < rtot; r += 4u) {
          const uint32_t da0 = x3_lds_read_b32(ra), db0 = x3_lds_read_b32(ra + 256u), da1 = x3_lds_read_b32(ra + 512u),
                         db1 = x3_lds_read_b32(ra + 768u);
          ra += 1024u;
          sa = m4096(sa) ^ da0;
          sb = m4096(sb) ^ db0;
          sa = m4096(sa) ^ da1;
          sb = m4096(sb) ^ db1;
        }
        if (r < rtot) {
          const uint32_t da = x3_lds_read_b32(ra), db = x3_lds_read_b32(ra + 256u);
          sa = m4096(sa) ^ da;
          sb = m4096(sb) ^ db;
        }
        // both states times x^4096, reduced to 16 bits; chain a ends one row in front of chain b.  s = (the lane's column
        // as a polynomial) * x^4096 mod P -- the factor x^(16 - 4096) that makes a CRC of it is in the lanes' weights
        sa = m4096(sa);
        sb = m4096(sb);
        const uint32_t s = (uint32_t)x3_lds_read_u16(tab_base + x3_sdwa_byte_x2(sa, 1), 2048u) ^
                           (uint32_t)x3_lds_read_u16(tab_base + x3_sdwa_byte_x2(sa, 0), 2560u) ^ sb;
        // times this lane's x^(32*(63-lane) + 16 - 4096): sixteen pre-shifted words
        const x3_u32x4 k0 = x3_lds_read_b128(tab_base + 3072u + 32u * lane);
        const x3_u32x4 k1 = x3_lds_read_b128(tab_base + 3072u + 32u * lane + 16u);
        const uint32_t kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
        uint32_t rr = 0;
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) {
          const uint32_t kv = (bit & 1) ? (kk[bit >> 1] >> 16) : (kk[bit >> 1] & 0xFFFFu);
          rr ^= (0u - ((s >> bit) & 1u)) & kv;
        }
        crc = (uint32_t)__builtin_amdgcn_readlane((int)x3_wave_xor_to_lane63_dpp(rr), 63) & 0xFFFFu;
        // the rows counted 256 rtot bytes, the payload has L: undo the zero bytes behind it
        const uint32_t zb = 256u * rtot - L;
        if (zb) {
          const uint32_t kx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tab[2560u + (zb >> 1)]);
          crc = x3_gf_mul(crc, kx) & 0xFFFFu;
        }
      }
      // the next frame's second half: behind the CRC pass (whose look-ups want the registers), in front of the
      // analysis of its first half
      if (have_next && !(X3W_SKIP_EMPTY_HALF && n_next <= X3W_PART + 1u)) load_half(X1, src_next, n_next, 1);
      X3_STAMP(4);

#ifdef X3W_SKEW2_TIMING
      // timing experiment only (the stream is NOT valid): the frame before leaves its image BEHIND this frame's emission
      // and CRC pass, as it could if a wave had room for two images
      if (have_prev) {
        finish_prev();
        if (lost) break;
      }
#endif
      // ---- this frame waits in its image; the wave goes on to its next one
      have_prev = true;
      prev_ovf = ovf;
      prev_f = f;
      prev_wgi = wgi;
      prev_gen = gen;
      prev_n = n;
      prev_L = L;
      prev_crc = crc;
      if (!have_next) break;
      ++gen;
      wgi += a.nwg;
      f = wgi * a.m + w;
      src = src_next;
      n = n_next;
    }
    if (have_prev && !lost) finish_prev();
  }

#ifdef X3_DBG_STAMPS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) x3_dbg[(blockIdx.x * 16 + w) * 8 + k] = dbg_acc[k];
  if (lane == 0)
    for (int k = 0; k < 4; ++k) x3_dbg[32768 + (blockIdx.x * 16 + w) * 4 + k] = dbg_cnt[k];
#endif
  if (blockIdx.x == 0 && w == 0 && lane == 0) {   // the launch log: the shader clock this launch ran at (x3_ctx_launch_log)
    uint32_t* const lg = a.log + X3_LOG_WORDS * (a.log_epoch & (X3_LOG_ENTRIES - 1u));
    lg[0] = a.log_epoch << 20;
    lg[1] = 0;
    lg[2] = (uint32_t)(clock64() - log_c0);
    lg[3] = (uint32_t)(wall_clock64() - log_w0);
  }
  // statistics: the last wave to leave adds the workgroup's sums
  if (lane == 0) {
    const uint32_t old = __hip_atomic_fetch_add(&book[136], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == a.m - 1u) {
      for (uint32_t i = 0; i < 6; ++i) {
        uint32_t v = 0;
        for (uint32_t c8 = 0; c8 < 8; ++c8) v += __hip_atomic_load(&book[192u + 8u * c8 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v) atomicAdd(&stats[i], (unsigned long long)v);
      }
    }
  }
}
