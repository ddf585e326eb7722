// x3_decode_mc_kernel.h -- the multi-channel extension's decoder on the lane-per-frame machinery (round 4).
//
// Not in the reference (decoder.rs:90-94 refuses frames with more than one channel; oracle/x3_oracle.c,
// x3o_decode_stream_mc, says what the extension's layout is): payload = the first sample of channel 0 .. C-1 (16 bits
// each), then for every block index the block of channel 0, 1, .. C-1 -- each a mono block (decoder.rs:132-235) against
// its own channel's last sample -- then word_align.  Until round 4 this was ONE THREAD per frame over the reference's
// byte-wise reader (x3_decode_mc_kernel in x3_decode_replay.h: 1.8 Gsamples/s on the device).  Here a frame is a LANE,
// as in x3_decode_lanes_kernel, whose input ring, 64-bit bit window and block arithmetic this kernel shares: the bit
// stream of a frame is serial whatever the number of channels, the channels only say whose `last` a block continues
// and where its samples go.
//   * per channel and lane: the running sample (`last`) and a 16-byte staging slot in LDS; a slot is stored with ONE
//     global_store_dwordx4 when its eight samples are complete (rows 16-byte aligned: frame sample offsets and the
//     channel stride multiples of eight -- otherwise, and at a frame's ragged end, sample by sample);
//   * errors are not decided here: a frame with a table-bound or BFP error, a zero run of 32 bits or more, or a read
//     beyond its payload is flagged X3D_REPLAY and goes through the reference's reader (x3_decode_mc_kernel, which
//     then only takes the flagged frames) -- the same division of labour as in the mono decoders.
// Frames the check kernel has refused (cstatus[f] != 0) are skipped.
#pragma once
#include "x3_decode_kernel.h"

__global__ void __launch_bounds__(64)
x3_decode_mc_lanes_kernel(const uint8_t* __restrict__ x3, const uint64_t* __restrict__ frame_off,
                          const uint64_t* __restrict__ wav_off, uint64_t n_frames, X3DevParams p, uint32_t n_ch,
                          int16_t* __restrict__ wav, uint64_t ch_stride, uint64_t wav_cap,
                          const int32_t* __restrict__ cstatus, int32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint32_t ring[64 * X3_DEC_RING_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t slot[X3_MAX_CHANNELS * 64 * 4];  // [channel][lane] eight samples
  __shared__ int32_t s_last[X3_MAX_CHANNELS * 64];

  const uint32_t lane = threadIdx.x;
  const uint64_t f = (uint64_t)blockIdx.x * 64 + lane;
  uint32_t* const row = ring + lane * X3_DEC_RING_STRIDE;

  // ---- per-lane frame setup (the check kernel has validated the header: key, CRC, channel count, length)
  bool active = f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 2u * n_ch;
  uint64_t p0 = 0, wo = 0;
  if (active && cstatus[f] != X3D_OK) active = false;   // (its status comes from the check pass)
  if (active) {
    const uint8_t* __restrict__ h = x3 + frame_off[f];
    samples = ((uint32_t)h[4] << 8) | h[5];
    plen = ((uint32_t)h[6] << 8) | h[7];
    p0 = frame_off[f] + 20;
    wo = wav_off[f];
    if (samples == 0u || plen < 2u * n_ch || wo + samples > wav_cap) {
      st = X3D_BAD_ARG;
      active = false;
    }
  }
  if (!active) { p0 = 0; plen = 2u * n_ch; wo = 0; samples = 0; }
  // rows that can be stored sixteen bytes at a time (wave-uniform test on the uniform parts; the lane's own offset below)
  const bool aligned = active && ((wo | ch_stride) & 7ull) == 0 && (reinterpret_cast<uintptr_t>(wav) & 15u) == 0;

  // ---- input ring (as in x3_decode_lanes_kernel)
  const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
  const uint8_t* __restrict__ const x3b = x3 - adj;
  const uint64_t v_end = adj + p0 + plen;
  const uint64_t v_bits = adj + p0 + 2ull * n_ch;       // the bit stream starts behind the channels' first samples
  const uint64_t v_last = (v_end - 1) & ~15ull;
  uint64_t v_next = v_bits & ~15ull;
  uint32_t wr_abs = 0, rd_abs = 0;
  auto request = [&](uint64_t v) -> uint4 {
    const uint64_t a = v < v_last ? v : v_last;
    return *reinterpret_cast<const uint4*>(x3b + a);
  };
  auto park = [&](uint4 c, uint64_t v) {
    const int64_t left = (int64_t)(v_end - v);
    if (left < 16) {
      uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int64_t r = left - 4 * d;
        w[d] = r >= 4 ? w[d] : (r <= 0 ? 0u : (w[d] & ((1u << (8u * (uint32_t)r)) - 1u)));
      }
      c = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(row + (wr_abs & (X3_DEC_RING_DW - 1u))) = c;
    wr_abs += 4;
  };
  uint64_t win = 0;
  uint32_t have = 0;
  bool deferred = false;
  uint32_t nextw_raw = 0;
  {
    uint4 c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) park(c[k], v_next + 16u * k);
    v_next += 128;
    const uint32_t skip = (uint32_t)(v_bits & 15u);
    rd_abs = skip >> 2;
    const uint32_t a = skip & 3u;
    const uint32_t w0 = x3_bswap32(row[rd_abs & (X3_DEC_RING_DW - 1u)]);
    ++rd_abs;
    win = (uint64_t)w0 << (32u + 8u * a);
    have = 32u - 8u * a;
    nextw_raw = row[rd_abs & (X3_DEC_RING_DW - 1u)];
  }
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
  auto service = [&]() {
    const uint32_t free_dw = X3_DEC_RING_DW - (wr_abs - rd_abs) - 1u;
    const uint32_t fit = free_dw >> 2;
    if (fit > 0) park(ld0, v_req);
    if (fit > 1) park(ld1, v_req + 16);
    if (fit > 2) park(ld2, v_req + 32);
    v_next += 16u * (fit > 3u ? 3u : fit);
    v_req = v_next;
    ld0 = request(v_req);
    ld1 = request(v_req + 16);
    ld2 = request(v_req + 32);
  };

  // ---- the channels' first samples (<Audio State>), staged like every other sample
  uint16_t* const slot16 = reinterpret_cast<uint16_t*>(slot);
  auto put = [&](uint32_t c, uint32_t i, int32_t v) {   // sample i of channel c of this lane's frame
    int16_t* const o = wav + (uint64_t)c * ch_stride + wo;
    if (!aligned) {
      o[i] = (int16_t)v;
      return;
    }
    slot16[((c * 64u + lane) << 3) + (i & 7u)] = (uint16_t)v;
    if ((i & 7u) == 7u) {        // the slot is complete: one 16-byte store (a wave's DS instructions execute in order)
      // (the halfword stores above and this vector load alias through different types: keep the compiler from moving
      // the load above the last store -- ADVICE r4)
      X3_WAVE_LDS_ORDER();
      const uint4 q = *reinterpret_cast<const uint4*>(slot + ((c * 64u + lane) << 2));
      const x3_u32x4 vv = {q.x, q.y, q.z, q.w};
      x3_store_stream16(o + (i & ~7u), vv);
    } else if (i + 1u == samples) {  // the frame's ragged end: sample by sample
      for (uint32_t k = i & ~7u; k <= i; ++k) o[k] = (int16_t)slot16[((c * 64u + lane) << 3) + (k & 7u)];
    }
  };
  if (active) {
    for (uint32_t c = 0; c < n_ch; ++c) {
      const int32_t first = (int16_t)(uint16_t)(((uint32_t)x3[p0 + 2u * c] << 8) | x3[p0 + 2u * c + 1u]);
      s_last[c * 64u + lane] = first;
      put(c, 0u, first);
    }
  }
  X3_WAVE_LDS_ORDER();

  // ---- lock-step decode: block index, channel and in-block sample index are wave-uniform
  const uint32_t bl = p.block_len;
  uint32_t remaining = samples ? samples - 1u : 0u;
  uint32_t i0 = 1;   // uniform index of the first sample of the current block
  for (;;) {
    uint32_t cnt = remaining < bl ? remaining : bl;
    const uint32_t maxcnt = __any(cnt == bl) ? bl : x3_wave_max_u32(cnt);
    if (maxcnt == 0) break;
    for (uint32_t c = 0; c < n_ch; ++c) {
      service();
      int32_t last = s_last[c * 64u + lane];
      uint32_t zmask = 0, width = 1, bound = 0xFFFFFFFFu, level = 0, lit = 0, neg_thresh = 0xFFFFFFFFu, neg2 = 0;
      if (cnt) {
        refill();
        const uint32_t hdr = (uint32_t)(win >> 58);  // 6 header bits (decoder.rs:138-144)
        const uint32_t ftype = hdr >> 4;
        if (ftype == 0) {
          const uint32_t E = (hdr & 15u) + 1u;
          win <<= 6;
          have -= 6;
          width = E;
          lit = E == 16u ? 1u : 0u;
          neg_thresh = 1u << (E - 1u);
          neg2 = lit ? 0u : (neg_thresh << 1);
          if (E <= 5u) {   // decoder.rs:209-216: the reference's reader decides (replay)
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
      for (uint32_t j = 0; j < maxcnt; ++j) {
        if (j && (j % X3_DEC_CHUNK) == 0) service();
        if (j < cnt) {
          refill();
          uint32_t top = (uint32_t)(win >> 32);
          uint32_t z = (uint32_t)__clz(top) & zmask;
          uint32_t zextra = 0;
          if (zmask && top == 0) {  // a zero run of >= 32 bits: keep counting, and let the reference's reader decide
            deferred = true;
            do {
              win <<= 32;
              have -= 32;
              zextra += 32;
              refill();
              top = (uint32_t)(win >> 32);
            } while (top == 0 && zextra < 128);
            z = top ? (uint32_t)__clz(top) : 0u;
          }
          win <<= z;
          have -= z;
          refill();
          const uint32_t v = (uint32_t)(win >> 32) >> rsh;
          win <<= width;
          have -= width;
          z += zextra;
          const uint32_t ii = v + level * z - level;                       // decoder.rs:186
          const int32_t d_rice = (int32_t)(ii >> 1) ^ -(int32_t)(ii & 1u);  // x3.rs:200-204
          const int32_t d_bfp = (int32_t)v - (int32_t)(v > neg_thresh ? neg2 : 0u);  // decoder.rs:198-207
          const int32_t d = zmask ? d_rice : d_bfp;
          const int32_t nl = lit ? (int32_t)v : last + d;
          if (zmask && ii >= bound) {  // OutOfBoundsInverse (decoder.rs:160,187)
            st = X3D_OUT_OF_BOUNDS_INVERSE;
            cnt = 0;
            remaining = 0;
          } else {
            last = (int16_t)(uint16_t)nl;
            put(c, i0 + j, last);
          }
        }
      }
      s_last[c * 64u + lane] = last;
    }
    if (remaining) remaining -= cnt;
    i0 += maxcnt;
  }
  if (f < n_frames) {
    if (active || st != X3D_OK) {
      const uint64_t taken = 32ull * rd_abs - have, held = 8ull * (v_end - (v_bits & ~15ull));
      if (active && (st == X3D_OUT_OF_BOUNDS_INVERSE || st == X3D_FRAME_DECODE_INVALID_BPF || deferred || taken > held))
        st = X3D_REPLAY;
      status[f] = st;
    } else {
      status[f] = X3D_OK;   // (refused by the check pass: its status counts)
    }
  }
}
