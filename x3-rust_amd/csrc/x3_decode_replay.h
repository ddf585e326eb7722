// x3_decode_replay.h -- reference-exact decode of ONE frame by ONE thread.
//
// The lane-per-frame decoders (x3_decode_split_kernel, x3_decode_fast_kernel, x3_decode_lanes_kernel) read
// the payload as an MSB-first bit string that is zero beyond its last byte and count zero runs exactly.
// The reference's BitReader (src/bitreader.rs:51-176) does the same on every stream its encoder can
// produce, but not on short or corrupt payloads:
//   * count_zero_bits (:128-139) extends a run by at most ONE peeked word: a run that covers a whole
//     32-bit word of the reader's grid is cut at the end of that word, and a word-aligned all-zero word
//     returns 32 without peeking at all;
//   * behind the last byte get_next (:148-163) leaves {leading_word 0, rem_bit 0}, and inc_bits (:76-92)
//     then sets rem_bit = 32 - rem: later zero runs come back as those phantom counts (< 32, often a
//     valid index of the inverse table), not as "no terminator";
//   * the word grid is relative to payload[2..], the last word may hold 1..3 bytes (read_word, :29-49).
// A frame whose header `samples` asks for more than the payload encodes therefore decodes "successfully"
// in the reference, with values that depend on that state machine.  To be reference-exact there without
// touching the hot loop, the fast decoders FLAG every frame in which they saw a decode error, a zero run
// of 32 bits or more, or a read position behind the payload's last byte (status X3D_REPLAY), and
// x3_decode_merge_kernel runs such a frame again through x3_replay_frame below: a scalar restatement of
// decoder::decode_frame (src/decoder.rs:36-58) over the reference's own reader.  Conforming streams never
// get here.
#pragma once
#include "x3_device.h"

#define X3D_REPLAY 101  // internal: the fast decoder defers this frame to x3_replay_frame

// BitReader (src/bitreader.rs:51-176)
struct X3RefReader {
  const uint8_t* a;  // payload[2..]
  uint32_t len, idx, word, rem;

  // read_word (:29-49): (word, bytes taken); a short last word is zero-padded (its third byte only counts
  // when exactly three remain, as in the reference)
  __device__ __forceinline__ uint32_t load(uint32_t at, uint32_t& took) const {
    const uint32_t left = len - at;
    if (left >= 4u) {
      took = 4u;
      return ((uint32_t)a[at] << 24) | ((uint32_t)a[at + 1] << 16) | ((uint32_t)a[at + 2] << 8) | (uint32_t)a[at + 3];
    }
    uint32_t w = 0;
    if (left >= 1u) w |= (uint32_t)a[at] << 24;
    if (left >= 2u) w |= (uint32_t)a[at + 1] << 16;
    if (left == 3u) w |= (uint32_t)a[at + 2] << 8;
    took = left;
    return w;
  }
  __device__ __forceinline__ void open(const uint8_t* bytes, uint32_t n) {  // BitReader::new (:65-74)
    a = bytes;
    len = n;
    uint32_t took;
    word = load(0u, took);
    idx = took;
    rem = took * 8u;
  }
  __device__ __forceinline__ void next() {  // get_next over peek_next (:148-175)
    if (idx >= len) {
      word = 0u;
      rem = 0u;
    } else {
      uint32_t took;
      word = load(idx, took);
      idx += took;
      rem = took * 8u;
    }
  }
  // inc_bits (:76-92); a shift count of 32 is taken modulo 32, as release-mode Rust does
  __device__ __forceinline__ void skip(uint32_t n) {
    if (n < rem) {
      word <<= (n & 31u);
      rem -= n;
    } else if (n > rem) {
      const uint32_t over = n - rem;
      next();
      rem = 32u - over;
      word <<= (over & 31u);
    } else {
      next();
    }
  }
  __device__ __forceinline__ uint32_t bits(uint32_t n) {  // read_nbits (:105-119)
    if (n <= rem) {
      const uint32_t r = word >> ((32u - n) & 31u);
      skip(n);
      return r;
    }
    const uint32_t over = n - rem;
    uint32_t r = word >> ((32u - n) & 31u);
    skip(rem);
    r |= word >> ((32u - over) & 31u);
    skip(over);
    return r;
  }
  __device__ __forceinline__ uint32_t zeros() {  // count_zero_bits (:128-139)
    uint32_t count = word ? (uint32_t)__clz(word) : 32u;
    if (count > rem) {
      if (idx >= len) {
        count = rem;
      } else {
        uint32_t took;
        const uint32_t w = load(idx, took);
        count = rem + (w ? (uint32_t)__clz(w) : 32u);
      }
    }
    skip(count);
    return count;
  }
};

// decoder::decode_block (src/decoder.rs:132-145) with the three block decoders (:147-235): n samples from the
// reader's position into out[0..n); `last` = *last_wav, kept modulo 2^16 (i16 arithmetic wraps in release).
// Returns X3D_OK, X3D_OUT_OF_BOUNDS_INVERSE or X3D_FRAME_DECODE_INVALID_BPF.
__device__ __forceinline__ int32_t x3_replay_block(X3RefReader& br, uint32_t n, const X3DevParams& p, uint32_t& last,
                                                   int16_t* __restrict__ out) {
  const uint32_t ftype = br.bits(2u);
  if (ftype == 0u) {  // decode_bpf_block (:209-235)
    const uint32_t E = br.bits(4u) + 1u;
    if (E <= 5u) return X3D_FRAME_DECODE_INVALID_BPF;
    if (E == 16u) {
      for (uint32_t i = 0; i < n; ++i) {
        last = br.bits(16u) & 0xFFFFu;
        out[i] = (int16_t)(uint16_t)last;
      }
    } else {
      const uint32_t half = 1u << (E - 1u);
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = br.bits(E) & 0xFFFFu;
        if (v > half) v -= half << 1;  // unsigned_to_i16 (:198-207): strict compare
        last = (last + v) & 0xFFFFu;
        out[i] = (int16_t)(uint16_t)last;
      }
    }
    if (n == 0u) return X3D_BAD_ARG;  // `*last_wav = wav[wav.len() - 1]` on an empty block panics (:232)
  } else if (ftype == 1u) {  // decode_ricecode_block_r1 (:147-170)
    const uint32_t bound = p.inv_len[0];
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t ix = br.zeros();
      (void)br.bits(1u);
      if (ix >= bound) return X3D_OUT_OF_BOUNDS_INVERSE;
      const uint32_t d = (ix & 1u) ? 0u - ((ix + 1u) >> 1) : (ix >> 1);  // INV_RICE_CODE (x3.rs:200-204)
      last = (last + d) & 0xFFFFu;
      out[i] = (int16_t)(uint16_t)last;
    }
  } else {  // decode_ricecode_block_r2r3 (:172-196): nb hard-wired, i16 arithmetic, `as usize` sign-extends
    const uint32_t nb = ftype == 2u ? 2u : 4u;
    const int32_t level = 1 << p.k[ftype - 1u];
    const uint32_t bound = p.inv_len[ftype - 1u];
    for (uint32_t i = 0; i < n; ++i) {
      const int32_t nz = (int32_t)(int16_t)br.zeros();
      const int32_t r = (int32_t)(int16_t)br.bits(nb);
      const int32_t ix = (int32_t)(int16_t)(r + level * (nz - 1));
      if (ix < 0 || (uint32_t)ix >= bound) return X3D_OUT_OF_BOUNDS_INVERSE;
      const uint32_t u = (uint32_t)ix;
      const uint32_t d = (u & 1u) ? 0u - ((u + 1u) >> 1) : (u >> 1);
      last = (last + d) & 0xFFFFu;
      out[i] = (int16_t)(uint16_t)last;
    }
  }
  return X3D_OK;
}

// decoder::decode_frame (src/decoder.rs:36-58): payload[0..plen), plen >= 2; samples >= 1; out has
// room for `samples` values.
__device__ __noinline__ int32_t x3_replay_frame(const uint8_t* __restrict__ payload, uint32_t plen, uint32_t samples,
                                                const X3DevParams& p, int16_t* __restrict__ out) {
  uint32_t last = ((uint32_t)payload[0] << 8) | payload[1];
  out[0] = (int16_t)(uint16_t)last;
  X3RefReader br;
  br.open(payload + 2, plen - 2u);
  uint32_t at = 1u, remaining = samples - 1u;
  // (block_len == 0 -- a damaged archive header can say so: empty blocks, `remaining` stays; Rice blocks read their type
  // bits and nothing else, a BFP block is an error or the reference's panic, and behind the payload the reader yields
  // zeros = a BFP block with E = 1, so the loop ends there at the latest; the turn limit only guards this restatement)
  uint32_t turns = 0;
  while (remaining) {
    const uint32_t n = remaining < p.block_len ? remaining : p.block_len;
    if (n == 0u && ++turns > 4u * plen + 64u) return X3D_BAD_ARG;
    const int32_t st = x3_replay_block(br, n, p, last, out + at);
    if (st != X3D_OK) return st;
    remaining -= n;
    at += n;
  }
  return X3D_OK;
}

// One frame through the reference's reader, for x3_decode_frame calls outside the fast decoders' geometry (a payload
// longer than the walk's 24 KB read buffer or the header's length field, more samples than its 16-bit count): what
// decoder::decode_frame itself does not limit.  One thread; status -> *status.
__global__ void x3_replay_one_kernel(const uint8_t* __restrict__ payload, uint32_t plen, uint32_t samples, X3DevParams p,
                                     int16_t* __restrict__ out, int32_t* __restrict__ status) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *status = x3_replay_frame(payload, plen, samples, p, out);
}

// ---- multi-channel extension (not in the reference: decoder.rs:90-94 refuses such frames; oracle/x3_oracle.c says what
// the extension's layout is).  Since round 4 the REPLAY path of x3_decode_mc_lanes_kernel (x3_decode_mc_kernel.h), which
// decodes a frame per lane; option "mc_decode_threads" = 1 still sends every frame here.
// One thread per frame over the reference's own reader: the first sample of every channel,
// then for every block index the block of channel 0 .. n_ch-1, each against its own channel's last sample.  Channel c
// goes to wav + c * ch_stride.  status[f]: X3D_OK or the block decoder's error; frames the check kernel has refused
// (cstatus[f] != 0) are skipped.
// (X3_MAX_CHANNELS: x3_tables.h)
__global__ void __launch_bounds__(64)
x3_decode_mc_kernel(const uint8_t* __restrict__ x3, const uint64_t* __restrict__ frame_off, const uint64_t* __restrict__ wav_off,
                    uint64_t n_frames, X3DevParams p, uint32_t n_ch, int16_t* __restrict__ wav, uint64_t ch_stride,
                    uint64_t wav_cap, const int32_t* __restrict__ cstatus, int32_t* __restrict__ status,
                    uint32_t replay_only) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  // replay_only (round 4): the frames x3_decode_mc_lanes_kernel has flagged (x3_decode_mc_kernel.h), nothing else
  if (replay_only && status[f] != X3D_REPLAY) return;
  if (cstatus[f] != X3D_OK) { status[f] = X3D_OK; return; }
  const uint8_t* __restrict__ h = x3 + frame_off[f];
  const uint32_t samples = ((uint32_t)h[4] << 8) | h[5], plen = ((uint32_t)h[6] << 8) | h[7];
  const uint8_t* __restrict__ payload = h + 20;
  const uint64_t wo = wav_off[f];
  if (samples == 0u || plen < 2u * n_ch || wo + samples > wav_cap) { status[f] = X3D_BAD_ARG; return; }
  uint32_t last[X3_MAX_CHANNELS];
  for (uint32_t c = 0; c < n_ch; ++c) {
    last[c] = ((uint32_t)payload[2u * c] << 8) | payload[2u * c + 1u];
    wav[(uint64_t)c * ch_stride + wo] = (int16_t)(uint16_t)last[c];
  }
  X3RefReader br;
  br.open(payload + 2u * n_ch, plen - 2u * n_ch);
  uint32_t at = 1u, remaining = samples - 1u, turns = 0;
  int32_t st = X3D_OK;
  while (remaining && st == X3D_OK) {
    const uint32_t n = remaining < p.block_len ? remaining : p.block_len;
    if (n == 0u && ++turns > 4u * plen + 64u) { st = X3D_BAD_ARG; break; }
    for (uint32_t c = 0; c < n_ch && st == X3D_OK; ++c) {
      uint32_t l = last[c];
      st = x3_replay_block(br, n, p, l, wav + (uint64_t)c * ch_stride + wo + at);
      last[c] = l;
    }
    remaining -= n;
    at += n;
  }
  status[f] = st;
}
