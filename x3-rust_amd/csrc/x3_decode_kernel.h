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
#include "x3_device.h"

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
#define X3_XINV8_INDEX(t) (X3_XINV16_INDEX + 1 + (t))

__global__ void __launch_bounds__(256)
x3_frame_check_kernel(const uint32_t* __restrict__ xw, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                      uint64_t n_frames, const uint16_t* __restrict__ xpow, int32_t* __restrict__ status,
                      X3FrameMeta* __restrict__ meta) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t f = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (f >= n_frames) return;  // whole wave
  const uint64_t n_dw = (x3_len + 3) >> 2;
  const uint64_t off = frame_off[f];
  int32_t st = X3D_OK;
  uint32_t plen = 0, samples = 0, pcrc = 0;

  // ---- header (decoder.rs:69-118); every lane reads the same 20 bytes (broadcast loads)
  if (off + 20 > x3_len) {
    st = X3D_STREAM_ENDS_IN_FRAME;
  } else {
    const uint32_t h0 = x3_be32_at(xw, n_dw, off), h1 = x3_be32_at(xw, n_dw, off + 4);
    const uint32_t h2 = x3_be32_at(xw, n_dw, off + 8), h3 = x3_be32_at(xw, n_dw, off + 12);
    const uint32_t h4 = x3_be32_at(xw, n_dw, off + 16);
    uint32_t hc = 0xFFFFu;
    hc = x3_crc_be32(hc, h0);
    hc = x3_crc_be32(hc, h1);
    hc = x3_crc_be32(hc, h2);
    hc = x3_crc_be32(hc, h3);
    samples = h1 >> 16;
    plen = h1 & 0xFFFFu;
    pcrc = h4 & 0xFFFFu;
    if ((h4 >> 16) != hc) st = X3D_FRAME_HEADER_INVALID_HEADER_CRC;
    else if ((h0 >> 16) != 0x7833u) st = X3D_FRAME_HEADER_INVALID_KEY;
    else if ((h0 & 0xFFu) > 1u) st = X3D_MORE_THAN_ONE_CHANNEL;
    else if (plen >= 0x7fe0u) st = X3D_FRAME_LENGTH;
    else if (off + 20 + plen > x3_len) st = X3D_STREAM_ENDS_IN_FRAME;          // decodefile.rs:114-116
    else if (plen > 24576u) st = X3D_FRAME_HEADER_INVALID_PAYLOAD_LEN;         // decodefile.rs:118-121
  }

  // ---- payload CRC (decodefile.rs:96-100)
  if (st == X3D_OK) {
    const uint64_t p0 = off + 20;
    uint32_t crc;
    if (plen < 4) {
      crc = 0xFFFFu;
      for (uint32_t i = 0; i < plen; ++i) crc = x3_crc_byte(crc, x3_be32_at(xw, n_dw, p0 + i) >> 24);
    } else {
      const uint32_t lead = (uint32_t)(p0 & 3u);                 // header bytes in front, inside dword 0
      const uint64_t a_dw = p0 >> 2;
      const uint32_t nd = (lead + plen + 3u) >> 2;               // aligned dwords covering the payload
      const uint32_t tpad = 4u * nd - lead - plen;               // bytes behind the payload in the last dword
      const uint32_t c_dw = (nd + 63u) >> 6;
      const int32_t j0 = (int32_t)(lane * c_dw) - (int32_t)(64u * c_dw - nd);
      crc = 0;
      for (uint32_t i = 0; i < c_dw; ++i) {
        const int32_t j = j0 + (int32_t)i;
        if (j >= 0) {
          uint32_t be = x3_bswap32(xw[a_dw + (uint32_t)j]);
          if (j == 0) {
            be &= 0xFFFFFFFFu >> (8u * lead);
            be ^= lead <= 2 ? (0xFFFF0000u >> (8u * lead)) : 0x000000FFu;   // CRC init folded into bytes 0,1
          }
          if (j == 1 && lead == 3) be ^= 0xFF000000u;
          if ((uint32_t)j == nd - 1) be &= 0xFFFFFFFFu << (8u * tpad);
          crc = x3_crc_be32(crc, be);
        }
      }
#pragma unroll
      for (int lvl = 0; lvl < 6; ++lvl) {
        const uint32_t kx = x3_xp(xpow, lvl, c_dw);
        const uint32_t t = __shfl_up(crc, 1 << lvl, X3_WAVE);
        if (lane >= (1u << lvl)) crc = x3_gf_mul(t, kx) ^ crc;
      }
      crc = __shfl(crc, 63, X3_WAVE);
      if (tpad) crc = x3_gf_mul(crc, xpow[X3_XINV8_INDEX(tpad)]);  // undo the virtual trailing zero bytes
    }
    if (crc != pcrc) st = X3D_FRAME_HEADER_INVALID_PAYLOAD_CRC;
  }
  if (lane == 0) {
    status[f] = st;
    meta[f].payload_len = plen;
    meta[f].samples = samples;
  }
}

// inverse Rice map (x3.rs:200-204): 0,-1,1,-2,2,...
__device__ __forceinline__ int32_t x3_inv_rice(uint32_t i) { return (i & 1u) ? -(int32_t)((i + 1u) >> 1) : (int32_t)(i >> 1); }

#define X3_DEC_RING_DW 32u       // per-lane input ring: 32 dwords = 128 bytes
#define X3_DEC_RING_STRIDE 36u   // row stride in dwords (144 B: 16-byte aligned rows, spread over banks)
#define X3_DEC_WIN 160u          // samples staged per lane between flushes (320 bytes)
#define X3_DEC_OUT_STRIDE 82u    // row stride in dwords (80 + 2: 8-byte aligned rows)
#define X3_DEC_CHUNK 20u         // samples decoded between two services of the input ring

__global__ void __launch_bounds__(64)
x3_decode_lanes_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                       uint64_t n_frames, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                       int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                       const X3FrameMeta* __restrict__ meta) {
  __shared__ __attribute__((aligned(16))) uint32_t ring[64 * X3_DEC_RING_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t outs[64 * X3_DEC_OUT_STRIDE];
  __shared__ unsigned long long s_base[64];  // output address of each lane's frame (0 = not cooperative)
  __shared__ uint32_t s_ns[64];              // samples of each lane's frame

  const uint32_t lane = threadIdx.x;
  const uint64_t f = (uint64_t)blockIdx.x * 64 + lane;
  uint32_t* const row = ring + lane * X3_DEC_RING_STRIDE;
  uint32_t* const orow = outs + lane * X3_DEC_OUT_STRIDE;

  // ---- per-lane frame setup
  bool active = f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 0;
  uint64_t p0 = 0;
  int16_t* o = nullptr;
  if (active) {
    st = status[f];
    samples = meta[f].samples;
    plen = meta[f].payload_len;
    p0 = frame_off[f] + 20;
    if (st != X3D_OK) {
      active = false;
    } else if (samples == 0 || plen < 2) {
      st = X3D_BAD_ARG;  // the reference panics (decoder.rs:42,47)
      active = false;
    } else {
      uint64_t wo;
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
      } else {
        o = wav + wo;
      }
    }
  }
  // frames whose output is 16-byte aligned are flushed cooperatively, the others store directly
  const bool coop = active && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0);
  s_base[lane] = coop ? (unsigned long long)reinterpret_cast<uintptr_t>(o) : 0ull;
  s_ns[lane] = active ? samples : 0u;

  // ---- input ring: absolute addresses, 16-byte chunks
  const uintptr_t pay_addr = reinterpret_cast<uintptr_t>(x3) + p0;
  const uintptr_t end_addr = pay_addr + plen;          // first byte that must read as zero
  const uintptr_t bits_addr = pay_addr + 2;            // the bit stream starts behind the raw first sample
  uintptr_t next_chunk = bits_addr & ~(uintptr_t)15;   // next 16-byte chunk to request
  uint32_t wr_abs = 0;                                 // dwords written to the ring so far
  uint32_t rd_abs = 0;                                 // dwords taken out of the ring so far

  auto load_chunk = [&](uintptr_t a) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (a < end_addr) {
      v = *reinterpret_cast<const uint4*>(a);
      if (a + 16 > end_addr) {  // zero the bytes past the payload
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uintptr_t ad = a + 4 * d;
          if (ad >= end_addr) w[d] = 0;
          else if (ad + 4 > end_addr) w[d] &= (1u << (8u * (uint32_t)(end_addr - ad))) - 1u;
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return v;
  };
  auto park = [&](const uint4& v) {
    *reinterpret_cast<uint4*>(row + (wr_abs & (X3_DEC_RING_DW - 1u))) = v;
    wr_abs += 4;
  };

  uint64_t win = 0;
  uint32_t have = 0, nextw = 0;
  int32_t last = 0;
  uint32_t remaining = 0;  // samples of this lane's frame still to decode
  if (active) {
    for (int k = 0; k < 8; ++k) {  // fill the ring: 128 bytes
      park(load_chunk(next_chunk));
      next_chunk += 16;
    }
    const uint32_t skip = (uint32_t)(bits_addr & 15u);  // bytes of the first chunk in front of the bit stream
    rd_abs = skip >> 2;
    const uint32_t a = skip & 3u;
    const uint32_t w0 = x3_bswap32(row[rd_abs & (X3_DEC_RING_DW - 1u)]);
    ++rd_abs;
    win = (uint64_t)w0 << (32u + 8u * a);
    have = 32u - 8u * a;
    nextw = x3_bswap32(row[rd_abs & (X3_DEC_RING_DW - 1u)]);
    last = (int16_t)(uint16_t)((uint32_t)x3[p0] << 8 | x3[p0 + 1]);  // <Audio State> (decoder.rs:42)
  }
  // in-flight chunk requests (issued one service ahead of use)
  uint4 ld0 = make_uint4(0, 0, 0, 0), ld1 = ld0, ld2 = ld0;
  uint32_t n_inflight = 0;

  auto refill = [&]() {
    if (have <= 32u) {
      win |= (uint64_t)nextw << (32u - have);
      have += 32u;
      ++rd_abs;
      nextw = x3_bswap32(row[rd_abs & (X3_DEC_RING_DW - 1u)]);
    }
  };
  auto read_bits = [&](uint32_t n) -> uint32_t {  // n in 1..32
    refill();
    const uint32_t v = (uint32_t)(win >> (64u - n));
    win <<= n;
    have -= n;
    return v;
  };
  auto count_zeros = [&]() -> uint32_t {  // consumes the zero run, not the terminating 1
    refill();
    uint32_t top = (uint32_t)(win >> 32);
    if (top) {
      const uint32_t z = (uint32_t)__clz(top);
      win <<= z;
      have -= z;
      return z;
    }
    uint32_t total = 0;
    for (;;) {
      win <<= 32;
      have -= 32;
      total += 32;
      if (total >= 128) return total;  // far beyond every table bound
      refill();
      top = (uint32_t)(win >> 32);
      if (top) {
        const uint32_t z = (uint32_t)__clz(top);
        win <<= z;
        have -= z;
        return total + z;
      }
    }
  };
  // keep the ring ahead of the reader: park what was requested one service ago, request up to 3
  // more chunks (48 B >= the 41 B a 20-sample chunk can consume), top up synchronously if short
  auto service = [&]() {
    if (n_inflight > 0) park(ld0);
    if (n_inflight > 1) park(ld1);
    if (n_inflight > 2) park(ld2);
    n_inflight = 0;
    if (remaining > 0) {
      while (wr_abs - rd_abs < 14u) {  // never in steady state; guards ring underflow
        park(load_chunk(next_chunk));
        next_chunk += 16;
      }
      const uint32_t free_dw = X3_DEC_RING_DW - (wr_abs - rd_abs) - 1u;  // keep nextw's slot
      const uint32_t want = free_dw >> 2;
      n_inflight = want > 3u ? 3u : want;
      if (n_inflight > 0) { ld0 = load_chunk(next_chunk); next_chunk += 16; }
      if (n_inflight > 1) { ld1 = load_chunk(next_chunk); next_chunk += 16; }
      if (n_inflight > 2) { ld2 = load_chunk(next_chunk); next_chunk += 16; }
    }
  };

  // ---- output staging
  uint32_t wbase = 0;   // first sample index of the staged window (uniform, multiple of X3_DEC_WIN)
  int32_t carry = 0;    // even-indexed sample waiting for its odd partner
  auto flush = [&](uint32_t upto) {  // stage holds samples [wbase, upto) of every lane's frame
    __syncthreads();
    const uint32_t pieces = (upto - wbase + 7u) >> 3;  // 16-byte pieces per frame in this window
    const uint32_t total = pieces * 64u;
    for (uint32_t t = lane; t < total; t += 64u) {
      const uint32_t r = t / pieces, q = t - r * pieces;
      const unsigned long long base = s_base[r];
      const uint32_t ns = s_ns[r];
      if (base && wbase + 8u * q + 8u <= ns) {
        const uint2* src = reinterpret_cast<const uint2*>(outs + r * X3_DEC_OUT_STRIDE + 4u * q);
        const uint2 lo = src[0], hi = src[1];
        *reinterpret_cast<uint4*>(base + 2ull * wbase + 16ull * q) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    __syncthreads();
  };

  if (active) {
    carry = last;  // sample 0
    if (!coop || samples == 1u) o[0] = (int16_t)last;
  }

  // ---- lock-step decode: block index and in-block sample index are wave-uniform
  remaining = active ? samples - 1u : 0u;
  const uint32_t bl = p.block_len;
  uint32_t i = 1;  // uniform index of the sample being produced
  for (;;) {
    if (!__any(remaining > 0)) break;
    uint32_t cnt = remaining < bl ? remaining : bl;
    // block header (decoder.rs:138-144)
    uint32_t mode_rice = 0, E = 0, nb = 0, bound = 0;
    int32_t level = 0;
    service();
    if (cnt) {
      const uint32_t ftype = read_bits(2);
      if (ftype == 0) {
        E = read_bits(4) + 1u;  // decode_bpf_block (decoder.rs:209-235)
        if (E <= 5u) {
          st = X3D_FRAME_DECODE_INVALID_BPF;
          cnt = 0;
          remaining = 0;
        }
      } else {
        mode_rice = 1;
        // r1: count zeros, skip the 1 (decoder.rs:156-163)  ==  nb = 1, level = 1 below
        // r2r3: nb hard-wired 2 / 4, level = 1 << nsubs (decoder.rs:180-186)
        nb = ftype == 1u ? 1u : (ftype == 2u ? 2u : 4u);
        level = ftype == 1u ? 1 : (1 << (ftype == 2u ? p.k[1] : p.k[2]));
        bound = ftype == 1u ? p.inv_len[0] : (ftype == 2u ? p.inv_len[1] : p.inv_len[2]);
      }
    }
    const uint32_t neg_thresh = E ? (1u << (E - 1u)) : 0u;
    for (uint32_t j = 0; j < bl; ++j) {
      if (!__any(j < cnt)) break;
      if (j && (j % X3_DEC_CHUNK) == 0) service();
      if (j < cnt) {
        if (mode_rice) {
          const int32_t nz = (int32_t)count_zeros();
          const int32_t r = (int32_t)read_bits(nb);
          const int32_t ii = (int32_t)(int16_t)(r + level * (nz - 1));
          if (ii < 0 || (uint32_t)ii >= bound) {
            st = X3D_OUT_OF_BOUNDS_INVERSE;
            cnt = 0;
            remaining = 0;
          } else {
            last = (int16_t)(uint16_t)(last + x3_inv_rice((uint32_t)ii));
          }
        } else {
          int32_t a = (int32_t)read_bits(E);
          if (E == 16u) {
            last = (int16_t)(uint16_t)a;  // literal block (decoder.rs:218-222)
          } else {
            if ((uint32_t)a > neg_thresh) a -= (int32_t)(neg_thresh << 1);  // unsigned_to_i16 (decoder.rs:198-207)
            last = (int16_t)(uint16_t)(last + a);
          }
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
        flush(i);
        wbase = i;
      }
    }
    if (remaining) {
      remaining -= cnt;
      if (remaining == 0 && coop && st == X3D_OK) {
        // this lane's frame is complete: the cooperative flush stores only 16-byte pieces that lie
        // fully inside the frame, so the owner stores the < 8 samples behind the last full piece
        if (samples & 1u) orow[(samples - 1u - wbase) >> 1] = (uint32_t)carry & 0xFFFFu;
        const uint32_t done = samples & ~7u;
        const uint32_t from = done > wbase ? done : wbase;
        const uint16_t* h = reinterpret_cast<const uint16_t*>(orow);
        for (uint32_t s = from; s < samples; ++s) o[s] = (int16_t)h[s - wbase];
      }
    }
  }
  if (i > wbase) flush(i);  // the partial last window
  if (f < n_frames) status[f] = st;
}

// first frame with a non-zero status, and the samples of the good frames before it
struct X3DecodeSummary {
  unsigned long long first_bad;
  unsigned long long samples_before;
  int first_bad_status;
  int pad;
};

__global__ void __launch_bounds__(1024)
x3_decode_summary_kernel(const int32_t* __restrict__ status, const X3FrameMeta* __restrict__ meta,
                         uint64_t n_frames, X3DecodeSummary* __restrict__ out) {
  __shared__ unsigned long long s_first;
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) { s_first = n_frames; s_sum = 0; }
  __syncthreads();
  unsigned long long mine = n_frames;
  for (uint64_t f = threadIdx.x; f < n_frames; f += blockDim.x)
    if (status[f] != 0) { mine = f; break; }
  if (mine < n_frames) atomicMin(&s_first, mine);
  __syncthreads();
  const unsigned long long first = s_first;
  unsigned long long sum = 0;
  for (uint64_t f = threadIdx.x; f < first; f += blockDim.x) sum += meta[f].samples;
  if (sum) atomicAdd(&s_sum, sum);
  __syncthreads();
  if (threadIdx.x == 0) {
    out->first_bad = first;
    out->samples_before = s_sum;
    out->first_bad_status = first < n_frames ? status[first] : 0;
    out->pad = 0;
  }
}
