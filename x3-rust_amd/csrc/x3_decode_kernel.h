// x3_decode_kernel.h -- frame decoder for gfx950: one frame per lane.
//
// Decoding a frame is sequential (variable-length codes; a block's start is known only once
// the previous block is decoded), frames are independent (each re-seeds the predictor with a
// raw sample, decoder.rs:42-46).  So the parallel axis is the frame: lane f validates frame f's
// header (decoder::read_frame_header, decoder.rs:69-118), checks the payload CRC
// (X3aReader::read_frame_payload, decodefile.rs:93-103) and runs decode_frame / decode_block
// (decoder.rs:36-58,132-235) with a per-lane MSB-first bit window in registers that replaces
// BitReader (bitreader.rs:29-176).
//
// Bit-window semantics: the payload is read as an infinite MSB-first bit string that is zero
// beyond the payload's last byte (bitreader.rs:34-48,157-161).  A zero run is counted exactly;
// the reference caps a run at the end of the NEXT 32-bit word (bitreader.rs:129-139), which only
// differs for runs >= 32 bits -- an OutOfBoundsInverse error in both for the default parameters.
#pragma once
#include "x3_device.h"

#define X3D_STREAM_ENDS_IN_FRAME (-1)  // quiet stop of the walk (decodefile.rs:107-116)

struct X3BitWindow {
  const uint8_t* q;      // next 4-byte-aligned address to fetch
  const uint8_t* lo;     // first valid byte
  const uint8_t* hi;     // one past the last valid byte (payload end)
  uint64_t win;          // next bits, MSB first
  uint32_t have;         // valid bits in win

  __device__ __forceinline__ uint32_t fetch() {
    uint32_t w = 0;
    if (q < hi) {
      w = x3_bswap32(*reinterpret_cast<const uint32_t*>(q));
      if (q + 4 > hi) w &= 0xFFFFFFFFu << (8u * (uint32_t)(q + 4 - hi));  // zero past the payload
    }
    q += 4;
    return w;
  }
  // `start` must be inside a 4-byte-aligned buffer that covers [start & ~3, roundup4(end))
  __device__ __forceinline__ void init(const uint8_t* start, const uint8_t* end) {
    lo = start;
    hi = end;
    const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(start) & 3u);
    q = start - a;
    uint32_t w = fetch();
    win = (uint64_t)w << (32 + 8 * a);
    have = 32 - 8 * a;
    refill();
  }
  __device__ __forceinline__ void refill() {
    if (have <= 32) {
      win |= (uint64_t)fetch() << (32 - have);
      have += 32;
    }
  }
  // n in 1..32; at most 32 bits may be consumed between refills
  __device__ __forceinline__ uint32_t read(uint32_t n) {
    refill();
    uint32_t v = (uint32_t)(win >> (64 - n));
    win <<= n;
    have -= n;
    return v;
  }
  // count and consume leading zero bits (not the terminating 1)
  __device__ __forceinline__ uint32_t zeros() {
    uint32_t total = 0;
    for (;;) {
      refill();
      uint32_t top = (uint32_t)(win >> 32);
      if (top) {
        uint32_t z = (uint32_t)__clz(top);
        win <<= z;
        have -= z;
        return total + z;
      }
      // 32 zero bits at least
      win <<= 32;
      have -= 32;
      total += 32;
      if (total >= 128) return total;  // far beyond every table bound
    }
  }
};

// inverse Rice map (x3.rs:200-204): 0,-1,1,-2,2,...
__device__ __forceinline__ int32_t x3_inv_rice(uint32_t i) { return (i & 1u) ? -(int32_t)((i + 1u) >> 1) : (int32_t)(i >> 1); }

__global__ void __launch_bounds__(256)
x3_decode_frames_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len,
                        const uint64_t* __restrict__ frame_off, uint64_t n_frames, X3Geom g,
                        const uint64_t* __restrict__ wav_off, X3DevParams p, int16_t* __restrict__ wav,
                        uint64_t wav_cap, int32_t* __restrict__ status, uint32_t* __restrict__ nsamp) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  int32_t st = X3D_OK;
  uint32_t samples = 0;

  do {
    const uint64_t off = frame_off[f];
    // ---- header (decoder.rs:69-118)
    if (off + 20 > x3_len) { st = X3D_STREAM_ENDS_IN_FRAME; break; }
    const uint16_t* h16 = reinterpret_cast<const uint16_t*>(x3 + off);  // frames start on even offsets
    uint32_t hb[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      uint32_t a = h16[2 * i], b = h16[2 * i + 1];  // little-endian loads of stream bytes
      hb[i] = x3_bswap32(a | (b << 16));            // -> big-endian numeric value of 4 stream bytes
    }
    uint32_t hc = 0xFFFFu;
    hc = x3_crc_be32(hc, hb[0]);
    hc = x3_crc_be32(hc, hb[1]);
    hc = x3_crc_be32(hc, hb[2]);
    hc = x3_crc_be32(hc, hb[3]);
    if ((hb[4] >> 16) != hc) { st = X3D_FRAME_HEADER_INVALID_HEADER_CRC; break; }
    if ((hb[0] >> 16) != 0x7833u) { st = X3D_FRAME_HEADER_INVALID_KEY; break; }
    if ((hb[0] & 0xFFu) > 1u) { st = X3D_MORE_THAN_ONE_CHANNEL; break; }
    samples = hb[1] >> 16;
    const uint32_t plen = hb[1] & 0xFFFFu;
    if (plen >= 0x7fe0u) { st = X3D_FRAME_LENGTH; break; }
    const uint32_t pcrc = hb[4] & 0xFFFFu;
    // ---- walk checks (decodefile.rs:114-121)
    if (off + 20 + plen > x3_len) { st = X3D_STREAM_ENDS_IN_FRAME; break; }
    if (plen > 24576u) { st = X3D_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; }
    const uint8_t* pay = x3 + off + 20;
    // ---- payload CRC (decodefile.rs:96-100)
    {
      const uint16_t* p16 = reinterpret_cast<const uint16_t*>(pay);
      uint32_t crc = 0xFFFFu;
      const uint32_t n2 = plen >> 1;
      for (uint32_t i = 0; i < n2; ++i) {
        uint32_t v = p16[i];
        crc = x3_crc_byte(crc, v & 0xFFu);
        crc = x3_crc_byte(crc, v >> 8);
      }
      if (plen & 1u) crc = x3_crc_byte(crc, pay[plen - 1]);
      if (crc != pcrc) { st = X3D_FRAME_HEADER_INVALID_PAYLOAD_CRC; break; }
    }
    // ---- decode_frame (decoder.rs:36-58)
    if (samples == 0 || plen < 2 || p.block_len == 0) { st = X3D_BAD_ARG; break; }
    uint64_t wo;
    if (wav_off) {
      wo = wav_off[f];
    } else {
      const uint64_t clip = f / g.fpc;
      const uint64_t idx = f - clip * g.fpc;
      wo = clip * g.clip_stride + idx * (uint64_t)p.spf;
    }
    if (wo + samples > wav_cap) { st = X3D_BAD_ARG; break; }
    int16_t* __restrict__ o = wav + wo;

    int32_t last = (int16_t)(uint16_t)((uint32_t)pay[0] << 8 | pay[1]);
    o[0] = (int16_t)last;
    X3BitWindow br;
    br.init(pay + 2, pay + plen);
    uint32_t remaining = samples - 1;
    uint32_t pw = 1;
    while (remaining > 0 && st == X3D_OK) {
      const uint32_t bl = remaining < p.block_len ? remaining : p.block_len;
      const uint32_t ftype = br.read(2);
      if (ftype == 0) {
        // decode_bpf_block (decoder.rs:209-235)
        const uint32_t E = br.read(4) + 1u;
        if (E <= 5u) { st = X3D_FRAME_DECODE_INVALID_BPF; break; }
        if (E == 16u) {
          for (uint32_t i = 0; i < bl; ++i) {
            last = (int16_t)(uint16_t)br.read(16);
            o[pw + i] = (int16_t)last;
          }
        } else {
          const int32_t neg_thresh = 1 << (E - 1), neg = 1 << E;
          for (uint32_t i = 0; i < bl; ++i) {
            int32_t a = (int32_t)br.read(E);
            if (a > neg_thresh) a -= neg;  // unsigned_to_i16 (decoder.rs:198-207)
            last = (int16_t)(uint16_t)(last + a);
            o[pw + i] = (int16_t)last;
          }
        }
      } else if (ftype == 1) {
        // decode_ricecode_block_r1 (decoder.rs:147-170)
        const uint32_t bound = p.inv_len[0];
        for (uint32_t i = 0; i < bl; ++i) {
          const uint32_t z = br.zeros();
          br.read(1);
          if (z >= bound) { st = X3D_OUT_OF_BOUNDS_INVERSE; break; }
          last = (int16_t)(uint16_t)(last + x3_inv_rice(z));
          o[pw + i] = (int16_t)last;
        }
      } else {
        // decode_ricecode_block_r2r3 (decoder.rs:172-196): nb hard-wired 2 / 4
        const uint32_t ft = ftype - 1u;
        const uint32_t nb = ftype == 2u ? 2u : 4u;
        const int32_t level = 1 << (ftype == 2u ? p.k[1] : p.k[2]);
        const uint32_t bound = ftype == 2u ? p.inv_len[1] : p.inv_len[2];
        (void)ft;
        for (uint32_t i = 0; i < bl; ++i) {
          const int32_t nz = (int32_t)br.zeros();
          const int32_t r = (int32_t)br.read(nb);
          const int32_t ii = (int32_t)(int16_t)(r + level * (nz - 1));
          if (ii < 0 || (uint32_t)ii >= bound) { st = X3D_OUT_OF_BOUNDS_INVERSE; break; }
          last = (int16_t)(uint16_t)(last + x3_inv_rice((uint32_t)ii));
          o[pw + i] = (int16_t)last;
        }
      }
      remaining -= bl;
      pw += bl;
    }
  } while (0);

  status[f] = st;
  nsamp[f] = samples;
}

// first frame with a non-zero status, and the samples of the good frames before it
struct X3DecodeSummary {
  unsigned long long first_bad;
  unsigned long long samples_before;
  int first_bad_status;
  int pad;
};

__global__ void __launch_bounds__(1024)
x3_decode_summary_kernel(const int32_t* __restrict__ status, const uint32_t* __restrict__ nsamp,
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
  for (uint64_t f = threadIdx.x; f < first; f += blockDim.x) sum += nsamp[f];
  if (sum) atomicAdd(&s_sum, sum);
  __syncthreads();
  if (threadIdx.x == 0) {
    out->first_bad = first;
    out->samples_before = s_sum;
    out->first_bad_status = first < n_frames ? status[first] : 0;
    out->pad = 0;
  }
}
