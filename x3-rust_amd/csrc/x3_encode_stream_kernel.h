// x3_encode_stream_kernel.h -- single-pass frame encoder for block_len = 20 (the default geometry).
//
// Same steps A-F as x3_encode_kernel.h, with three changes that matter on MI355X:
//
//  1. ONE pass over the samples.  Frame offsets in the stream come from a decoupled look-back over
//     per-frame descriptors instead of a size pass + scan: the grid is persistent (<= 3 workgroups
//     per CU, all co-resident, each looping over frames f = blockIdx.x + k*gridDim.x), a frame's
//     byte size is published as soon as its bit lengths are scanned, and after emission + CRC one
//     wave sums the predecessors' descriptors back to the nearest inclusive prefix.  A descriptor
//     is ONE 8-byte word {state:2 | value:62} written with an agent-scope relaxed atomic store and
//     polled with agent-scope relaxed atomic loads (write-through / L1-bypassing on gfx950; the
//     granule is its own flag, so no fence is needed -- cdna_hip_programming.md G16, form R2).
//     HBM traffic = 2 B/sample in + stream bytes out, nothing else.
//  2. The block lives in REGISTERS: 11 dwords (22 samples) per lane from LDS by ds_read_b64, the
//     20 first differences as ten v_pk_sub_i16 (saturating: |d| >= 16384 means a literal block
//     anyway, encoder.rs:308-311), min/max by v_pk_min/max_i16, zigzag and the Rice length sum in
//     packed 16-bit arithmetic.  No per-sample LDS traffic.
//  3. ONE emission loop for all three block families: per lane the packed source (zigzag / diff /
//     raw), an AND mask, an OR constant, a shift and a base length turn a sample into (code, len);
//     two samples share one flush test of the 64-bit accumulator.
//
// Blocks with fewer than 20 samples (the last block of a frame, tail frames) take a scalar path.
#pragma once
#include "x3_encode_kernel.h"

#define X3_DESC_AGG 1ull     // value = bytes of this frame
#define X3_DESC_PREFIX 2ull  // value = stream position behind this frame
#define X3_DESC_SHIFT 62
#define X3_DESC_MASK ((1ull << X3_DESC_SHIFT) - 1ull)
#define X3_SPIN_LIMIT (1u << 26)

typedef short x3_short2 __attribute__((ext_vector_type(2)));
typedef unsigned short x3_ushort2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t x3_pk_sub_sat(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_min_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_max_i16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_add_u16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_shl_b16(uint32_t a, uint32_t sh) {  // a << sh, per half
  uint32_t r;
  asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_shr_u16(uint32_t a, uint32_t sh) {  // logical
  uint32_t r;
  asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_pk_sar_i16(uint32_t a, uint32_t sh) {  // arithmetic
  uint32_t r;
  asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(sh * 0x10001u), "v"(a));
  return r;
}

__global__ void __launch_bounds__(512)
x3_encode_stream_kernel(const int16_t* __restrict__ wav, X3Geom g, X3DevParams p,
                        uint64_t* __restrict__ frame_off, uint8_t* __restrict__ out, uint64_t out_cap,
                        uint64_t start_pos, unsigned long long* __restrict__ desc,
                        unsigned long long* __restrict__ stats, int* __restrict__ status,
                        unsigned long long* __restrict__ end_pos, const uint16_t* __restrict__ xpow,
                        uint32_t lds_in_bytes, uint32_t img_dwords) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* part = reinterpret_cast<uint32_t*>(smem);  // [0..15] wave partials, [32..37] stats, [40] bad, [48..49] offset
  int16_t* in_s = reinterpret_cast<int16_t*>(smem + X3_ENC_SMEM_HDR);
  const uint32_t* in_w = reinterpret_cast<const uint32_t*>(smem + X3_ENC_SMEM_HDR);
  uint32_t* img = reinterpret_cast<uint32_t*>(smem + X3_ENC_SMEM_HDR + lds_in_bytes);

  const uint32_t tid = threadIdx.x, nthr = blockDim.x;  // 512
  const uint32_t lane = tid & 63u, wid = tid >> 6, nwaves = nthr >> 6;
  const uint64_t base_pos = (start_pos + 1ull) & ~1ull;  // writer.align::<2>() (encoder.rs:182)

  // rice parameters per ftype
  const uint32_t k0 = p.k[0], k1 = p.k[1], k2 = p.k[2];

  for (uint64_t f = blockIdx.x; f < g.n_frames; f += gridDim.x) {
    // ---- frame geometry
    const uint64_t clip = f / g.fpc;
    const uint64_t idx = f - clip * g.fpc;
    const uint64_t s_start = clip * g.clip_stride + idx * (uint64_t)p.spf;
    const uint64_t left = g.n_per_clip - idx * (uint64_t)p.spf;
    const uint32_t n = left < p.spf ? (uint32_t)left : p.spf;
    const int16_t* __restrict__ src = wav + s_start;

    // ---- A: stage samples (frames are 16-byte aligned on this path)
    {
      const uint32_t nvec = n >> 3;
      const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
      uint4* d4 = reinterpret_cast<uint4*>(in_s);
      for (uint32_t i = tid; i < nvec; i += nthr) d4[i] = s4[i];
      for (uint32_t i = nvec * 8 + tid; i < n; i += nthr) in_s[i] = src[i];
      if (tid < 2) in_s[n + tid] = 0;  // the dword behind the last sample is read (and ignored)
    }
    if (tid >= 32 && tid < 48) part[tid] = 0;
    __syncthreads();

    // ---- B: one block per lane, in registers
    const uint32_t nblocks = (n - 1 + 19) / 20;
    const uint32_t b = tid;
    const bool valid = b < nblocks;
    const uint32_t s0 = 1 + b * 20;
    const uint32_t cnt = valid ? (n - s0 < 20 ? n - s0 : 20) : 0;
    const bool full = cnt == 20;

    uint32_t W[11];   // samples 20b .. 20b+21 as (even, odd) pairs
    uint32_t S[10];   // emission source per pair of block samples (r = 2j+1, 2j+2)
    uint32_t type = 0, ft = 0, nb = 0, nbits = 0, bad = 0;
    uint32_t amask = 0, orc = 0, qsh = 0, qmask = 0, lbase = 0;  // (code,len) recipe, see file header
    if (full) {
      const uint2* r2 = reinterpret_cast<const uint2*>(in_w + 10 * b);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint2 v = r2[j];
        W[2 * j] = v.x;
        W[2 * j + 1] = v.y;
      }
      W[10] = in_w[10 * b + 10];
      uint32_t X[10], P[10];
      uint32_t mn = 0, mx = 0;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        X[j] = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // (s[2j+1], s[2j+2])
        P[j] = x3_pk_sub_sat(X[j], W[j]);                       // (d[2j+1], d[2j+2]), saturated
        mn = x3_pk_min_i16(mn, P[j]);
        mx = x3_pk_max_i16(mx, P[j]);
      }
      const int32_t dmin = min((int32_t)(int16_t)(mn & 0xFFFFu), (int32_t)mn >> 16);
      const int32_t dmax = max((int32_t)(int16_t)(mx & 0xFFFFu), (int32_t)mx >> 16);
      const int32_t maxabs = (-dmin) > dmax ? (-dmin) : dmax;
      if (maxabs <= (int32_t)p.thr[2]) {
        ft = (maxabs > (int32_t)p.thr[0] ? 1u : 0u) + (maxabs > (int32_t)p.thr[1] ? 1u : 0u);
        const uint32_t k = ft == 0 ? k0 : (ft == 1 ? k1 : k2);
        const int32_t lo = ft == 0 ? p.dmin[0] : (ft == 1 ? p.dmin[1] : p.dmin[2]);
        const int32_t hi = ft == 0 ? p.dmax[0] : (ft == 1 ? p.dmax[1] : p.dmax[2]);
        type = k;
        if (dmin < lo || dmax > hi) bad = 1;  // outside the reference's Rice table (panic there)
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          S[j] = x3_pk_shl_b16(P[j], 1) ^ x3_pk_sar_i16(P[j], 15);  // zigzag, per half
          sum = x3_pk_add_u16(sum, x3_pk_shr_u16(S[j], k));
        }
        nbits = bad ? 0u : 2u + 20u * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
        amask = (1u << k) - 1u;
        orc = 1u << k;
        qsh = k;
        qmask = 0xFFFFFFFFu;
        lbase = k + 1u;
      } else {
        nb = 32u - (uint32_t)__clz(maxabs);
        if (nb >= 15) {
          type = 5;
          nbits = 6 + 16 * 20;
#pragma unroll
          for (int j = 0; j < 10; ++j) S[j] = X[j];
          amask = 0xFFFFu;
          lbase = 16;
        } else {
          type = 4;
          nbits = 6 + 20 * (nb + 1);
#pragma unroll
          for (int j = 0; j < 10; ++j) S[j] = P[j];
          amask = (1u << (nb + 1)) - 1u;
          lbase = nb + 1;
        }
      }
    } else if (cnt) {
      // scalar path for a short block (x3_encode_kernel.h, step B)
      int32_t dmin = 0, dmax = 0, prev = in_s[s0 - 1];
      for (uint32_t i = 0; i < cnt; ++i) {
        const int32_t s = in_s[s0 + i], d = s - prev;
        prev = s;
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
      }
      const int32_t maxabs = (-dmin) > dmax ? (-dmin) : dmax;
      if (maxabs <= (int32_t)p.thr[2]) {
        ft = (maxabs > (int32_t)p.thr[0] ? 1u : 0u) + (maxabs > (int32_t)p.thr[1] ? 1u : 0u);
        const uint32_t k = ft == 0 ? k0 : (ft == 1 ? k1 : k2);
        const int32_t lo = ft == 0 ? p.dmin[0] : (ft == 1 ? p.dmin[1] : p.dmin[2]);
        const int32_t hi = ft == 0 ? p.dmax[0] : (ft == 1 ? p.dmax[1] : p.dmax[2]);
        type = k;
        if (dmin < lo || dmax > hi) {
          bad = 1;
        } else {
          uint32_t sum = 0;
          int32_t pv = in_s[s0 - 1];
          for (uint32_t i = 0; i < cnt; ++i) {
            const int32_t s = in_s[s0 + i], d = s - pv;
            pv = s;
            sum += (((uint32_t)d << 1) ^ (uint32_t)(d >> 31)) >> k;
          }
          nbits = 2 + cnt * (k + 1) + sum;
        }
      } else {
        nb = 32u - (uint32_t)__clz(maxabs);
        type = nb >= 15 ? 5u : 4u;
        nbits = nb >= 15 ? 6 + 16 * cnt : 6 + cnt * (nb + 1);
      }
    }

    // ---- C: workgroup exclusive scan of bit lengths; publish this frame's size
    const uint32_t incl = x3_wave_incl_scan(nbits, lane);
    if (lane == 63) part[wid] = incl;
    if (bad) part[40] = 1;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 8; ++w) {
      const uint32_t v = part[w];
      wave_base += (w < wid) ? v : 0u;
      total += v;
    }
    const uint32_t pos = 16u + wave_base + incl - nbits;
    const uint32_t total_bits = 16u + total;
    const uint32_t L = (((total_bits + 7u) >> 3) + 1u) & ~1u;  // word_align (bitpacker.rs:124-132)
    const uint32_t frame_bytes = 20u + L;
    const bool frame_bad = part[40] != 0;
    if (tid == 0) {
      const unsigned long long d = f == 0 ? ((X3_DESC_PREFIX << X3_DESC_SHIFT) | (base_pos + frame_bytes))
                                          : ((X3_DESC_AGG << X3_DESC_SHIFT) | (unsigned long long)frame_bytes);
      __hip_atomic_store(&desc[f], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // zero the image words this frame uses (header + payload, rounded up to 16 bytes)
    {
      uint4* z4 = reinterpret_cast<uint4*>(img);
      const uint32_t nz = (5u + ((L + 3u) >> 2) + 3u) >> 2;
      const uint4 zero = make_uint4(0, 0, 0, 0);
      for (uint32_t i = tid; i < nz; i += nthr) z4[i] = zero;
    }
    __syncthreads();
    if (tid == 0) atomicOr(&img[5], x3_bswap32(((uint32_t)(uint16_t)in_s[0]) << 16));  // <Audio State>

    // ---- D: emission
    if (nbits) {
      X3BitEmitter e;
      e.init(img + 5, pos);
      if (full) {
        e.put(type <= 3 ? ft + 1u : (type == 4 ? nb : 15u), type <= 3 ? 2u : 6u);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const uint32_t a = S[j] & 0xFFFFu, c = S[j] >> 16;
          const uint32_t la = ((a >> qsh) & qmask) + lbase, lc = ((c >> qsh) & qmask) + lbase;
          e.acc = (e.acc << la) | ((a & amask) | orc);
          e.acc = (e.acc << lc) | ((c & amask) | orc);
          e.cnt += la + lc;
          if (e.cnt >= 32u) {
            const uint32_t word = (uint32_t)(e.acc >> (e.cnt - 32u));
            atomicOr(&e.words[e.w], x3_bswap32(word));
            ++e.w;
            e.cnt -= 32u;
          }
        }
        e.finish();
      } else {
        if (type <= 3) {
          const uint32_t k = type;
          e.put(ft + 1, 2);
          const uint32_t mask = (1u << k) - 1u;
          int32_t pv = in_s[s0 - 1];
          for (uint32_t i = 0; i < cnt; ++i) {
            const int32_t s = in_s[s0 + i], d = s - pv;
            pv = s;
            const uint32_t u = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
            e.put((1u << k) | (u & mask), (u >> k) + 1u + k);
          }
        } else if (type == 4) {
          e.put(nb, 6);
          const uint32_t mask = (1u << (nb + 1)) - 1u;
          int32_t pv = in_s[s0 - 1];
          for (uint32_t i = 0; i < cnt; ++i) {
            const int32_t s = in_s[s0 + i], d = s - pv;
            pv = s;
            e.put((uint32_t)d & mask, nb + 1);
          }
        } else {
          e.put(15, 6);
          for (uint32_t i = 0; i < cnt; ++i) e.put((uint32_t)(uint16_t)in_s[s0 + i], 16);
        }
        e.finish();
      }
    }
    // statistics (encoder.rs:199): stats[type] += block.len()
#pragma unroll
    for (uint32_t t = 0; t < 6; ++t) {
      const unsigned long long m = __ballot(valid && type == t);
      if (lane == 0 && m) atomicAdd(&part[32 + t], (uint32_t)__popcll(m) * 20u);
    }
    if (valid && cnt != 20) atomicSub(&part[32 + type], 20u - cnt);
    __syncthreads();  // emission complete

    // ---- E: payload CRC-16 (segmented reduction) + frame header
    const uint32_t Lw = (L + 3u) >> 2;
    const uint32_t c_dw = (Lw + nthr - 1) / nthr;
    const int32_t j0 = (int32_t)(tid * c_dw) - (int32_t)(nthr * c_dw - Lw);
    uint32_t crc = 0;
    for (uint32_t i = 0; i < c_dw; ++i) {
      const int32_t j = j0 + (int32_t)i;
      if (j >= 0) {
        uint32_t be = x3_bswap32(img[5 + j]);
        if (j == 0) be ^= 0xFFFF0000u;
        crc = x3_crc_be32(crc, be);
      }
    }
#pragma unroll
    for (int lvl = 0; lvl < 6; ++lvl) {
      const uint32_t kx = x3_xp(xpow, lvl, c_dw);
      const uint32_t t = __shfl_up(crc, 1 << lvl, X3_WAVE);
      if (lane >= (1u << lvl)) crc = x3_gf_mul(t, kx) ^ crc;
    }
    if (lane == 63) part[wid] = crc;
    __syncthreads();
    if (wid == 0) {
      uint32_t v = lane < nwaves ? part[lane] : 0u;
      for (uint32_t lvl = 6, d = 1; d < nwaves; ++lvl, d <<= 1) {
        const uint32_t kx = x3_xp(xpow, lvl, c_dw);
        const uint32_t t = __shfl_up(v, d, X3_WAVE);
        if (lane >= d) v = x3_gf_mul(t, kx) ^ v;
      }
      if (lane == nwaves - 1) {
        if (L & 2u) v = x3_gf_mul(v, xpow[X3_XINV16_INDEX]);
        const uint32_t h0 = 0x78330101u;
        const uint32_t h1 = ((n & 0xFFFFu) << 16) | (L & 0xFFFFu);
        uint32_t hc = 0xFFFFu;
        hc = x3_crc_be32(hc, h0);
        hc = x3_crc_be32(hc, h1);
        hc = x3_crc_be32(hc, 0);
        hc = x3_crc_be32(hc, 0);
        img[0] = x3_bswap32(h0);
        img[1] = x3_bswap32(h1);
        img[2] = 0;
        img[3] = 0;
        img[4] = x3_bswap32((hc << 16) | (v & 0xFFFFu));
      }
    } else if (wid == 1 && f > 0) {
      // ---- F1: decoupled look-back (one wave): sum predecessors' sizes back to an inclusive prefix
      unsigned long long run = 0;
      uint64_t look = f;  // descriptors [look-64, look) are inspected next
      uint32_t spins = 0;
      bool done = false, timeout = false;
      while (!done) {
        const bool in_range = look > lane;
        const uint64_t gi = in_range ? look - 1 - lane : 0;
        unsigned long long d = in_range ? __hip_atomic_load(&desc[gi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : (X3_DESC_PREFIX << X3_DESC_SHIFT) | base_pos;  // virtual frame -1
        const uint32_t state = (uint32_t)(d >> X3_DESC_SHIFT);
        const unsigned long long m_inv = __ballot(state == 0);
        const unsigned long long m_pre = __ballot(state == (uint32_t)X3_DESC_PREFIX);
        // lanes nearer than the first PREFIX must all be valid
        const int first_pre = m_pre ? __ffsll((long long)m_pre) - 1 : 64;
        const unsigned long long need = first_pre >= 64 ? ~0ull : ((1ull << first_pre) - 1ull);
        if (m_inv & need) {
          if (++spins > X3_SPIN_LIMIT) { timeout = true; break; }
          __builtin_amdgcn_s_sleep(1);
          continue;  // poll again
        }
        unsigned long long v = ((int)lane <= first_pre) ? (d & X3_DESC_MASK) : 0ull;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, X3_WAVE);
        run += v;
        if (first_pre < 64) done = true;
        else look -= 64;
      }
      if (lane == 0) {
        if (timeout) {
          atomicMax(&status[1], X3D_BAD_ARG);
          run = 0;
        }
        part[48] = (uint32_t)run;
        part[49] = (uint32_t)(run >> 32);
        __hip_atomic_store(&desc[f], (X3_DESC_PREFIX << X3_DESC_SHIFT) | (run + frame_bytes), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (wid == 1 && lane == 0) {
      part[48] = (uint32_t)base_pos;
      part[49] = (uint32_t)(base_pos >> 32);
    }
    __syncthreads();

    // ---- F2: copy header + payload to the final stream position
    const uint64_t off = (uint64_t)part[48] | ((uint64_t)part[49] << 32);
    const uint32_t total_bytes = frame_bytes;
    const bool fits = off + total_bytes <= out_cap;
    if (fits && !frame_bad) {
      uint8_t* dst = out + off;
      const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
      if (mis == 0) {
        const uint32_t ndw = total_bytes >> 2;
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
        for (uint32_t i = tid; i < ndw; i += nthr) d32[i] = img[i];
        if ((total_bytes & 2u) && tid == 0) *reinterpret_cast<uint16_t*>(dst + 4 * ndw) = (uint16_t)img[ndw];
      } else if (mis == 2) {
        if (tid == 0) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)img[0];
        const uint32_t rem = total_bytes - 2u;
        const uint32_t ndw = rem >> 2;
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + 2);
        for (uint32_t i = tid; i < ndw; i += nthr) d32[i] = (img[i] >> 16) | (img[i + 1] << 16);
        if ((rem & 2u) && tid == 0) *reinterpret_cast<uint16_t*>(dst + 2 + 4 * ndw) = (uint16_t)(img[ndw] >> 16);
      } else {
        for (uint32_t i = tid; i < total_bytes; i += nthr) dst[i] = (uint8_t)(img[i >> 2] >> (8 * (i & 3u)));
      }
    }
    if (tid == 0) {
      frame_off[f] = off;
      if (!fits) atomicMax(&status[0], X3D_BYTE_WRITER_INSUFFICIENT_MEMORY);
      if (frame_bad) atomicMax(&status[0], X3D_BAD_ARG);
      if (f == g.n_frames - 1) {
        frame_off[g.n_frames] = off + total_bytes;
        *end_pos = off + total_bytes;
      }
      if (f == 0 && (start_pos & 1ull) && start_pos < out_cap) out[start_pos] = 0;  // align pad byte
    }
    if (tid < 6) {
      const uint32_t v = part[32 + tid];
      if (v) atomicAdd(&stats[tid], (unsigned long long)v);
    }
    __syncthreads();  // LDS is reused by the next frame
  }
}
