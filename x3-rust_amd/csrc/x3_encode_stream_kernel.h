// x3_encode_stream_kernel.h -- single-pass frame encoder for block_len = 20 (the default geometry).
//
// Same steps A-F as x3_encode_kernel.h, with three changes that matter on MI355X:
//
//  1. ONE pass over the samples.  The grid is persistent (2 workgroups per CU, all co-resident, each looping
//     over frames f = blockIdx.x + k*G) and frame offsets in the stream come from the frames' sizes, which
//     every workgroup publishes as soon as the bit lengths are scanned -- one 4-byte word
//     {epoch:12 | bytes:20}, agent-scope relaxed atomic store, polled with agent-scope relaxed atomic loads
//     (write-through / L1-bypassing on gfx950; the word is its own flag, so no fence is needed --
//     cdna_hip_programming.md G16, form R2).  A workgroup's consecutive frames are f-G and f, hence
//         off(f) = off(f-G) + bytes(f-G) + SUM bytes(j), j in (f-G, f):
//     G-1 words of the OTHER workgroups, no prefix chain.  The words are requested while frame f+G is
//     analysed and emitted, and frame f is copied out one frame late from the second of two frame images.
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

#define X3_SPIN_LIMIT (1u << 16)  // polls of >= 1 memory round trip each (~0.1 s): a bounded spin, never a hang
#define X3D_SIZE_WAIT_TIMEOUT 100  // internal: the host re-runs the two-pass encoder (x3_encode_result)

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

// SDWA operand selects (gfx9 encoding, available on gfx950): halves of a dword as operands of a 32-bit op
__device__ __forceinline__ uint32_t x3_sdwa_add_w0_w1(uint32_t a) {  // a.lo16 + a.hi16
  uint32_t r;
  asm("v_add_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ uint32_t x3_sdwa_shl_w0_by_w1(uint32_t v, uint32_t sh) {  // v.lo16 << sh.hi16
  uint32_t r;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"
      : "=v"(r) : "v"(sh), "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t x3_sdwa_or_w1(uint32_t a, uint32_t b) {  // a | b.hi16
  uint32_t r;
  asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
      : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// ds_or_b32 on an LDS byte address (no return value)
__device__ __forceinline__ void x3_lds_or_b32(uint32_t addr, uint32_t v) {
  __hip_atomic_fetch_or(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)addr), v,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#define X3_STREAM_THREADS 576u  // 8 compute waves (one block per lane) + 1 helper wave

// One 16-byte-per-lane direct-to-LDS load, hidden from hipcc (cdna_hip_programming.md section 5.7): the
// compiler would otherwise wait vmcnt(0) before every later LDS access of the wave.  lds_dst is the
// wave-uniform LDS byte address; lane l lands at lds_dst + 16*l.  The caller waits (x3_dma_wait)
// before the barrier that precedes the first read of the staged bytes.
__device__ __forceinline__ void x3_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// vmcnt(0) as the BUILTIN, not as asm text: hipcc's wait-count pass sees it and clears its scoreboard, so it
// does not add conservative vmcnt(0) waits of its own later (those would also wait for the hidden LDS-DMA).
__device__ __forceinline__ void x3_dma_wait() {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  asm volatile("" ::: "memory");
}

// a(x) * C(x) mod 0x11021 for a compile-time constant C: the sixteen C*x^b are immediates
constexpr uint32_t x3_gf_xtime(uint32_t k) { return ((k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u)) & 0xFFFFu; }
constexpr uint32_t x3_gf_mul_c(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 0; i < 16; ++i) {
    if ((a >> i) & 1u) r ^= b;
    b = x3_gf_xtime(b);
  }
  return r;
}
constexpr uint32_t x3_gf_pow_c(uint32_t base, int e) {
  uint32_t r = 1;
  for (int i = 0; i < e; ++i) r = x3_gf_mul_c(r, base);
  return r;
}
constexpr uint32_t x3_crc16_step_c(uint32_t crc, uint32_t byte) {
  uint32_t t = ((crc >> 8) ^ byte) & 0xFFu;
  t ^= t >> 4;
  return ((crc << 8) ^ (t << 12) ^ (t << 5) ^ t) & 0xFFFFu;
}
constexpr uint32_t x3_crc16_const4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
  return x3_crc16_step_c(x3_crc16_step_c(x3_crc16_step_c(x3_crc16_step_c(0xFFFFu, b0), b1), b2), b3);
}
// x^-1 = x^15 + x^11 + x^4 (x * that = x^16 + x^12 + x^5 = P + 1); x^-16 = (x^-1)^16
constexpr uint32_t X3_XINV16_C = x3_gf_pow_c(0x8810u, 16);
template <uint32_t C>
__device__ __forceinline__ uint32_t x3_gf_mul_const(uint32_t a) {
  uint32_t r = 0, k = C;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    r ^= (0u - ((a >> b) & 1u)) & k;
    k = x3_gf_xtime(k);
  }
  return r;
}

// Four 16-byte-per-lane direct-to-LDS loads (4 KB) behind ONE M0 write: the instruction offset advances the
// global address and the LDS address together (LDS address = M0 + inst_offset + 16*lane).
__device__ __forceinline__ void x3_glds16x4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// stage frame samples [0, n) of `src` into LDS with direct-to-LDS loads (no VGPRs, all in flight at
// once); executed by ONE wave.  4 KB groups, then 1 KB pieces, then the < 8 samples behind the last
// full 16-byte piece by lanes.
__device__ __forceinline__ void x3_stage_frame_dma(const int16_t* __restrict__ src, uint32_t n, int16_t* in_s,
                                                   uint32_t lane) {
  const uint32_t npieces = n >> 3;  // 16-byte pieces
  const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(x3_lds_addr(in_s));
  uint32_t base = 0;
  for (; base + 256u <= npieces; base += 256u)
    x3_glds16x4(s4 + base + lane, __builtin_amdgcn_readfirstlane(lds0 + 16u * base));
  for (; base < npieces; base += 64u) {
    if (base + lane < npieces) x3_glds16(s4 + base + lane, __builtin_amdgcn_readfirstlane(lds0 + 16u * base));
  }
  const uint32_t done = npieces << 3;
  if (done + lane < n) in_s[done + lane] = src[done + lane];
  if (lane < 2) in_s[n + lane] = 0;  // the dword behind the last sample is read (and ignored)
}

#ifndef X3_STREAM_MIN_WAVES
#define X3_STREAM_MIN_WAVES 6  // <= 80 VGPRs: two 9-wave workgroups per CU
#endif
#define X3_DESC_BYTES_BITS 20u                       // frame bytes <= 20 + 65535 < 2^20
#define X3_DESC_BYTES_MASK ((1u << X3_DESC_BYTES_BITS) - 1u)
#define X3_LB_WINDOWS 8                              // 8 x 64 descriptors requested at once
#define X3_DESC_PAD 576u                             // words in front of desc[0]: windows may reach below frame 0

__global__ void __launch_bounds__(X3_STREAM_THREADS, X3_STREAM_MIN_WAVES)
x3_encode_stream_kernel(const int16_t* __restrict__ wav, X3Geom g, X3DevParams p,
                        uint64_t* __restrict__ frame_off, uint8_t* __restrict__ out, uint64_t out_cap,
                        uint64_t start_pos, uint32_t* __restrict__ desc, uint32_t epoch,
                        unsigned long long* __restrict__ stats, int* __restrict__ status,
                        unsigned long long* __restrict__ end_pos, const uint32_t* __restrict__ xk16,
                        const uint16_t* __restrict__ crc_tab_g, uint32_t lds_in_bytes, uint32_t img_dwords) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // part: [0..7] scan partials, [16..23] CRC partials, [32..37] stats, [40] bad, [41] offsets lost, [48..49] stream offset of
  // the frame being copied out, [50] header CRC
  uint32_t* part = reinterpret_cast<uint32_t*>(smem);
  int16_t* in_s = reinterpret_cast<int16_t*>(smem + X3_ENC_SMEM_HDR);
  const uint32_t* in_w = reinterpret_cast<const uint32_t*>(smem + X3_ENC_SMEM_HDR);
  // TWO frame images: frame k is copied out while frame k+1 is analysed and emitted (see file header)
  uint32_t* img0 = reinterpret_cast<uint32_t*>(smem + X3_ENC_SMEM_HDR + lds_in_bytes);
  // slicing-by-4 CRC tables behind the images: T[j][v] = crc0 of byte v followed by j zero bytes
  uint16_t* crc_tab = reinterpret_cast<uint16_t*>(img0 + 2u * img_dwords);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u, wid = tid >> 6;
  const bool helper_wave = wid == 8;         // wave 8: frame sizes and offsets, prefetch of the next frame
  const uint32_t nthr = 512;                 // compute threads
  const uint64_t base_pos = (start_pos + 1ull) & ~1ull;  // writer.align::<2>() (encoder.rs:182)
  const uint32_t k0 = p.k[0], k1 = p.k[1], k2 = p.k[2];
  const uint32_t G = gridDim.x;
  const uint32_t ready_tag = epoch << X3_DESC_BYTES_BITS;

  // frame f = (clip, idx): each role walks its frames f = blockIdx.x + k*G without dividing per frame
  auto geom_at = [&](uint64_t clip, uint32_t idx, const int16_t*& src, uint32_t& n) __attribute__((always_inline)) {
    const uint64_t left = g.n_per_clip - (uint64_t)idx * (uint64_t)p.spf;
    n = left < p.spf ? (uint32_t)left : p.spf;
    src = wav + clip * g.clip_stride + (uint64_t)idx * (uint64_t)p.spf;
  };
  auto geom_advance = [&](uint64_t& clip, uint32_t& idx) __attribute__((always_inline)) {
    if (G < g.fpc) {
      idx += G;
      if (idx >= g.fpc) { idx -= g.fpc; ++clip; }
    } else {  // clips shorter than the grid is wide
      const uint64_t t = (uint64_t)idx + G;
      clip += t / g.fpc;
      idx = (uint32_t)(t % g.fpc);
    }
  };
  uint64_t clip_f = blockIdx.x / g.fpc;
  uint32_t idx_f = (uint32_t)(blockIdx.x - clip_f * g.fpc);

#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  // ---- prologue: the first frame is staged by the helper wave
  if (tid < 64) part[tid] = 0;
  for (uint32_t i = tid; i < (2u * img_dwords) >> 2; i += X3_STREAM_THREADS)
    reinterpret_cast<uint4*>(img0)[i] = make_uint4(0, 0, 0, 0);  // both frame images start clear
  if (tid < 512) reinterpret_cast<uint32_t*>(crc_tab)[tid] = reinterpret_cast<const uint32_t*>(crc_tab_g)[tid];
  if (helper_wave) {
    const int16_t* src;
    uint32_t n;
    geom_at(clip_f, idx_f, src, n);
    x3_stage_frame_dma(src, n, in_s, lane);
    x3_dma_wait();
  }
  __syncthreads();

  // The two roles run their own frame loops (the branch is outside the loops so that loop-carried and
  // hoisted values of one role are not live in the other) and meet at four barriers per frame (B1, B3, B4, B4b).
  if (helper_wave) {
    // =============================================================== helper wave
    // Few instructions, all of them on the critical path of the eight compute waves (which wait for this
    // wave at every barrier): issue them ahead of the compute waves sharing the SIMD.  Measured without it:
    // ~25 cycles per helper instruction while the compute waves emit.
    __builtin_amdgcn_s_setprio(3);
    // Offsets: this workgroup's consecutive frames are f-G and f, so
    //   off(f) = off(f-G) + bytes(f-G) + SUM bytes(j), j in (f-G, f)      (first frame: base_pos + SUM j < f)
    // -- at most G-1 sizes of OTHER workgroups, each published as one 4-byte word {epoch:12 | bytes:20} with an
    // agent-scope relaxed atomic store as soon as the frame's bit lengths are scanned (the word is its own flag:
    // cdna_hip_programming.md G16, form R2).  No prefix chain.  The loads for frame f are REQUESTED when f's
    // emission is done and CONSUMED one frame later (a loaded-latency of several microseconds under the
    // kernel's own streaming traffic, measured, hides behind the next frame's analysis and emission).
    uint32_t dq[X3_LB_WINDOWS];
    uint32_t hdr_crc = 0, cur_img = 0;
    uint64_t pend_f = 0, my_off = 0;
    uint32_t pend_bytes = 0, my_bytes = 0;
    bool pending = false, first = true, lost = false;
    // Window w of a frame = the sizes of frames f-1-lane-64w.  Requests are raw loads at constant offsets
    // from one per-lane pointer (the array has X3_DESC_PAD words in front, so windows reaching below frame
    // 0 read padding); range and readiness are checked when a word is USED, so that requesting never waits.
    const uint32_t lane1 = lane + 1u;
    auto request = [&]() __attribute__((always_inline)) {
      const uint32_t* p0 = desc + pend_f - lane1;
#pragma unroll
      for (uint32_t w = 0; w < X3_LB_WINDOWS; ++w)
        dq[w] = __hip_atomic_load(p0 - 64 * (int)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto resolve = [&]() __attribute__((always_inline)) {
      // finish the pending frame's offset: all of its predecessors' sizes must have arrived
      const uint32_t needed = first ? (uint32_t)pend_f : G - 1u;  // sizes in front of pend_f that count
      const uint32_t nwin = (needed + 63u) >> 6;
      const uint32_t* p0 = desc + pend_f - lane1;
      uint32_t sum = 0, spins = 0;
      bool timeout = false;
#pragma unroll
      for (uint32_t w = 0; w < X3_LB_WINDOWS; ++w) {
        if (w < nwin) {
          const bool in = (int32_t)lane1 <= (int32_t)needed - 64 * (int32_t)w;
          uint32_t v = in ? dq[w] : ready_tag;  // "ready, 0 bytes" outside the range
#ifdef X3_DBG_STAMPS
          {
            const unsigned long long miss = __ballot((v >> X3_DESC_BYTES_BITS) != epoch);
            if (miss) dbg_acc[7] += (1ull << 32) + ((unsigned long long)__popcll(miss) << 48);
          }
#endif
          while (__any((v >> X3_DESC_BYTES_BITS) != epoch)) {
#ifdef X3_DBG_STAMPS
            dbg_acc[7] += 1;
#endif
            // give up after the bounded spin -- or as soon as ANY workgroup has (then the launch is lost anyway and
            // every further wait would only add its own 0.1 s): the host re-encodes with the two-pass kernels
            if (++spins > X3_SPIN_LIMIT ||
                __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT) {
              timeout = true;
              break;
            }
            __builtin_amdgcn_s_sleep(8);
            const uint32_t r = __hip_atomic_load(p0 - 64 * (int)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = in ? r : ready_tag;
          }
          sum += v & X3_DESC_BYTES_MASK;
        }
      }
      unsigned long long tot = 0;
      if (nwin > X3_LB_WINDOWS) {  // grids wider than 512 workgroups: the rest synchronously, summed in 64 bits
        unsigned long long acc = 0;
        for (uint32_t w = X3_LB_WINDOWS; w < nwin && !timeout; ++w) {
          const bool in = lane1 + 64u * w <= needed;
          uint32_t v;
          for (;;) {
            const uint32_t r = __hip_atomic_load(p0 - 64 * (int64_t)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = in ? r : ready_tag;
            if (!__any((v >> X3_DESC_BYTES_BITS) != epoch)) break;
            if (++spins > X3_SPIN_LIMIT ||
                __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT) {
              timeout = true;
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
          acc += v & X3_DESC_BYTES_MASK;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, X3_WAVE);
        tot = acc;
      }
      X3_STAMP(1);  // windows examined
      // 8 windows x 64 lanes x 2^20 < 2^32: a 32-bit DPP scan, total in lane 63
      tot += (unsigned long long)__builtin_amdgcn_readlane(x3_wave_incl_scan_dpp(sum), 63);
      const uint64_t off = (first ? base_pos : my_off + my_bytes) + tot;
      if (timeout && !lost) {
        // This workgroup no longer knows where its frames go -- this one and, since each offset builds on the
        // last, every later one.  Nothing of them may reach the output, the frame index or the end position:
        // x3_encode_result re-encodes the whole call with the two-pass kernels, and those rewrite
        // d_out[start_pos..) only -- bytes in front of start_pos belong to the caller.
        lost = true;
        if (lane == 0) {
          atomicMax(&status[1], X3D_SIZE_WAIT_TIMEOUT);
          part[41] = 1;  // read by copy_out
        }
      }
      my_off = off;
      my_bytes = pend_bytes;
      first = false;
      X3_STAMP(0);  // reduced
      if (lane == 0 && !lost) {
        part[48] = (uint32_t)off;
        part[49] = (uint32_t)(off >> 32);
        frame_off[pend_f] = off;
        if (off + pend_bytes > out_cap) atomicMax(&status[0], X3D_BYTE_WRITER_INSUFFICIENT_MEMORY);
        if (part[40] != 0) atomicMax(&status[0], X3D_BAD_ARG);
        if (pend_f == g.n_frames - 1) {
          frame_off[g.n_frames] = off + pend_bytes;
          *end_pos = off + pend_bytes;
        }
        if (pend_f == 0 && (start_pos & 1ull) && start_pos < out_cap) out[start_pos] = 0;  // align pad byte
      }
      pending = false;
    };

    for (uint64_t f = blockIdx.x; f < g.n_frames; f += G) {
      // nothing of this wave is in flight here except last frame's bookkeeping stores: tell hipcc so, or its
      // wait-count pass protects registers of the polling loads (maybe pending on the loop's back edge) with
      // vmcnt(0) waits in the middle of the prefetch below -- which then waits for the prefetch itself
      x3_dma_wait();
      const int16_t* src;
      uint32_t n;
      geom_at(clip_f, idx_f, src, n);
      (void)src;
      geom_advance(clip_f, idx_f);  // now the geometry of f + G
      X3_STAMP(0);
      __syncthreads();  // B1: bit-length partials ready; every block is in registers, in_s is free
      X3_STAMP(2);
      uint32_t total = 0;
#pragma unroll
      for (uint32_t w = 0; w < 8; ++w) total += part[w];
      const uint32_t L = (((16u + total + 7u) >> 3) + 1u) & ~1u;  // word_align (bitpacker.rs:124-132)
      const uint32_t frame_bytes = 20u + L;
      // publish this frame's size as early as possible
      if (lane == 0)
        __hip_atomic_store(&desc[f], ready_tag | frame_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // prefetch the next frame of this workgroup straight into LDS; the requests are hidden from hipcc
      // and waited for before B3
      const uint64_t fn = f + G;
      if (fn < g.n_frames) {
        const int16_t* nsrc;
        uint32_t nn;
        geom_at(clip_f, idx_f, nsrc, nn);
        x3_stage_frame_dma(nsrc, nn, in_s, lane);
      }
      // request the sizes in front of the PREVIOUS frame of this workgroup (published a whole frame time
      // ago unless a workgroup lags by more than that).  Memory operations retire in order, so these words
      // arrive behind the prefetch -- which has to be complete before B3 anyway.
      if (pending) request();
      X3_STAMP(3);
      // header CRC (encoder.rs:153-154): it needs only the sample count and the payload length.  The state
      // behind the constant bytes "x3", id, id is a constant; the (samples, payload_len) word and the
      // eight zero time bytes go through the slicing tables (a zero word is two look-ups).
      if (lane == 0) {
        constexpr uint32_t K4 = x3_crc16_const4(0x78u, 0x33u, 0x01u, 0x01u);
        const uint32_t m = (((n & 0xFFFFu) << 16) | (L & 0xFFFFu)) ^ (K4 << 16);
        uint32_t hc = (uint32_t)crc_tab[768u + (m >> 24)] ^ (uint32_t)crc_tab[512u + ((m >> 16) & 0xFFu)] ^
                      (uint32_t)crc_tab[256u + ((m >> 8) & 0xFFu)] ^ (uint32_t)crc_tab[m & 0xFFu];
        hc = (uint32_t)crc_tab[768u + (hc >> 8)] ^ (uint32_t)crc_tab[512u + (hc & 0xFFu)];
        hc = (uint32_t)crc_tab[768u + (hc >> 8)] ^ (uint32_t)crc_tab[512u + (hc & 0xFFu)];
        hdr_crc = hc;
      }
      X3_STAMP(4);
      __syncthreads();  // B3: emission complete (a barrier of the compute waves; nothing of this wave's is due yet)
      X3_STAMP(2);
      // The prefetch and the size words are a memory round trip (several microseconds under the kernel's own
      // traffic) behind B1.  They are due at B4, not at B3: the compute waves' CRC pass lies in between, and
      // behind B4 they take the previous frame out (its offset) and analyse the next one (its samples).
      x3_dma_wait();    // the next frame's samples have landed in LDS (and the requested sizes behind them)
      X3_STAMP(6);
      if (pending) resolve();  // the previous frame's offset: consumed by the compute waves behind B4
      X3_STAMP(5);
      __syncthreads();  // B4: CRC partials ready, next frame's samples landed, previous frame's offset known
      X3_STAMP(2);
      // this frame's header (encoder.rs:122-162): "x3", id, id, samples, payload_len, 8 zero time bytes, header
      // crc over bytes 0..16, payload crc; audio frames use id 1.  Here, not in a compute wave: that one would
      // keep the other seven waiting at the next barrier for its hundred scalar-like instructions.
      if (lane == 0) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; ++w) v ^= part[16 + w];
        if (L & 2u) v = x3_gf_mul_const<X3_XINV16_C>(v);  // undo the 2 virtual pad-to-4 bytes
        uint32_t* img = img0 + cur_img * img_dwords;
        img[0] = x3_bswap32(0x78330101u);
        img[1] = x3_bswap32(((n & 0xFFFFu) << 16) | (L & 0xFFFFu));
        img[2] = 0;
        img[3] = 0;
        img[4] = x3_bswap32((hdr_crc << 16) | (v & 0xFFFFu));
      }
      cur_img ^= 1u;
      pend_f = f;
      pend_bytes = frame_bytes;
      pending = true;
      __syncthreads();  // B4b: the previous frame has been copied out (its image is cleared behind this)
      X3_STAMP(2);
    }
    if (pending) {
      request();
      resolve();
    }
    __syncthreads();  // B5: the last frame's offset
  } else {
    // =============================================================== compute waves
    uint32_t prev_bytes = 0;
    bool have_prev = false;
    uint32_t cur = 0;
    auto copy_out = [&](const uint32_t* img, uint32_t total_bytes) __attribute__((always_inline)) {
      // header + payload of a finished frame to its final stream position (part[48..49])
      const uint64_t off = (uint64_t)part[48] | ((uint64_t)part[49] << 32);
      if (off + total_bytes <= out_cap && part[40] == 0 && part[41] == 0) {  // [41]: offsets lost to a size-wait time-out
        uint8_t* dst = out + off;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
        // sixteen bytes per lane and trip (a default frame is ~330 of them: one trip, two thirds of the lanes);
        // the stream position is even, so the image is either dword-aligned to it or two bytes off
        if (mis == 0) {
          const uint32_t ndw = total_bytes >> 2, nq = ndw >> 2;
          uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
          const uint4* img4 = reinterpret_cast<const uint4*>(img);
          for (uint32_t i = tid; i < nq; i += nthr) *reinterpret_cast<uint4*>(d32 + 4u * i) = img4[i];
          if (tid < (ndw & 3u)) d32[4u * nq + tid] = img[4u * nq + tid];
          if ((total_bytes & 2u) && tid == 0) *reinterpret_cast<uint16_t*>(dst + 4 * ndw) = (uint16_t)img[ndw];
        } else if (mis == 2) {
          if (tid == 0) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)img[0];
          const uint32_t rem = total_bytes - 2u;
          const uint32_t ndw = rem >> 2, nq = ndw >> 2;
          uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + 2);
          const uint4* img4 = reinterpret_cast<const uint4*>(img);
          for (uint32_t i = tid; i < nq; i += nthr) {
            const uint4 a = img4[i];
            const uint32_t e = img[4u * i + 4u];
            *reinterpret_cast<uint4*>(d32 + 4u * i) =
                make_uint4(__builtin_amdgcn_alignbit(a.y, a.x, 16), __builtin_amdgcn_alignbit(a.z, a.y, 16),
                           __builtin_amdgcn_alignbit(a.w, a.z, 16), __builtin_amdgcn_alignbit(e, a.w, 16));
          }
          if (tid < (ndw & 3u)) d32[4u * nq + tid] = (img[4u * nq + tid] >> 16) | (img[4u * nq + tid + 1u] << 16);
          if ((rem & 2u) && tid == 0) *reinterpret_cast<uint16_t*>(dst + 2 + 4 * ndw) = (uint16_t)(img[ndw] >> 16);
        } else {
          for (uint32_t i = tid; i < total_bytes; i += nthr) dst[i] = (uint8_t)(img[i >> 2] >> (8 * (i & 3u)));
        }
      }
    };

    for (uint64_t f = blockIdx.x; f < g.n_frames; f += G) {
      const int16_t* src;
      uint32_t n;
      geom_at(clip_f, idx_f, src, n);
      (void)src;
      geom_advance(clip_f, idx_f);
      uint32_t* img = img0 + cur * img_dwords;

      // ---- B: one block per lane, in registers (blocks of 19 or 20 samples; shorter ones: scalar path)
      const uint32_t nblocks = (n - 1 + 19) / 20;
      const uint32_t b = tid;
      const bool valid = b < nblocks;
      const uint32_t s0 = 1 + b * 20;
      const uint32_t cnt = valid ? (n - s0 < 20 ? n - s0 : 20) : 0;
      const bool regs = cnt >= 19;
      const uint32_t s_first = (uint32_t)(uint16_t)in_s[0];

      uint32_t S[10];   // emission source per pair of block samples (r = 2j+1, 2j+2)
      uint32_t type = 0, ft = 0, nb = 0, nbits = 0, bad = 0;
      uint32_t amask = 0, orc = 0, qsh = 0, qmask = 0, lbase = 0;  // (code,len) recipe, see file header
      if (regs) {
        uint32_t W[11];   // samples 20b .. 20b+21 as (even, odd) pairs
        const uint2* r2 = reinterpret_cast<const uint2*>(in_w + 10 * b);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const uint2 v = r2[j];
          W[2 * j] = v.x;
          W[2 * j + 1] = v.y;
        }
        W[10] = in_w[10 * b + 10];
        uint32_t mn = 0, mx = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const uint32_t Xj = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // (s[2j+1], s[2j+2])
          S[j] = x3_pk_sub_sat(Xj, W[j]);                                      // (d[2j+1], d[2j+2]), saturated
          if (j == 9 && cnt == 19) S[9] &= 0xFFFFu;                            // sample 20 does not exist
          mn = x3_pk_min_i16(mn, S[j]);
          mx = x3_pk_max_i16(mx, S[j]);
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
            S[j] = x3_pk_shl_b16(S[j], 1) ^ x3_pk_sar_i16(S[j], 15);  // zigzag, per half
            sum = x3_pk_add_u16(sum, x3_pk_shr_u16(S[j], k));
          }
          nbits = bad ? 0u : 2u + cnt * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
          amask = (1u << k) - 1u;
          orc = 1u << k;
          qsh = k;
          qmask = 0xFFFFFFFFu;
          lbase = k + 1u;
        } else {
          nb = 32u - (uint32_t)__clz(maxabs);
          if (nb >= 15) {
            type = 5;
            nbits = 6 + 16 * cnt;
#pragma unroll
            for (int j = 0; j < 10; ++j) S[j] = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // raw samples
            amask = 0xFFFFu;
            lbase = 16;
          } else {
            type = 4;
            nbits = 6 + cnt * (nb + 1);
            amask = (1u << (nb + 1)) - 1u;  // S already holds the exact diffs
            lbase = nb + 1;
          }
        }
      } else if (cnt) {
        // scalar path for a short block (tail frames only): sizes now, emission below from a private copy
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
        // keep the (< 19) samples of a short block in the S registers: in_s is overwritten by the prefetch
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const uint32_t i0 = 2u * j, i1 = 2u * j + 1u;
          const uint32_t a = i0 <= cnt ? (uint32_t)(uint16_t)in_s[s0 - 1 + i0] : 0u;   // sample r = i0 (r = 0: predecessor)
          const uint32_t c = i1 <= cnt ? (uint32_t)(uint16_t)in_s[s0 - 1 + i1] : 0u;
          S[j] = a | (c << 16);
        }
      }


      X3_STAMP(0);
      // ---- C: workgroup exclusive scan of bit lengths
      const uint32_t incl = x3_wave_incl_scan_dpp(nbits);
      if (lane == 63) part[wid] = incl;
      if (bad) part[40] = 1;
      X3_STAMP(1);
      __syncthreads();  // B1: partials ready; every compute lane holds its block in registers, in_s is free
      X3_STAMP(2);
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

      if (tid == 0) atomicOr(&img[5], x3_bswap32(s_first << 16));  // <Audio State> (encoder.rs:189)
      // ---- D: emission
      if (nbits) {
        X3BitEmitter e;
        e.init(img + 5, pos);
        if (regs) {
          e.put(type <= 3 ? ft + 1u : (type == 4 ? nb : 15u), type <= 3 ? 2u : 6u);
          // (code, len) for BOTH samples of a pair in packed 16-bit arithmetic, the halves combined with
          // SDWA operand selects: 11 VALU per pair in front of the flush test
          const uint32_t qsh2 = qsh * 0x10001u, lbase2 = lbase * 0x10001u;
          const uint32_t amask2 = amask * 0x10001u, orc2 = orc * 0x10001u;
          const uint32_t last_on = cnt == 20 ? 0xFFFFFFFFu : 0x0000FFFFu;  // a 19-sample block has no sample 20
          uint32_t waddr = x3_lds_addr(e.words + e.w);  // byte address of the word the accumulator flushes to
          uint64_t acc = e.acc;
          uint32_t pend = e.cnt;
#pragma unroll
          for (int j = 0; j < 10; ++j) {
            uint32_t Lp = x3_pk_add_u16(x3_pk_lshr_b16(S[j], qsh2) & qmask, lbase2);  // (la, lc)
            uint32_t Cp = (S[j] & amask2) | orc2;                                        // (ca, cc)
            if (j == 9) { Lp &= last_on; Cp &= last_on; }
            const uint32_t tot = x3_sdwa_add_w0_w1(Lp);                                  // la + lc <= 32
            const uint32_t pair = x3_sdwa_or_w1(x3_sdwa_shl_w0_by_w1(Cp, Lp), Cp);       // (ca << lc) | cc
            acc = (acc << tot) | (unsigned long long)pair;
            pend += tot;
            if (pend >= 32u) {
              pend -= 32u;
              x3_lds_or_b32(waddr, x3_bswap32((uint32_t)(acc >> pend)));
              waddr += 4u;
            }
          }
          if (pend) x3_lds_or_b32(waddr, x3_bswap32((uint32_t)(acc << (32u - pend))));
        } else {
          // short block: samples r = 0..cnt are in S as raw 16-bit values (r = 0 is the predecessor)
          auto smp = [&](uint32_t r) -> int32_t {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) w = (r >> 1) == (uint32_t)j ? S[j] : w;
            return (int32_t)(int16_t)((r & 1u) ? (w >> 16) : (w & 0xFFFFu));
          };
          if (type <= 3) {
            const uint32_t k = type;
            e.put(ft + 1, 2);
            const uint32_t mask = (1u << k) - 1u;
            for (uint32_t i = 1; i <= cnt; ++i) {
              const int32_t d = smp(i) - smp(i - 1);
              const uint32_t u = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
              e.put((1u << k) | (u & mask), (u >> k) + 1u + k);
            }
          } else if (type == 4) {
            e.put(nb, 6);
            const uint32_t mask = (1u << (nb + 1)) - 1u;
            for (uint32_t i = 1; i <= cnt; ++i) e.put((uint32_t)(smp(i) - smp(i - 1)) & mask, nb + 1);
          } else {
            e.put(15, 6);
            for (uint32_t i = 1; i <= cnt; ++i) e.put((uint32_t)(uint16_t)smp(i), 16);
          }
          e.finish();
        }
      }
      // statistics (encoder.rs:199): stats[type] += block.len(); one LDS atomic per lane, summed over
      // all frames of this workgroup and flushed once at the end
      if (cnt) atomicAdd(&part[32 + type], cnt);

      X3_STAMP(4);
      __syncthreads();  // B3: emission complete
#ifdef X3_DBG_ALLWAVES
      X3_STAMP(3);
#else
      X3_STAMP(2);
#endif

      X3_STAMP(6);

      uint32_t crc = 0;
      // ---- E: payload CRC-16 as a segmented reduction:
      // crc0(payload) = XOR over lanes t of crc0(chunk_t) * x^(32*c_dw*(511-t)) mod P; each lane
      // multiplies by ITS OWN power of x (table xk) and the products are XOR-reduced.
      const uint32_t Lw = (L + 3u) >> 2;
      const uint32_t c_dw = (Lw + nthr - 1) / nthr;  // 1..11 on this path
      const int32_t j0 = (int32_t)(tid * c_dw) - (int32_t)(nthr * c_dw - Lw);
      // this lane's multiplier x^(32*c_dw*(511-tid)), as its sixteen shifts K*x^b (requested now, used below)
      const uint4* kp = reinterpret_cast<const uint4*>(xk16 + ((size_t)(c_dw - 1u) * 512u + tid) * 16u);
      const uint4 kq0 = kp[0], kq1 = kp[1], kq2 = kp[2], kq3 = kp[3];
      for (uint32_t i = 0; i < c_dw; ++i) {
        const int32_t j = j0 + (int32_t)i;
        if (j >= 0) {
          uint32_t be = x3_bswap32(img[5 + j]);
          // CRC init 0xFFFF folded into the first 16 message bits (dword 0 only); as arithmetic, not a select:
          // a v_cndmask on a stale VCC is a SIMD-wide bottleneck on gfx950 (tools/ubench/issue_cost.hip)
          be ^= x3_mask_if_zero(j) & 0xFFFF0000u;
          // slicing-by-4: fold the running CRC into the top 16 message bits, one table per byte
          const uint32_t m = be ^ (crc << 16);
          crc = (uint32_t)crc_tab[768u + (m >> 24)] ^ (uint32_t)crc_tab[512u + ((m >> 16) & 0xFFu)] ^
                (uint32_t)crc_tab[256u + ((m >> 8) & 0xFFu)] ^ (uint32_t)crc_tab[m & 0xFFu];
        }
      }
      {
        const uint32_t kk[16] = {kq0.x, kq0.y, kq0.z, kq0.w, kq1.x, kq1.y, kq1.z, kq1.w,
                                 kq2.x, kq2.y, kq2.z, kq2.w, kq3.x, kq3.y, kq3.z, kq3.w};
        uint32_t r = 0;
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) r ^= (0u - ((crc >> bit) & 1u)) & kk[bit];
        crc = r;
      }
      crc = x3_wave_xor_to_lane63_dpp(crc);
      if (lane == 63) part[16 + wid] = crc;

      X3_STAMP(5);
      __syncthreads();  // B4: CRC partials are in LDS; the next frame's samples and the previous frame's offset too
#ifdef X3_DBG_ALLWAVES
      X3_STAMP(7);
#else
      X3_STAMP(2);
#endif
      // (the helper wave writes this frame's header into the image behind this barrier)
      // ---- F: the PREVIOUS frame goes out now: its offset needed every predecessor's size, and that wait (in
      // the helper wave) overlapped this frame's analysis, emission and CRC
      if (have_prev) copy_out(img0 + (cur ^ 1u) * img_dwords, prev_bytes);
      X3_STAMP(6);
      __syncthreads();  // B4b: the previous frame's image has been read by every wave
#ifdef X3_DBG_ALLWAVES
      X3_STAMP(7);
#else
      X3_STAMP(2);
#endif
      // clear what it used, ready for the frame after this one.  The next B1 separates this from emission.
      if (have_prev) {
        uint4* z4 = reinterpret_cast<uint4*>(img0 + (cur ^ 1u) * img_dwords);
        const uint32_t nz = (((prev_bytes + 3u) >> 2) + 3u) >> 2;
        const uint4 zero = make_uint4(0, 0, 0, 0);
        for (uint32_t i = tid; i < nz; i += nthr) z4[i] = zero;
      }
      prev_bytes = frame_bytes;
      have_prev = true;
      cur ^= 1u;
    }
    __syncthreads();  // B5: the last frame's header and offset
    copy_out(img0 + (cur ^ 1u) * img_dwords, prev_bytes);
  }
#ifdef X3_DBG_STAMPS
#ifdef X3_DBG_ALLWAVES
  // (blocks 0..63 and 256..319: with 512 workgroups on 256 CUs those are first and second on their CUs)
  if (lane == 0 && (blockIdx.x < 64 || (blockIdx.x >= 256 && blockIdx.x < 320)))
    for (int k = 0; k < 8; ++k) x3_dbg[(((blockIdx.x & 63u) + (blockIdx.x >= 256 ? 64u : 0u)) * 9 + wid) * 8 + k] = dbg_acc[k];
#else
  if (lane == 0 && (wid == 0 || wid == 8) && blockIdx.x < 2048)
    for (int k = 0; k < 8; ++k) x3_dbg[(blockIdx.x * 2 + (wid == 8)) * 8 + k] = dbg_acc[k];
#endif
#endif
  __syncthreads();
  if (tid < 6) {
    const uint32_t v = part[32 + tid];  // < 2^32: at most 135 frames x 10 000 samples per workgroup... see host check
    if (v) atomicAdd(&stats[tid], (unsigned long long)v);
  }
}
