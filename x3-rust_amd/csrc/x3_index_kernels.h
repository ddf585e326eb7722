// x3_index_kernels.h -- the frame walk of X3aReader::decode_next_frame (src/decodefile.rs:105-121) on the
// GPU, for streams that are already in HBM and whose frame offsets are not known.
//
// The walk is a linked list threaded through the stream: frame i+1 starts at off_i + 20 + payload_len_i.
// Followed serially that is one dependent memory access per frame; here
//   1. every byte offset is tested IN PARALLEL for the key bytes "x3" and, on a hit, for a valid header
//      (decoder::read_frame_header, decoder.rs:69-118: header CRC, key, channels, length) -> candidates
//      {offset, payload_len, samples, kind}; a random pair of bytes is the key once in 65 536 and then still
//      has to pass a 16-bit CRC, so candidates ~ frames;
//   2. candidates go into an open-addressing hash table keyed by offset, and every candidate that the walk
//      would step over looks up its successor;
//   3. pointer doubling builds, per level r, the 2^r-th successor together with the sample count and node
//      count of the span, and
//   4. frame k of the stream is found from the start node by the binary digits of k -- all k in parallel --
//      which also yields its sample offset (an exclusive prefix sum along the list for free).
// A single thread then states how the walk ends, with the reference's rules: <= 20 bytes left, a header
// that does not validate (its error), a payload that runs past the end (quiet stop), a payload longer than
// the reader's buffer (FrameHeaderInvalidPayloadLen), a frame the decoder cannot take (BAD_ARG, included).
#pragma once
#include "x3_tables.h"
#include "x3_decode_kernel.h"

#define X3I_CONT 0u     // the walk steps over this frame
#define X3I_LAST_BAD 1u // pushed, then the walk stops with BAD_ARG (samples == 0, payload < 2 bytes, ...)
#define X3I_QUIET 2u    // header fine, payload runs past the end of the data: quiet stop, not a frame
#define X3I_PLEN 3u     // header fine, payload longer than X3_READ_BUFFER_SIZE: hard error, not a frame
#define X3I_IO 4u       // header fine, payload inside the bytes the reader believes in but past the real end: Io
#define X3I_NONE 0xFFFFFFFFu
#define X3I_READ_BUFFER 24576u
#ifndef X3I_DEFER_CHECK
#define X3I_DEFER_CHECK 1
#endif
#define X3I_WG_RAW 384u     // places with the key that a workgroup of x3_index_candidates_kernel lists before it checks them
#define X3I_WG_CANDS 256u   // candidates a workgroup of x3_index_candidates_kernel collects before it touches the global counter

struct X3Cand {
  unsigned long long off;
  uint32_t plen_kind;  // payload_len | kind << 16
  uint32_t samples;
};

// (struct X3IndexSummary: x3_tables.h)

// decoder::read_frame_header only (no walk checks): status, payload_len, samples
__device__ __forceinline__ int32_t x3i_read_header(const uint32_t* __restrict__ xw, uint64_t n_dw, uint64_t off,
                                                   uint32_t& plen, uint32_t& samples) {
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
  if ((h4 >> 16) != hc) return X3D_FRAME_HEADER_INVALID_HEADER_CRC;
  if ((h0 >> 16) != 0x7833u) return X3D_FRAME_HEADER_INVALID_KEY;
  if ((h0 & 0xFFu) > 1u) return X3D_MORE_THAN_ONE_CHANNEL;
  if (plen >= 0x7fe0u) return X3D_FRAME_LENGTH;
  return X3D_OK;
}

// what the walk does with a VALID header at `off` (decodefile.rs:114-121 and the decoder's preconditions)
// `believed` >= len: what the reader thinks the stream holds (X3aReader::open's 8 phantom bytes, decodefile.rs:62-66)
__device__ __forceinline__ uint32_t x3i_kind(uint64_t len, uint64_t believed, uint64_t off, uint32_t plen,
                                             uint32_t samples, uint32_t bl0) {
  if (believed - off - 20 < plen) return X3I_QUIET;
  if (plen > X3I_READ_BUFFER) return X3I_PLEN;  // tested before the payload is read (decodefile.rs:118-124)
  if (len - off - 20 < plen) return X3I_IO;
  if (samples == 0 || plen < 2 || (bl0 && samples > 1)) return X3I_LAST_BAD;
  return X3I_CONT;
}

// 0. the summary and the candidate counter start clean (one launch instead of two small copies)
__global__ void x3_index_init_kernel(X3IndexSummary* __restrict__ sum, unsigned int* __restrict__ count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  sum->n_frames = 0;
  sum->n_samples = 0;
  sum->terminal = 0;
  sum->last_node = X3I_NONE;
  sum->first_over = ~0ull;
  sum->n_chain = 0;
  sum->start = X3I_NONE;
  sum->pad = 0;
  sum->unaligned = 0;
  sum->pad2 = 0;
  *count = 0;
}

// 1. candidates: thread t looks at the 16 byte offsets of 16-byte chunk t.  cand == nullptr: count only.
// (Every offset, not only the even ones an encoder produces: read_frame_header accepts any payload_len, and a
// frame with an odd one puts its successor on an odd offset -- the host walk follows it there, so does this one.)
//
// ORDERED (round 4, the fast path of index_dev_impl): nothing goes through the global counter.  A workgroup scans ONE
// contiguous span of the stream, so its candidates are a contiguous run of the stream's candidates: it sorts its (few)
// candidates by offset and leaves them, their number and their sample sum in ITS slots of cand / count / samp
// (cand[blockIdx * X3I_WG_CANDS ...], count[blockIdx], samp[blockIdx]); x3_index_chain_kernel scans the counts and
// x3_index_link_kernel puts the candidates in order and checks, all at once, that each one ends where the next one
// begins -- which is all the pointer chasing of the general path amounts to on a stream that is one clean chain.
// A workgroup with more than X3I_WG_CANDS candidates says so in *not_simple (the general path then takes the stream).
template <bool ORDERED>
__global__ void __launch_bounds__(256)
x3_index_candidates_kernel(const uint32_t* __restrict__ xw, uint64_t len, uint64_t believed, uint32_t bl0,
                           X3Cand* __restrict__ cand, uint32_t cap, unsigned int* __restrict__ count,
                           unsigned long long* __restrict__ samp, uint32_t* __restrict__ not_simple) {
  __shared__ X3Cand s_c[X3I_WG_CANDS];
  __shared__ uint32_t s_n, s_base, s_nraw;
  __shared__ unsigned long long s_raw[X3I_WG_RAW];
  if (threadIdx.x == 0) { s_n = 0; s_nraw = 0; }
  __syncthreads();
  const uint64_t n_dw = (len + 3) >> 2;
  const uint64_t chunks = (len + 15) >> 4;
  // every workgroup walks ONE contiguous span of the stream (consecutive trips touch consecutive 4 KB)
  const uint64_t per_wg = ((chunks + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
  const uint64_t t_end = (uint64_t)(blockIdx.x + 1) * per_wg < chunks ? (uint64_t)(blockIdx.x + 1) * per_wg : chunks;
  // (the chunks of the next TWO trips are requested before this trip's is looked at; with one 16-byte load in flight per
  // lane the kernel took 118 us for config 3's 363 MB, with three it takes 115: 3.2 TB/s either way.  Round 4: hipcc waits
  // for all of them at the top of every trip -- it cannot count guarded loads --, but range-checked buffer loads in a loop
  // unrolled by three, which it does count (vmcnt(3) at the first use), made the kernel SLOWER: 136 us; the same loads with
  // nothing requested ahead and half the instructions per trip: 113 us; without the fifth dword: 103.  A bare read of the
  // same bytes, one span per workgroup, takes 58 us (tools/ubench/read_rate.hip: 6.3 TB/s): it is not the loads.  It was
  // the candidates: each header read in the loop held its wave for a memory round trip -- checked behind the loop, all at
  // once (X3I_DEFER_CHECK): 74 us.)
  auto fetch = [&](uint64_t t, uint32_t (&w)[5]) {
    if (t >= t_end) {
#pragma unroll
      for (int d = 0; d < 5; ++d) w[d] = 0u;
    } else if (4 * t + 4 < n_dw) {  // (all but the stream's last chunk)
      const uint4 v = reinterpret_cast<const uint4*>(xw)[t];
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      w[4] = xw[4 * t + 4];  // (taking it from the next lane's chunk by ds_bpermute instead: 125 against 115 us; by DPP wave_shl:1, round 4: the whole call 0.89-0.91 against 0.86-0.88 ms)
    } else {
#pragma unroll
      for (int d = 0; d < 5; ++d) w[d] = 4 * t + d < n_dw ? xw[4 * t + d] : 0u;
    }
  };
  auto consider = [&](uint64_t off) {
      uint32_t plen, samples;
      if (x3i_read_header(xw, n_dw, off, plen, samples) != X3D_OK) return;
      // (collected per workgroup: 70 000 atomics on ONE global counter serialise in L2, ~10 ns each -- that was
      // 0.75 of this kernel's 0.81 ms on config 3)
      X3Cand cd;
      cd.off = off;
      cd.plen_kind = plen | (x3i_kind(len, believed, off, plen, samples, bl0) << 16);
      cd.samples = samples;
      const uint32_t li = atomicAdd(&s_n, 1u);
      if (li < X3I_WG_CANDS) {
        s_c[li] = cd;
      } else if (!ORDERED) {  // (a span with more candidates than the workgroup's buffer holds: straight to the global counter)
        const unsigned int slot = atomicAdd(count, 1u);
        if (cand && slot < cap) cand[slot] = cd;
      }
  };
  const uint64_t t_first = (uint64_t)blockIdx.x * per_wg + threadIdx.x;
  uint32_t wa[5], wb[5];
  fetch(t_first, wa);
  fetch(t_first + blockDim.x, wb);
  for (uint64_t t = t_first; t < t_end; t += blockDim.x) {
    uint32_t w[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) { w[d] = wa[d]; wa[d] = wb[d]; }
    fetch(t + 2 * (uint64_t)blockDim.x, wb);
    // filter: is the key 0x78 0x33 at ANY of the sixteen byte offsets?  Halfword-zero test (x - 0x0001..) & ~x & 0x8000..
    // on the dwords XOR the key, for the even offsets as they are and for the odd ones shifted by a byte.  One chunk
    // in 4 000 passes (random bytes), so the exact per-offset work below is rare.
    uint32_t hit = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t e = w[d] ^ 0x33783378u;
      const uint32_t o = __builtin_amdgcn_alignbit(w[d + 1], w[d], 8) ^ 0x33783378u;
      hit |= ((e - 0x00010001u) & ~e) | ((o - 0x00010001u) & ~o);
    }
    if ((hit & 0x80008000u) == 0) continue;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      // bytes b, b+1 of the chunk in memory order
      const uint32_t lo = w[b >> 2] >> (8 * (b & 3));
      const uint32_t hw = ((b & 3) == 3 ? (lo & 0xFFu) | ((w[(b >> 2) + 1] & 0xFFu) << 8) : lo) & 0xFFFFu;
      if (hw != 0x3378u) continue;  // bytes 0x78 0x33
      const uint64_t off = 16 * t + b;
      if (off + 20 > len) continue;
      // the key is there: the header is read and checked BEHIND the loop, all of the workgroup's at once (X3I_DEFER_CHECK;
      // in the loop every one of them held its wave for a memory round trip)
      const uint32_t ri = atomicAdd(&s_nraw, 1u);
      if (X3I_DEFER_CHECK && ri < X3I_WG_RAW) {
        s_raw[ri] = off;
        continue;
      }
      consider(off);
    }
  }
  __syncthreads();
  {
    const uint32_t nraw = s_nraw < X3I_WG_RAW ? s_nraw : X3I_WG_RAW;
    if (X3I_DEFER_CHECK)
      for (uint32_t i = threadIdx.x; i < nraw; i += blockDim.x) consider(s_raw[i]);
  }
  __syncthreads();
  const uint32_t mine = s_n < X3I_WG_CANDS ? s_n : X3I_WG_CANDS;
  if (ORDERED) {
    if (threadIdx.x == 0) {
      if (s_n > X3I_WG_CANDS) *not_simple = 1u;
      count[blockIdx.x] = mine;
    }
    // rank sort by offset (a dozen candidates per workgroup on config 3; 256 at most), samples summed on the way
    X3Cand* const dst = cand + (size_t)blockIdx.x * X3I_WG_CANDS;
    unsigned long long tot = 0;
    for (uint32_t i = threadIdx.x; i < mine; i += blockDim.x) {
      const X3Cand me = s_c[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < mine; ++j) rank += s_c[j].off < me.off ? 1u : 0u;
      dst[rank] = me;
    }
    if (threadIdx.x == 0) {
      for (uint32_t j = 0; j < mine; ++j) tot += s_c[j].samples;
      samp[blockIdx.x] = tot;
    }
    return;
  }
  if (threadIdx.x == 0) s_base = mine ? atomicAdd(count, mine) : 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < mine; i += blockDim.x)
    if (cand && s_base + i < cap) cand[s_base + i] = s_c[i];
}

__device__ __forceinline__ uint32_t x3i_hash(unsigned long long off, uint32_t mask) {
  unsigned long long x = off * 0x9E3779B97F4A7C15ull;
  return (uint32_t)(x >> 32) & mask;
}

// 2a. hash table: keys[] = offset + 1 (0 = empty), vals[] = candidate index
__global__ void __launch_bounds__(256)
x3_index_hash_insert_kernel(const X3Cand* __restrict__ cand, uint32_t n, unsigned long long* __restrict__ keys,
                            uint32_t* __restrict__ vals, uint32_t mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = cand[i].off + 1ull;
  uint32_t h = x3i_hash(cand[i].off, mask);
  for (;;) {
    const unsigned long long prev = atomicCAS(&keys[h], 0ull, key);
    if (prev == 0ull) { vals[h] = i; return; }
    h = (h + 1u) & mask;
  }
}

__device__ __forceinline__ uint32_t x3i_lookup(unsigned long long off, const unsigned long long* __restrict__ keys,
                                               const uint32_t* __restrict__ vals, uint32_t mask) {
  uint32_t h = x3i_hash(off, mask);
  for (;;) {
    const unsigned long long k = keys[h];
    if (k == 0ull) return X3I_NONE;
    if (k == off + 1ull) return vals[h];
    h = (h + 1u) & mask;
  }
}

// 2b. level 0: successor, samples and node count of every candidate
__global__ void __launch_bounds__(256)
x3_index_succ_kernel(const X3Cand* __restrict__ cand, uint32_t n, const unsigned long long* __restrict__ keys,
                     const uint32_t* __restrict__ vals, uint32_t mask, uint32_t* __restrict__ J0,
                     unsigned long long* __restrict__ S0, uint32_t* __restrict__ L0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t kind = cand[i].plen_kind >> 16;
  uint32_t succ = X3I_NONE;
  if (kind == X3I_CONT) {
    const uint32_t j = x3i_lookup(cand[i].off + 20ull + (cand[i].plen_kind & 0xFFFFu), keys, vals, mask);
    if (j != X3I_NONE && (cand[j].plen_kind >> 16) <= X3I_LAST_BAD) succ = j;
  }
  J0[i] = succ;
  S0[i] = kind == X3I_CONT ? cand[i].samples : 0ull;  // a LAST_BAD frame is pushed but its samples are not counted
  L0[i] = 1u;
}

// 3. one doubling step: level r from level r-1
__global__ void __launch_bounds__(256)
x3_index_double_kernel(uint32_t n, const uint32_t* __restrict__ Jp, const unsigned long long* __restrict__ Sp,
                       const uint32_t* __restrict__ Lp, uint32_t* __restrict__ J, unsigned long long* __restrict__ S,
                       uint32_t* __restrict__ L) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = Jp[i];
  if (j == X3I_NONE) {
    J[i] = X3I_NONE;
    S[i] = Sp[i];
    L[i] = Lp[i];
  } else {
    J[i] = Jp[j];
    S[i] = Sp[i] + Sp[j];
    L[i] = Lp[i] + Lp[j];
  }
}

// 3b. FOUR doubling steps in one launch: levels r .. r+3 from level r-1 alone -- the 2^(r+3)-th successor is the
// 2^(r-1)-th successor taken sixteen times, and the levels in between fall out on the way (after 2, 4 and 8 steps).
// Sixteen dependent gathers from L2 are cheaper than three more launches of a kernel that does one.
__global__ void __launch_bounds__(256)
x3_index_double4_kernel(uint32_t n, const uint32_t* __restrict__ Jp, const unsigned long long* __restrict__ Sp,
                        const uint32_t* __restrict__ Lp, uint32_t* __restrict__ J, unsigned long long* __restrict__ S,
                        uint32_t* __restrict__ L) {  // J/S/L: level r; the three levels behind it follow at strides of n
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = Jp[i], lacc = Lp[i];
  unsigned long long sacc = Sp[i];
  uint32_t out = 0;
#pragma unroll
  for (uint32_t step = 2; step <= 16; ++step) {  // x = the node `step - 1` jumps of 2^(r-1) behind i
    if (x != X3I_NONE) {
      sacc += Sp[x];
      lacc += Lp[x];
      x = Jp[x];
    }
    if ((step & (step - 1u)) == 0u) {  // 2, 4, 8, 16 jumps: levels r, r+1, r+2, r+3
      J[(size_t)out * n + i] = x;
      S[(size_t)out * n + i] = sacc;
      L[(size_t)out * n + i] = lacc;
      ++out;
    }
  }
}

// 4. frame k = the k-th successor of the start node; its sample offset = the samples of the k nodes before it.
// levels: J/S arrays of level r start at r*n.  The start node and the chain's length are looked up here, so the
// host does not have to fetch them in between: the grid covers all n candidates.
// A chain longer than max_frames: sum->pad = 1 (the caller's arrays are too small), nothing behind them is written.
__global__ void __launch_bounds__(256)
x3_index_emit_kernel(const X3Cand* __restrict__ cand, uint32_t n, uint32_t levels, const uint32_t* __restrict__ J,
                     const unsigned long long* __restrict__ S, unsigned long long max_frames,
                     unsigned long long wav_cap, unsigned long long* __restrict__ frame_off,
                     unsigned long long* __restrict__ wav_off, X3IndexSummary* __restrict__ sum,
                     const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t mask,
                     const uint32_t* __restrict__ L_top) {
  const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  // the start node = the candidate at offset 0, if the walk can step onto it, and the length of its chain: every
  // thread looks them up for itself (one launch less), thread 0 leaves them in the summary for the last kernel
  uint32_t start = x3i_lookup(0ull, keys, vals, mask);
  if (start != X3I_NONE && (cand[start].plen_kind >> 16) > X3I_LAST_BAD) start = X3I_NONE;
  const unsigned long long n_chain = start == X3I_NONE ? 0ull : (unsigned long long)L_top[start];
  if (k == 0) {
    sum->start = start;
    sum->n_chain = n_chain;
  }
  if (start == X3I_NONE || k >= n_chain) return;
  if (n_chain > max_frames) {
    if (k == 0) sum->pad = 1;
    return;
  }
  uint32_t node = start;
  unsigned long long acc = 0;
  for (uint32_t r = 0; r < levels; ++r) {
    if ((k >> r) & 1ull) {
      acc += S[(size_t)r * n + node];
      node = J[(size_t)r * n + node];
    }
  }
  frame_off[k] = cand[node].off;
  wav_off[k] = acc;
  if ((acc & 3ull) && sum->unaligned == 0) sum->unaligned = 1;  // a sample offset that is not a multiple of four
  // decodefile walk as the host runs it: a frame that does not fit the output is pushed and ends the walk
  if ((cand[node].plen_kind >> 16) == X3I_CONT && acc + cand[node].samples > wav_cap) atomicMin(&sum->first_over, k);
  if (k == n_chain - 1) sum->last_node = node;
}

// 5. how the walk ends (one thread)
__global__ void x3_index_finalize_kernel(const uint32_t* __restrict__ xw, uint64_t len, uint64_t believed, uint32_t bl0,
                                         const X3Cand* __restrict__ cand,
                                         const unsigned long long* __restrict__ wav_off, X3IndexSummary* __restrict__ sum) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t start = sum->start;
  const unsigned long long n_chain = sum->n_chain;
  if (sum->pad) return;  // more frames than the caller's arrays hold: the host reports it
  const uint64_t n_dw = (len + 3) >> 2;
  auto ending_at = [&](uint64_t pos) -> int {  // the walk arrives at `pos` and finds no frame to push
    if (believed - pos <= 20) return X3D_OK;
    if (len - pos < 20) return X3D_IO;  // read_exact of a header the reader believes in
    uint32_t plen, samples;
    const int32_t st = x3i_read_header(xw, n_dw, pos, plen, samples);
    if (st != X3D_OK) return st;
    const uint32_t kind = x3i_kind(len, believed, pos, plen, samples, bl0);
    if (kind == X3I_QUIET) return X3D_OK;
    if (kind == X3I_IO) return X3D_IO;
    if (kind == X3I_PLEN) return X3D_FRAME_HEADER_INVALID_PAYLOAD_LEN;
    return X3D_BAD_ARG;  // unreachable: such a header is a candidate and would be part of the chain
  };
  if (start == X3I_NONE) {
    sum->n_frames = 0;
    sum->n_samples = 0;
    sum->terminal = ending_at(0);
    return;
  }
  if (sum->first_over < n_chain) {  // the first frame that does not fit the output: pushed, BAD_ARG
    sum->n_frames = sum->first_over + 1;
    sum->n_samples = wav_off[sum->first_over];
    sum->terminal = X3D_BAD_ARG;
    return;
  }
  const X3Cand last = cand[sum->last_node];
  sum->n_frames = n_chain;
  if ((last.plen_kind >> 16) == X3I_LAST_BAD) {
    sum->n_samples = wav_off[n_chain - 1];
    sum->terminal = X3D_BAD_ARG;
  } else {
    sum->n_samples = wav_off[n_chain - 1] + last.samples;
    sum->terminal = ending_at(last.off + 20ull + (last.plen_kind & 0xFFFFu));
  }
}


// ---- the fast path's two small kernels (see x3_index_candidates_kernel<true>)
// F1. exclusive scans of the workgroups' candidate counts and sample sums (one workgroup; G <= 4 096 spans)
__global__ void __launch_bounds__(1024)
x3_index_chain_kernel(const unsigned int* __restrict__ count, const unsigned long long* __restrict__ samp, uint32_t G,
                      uint32_t* __restrict__ base, unsigned long long* __restrict__ sbase, X3IndexSummary* __restrict__ sum) {
  __shared__ uint32_t s_c[1024];
  __shared__ unsigned long long s_s[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (G + 1023u) / 1024u;
  uint32_t c = 0;
  unsigned long long sv = 0;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t b = t * per + i;
    if (b < G) { c += count[b]; sv += samp[b]; }
  }
  s_c[t] = c;
  s_s[t] = sv;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {   // inclusive Hillis-Steele scan
    const uint32_t cv = t >= d ? s_c[t - d] : 0u;
    const unsigned long long sv2 = t >= d ? s_s[t - d] : 0ull;
    __syncthreads();
    s_c[t] += cv;
    s_s[t] += sv2;
    __syncthreads();
  }
  uint32_t cb = s_c[t] - c;
  unsigned long long sb = s_s[t] - sv;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t b = t * per + i;
    if (b < G) {
      base[b] = cb;
      sbase[b] = sb;
      cb += count[b];
      sb += samp[b];
    }
  }
  if (t == 1023u) sum->n_chain = s_c[1023];   // candidates in all (= frames, if the chain turns out clean)
}

// F2. candidate i of span b is candidate k = base[b] + i of the stream.  It is frame k of a CLEAN chain iff the first
// one sits at offset 0, every one is a frame the walk steps over (X3I_CONT), and each ends where the next begins;
// anything else sets sum->pad2 (the host then runs the general path).  Writes the sorted candidates (for
// x3_index_finalize_kernel), the frame offsets and the sample offsets.
__global__ void __launch_bounds__(64)
x3_index_link_kernel(const X3Cand* __restrict__ cand_wg, const unsigned int* __restrict__ count, uint32_t G,
                     const uint32_t* __restrict__ base, const unsigned long long* __restrict__ sbase,
                     X3Cand* __restrict__ sorted, uint32_t sorted_cap, unsigned long long max_frames,
                     unsigned long long wav_cap, unsigned long long* __restrict__ frame_off,
                     unsigned long long* __restrict__ wav_off, X3IndexSummary* __restrict__ sum) {
  const uint32_t b = blockIdx.x;
  const uint32_t n = count[b];
  const unsigned long long total = sum->n_chain;
  if (total == 0ull || total > max_frames || total > sorted_cap) {
    // no candidate at all: the general path states how the walk ends.  More CANDIDATES than max_frames says nothing about
    // the chain -- a truncated last frame, a frame behind junk or a false header inside a payload is a candidate and not a
    // frame -- so that case, too, goes to the general walk, which sets `pad` from the real chain length.
    if (b == 0 && threadIdx.x == 0) sum->pad2 = 1;
    return;
  }
  const X3Cand* const mine = cand_wg + (size_t)b * X3I_WG_CANDS;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const X3Cand cd = mine[i];
    const unsigned long long k = (unsigned long long)base[b] + i;
    unsigned long long acc = sbase[b];
    for (uint32_t j = 0; j < i; ++j) acc += mine[j].samples;
    bool ok = (cd.plen_kind >> 16) == X3I_CONT;
    if (k == 0ull) ok = ok && cd.off == 0ull;
    // the next candidate of the stream: the next one of this span, or the first one of the next span that has any
    unsigned long long next_off = ~0ull;
    if (i + 1u < n) {
      next_off = mine[i + 1u].off;
    } else {
      for (uint32_t b2 = b + 1u; b2 < G; ++b2)
        if (count[b2]) { next_off = cand_wg[(size_t)b2 * X3I_WG_CANDS].off; break; }
    }
    if (next_off != ~0ull) ok = ok && cd.off + 20ull + (cd.plen_kind & 0xFFFFu) == next_off;
    else sum->last_node = (uint32_t)k;          // the stream's last candidate
    if (!ok) sum->pad2 = 1;
    sorted[k] = cd;
    frame_off[k] = cd.off;
    wav_off[k] = acc;
    if ((acc & 3ull) && sum->unaligned == 0) sum->unaligned = 1;
    if (acc + cd.samples > wav_cap) atomicMin(&sum->first_over, k);
    if (k == 0ull) sum->start = 0u;
  }
}
