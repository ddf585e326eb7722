// x3_encode_stream2_kernel.h -- single-pass frame encoder for block_len = 20 (and, round 6, 10 and 40: template parameter BL),
// second generation.
//
// Same single pass as round 1's first-generation kernel (retired in round 4; persistent co-resident grid, frame f = blockIdx.x + k*G, stream
// offsets from the frame sizes every workgroup publishes as {epoch:12 | bytes:20} words -- no prefix chain, HBM
// traffic = 2 B/sample in + stream bytes out), restructured around what bounded the first one (profiles/r1:
// 303 M VALU + 112 M SALU per launch at 68 % VALU-busy, two 9-wave workgroups per CU):
//
//  * EIGHT waves, no helper wave.  Nine waves land 3+2+2+2 on the four SIMDs, and two such workgroups make one
//    SIMD carry five or six waves while the others carry four: the CU runs at the pace of its fullest SIMD.
//    Eight waves are two per SIMD, always.  The helper's duties are spread over the compute waves:
//      - sizes: every thread holds TWO of the (up to 1023) size words in front of the workgroup's previous frame,
//        requested at the end of one iteration and due at the end of the next one's emission: per wave a readiness
//        test, a DPP sum and one LDS partial; behind the barrier every thread adds the eight partials;
//      - header: wave 0, behind the CRC barrier, while the others clear the image that has just gone out.
//  * NO sample tile in LDS.  A lane's block is 40 contiguous bytes (+ 4): it loads them straight from HBM into
//    the registers the analysis uses (16 + 16 + 8 + 4 bytes per lane, a wave covers 2 560 contiguous bytes), for
//    the NEXT frame as soon as this frame's block has been emitted, so the loads fly under the CRC pass, the
//    copy-out, the barriers and the other workgroups of the CU.  They are range-checked buffer loads (descriptor =
//    the frame): a lane behind the frame's last sample reads zeros, no branch, nothing behind the frame is touched.
//    That removes the 20 KB tile (LDS per workgroup 63 -> 49 KB: THREE workgroups per CU), its per-lane
//    ds_read_b64 at a 40-byte stride (the bank conflicts SURVEY section 7 warns about) and the DMA bookkeeping.
//  * Three barriers per frame instead of four: the previous frame goes out in the same phase as the CRC pass of the
//    current one (its offset is complete at the emission barrier), and the barrier behind both also opens the image
//    for clearing.
//  * Nothing in the loop waits for a VMEM result except the analysis (for its samples) and the size words (issued
//    before them): loads return in order, so every other wait would also wait for the samples in flight.  The CRC
//    multipliers therefore come from LDS (below), and the kernel must not spill (a scratch reload is a VMEM load).
//  * ONE code path for blocks of any length 1..20 (lane masks per pair); the plain case -- every block of the wave
//    has 20 samples, or 19 in the frame's last block -- skips the masks.
//
// Closed-form bit lengths, the (code, len) emission recipe, the slicing-by-4 chunk CRC and the bounded size wait with
// its time-out protocol are those of the first kernel (helpers are shared from its header).
#pragma once
#include <type_traits>

#include "x3_encode_common.h"

#ifndef X3E_PACE_CLIMB
#define X3E_PACE_CLIMB 1
#endif
#ifndef X3E_PACE_DIV
#define X3E_PACE_DIV 16u  // the target is the slowest workgroup's pace of the launch before less 1/DIV
#endif
#ifndef X3E_BAND
#define X3E_BAND 16  // sixteenths of a frame ahead / behind that move a workgroup one priority level
#endif
#define X3_STREAM2_THREADS 512u
#define X3_STREAM2_MAX_GRID 1024u  // two size words per thread cover 1023 predecessors
#define X3_STREAM2_DESC_PAD 1088u  // words in front of desc[0]: the windows of the first frames reach below frame 0

// The payload CRC pass can be left to the first X3E_CRC_WAVES waves of the workgroup while the others copy the previous
// frame out: fewer waves pay the pass's fixed ~75 instructions (multipliers, reduction) for the same table work.  It
// does not pay -- the pass is a chain of dependent LDS look-ups per lane, and a longer chain in fewer waves is time the
// other waves then spend at B4: config 3 encodes in 0.78 / 0.66 / 0.63 / 0.613 / 0.615 ms with 1 / 2 / 3 / 4 / 8 waves.
#include "x3_tables.h"   // X3E_CRC_WAVES / _LANES, X3_STREAM2_MAX_PAYLOAD_DWORDS, X3_K2_*

// CRC multipliers (x3_ctx.hip builds them, one block of X3_K2_DWORDS per chunk size c = 1..X3_K2_MAXC):
//   KN[l][j][v], l < 64, j < 4, v < 16 (uint16; rows of 33 dwords, 32 used: consecutive lanes start one bank apart):
//             (v << 4j) * x^(32*c*(63-l)) mod P -- lane l's dwords are followed by c*(63-l) dwords of its WAVE's segment,
//             and its 16-bit partial is multiplied by that power nibble by nibble: four look-ups and ten instructions
//             where sixteen pre-shifted words and their bit tests took 48;
//   KA[w][b], w < 8: x^(32*c*64*(7-w)) * x^b -- a wave's segment is followed by those of the CRC waves behind it
//             (CRC wave w of X3E_CRC_WAVES uses row 8 - X3E_CRC_WAVES + w).
// crc0(payload) = XOR_w KA[w] * (XOR_l KN[l] * crc0(chunk of lane l of wave w)).  The block of the current chunk
// size lives in LDS (8.75 KB; reloaded by the workgroup when c changes between frames, which it rarely does).

typedef uint32_t x3_v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t x3_v2u32 __attribute__((ext_vector_type(2)));

// LDS: part[64] | image 0 | image 1 | CRC tables (4 x 256 x u16) | multipliers of the current chunk size
// part: [0..7] scan partials, [8..15] size-sum partials, [16..23] CRC partials, [32..37] stats,
//       [41] offsets lost to a size-wait time-out
// ctl: the context's 128-byte control block: int status[8] | u64 stats[6] | u64 end_pos
// The host only takes this path for parameter sets whose thresholds keep every Rice block inside the reference's
// table for its code (x3_encode.hip, stream_safe_thresholds), so there is no "outside the table" test here.
// Waves per SIMD the register budget is cut for: 6 = 80 VGPRs (79 used) = three workgroups per CU, which is also what
// two worst-case frame images per workgroup leave room for in LDS.  Four per CU were tried with 12 KB images (enough for
// config 3) and a 64-VGPR build: 9 registers spill, 0.72 ms with three workgroups, 0.67 ms with four -- against 0.615.
#ifndef X3E_WAVES_PER_SIMD
#define X3E_WAVES_PER_SIMD 6
#endif
// A payload of more than this many bytes does not fit the wave encoder's LDS image (x3_encode_wave_kernel.h: 38 rows of
// 256 bytes).  The control block's word X3_CTL_DENSE_COUNT counts a call's frames beyond it: the wave encoder leaves
// them to this kernel's LIST form, and the host reads the count as a hint for the next call (x3_encode.hip).
#define X3_DENSE_PAYLOAD_BYTES 9728u
#define X3_CTL_DENSE_COUNT 96u   // byte offset in ctl (behind status[8], stats[6], end_pos)
// LIST = false: the single-pass encoder of a whole call (above).
// LIST = true (round 4): the DENSE PASS behind x3_encode_wave_kernel -- the frames that kernel could not hold in its
// image (`dense_list`, their number in the control block) are encoded here, one workgroup per frame with worst-case
// images, and written at the offsets the wave kernel has already assigned (frame_off[f]: every frame's size comes out
// of the analysis, whether it is emitted or not).  No size words, no waits, no residency requirement, no pacing; the
// statistics were counted by the wave kernel.  encoder.rs:289-315 makes no difference between a loud frame and a quiet
// one, and neither does a call any more: one loud frame costs one workgroup a few microseconds, not a second encode of
// the whole call (rounds 2-3).
// TAB: the frames come from a table (x3_encode_frames_dev; g.src_off / g.src_n).  A template parameter: as a run-time test
// it cost the uniform layout 17 % (white noise 0.94 ms against 0.80).
// BL (round 6): the block length.  A lane always holds 20 samples -- loads, pair masks, the scan of bit lengths, the CRC pass
// and the copy-out do not know about blocks at all.  What a block decides is the filter (its largest |difference|), its
// header and the statistics: with BL = 10 a lane holds TWO blocks (two maxima, two headers, one run of bits), with BL = 40
// a block is TWO neighbouring lanes (one quad-permute DPP shares the maximum, the even lane writes the header).  The
// BL = 20 instantiation is the code of rounds 2-5, untouched (`if constexpr`; its instruction stream compared equal when this
// was added); tools/check_encoder_isa.py keeps every instantiation inside its register budget.
template <bool LIST, bool TAB = false, uint32_t BL = 20u>
__global__ void __launch_bounds__(X3_STREAM2_THREADS, X3E_WAVES_PER_SIMD)
x3_encode_stream2_kernel(const int16_t* __restrict__ wav, X3Geom g, X3DevParams p,
                         uint64_t* __restrict__ frame_off, uint8_t* __restrict__ out, uint64_t out_cap,
                         uint64_t start_pos, uint32_t* __restrict__ desc, uint32_t epoch,
                         unsigned char* __restrict__ ctl, const uint32_t* __restrict__ xk2,
                         const uint16_t* __restrict__ crc_tab_g, uint32_t img_dwords, uint32_t* __restrict__ pace,
                         const uint32_t* __restrict__ dense_list, uint32_t* __restrict__ ctl_next = nullptr) {
  // (round 5) the dense pass is the last kernel of a wave-encoder call: it clears the control block the context's NEXT encode
  // call uses (the context alternates between two), which then needs no memset of its own in front of its first kernel
  if (LIST && ctl_next && blockIdx.x == 0 && threadIdx.x < 32u) ctl_next[threadIdx.x] = 0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* part = reinterpret_cast<uint32_t*>(smem);
  uint32_t* img0 = reinterpret_cast<uint32_t*>(smem + X3_ENC_SMEM_HDR);
  uint16_t* crc_tab = reinterpret_cast<uint16_t*>(img0 + 2u * img_dwords);
  uint32_t* ktab = reinterpret_cast<uint32_t*>(crc_tab + 1024);
  int* const status = reinterpret_cast<int*>(ctl);
  unsigned long long* const stats = reinterpret_cast<unsigned long long*>(ctl + 32);
  unsigned long long* const end_pos = stats + 6;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t nthr = X3_STREAM2_THREADS;
  const uint64_t base_pos = (start_pos + 1ull) & ~1ull;  // writer.align::<2>() (encoder.rs:182)
  const uint32_t kpack = p.k[0] | (p.k[1] << 8) | (p.k[2] << 16);
  const uint32_t G = gridDim.x;
  const uint32_t ready_tag = epoch << X3_DESC_BYTES_BITS;

  // (a frame table, x3_encode_frames_dev: `clip` carries the frame number and idx stays 0)
  auto geom_at = [&](uint64_t clip, uint32_t idx, const int16_t*& src, uint32_t& n) __attribute__((always_inline)) {
    if (TAB) {
      n = g.src_n[clip];
      src = wav + g.src_off[clip];
      return;
    }
    const uint64_t left = g.n_per_clip - (uint64_t)idx * (uint64_t)p.spf;
    n = left < p.spf ? (uint32_t)left : p.spf;
    src = wav + clip * g.clip_stride + (uint64_t)idx * (uint64_t)p.spf;
  };
  auto geom_advance = [&](uint64_t& clip, uint32_t& idx) __attribute__((always_inline)) {
    if (TAB) { clip += G; return; }
    if (G < g.fpc) {
      idx += G;
      if (idx >= g.fpc) { idx -= g.fpc; ++clip; }
    } else {  // clips shorter than the grid is wide
      const uint64_t t = (uint64_t)idx + G;
      clip += t / g.fpc;
      idx = (uint32_t)(t % g.fpc);
    }
  };
  uint32_t* const dense_count = reinterpret_cast<uint32_t*>(ctl + X3_CTL_DENSE_COUNT);
  // LIST: this workgroup takes entries blockIdx.x, blockIdx.x + G, ... of the list
  const uint32_t n_list = LIST ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*dense_count) : 0u;
  if (LIST && blockIdx.x >= n_list) return;
  auto geom_of = [&](uint64_t f_, uint64_t& clip, uint32_t& idx) __attribute__((always_inline)) {
    if (TAB) { clip = f_; idx = 0; return; }
    clip = f_ / g.fpc;
    idx = (uint32_t)(f_ - clip * g.fpc);
  };
  uint64_t clip_f;
  uint32_t idx_f;
  geom_of(LIST ? (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)dense_list[blockIdx.x]) : (uint64_t)blockIdx.x, clip_f, idx_f);
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  unsigned long long dbg_spins = 0, dbg_late = 0;  // polls of settle(); frames whose sizes were not all there at first look
#endif

  // this lane's block of a frame of n samples at src: samples 20b .. 20b+21 as eleven (even, odd) pairs.  Block b
  // holds samples 20b+1 .. 20b+20 and is predicted from sample 20b.  The buffer descriptor covers the frame's
  // bytes rounded up to a dword (an odd frame's last dword also holds the sample behind the frame, in the same
  // page: it is masked by the block's sample count); dwords behind it read as zero.
  uint32_t W[11];
  auto load_block = [&](const int16_t* src, uint32_t n) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(src), 0, (int)((2u * n + 3u) & ~3u), 0x00020000);
    const uint32_t vo = 40u * tid;
    const x3_v4u32 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0);
    const x3_v4u32 c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16u), 0, 0);
    const x3_v2u32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(vo + 32u), 0, 0);
    W[10] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(vo + 40u), 0, 0);
    W[0] = a.x; W[1] = a.y; W[2] = a.z; W[3] = a.w;
    W[4] = c.x; W[5] = c.y; W[6] = c.z; W[7] = c.w;
    W[8] = d.x; W[9] = d.y;
  };

  // ---- prologue
  if (tid < 64) part[tid] = 0;
  for (uint32_t i = tid; i < (2u * img_dwords) >> 2; i += nthr)
    reinterpret_cast<uint4*>(img0)[i] = make_uint4(0, 0, 0, 0);  // both frame images start clear
  reinterpret_cast<uint32_t*>(crc_tab)[tid] = reinterpret_cast<const uint32_t*>(crc_tab_g)[tid];  // 4 x 256 x u16
  {
    const int16_t* src;
    uint32_t n;
    geom_at(clip_f, idx_f, src, n);
    load_block(src, n);
    // (waited for here, once: the loop is then entered with nothing in flight, and the wait counts hipcc derives for
    // the analysis come from the loop's own order of loads alone -- samples, then size words -- so that the analysis
    // waits for its samples and not for the size words issued behind them)
    x3_dma_wait();
  }
  __syncthreads();

  // the frame that has been encoded and waits for its offset
  uint32_t prev_bytes = 0, q0 = 0, q1 = 0;
  uint64_t prev_f = 0, my_off = 0;
  uint32_t my_bytes = 0;
  bool have_prev = false, first = true;
  uint32_t cur = 0;
  uint32_t ktab_c = 0;  // chunk size whose multipliers are in LDS (0: none yet)
  uint32_t n_dense = 0;  // thread 0: this workgroup's frames beyond X3_DENSE_PAYLOAD_BYTES (one atomic per workgroup: 69 120
                         // adds to one address cost a white-noise launch 2 ms)

  // The size words in front of frame pf: thread t holds those of frames pf-1-t and pf-513-t (the array has
  // X3_STREAM2_DESC_PAD words in front, so a window that reaches below frame 0 reads padding; range and readiness
  // are tested when a word is USED, so requesting never waits).  sc1 loads: the words are written by other
  // workgroups in this launch (cdna_hip_programming.md G16, form R2: the word is its own flag).
  auto desc_load2 = [&](uint64_t pf, uint32_t& a, uint32_t& b) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(desc + pf - 1024u, 0, 4096, 0x00020000);
    a = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * (1023u - tid)), 0, 16);
    b = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * (511u - tid)), 0, 16);
  };
  // this wave's share of the sum of the sizes in front of prev_f -> part[8 + wid]
  auto settle = [&]() __attribute__((always_inline)) {
    const uint32_t needed = first ? (uint32_t)prev_f : G - 1u;  // sizes in front of prev_f that count
    const bool in0 = tid < needed, in1 = tid + 512u < needed;
    uint32_t v0 = in0 ? q0 : ready_tag, v1 = in1 ? q1 : ready_tag;  // "ready, 0 bytes" outside the range
    uint32_t spins = 0;
    bool timeout = false;
    while (__any(((v0 >> X3_DESC_BYTES_BITS) != epoch) || ((v1 >> X3_DESC_BYTES_BITS) != epoch))) {
      // give up after the bounded spin -- or as soon as ANY workgroup has (then the launch is lost anyway and
      // every further wait would only add its own 15 ms): the host re-encodes with the two-pass kernels
      if (++spins > X3_SPIN_LIMIT ||
          __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == X3D_SIZE_WAIT_TIMEOUT) {
        timeout = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      uint32_t r0, r1;
      desc_load2(prev_f, r0, r1);
      v0 = in0 ? r0 : ready_tag;
      v1 = in1 ? r1 : ready_tag;
    }
#ifdef X3_DBG_STAMPS
    dbg_spins += spins;
    dbg_late += spins ? 1u : 0u;
#endif
    // 128 words x 2^20 < 2^32: a 32-bit DPP scan, total in lane 63
    const uint32_t sum = x3_wave_incl_scan_dpp((v0 & X3_DESC_BYTES_MASK) + (v1 & X3_DESC_BYTES_MASK));
    if (lane == 63) part[8 + wid] = sum;
    if (timeout && lane == 0) {
      // This workgroup no longer knows where its frames go -- this one and, since each offset builds on the
      // last, every later one.  Nothing of them may reach the output, the frame index or the end position:
      // x3_encode_result re-encodes the whole call with the two-pass kernels, and those rewrite
      // d_out[start_pos..) only -- bytes in front of start_pos belong to the caller.
      atomicMax(&status[1], X3D_SIZE_WAIT_TIMEOUT);
      part[41] = 1;
    }
  };
  // behind the barrier that follows settle(): the previous frame's stream offset, on every thread (in SGPRs)
  auto resolve = [&]() __attribute__((always_inline)) -> uint64_t {
    const uint4 a = reinterpret_cast<const uint4*>(part + 8)[0], b = reinterpret_cast<const uint4*>(part + 8)[1];
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w));
    if (LIST) return frame_off[prev_f];  // the wave kernel put every frame's offset there
    const uint64_t off = (first ? base_pos : my_off + my_bytes) + tot;  // (1023 sizes < 2^20 each: < 2^30)
    my_off = off;
    my_bytes = prev_bytes;
    first = false;
    if (tid == 0 && part[41] == 0) {
      frame_off[prev_f] = off;
      if (off + prev_bytes > out_cap) atomicMax(&status[0], X3D_BYTE_WRITER_INSUFFICIENT_MEMORY);
      if (prev_f == g.n_frames - 1) {
        frame_off[g.n_frames] = off + prev_bytes;
        *end_pos = off + prev_bytes;
      }
      if (prev_f == 0 && (start_pos & 1ull) && start_pos < out_cap) out[start_pos] = 0;  // align pad byte
    }
    return off;
  };
  auto copy_out = [&](const uint32_t* img, uint64_t off, uint32_t total_bytes, uint32_t tid, uint32_t nthr)
                      __attribute__((always_inline)) {
    // (tid, nthr: index in and size of the team of waves that copies)
    // header + payload of a finished frame to its final stream position.  Buffer stores over [dst, dst + bytes):
    // a uniform base and 32-bit per-lane offsets (no 64-bit address arithmetic per lane: registers are tight).
    if (off + total_bytes <= out_cap && part[41] == 0) {
      uint8_t* dst = out + off;
      const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u);
      // sixteen bytes per lane and trip (a default frame is ~330 of them: one trip, two thirds of the lanes);
      // the stream position is even, so the image is either dword-aligned to it or two bytes off
      if (mis == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)total_bytes, 0x00020000);
        const uint32_t ndw = total_bytes >> 2, nq = ndw >> 2;
        const uint4* img4 = reinterpret_cast<const uint4*>(img);
        for (uint32_t i = tid; i < nq; i += nthr) {
          const uint4 a = img4[i];
          const x3_v4u32 v = {a.x, a.y, a.z, a.w};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(16u * i), 0, 0);
        }
        if (tid < (ndw & 3u)) __builtin_amdgcn_raw_buffer_store_b32(img[4u * nq + tid], rs, (int)(16u * nq + 4u * tid), 0, 0);
        if ((total_bytes & 2u) && tid == 0) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)img[ndw], rs, (int)(4u * ndw), 0, 0);
      } else if (mis == 2) {
        if (tid == 0) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)img[0];
        const uint32_t rem = total_bytes - 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + 2, 0, (int)rem, 0x00020000);
        const uint32_t ndw = rem >> 2, nq = ndw >> 2;
        const uint4* img4 = reinterpret_cast<const uint4*>(img);
        for (uint32_t i = tid; i < nq; i += nthr) {
          const uint4 a = img4[i];
          const uint32_t e = img[4u * i + 4u];
          const x3_v4u32 v = {__builtin_amdgcn_alignbit(a.y, a.x, 16), __builtin_amdgcn_alignbit(a.z, a.y, 16),
                              __builtin_amdgcn_alignbit(a.w, a.z, 16), __builtin_amdgcn_alignbit(e, a.w, 16)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(16u * i), 0, 0);
        }
        if (tid < (ndw & 3u))
          __builtin_amdgcn_raw_buffer_store_b32((img[4u * nq + tid] >> 16) | (img[4u * nq + tid + 1u] << 16), rs,
                                                (int)(16u * nq + 4u * tid), 0, 0);
        if ((rem & 2u) && tid == 0) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(img[ndw] >> 16), rs, (int)(4u * ndw), 0, 0);
      } else {
        for (uint32_t i = tid; i < total_bytes; i += nthr) dst[i] = (uint8_t)(img[i >> 2] >> (8 * (i & 3u)));
      }
    }
  };

  // Pacing (as in x3_decode_split_kernel.h): the SQ issues oldest-first, the first workgroup of a CU runs ahead and then
  // polls for the sizes of the third.  Per frame a workgroup compares its frame count with where the clock says it
  // should be -- from the pace of the slowest workgroup of the launch before, below -- and sets its priority: ahead ->
  // lower, behind -> higher.  0.684 -> 0.62 ms on config 3; a target that does not fit pins the priority: as unpaced.
  const unsigned long long pace_t0 = wall_clock64();
  uint32_t pace_k = 0, pace_inv = 0;  // frames done; sixteenths of a frame per 10 ns tick, 16.16 (0: no pacing)
  uint32_t pace_target = 0;
  {
    // pace[0]: ticks per frame of the slowest workgroup of the launch before (P); pace[2]: what that launch aimed at (T).
    // The controller of x3_decode_split_kernel.h with this kernel's numbers: at its best it achieves ~6 % more than it
    // aims at; met to within 2 % -> aim 4 % faster, to within 4 % -> 1.5 % faster; missed by more than 9 % -> back to 1/16
    // under what was achieved.  (Wider "met" bands settle faster and then overshoot every few launches: tools/pace_trace.py.)
    const uint32_t wp = LIST ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)pace[0]);
    const uint32_t wt = LIST ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)pace[2]);
    const uint32_t P = (wp >> X3_DESC_BYTES_BITS) == epoch - 1u ? (wp & X3_DESC_BYTES_MASK) : 0u;
    const uint32_t T = (wt >> X3_DESC_BYTES_BITS) == epoch - 1u ? (wt & X3_DESC_BYTES_MASK) : 0u;
    if (LIST) {
      // (a handful of frames per workgroup: nothing to pace)
    } else if (P < 64u) {
      // nothing to go by (a context's first launch): 6.6 us per frame of 10 000 samples, what config 3 settles on --
      // a guess that does not fit the data pins the priorities, which is the unpaced kernel
      pace_target = (p.spf * 66u) / 1000u;
      if (pace_target >= 64u) pace_inv = (16u << 16) / pace_target;
    } else {
      uint32_t t = P - P / X3E_PACE_DIV;
#if X3E_PACE_CLIMB
      if (T >= 64u) {
        const uint32_t r = (P << 8) / T;  // 256 = met exactly
        t = r < 261u ? T - T / 24u : (r < 266u ? T - T / 64u : (r <= 279u ? T : P - P / X3E_PACE_DIV));  // (met by a wide margin: 4 % faster)
      }
#endif
      pace_target = t;
      pace_inv = (16u << 16) / t;
    }
  }
  for (uint64_t fi = blockIdx.x; fi < (LIST ? (uint64_t)n_list : g.n_frames); fi += G) {
    const uint64_t f = LIST ? (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)dense_list[fi]) : fi;
    const bool more = LIST ? fi + G < n_list : fi + G < g.n_frames;  // this workgroup has another frame behind this one
    if (pace_inv) {
      const uint32_t el = (uint32_t)(wall_clock64() - pace_t0);  // 10 ns ticks
      const int32_t d = (int32_t)(pace_k * 16u) - (int32_t)((el * pace_inv) >> 16);  // sixteenths of a frame
      if (d > X3E_BAND) __builtin_amdgcn_s_setprio(0);
      else if (d > 0) __builtin_amdgcn_s_setprio(1);
      else if (d > -X3E_BAND) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(3);
    }
    ++pace_k;
    const int16_t* src;
    uint32_t n;
    geom_at(clip_f, idx_f, src, n);
    (void)src;
    if (!LIST) geom_advance(clip_f, idx_f);  // now the geometry of f + G
    else if (more) geom_of((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)dense_list[fi + G]), clip_f, idx_f);
    uint32_t* img = img0 + cur * img_dwords;

    // ---- B: one block per lane, in registers.  cnt = samples of this lane's block: 20, 19 in the last block of a
    // full frame, anything down to 1 at the end of a tail frame, 0 behind the frame.
    const uint32_t s0 = 1u + tid * 20u;
    const uint32_t cnt = s0 < n ? (n - s0 < 20u ? n - s0 : 20u) : 0u;
    const uint32_t s_first = W[0] & 0xFFFFu;  // (thread 0: the frame's first sample)
    // plain wave: whole blocks, or the 19-sample last block of a full frame (the pair masks are skipped)
    const bool plain = __all(cnt >= 19u || cnt == 0u);
    // pair j holds block samples r = 2j+1, 2j+2: which of them exist
    auto pair_mask = [&](uint32_t j) __attribute__((always_inline)) -> uint32_t {
      return cnt >= 2u * j + 2u ? 0xFFFFFFFFu : (cnt == 2u * j + 1u ? 0x0000FFFFu : 0u);
    };

    uint32_t S[10];   // emission source per pair of block samples
    uint32_t type = 0, ft = 0, nb = 0, nbits = 0;
    uint32_t amask = 0, orc = 0, qsh = 0, qmask = 0, lbase = 0;  // (code,len) recipe, see x3_encode_common.h
    // (general block lengths) blocks of a lane / lanes of a block / pairs of a lane's (part of a) block
    constexpr uint32_t BU = BL < 20u ? 20u / BL : 1u, BV = BL > 20u ? BL / 20u : 1u, BP = 10u / BU;
    static_assert(BL == 10u || BL == 20u || BL == 40u, "block lengths 10, 20 and 40");
    uint32_t hd[BU];   // per block of the lane: type | header value << 8
    const bool lead = BV == 1u || (lane & (BV - 1u)) == 0u;   // this lane holds the block's first samples: it writes the header
    // samples of block u of this lane
    auto cnt_of = [&](uint32_t u) __attribute__((always_inline)) -> uint32_t {
      if (BU == 1u) return cnt;
      return cnt > u * BL ? (cnt - u * BL < BL ? cnt - u * BL : BL) : 0u;
    };
    if constexpr (BL == 20u) {
      if (cnt) {
        uint32_t mn = 0, mx = 0;
        // (two copies of the loop behind ONE wave-uniform branch: left inside the loop, hipcc turns the test into
        // per-pair selects and every pair of every block pays for the masks)
        auto diffs = [&](auto plain_tag) __attribute__((always_inline)) {
          constexpr bool PLAIN = decltype(plain_tag)::value;
  #pragma unroll
          for (int j = 0; j < 10; ++j) {
            const uint32_t Xj = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // (s[2j+1], s[2j+2])
            S[j] = x3_pk_sub_sat(Xj, W[j]);                                      // (d[2j+1], d[2j+2]), saturated
            if (PLAIN) {
              if (j == 9) S[9] &= cnt == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;        // a 19-sample block has no sample 20
            } else {
              S[j] &= pair_mask(j);
            }
            mn = x3_pk_min_i16(mn, S[j]);
            mx = x3_pk_max_i16(mx, S[j]);
          }
        };
        if (plain) diffs(std::true_type{}); else diffs(std::false_type{});
        const int32_t dmin = min((int32_t)(int16_t)(mn & 0xFFFFu), (int32_t)mn >> 16);
        const int32_t dmax = max((int32_t)(int16_t)(mx & 0xFFFFu), (int32_t)mx >> 16);
        const int32_t maxabs = (-dmin) > dmax ? (-dmin) : dmax;
        if (maxabs <= (int32_t)p.thr[2]) {
          ft = (maxabs > (int32_t)p.thr[0] ? 1u : 0u) + (maxabs > (int32_t)p.thr[1] ? 1u : 0u);
          const uint32_t k = (kpack >> (8u * ft)) & 0xFFu;
          type = k;
          uint32_t sum = 0;
  #pragma unroll
          for (int j = 0; j < 10; ++j) {
            S[j] = x3_pk_shl_b16(S[j], 1) ^ x3_pk_sar_i16(S[j], 15);  // zigzag, per half (0 stays 0)
            sum = x3_pk_add_u16(sum, x3_pk_shr_u16(S[j], k));
          }
          nbits = 2u + cnt * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
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
      }
    } else {
      uint32_t mn[BU], mx[BU];
#pragma unroll
      for (uint32_t u = 0; u < BU; ++u) mn[u] = mx[u] = 0;
      auto diffs = [&](auto plain_tag) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const uint32_t Xj = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // (s[2j+1], s[2j+2])
          S[j] = x3_pk_sub_sat(Xj, W[j]);                                      // (d[2j+1], d[2j+2]), saturated
          if (PLAIN) {
            if (j == 9) S[9] &= cnt == 20u ? 0xFFFFFFFFu : 0x0000FFFFu;
          } else {
            S[j] &= pair_mask(j);
          }
          mn[j / BP] = x3_pk_min_i16(mn[j / BP], S[j]);
          mx[j / BP] = x3_pk_max_i16(mx[j / BP], S[j]);
        }
      };
      if (plain) diffs(std::true_type{}); else diffs(std::false_type{});
      // (a plain wave's lanes behind the frame hold whatever their loads returned: they count as silence)
      const uint32_t on = cnt ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (uint32_t u = 0; u < BU; ++u) {
        const int32_t dmin = min((int32_t)(int16_t)(mn[u] & 0xFFFFu), (int32_t)mn[u] >> 16);
        const int32_t dmax = max((int32_t)(int16_t)(mx[u] & 0xFFFFu), (int32_t)mx[u] >> 16);
        uint32_t maxabs = (uint32_t)((-dmin) > dmax ? (-dmin) : dmax) & on;
        // the block's other lane (quad_perm [1,0,3,2]; every lane of the wave is here)
        if (BV == 2u) maxabs = max(maxabs, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)maxabs, 0xB1, 0xF, 0xF, false));
        const uint32_t cu = cnt_of(u);
        uint32_t ty = 0, hv = 0, nbu = 0;
        if (cu) {
          if (maxabs <= p.thr[2]) {
            const uint32_t f_ = (maxabs > p.thr[0] ? 1u : 0u) + (maxabs > p.thr[1] ? 1u : 0u);
            const uint32_t k = (kpack >> (8u * f_)) & 0xFFu;
            ty = k;
            hv = f_ + 1u;
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t j = u * BP; j < (u + 1u) * BP; ++j) {
              S[j] = x3_pk_shl_b16(S[j], 1) ^ x3_pk_sar_i16(S[j], 15);  // zigzag, per half (0 stays 0)
              sum = x3_pk_add_u16(sum, x3_pk_shr_u16(S[j], k));
            }
            nbu = (lead ? 2u : 0u) + cu * (k + 1u) + (sum & 0xFFFFu) + (sum >> 16);
          } else {
            const uint32_t nbw = 32u - (uint32_t)__clz(maxabs);
            if (nbw >= 15) {
              ty = 5;
              hv = 15;
              nbu = (lead ? 6u : 0u) + 16u * cu;
#pragma unroll
              for (uint32_t j = u * BP; j < (u + 1u) * BP; ++j) S[j] = __builtin_amdgcn_alignbit(W[j + 1], W[j], 16);  // raw samples
            } else {
              ty = 4;
              hv = nbw;
              nbu = (lead ? 6u : 0u) + cu * (nbw + 1u);
            }
          }
        }
        hd[u] = ty | (hv << 8);
        nbits += nbu;
      }
    }

    // ---- C: workgroup exclusive scan of bit lengths
    const uint32_t incl = x3_wave_incl_scan_dpp(nbits);
    if (lane == 63) part[wid] = incl;
    X3_STAMP(0);
    __syncthreads();  // B1: bit-length partials ready
    X3_STAMP(1);
    // the sizes in front of the PREVIOUS frame, due behind this frame's emission: asked for as late as their latency
    // allows -- the other workgroups publish theirs behind their B1, and a word that is read before it is written
    // costs a poll (asked for a whole iteration ahead, 45 % of the frames found a size missing; tools/dbg_stamps_enc2.py)
    if (!LIST && have_prev) desc_load2(prev_f, q0, q1);
    uint32_t wave_base = 0, total = 0;
    {
      const uint4 a = reinterpret_cast<const uint4*>(part)[0], c = reinterpret_cast<const uint4*>(part)[1];
      const uint32_t v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (uint32_t w = 0; w < 8; ++w) {
        wave_base += (w < wid) ? v[w] : 0u;
        total += v[w];
      }
    }
    wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_base);
    total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
    const uint32_t pos = 16u + wave_base + incl - nbits;
    const uint32_t total_bits = 16u + total;
    const uint32_t L = (((total_bits + 7u) >> 3) + 1u) & ~1u;  // word_align (bitpacker.rs:124-132)
    const uint32_t frame_bytes = 20u + L;
    const uint32_t Lw = (L + 3u) >> 2;
    const uint32_t c_dw = (Lw + X3E_CRC_LANES - 1u) / X3E_CRC_LANES;  // payload dwords per lane in the CRC pass: 1..X3_K2_MAXC
    if (c_dw != ktab_c) {  // (workgroup-uniform, rare) the multipliers of this chunk size: used behind B3
      const uint32_t* __restrict__ src_k = xk2 + (size_t)(c_dw - 1u) * X3_K2_DWORDS;
      for (uint32_t i = tid; i < X3_K2_DWORDS; i += nthr) ktab[i] = src_k[i];
      ktab_c = c_dw;
    }
    if (tid == 0) {
      // publish this frame's size as early as possible (the word is its own flag: cdna_hip_programming.md G16, R2)
      if (!LIST) {
        __hip_atomic_store(&desc[f], ready_tag | frame_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (L > X3_DENSE_PAYLOAD_BYTES) ++n_dense;  // (a hint for the host: how dense this call's content is; added once, at the end)
      }
      atomicOr(&img[5], x3_bswap32(s_first << 16));  // <Audio State> (encoder.rs:189)
    }

    // ---- D: emission
    if constexpr (BL == 20u) {
      if (nbits) {
        X3BitEmitter e;
        e.init(img + 5, pos);
        e.put(type <= 3 ? ft + 1u : (type == 4 ? nb : 15u), type <= 3 ? 2u : 6u);
        // (code, len) for BOTH samples of a pair in packed 16-bit arithmetic, the halves combined with
        // SDWA operand selects: 11 VALU per pair in front of the flush test
        const uint32_t qsh2 = qsh * 0x10001u, lbase2 = lbase * 0x10001u;
        const uint32_t amask2 = amask * 0x10001u, orc2 = orc * 0x10001u;
        const uint32_t last_on = cnt == 20 ? 0xFFFFFFFFu : 0x0000FFFFu;  // a 19-sample block has no sample 20
        uint32_t waddr = x3_lds_addr(e.words + e.w);  // byte address of the word the accumulator flushes to
        uint64_t acc = e.acc;
        uint32_t pend = e.cnt;
        auto pairs = [&](auto plain_tag) __attribute__((always_inline)) {
          constexpr bool PLAIN = decltype(plain_tag)::value;
  #pragma unroll
          for (int j = 0; j < 10; ++j) {
            uint32_t Lp = x3_pk_add_u16(x3_pk_lshr_b16(S[j], qsh2) & qmask, lbase2);  // (la, lc)
            uint32_t Cp = (S[j] & amask2) | orc2;                                        // (ca, cc)
            if (PLAIN) {
              if (j == 9) { Lp &= last_on; Cp &= last_on; }
            } else {
              const uint32_t m = pair_mask(j);
              Lp &= m;
              Cp &= m;
            }
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
        };
        if (plain) pairs(std::true_type{}); else pairs(std::false_type{});
        if (pend) x3_lds_or_b32(waddr, x3_bswap32((uint32_t)(acc << (32u - pend))));
      }
    } else if (nbits) {
      X3BitEmitter e;
      e.init(img + 5, pos);
      uint32_t waddr = x3_lds_addr(e.words + e.w);  // byte address of the word the accumulator flushes to
      uint64_t acc = 0;
      uint32_t pend = e.cnt;
      const uint32_t last_on = cnt == 20 ? 0xFFFFFFFFu : 0x0000FFFFu;  // a 19-sample lane has no sample 20
#pragma unroll
      for (uint32_t u = 0; u < BU; ++u) {
        const uint32_t cu = cnt_of(u);
        if (cu) {
          const uint32_t ty = hd[u] & 0xFFu, hv = hd[u] >> 8;
          if (lead) {
            const uint32_t hl = ty <= 3u ? 2u : 6u;
            acc = (acc << hl) | (unsigned long long)hv;
            pend += hl;
            if (pend >= 32u) {
              pend -= 32u;
              x3_lds_or_b32(waddr, x3_bswap32((uint32_t)(acc >> pend)));
              waddr += 4u;
            }
          }
          // the (code, len) recipe of this block (x3_encode_common.h)
          const bool rice = ty <= 3u;
          const uint32_t wdt = ty == 4u ? hv + 1u : 16u;   // BFP: nb + 1 bits; literal: 16
          const uint32_t qsh2 = (rice ? ty : 0u) * 0x10001u, lbase2 = (rice ? ty + 1u : wdt) * 0x10001u;
          const uint32_t amask2 = ((1u << (rice ? ty : wdt)) - 1u) * 0x10001u, orc2 = (rice ? 1u << ty : 0u) * 0x10001u;
          const uint32_t qmask = rice ? 0xFFFFFFFFu : 0u;
          auto pairs = [&](auto plain_tag) __attribute__((always_inline)) {
            constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
            for (uint32_t j = u * BP; j < (u + 1u) * BP; ++j) {
              uint32_t Lp = x3_pk_add_u16(x3_pk_lshr_b16(S[j], qsh2) & qmask, lbase2);  // (la, lc)
              uint32_t Cp = (S[j] & amask2) | orc2;                                        // (ca, cc)
              if (PLAIN) {
                if (j == 9) { Lp &= last_on; Cp &= last_on; }
              } else {
                const uint32_t m = pair_mask(j);
                Lp &= m;
                Cp &= m;
              }
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
          };
          if (plain) pairs(std::true_type{}); else pairs(std::false_type{});
          if (!LIST) atomicAdd(&part[32 + ty], cu);   // statistics (encoder.rs:199): stats[type] += block.len()
        }
      }
      if (pend) x3_lds_or_b32(waddr, x3_bswap32((uint32_t)(acc << (32u - pend))));
    }
    // statistics (encoder.rs:199): stats[type] += block.len(); one LDS atomic per lane, summed over
    // all frames of this workgroup and flushed once at the end
    if (BL == 20u && !LIST && cnt) atomicAdd(&part[32 + type], cnt);
    X3_STAMP(2);
    // the sizes in front of the previous frame were requested a whole iteration ago
    if (!LIST && have_prev) settle();
    // Nothing of this wave is in flight here but stores.  Said with the BUILTIN, so that hipcc's wait-count pass
    // clears its scoreboard: it would otherwise protect registers that "may" still be load targets (the polls of
    // settle(), the size words across the loop's back edge) with vmcnt(0) waits all over the CRC pass and the
    // copy-out -- which would then wait for the sample loads issued just below (loads return in order).
    x3_dma_wait();
    // the next frame of this workgroup: its samples fly under the CRC pass, the copy-out and the barriers (the
    // registers are free: the block has been emitted)
    if (more) {
      const int16_t* nsrc;
      uint32_t nn;
      geom_at(clip_f, idx_f, nsrc, nn);
      load_block(nsrc, nn);
    }
    X3_STAMP(3);
    __syncthreads();  // B3: emission complete; the size sums of the previous frame are in LDS
    X3_STAMP(4);

    // ---- F: the PREVIOUS frame goes out now: its offset needed every predecessor's size, and that wait
    // overlapped this frame's analysis and emission
    if (have_prev) {
      const uint64_t off = resolve();
      if (X3E_CRC_WAVES >= 8u) copy_out(img0 + (cur ^ 1u) * img_dwords, off, prev_bytes, tid, nthr);
      else if (wid >= X3E_CRC_WAVES) copy_out(img0 + (cur ^ 1u) * img_dwords, off, prev_bytes, tid - X3E_CRC_LANES, nthr - X3E_CRC_LANES);
    }
    X3_STAMP(5);
    // ---- E: payload CRC-16 as a segmented reduction (see the multiplier tables above): lane chunks are right-
    // aligned in the payload, c_dw dwords each
    uint32_t crc = 0;
    const int32_t j0 = (int32_t)(tid * c_dw) - (int32_t)(X3E_CRC_LANES * c_dw - Lw);
    if (wid < X3E_CRC_WAVES && __any(j0 + (int32_t)c_dw > 0)) {  // (the first waves of a short payload hold nothing)
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
        // times this lane's x^(32*c*(63-lane)): one 16-entry table per nibble (KN above)
        const uint32_t rowb = x3_lds_addr(ktab) + lane * (X3_K2_ROW * 4u);
        crc = (uint32_t)x3_lds_read_u16(rowb + ((crc & 15u) << 1), 0u) ^
              (uint32_t)x3_lds_read_u16(rowb + (((crc >> 4) & 15u) << 1), 32u) ^
              (uint32_t)x3_lds_read_u16(rowb + (((crc >> 8) & 15u) << 1), 64u) ^
              (uint32_t)x3_lds_read_u16(rowb + (((crc >> 12) & 15u) << 1), 96u);
      }
      crc = x3_wave_xor_to_lane63_dpp(crc);
    }
    if (lane == 63 && wid < X3E_CRC_WAVES) part[16 + wid] = crc;
    // header CRC (encoder.rs:153-154): it needs only the sample count and the payload length.  The state behind
    // the constant bytes "x3", id, id is a constant; the (samples, payload_len) word and the eight zero time
    // bytes go through the slicing tables (a zero word is two look-ups).  Lane 7 of wave 0 keeps it for the header.
    uint32_t hdr_crc = 0;
    if (tid == 7) {
      constexpr uint32_t K4 = x3_crc16_const4(0x78u, 0x33u, 0x01u, 0x01u);
      const uint32_t m = (((n & 0xFFFFu) << 16) | (L & 0xFFFFu)) ^ (K4 << 16);
      uint32_t hc = (uint32_t)crc_tab[768u + (m >> 24)] ^ (uint32_t)crc_tab[512u + ((m >> 16) & 0xFFu)] ^
                    (uint32_t)crc_tab[256u + ((m >> 8) & 0xFFu)] ^ (uint32_t)crc_tab[m & 0xFFu];
      hc = (uint32_t)crc_tab[768u + (hc >> 8)] ^ (uint32_t)crc_tab[512u + (hc & 0xFFu)];
      hc = (uint32_t)crc_tab[768u + (hc >> 8)] ^ (uint32_t)crc_tab[512u + (hc & 0xFFu)];
      hdr_crc = hc;
    }
    X3_STAMP(6);
    __syncthreads();  // B4: CRC partials in LDS; the previous frame's image has been read by every wave
    X3_STAMP(7);

    // this frame's header (encoder.rs:122-162): "x3", id, id, samples, payload_len, 8 zero time bytes, header crc
    // over bytes 0..16, payload crc; audio frames use id 1.  The first lanes of wave 0 weigh the CRC waves' partials
    // (KA), lane 7 gathers them and writes the five words, while the other waves clear the old image.
    if (wid == 0) {
      uint32_t v = 0;
      if (lane < X3E_CRC_WAVES) {
        const uint32_t pw = part[16 + lane];
        const uint4* kp = reinterpret_cast<const uint4*>(ktab + X3_K2_KA + (8u - X3E_CRC_WAVES + lane) * 16u);
        const uint4 kq0 = kp[0], kq1 = kp[1], kq2 = kp[2], kq3 = kp[3];
        const uint32_t kk[16] = {kq0.x, kq0.y, kq0.z, kq0.w, kq1.x, kq1.y, kq1.z, kq1.w,
                                 kq2.x, kq2.y, kq2.z, kq2.w, kq3.x, kq3.y, kq3.z, kq3.w};
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) v ^= (0u - ((pw >> bit) & 1u)) & kk[bit];
      }
      v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
      v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
      v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4 -> lane 7: all eight
      if (lane == 7) {
        if (L & 2u) v = x3_gf_mul_const<X3_XINV16_C>(v);  // undo the 2 virtual pad-to-4 bytes
        img[0] = x3_bswap32(0x78330101u);
        img[1] = x3_bswap32(((n & 0xFFFFu) << 16) | (L & 0xFFFFu));
        img[2] = 0;
        img[3] = 0;
        img[4] = x3_bswap32((hdr_crc << 16) | (v & 0xFFFFu));
      }
    }
    // clear what the previous frame used, ready for the frame after this one (the next B1 separates this from
    // the emission into it)
    if (have_prev) {
      uint4* z4 = reinterpret_cast<uint4*>(img0 + (cur ^ 1u) * img_dwords);
      const uint32_t nz = (((prev_bytes + 3u) >> 2) + 3u) >> 2;
      const uint4 zero = make_uint4(0, 0, 0, 0);
      for (uint32_t i = tid; i < nz; i += nthr) z4[i] = zero;
    }
    prev_f = f;
    prev_bytes = frame_bytes;
    have_prev = true;
    cur ^= 1u;
    X3_STAMP(0);
  }
  // ---- the workgroup's last frame
  if (!LIST && have_prev) {
    desc_load2(prev_f, q0, q1);
    settle();
  }
  __syncthreads();  // its size sums, its header
  if (have_prev) {
    const uint64_t off = resolve();
    copy_out(img0 + (cur ^ 1u) * img_dwords, off, prev_bytes, tid, nthr);
  }
#ifdef X3_DBG_STAMPS
  if (lane == 0 && blockIdx.x < 400) {
    for (int k = 0; k < 8; ++k) x3_dbg[(blockIdx.x * 8 + wid) * 8 + k] = dbg_acc[k];
    x3_dbg[8 * 8 * 400 + (blockIdx.x * 8 + wid) * 2] = dbg_spins;
    x3_dbg[8 * 8 * 400 + (blockIdx.x * 8 + wid) * 2 + 1] = dbg_late;
  }
#endif
  if (!LIST && tid == 0 && pace_k >= 16u) {  // this workgroup's pace, for the next launch
    uint64_t t = (wall_clock64() - pace_t0) / pace_k;
    if (t > X3_DESC_BYTES_MASK) t = X3_DESC_BYTES_MASK;
    atomicMax(pace, (epoch << X3_DESC_BYTES_BITS) | (uint32_t)t);
    if (blockIdx.x == 0) pace[2] = (epoch << X3_DESC_BYTES_BITS) | pace_target;
  }
  if (!LIST && tid == 0 && n_dense) atomicAdd(dense_count, n_dense);
  if (!LIST && tid < 6) {
    const uint32_t v = part[32 + tid];  // < 2^32: at most ~270 frames x 10 000 samples per workgroup
    if (v) atomicAdd(&stats[tid], (unsigned long long)v);
  }
}
