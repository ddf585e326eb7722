// x3_decode.hip -- decoder, frame-check and index kernels and the x3_decode* / x3_index_dev entry points, the stream walks,
// the BitReader / BitPacker handles (C ABI: include/x3hip.h; units: x3_internal.h).
#include <chrono>
#include "x3_internal.h"
#include "x3_decode_kernel.h"
#include "x3_decode_split_kernel.h"
#include "x3_decode_blocks_kernel.h"
#include "x3_index_kernels.h"
#include "x3_decode_mc_kernel.h"

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// wav_off_aligned: the caller knows that every d_wav_offsets[f] is a multiple of FOUR samples (the three-wave decoder
// moves rows in 8- or 16-byte pieces); without that knowledge caller-supplied offsets go to the single-wave kernels
int decode_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                           uint64_t F, const x3_batch* batch, const uint64_t* d_wav_offsets, const x3_params* p,
                           int16_t* d_wav, uint64_t wav_cap, int32_t* d_status, bool wav_off_aligned,
                           bool bl0,    // bl0: the caller's block_len is 0 and p carries 1 (x3_decode_merge_kernel)
                           const X3SegSpec* seg, const unsigned long long* d_nf) {
  if (reinterpret_cast<uintptr_t>(d_x3) & 3u) return X3_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d_wav) & 1u) return X3_ERR_BAD_ARG;
  if (F == 0 || F > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  const uint64_t spf = spf_of(p);
  X3DevParams dp;
  int rc = derive(p, spf > 0xFFFFFFFFull ? 0 : spf, &dp);
  if (rc) return rc;
  X3Geom g{0, 0, 1, F};
  if (!d_wav_offsets) {
    if (!batch || spf == 0) return X3_ERR_BAD_ARG;
    const uint64_t fpc = (batch->n_per_clip + spf - 1) / spf;
    if (fpc == 0 || fpc > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
    g.n_per_clip = batch->n_per_clip;
    g.clip_stride = batch->clip_stride;
    g.fpc = (uint32_t)fpc;
  }
  c->last_seg_stretches = 0;
  if ((rc = ensure(c, c->dec_meta, F * sizeof(X3FrameMeta)))) return rc;
  if (!d_status) {
    if ((rc = ensure(c, c->dec_status, F * sizeof(int32_t)))) return rc;
    d_status = (int32_t*)c->dec_status.p;
  }
  if ((rc = ensure(c, c->dec_cstatus, F * sizeof(int32_t)))) return rc;
  // fork: header + payload-CRC pass on the side stream, decoder on the main stream (independent;
  // the decoder's one-wave-per-SIMD dependency chains leave the CUs mostly idle)
#ifdef X3_PROFILING
  hipStream_t check_stream = c->opt.check_serial ? c->stream : c->stream2;
  const bool no_check = c->opt.no_check != 0;
#else
  // (option check_main: the two passes swap streams -- the one on the side stream starts a cross-queue event later)
  hipStream_t check_stream = c->opt.check_main ? c->stream : c->stream2;
  const bool no_check = false;
#endif
  hipStream_t dec_stream = check_stream == c->stream2 ? c->stream : c->stream2;
  const uint64_t check_wgs_per_cu = (uint64_t)c->opt.check_wgs;
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  // The check pass and the decoder are independent and run side by side; the decoder is enqueued first.  (Neither the
  // order nor the side stream's queue priority selects between the decoder's timing modes -- 0.81 / 0.87 / 0.94 ms
  // per process on one box, 0.98 on another with no check kernel at all: measured, tools/dbg_modes.sh, DESIGN.md.)
  auto launch_check = [&]() -> int {
    if (no_check) {
      // profiling builds only: time the decoder without the check pass beside it (payload CRCs are NOT verified)
      HIPCHK(c, hipMemsetAsync(c->dec_cstatus.p, 0, F * sizeof(int32_t), c->stream2));

    } else {
      TimerScope ts(c, 4, check_stream, true);
      const uint64_t check_grid = std::min<uint64_t>((F + 3) / 4, (uint64_t)c->n_cus * check_wgs_per_cu);
      X3_LAUNCH_TIMED(ts, x3_frame_check_kernel, dim3((unsigned)check_grid), dim3(256), 0, check_stream,
                         reinterpret_cast<const uint32_t*>(d_x3), x3_len, d_frame_offsets, F,
                         (const uint16_t*)c->d_xinv8, (const uint16_t*)c->d_chktab, (const uint32_t*)c->d_kx64,
                         (int32_t*)c->dec_cstatus.p, reinterpret_cast<unsigned long long*>(c->d_summary), 1u, d_nf);
    }
    if (check_stream == c->stream2) HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
    return X3_OK;
  };
  if (c->opt.check_first && (rc = launch_check())) return rc;
  {
    // the branch-free kernel needs every valid Rice codeword (zeros + terminator + sub-code) to fit 32 bits
    bool fast = true;
    const uint32_t widths[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
      const uint32_t level = k == 0 ? 1u : (1u << dp.k[k]);
      fast = fast && (dp.inv_len[k] / level + 1 + widths[k] <= 32);
    }
    // two waves per group of 64 frames (parser + valuer) when the geometry is the plain one.  Its parser hands
    // over i = (z << k) + r with r the k bits behind the terminating one, which equals the reference's
    // r' + level * (n - 1) (decoder.rs:186, r' = the hard-wired 2 / 4 bits INCLUDING the one) only when the
    // code of ftype 2 has one sub-bit and that of ftype 3 three -- the default codes; the single-wave kernels
    // follow the reference's formula literally and take every other code set.
    const bool split = fast && dp.block_len == X3S_BL && dp.k[1] == 1u && dp.k[2] == 3u &&
                       (!d_wav_offsets || wav_off_aligned) && !c->force_single_wave_decode &&
                       !c->opt.decode_single && (reinterpret_cast<uintptr_t>(d_wav) & 7u) == 0 &&
                       (d_wav_offsets || ((dp.spf % 4u) == 0 && (g.fpc * (uint64_t)1 >= g.n_frames || (g.clip_stride % 4u) == 0)));
    // (rows on 16-byte boundaries: whole lines in 16-byte pieces; on 8-byte boundaries -- an output pointer, a clip stride
    // or a frame length of 4 (mod 8) samples --: the same lines in 8-byte pieces, x3_decode_split_kernel.h flush_rows)
    // (X3HIP_DECODE_DYN_LDS: extra LDS per decoder group = fewer groups per CU; an occupancy experiment)
    const size_t dyn_lds = (size_t)c->opt.dyn_lds;
    // Round 6: a BLOCK per lane (x3_decode_blocks_kernel.h).  Block length 20: option "decode_blocks" -- wherever the three-wave
    // kernel would run frame by frame (that one keeps the segment index, decoding by it and recording it, and stays the default:
    // the block-per-lane kernel walks every frame twice and is slower on config 3, profiles/r6/decoder_blocks_kernel.txt).
    // Block lengths 10 and 40 (the default codes, rows on 8-byte boundaries): the block-per-lane kernel IS the default -- the
    // three-wave kernel is written for blocks of 20, and the single-wave kernels such streams took until round 6 are one serial
    // chain per frame with nothing beside it (1.8 / 1.1 ms at config 3's size).
    const bool by_seg = seg && seg->mode && seg->d_index && seg->seg_blocks;
    const bool blocks_geom = fast && dp.k[1] == 1u && dp.k[2] == 3u && (!d_wav_offsets || wav_off_aligned) &&
                             !c->force_single_wave_decode && !c->opt.decode_single;
    // (rows: a frame's samples begin where the layout says -- multiples of spf samples, of the clip stride, or the caller's
    // offsets -- so the row alignment follows from the pointer's and theirs)
    auto rows_on = [&](uint64_t bytes) {
      const uint64_t samples_unit = bytes / 2;
      if (reinterpret_cast<uintptr_t>(d_wav) % bytes) return false;
      if (d_wav_offsets) return bytes <= 8;            // (the caller vouches for multiples of four samples, no more)
      return (dp.spf % samples_unit) == 0 && (g.fpc * (uint64_t)1 >= g.n_frames || (g.clip_stride % samples_unit) == 0);
    };
    const bool blocks40 = blocks_geom && dp.block_len == 40u && rows_on(8) && !c->opt.decode_blocks_off;
    const bool blocks10 = blocks_geom && dp.block_len == 10u && rows_on(8) && !c->opt.decode_blocks_off;
    if (d_nf && !split) return X3_ERR_BAD_ARG;   // (only the block_len 20 decoders take the frame count from device memory)
    TimerScope ts(c, 1, dec_stream, split || blocks40 || blocks10);   // (events on the dispatch packet of the kernels launched with X3_LAUNCH_TIMED; the rarer single-wave kernels below: bracketed)
    if ((split && !by_seg && c->opt.decode_blocks) || blocks40 || blocks10) {
      // as many groups as give every CU the same number (five are resident per CU: the walkers of ALL groups must run
      // side by side, a frame's walk is the kernel's critical path): config 3 is 1 280 groups of 54 frames
      const uint64_t slots = (uint64_t)c->n_cus * 5;
      uint64_t fpg = 1;
      if (F >= slots) {
        const uint64_t rounds = (F + 64 * slots - 1) / (64 * slots);
        fpg = (F + rounds * slots - 1) / (rounds * slots);
      }
      const uint64_t groups = (F + fpg - 1) / fpg;
      if ((c->dec_epoch & 0xFFFu) == 0u) {
        HIPCHK(c, hipMemsetAsync(c->d_pace, 0, 16, dec_stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + 8, 0, 8, dec_stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + X3_LOG_BASE, 0, X3_LOG_ENTRIES * X3_LOG_WORDS * sizeof(uint32_t), dec_stream));
        ++c->dec_epoch;
      }
      c->last_decode_kernel = 3;
#define X3B_LAUNCH(UNIT, UPB)                                                                                             \
      X3_LAUNCH_TIMED(ts, (x3_decode_blocks_kernel<UNIT, UPB>), dim3((unsigned)groups), dim3(64 * X3B_WAVES), dyn_lds, dec_stream, \
                      d_x3, x3_len, d_frame_offsets, F, (uint32_t)fpg, g, d_wav_offsets, dp, d_wav, wav_cap, d_status,     \
                      (X3FrameMeta*)c->dec_meta.p, c->d_pace, c->dec_epoch & 0xFFFu, d_nf)
      if (blocks40) X3B_LAUNCH(20u, 2u);
      else if (blocks10) X3B_LAUNCH(10u, 1u);
      else X3B_LAUNCH(20u, 1u);
#undef X3B_LAUNCH
      ++c->dec_epoch;
    }
    else if (split) {
      c->last_decode_kernel = 2;
      // the pace word's 12-bit epoch: launches 1, 2, ... 4095, then the word starts over
      if ((c->dec_epoch & 0xFFFu) == 0u) {
        HIPCHK(c, hipMemsetAsync(c->d_pace, 0, 16, dec_stream));  // (achieved and aimed-at, one word per launch parity)
        HIPCHK(c, hipMemsetAsync(c->d_pace + 8, 0, 8, dec_stream));   // (the launch shapes)
        HIPCHK(c, hipMemsetAsync(c->d_pace + X3_LOG_BASE, 0, X3_LOG_ENTRIES * X3_LOG_WORDS * sizeof(uint32_t), dec_stream));  // (its atomicMax entries carry the epoch too)
        ++c->dec_epoch;
      }
      // The segment index (x3_decode_split_kernel.h, "STRETCHES"): decode by it -- a lane per stretch of `seg_blocks` blocks,
      // nseg times the groups -- or record it while decoding frame by frame.  Frames of at most one stretch need none.
      X3SegArgs sg{nullptr, nullptr, 0u, 1u, 0u, 1u};
      uint64_t groups = (F + 63) / 64;
      if (seg && seg->mode && seg->d_index && seg->seg_blocks) {
        // (the index's pitch follows the PARAMETERS -- blocks of a full frame -- not the call: x3_seg_index_entries, the
        // encoder and this launch must agree on it whatever the clips' lengths are)
        const uint64_t bpf = (dp.spf + X3S_BL - 1) / X3S_BL;
        const uint64_t nidx = (bpf + seg->seg_blocks - 1) / seg->seg_blocks;   // stretches the index can tell apart
        if (nidx >= 2) {
          sg.pitch = (uint32_t)(nidx - 1);
          if (seg->mode == 1) {
            // As many stretches as fill the chip and no more: a launch whose groups fill it anyway is faster frame by frame
            // (a stretch pays a group's start: header, ring fill, the flusher's tables; 1 080 groups: 0.70 against 0.78 ms,
            // profiles/r5/segments_split_kernel.txt), a stream of 42 groups five times faster by stretches.
            const uint64_t want = c->opt.seg_stretches > 0 ? (uint64_t)c->opt.seg_stretches
                                  : (groups * 2 >= (uint64_t)c->n_cus * 5 ? 1 : ((uint64_t)c->n_cus * 4 + groups - 1) / groups);
            const uint64_t mul = (nidx + std::min<uint64_t>(want, nidx) - 1) / std::min<uint64_t>(want, nidx);
            const uint64_t nseg = (nidx + mul - 1) / mul;
            if (nseg >= 2 && groups * nseg <= 0x7FFFFFFFull) {
              sg.in = reinterpret_cast<const uint2*>(seg->d_index);
              sg.mul = (uint32_t)mul;
              sg.sb = (uint32_t)(seg->seg_blocks * mul);
              sg.nseg = (uint32_t)nseg;
              groups *= nseg;
              HIPCHK(c, hipMemsetAsync(d_status, 0, F * sizeof(int32_t), dec_stream));  // (the stretches of a frame add to its status)
            }
          } else {
            sg.out = reinterpret_cast<uint2*>(seg->d_index);
            sg.sb = seg->seg_blocks;
            sg.nseg = (uint32_t)nidx;
            HIPCHK(c, hipMemsetAsync(seg->d_index, 0, (1 + F * (nidx - 1)) * sizeof(uint2), dec_stream));   // (no entry: not valid)
          }
        }
      }
      c->last_seg_stretches = sg.in ? (int)sg.nseg : (sg.out ? -1 : 0);
      X3_LAUNCH_TIMED(ts, x3_decode_split_kernel, dim3((unsigned)groups), dim3(64 * X3S_WAVES),
                         dyn_lds, dec_stream, d_x3, x3_len,
                         d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status, (X3FrameMeta*)c->dec_meta.p,
                         c->d_pace, c->dec_epoch & 0xFFFu, sg, d_nf);
      ++c->dec_epoch;
    }
    else if (fast) {
      c->last_decode_kernel = 1;
      hipLaunchKernelGGL(x3_decode_fast_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, dec_stream, d_x3, x3_len,
                         d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status,
                         (X3FrameMeta*)c->dec_meta.p);
    } else {
      c->last_decode_kernel = 0;
      hipLaunchKernelGGL((x3_decode_lanes_kernel<false, 64>), dim3((unsigned)((F + 63) / 64)), dim3(64), 0, dec_stream,
                         d_x3, x3_len, d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status,
                         (X3FrameMeta*)c->dec_meta.p);
    }
  }
  if (dec_stream == c->stream2) HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
  if (!c->opt.check_first && (rc = launch_check())) return rc;
  // join, then merge the two status arrays and summarise
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  if (no_check) {  // otherwise the check kernel's first thread does this
    X3DecodeSummary init;
    init.first_bad = F;
    init.samples_before = 0;
    init.first_bad_status = 0;
    init.pad = 0;
    *c->h_summary_init = init;
    HIPCHK(c, hipMemcpyAsync(c->d_summary, c->h_summary_init, sizeof init, hipMemcpyHostToDevice, c->stream));
  }
  hipLaunchKernelGGL(x3_decode_merge_kernel, dim3((unsigned)std::min<uint64_t>((F + 255) / 256, 64)), dim3(256), 0, c->stream,
                     (const int32_t*)c->dec_cstatus.p, d_status, (const X3FrameMeta*)c->dec_meta.p, F, c->d_summary,
                     d_x3, d_frame_offsets, g, d_wav_offsets, dp, d_wav, bl0 ? 1u : 0u, d_nf);
  c->dec_status_ptr = d_status;
  HIPCHK(c, hipGetLastError());
  c->decode_pending = true;
  c->dec_frames = F;
  return X3_OK;
}

extern "C" int x3_decode_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                             uint64_t n_frames, const x3_batch* batch, const uint64_t* d_wav_offsets,
                             const x3_params* p, int16_t* d_wav, uint64_t wav_cap, int32_t* d_status) {
  if (!c || !d_x3 || !d_frame_offsets || !p || !d_wav) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  // (caller-supplied sample offsets: the three-wave decoder needs rows on 8-byte boundaries, and only the caller can know
  // that without a pass over the offsets -- option "wav_offsets_x4")
  return decode_dev_impl(c, d_x3, x3_len, d_frame_offsets, n_frames, batch, d_wav_offsets, p, d_wav, wav_cap, d_status,
                         d_wav_offsets != nullptr && c->opt.wav_offsets_x4 != 0);
}

// x3_decode_dev with the SEGMENT INDEX (include/x3hip.h): record = 0 decodes by it, record = 1 decodes frame by frame and
// leaves it in d_seg_index for the next decode of the same stream
extern "C" uint64_t x3_seg_index_entries(uint64_t n_frames, const x3_params* p, uint32_t seg_blocks) {
  if (!p || seg_blocks == 0 || p->blocks_per_frame == 0) return 0;
  const uint64_t nseg = ((uint64_t)p->blocks_per_frame + seg_blocks - 1) / seg_blocks;
  return nseg >= 2 ? 1 + n_frames * (nseg - 1) : 0;   // (a header word, then nseg - 1 entries per frame)
}

extern "C" int x3_decode_dev_seg(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                                 uint64_t n_frames, const x3_batch* batch, const uint64_t* d_wav_offsets,
                                 const x3_params* p, int16_t* d_wav, uint64_t wav_cap, int32_t* d_status,
                                 uint64_t* d_seg_index, uint32_t seg_blocks, int record) {
  if (!c || !d_x3 || !d_frame_offsets || !p || !d_wav) return X3_ERR_BAD_ARG;
  if (d_seg_index && (seg_blocks == 0 || (seg_blocks & 3u) || seg_blocks > 3200u || (reinterpret_cast<uintptr_t>(d_seg_index) & 7u)))
    return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const X3SegSpec seg{d_seg_index, seg_blocks, d_seg_index ? (record ? 2 : 1) : 0};
  return decode_dev_impl(c, d_x3, x3_len, d_frame_offsets, n_frames, batch, d_wav_offsets, p, d_wav, wav_cap, d_status,
                         d_wav_offsets != nullptr && c->opt.wav_offsets_x4 != 0, false, &seg);
}

extern "C" int x3_decode_result(x3_ctx* c, uint64_t* first_bad, int* first_bad_status, uint64_t* samples_before) {
  if (!c || !c->decode_pending) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(X3DecodeSummary), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->decode_pending = false;
  if (c->h_summary->first_bad < c->dec_frames) {
    // rare: a frame is bad -- its status and the samples of the good frames before it
    hipLaunchKernelGGL(x3_decode_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, (const int32_t*)c->dec_status_ptr,
                       (const X3FrameMeta*)c->dec_meta.p, c->dec_frames, c->d_summary);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(X3DecodeSummary), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (first_bad) *first_bad = c->h_summary->first_bad;
  if (first_bad_status) *first_bad_status = c->h_summary->first_bad_status;
  if (samples_before) *samples_before = c->h_summary->samples_before;
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// GPU-side frame index of a device-resident stream (x3_index_kernels.h)
// ------------------------------------------------------------------------------------------------
// The fast path's five launches (x3_index_kernels.h), nothing waited for: the summary -- frame and sample count, how the
// walk ends, pad2 = "not one clean chain: take the general walk" -- stays in *d_sum_out on the device.
static int index_fast_launch(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, uint32_t bl0, uint64_t wav_cap,
                             uint64_t max_frames, uint64_t* d_frame_offsets, uint64_t* d_wav_offsets,
                             X3IndexSummary** d_sum_out) {
  const uint32_t* xw = reinterpret_cast<const uint32_t*>(d_x3);
  int rc;
  if ((rc = ensure(c, c->idx_sum, 256))) return rc;
  unsigned int* d_count = reinterpret_cast<unsigned int*>((char*)c->idx_sum.p + 128);
  X3IndexSummary* d_sum = reinterpret_cast<X3IndexSummary*>(c->idx_sum.p);
  const uint64_t chunks = (len + 15) >> 4;
  const unsigned grid = (unsigned)std::min<uint64_t>((chunks + 255) / 256, (uint64_t)c->n_cus * 16);
  *d_sum_out = d_sum;
  if (!grid) return X3_ERR_BAD_ARG;
  {
    const size_t G = grid;
    if ((rc = ensure(c, c->idx_wg, G * X3I_WG_CANDS * sizeof(X3Cand)))) return rc;
    if ((rc = ensure(c, c->idx_sorted, G * X3I_WG_CANDS * sizeof(X3Cand)))) return rc;
    if ((rc = ensure(c, c->idx_scan, G * 32))) return rc;   // count u32 | base u32 | samp u64 | sbase u64 per span
    unsigned int* cnt = reinterpret_cast<unsigned int*>(c->idx_scan.p);
    uint32_t* base = reinterpret_cast<uint32_t*>(cnt + G);
    unsigned long long* samp = reinterpret_cast<unsigned long long*>(base + G);
    unsigned long long* sbase = samp + G;
    hipLaunchKernelGGL(x3_index_init_kernel, dim3(1), dim3(64), 0, c->stream, d_sum, d_count);
    hipLaunchKernelGGL(x3_index_candidates_kernel<true>, dim3(grid), dim3(256), 0, c->stream, xw, len, len + phantom, bl0,
                       (X3Cand*)c->idx_wg.p, 0u, cnt, samp, &d_sum->pad2);
    hipLaunchKernelGGL(x3_index_chain_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned int*)cnt,
                       (const unsigned long long*)samp, (uint32_t)G, base, sbase, d_sum);
    hipLaunchKernelGGL(x3_index_link_kernel, dim3(grid), dim3(64), 0, c->stream, (const X3Cand*)c->idx_wg.p,
                       (const unsigned int*)cnt, (uint32_t)G, (const uint32_t*)base, (const unsigned long long*)sbase,
                       (X3Cand*)c->idx_sorted.p, (uint32_t)std::min<uint64_t>(G * X3I_WG_CANDS, 0xFFFFFFFFull),
                       (unsigned long long)max_frames, (unsigned long long)wav_cap, (unsigned long long*)d_frame_offsets,
                       (unsigned long long*)d_wav_offsets, d_sum);
    hipLaunchKernelGGL(x3_index_finalize_kernel, dim3(1), dim3(64), 0, c->stream, xw, len, len + phantom, bl0,
                       (const X3Cand*)c->idx_sorted.p, (const unsigned long long*)d_wav_offsets, d_sum);
    HIPCHK(c, hipGetLastError());
  }
  return X3_OK;
}

static int index_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, uint32_t bl0, uint64_t wav_cap,
                          uint64_t max_frames, uint64_t* d_frame_offsets, uint64_t* d_wav_offsets,
                          X3IndexSummary* result) {
  if (reinterpret_cast<uintptr_t>(d_x3) & 3u) return X3_ERR_BAD_ARG;
  const uint32_t* xw = reinterpret_cast<const uint32_t*>(d_x3);
  int rc;
  if ((rc = ensure(c, c->idx_sum, 256))) return rc;
  unsigned int* d_count = reinterpret_cast<unsigned int*>((char*)c->idx_sum.p + 128);
  X3IndexSummary* d_sum = reinterpret_cast<X3IndexSummary*>(c->idx_sum.p);
  const uint64_t chunks = (len + 15) >> 4;
  const unsigned grid = (unsigned)std::min<uint64_t>((chunks + 255) / 256, (uint64_t)c->n_cus * 16);
  // ---- the fast path (round 4): a stream that is ONE CLEAN CHAIN -- what an encoder writes -- needs no hash table and
  // no pointer doubling.  The scanning workgroups keep their candidates in stream order, two scans number them, and one
  // kernel checks all at once that the first frame sits at offset 0 and that every frame ends where the next begins
  // (x3_index_kernels.h).  Five launches and one trip to the host where the general walk below takes thirteen and two;
  // anything else -- a false candidate inside a payload, damage, a frame the walk does not step over -- sets pad2, and
  // the general walk takes the stream as before.
  if (grid && !c->opt.index_no_fast) {
    X3IndexSummary* ds = nullptr;
    if ((rc = index_fast_launch(c, d_x3, len, phantom, bl0, wav_cap, max_frames, d_frame_offsets, d_wav_offsets, &ds))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_summary_init, d_sum, sizeof *result, hipMemcpyDeviceToHost, c->stream));  // (pinned)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(result, c->h_summary_init, sizeof *result);
    if (result->pad) {
      c->last_error = "x3_index_dev: more frames in the stream than max_frames";
      return X3_ERR_BAD_ARG;
    }
    if (!result->pad2) {
      ++c->index_fast;
      return X3_OK;
    }
  }
  ++c->index_general;
  hipLaunchKernelGGL(x3_index_init_kernel, dim3(1), dim3(64), 0, c->stream, d_sum, d_count);
  // ONE pass over the stream: the candidates go into a buffer sized for a frame every 256 bytes (the context keeps
  // it; typical streams hold one every few kilobytes); only a stream with more than that is scanned a second time.
  unsigned int n_cand = 0;
  if (grid) {
    const uint64_t guess = std::max<uint64_t>(4096, len / 256 + 1024);
    if ((rc = ensure(c, c->idx_cand, (size_t)std::min<uint64_t>(guess, 0x7FFFFFFFull) * sizeof(X3Cand)))) return rc;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(c->idx_cand.cap / sizeof(X3Cand), 0x7FFFFFFFull);
    hipLaunchKernelGGL(x3_index_candidates_kernel<false>, dim3(grid), dim3(256), 0, c->stream, xw, len, len + phantom, bl0,
                       (X3Cand*)c->idx_cand.p, cap, d_count, (unsigned long long*)nullptr, (uint32_t*)nullptr);
    HIPCHK(c, hipMemcpyAsync(c->h_crc, d_count, sizeof n_cand, hipMemcpyDeviceToHost, c->stream));  // (pinned)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(&n_cand, c->h_crc, sizeof n_cand);
    if (n_cand > cap) {  // rare: denser than one frame per 256 bytes
      if ((rc = ensure(c, c->idx_cand, (size_t)n_cand * sizeof(X3Cand)))) return rc;
      HIPCHK(c, hipMemsetAsync(d_count, 0, sizeof(unsigned int), c->stream));
      hipLaunchKernelGGL(x3_index_candidates_kernel<false>, dim3(grid), dim3(256), 0, c->stream, xw, len, len + phantom, bl0,
                         (X3Cand*)c->idx_cand.p, n_cand, d_count, (unsigned long long*)nullptr, (uint32_t*)nullptr);
    }
  }
  if (n_cand) {
    uint32_t tsize = 1024;
    while (tsize < 2u * n_cand && tsize < 0x80000000u) tsize <<= 1;
    uint32_t levels = 1;
    while ((1ull << (levels - 1)) < n_cand) ++levels;  // the top level spans 2^(levels-1) >= n_cand >= any chain
    if ((rc = ensure(c, c->idx_keys, (size_t)tsize * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, c->idx_vals, (size_t)tsize * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->idx_J, (size_t)levels * n_cand * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->idx_S, (size_t)levels * n_cand * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, c->idx_L, (size_t)levels * n_cand * sizeof(uint32_t)))) return rc;
    X3Cand* cand = (X3Cand*)c->idx_cand.p;
    unsigned long long* keys = (unsigned long long*)c->idx_keys.p;
    uint32_t* vals = (uint32_t*)c->idx_vals.p;
    uint32_t* J = (uint32_t*)c->idx_J.p;
    unsigned long long* S = (unsigned long long*)c->idx_S.p;
    uint32_t* L = (uint32_t*)c->idx_L.p;
    HIPCHK(c, hipMemsetAsync(keys, 0, (size_t)tsize * sizeof(unsigned long long), c->stream));
    const unsigned cg = (n_cand + 255) / 256;
    hipLaunchKernelGGL(x3_index_hash_insert_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand,
                       keys, vals, tsize - 1);
    hipLaunchKernelGGL(x3_index_succ_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand,
                       (const unsigned long long*)keys, (const uint32_t*)vals, tsize - 1, J, S, L);
    uint32_t r = 1;
    for (; r + 4 <= levels; r += 4)  // four levels a launch
      hipLaunchKernelGGL(x3_index_double4_kernel, dim3(cg), dim3(256), 0, c->stream, n_cand,
                         (const uint32_t*)(J + (size_t)(r - 1) * n_cand),
                         (const unsigned long long*)(S + (size_t)(r - 1) * n_cand),
                         (const uint32_t*)(L + (size_t)(r - 1) * n_cand), J + (size_t)r * n_cand,
                         S + (size_t)r * n_cand, L + (size_t)r * n_cand);
    for (; r < levels; ++r)
      hipLaunchKernelGGL(x3_index_double_kernel, dim3(cg), dim3(256), 0, c->stream, n_cand,
                         (const uint32_t*)(J + (size_t)(r - 1) * n_cand),
                         (const unsigned long long*)(S + (size_t)(r - 1) * n_cand),
                         (const uint32_t*)(L + (size_t)(r - 1) * n_cand), J + (size_t)r * n_cand,
                         S + (size_t)r * n_cand, L + (size_t)r * n_cand);
    // start node and chain length stay on the device: the emit grid covers every candidate
    hipLaunchKernelGGL(x3_index_emit_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand, levels,
                       (const uint32_t*)J, (const unsigned long long*)S, (unsigned long long)max_frames,
                       (unsigned long long)wav_cap, (unsigned long long*)d_frame_offsets,
                       (unsigned long long*)d_wav_offsets, d_sum, (const unsigned long long*)keys, (const uint32_t*)vals,
                       tsize - 1, (const uint32_t*)(L + (size_t)(levels - 1) * n_cand));
    hipLaunchKernelGGL(x3_index_finalize_kernel, dim3(1), dim3(64), 0, c->stream, xw, len, len + phantom, bl0,
                       (const X3Cand*)cand, (const unsigned long long*)d_wav_offsets, d_sum);
  } else {
    hipLaunchKernelGGL(x3_index_finalize_kernel, dim3(1), dim3(64), 0, c->stream, xw, len, len + phantom, bl0,
                       (const X3Cand*)nullptr, (const unsigned long long*)nullptr, d_sum);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_summary_init, d_sum, sizeof *result, hipMemcpyDeviceToHost, c->stream));  // (pinned)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(result, c->h_summary_init, sizeof *result);
  if (result->pad) {
    c->last_error = "x3_index_dev: more frames in the stream than max_frames";
    return X3_ERR_BAD_ARG;
  }
  return X3_OK;
}

extern "C" int x3_index_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t max_frames,
                            uint64_t* d_frame_offsets, uint64_t* d_wav_offsets, uint64_t* n_frames,
                            uint64_t* n_samples, int* terminal) {
  if (!c || (!d_x3 && len) || !d_frame_offsets || !d_wav_offsets) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  X3IndexSummary r;
  int rc = index_dev_impl(c, d_x3, len, 0, 0, ~0ull, max_frames, d_frame_offsets, d_wav_offsets, &r);
  if (rc) return rc;
  if (n_frames) *n_frames = r.n_frames;
  if (n_samples) *n_samples = r.n_samples;
  if (terminal) *terminal = r.terminal;
  return X3_OK;
}


extern "C" int x3_decode_stream_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t len, const x3_params* p, int16_t* d_wav,
                                    uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  if (!c || !p || (!d_x3 && len) || (!d_wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  return decode_stream_dev_impl(c, d_x3, len, 0, p, d_wav, wav_cap, nullptr, n_out, frames_ok, frame_errors);
}

// the walk on the GPU (x3_index_kernels.h), then one decode launch; `phantom` as in walk_host.  own_out: decode
// into this scratch buffer, sized once the index knows the sample count, instead of d_wav.
int decode_stream_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                  int16_t* d_wav, uint64_t wav_cap, DevBuf* own_out, uint64_t* n_out,
                                  uint64_t* frames_ok, uint64_t* frame_errors) {
  // every frame is at least 22 bytes; index into internal buffers sized for the real count
  const uint64_t max_frames = len / 22 + 1;
  int rc;
  X3IndexSummary r;
  // sizes are not known before the candidate count: index_dev_impl sizes its own scratch, the two output arrays
  // are sized here from an upper bound that is refined by a first call when it is large
  uint64_t cap_frames = std::min<uint64_t>(max_frames, 1u << 20);
  // ---- ONE trip to the host (round 5).  A stream that is one clean chain of frames -- what an encoder
  // writes -- is walked by five launches that leave the frame count on the device; the decoder, the check pass and the merge
  // are enqueued right behind them with a grid for an upper bound of the count (a frame per KiB of stream) and read the
  // count from there (the groups behind it leave at once), and ONE copy brings both summaries back.  Anything the walk's
  // summary then objects to -- not a clean chain, more frames than the bound, sample offsets off the 8-byte grid -- is done
  // again the old way (the walk, a trip, the decode, a trip): the speculative decode wrote nothing that is not rewritten.
  {
    X3DevParams dq;
    // (frames of fewer than ~1 000 samples are more than one per KiB of stream: the bound below cannot hold, and every call
    // would pay the speculative launches and a sync before it takes the two trips anyway -- ADVICE r5)
    const bool try_one = !own_out && !c->opt.index_no_fast && !c->opt.two_trips && p->block_len == X3S_BL && !c->opt.decode_single &&
                         spf_of(p) >= 2048 &&
                         !c->force_single_wave_decode && derive(p, spf_of(p) > 0xFFFFFFFFull ? 0 : spf_of(p), &dq) == X3_OK &&
                         dq.k[1] == 1u && dq.k[2] == 3u && (reinterpret_cast<uintptr_t>(d_wav) & 7u) == 0 &&
                         (reinterpret_cast<uintptr_t>(d_x3) & 3u) == 0 && len >= 22;
    const uint64_t bound = std::min<uint64_t>(cap_frames, len / 1024 + 64);
    if (try_one) {
      if ((rc = ensure(c, c->frame_off, (bound + 1) * sizeof(uint64_t)))) return rc;
      if ((rc = ensure(c, c->wav_off, bound * sizeof(uint64_t)))) return rc;
      X3IndexSummary* ds = nullptr;
      rc = index_fast_launch(c, d_x3, len, phantom, 0u, wav_cap, bound, (uint64_t*)c->frame_off.p, (uint64_t*)c->wav_off.p, &ds);
      if (rc == X3_OK) {
        x3_params pq = *p;
        rc = decode_dev_impl(c, d_x3, len, (const uint64_t*)c->frame_off.p, bound, nullptr, (const uint64_t*)c->wav_off.p, &pq,
                             d_wav, wav_cap, nullptr, true, false, nullptr, &ds->n_frames);
        if (rc == X3_OK) {
          HIPCHK(c, hipMemcpyAsync(c->h_summary_init, ds, sizeof(X3IndexSummary), hipMemcpyDeviceToHost, c->stream));  // (pinned)
          HIPCHK(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(X3DecodeSummary), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
          std::memcpy(&r, c->h_summary_init, sizeof r);
          c->decode_pending = false;
          if (!r.pad && !r.pad2 && !r.unaligned && r.n_frames <= bound) {
            ++c->index_fast;
            ++c->stream_one_trip;
            const uint64_t F1 = r.n_frames;
            if (F1 == 0) return r.terminal;
            uint64_t first_bad = c->h_summary->first_bad, before = c->h_summary->samples_before;
            int bad_status = c->h_summary->first_bad_status;
            if (first_bad < F1) {   // rare: a frame is bad -- its status and the samples of the good frames before it
              c->dec_frames = F1;
              c->decode_pending = true;
              if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
            }
            if (n_out) *n_out = before;
            if (frames_ok) *frames_ok = first_bad;
            if (first_bad < F1) {
              if (bad_status == X3_ERR_OUT_OF_BOUNDS_INVERSE || bad_status == X3_ERR_FRAME_DECODE_INVALID_BPF) {
                if (frame_errors) *frame_errors = 1;
                return X3_OK;
              }
              return bad_status;
            }
            return r.terminal;
          }
        } else if (rc != X3_ERR_BAD_ARG) {
          return rc;
        }
      } else if (rc != X3_ERR_BAD_ARG) {
        return rc;
      }
    }
  }
  for (;;) {
    if ((rc = ensure(c, c->frame_off, (cap_frames + 1) * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(c, c->wav_off, cap_frames * sizeof(uint64_t)))) return rc;
    rc = index_dev_impl(c, d_x3, len, phantom, p->block_len == 0 ? 1u : 0u, wav_cap, cap_frames, (uint64_t*)c->frame_off.p,
                        (uint64_t*)c->wav_off.p, &r);
    if (rc == X3_ERR_BAD_ARG && cap_frames < max_frames) { cap_frames = max_frames; continue; }
    break;
  }
  if (rc) return rc;
  const uint64_t F = r.n_frames;
  const int terminal = r.terminal;
  if (F == 0) return terminal;
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;  // frames that need block_len are BAD_ARG frames of the index
  if (own_out) {
    if ((rc = ensure(c, *own_out, (r.n_samples + 65536) * sizeof(int16_t)))) return rc;
    d_wav = (int16_t*)own_out->p;
    wav_cap = std::min<uint64_t>(wav_cap, r.n_samples + 65535);
  }
  if ((rc = decode_dev_impl(c, d_x3, len, (const uint64_t*)c->frame_off.p, F, nullptr, (const uint64_t*)c->wav_off.p,
                            &pp, d_wav, wav_cap, nullptr, r.unaligned == 0, p->block_len == 0)))
    return rc;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
  if (n_out) *n_out = before;
  if (frames_ok) *frames_ok = first_bad;
  if (first_bad < F) {
    if (bad_status == X3_ERR_OUT_OF_BOUNDS_INVERSE || bad_status == X3_ERR_FRAME_DECODE_INVALID_BPF) {
      if (frame_errors) *frame_errors = 1;  // counted, the walk ends quietly (decodefile.rs:129-135)
      return X3_OK;
    }
    return bad_status;
  }
  return terminal;
}


extern "C" int x3_decode_stream(x3_ctx* c, const uint8_t* x3, uint64_t len, const x3_params* p, int16_t* wav,
                                uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  return decode_stream_impl(c, x3, len, 0, p, wav, wav_cap, n_out, frames_ok, frame_errors);
}

void walk_host(const uint8_t* buf, uint64_t buf_len, uint64_t real_total, uint64_t believed_total,
                      const x3_params* p, uint64_t wav_cap, uint64_t max_samples, HostWalk* w, uint32_t n_ch) {
  uint64_t pos = 0, remaining = believed_total, nsamp = 0;
  for (;;) {
    if (remaining <= 20) break;
    if (real_total - pos < 20) { w->terminal = X3_ERR_IO; break; }  // read_exact past the real end of the data
    if (buf_len - pos < 20) { w->need_more = true; break; }
    x3_frame_header h;
    int rc = read_frame_header_ch(buf + pos, 20, &h, n_ch);
    if (rc) { w->terminal = rc; break; }
    if (remaining - 20 < h.payload_len) break;
    // the buffer-size test comes before the payload is read (decodefile.rs:118-124): a payload that is both too
    // long and cut off by the real end of the data is FrameHeaderInvalidPayloadLen, not Io
    if (h.payload_len > X3_READ_BUFFER_SIZE) { w->terminal = X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; }
    if (real_total - pos - 20 < h.payload_len) { w->terminal = X3_ERR_IO; break; }
    if (buf_len - pos - 20 < h.payload_len) { w->need_more = true; break; }
    if (h.samples == 0 || h.payload_len < 2 * n_ch || nsamp + h.samples > wav_cap || (p->block_len == 0 && h.samples > 1)) {
      // payload CRC is checked before decode_frame runs, so let the GPU look at this frame too:
      // it reports the CRC error if there is one, BAD_ARG (reference panic) otherwise
      w->offs.push_back(pos);
      w->woffs.push_back(nsamp);
      w->terminal = X3_ERR_BAD_ARG;
      pos += 20 + (uint64_t)h.payload_len;
      break;
    }
    if (nsamp + h.samples > max_samples && !w->offs.empty()) { w->need_more = true; break; }
    w->offs.push_back(pos);
    w->woffs.push_back(nsamp);
    nsamp += h.samples;
    pos += 20 + h.payload_len;
    remaining -= 20 + h.payload_len;
  }
  w->nsamp = nsamp;
  w->end_pos = pos;
}

// decode the frames a walk collected from host memory into host memory: H2D, one decode launch, D2H of the
// samples in front of the first frame that fails.  *first_bad == F: all of them decoded.
int decode_frames_host(x3_ctx* c, const uint8_t* x3, uint64_t len, const HostWalk& w, const x3_params* p,
                              int16_t* wav, uint64_t wav_cap, uint64_t* before, uint64_t* first_bad, int* bad_status,
                              bool download,  // !download: the samples stay in c->out (x3_mgpu_decode_stream)
                              const uint8_t* d_x3) {  // the frames' bytes are on the device already
  const uint64_t F = w.offs.size();
  *before = 0;
  *first_bad = 0;
  *bad_status = 0;
  if (F == 0) return X3_OK;
  int rc;
  if (!d_x3) {
    if ((rc = ensure(c, c->in, len + 16))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, x3, len, hipMemcpyHostToDevice, c->stream));
    d_x3 = (const uint8_t*)c->in.p;
  }
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, F * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, (w.nsamp + 65536) * sizeof(int16_t)))) return rc;
  // (through pinned memory: a copy from pageable memory is staged by the runtime under a lock it shares with the large
  // pageable copies the chunked front ends have in flight on other threads -- 0.2-0.4 ms per call when they collide)
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (the copy before this one has left the pinned block)
  if (c->h_walk_cap < 2 * F * sizeof(uint64_t)) {
    if (c->h_walk) HIPCHK(c, hipHostFree(c->h_walk));
    c->h_walk = nullptr;
    c->h_walk_cap = 0;
    const size_t want = (2 * F * sizeof(uint64_t) * 5 / 4 + 4095) & ~(size_t)4095;
    HIPCHK(c, hipHostMalloc(&c->h_walk, want));
    c->h_walk_cap = want;
  }
  std::memcpy(c->h_walk, w.offs.data(), F * sizeof(uint64_t));
  std::memcpy((uint64_t*)c->h_walk + F, w.woffs.data(), F * sizeof(uint64_t));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, c->h_walk, F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, (uint64_t*)c->h_walk + F, F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;  // frames that need block_len were routed to BAD_ARG by the walk
  const uint64_t dev_wav_cap = std::min<uint64_t>(wav_cap, w.nsamp + 65535);
  bool aligned = true;
  for (uint64_t v : w.woffs) aligned = aligned && (v & 3ull) == 0;   // (rows on the 8-byte grid)
  if ((rc = decode_dev_impl(c, d_x3, len, (const uint64_t*)c->frame_off.p, F, nullptr,
                            (const uint64_t*)c->wav_off.p, &pp, (int16_t*)c->out.p, dev_wav_cap, nullptr, aligned,
                            p->block_len == 0)))
    return rc;
  if ((rc = x3_decode_result(c, first_bad, bad_status, before))) return rc;
  if (download && *before)
    HIPCHK(c, hipMemcpyAsync(wav, c->out.p, *before * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

// how the walk ends when frame `first_bad` of F failed with `bad_status` (decodefile.rs:96-100, 128-135)
int walk_result(uint64_t F, uint64_t first_bad, int bad_status, int terminal, uint64_t* frame_errors) {
  if (first_bad < F) {
    if (bad_status == X3_ERR_OUT_OF_BOUNDS_INVERSE || bad_status == X3_ERR_FRAME_DECODE_INVALID_BPF) {
      if (frame_errors) *frame_errors = 1;  // counted, the walk ends quietly (decodefile.rs:129-135)
      return X3_OK;
    }
    return bad_status;  // payload CRC mismatch (hard error) or BAD_ARG (reference panic)
  }
  return terminal;
}

// x3_decode_stream on a LONG stream in host memory, in chunks of whole frames (the file pipeline's scheme in one
// context).  The samples are 79 % of the bytes this entry point moves (config 3) and the link carries both directions at
// once, so: one host thread walks the headers of chunk i+2 (decodefile.rs:105-121, one dependent cache miss per frame)
// and sends its bytes up, this one decodes chunk i+1 (a launch of a few thousand frames lasts as long as one group of 64
// does: 0.6 ms, whatever the GPU could do beside it), a third brings the samples of chunk i down.  Two device buffers
// take turns on either side of the decoder.
static int decode_stream_host_chunked(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                      uint64_t chunk_samples, bool grow, int16_t* wav, uint64_t wav_cap, uint64_t* n_out,
                                      uint64_t* frames_ok, uint64_t* frame_errors) {
  int rc = x3_pipe_streams(c);
  if (rc) return rc;
  struct Chunk { HostWalk hw; uint64_t a, sample_off; const uint8_t* d_x3; };
  struct Piece { const void* src; uint64_t sample_off, count; };
  X3Handoff<Chunk> ready;
  X3Handoff<Piece> down;
  X3Progress decoded, landed;
  hipError_t up_err = hipSuccess, dl_err = hipSuccess;
  std::thread uploader, downloader;
  try {
  uploader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
    e = hipSetDevice(c->device);
    uint64_t a = 0, sample_off = 0;
    for (uint64_t k = 0; e == hipSuccess; ++k) {
      { std::lock_guard<std::mutex> g(decoded.mu); if (decoded.stop) break; }
      Chunk ck;
      const uint64_t real_total = len - a;
      // grow: a short first chunk so that the downloads start early, then longer ones (x1.5 up to x8: a chunk's walk and
      // upload take 0.7 of the time its predecessor's samples need to come down) -- every chunk costs a launch set and a
      // handful of runtime calls whatever its size, and those calls now and then stall for milliseconds beside the
      // pageable copies of the other two threads
      uint64_t budget = chunk_samples;
      if (grow)
        for (uint64_t g = 0; g < k && budget < 8 * chunk_samples; ++g) budget += budget >> 1;
      walk_host(x3 + a, real_total, real_total, real_total + phantom, p, wav_cap - sample_off, budget, &ck.hw);
      ck.a = a;
      ck.sample_off = sample_off;
      ck.d_x3 = nullptr;
      const bool last = !ck.hw.need_more;
      if (!ck.hw.offs.empty()) {
        // this chunk's bytes go where those of chunk k-3 were: not before that chunk has been decoded
        if (k >= 3 && !decoded.wait_for(k - 2)) break;
        DevBuf& buf = (k % 3) ? c->in_more[k % 3 - 1] : c->in;
        if (buf.cap < ck.hw.end_pos + 16) {
          if (buf.p) e = x3_dfree(buf.p);
          buf.p = nullptr;
          buf.cap = 0;
          const size_t want = (size_t)((ck.hw.end_pos * (grow ? 2 : 1) + 16 + (ck.hw.end_pos >> 3) + 255) & ~255ull);
          if (e == hipSuccess) e = x3_dmalloc(&buf.p, want);
          if (e == hipSuccess) buf.cap = want;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(buf.p, x3 + a, ck.hw.end_pos, hipMemcpyHostToDevice, c->ul_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->ul_stream);
        if (e != hipSuccess) break;
        ck.d_x3 = static_cast<const uint8_t*>(buf.p);
      }
      a += ck.hw.end_pos;
      sample_off += ck.hw.nsamp;
      ready.push(std::move(ck));
      if (last) break;
    }
    } catch (...) { e = hipErrorOutOfMemory; }   // (walk_host's vectors, the hand-off: never std::terminate -- ADVICE r3)
    up_err = e;
    ready.close();
  });
  downloader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      Piece pc;
      while (down.pop(&pc)) {
        if (e == hipSuccess && pc.count) {
          e = hipMemcpyAsync(wav + pc.sample_off, pc.src, pc.count * sizeof(int16_t), hipMemcpyDeviceToHost, c->dl_stream);
          if (e == hipSuccess) e = hipStreamSynchronize(c->dl_stream);
        }
        landed.advance();
      }
    } catch (...) { e = hipErrorOutOfMemory; landed.halt(); }
    dl_err = e;
  });
  } catch (const std::system_error&) {  // no thread to be had: the caller takes the stream in one piece
    decoded.halt();
    if (uploader.joinable()) uploader.join();
    return X3_PIPE_UNAVAILABLE;
  }
  uint64_t total = 0, frames = 0, ferr = 0, k = 0;
  int result = X3_OK;
  bool ended = false;
  Chunk ck;
  try {   // (as in encode_host_chunked: nothing thrown may pass the joins below)
  while (!ended && ready.pop(&ck)) {
    const uint64_t F = ck.hw.offs.size();
    uint64_t before = 0, first_bad = 0;
    int bad_status = 0;
    if (F) {
      // the buffer this chunk decodes into was the source of the download three chunks back
      if (k >= 3) landed.wait_for(k - 2);
      DevBuf* turn = (k % 3) ? &c->out_more[k % 3 - 1] : nullptr;
      if (turn) std::swap(c->out, *turn);
      if (grow && c->out.cap < (ck.hw.nsamp + 65536) * sizeof(int16_t))  // (room for the longer chunks that follow)
        rc = ensure(c, c->out, (2 * ck.hw.nsamp + 65536) * sizeof(int16_t));
      if (rc == X3_OK)
        rc = decode_frames_host(c, nullptr, ck.hw.end_pos, ck.hw, p, nullptr, wav_cap - ck.sample_off, &before, &first_bad,
                                &bad_status, false, ck.d_x3);
      const void* src = c->out.p;
      if (turn) std::swap(c->out, *turn);
      if (rc) break;
      down.push({src, ck.sample_off, before});
      decoded.advance();
      ++k;
    }
    uint64_t fe = 0;
    result = walk_result(F, first_bad, bad_status, ck.hw.need_more ? X3_OK : ck.hw.terminal, &fe);
    ferr += fe;
    frames += first_bad < F ? first_bad : F;
    total = ck.sample_off + before;
    ended = first_bad < F || !ck.hw.need_more;
  }
  } catch (...) {
    c->last_error = "x3_decode_stream: out of host memory in the chunked pipeline";
    rc = X3_ERR_HIP;
  }
  decoded.halt();  // (an uploader that waits for a decode that will not come)
  try { while (ready.pop(&ck)) {} } catch (...) {}
  down.close();
  uploader.join();
  downloader.join();
  if (rc) return rc;
  if (!ended) HIPCHK(c, up_err);
  HIPCHK(c, dl_err);
  if (n_out) *n_out = total;
  if (frames_ok) *frames_ok = frames;
  if (frame_errors) *frame_errors = ferr;
  return result;
}

// `phantom`: bytes the reader BELIEVES remain beyond the real data; a read that runs past the real end is
// X3Error::Io.
int decode_stream_impl(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                              int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                              uint64_t* frame_errors) {
  if (!c || !p || (!x3 && len) || (!wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  // Long streams: the header chain is one dependent cache miss per frame on the host (10 ms for the 69 120
  // frames of config 3), and a few launches on the GPU once the bytes are there anyway.  Short ones: the
  // other way round.  Option "host_walk" = 0/1 forces one or the other (tests run both).
  bool gpu_walk = len >= (4u << 20);
  if (c->opt.host_walk >= 0) gpu_walk = c->opt.host_walk == 0;
  // long streams in chunks, downloads beside uploads (unless a test pins the walk to the GPU)
  if (c->opt.host_walk != 0 && c->opt.host_chunk_frames >= 0 && len > 20 &&
      (c->opt.host_chunk_frames > 0 || len >= (16u << 20))) {
    const uint64_t spf = std::max<uint64_t>(spf_of(p), 1);
    const uint64_t chunk = c->opt.host_chunk_frames > 0 ? (uint64_t)c->opt.host_chunk_frames * spf : 16ull << 20;
    const int rc = decode_stream_host_chunked(c, x3, len, phantom, p, std::max<uint64_t>(chunk, 1), c->opt.host_chunk_frames == 0,
                                              wav, wav_cap, n_out, frames_ok, frame_errors);
    if (rc != X3_PIPE_UNAVAILABLE) return rc;
  }
  if (gpu_walk && len > 0) {
    int rc = ensure(c, c->in, len + 16);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, x3, len, hipMemcpyHostToDevice, c->stream));
    uint64_t before = 0;
    rc = decode_stream_dev_impl(c, (const uint8_t*)c->in.p, len, phantom, p, nullptr, wav_cap, &c->out, &before,
                                frames_ok, frame_errors);
    if (before) HIPCHK(c, hipMemcpyAsync(wav, c->out.p, before * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_out) *n_out = before;
    return rc;
  }
  HostWalk w;
  walk_host(x3, len, len, len + phantom, p, wav_cap, ~0ull, &w);
  const uint64_t F = w.offs.size();
  if (F == 0) return w.terminal;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  int rc = decode_frames_host(c, x3, len, w, p, wav, wav_cap, &before, &first_bad, &bad_status);
  if (rc) return rc;
  if (n_out) *n_out = before;
  if (frames_ok) *frames_ok = first_bad;
  return walk_result(F, first_bad, bad_status, w.terminal, frame_errors);
}

// single bare payload (decoder::decode_frame): wrap it in a frame header so that the one decode
// kernel serves both paths; decode_frame itself checks no CRC, so a correct one is supplied.

static int decode_frame_impl(x3_ctx* c, const uint8_t* payload, uint64_t len, int16_t* wav, uint64_t wav_cap,
                             const x3_params* p, uint64_t samples, uint64_t* n_out, bool use_cache) {
  if (!c || !p || !payload || !wav) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (len < 2 || samples == 0 || wav_cap < 1) return X3_ERR_BAD_ARG;  // reference panics
  // block_len == 0: every block is empty, the frame's fate is in its block type bits (x3_replay_frame) and no block
  // ever fails to fit
  const bool bl0 = p->block_len == 0 && samples > 1;
  if (samples > wav_cap && !bl0) {
    // decode_frame slices wav block by block (decoder.rs:49) and panics at the first block that does not fit -- but an
    // error in a block in front of that one is returned first.  The blocks in front of it are a frame of their own:
    const uint64_t bl = p->block_len ? p->block_len : 1;
    const uint64_t n_fit = 1 + ((wav_cap - 1) / bl) * bl;  // the first sample and the whole blocks that fit
    if (n_fit > 1) {
      std::vector<int16_t> tmp(n_fit);
      const int rc_fit = decode_frame_impl(c, payload, len, tmp.data(), n_fit, p, n_fit, nullptr, false);
      if (rc_fit != X3_OK) return rc_fit;
    }
    return X3_ERR_BAD_ARG;  // slice index panic
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if (bl0 || samples > 0xFFFF || len >= X3_FRAME_MAX_LENGTH || len > X3_READ_BUFFER_SIZE) {
    // not a frame the walk would hand over (decodefile.rs:118-121, x3.rs:145) and not one a header can describe, but
    // decode_frame itself has no such limits: the reference's reader, one thread (x3_decode_replay.h)
    if (samples > 0xFFFFFFFFull || len > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
    X3DevParams dpr;
    x3_params pr = *p;
    if (pr.block_len == 0) pr.block_len = 1;
    if ((rc = derive(&pr, 0, &dpr))) return rc;
    if (bl0) dpr.block_len = 0;
    if ((rc = ensure(c, c->in, len + 16))) return rc;
    if ((rc = ensure(c, c->out, (samples + 16) * sizeof(int16_t)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, payload, len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(x3_replay_one_kernel, dim3(1), dim3(64), 0, c->stream, (const uint8_t*)c->in.p, (uint32_t)len,
                       (uint32_t)samples, dpr, (int16_t*)c->out.p, (int32_t*)c->d_crc);
    HIPCHK(c, hipGetLastError());
    int32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, c->d_crc, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st != X3D_OK) return st;
    HIPCHK(c, hipMemcpyAsync(wav, c->out.p, samples * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_out) *n_out = samples;
    return X3_OK;
  }
  if (c->fcache && use_cache) {  // a frame of the stream announced with x3_decode_prefetch: decoded ahead, a window at a time
    rc = frame_cache_serve(c, payload, len, p, samples, wav);
    if (rc == X3_OK) {
      if (n_out) *n_out = samples;
      return X3_OK;
    }
    if (rc != X3_FRAME_CACHE_MISS) return rc;
  }
  if ((rc = ensure(c, c->in, 20 + len + 16))) return rc;
  if ((rc = ensure(c, c->frame_off, 2 * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, (samples + 16) * sizeof(int16_t)))) return rc;
  HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in.p + 20, payload, len, hipMemcpyHostToDevice, c->stream));
  uint16_t pcrc = 0;
  if ((rc = crc_dev_async(c, (const uint8_t*)c->in.p + 20, len))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  pcrc = *c->h_crc;
  uint8_t hdr[20];
  x3_write_frame_header(samples, 1, len, pcrc, hdr);
  const uint64_t zero = 0;
  HIPCHK(c, hipMemcpyAsync(c->in.p, hdr, 20, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;
  if ((rc = decode_dev_impl(c, (const uint8_t*)c->in.p, 20 + len, (const uint64_t*)c->frame_off.p, 1, nullptr,
                            (const uint64_t*)c->wav_off.p, &pp, (int16_t*)c->out.p, samples, nullptr)))
    return rc;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
  if (first_bad == 0) return bad_status;
  HIPCHK(c, hipMemcpyAsync(wav, c->out.p, samples * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_out) *n_out = samples;
  return X3_OK;
}

extern "C" int x3_decode_frame(x3_ctx* c, const uint8_t* payload, uint64_t len, int16_t* wav, uint64_t wav_cap,
                               const x3_params* p, uint64_t samples, uint64_t* n_out) {
  return decode_frame_impl(c, payload, len, wav, wav_cap, p, samples, n_out, true);
}


#ifdef X3_DBG_STAMPS
// stamp builds: the decoders' per-phase clocks (this unit's copy of x3_dbg; tools/scratch/dbg_stamps_split.py)
extern "C" int x3_dbg_read(x3_ctx* c, unsigned long long* out, uint64_t n) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_dbg), n * sizeof(unsigned long long)));
  return X3_OK;
}
#endif

#include "x3_bits.h"
#define X3_MC_DECODE
#include "x3_mc.h"

// ---- x3_place_buffers (include/x3hip.h, "Placement"): the round trip timed on every pair of candidate buffers
extern "C" int x3_place_buffers(x3_ctx* c, const int16_t* d_wav, uint64_t n, const x3_params* p, uint8_t* const* d_streams,
                                uint32_t n_streams, uint64_t cap, uint64_t* d_frame_offsets, int16_t* const* d_backs,
                                uint32_t n_backs, uint32_t warm, uint32_t steps, double* ms_per_step) {
  if (!c || !d_wav || !n || !p || !d_streams || !n_streams || !d_frame_offsets || !d_backs || !n_backs || !steps || !ms_per_step)
    return X3_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n_streams; ++i) if (!d_streams[i]) return X3_ERR_BAD_ARG;
  for (uint32_t j = 0; j < n_backs; ++j) if (!d_backs[j]) return X3_ERR_BAD_ARG;
  const uint64_t F = x3_num_frames(n, p);
  const x3_batch b{n, n, 1};
  int rc = X3_OK;
  auto trips = [&](uint8_t* d_out, int16_t* d_back, uint32_t k) -> int {
    for (uint32_t t = 0; t < k; ++t) {
      if ((rc = x3_encode_dev(c, d_wav, &b, p, d_out, cap, 0, d_frame_offsets))) return rc;
      if ((rc = x3_decode_dev(c, d_out, cap, d_frame_offsets, F, &b, nullptr, p, d_back, n, nullptr))) return rc;
    }
    uint64_t pos = 0, first_bad = 0, before = 0;
    int bad_status = 0;
    if ((rc = x3_encode_result(c, &pos, nullptr))) return rc;       // (ByteWriterInsufficientMemory: cap does not hold the stream)
    if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
    return bad_status ? bad_status : X3_OK;
  };
  // (the clocks and the caches of a process that has just started: the first pair would pay for them)
  if ((rc = trips(d_streams[0], d_backs[0], 6 * warm + 1))) return rc;
  for (uint32_t i = 0; i < n_streams; ++i) {
    for (uint32_t j = 0; j < n_backs; ++j) {
      if (warm && (rc = trips(d_streams[i], d_backs[j], warm))) return rc;   // (the decoder's pace controller settles)
      const auto t0 = std::chrono::steady_clock::now();
      if ((rc = trips(d_streams[i], d_backs[j], steps))) return rc;
      const auto t1 = std::chrono::steady_clock::now();
      ms_per_step[(size_t)i * n_backs + j] = std::chrono::duration<double, std::milli>(t1 - t0).count() / steps;
    }
  }
  return X3_OK;
}
