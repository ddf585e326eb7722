// x3_encode.hip -- encoder kernels and the x3_encode* entry points (C ABI: include/x3hip.h; units: x3_internal.h).
#include "x3_internal.h"
#include "x3_encode_kernel.h"
#include "x3_encode_stream2_kernel.h"
#include "x3_encode_wave_kernel.h"

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// x3_encode_stream2_kernel has no "diff outside the reference's Rice table" test (the reference panics there:
// X3_ERR_BAD_ARG from the two-pass kernels): true when no block can need it.  A block with max|d| = m <= thr[2] is
// coded with code[ft(m)], ft = [m > thr0] + [m > thr1] (encoder.rs:241-247); it is inside that code's table when
// m <= min(offset, len - offset - 1).
#ifndef X3_ENC_FRAME_ALIGN
// samples between frame starts that the single-pass encoders take: 4 = frames on 8-byte boundaries of a buffer that is
// itself dword aligned.  (Their sample loads are range-checked 16-byte BUFFER loads, which need dword alignment only; until round 4 this was 8 -- frames of an odd
// number of blocks went to the two-pass kernels, 3.6 ms against 0.45 for 501 blocks a frame.)
#define X3_ENC_FRAME_ALIGN 4u
#endif
static bool stream_safe_thresholds(const x3_params* p) {
  uint32_t mmax[3] = {0, 0, 0};
  bool used[3] = {false, false, false};
  const uint32_t top = std::min<uint32_t>(p->thresholds[2], 70000u);
  for (uint32_t m = 0; m <= top; ++m) {
    const uint32_t ft = (m > p->thresholds[0] ? 1u : 0u) + (m > p->thresholds[1] ? 1u : 0u);
    used[ft] = true;
    mmax[ft] = m;
  }
  for (int ft = 0; ft < 3; ++ft) {
    if (!used[ft]) continue;
    const uint32_t cde = p->codes[ft];
    if (cde > 3) return false;
    const uint32_t inside = std::min(X3_RICE_OFFSET[cde], X3_RICE_LEN[cde] - X3_RICE_OFFSET[cde] - 1u);
    if (mmax[ft] > inside) return false;
  }
  return true;
}

int plan_encode(x3_ctx* c, const x3_batch* b, const x3_params* p, uint64_t spf, EncPlan* pl) {
  int rc = derive(p, spf, &pl->dp);
  if (rc) return rc;
  if (spf == 0 || spf > 0xFFFFFFFFull || p->block_len == 0) return X3_ERR_BAD_ARG;
  const uint64_t fpc = (b->n_per_clip + spf - 1) / spf;
  if (fpc == 0 || fpc > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
  const uint64_t F = fpc * b->n_clips;
  if (F == 0 || F > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  pl->g.n_per_clip = b->n_per_clip;
  pl->g.clip_stride = b->clip_stride;
  pl->g.fpc = (uint32_t)fpc;
  pl->g.n_frames = F;
  const uint64_t nmax = std::min<uint64_t>(spf, b->n_per_clip);   // samples in the largest frame
  // a block longer than MAX_BLOCK_LENGTH = 60 samples overruns the reference's diff array (encoder.rs:296-299)
  if (std::min<uint64_t>(p->block_len, nmax - 1) > 60) return X3_ERR_BAD_ARG;
  const uint64_t nblocks = (nmax - 1 + p->block_len - 1) / p->block_len;
  uint32_t nthr = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(64, (nblocks + 63) & ~63ull));
  pl->nthr = nthr;
  const uint64_t in_bytes = (2 * nmax + 4 + 15) & ~15ull;  // + the dword read behind the last sample
  const uint64_t img_dw = ((5 + (max_payload_bytes(nmax, p->block_len) + 3) / 4 + 4) + 3) & ~3ull;
  const uint64_t smem = X3_ENC_SMEM_HDR + in_bytes + img_dw * 4;
  if (smem > 160 * 1024) {
    c->last_error = "frame too large for the LDS-resident encoder (block_len*blocks_per_frame)";
    return X3_ERR_BAD_ARG;
  }
  pl->lds_in_bytes = (uint32_t)in_bytes;
  pl->img_dwords = (uint32_t)img_dw;
  pl->smem = (size_t)smem;
  return X3_OK;
}

int ctl_begin(x3_ctx* c) {
  c->ctl_half ^= 1;
  c->d_status = c->d_ctl_base + 32 * c->ctl_half;   // (128 bytes each)
  c->d_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_status) + 32);
  c->d_end_pos = c->d_stats + 6;
  const bool clean = c->ctl_clean[c->ctl_half] && !c->capturing;
  c->ctl_clean[c->ctl_half] = false;
  if (!clean) HIPCHK(c, hipMemsetAsync(c->d_status, 0, 128, c->stream));
  return X3_OK;
}

int encode_dev_impl(x3_ctx* c, const int16_t* d_wav, const x3_batch* b, const x3_params* p, uint64_t spf,
                           uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets,
                           const X3FrameTable* tab) {
  EncPlan pl;
  int rc = plan_encode(c, b, p, spf, &pl);
  if (rc) return rc;
  if (tab) {   // (the batch is then "n_frames clips of at most one frame each": plan_encode's geometry of the largest frame)
    pl.g.src_off = tab->src_off;
    pl.g.src_n = tab->src_n;
  }
  if (reinterpret_cast<uintptr_t>(d_out) & 1u) return X3_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d_wav) & 1u) return X3_ERR_BAD_ARG;
  const uint64_t F = pl.g.n_frames;
  if ((rc = ensure(c, c->frame_bytes, F * sizeof(uint32_t)))) return rc;
  uint64_t* d_off = d_frame_offsets;
  if (!d_off) {
    if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
    d_off = (uint64_t*)c->frame_off.p;
  }
  if ((rc = ctl_begin(c))) return rc;
  // the segment index of this call (x3_encode_dev_seg): only the wave encoder fills it; every other path leaves a header
  // that says "no index" (the decoder then takes whole frames per lane)
  const X3SegSpec seg = c->enc_seg;
  c->enc_seg = X3SegSpec{nullptr, 0, 0};
  bool seg_header_open = seg.d_index != nullptr;   // (a wave-encoder call that fills the index writes the header itself)
  auto seg_header_none = [&]() -> int {
    if (seg_header_open) HIPCHK(c, hipMemsetAsync(seg.d_index, 0, sizeof(uint64_t), c->stream));
    seg_header_open = false;
    return X3_OK;
  };
  // ---- single-pass path: default block length, frames on dword boundaries (buffer loads)
  // (block lengths 10 and 40, round 6: the second-generation kernel's lanes hold 20 samples whatever a block is -- half a
  // block, one, or two -- so the bound on a frame is the same 512 lanes; the wave encoder stays with the default length)
  const bool stream_path = (p->block_len == 20 || p->block_len == 10 || p->block_len == 40) &&
                           (std::min<uint64_t>(spf, b->n_per_clip) + 18) / 20 <= 512 &&
                           (spf % X3_ENC_FRAME_ALIGN) == 0 &&
                           (b->n_clips == 1 || (b->clip_stride % X3_ENC_FRAME_ALIGN) == 0) &&
                           (reinterpret_cast<uintptr_t>(d_wav) & 3u) == 0 && !c->force_two_pass && !c->opt.two_pass &&
                           (!tab || tab->even);
  c->last_enc = {d_wav, *b, *p, spf, d_out, out_cap, start_pos, d_frame_offsets, tab ? tab->src_off : nullptr,
                 tab ? tab->src_n : nullptr, tab ? tab->even : false, seg};
  // part + two worst-case frame images + CRC tables + the multipliers of one chunk size (x3_encode_stream2_kernel.h)
  const size_t smem2 = X3_ENC_SMEM_HDR + 2 * (size_t)pl.img_dwords * 4 + 2048 + X3_K2_DWORDS * 4;
  c->last_enc_gen = 0;
  if (stream_path && c->opt.enc_gen == 3 && c->opt.stream_wgs == 0 &&
      stream_safe_thresholds(p) && smem2 <= 160 * 1024) {
    if (c->prefer_gen2) {
      // the last call's content was mostly dense: the second generation, until it counts few dense frames again
    } else {
      // third generation (x3_encode_wave_kernel.h): one wave per frame, sixteen waves per CU, one workgroup per CU
      static_assert(X3W_SMEM <= 160 * 1024, "LDS");
      typedef decltype(&x3_encode_wave_kernel<false, 20u>) gen3_fn;
      // (a block of 40 is two of a lane's runs of 20, a run is two blocks of 10: x3w_analyse40, x3w_analyse10)
      const uint32_t bl = p->block_len;
      const gen3_fn wave_fn = pl.g.src_off ? (bl == 40 ? &x3_encode_wave_kernel<true, 40u> : bl == 10 ? &x3_encode_wave_kernel<true, 10u>
                                                                                                       : &x3_encode_wave_kernel<true, 20u>)
                                           : (bl == 40 ? &x3_encode_wave_kernel<false, 40u> : bl == 10 ? &x3_encode_wave_kernel<false, 10u>
                                                                                                       : &x3_encode_wave_kernel<false, 20u>);
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(wave_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3W_SMEM));
      uint64_t nwg_max = std::min<uint64_t>((uint64_t)c->n_cus, X3W_MAX_NWG);
      if (c->opt.wave_nwg > 0) nwg_max = std::min<uint64_t>(nwg_max, (uint64_t)c->opt.wave_nwg);
      X3WaveArgs wa;
      wa.m = (uint32_t)std::min<uint64_t>(X3W_WAVES, (F + nwg_max - 1) / nwg_max);
      if (c->opt.wave_m > 0) wa.m = (uint32_t)c->opt.wave_m;
      const uint64_t n_wggen = (F + wa.m - 1) / wa.m;
      wa.nwg = (uint32_t)std::min<uint64_t>(nwg_max, n_wggen);
      wa.n_wggen = (uint32_t)n_wggen;
      const uint64_t step = (uint64_t)wa.nwg * wa.m;
      wa.step_clip = (uint32_t)(step / pl.g.fpc);
      wa.step_idx = (uint32_t)(step % pl.g.fpc);
      const size_t desc_bytes = (n_wggen + X3W_DESC_PAD) * sizeof(uint32_t);
      const bool fresh = c->desc.cap < desc_bytes;
      if ((rc = ensure(c, c->desc, desc_bytes))) return rc;
      if (fresh || c->capturing || ++c->desc_epoch > 0xFFFu) {   // (recorded into a graph: cleared by a node of it, every replay)
        HIPCHK(c, hipMemsetAsync(c->desc.p, 0, c->desc.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + 4, 0, 16, c->stream));
        c->desc_epoch = 1;
      }
      wa.wav = d_wav;
      wa.out = d_out;
      wa.frame_off = d_off;
      wa.desc = (uint32_t*)c->desc.p + X3W_DESC_PAD;
      wa.ctl = reinterpret_cast<unsigned char*>(c->d_status);
      wa.tabs = c->d_wtab;
      wa.log = c->d_pace + X3_LOG_ENC_BASE;
      wa.log_epoch = ++c->enc_log_epoch & 0xFFFu;
      if ((rc = ensure(c, c->dense_list, F * sizeof(uint32_t)))) return rc;
      wa.dense_list = (uint32_t*)c->dense_list.p;
      wa.out_cap = out_cap;
      wa.start_pos = start_pos;
      wa.n_per_clip = pl.g.n_per_clip;
      wa.clip_stride = pl.g.clip_stride;
      wa.n_frames = F;
      wa.src_off = pl.g.src_off;
      wa.src_n = pl.g.src_n;
      wa.fpc = pl.g.fpc;
      wa.spf = pl.dp.spf;
      wa.epoch = c->desc_epoch;
      wa.thr0 = pl.dp.thr[0];
      wa.thr1 = pl.dp.thr[1];
      wa.thr2 = pl.dp.thr[2];
      wa.kpack = pl.dp.k[0] | (pl.dp.k[1] << 8) | (pl.dp.k[2] << 16);
      wa.drop_wgi = c->opt.wave_drop >= 0 ? (uint32_t)c->opt.wave_drop : 0xFFFFFFFFu;
      wa.seg = nullptr;
      wa.seg_log2 = 0;
      wa.seg_pitch = 0;
      if (seg.d_index && bl == 20 && seg.seg_blocks >= 4 && (seg.seg_blocks & (seg.seg_blocks - 1)) == 0) {
        const uint64_t bpf = (spf + 19) / 20;   // blocks of a full frame (<= 512 here): the pitch follows the parameters, not the call
        const uint64_t nidx = (bpf + seg.seg_blocks - 1) / seg.seg_blocks;
        if (nidx >= 2 && !tab) {
          wa.seg = reinterpret_cast<uint2*>(seg.d_index);
          wa.seg_log2 = (uint32_t)__builtin_ctz(seg.seg_blocks);
          wa.seg_pitch = (uint32_t)(nidx - 1);
        }
      }
      if (!wa.seg && (rc = seg_header_none())) return rc;
      {
        TimerScope ts(c, 0, nullptr, true);
        X3_LAUNCH_TIMED(ts, wave_fn, dim3(wa.nwg), dim3(X3W_THREADS), X3W_SMEM, c->stream, wa);
      }
      {
        // The dense pass, always: the frames the wave kernel listed (none, in most recordings: the workgroups read a zero
        // count and leave, ~2 us of queue) written at the offsets it assigned.  In the stream, not in x3_encode_result:
        // whatever the caller enqueues behind this call -- x3_decode_dev, a copy -- finds the whole stream.
        typedef decltype(&x3_encode_stream2_kernel<true, false, 20u>) dense_fn_t;
        const dense_fn_t dense_fn =
            pl.g.src_off ? (bl == 40 ? &x3_encode_stream2_kernel<true, true, 40u> : bl == 10 ? &x3_encode_stream2_kernel<true, true, 10u>
                                                                                             : &x3_encode_stream2_kernel<true, true, 20u>)
                         : (bl == 40 ? &x3_encode_stream2_kernel<true, false, 40u> : bl == 10 ? &x3_encode_stream2_kernel<true, false, 10u>
                                                                                              : &x3_encode_stream2_kernel<true, false, 20u>);
        if (smem2 > 64 * 1024)
          HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
        const uint64_t per_cu = std::max<uint64_t>(1, (160 * 1024) / (smem2 + 256));
        const uint64_t grid = std::min<uint64_t>(F, (uint64_t)c->n_cus * std::min<uint64_t>(per_cu, 3));
        TimerScope ts(c, 5, nullptr, true);
        X3_LAUNCH_TIMED(ts, dense_fn, dim3((unsigned)grid), dim3(X3_STREAM2_THREADS), smem2,
                        c->stream, d_wav, pl.g, pl.dp, d_off, d_out, out_cap, start_pos, (uint32_t*)nullptr, 0u,
                        reinterpret_cast<unsigned char*>(c->d_status), (const uint32_t*)c->d_xk2,
                        (const uint16_t*)c->d_crctab, pl.img_dwords, (uint32_t*)nullptr, (const uint32_t*)c->dense_list.p,
                        reinterpret_cast<uint32_t*>(c->d_ctl_base + 32 * (c->ctl_half ^ 1)));
      }
      HIPCHK(c, hipGetLastError());
      c->ctl_clean[c->ctl_half ^ 1] = !c->capturing;   // (the dense pass clears the next call's control block -- when it runs)
      c->last_enc_gen = 3;
      c->encode_pending = true;
      c->enc_start_pos = start_pos;
      return X3_OK;
    }
  }
  if ((rc = seg_header_none())) return rc;
  if (stream_path && stream_safe_thresholds(p)) {
    // second generation (x3_encode_stream2_kernel.h): eight waves, no sample tile in LDS
    // the instantiation of this call: block length x (frames from a table?)
    typedef decltype(&x3_encode_stream2_kernel<false, false, 20u>) gen2_fn;
    const gen2_fn fn_plain = p->block_len == 10 ? &x3_encode_stream2_kernel<false, false, 10u>
                           : p->block_len == 40 ? &x3_encode_stream2_kernel<false, false, 40u>
                                                : &x3_encode_stream2_kernel<false, false, 20u>;
    const gen2_fn fn_tab = p->block_len == 10 ? &x3_encode_stream2_kernel<false, true, 10u>
                         : p->block_len == 40 ? &x3_encode_stream2_kernel<false, true, 40u>
                                              : &x3_encode_stream2_kernel<false, true, 20u>;
    const gen2_fn fn = pl.g.src_off ? fn_tab : fn_plain;
    const uint64_t wg_key = (uint64_t)smem2 | ((uint64_t)p->block_len << 32) | (pl.g.src_off ? 1ull << 40 : 0ull);
    if (c->stream_wg_per_cu < 0 || c->stream_wg_key != wg_key) {
      c->stream_wg_key = wg_key;
      // Offsets wait on the other workgroups' frame sizes, so EVERY workgroup of the grid must be resident.  The
      // occupancy API can over-report by one block per CU (MI355X_MICROARCH.md, "Residency"), so it is capped
      // by the kernel's own register/LDS footprint: eight waves are two per SIMD, whatever the placement.
      if (smem2 > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
      int nb = 0;
      HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), X3_STREAM2_THREADS, smem2));
      hipFuncAttributes fa;
      HIPCHK(c, hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn)));
      const int alloc = ((fa.numRegs + 7) / 8) * 8;
      const int wps = std::min(8, 512 / std::max(alloc, 8));
      const int by_regs = (4 * wps) / 8;
      const int by_lds = (int)((160 * 1024) / (smem2 + 256));  // (+ allocation granularity)
      c->stream_wg_per_cu = std::max(0, std::min(std::min(nb, 4), std::min(by_regs, by_lds)));
      // experiments and the fallback test: force a grid (one that is too large cannot be resident: the size
      // waits time out and x3_encode_result re-encodes with the two-pass kernels)
      if (c->opt.stream_wgs > 0) c->stream_wg_per_cu = c->opt.stream_wgs;
      if (c->opt.verbose)
        std::fprintf(stderr, "x3hip: stream encoder v2 (block length %u) %d VGPRs, %zu B LDS, occupancy API %d, by_regs %d, by_lds %d -> %d workgroups/CU\n",
                     p->block_len, fa.numRegs, smem2, nb, by_regs, by_lds, c->stream_wg_per_cu);
    }
    if (c->stream_wg_per_cu >= 1 && smem2 <= 160 * 1024) {
      // frame-size descriptors {epoch:12 | bytes:20}: the epoch makes last launch's words "not ready"
      // without clearing the array (cleared when it is (re)allocated and when the epoch wraps)
      const uint64_t grid = std::min<uint64_t>(std::min<uint64_t>(F, X3_STREAM2_MAX_GRID), (uint64_t)c->n_cus * c->stream_wg_per_cu);
      const size_t desc_pad = 1024 + 64;  // words in front of desc[0]: the windows of the first frames reach below frame 0
      const size_t desc_bytes = (F + desc_pad) * sizeof(uint32_t);
      const bool fresh = c->desc.cap < desc_bytes;
      if ((rc = ensure(c, c->desc, desc_bytes))) return rc;
      if (fresh || c->capturing || ++c->desc_epoch > 0xFFFu) {   // (recorded into a graph: cleared by a node of it, every replay)
        HIPCHK(c, hipMemsetAsync(c->desc.p, 0, c->desc.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + 4, 0, 16, c->stream));  // (the encoder's pace words carry the same epoch)
        c->desc_epoch = 1;
      }
      {
        TimerScope ts(c, 0);
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(X3_STREAM2_THREADS), smem2,
                           c->stream, d_wav, pl.g, pl.dp, d_off, d_out, out_cap, start_pos, (uint32_t*)c->desc.p + desc_pad,
                           c->desc_epoch, reinterpret_cast<unsigned char*>(c->d_status), (const uint32_t*)c->d_xk2,
                           (const uint16_t*)c->d_crctab, pl.img_dwords, c->d_pace + 4, (const uint32_t*)nullptr,
                           (uint32_t*)nullptr);
      }
      HIPCHK(c, hipGetLastError());
      c->last_enc_gen = 2;
      c->encode_pending = true;
      c->enc_start_pos = start_pos;
      return X3_OK;
    }
  }
  if (pl.smem > 64 * 1024) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
  }
  // ---- any geometry in ONE pass (round 4): sizes by decoupled look-back (x3_encode_kernel.h, LOOKBACK).  The two passes
  // below stay as what a launch falls back to whose look-back gave up (x3_encode_result), and for option two_pass.
  if (!c->force_two_pass && !c->opt.two_pass) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(pl.smem, 64 * 1024)));
    const size_t lb_bytes = F * sizeof(unsigned long long);
    const bool fresh = c->lb_desc.cap < lb_bytes;
    if ((rc = ensure(c, c->lb_desc, lb_bytes))) return rc;
    if (fresh || c->capturing || ++c->lb_epoch > 0xFFFu) {
      HIPCHK(c, hipMemsetAsync(c->lb_desc.p, 0, c->lb_desc.cap, c->stream));
      c->lb_epoch = 1;
    }
    {
      TimerScope ts(c, 0);
      hipLaunchKernelGGL((x3_encode_frames_kernel<false, true>), dim3((unsigned)F), dim3(pl.nthr), pl.smem, c->stream, d_wav,
                         pl.g, pl.dp, (const uint64_t*)d_off, (uint32_t*)c->lb_desc.p, d_out, start_pos, c->d_stats,
                         c->d_status, (const uint16_t*)c->d_xpow, pl.lds_in_bytes, pl.img_dwords, 1u, (uint64_t)0,
                         c->lb_epoch, out_cap, c->d_end_pos, c->opt.lb_drop >= 0 ? (uint32_t)c->opt.lb_drop : 0xFFFFFFFFu);
    }
    HIPCHK(c, hipGetLastError());
    c->last_enc_gen = 1;
    c->encode_pending = true;
    c->enc_start_pos = start_pos;
    return X3_OK;
  }
  {
    TimerScope ts(c, 2);
    hipLaunchKernelGGL(x3_encode_frames_kernel<true>, dim3((unsigned)F), dim3(pl.nthr),
                       X3_ENC_SMEM_HDR + pl.lds_in_bytes, c->stream, d_wav, pl.g, pl.dp, (const uint64_t*)nullptr,
                       (uint32_t*)c->frame_bytes.p, (uint8_t*)nullptr, start_pos, c->d_stats, c->d_status,
                       (const uint16_t*)c->d_xpow, pl.lds_in_bytes, 0u, 1u, (uint64_t)0);
  }
  {
    TimerScope ts(c, 3);
    hipLaunchKernelGGL(x3_scan_frame_offsets_kernel, dim3(1), dim3(1024), 0, c->stream,
                       (const uint32_t*)c->frame_bytes.p, F, start_pos, out_cap, d_off, c->d_end_pos, c->d_status);
  }
  {
    TimerScope ts(c, 0);
    hipLaunchKernelGGL(x3_encode_frames_kernel<false>, dim3((unsigned)F), dim3(pl.nthr), pl.smem, c->stream, d_wav,
                       pl.g, pl.dp, (const uint64_t*)d_off, (uint32_t*)nullptr, d_out, start_pos, c->d_stats,
                       c->d_status, (const uint16_t*)c->d_xpow, pl.lds_in_bytes, pl.img_dwords, 1u, (uint64_t)0);
  }
  HIPCHK(c, hipGetLastError());
  c->encode_pending = true;
  c->enc_start_pos = start_pos;
  return X3_OK;
}

extern "C" int x3_encode_dev(x3_ctx* c, const int16_t* d_wav, const x3_batch* batch, const x3_params* p,
                             uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets) {
  if (!c || !d_wav || !batch || !p || !d_out) return X3_ERR_BAD_ARG;
  if (batch->n_per_clip == 0 || batch->n_clips == 0) return X3_ERR_BAD_ARG;
  if (batch->n_clips > 1 && batch->clip_stride < batch->n_per_clip) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return encode_dev_impl(c, d_wav, batch, p, spf_of(p), d_out, out_cap, start_pos, d_frame_offsets);
}

// x3_encode_dev that also leaves the SEGMENT INDEX of the stream (include/x3hip.h): the encoder's prefix scan over the
// blocks' bit lengths knows where every block begins, its input knows the sample in front of it
extern "C" int x3_encode_dev_seg(x3_ctx* c, const int16_t* d_wav, const x3_batch* batch, const x3_params* p,
                                 uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets,
                                 uint64_t* d_seg_index, uint32_t seg_blocks) {
  if (!c || !d_wav || !batch || !p || !d_out) return X3_ERR_BAD_ARG;
  if (batch->n_per_clip == 0 || batch->n_clips == 0) return X3_ERR_BAD_ARG;
  if (batch->n_clips > 1 && batch->clip_stride < batch->n_per_clip) return X3_ERR_BAD_ARG;
  if (d_seg_index && (seg_blocks < 4 || (seg_blocks & (seg_blocks - 1)) || seg_blocks > 2048 || (reinterpret_cast<uintptr_t>(d_seg_index) & 7u)))
    return X3_ERR_BAD_ARG;   // (the encoder's entries fall on its lanes' block runs: a power of two >= 4)
  HIPCHK(c, hipSetDevice(c->device));
  c->enc_seg = X3SegSpec{d_seg_index, seg_blocks, d_seg_index ? 2 : 0};
  return encode_dev_impl(c, d_wav, batch, p, spf_of(p), d_out, out_cap, start_pos, d_frame_offsets);
}


// Frames from anywhere: frame f is src_samples[f] samples (1 .. block_len * blocks_per_frame) at d_wav + src_offsets[f].
// Host arrays: they are checked, and copied to the device, here.  (Clips of different lengths are runs of such frames.)
extern "C" int x3_encode_frames_dev(x3_ctx* c, const int16_t* d_wav, const uint64_t* src_offsets, const uint32_t* src_samples,
                                    uint64_t n_frames, const x3_params* p, uint8_t* d_out, uint64_t out_cap,
                                    uint64_t start_pos, uint64_t* d_frame_offsets) {
  if (!c || !d_wav || !src_offsets || !src_samples || !n_frames || !p || !d_out) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t spf = spf_of(p);
  if (spf == 0 || n_frames > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  uint32_t n_max = 0;
  bool even = true;
  for (uint64_t f = 0; f < n_frames; ++f) {
    if (src_samples[f] == 0 || src_samples[f] > spf) return X3_ERR_BAD_ARG;
    n_max = std::max(n_max, src_samples[f]);
    even = even && (src_offsets[f] & 1ull) == 0;
  }
  int rc;
  if ((rc = ensure(c, c->src_tab, n_frames * 12 + 16))) return rc;
  uint64_t* d_so = (uint64_t*)c->src_tab.p;
  uint32_t* d_sn = (uint32_t*)(d_so + n_frames);
  // the caller's arrays are read HERE (into a pinned copy of the context's); the copy to the device is asynchronous
  const size_t tab_bytes = n_frames * 12;
  if (c->ev_src_tab) HIPCHK(c, hipEventSynchronize(c->ev_src_tab));   // (the last call's copy out of the pinned buffer)
  else HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_tab, hipEventDisableTiming));
  if (c->h_src_tab_cap < tab_bytes) {
    if (c->h_src_tab) (void)hipHostFree(c->h_src_tab);
    c->h_src_tab = nullptr;
    c->h_src_tab_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_src_tab, tab_bytes + tab_bytes / 4 + 64));
    c->h_src_tab_cap = tab_bytes + tab_bytes / 4 + 64;
  }
  std::memcpy(c->h_src_tab, src_offsets, n_frames * sizeof(uint64_t));
  std::memcpy((char*)c->h_src_tab + n_frames * sizeof(uint64_t), src_samples, n_frames * sizeof(uint32_t));
  HIPCHK(c, hipMemcpyAsync(d_so, c->h_src_tab, tab_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_src_tab, c->stream));
  const x3_batch b{n_max, 0, n_frames};   // n_frames clips of one frame each as far as the plan is concerned
  const X3FrameTable tab{d_so, d_sn, even};
  return encode_dev_impl(c, d_wav, &b, p, spf, d_out, out_cap, start_pos, d_frame_offsets, &tab);
}

extern "C" int x3_encode_result(x3_ctx* c, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c) return X3_ERR_BAD_ARG;
  if (!c->encode_pending) return X3_ERR_BAD_ARG;
  // status and statistics are fetched here, not behind every launch: a small copy is a packet of its own in
  // the queue (~8 us), and a pipeline that launches encode and decode back to back asks once per batch
  HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status, 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->encode_pending = false;
  if (c->h_status[1] != X3D_SIZE_WAIT_TIMEOUT && (c->last_enc_gen == 3 || c->last_enc_gen == 2)) {
    // how dense the call's content was (both single-pass generations count the frames beyond the wave encoder's image):
    // the hint for the context's next call, nothing else
    uint32_t dense;
    std::memcpy(&dense, reinterpret_cast<const unsigned char*>(c->h_status) + X3_CTL_DENSE_COUNT, sizeof dense);
    const uint64_t frames = ((c->last_enc.b.n_per_clip + c->last_enc.spf - 1) / c->last_enc.spf) * c->last_enc.b.n_clips;
    c->last_dense_frames = dense;
    if (c->last_enc_gen == 3) {
      c->encode_dense_frames += dense;
      if ((uint64_t)dense * 4u > frames) c->prefer_gen2 = true;
    } else if (c->prefer_gen2 && (uint64_t)dense * 8u <= frames) {
      c->prefer_gen2 = false;
    }
  }
  if (c->h_status[1] == X3D_SIZE_WAIT_TIMEOUT) {
    if (c->opt.verbose)
      std::fprintf(stderr, "x3hip: size wait gave up: kind %d generation %d wave %d gen %d masks %08x %08x %06x\n", c->h_status[2],
                   c->h_status[3], c->h_status[4] & 0xFF, c->h_status[4] >> 8, c->h_status[5], c->h_status[6], c->h_status[7]);
    // the single-pass kernel's workgroups were not all resident (GPU shared with other work): its
    // bounded wait for frame sizes gave up.  Encode again with the two-pass kernels, which need no residency.
    ++c->encode_fallbacks;
    if (c->opt.verbose)
      std::fprintf(stderr, "x3hip: stream encoder gave up waiting for frame sizes (grid not co-resident): two-pass fallback\n");
    if (c->last_enc_mc) return X3_RETRY_TWO_PASS;   // (several channels: x3_encode_mc runs its own two passes)
    c->force_two_pass = true;
    auto a = c->last_enc;
    c->enc_seg = a.seg;   // (the index, if the call had one: the re-run leaves a header that says "none")
    const X3FrameTable tab{a.src_off, a.src_n, a.src_even};
    int rc = encode_dev_impl(c, a.d_wav, &a.b, &a.p, a.spf, a.d_out, a.out_cap, a.start_pos, a.d_off, a.src_off ? &tab : nullptr);
    c->force_two_pass = false;
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status, 128, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->encode_pending = false;
  }
  if (out_pos) *out_pos = c->h_stats[6];
  if (stats)
    for (int i = 0; i < 6; ++i) stats[i] = c->h_stats[i];
  return std::max(c->h_status[0], c->h_status[1]);
}

// host-buffer front end shared by x3_encode / x3_encode_frame / x3_encode_batch
int encode_host(x3_ctx* c, const int16_t* const* wavs, uint64_t n_per_clip, uint64_t n_clips,
                       const x3_params* p, uint64_t spf, uint8_t* out, uint64_t out_cap, uint64_t start_pos,
                       uint64_t* out_pos, uint64_t* clip_offsets, uint64_t stats[6]) {
  HIPCHK(c, hipSetDevice(c->device));
  // (the clips side by side on the device at a stride of a multiple of eight samples: whatever their length, the batch then
  // takes the single-pass encoders -- x3_encode_dev's "Layout")
  const uint64_t stride = n_clips > 1 ? (n_per_clip + 7) & ~7ull : n_per_clip;
  const uint64_t total = stride * n_clips;
  int rc = ensure(c, c->in, total * sizeof(int16_t) + 16);
  if (rc) return rc;
  for (uint64_t k = 0; k < n_clips; ++k)
    HIPCHK(c, hipMemcpyAsync((int16_t*)c->in.p + k * stride, wavs[k], n_per_clip * sizeof(int16_t),
                             hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  x3_batch b{n_per_clip, stride, n_clips};
  uint64_t bound;
  {
    uint64_t full = n_per_clip / spf, tail = n_per_clip % spf;
    bound = n_clips * (full * (20 + max_payload_bytes(spf, p->block_len)) +
                       (tail ? 20 + max_payload_bytes(tail, p->block_len) : 0));
  }
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  const uint64_t dev_cap = std::min<uint64_t>(out_cap, start_pos + 1 + bound);
  if ((rc = ensure(c, c->out, dev_cap + 16))) return rc;
  uint64_t pos = 0;
  {
    std::unique_lock<std::mutex> gate;
    if (c->enc_gate) {
      HIPCHK(c, hipStreamSynchronize(c->stream));  // the upload is not the gate's business
      gate = std::unique_lock<std::mutex>(*c->enc_gate);
    }
    if ((rc = encode_dev_impl(c, (const int16_t*)c->in.p, &b, &pp, spf, (uint8_t*)c->out.p, out_cap, start_pos, nullptr)))
      return rc;
    rc = x3_encode_result(c, &pos, stats);
  }
  if (rc == X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY) {
    // The reference's slice keeps every frame that fitted (bytewriter.rs:86-99 fails the write that does not fit,
    // encoder.rs:67-73 stops there): the caller gets those frames, complete and in place, *out_pos = the end of the last
    // of them, and nothing behind it is touched.  Every encoder writes a frame only if all of it fits, at the offset the
    // frame index says, so the device buffer holds exactly that prefix; the index says where it ends.  (An error path:
    // one copy of the index, nobody times it.  c->needed_pos keeps the position the whole stream would have reached.)
    c->needed_pos = pos;
    const uint64_t F = ((n_per_clip + spf - 1) / spf) * n_clips;
    std::vector<uint64_t> offs(F + 1);
    HIPCHK(c, hipMemcpyAsync(offs.data(), c->frame_off.p, offs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t end = start_pos, first = start_pos;
    if ((start_pos & 1ull) && start_pos < out_cap) {   // the pad byte in front of the first frame (writer.align, encoder.rs:182)
      out[start_pos] = 0;
      end = first = start_pos + 1;
    }
    for (uint64_t f = 0; f < F; ++f) {
      if (offs[f] != end || offs[f + 1] < offs[f] + 22 || offs[f + 1] > out_cap) break;
      end = offs[f + 1];
    }
    if (end > first)
      HIPCHK(c, hipMemcpy(out + first, (uint8_t*)c->out.p + first, end - first, hipMemcpyDeviceToHost));
    if (out_pos) *out_pos = end;
    return rc;
  }
  if (out_pos) *out_pos = pos;
  if (rc) return rc;
  if (pos > start_pos)
    HIPCHK(c, hipMemcpyAsync(out + start_pos, (uint8_t*)c->out.p + start_pos, pos - start_pos, hipMemcpyDeviceToHost,
                             c->stream));
  if (clip_offsets) {
    const uint64_t fpc = (n_per_clip + spf - 1) / spf;
    std::vector<uint64_t> offs(fpc * n_clips + 1);
    HIPCHK(c, hipMemcpyAsync(offs.data(), c->frame_off.p, offs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t k = 0; k <= n_clips; ++k) clip_offsets[k] = offs[k * fpc];
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

// x3_encode on a LONG host buffer: the same bytes as encode_host, in chunks of whole frames.  The link to the host is
// the whole cost of this entry point (config 3: 1.38 GB up, 0.36 GB down, 0.4 ms of kernel) and it carries both
// directions at once (tools/ubench/pcie_duplex.hip: 24.7 ms for both against 31.0 one after the other), so three host
// threads work side by side -- pageable copies hold their caller: one sends chunk i+2 up, this one encodes chunk i+1
// where chunk i ended (frames do not depend on each other; x3_encode_result's position is all a chunk waits for),
// one brings the bytes of chunk i down.
static int encode_host_chunked(x3_ctx* c, const int16_t* wav, uint64_t n, const x3_params* p, uint64_t spf,
                               uint64_t chunk, uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos,
                               uint64_t stats[6]) {
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->in, n * sizeof(int16_t) + 16);
  if (rc) return rc;
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  {
    const uint64_t full = n / spf, tail = n % spf;
    const uint64_t bound = full * (20 + max_payload_bytes(spf, p->block_len)) + (tail ? 20 + max_payload_bytes(tail, p->block_len) : 0);
    if ((rc = ensure(c, c->out, std::min<uint64_t>(out_cap, start_pos + 1 + bound) + 16))) return rc;
  }
  if ((rc = x3_pipe_streams(c))) return rc;
  struct Piece { uint64_t lo, hi; };
  X3Handoff<Piece> down;
  X3Progress up;
  hipError_t up_err = hipSuccess, dl_err = hipSuccess;
  int16_t* d_in = (int16_t*)c->in.p;
  std::thread uploader, downloader;
  try {
  // (the helper threads never let an exception out -- that would be std::terminate, and x3hip.h promises that the library
  // does not abort: whatever is thrown in them ends the pipeline with an error -- ADVICE r3)
  uploader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      for (uint64_t s0 = 0; s0 < n && e == hipSuccess; s0 += chunk) {
        { std::lock_guard<std::mutex> g(up.mu); if (up.stop) return; }
        e = hipMemcpyAsync(d_in + s0, wav + s0, std::min<uint64_t>(chunk, n - s0) * sizeof(int16_t), hipMemcpyHostToDevice, c->ul_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->ul_stream);
        if (e == hipSuccess) up.advance();
      }
    } catch (...) { e = hipErrorOutOfMemory; }
    if (e != hipSuccess) { up_err = e; up.halt(); }
  });
  downloader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      Piece pc;
      while (down.pop(&pc)) {
        if (e != hipSuccess) continue;  // (drain)
        e = hipMemcpyAsync(out + pc.lo, (const uint8_t*)c->out.p + pc.lo, pc.hi - pc.lo, hipMemcpyDeviceToHost, c->dl_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->dl_stream);
      }
    } catch (...) { e = hipErrorOutOfMemory; }
    dl_err = e;
  });
  } catch (const std::system_error&) {  // no thread to be had: the call goes through in one piece
    up.halt();
    down.close();
    if (uploader.joinable()) uploader.join();
    return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
  }
  x3_params pp = *p;
  uint64_t pos = start_pos, k = 0;
  bool halted = false;
  // (nothing thrown between here and the joins may leave this frame: a joinable std::thread that is destroyed calls
  // std::terminate.  The hand-off's push allocates; everything else reports through return codes.)
  try {
    for (uint64_t s0 = 0; s0 < n && rc == X3_OK; s0 += chunk, ++k) {
      if (!up.wait_for(k + 1)) { halted = true; break; }
      const uint64_t cnt = std::min<uint64_t>(chunk, n - s0);
      x3_batch b{cnt, cnt, 1};
      uint64_t st[6] = {0, 0, 0, 0, 0, 0}, end = pos;
      {
        std::unique_lock<std::mutex> gate;
        if (c->enc_gate) gate = std::unique_lock<std::mutex>(*c->enc_gate);
        rc = encode_dev_impl(c, d_in + s0, &b, &pp, spf, (uint8_t*)c->out.p, out_cap, pos, nullptr);
        if (rc == X3_OK) rc = x3_encode_result(c, &end, st);
      }
      if (rc != X3_OK) break;
      if (stats)
        for (int i = 0; i < 6; ++i) stats[i] += st[i];
      if (end > pos) down.push({pos, end});
      pos = end;
    }
  } catch (...) {
    c->last_error = "x3_encode: out of host memory in the chunked pipeline";
    rc = X3_ERR_HIP;
  }
  up.halt();
  down.close();
  uploader.join();
  downloader.join();
  if (halted) HIPCHK(c, up_err);
  if (rc == X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY) {
    // the position that WOULD have been reached is part of the contract (include/x3hip.h): the sizes of all frames,
    // in one piece (an error path; nobody times it)
    if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
    return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
  }
  if (out_pos) *out_pos = pos;
  if (rc) return rc;
  HIPCHK(c, dl_err);
  return X3_OK;
}

extern "C" int x3_encode(x3_ctx* c, const int16_t* wav, uint64_t n, uint32_t n_channels, const x3_params* p,
                         uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c || !p || (!wav && n) || (!out && out_cap)) return X3_ERR_BAD_ARG;
  if (n_channels > 1) return X3_ERR_MORE_THAN_ONE_CHANNEL;  // encoder.rs:55-57
  if (n_channels == 0) return X3_ERR_BAD_ARG;                // channels[0] panics
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0 || n == 0) return X3_OK;  // take(0) / empty iterator: nothing is written (encoder.rs:67-73)
  if (c->opt.host_chunk_frames >= 0) {
    // chunks of whole frames, a multiple of eight of them (16-byte aligned chunk starts on the device)
    uint64_t frames = c->opt.host_chunk_frames ? (uint64_t)c->opt.host_chunk_frames : (16ull << 20) / spf;
    frames = std::max<uint64_t>(8, (frames + 7) & ~7ull);
    if (frames <= (~0ull >> 1) / spf && n / spf >= 2 * frames)
      return encode_host_chunked(c, wav, n, p, spf, frames * spf, out, out_cap, start_pos, out_pos, stats);
  }
  return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
}

extern "C" int x3_encode_frame(x3_ctx* c, const int16_t* wav, uint64_t n, const x3_params* p, uint8_t* out,
                               uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c || !p || !wav || (!out && out_cap)) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  if (n == 0) return X3_ERR_BAD_ARG;  // wav[0] panics (encoder.rs:189)
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  if (p->block_len == 0 && n > 1) return X3_ERR_BAD_ARG;  // chunks(0) panics
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;
  // the whole slice is ONE frame, whatever blocks_per_frame says
  return encode_host(c, &wav, n, 1, &pp, n, out, out_cap, start_pos, out_pos, nullptr, stats);
}

extern "C" int x3_encode_batch(x3_ctx* c, const int16_t* const* wavs, const uint64_t* ns, uint64_t count,
                               const x3_params* p, uint8_t* out, uint64_t out_cap, uint64_t* clip_offsets,
                               uint64_t stats[6]) {
  if (!c || !p || !wavs || !ns || !count || !clip_offsets) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0) return X3_ERR_BAD_ARG;
  bool uniform = true;
  for (uint64_t k = 1; k < count; ++k) uniform = uniform && ns[k] == ns[0];
  if (uniform && ns[0] > 0) {
    uint64_t pos = 0;
    return encode_host(c, wavs, ns[0], count, p, spf, out, out_cap, 0, &pos, clip_offsets, stats);
  }
  // ragged batch: the clips side by side on the device (each on a multiple of eight samples), every clip cut into frames as
  // encoder::encode cuts it, ONE launch set for the list of all frames (x3_encode_frames_dev; until round 4 a launch set,
  // a wait and two copies per clip)
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint64_t> so, cstart(count), first(count + 1);
  std::vector<uint32_t> sn;
  uint64_t total = 0, bound = 0;
  for (uint64_t k = 0; k < count; ++k) {
    cstart[k] = total;
    first[k] = so.size();
    for (uint64_t s0 = 0; s0 < ns[k]; s0 += spf) {
      so.push_back(total + s0);
      sn.push_back((uint32_t)std::min<uint64_t>(spf, ns[k] - s0));
    }
    bound += (ns[k] / spf) * (20 + max_payload_bytes(spf, p->block_len)) +
             (ns[k] % spf ? 20 + max_payload_bytes(ns[k] % spf, p->block_len) : 0);
    total += (ns[k] + 7) & ~7ull;
  }
  first[count] = so.size();
  clip_offsets[0] = 0;
  if (so.empty()) {
    for (uint64_t k = 0; k < count; ++k) clip_offsets[k + 1] = 0;
    return X3_OK;
  }
  if ((rc = ensure(c, c->in, total * sizeof(int16_t) + 16))) return rc;
  for (uint64_t k = 0; k < count; ++k)
    if (ns[k])
      HIPCHK(c, hipMemcpyAsync((int16_t*)c->in.p + cstart[k], wavs[k], ns[k] * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
  const uint64_t dev_cap = std::min<uint64_t>(out_cap, 1 + bound);
  if ((rc = ensure(c, c->out, dev_cap + 16))) return rc;
  const uint64_t F = so.size();
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  uint64_t pos = 0;
  {
    std::unique_lock<std::mutex> gate;
    if (c->enc_gate) {
      HIPCHK(c, hipStreamSynchronize(c->stream));  // the upload is not the gate's business
      gate = std::unique_lock<std::mutex>(*c->enc_gate);
    }
    if ((rc = x3_encode_frames_dev(c, (const int16_t*)c->in.p, so.data(), sn.data(), F, p, (uint8_t*)c->out.p, dev_cap, 0,
                                   (uint64_t*)c->frame_off.p)))
      return rc;
    rc = x3_encode_result(c, &pos, stats);
  }
  if (rc) return rc;
  if (pos) HIPCHK(c, hipMemcpyAsync(out, c->out.p, pos, hipMemcpyDeviceToHost, c->stream));
  std::vector<uint64_t> offs(F + 1);
  HIPCHK(c, hipMemcpyAsync(offs.data(), c->frame_off.p, offs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (uint64_t k = 0; k <= count; ++k) clip_offsets[k] = first[k] < F ? offs[first[k]] : pos;
  return X3_OK;
}


#ifdef X3_DBG_STAMPS
// stamp builds: the encoders' per-phase clocks (this unit's copy of x3_dbg; tools/scratch/dbg_stamps_wave.py)
extern "C" int x3_dbg_read_enc(x3_ctx* c, unsigned long long* out, uint64_t n) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_dbg), n * sizeof(unsigned long long)));
  return X3_OK;
}
#endif

#define X3_MC_ENCODE
#include "x3_mc.h"
