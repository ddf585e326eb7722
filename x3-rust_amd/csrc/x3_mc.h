// x3_mc.h -- multi-channel extension (included by x3_encode.hip and x3_decode.hip, one half each).
//
// NOT in the reference: encoder::encode returns MoreThanOneChannel for more than one channel (encoder.rs:55-57) and
// read_frame_header refuses a frame whose <Num Channels> is above one (decoder.rs:90-94).  The library keeps that
// behaviour in every entry point that mirrors the reference (x3_encode, x3_decode_stream, ...).  What the format foresees
// is the header's channel count (x3.rs:155-156, encoder.rs:134) and encode_frame's own comment, "pack the data block for
// each channel" (encoder.rs:197); the two entry points here follow that and nothing else:
//
//   frame = header { "x3", source id 1, <Num Channels> = C, samples PER CHANNEL, payload_len, ..., CRCs }
//           payload { first sample of channel 0 .. C-1, 16 bits each;
//                     for every block index: the block of channel 0, 1, .. C-1, each coded exactly like a mono block
//                     (x3_encode_block, encoder.rs:289-315) against its own channel's previous sample;
//                     word_align }
//
// With C = 1 every byte is the reference's (tests/test_gpu_multichannel.py pins the extension to the mono path that
// way); for C > 1 parity is UNPINNED by the reference -- the oracle (oracle/x3_oracle.c, x3o_encode_mc / x3o_decode_stream_mc)
// is the definition, and the tests hold the GPU path against it.  A frame whose payload would pass the 24 KB a reader
// takes (decodefile.rs:118-121) is X3_ERR_FRAME_LENGTH.
//
// GPU work: the general frame encoder (x3_encode_kernel.h, one workgroup per frame, one block per lane -- items in the
// order (block index, channel); one pass with look-back since round 5, two passes as its fallback), the frame check kernel (header + payload CRC, told how many channels a frame must
// announce), and one thread per frame over the reference's own reader for the samples (x3_decode_mc_kernel).  This is
// the generic path, not the tuned mono one: beyond parity, correctness first.
// (no include guard: x3_encode.hip takes the encoder with X3_MC_ENCODE, x3_decode.hip the decoder with X3_MC_DECODE)

#ifdef X3_MC_ENCODE   // (x3_encode.hip)
extern "C" int x3_encode_mc(x3_ctx* c, const int16_t* const* wavs, uint32_t n_ch, uint64_t n, const x3_params* p,
                            uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c || !p || !wavs || (!out && out_cap) || n_ch == 0 || n_ch > X3_MAX_CHANNELS) return X3_ERR_BAD_ARG;
  for (uint32_t k = 0; k < n_ch; ++k)
    if (!wavs[k] && n) return X3_ERR_BAD_ARG;
  // One channel IS the reference's stream: the mono entry point, whatever the geometry (until round 4 this path took
  // one channel itself, with a frame image capped at the 24 KB that only counts for several channels: long dense mono
  // frames -- blocks_per_frame 1 000 of white noise -- came back as BAD_ARG where x3_encode makes the stream: ADVICE r3).
  if (n_ch == 1) return x3_encode(c, wavs[0], n, 1, p, out, out_cap, start_pos, out_pos, stats);
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0 || n == 0) return X3_OK;
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  HIPCHK(c, hipSetDevice(c->device));
  // the planes side by side on the device, 16-byte aligned
  const uint64_t ch_stride = (n + 7) & ~7ull;
  if ((rc = ensure(c, c->in, n_ch * ch_stride * sizeof(int16_t) + 16))) return rc;
  for (uint32_t k = 0; k < n_ch; ++k)
    HIPCHK(c, hipMemcpyAsync((int16_t*)c->in.p + k * ch_stride, wavs[k], n * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
  EncPlan pl;
  x3_batch b{n, n, 1};
  if ((rc = plan_encode(c, &b, p, spf, &pl))) return rc;
  // the plan of one channel, times n_ch: sample rows, frame image, lanes
  const uint64_t nmax = std::min<uint64_t>(spf, n);
  const uint64_t items = ((nmax - 1 + p->block_len - 1) / p->block_len) * n_ch;
  const uint32_t nthr = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(64, (items + 63) & ~63ull));
  // one channel: the sample row in LDS as x3_encode has it; several: the blocks read global memory, LDS holds the frame
  // image only -- and no image beyond the 24 KB a payload may have (the size pass refuses longer frames: FrameLength)
  const uint64_t in_bytes = n_ch == 1 ? (uint64_t)pl.lds_in_bytes : 16;
  const uint64_t payload_max = n_ch * max_payload_bytes(nmax, p->block_len);
  const uint64_t img_dw = ((5 + (std::min<uint64_t>(payload_max, 24576 + 64) + 3) / 4 + 4) + 3) & ~3ull;
  const uint64_t smem = X3_ENC_SMEM_HDR + in_bytes + img_dw * 4;
  const uint64_t F = pl.g.n_frames;
  const uint64_t bound = F * (20 + payload_max);
  const uint64_t dev_cap = std::min<uint64_t>(out_cap, start_pos + 1 + bound);
  if ((rc = ensure(c, c->out, dev_cap + 16))) return rc;
  if ((rc = ensure(c, c->frame_bytes, F * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  uint64_t* d_off = (uint64_t*)c->frame_off.p;
  const int16_t* d_wav = (const int16_t*)c->in.p;
  uint64_t pos = 0;
  // One pass with decoupled look-back (x3_encode_kernel.h, LOOKBACK: the kernel of the mono general path since round 4,
  // taking several channels since round 5); the two passes -- sizes, scan, emission -- are what a launch whose look-back
  // gave up falls back to, and option two_pass.
  for (int attempt = (c->opt.two_pass || c->force_two_pass) ? 1 : 0; attempt < 2; ++attempt) {
    const bool one_pass = attempt == 0;
    if (int rc_ctl = ctl_begin(c)) return rc_ctl;
    if (one_pass) {
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(smem, 64 * 1024)));
      const size_t lb_bytes = F * sizeof(unsigned long long);
      const bool fresh = c->lb_desc.cap < lb_bytes;
      if ((rc = ensure(c, c->lb_desc, lb_bytes))) return rc;
      if (fresh || c->capturing || ++c->lb_epoch > 0xFFFu) {
        HIPCHK(c, hipMemsetAsync(c->lb_desc.p, 0, c->lb_desc.cap, c->stream));
        c->lb_epoch = 1;
      }
      hipLaunchKernelGGL((x3_encode_frames_kernel<false, true>), dim3((unsigned)F), dim3(nthr), smem, c->stream, d_wav, pl.g,
                         pl.dp, (const uint64_t*)d_off, (uint32_t*)c->lb_desc.p, (uint8_t*)c->out.p, start_pos, c->d_stats,
                         c->d_status, (const uint16_t*)c->d_xpow, (uint32_t)in_bytes, (uint32_t)img_dw, n_ch, ch_stride,
                         c->lb_epoch, dev_cap, c->d_end_pos, c->opt.lb_drop >= 0 ? (uint32_t)c->opt.lb_drop : 0xFFFFFFFFu);
    } else {
      if (smem > 64 * 1024) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      }
      hipLaunchKernelGGL(x3_encode_frames_kernel<true>, dim3((unsigned)F), dim3(nthr), X3_ENC_SMEM_HDR + in_bytes, c->stream,
                         d_wav, pl.g, pl.dp, (const uint64_t*)nullptr, (uint32_t*)c->frame_bytes.p, (uint8_t*)nullptr, start_pos,
                         c->d_stats, c->d_status, (const uint16_t*)c->d_xpow, (uint32_t)in_bytes, 0u, n_ch, ch_stride);
      hipLaunchKernelGGL(x3_scan_frame_offsets_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)c->frame_bytes.p, F,
                         start_pos, out_cap, d_off, c->d_end_pos, c->d_status);
      hipLaunchKernelGGL(x3_encode_frames_kernel<false>, dim3((unsigned)F), dim3(nthr), smem, c->stream, d_wav, pl.g, pl.dp,
                         (const uint64_t*)d_off, (uint32_t*)nullptr, (uint8_t*)c->out.p, start_pos, c->d_stats, c->d_status,
                         (const uint16_t*)c->d_xpow, (uint32_t)in_bytes, (uint32_t)img_dw, n_ch, ch_stride);
    }
    HIPCHK(c, hipGetLastError());
    x3_batch all{n * n_ch, n * n_ch, 1};
    c->last_enc = {d_wav, all, *p, spf, (uint8_t*)c->out.p, out_cap, start_pos, nullptr};
    c->last_enc_gen = one_pass ? 1 : 0;
    c->last_enc_mc = true;   // (x3_encode_result: a look-back that gave up is re-run HERE, not as a mono call)
    c->encode_pending = true;
    c->enc_start_pos = start_pos;
    rc = x3_encode_result(c, &pos, stats);
    c->last_enc_mc = false;
    if (rc != X3_RETRY_TWO_PASS) break;
    if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  }
  if (out_pos) *out_pos = pos;
  if (rc) return rc;
  if (pos > start_pos)
    HIPCHK(c, hipMemcpyAsync(out + start_pos, (uint8_t*)c->out.p + start_pos, pos - start_pos, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

#endif  // X3_MC_ENCODE

#ifdef X3_MC_DECODE   // (x3_decode.hip)
// the frame walk of x3_decode_stream for frames that announce n_ch channels: wavs[k][0 .. *n_samples) per channel
extern "C" int x3_decode_stream_mc(x3_ctx* c, const uint8_t* x3, uint64_t len, uint32_t n_ch, const x3_params* p,
                                   int16_t* const* wavs, uint64_t wav_cap, uint64_t* n_samples, uint64_t* frames_ok,
                                   uint64_t* frame_errors) {
  if (!c || !p || (!x3 && len) || !wavs || n_ch == 0 || n_ch > X3_MAX_CHANNELS) return X3_ERR_BAD_ARG;
  for (uint32_t k = 0; k < n_ch; ++k)
    if (!wavs[k] && wav_cap) return X3_ERR_BAD_ARG;
  if (n_samples) *n_samples = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  HostWalk w;
  walk_host(x3, len, len, len, p, wav_cap, ~0ull, &w, n_ch);
  const uint64_t F = w.offs.size();
  if (F == 0) return w.terminal;
  X3DevParams dp;
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;  // (frames that need block_len were routed to BAD_ARG by the walk)
  int rc = derive(&pp, 0, &dp);
  if (rc) return rc;
  const uint64_t ch_stride = (w.nsamp + 65536 + 7) & ~7ull;
  if ((rc = ensure(c, c->in, len + 16))) return rc;
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, F * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, n_ch * ch_stride * sizeof(int16_t)))) return rc;
  if ((rc = ensure(c, c->dec_cstatus, F * sizeof(int32_t)))) return rc;
  if ((rc = ensure(c, c->dec_status, F * sizeof(int32_t)))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->in.p, x3, len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, w.offs.data(), F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, w.woffs.data(), F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  const uint64_t check_grid = std::min<uint64_t>((F + 3) / 4, (uint64_t)c->n_cus * 8);
  hipLaunchKernelGGL(x3_frame_check_kernel, dim3((unsigned)check_grid), dim3(256), 0, c->stream,
                     reinterpret_cast<const uint32_t*>(c->in.p), len, (const uint64_t*)c->frame_off.p, F,
                     (const uint16_t*)c->d_xinv8, (const uint16_t*)c->d_chktab, (const uint32_t*)c->d_kx64,
                     (int32_t*)c->dec_cstatus.p, reinterpret_cast<unsigned long long*>(c->d_summary), n_ch);
  {
    // a frame per LANE (x3_decode_mc_kernel.h; until round 4: one thread per frame over the byte-wise reader), then the
    // reference's reader for the frames it has flagged.  Block lengths the ring cannot keep ahead of (> 60: the
    // reference's own limit is 60, encoder.rs:296-299) and option "mc_decode_threads" take the old path for all frames.
    const bool lanes = !c->opt.mc_decode_threads && pp.block_len <= 60;
    TimerScope ts(c, 1);
    if (lanes)
      hipLaunchKernelGGL(x3_decode_mc_lanes_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, c->stream, (const uint8_t*)c->in.p,
                         (const uint64_t*)c->frame_off.p, (const uint64_t*)c->wav_off.p, F, dp, n_ch, (int16_t*)c->out.p, ch_stride,
                         std::min<uint64_t>(wav_cap, w.nsamp + 65535), (const int32_t*)c->dec_cstatus.p, (int32_t*)c->dec_status.p);
    hipLaunchKernelGGL(x3_decode_mc_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, c->stream, (const uint8_t*)c->in.p,
                       (const uint64_t*)c->frame_off.p, (const uint64_t*)c->wav_off.p, F, dp, n_ch, (int16_t*)c->out.p, ch_stride,
                       std::min<uint64_t>(wav_cap, w.nsamp + 65535), (const int32_t*)c->dec_cstatus.p, (int32_t*)c->dec_status.p,
                       lanes ? 1u : 0u);
  }
  HIPCHK(c, hipGetLastError());
  std::vector<int32_t> cst(F), dst(F);
  HIPCHK(c, hipMemcpyAsync(cst.data(), c->dec_cstatus.p, F * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(dst.data(), c->dec_status.p, F * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // the first frame that fails ends the walk: header / payload CRC errors come first (decodefile.rs:96-100)
  uint64_t first_bad = F, before = w.nsamp;
  int bad_status = 0;
  for (uint64_t f = 0; f < F; ++f) {
    const int st = cst[f] ? cst[f] : dst[f];
    if (st) {
      first_bad = f;
      bad_status = st;
      before = w.woffs[f];
      break;
    }
  }
  for (uint32_t k = 0; k < n_ch && before; ++k)
    HIPCHK(c, hipMemcpyAsync(wavs[k], (int16_t*)c->out.p + k * ch_stride, before * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_samples) *n_samples = before;
  if (frames_ok) *frames_ok = first_bad;
  return walk_result(F, first_bad, bad_status, w.terminal, frame_errors);
}
#endif  // X3_MC_DECODE
