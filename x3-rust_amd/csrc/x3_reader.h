// x3_reader.h -- `X3aReader` (src/decodefile.rs:47-137): open / spec / decode_next_frame, one frame per call.
//
// The reference reads a 20-byte header, then the payload, checks its CRC and decodes it, per call.  Done that
// way through a GPU, every call would be a dispatch of its own (a few copies, two kernels, two synchronisations
// for 10 000 samples).  Here a call that finds nothing prepared reads a WINDOW of the stream ahead (up to
// X3R_WINDOW_FRAMES frames -- option "reader_window_frames" -- / X3R_WINDOW_BYTES bytes), walks its header chain, checks and decodes all of its
// frames in one launch set, and keeps their samples and per-frame status in pinned host memory; the following
// calls are a header parse and a memcpy.  What a call returns, what it consumes from the stream and what it
// counts are the reference's, frame by frame -- including its habit of going on behind a frame that failed:
//   remaining <= 20                          -> Ok(None)                              (:107-109)
//   header does not validate                 -> Err(..), 20 bytes consumed            (:112, decoder.rs:69-118)
//   payload longer than what remains         -> Ok(None), 20 bytes consumed           (:114-116)
//   payload longer than the read buffer      -> Err(FrameHeaderInvalidPayloadLen)     (:118-121)
//   payload CRC                              -> Err(FrameHeaderInvalidPayloadCRC), header + payload consumed (:96-100)
//   decode error                             -> Ok(None), frame_errors += 1           (:128-135)
//   a read behind the real end of the data   -> Err(Io)                               (read_exact)
// "remaining" is what the reader BELIEVES remains: X3aReader::open subtracts the archive header without its
// 8-byte id from the file length (:61-66), so it believes in 8 bytes that do not exist.
#pragma once

#define X3R_WINDOW_FRAMES 4096u
#define X3R_WINDOW_BYTES (48u << 20)

struct x3_reader {
  x3_ctx* c = nullptr;
  int fd = -1;                    // file variant
  const uint8_t* mem = nullptr;   // memory variant: the audio frames (behind the archive header)
  uint64_t start = 0;             // file offset of the first audio frame
  uint64_t real_total = 0;        // bytes that really follow the archive header
  uint64_t pos = 0;               // bytes consumed of them
  uint64_t remaining = 0;         // what the reader believes remains
  x3_params p;
  uint32_t rate = 0;
  uint8_t channels = 0;
  uint64_t frame_errors = 0;
  // the window that has been decoded ahead
  uint64_t w_pos = 0;             // stream position of its first frame
  std::vector<uint64_t> w_off, w_woff;  // per frame: offset from w_pos, sample offset in w_samples
  std::vector<int32_t> w_status;
  size_t w_next = 0;              // first frame of the window that has not been handed out
  PinBuf w_bytes, w_samples, w_stat;
  uint64_t windows = 0;           // (statistics) windows decoded
};

static int reader_fetch(x3_reader* r, uint64_t at, void* dst, uint64_t n) {  // n bytes of the stream at `at`
  if (r->mem) {
    std::memcpy(dst, r->mem + at, n);
    return X3_OK;
  }
  return pread_full(r->fd, dst, n, r->start + at) ? X3_OK : X3_ERR_IO;
}

// decode a window of frames starting at r->pos
static int reader_fill(x3_reader* r) {
  x3_ctx* c = r->c;
  r->w_off.clear();
  r->w_woff.clear();
  r->w_status.clear();
  r->w_next = 0;
  r->w_pos = r->pos;
  const uint64_t left = r->real_total - r->pos;
  const uint64_t len = std::min<uint64_t>(left, X3R_WINDOW_BYTES);
  if (len <= 20) return X3_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const uint8_t* bytes;
  if (r->mem) {
    bytes = r->mem + r->pos;
  } else {
    if (!r->w_bytes.ensure(len + 16)) return X3_ERR_HIP;
    if (!pread_full(r->fd, r->w_bytes.p, len, r->start + r->pos)) return X3_ERR_IO;
    bytes = static_cast<const uint8_t*>(r->w_bytes.p);
  }
  HostWalk hw;
  // the frames the walk steps over from here; it stops where a call will have to look for itself (a header that
  // does not validate, the end of the window, ...).  max_samples bounds the window's sample cache.
  walk_host(bytes, len, left, r->remaining, &r->p, ~0ull, (uint64_t)X3R_WINDOW_FRAMES * 65535ull, &hw);
  size_t F = hw.offs.size();
  const size_t max_frames = (size_t)std::max(1ll, c->opt.reader_window_frames);
  if (F > max_frames) {
    F = max_frames;
    hw.nsamp = hw.woffs[F];
    hw.offs.resize(F);
    hw.woffs.resize(F);
  }
  if (F == 0) return X3_OK;
  // bytes of the F frames: up to the end of the last one
  x3_frame_header hl;
  if (x3_read_frame_header(bytes + hw.offs[F - 1], 20, &hl)) return X3_ERR_BAD_ARG;  // (validated by the walk)
  const uint64_t span = hw.offs[F - 1] + 20 + hl.payload_len;
  int rc;
  if ((rc = ensure(c, c->in, span + 16))) return rc;
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, F * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, (hw.nsamp + 65536) * sizeof(int16_t)))) return rc;
  if ((rc = ensure(c, c->dec_status, F * sizeof(int32_t)))) return rc;
  if (!r->w_samples.ensure((hw.nsamp + 16) * sizeof(int16_t)) || !r->w_stat.ensure(F * sizeof(int32_t))) return X3_ERR_HIP;
  HIPCHK(c, hipMemcpyAsync(c->in.p, bytes, span, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, hw.offs.data(), F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, hw.woffs.data(), F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  x3_params pp = r->p;
  if (pp.block_len == 0) pp.block_len = 1;  // (frames that need block_len are BAD_ARG frames of the walk)
  bool aligned = true;
  for (size_t i = 0; i < F; ++i) aligned = aligned && (hw.woffs[i] & 3ull) == 0;
  if ((rc = decode_dev_impl(c, (const uint8_t*)c->in.p, span, (const uint64_t*)c->frame_off.p, F, nullptr,
                            (const uint64_t*)c->wav_off.p, &pp, (int16_t*)c->out.p, hw.nsamp + 65535,
                            (int32_t*)c->dec_status.p, aligned, r->p.block_len == 0)))
    return rc;
  HIPCHK(c, hipMemcpyAsync(r->w_stat.p, c->dec_status.p, F * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (hw.nsamp)
    HIPCHK(c, hipMemcpyAsync(r->w_samples.p, c->out.p, hw.nsamp * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  uint64_t fb, before;
  int bs;
  if ((rc = x3_decode_result(c, &fb, &bs, &before))) return rc;  // (also waits for the copies)
  r->w_off = hw.offs;
  r->w_woff = hw.woffs;
  r->w_status.assign(static_cast<const int32_t*>(r->w_stat.p), static_cast<const int32_t*>(r->w_stat.p) + F);
  ++r->windows;
  return X3_OK;
}

static int reader_open_common(x3_reader* r, const uint8_t* head, uint64_t head_len, uint64_t file_len) {
  uint64_t hsize = 0;
  int rc = x3_archive_header_read(head, head_len, &r->rate, &r->p, &r->channels, &hsize);
  if (rc) return rc;
  r->start = 8 + hsize;
  r->real_total = file_len - r->start;
  r->remaining = file_len - hsize;  // decodefile.rs:61-66: the 8-byte id is not subtracted
  r->pos = 0;
  return X3_OK;
}

extern "C" void x3_reader_close(x3_reader* r) {
  if (!r) return;
  if (r->c) (void)hipSetDevice(r->c->device);
  if (r->fd >= 0) ::close(r->fd);
  delete r;
}

extern "C" int x3_reader_open(x3_ctx* c, const char* path, x3_reader** out) {
  if (!c || !path || !out) return X3_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  x3_reader* r = new x3_reader();
  r->c = c;
  r->fd = ::open(path, O_RDONLY);
  if (r->fd < 0) {  // File::open(..).unwrap() panics
    c->last_error = std::string("cannot open ") + path;
    delete r;
    return X3_ERR_IO;
  }
  struct stat sb;
  if (fstat(r->fd, &sb) != 0) { x3_reader_close(r); return X3_ERR_IO; }
  const uint64_t file_len = (uint64_t)sb.st_size;
  std::vector<uint8_t> head((size_t)std::min<uint64_t>(file_len, 28 + 0x8000));
  if (!head.empty() && !pread_full(r->fd, head.data(), head.size(), 0)) { x3_reader_close(r); return X3_ERR_IO; }
  const int rc = reader_open_common(r, head.data(), head.size(), file_len);
  if (rc) { x3_reader_close(r); return rc; }
  *out = r;
  return X3_OK;
}

extern "C" int x3_reader_open_mem(x3_ctx* c, const uint8_t* x3a, uint64_t len, x3_reader** out) {
  if (!c || (!x3a && len) || !out) return X3_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  x3_reader* r = new x3_reader();
  r->c = c;
  const int rc = reader_open_common(r, x3a, len, len);
  if (rc) { delete r; return rc; }
  r->mem = x3a + r->start;
  *out = r;
  return X3_OK;
}

extern "C" int x3_reader_spec(const x3_reader* r, uint32_t* sample_rate, x3_params* p, uint8_t* channels) {
  if (!r) return X3_ERR_BAD_ARG;
  if (sample_rate) *sample_rate = r->rate;
  if (p) *p = r->p;
  if (channels) *channels = r->channels;
  return X3_OK;
}

extern "C" uint64_t x3_reader_frame_errors(const x3_reader* r) { return r ? r->frame_errors : 0; }
extern "C" uint64_t x3_reader_position(const x3_reader* r) { return r ? r->start + r->pos : 0; }

extern "C" int x3_reader_next_frame(x3_reader* r, int16_t* wav, uint64_t wav_cap, uint64_t* n_out) {
  if (!r || (!wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (r->remaining <= 20) return X3_OK;  // end of the file
  // ---- header (read_bytes clamps to what is believed to remain; read_exact fails behind the real end)
  if (r->real_total - r->pos < 20) {
    r->remaining -= 20;
    r->pos = r->real_total;
    return X3_ERR_IO;
  }
  uint8_t hb[20];
  int rc = reader_fetch(r, r->pos, hb, 20);
  if (rc) return rc;
  const uint64_t frame_pos = r->pos;
  r->pos += 20;
  r->remaining -= 20;
  x3_frame_header h;
  if ((rc = x3_read_frame_header(hb, 20, &h))) return rc;
  if (r->remaining < h.payload_len) return X3_OK;  // Ok(None)
  if (h.payload_len > X3_READ_BUFFER_SIZE) return X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN;
  // ---- payload
  if (r->real_total - r->pos < h.payload_len) {
    r->remaining -= h.payload_len;
    r->pos = r->real_total;
    return X3_ERR_IO;
  }
  r->pos += h.payload_len;
  r->remaining -= h.payload_len;
  // ---- CRC + decode: prepared ahead, a window at a time
  if (!(r->w_next < r->w_off.size() && r->w_pos + r->w_off[r->w_next] == frame_pos)) {
    // nothing prepared for this position (first call, end of the window, or the caller went on behind an error
    // that took the chain somewhere else): decode a window from here
    const uint64_t keep_pos = r->pos, keep_rem = r->remaining;
    r->pos = frame_pos;
    r->remaining = keep_rem + 20 + h.payload_len;
    rc = reader_fill(r);
    r->pos = keep_pos;
    r->remaining = keep_rem;
    if (rc) return rc;
    if (r->w_off.empty() || r->w_off[0] != 0) {
      r->c->last_error = "x3_reader: the window walk does not start at the frame";
      return X3_ERR_BAD_ARG;
    }
  }
  const size_t i = r->w_next++;
  const int st = r->w_status[i];
  if (st == X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC || st == X3_ERR_BAD_ARG || st == X3_ERR_HIP) return st;
  if (st != X3_OK) {  // decode error: counted, Ok(None) (decodefile.rs:129-135)
    r->frame_errors += 1;
    return X3_OK;
  }
  if (h.samples > wav_cap) return X3_ERR_BAD_ARG;  // (the reference's buffer holds any frame)
  std::memcpy(wav, static_cast<const int16_t*>(r->w_samples.p) + r->w_woff[i], (size_t)h.samples * sizeof(int16_t));
  if (n_out) *n_out = h.samples;
  return X3_OK;
}


// ------------------------------------------------------------------------------------------------ x3_decode_prefetch
// `decoder::decode_frame` (src/decoder.rs:36-58) takes one payload per call; a loop over the frames of a stream is then
// one dispatch per 10 000 samples.  A caller that has the whole stream in memory can announce it: x3_decode_frame calls
// whose payload lies in the announced buffer are served from windows that are walked, checked and decoded ahead (the
// reader's machinery over the frame stream, without an archive header).  A frame that the window did not decode
// cleanly -- a payload CRC that does not match its header, which decode_frame does not look at; a decode error -- or
// that is called with another length or sample count than its header says falls through to the per-call path, so the
// results are those of x3_decode_frame without the announcement.
extern "C" int x3_decode_prefetch(x3_ctx* c, const uint8_t* x3, uint64_t len, const x3_params* p) {
  if (!c || (x3 && !p)) return X3_ERR_BAD_ARG;
  if (c->fcache) {
    x3_reader_close(c->fcache);
    c->fcache = nullptr;
  }
  if (!x3 || len <= 20) return X3_OK;
  x3_reader* r = new x3_reader();
  r->c = c;
  r->mem = x3;
  r->start = 0;
  r->real_total = len;
  r->remaining = len;
  r->p = *p;
  c->fcache = r;
  return X3_OK;
}

int frame_cache_serve(x3_ctx* c, const uint8_t* payload, uint64_t len, const x3_params* p, uint64_t samples, int16_t* wav) {
  x3_reader* r = c->fcache;
  if (std::memcmp(p, &r->p, sizeof(x3_params)) != 0) return X3_FRAME_CACHE_MISS;
  if (payload < r->mem + 20 || payload + len > r->mem + r->real_total) return X3_FRAME_CACHE_MISS;
  const uint64_t fpos = (uint64_t)(payload - r->mem) - 20;
  auto find = [&]() -> long {
    if (fpos < r->w_pos || r->w_off.empty()) return -1;
    const uint64_t rel = fpos - r->w_pos;
    auto it = std::lower_bound(r->w_off.begin(), r->w_off.end(), rel);
    return (it != r->w_off.end() && *it == rel) ? (long)(it - r->w_off.begin()) : -1;
  };
  long i = find();
  if (i < 0) {
    x3_frame_header h;  // only a position that holds a valid header of this very frame starts a window
    if (x3_read_frame_header(r->mem + fpos, 20, &h) || h.payload_len != len || h.samples != samples) return X3_FRAME_CACHE_MISS;
    r->pos = fpos;
    r->remaining = r->real_total - fpos;
    const int rc = reader_fill(r);
    if (rc) return rc;
    i = find();
    if (i < 0) return X3_FRAME_CACHE_MISS;
  }
  if (r->w_status[(size_t)i] != 0) return X3_FRAME_CACHE_MISS;
  x3_frame_header h;
  if (x3_read_frame_header(r->mem + fpos, 20, &h) || h.payload_len != len || h.samples != samples) return X3_FRAME_CACHE_MISS;
  std::memcpy(wav, static_cast<const int16_t*>(r->w_samples.p) + r->w_woff[(size_t)i], samples * sizeof(int16_t));
  return X3_OK;
}
