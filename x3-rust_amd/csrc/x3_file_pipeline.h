// x3_file_pipeline.h -- the file level of the reference: encodefile::wav_to_x3a (encodefile.rs:48-77) and
// decodefile::x3a_to_wav (decodefile.rs:189-227), as a streaming pipeline.
//
// The reference reads the whole WAV through an iterator and pushes the frames through a BufWriter one
// byte at a time; its README lists "allocates everything up front" as a to-do.  Here a file is cut into
// chunks of whole frames (16 MB of samples by default) that move through a small pool of workers, each
// with its own context (stream + device scratch) and its own pinned staging buffers:
//
//   wav -> x3a : pread chunk i into pinned memory | H2D, encode, D2H (x3_encode's path) | pwrite at the
//                position the chunks before it ended at (a turnstile keeps the order; frames are even-sized,
//                so chunks concatenate exactly as the reference's single stream does)
//   x3a -> wav : one worker at a time reads the next window and walks its frame headers (the only serial
//                part: where a chunk ends is where the next begins) | H2D, check + decode, D2H | pwrite at
//                the sample offset the walk already knows
//
// so file reads, PCIe transfers, kernels and file writes of neighbouring chunks overlap, and memory is
// bounded by workers x chunk size whatever the file size.  Bytes on disk are exactly those of the
// in-memory x3_x3a_encode / x3_x3a_decode on the whole file.
//
// WAV container: `hound` 3.4.0 in the reference (Cargo.toml:24, not in its tree); only what the reference
// accepts is supported -- 16-bit integer PCM, one channel (encodefile.rs:53,56 assert both) -- and the
// writer emits hound's canonical 44-byte header for that format.
#pragma once
#include <condition_variable>
#include <fcntl.h>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

struct FilePipeCfg {
  uint64_t chunk_frames;
  int workers;
};
static FilePipeCfg file_pipe_cfg(const x3_ctx* c) {  // options "file_chunk_frames" / "file_workers"
  return FilePipeCfg{(uint64_t)c->opt.file_chunk_frames, c->opt.file_workers};
}

static bool pread_full(int fd, void* buf, uint64_t n, uint64_t off) {
  uint8_t* b = static_cast<uint8_t*>(buf);
  while (n) {
    const ssize_t r = pread(fd, b, n, (off_t)off);
    if (r <= 0) return false;
    b += r; off += (uint64_t)r; n -= (uint64_t)r;
  }
  return true;
}
static bool pwrite_full(int fd, const void* buf, uint64_t n, uint64_t off) {
  const uint8_t* b = static_cast<const uint8_t*>(buf);
  while (n) {
    const ssize_t r = pwrite(fd, b, n, (off_t)off);
    if (r <= 0) return false;
    b += r; off += (uint64_t)r; n -= (uint64_t)r;
  }
  return true;
}

// pinned host buffer that grows on demand
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = (std::max(n, (size_t)4096) + 4095) & ~(size_t)4095;
    if (hipHostMalloc(&p, want) != hipSuccess) { p = nullptr; return false; }
    cap = want;
    return true;
  }
  ~PinBuf() { if (p) (void)hipHostFree(p); }
};

// the worker contexts: the caller's context serves worker 0, the others are created on the same device
struct WorkerCtxs {
  std::vector<x3_ctx*> ctx;
  int rc = X3_OK;
  // `encoders`: the workers encode on one GPU.  The single-pass encoders are persistent grids whose workgroups wait
  // for each other: two of them in flight at once are not both resident, both spin to the bounded-wait limit (~0.1 s)
  // and fall back to the two-pass kernels.  The workers therefore share ONE gate (x3_ctx::enc_gate): a chunk's kernel
  // -- tens of microseconds -- runs alone, the uploads, downloads and file I/O of the other workers run beside it.
  // (Until round 3 the workers were simply put on the two-pass kernels, four times slower.)
  std::mutex gate;
  WorkerCtxs(x3_ctx* c, int n, bool encoders = false) {
    ctx.push_back(c);
    for (int k = 1; k < n; ++k) {
      x3_ctx* w = nullptr;
      rc = x3_ctx_create(c->device, &w);
      if (rc) { c->last_error = "file pipeline: cannot create a worker context"; break; }
      w->opt = c->opt;
      ctx.push_back(w);
    }
    if (encoders && ctx.size() > 1)
      for (x3_ctx* w : ctx) w->enc_gate = &gate;
  }
  ~WorkerCtxs() {
    ctx[0]->enc_gate = nullptr;
    for (size_t k = 1; k < ctx.size(); ++k) {
      ctx[0]->encode_fallbacks += ctx[k]->encode_fallbacks;
      ctx[0]->encode_dense_frames += ctx[k]->encode_dense_frames;
      x3_ctx_destroy(ctx[k]);
    }
  }
};

// ---- RIFF/WAVE (hound::WavReader::new restated: chunks up to "data"; "fmt " must precede it)
struct WavInfo {
  uint32_t sample_rate = 0;
  uint16_t channels = 0, bits = 0;
  uint64_t data_off = 0, data_len = 0;
};
static uint32_t rd_le32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
static uint16_t rd_le16(const uint8_t* b) { return (uint16_t)(b[0] | (b[1] << 8)); }
static void wr_le32(uint8_t* b, uint32_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 24); }
static void wr_le16(uint8_t* b, uint16_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); }

static int wav_parse_fd(int fd, uint64_t file_len, WavInfo* wi) {
  uint8_t h[64];
  if (file_len < 12 || !pread_full(fd, h, 12, 0)) return X3_ERR_IO;
  if (std::memcmp(h, "RIFF", 4) != 0 || std::memcmp(h + 8, "WAVE", 4) != 0) return X3_ERR_BAD_ARG;
  uint64_t pos = 12;
  bool have_fmt = false;
  for (;;) {
    if (file_len - pos < 8 || !pread_full(fd, h, 8, pos)) return X3_ERR_IO;  // no data chunk
    const uint32_t clen = rd_le32(h + 4);
    const bool is_fmt = std::memcmp(h, "fmt ", 4) == 0, is_data = std::memcmp(h, "data", 4) == 0;
    pos += 8;
    if (is_fmt) {
      if (clen < 16) return X3_ERR_BAD_ARG;
      if (file_len - pos < clen) return X3_ERR_IO;
      uint8_t f[40];
      const uint32_t take = clen < 40 ? clen : 40;
      if (!pread_full(fd, f, take, pos)) return X3_ERR_IO;
      uint16_t tag = rd_le16(f);
      wi->channels = rd_le16(f + 2);
      wi->sample_rate = rd_le32(f + 4);
      wi->bits = rd_le16(f + 14);
      if (tag == 0xFFFE) {  // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        if (clen < 40) return X3_ERR_BAD_ARG;
        tag = rd_le16(f + 24);
      }
      if (tag != 1) return X3_ERR_BAD_ARG;  // integer PCM only
      have_fmt = true;
    } else if (is_data) {
      if (!have_fmt) return X3_ERR_BAD_ARG;
      wi->data_off = pos;
      wi->data_len = clen;
      return X3_OK;
    }
    const uint64_t skip = (uint64_t)clen + (clen & 1u);  // chunks are word aligned
    if (file_len - pos < skip) return X3_ERR_IO;
    pos += skip;
  }
}

// hound::WavWriter for {channels 1, 16 bit, Int}: PCMWAVEFORMAT, sizes as finalize() leaves them
static void wav_header(uint32_t sample_rate, uint64_t n_samples, uint8_t out[44]) {
  const uint32_t data_len = (uint32_t)(n_samples * 2);
  std::memcpy(out, "RIFF", 4);
  wr_le32(out + 4, 36 + data_len);
  std::memcpy(out + 8, "WAVEfmt ", 8);
  wr_le32(out + 16, 16);
  wr_le16(out + 20, 1);
  wr_le16(out + 22, 1);
  wr_le32(out + 24, sample_rate);
  wr_le32(out + 28, sample_rate * 2);
  wr_le16(out + 32, 2);
  wr_le16(out + 34, 16);
  std::memcpy(out + 36, "data", 4);
  wr_le32(out + 40, data_len);
}

struct Fd {
  int fd = -1;
  ~Fd() { if (fd >= 0) ::close(fd); }
};

// ------------------------------------------------------------------------------------------------
// wav -> x3a
// ------------------------------------------------------------------------------------------------
extern "C" int x3_wav_to_x3a(x3_ctx* c, const char* wav_path, const char* x3a_path, uint64_t stats[6]) {
  if (!c || !wav_path || !x3a_path) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  HIPCHK(c, hipSetDevice(c->device));
  Fd in, out;
  in.fd = ::open(wav_path, O_RDONLY);
  if (in.fd < 0) { c->last_error = std::string("cannot open ") + wav_path; return X3_ERR_IO; }  // .unwrap() panics
  struct stat sb;
  if (fstat(in.fd, &sb) != 0) return X3_ERR_IO;
  const uint64_t file_len = (uint64_t)sb.st_size;
  WavInfo wi;
  int rc = wav_parse_fd(in.fd, file_len, &wi);
  if (rc) return rc;
  if (wi.bits != 16 || wi.channels != 1) return X3_ERR_BAD_ARG;  // assert_eq! (encodefile.rs:53,56)
  if (wi.data_len & 1u) return X3_ERR_BAD_ARG;                    // hound: not a multiple of the sample size
  out.fd = ::open(x3a_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);  // File::create(..)? (encodefile.rs:66)
  if (out.fd < 0) { c->last_error = std::string("cannot create ") + x3a_path; return X3_ERR_IO; }
  uint64_t n = wi.data_len / 2;
  const uint64_t avail = (file_len - wi.data_off) / 2;
  const bool truncated = avail < n;  // samples().map(|x| x.unwrap()) panics at the first missing sample
  if (truncated) n = avail;

  x3_params p;
  x3_params_default(&p);  // encodefile.rs:57
  uint8_t hdr[512];
  uint64_t hlen = 0;
  if ((rc = x3_archive_header_write(wi.sample_rate, &p, hdr, sizeof hdr, &hlen))) return rc;
  if (!pwrite_full(out.fd, hdr, hlen, 0)) return X3_ERR_IO;

  const FilePipeCfg cfg = file_pipe_cfg(c);
  const uint64_t spf = spf_of(&p);
  const uint64_t chunk_samples = cfg.chunk_frames * spf;
  const uint64_t n_chunks = (n + chunk_samples - 1) / chunk_samples;
  const int n_workers = (int)std::min<uint64_t>((uint64_t)cfg.workers, std::max<uint64_t>(n_chunks, 1));
  WorkerCtxs pool(c, n_workers, true);
  if (pool.rc) return pool.rc;

  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    uint64_t next_chunk = 0, write_turn = 0, write_pos = 0;
    int err = X3_OK;
    uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
  } sh;
  sh.write_pos = hlen;  // even: 8 + 20 + the XML padded to a word (encodefile.rs:123-128)

  auto worker = [&](x3_ctx* w) {
    (void)hipSetDevice(w->device);
    PinBuf pin_in, pin_out;
    for (;;) {
      uint64_t i;
      {
        std::lock_guard<std::mutex> lk(sh.m);
        if (sh.err || sh.next_chunk >= n_chunks) return;
        i = sh.next_chunk++;
      }
      const uint64_t first = i * chunk_samples, cnt = std::min(chunk_samples, n - first);
      const uint64_t bound = x3_encode_bound(cnt, &p);
      int e = X3_OK;
      uint64_t pos = 0, st[6] = {0, 0, 0, 0, 0, 0};
      if (!pin_in.ensure(cnt * 2) || !pin_out.ensure(bound)) {
        e = X3_ERR_HIP;
      } else if (!pread_full(in.fd, pin_in.p, cnt * 2, wi.data_off + first * 2)) {
        e = X3_ERR_IO;
      } else {
        const int16_t* src = static_cast<const int16_t*>(pin_in.p);  // WAV samples are little-endian, as is the host
        e = encode_host(w, &src, cnt, 1, &p, spf, static_cast<uint8_t*>(pin_out.p), bound, 0, &pos, nullptr, st);
        if (e && w != c) c->last_error = w->last_error;
      }
      uint64_t my_pos = 0;
      {
        std::unique_lock<std::mutex> lk(sh.m);
        sh.cv.wait(lk, [&] { return sh.write_turn == i || sh.err; });
        if (e && !sh.err) sh.err = e;
        if (!sh.err) {
          my_pos = sh.write_pos;
          sh.write_pos += pos;
          for (int k = 0; k < 6; ++k) sh.stats[k] += st[k];
        }
        sh.write_turn = i + 1;
        const bool stop = sh.err != X3_OK;
        lk.unlock();
        sh.cv.notify_all();
        if (stop) return;
      }
      if (!pwrite_full(out.fd, pin_out.p, pos, my_pos)) {
        std::lock_guard<std::mutex> lk(sh.m);
        if (!sh.err) sh.err = X3_ERR_IO;
        sh.cv.notify_all();
        return;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (int k = 1; k < n_workers; ++k) th.emplace_back(worker, pool.ctx[k]);
    worker(pool.ctx[0]);
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(c->device);
  if (stats)
    for (int k = 0; k < 6; ++k) stats[k] = sh.stats[k];
  if (sh.err) return sh.err;
  return truncated ? X3_ERR_IO : X3_OK;
}

// ------------------------------------------------------------------------------------------------
// x3a -> wav
// ------------------------------------------------------------------------------------------------
extern "C" int x3_x3a_to_wav(x3_ctx* c, const char* x3a_path, const char* wav_path, uint64_t* n_samples,
                             uint64_t* frame_errors) {
  if (!c || !x3a_path || !wav_path) return X3_ERR_BAD_ARG;
  if (n_samples) *n_samples = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  Fd in, out;
  in.fd = ::open(x3a_path, O_RDONLY);
  if (in.fd < 0) { c->last_error = std::string("cannot open ") + x3a_path; return X3_ERR_IO; }  // .unwrap() panics
  struct stat sb;
  if (fstat(in.fd, &sb) != 0) return X3_ERR_IO;
  const uint64_t file_len = (uint64_t)sb.st_size;
  // X3aReader::open (decodefile.rs:59-75): the archive header comes first, before the writer exists
  x3_params p;
  uint32_t rate = 0;
  uint64_t hsize = 0;
  {
    std::vector<uint8_t> head((size_t)std::min<uint64_t>(file_len, 28 + 0x8000));
    if (!head.empty() && !pread_full(in.fd, head.data(), head.size(), 0)) return X3_ERR_IO;
    uint8_t ch = 0;
    int rc = x3_archive_header_read(head.data(), head.size(), &rate, &p, &ch, &hsize);
    if (rc) return rc;
  }
  out.fd = ::open(wav_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);  // WavWriter::create(..)? (decodefile.rs:201)
  if (out.fd < 0) { c->last_error = std::string("cannot create ") + wav_path; return X3_ERR_HOUND; }
  const uint64_t start = 8 + hsize;
  const uint64_t real_total0 = file_len - start;  // bytes that follow the archive header
  const uint64_t phantom = 8;                     // remaing_bytes = file length - header_size (decodefile.rs:62-66)

  const FilePipeCfg cfg = file_pipe_cfg(c);
  const uint64_t spf = std::max<uint64_t>(spf_of(&p), 1);
  const uint64_t chunk_samples = cfg.chunk_frames * spf;
  // window: the chunk's samples at the worst-case rate plus one maximal frame, so that typical streams fill
  // the sample budget before the window ends
  const uint64_t window = std::max<uint64_t>(chunk_samples / 2, 1u << 20) + 20 + X3_FRAME_MAX_LENGTH;
  const int n_workers = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cfg.workers, real_total0 / window + 1));
  WorkerCtxs pool(c, n_workers);
  if (pool.rc) return pool.rc;

  struct ChunkEnd {  // a chunk whose walk or decode did not simply continue
    uint64_t idx, sample_off, before;
    int rc;
    uint64_t frame_errors;
  };
  struct Shared {
    std::mutex m;
    uint64_t cursor = 0, sample_off = 0, chunk_idx = 0;  // next window, relative to `start`
    bool done = false;
    int err = X3_OK;  // pipeline failures (I/O, HIP), not stream errors
    std::vector<ChunkEnd> ends;
  } sh;

  auto worker = [&](x3_ctx* w) {
    (void)hipSetDevice(w->device);
    PinBuf pin_x3, pin_wav;
    for (;;) {
      HostWalk hw;
      uint64_t idx, sample_off, a;
      {
        // the serial part: read the next window and find where its last whole frame ends
        std::lock_guard<std::mutex> lk(sh.m);
        if (sh.done || sh.err) return;
        a = sh.cursor;
        const uint64_t real_total = real_total0 - a;
        const uint64_t len = std::min(window, real_total);
        if (!pin_x3.ensure(len + 16)) { sh.err = X3_ERR_HIP; return; }
        if (len && !pread_full(in.fd, pin_x3.p, len, start + a)) { sh.err = X3_ERR_IO; return; }
        walk_host(static_cast<const uint8_t*>(pin_x3.p), len, real_total, real_total + phantom, &p, ~0ull,
                  chunk_samples, &hw);
        idx = sh.chunk_idx++;
        sample_off = sh.sample_off;
        sh.cursor += hw.end_pos;
        sh.sample_off += hw.nsamp;
        if (!hw.need_more) sh.done = true;
      }
      uint64_t before = 0, first_bad = 0, ferr = 0;
      int bad_status = 0, rc = X3_OK;
      const uint64_t F = hw.offs.size();
      if (F) {
        if (!pin_wav.ensure((hw.nsamp + 16) * sizeof(int16_t))) rc = X3_ERR_HIP;
        if (!rc)
          rc = decode_frames_host(w, static_cast<const uint8_t*>(pin_x3.p), hw.end_pos, hw, &p,
                                  static_cast<int16_t*>(pin_wav.p), ~0ull, &before, &first_bad, &bad_status);
        if (rc) {
          std::lock_guard<std::mutex> lk(sh.m);
          if (!sh.err) { sh.err = rc; if (w != c) c->last_error = w->last_error; }
          return;
        }
        if (before && !pwrite_full(out.fd, pin_wav.p, before * 2, 44 + sample_off * 2)) {
          std::lock_guard<std::mutex> lk(sh.m);
          if (!sh.err) sh.err = X3_ERR_IO;
          return;
        }
      }
      const int res = walk_result(F, first_bad, bad_status, hw.need_more ? X3_OK : hw.terminal, &ferr);
      if (first_bad < F || !hw.need_more) {
        std::lock_guard<std::mutex> lk(sh.m);
        sh.ends.push_back(ChunkEnd{idx, sample_off, before, res, ferr});
        sh.done = true;  // nothing behind a failed frame is part of the output
        if (first_bad < F) return;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (int k = 1; k < n_workers; ++k) th.emplace_back(worker, pool.ctx[k]);
    worker(pool.ctx[0]);
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(c->device);
  // the walk ends at the EARLIEST chunk that ended it; chunks behind it may have been written already
  uint64_t total = 0, ferr = 0;
  int rc = X3_OK;
  {
    const ChunkEnd* first = nullptr;
    for (const ChunkEnd& e : sh.ends)
      if (!first || e.idx < first->idx) first = &e;
    if (first) {
      total = first->sample_off + first->before;
      rc = first->rc;
      ferr = first->frame_errors;
    }
  }
  // the writer is dropped -- and finalised -- on every path once it exists (hound's Drop)
  uint8_t hdr[44];
  wav_header(rate, total, hdr);
  bool io_ok = pwrite_full(out.fd, hdr, 44, 0);
  io_ok = ftruncate(out.fd, (off_t)(44 + total * 2)) == 0 && io_ok;
  if (n_samples) *n_samples = total;
  if (frame_errors) *frame_errors = ferr;
  if (sh.err) return sh.err;
  if (rc) return rc;
  return io_ok ? X3_OK : X3_ERR_IO;
}
