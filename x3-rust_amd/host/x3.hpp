// x3.hpp -- host-side mirror of the psiphi75/x3-rust public API on top of the C ABI (include/x3hip.h).
//
// The reference is a Rust crate; this image has no Rust toolchain, so the host layer above the
// C ABI is written in C++ with the reference's module / item names and argument meaning:
//
//   reference (Rust)                                   here (C++)
//   ------------------------------------------------   ---------------------------------------------
//   x3::Parameters::{default,new}     x3.rs:81-134      x3::Parameters{}, x3::Parameters::create()
//   x3::Channel / x3::IterChannel     x3.rs:29-69       x3::Channel, x3::IterChannel<It>
//   x3::FrameHeader / Frame / Archive x3.rs:136-184     x3::FrameHeader, x3::Frame, x3::Archive
//   error::X3Error                    error.rs:27-62    x3::X3Error (same variant order)
//   bytewriter::{ByteWriter,SliceByteWriter,StreamByteWriter}  bytewriter.rs:14-165   x3::bytewriter::*
//   crc::{crc16,update_crc16}         crc.rs:44-58      x3::crc::*
//   encoder::{encode,encode_frame,write_frame_header}  encoder.rs:51-214   x3::encoder::*
//   decoder::{read_frame_header,decode_frame}          decoder.rs:36-118   x3::decoder::*
//   x3::RiceCode / RiceCodes::get     x3.rs:187-260     x3::RiceCode, x3::RiceCodes::get, Parameters::rice_codes
//   decoder::decode_block             decoder.rs:132-145     x3::decoder::decode_block
//   bitreader::BitReader              bitreader.rs:51-176    x3::bitreader::BitReader
//   bitpacker::BitPacker              bitpacker.rs:46-190    x3::bitpacker::BitPacker
//   decodefile::X3aReader             decodefile.rs:47-136   x3::decodefile::X3aReader
//   X3aReader::decode_next_frame loop decodefile.rs:105-136,200-209        x3::decoder::decode_stream
//
// Every function exists twice: with the reference's own argument list (it then runs on x3::default_context(), a
// process-wide context on device $X3HIP_DEVICE or 0, created on first use and serialized by a mutex), and with a
// leading `Context&` for callers that manage devices and streams themselves.
//
// Every bulk call goes to libx3hip.so (HIP kernels on the MI355X); nothing here computes on the CPU
// beyond argument marshalling.  BitPacker / BitReader are handles to device state: their bit work runs in
// kernels too (x3_bits.h) -- they exist for callers of the reference's building blocks, the frame kernels
// do not go through them.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <array>
#include <cstdlib>
#include <iterator>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/x3hip.h"

namespace x3 {

// error.rs:27-62 (Ok added as 0; Hip / BadArg appended for the C ABI)
enum class X3Error : int {
  Ok = 0, Io, Hound, BitPack, InvalidEncodingThresh, OutOfBoundsInverse, MoreThanOneChannel,
  ArchiveHeaderXMLInvalid, ArchiveHeaderXMLRiceCode, ArchiveHeaderXMLInvalidKey, FrameLength,
  FrameHeaderInvalidKey, FrameHeaderInvalidPayloadLen, FrameHeaderInvalidHeaderCRC, FrameHeaderInvalidPayloadCRC,
  FrameDecodeInvalidBlockLength, FrameDecodeInvalidIndex, FrameDecodeInvalidNTOGO, FrameDecodeInvalidFType,
  FrameDecodeInvalidRiceCode, FrameDecodeInvalidBPF, FrameDecodeUnexpectedEnd, ByteWriterInsufficientMemory,
  Hip, BadArg
};
inline const char* to_string(X3Error e) { return x3_strerror(static_cast<int>(e)); }

// One GPU + stream + scratch (x3_ctx).  Creation throws when no HIP device is usable: there is no CPU path.
class Context {
 public:
  explicit Context(int device = 0) {
    int rc = x3_ctx_create(device, &ctx_);
    if (rc) throw std::runtime_error("x3::Context: no usable HIP device (libx3hip has no CPU fallback)");
  }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  ~Context() { x3_ctx_destroy(ctx_); }
  x3_ctx* raw() const { return ctx_; }

 private:
  x3_ctx* ctx_ = nullptr;
};

// The context behind the reference-shaped overloads (those without a Context& argument).  One per process, made on
// first use; `lock()` serializes callers, a context is not re-entrant.
inline Context& default_context() {
  static Context ctx([] {
    const char* e = std::getenv("X3HIP_DEVICE");
    return e ? std::atoi(e) : 0;
  }());
  return ctx;
}
inline std::mutex& default_context_mutex() {
  static std::mutex m;
  return m;
}
struct MaybeLock {  // locks the default context's mutex for handles that were made on it
  explicit MaybeLock(std::mutex* m) : m_(m) { if (m_) m_->lock(); }
  ~MaybeLock() { if (m_) m_->unlock(); }
  MaybeLock(const MaybeLock&) = delete;
  MaybeLock& operator=(const MaybeLock&) = delete;
  std::mutex* m_;
};
#define X3_WITH_DEFAULT_CONTEXT(expr)                               \
  do {                                                              \
    std::lock_guard<std::mutex> lk_(::x3::default_context_mutex()); \
    ::x3::Context& ctx = ::x3::default_context();                   \
    return (expr);                                                  \
  } while (0)

// x3.rs:187-194.  The tables live in the library (x3_rice_code_get); `code`/`num_bits` have `len` entries.
struct RiceCode {
  size_t nsubs = 0, offset = 0;
  const uint32_t* code = nullptr;
  const uint32_t* num_bits = nullptr;
  const int16_t* inv = nullptr;
  size_t inv_len = 0;
  size_t len = 0;  // entries in code / num_bits (the slices' .len() in the reference)
};
// x3.rs:196-260
struct RiceCodes {
  static const RiceCode* code(size_t k) {
    static const std::array<RiceCode, 4> t = [] {
      std::array<RiceCode, 4> a;
      for (uint32_t i = 0; i < 4; ++i) {
        x3_rice_code c;
        x3_rice_code_get(i, &c);
        a[i].nsubs = c.nsubs; a[i].offset = c.offset; a[i].code = c.code; a[i].num_bits = c.num_bits;
        a[i].inv = c.inv; a[i].inv_len = c.inv_len; a[i].len = c.len;
      }
      return a;
    }();
    if (k > 3) throw std::out_of_range("RiceCodes::get: code number > 3 (the reference panics on the index)");
    return &t[k];
  }
  static std::array<const RiceCode*, 3> get(const size_t (&code_list)[3]) {
    return {code(code_list[0]), code(code_list[1]), code(code_list[2])};
  }
};

// x3.rs:81-134
struct Parameters {
  static constexpr size_t MAX_BLOCK_LENGTH = 60;
  static constexpr size_t WAV_BIT_SIZE = 16;
  static constexpr size_t DEFAULT_BLOCK_LENGTH = 20;
  static constexpr size_t DEFAULT_BLOCKS_PER_FRAME = 500;
  size_t block_len = DEFAULT_BLOCK_LENGTH;
  size_t blocks_per_frame = DEFAULT_BLOCKS_PER_FRAME;
  size_t codes[3] = {0, 1, 3};
  size_t thresholds[3] = {3, 8, 20};
  std::array<const RiceCode*, 3> rice_codes = {RiceCodes::code(0), RiceCodes::code(1), RiceCodes::code(3)};

  // Parameters::new (x3.rs:98-122)
  static X3Error create(size_t block_len, size_t blocks_per_frame, const size_t (&codes)[3],
                        const size_t (&thresholds)[3], Parameters* out) {
    Parameters p;
    p.block_len = block_len;
    p.blocks_per_frame = blocks_per_frame;
    for (int k = 0; k < 3; ++k) { p.codes[k] = codes[k]; p.thresholds[k] = thresholds[k]; }
    x3_params c = p.c_params();
    int rc = x3_params_validate(&c);
    if (rc) return static_cast<X3Error>(rc);
    p.rice_codes = RiceCodes::get(p.codes);
    *out = p;
    return X3Error::Ok;
  }
  x3_params c_params() const {
    x3_params c;
    c.block_len = (uint32_t)block_len;
    c.blocks_per_frame = (uint32_t)blocks_per_frame;
    for (int k = 0; k < 3; ++k) { c.codes[k] = (uint32_t)codes[k]; c.thresholds[k] = (uint32_t)thresholds[k]; }
    return c;
  }
  static Parameters from_c(const x3_params& c) {
    Parameters p;
    p.block_len = c.block_len;
    p.blocks_per_frame = c.blocks_per_frame;
    for (int k = 0; k < 3; ++k) { p.codes[k] = c.codes[k]; p.thresholds[k] = c.thresholds[k]; }
    p.rice_codes = RiceCodes::get(p.codes);
    return p;
  }
};

// x3.rs:29-45: slice-based channel
struct Channel {
  uint16_t id;
  const int16_t* wav;
  size_t len;
  uint32_t sample_rate;
  Parameters params;
  Channel(uint16_t id_, const int16_t* wav_, size_t len_, uint32_t rate, Parameters p)
      : id(id_), wav(wav_), len(len_), sample_rate(rate), params(p) {}
};

// x3.rs:47-69: iterator-based channel (what encoder::encode takes at this commit)
template <class It>
struct IterChannel {
  uint16_t id;
  It wav, wav_end;
  uint32_t sample_rate;
  Parameters params;
  IterChannel(uint16_t id_, It b, It e, uint32_t rate, Parameters p)
      : id(id_), wav(b), wav_end(e), sample_rate(rate), params(p) {}
};

// x3.rs:70-79
struct X3aSpec {
  uint32_t sample_rate;
  Parameters params;
  uint8_t channels;
};

struct Archive {  // x3.rs:136-141
  static constexpr const char* ID = "X3ARCHIV";
  static constexpr size_t ID_LEN = 8;
};
struct Frame {  // x3.rs:143-146
  static constexpr size_t MAX_LENGTH = 0x7fe0;
};
struct FrameHeader {  // x3.rs:148-184
  uint8_t source_id = 0;
  uint16_t samples = 0;
  uint8_t channels = 0;
  size_t payload_len = 0;
  uint16_t payload_crc = 0;
  static constexpr size_t LENGTH = 20;
  static constexpr uint16_t KEY = 30771;
  static constexpr size_t P_KEY = 0, P_SOURCE_ID = 2, P_CHANNELS = 3, P_SAMPLES = 4, P_PAYLOAD_SIZE = 6, P_TIME = 8,
                          P_HEADER_CRC = 16, P_PAYLOAD_CRC = 18;
};

namespace bytewriter {
enum class SeekFrom { Start, Current, End };

// bytewriter.rs:14-22
struct ByteWriter {
  virtual ~ByteWriter() = default;
  virtual X3Error align(size_t n, size_t* written = nullptr) = 0;
  virtual X3Error write_all(const uint8_t* v, size_t n) = 0;
  virtual X3Error flush() = 0;
  virtual X3Error seek(SeekFrom from, int64_t pos, uint64_t* out = nullptr) = 0;
  virtual X3Error stream_position(uint64_t* pos) = 0;
};

// bytewriter.rs:27-100
class SliceByteWriter : public ByteWriter {
 public:
  SliceByteWriter(uint8_t* slice, size_t len) : slice_(slice), len_(len) {}
  X3Error align(size_t n, size_t* written = nullptr) override {
    size_t residual = p_byte_ % n;
    if (written) *written = residual ? n - residual : 0;
    if (!residual) return X3Error::Ok;
    std::vector<uint8_t> z(n - residual, 0);
    return write_all(z.data(), z.size());
  }
  X3Error write_all(const uint8_t* v, size_t n) override {
    if (n > len_ - p_byte_) return X3Error::ByteWriterInsufficientMemory;
    std::memcpy(slice_ + p_byte_, v, n);
    p_byte_ += n;
    if (p_byte_ > stream_length_) stream_length_ = p_byte_;
    return X3Error::Ok;
  }
  X3Error flush() override { return X3Error::Ok; }
  X3Error seek(SeekFrom from, int64_t pos, uint64_t* out = nullptr) override {
    size_t abs_pos = from == SeekFrom::Start ? (size_t)pos
                     : from == SeekFrom::Current ? (size_t)((int64_t)p_byte_ + pos)
                                                 : (size_t)((int64_t)stream_length_ + pos);
    if (abs_pos > len_) return X3Error::ByteWriterInsufficientMemory;
    p_byte_ = abs_pos;
    if (p_byte_ > stream_length_) stream_length_ = p_byte_;
    if (out) *out = p_byte_;
    return X3Error::Ok;
  }
  X3Error stream_position(uint64_t* pos) override { *pos = p_byte_; return X3Error::Ok; }
  // direct access for the zero-copy encode path
  uint8_t* data() { return slice_; }
  size_t capacity() const { return len_; }
  size_t position() const { return p_byte_; }
  void advance_to(size_t pos) { p_byte_ = pos; if (p_byte_ > stream_length_) stream_length_ = p_byte_; }

 private:
  uint8_t* slice_;
  size_t len_;
  size_t p_byte_ = 0, stream_length_ = 0;
};

// bytewriter.rs:106-165 over any seekable std::ostream
class StreamByteWriter : public ByteWriter {
 public:
  explicit StreamByteWriter(std::ostream& os) : os_(os) {}
  X3Error align(size_t n, size_t* written = nullptr) override {
    uint64_t pos = 0;
    stream_position(&pos);
    size_t residual = pos % n;
    if (written) *written = residual ? n - residual : 0;
    if (!residual) return X3Error::Ok;
    std::vector<uint8_t> z(n - residual, 0);
    return write_all(z.data(), z.size());
  }
  X3Error write_all(const uint8_t* v, size_t n) override {
    os_.write(reinterpret_cast<const char*>(v), (std::streamsize)n);
    return os_ ? X3Error::Ok : X3Error::Io;
  }
  X3Error flush() override { os_.flush(); return os_ ? X3Error::Ok : X3Error::Io; }
  X3Error seek(SeekFrom from, int64_t pos, uint64_t* out = nullptr) override {
    os_.seekp(pos, from == SeekFrom::Start ? std::ios::beg : from == SeekFrom::Current ? std::ios::cur : std::ios::end);
    if (!os_) return X3Error::Io;
    if (out) *out = (uint64_t)os_.tellp();
    return X3Error::Ok;
  }
  X3Error stream_position(uint64_t* pos) override { *pos = (uint64_t)os_.tellp(); return X3Error::Ok; }

 private:
  std::ostream& os_;
};
}  // namespace bytewriter

namespace crc {
// crc.rs:44-47
inline uint16_t update_crc16(uint16_t c, uint8_t data) { return x3_crc16_update(c, data); }
// crc.rs:49-58 (GPU segmented reduction)
inline X3Error crc16(Context& ctx, const uint8_t* data, size_t n, uint16_t* out) {
  return static_cast<X3Error>(x3_crc16(ctx.raw(), data, n, out));
}
inline uint16_t crc16(const uint8_t* data, size_t n) {  // the reference's own signature: no error path
  std::lock_guard<std::mutex> lk(default_context_mutex());
  uint16_t c = 0;
  if (x3_crc16(default_context().raw(), data, n, &c)) throw std::runtime_error(x3_last_error(default_context().raw()));
  return c;
}
}  // namespace crc

// bitpacker.rs:46-190 (the subset the encoder uses: write_bits, write_packed_zeros, word_align, len, crc; dropping
// it flushes).  The fields are packed by a kernel when the packer is finished or dropped.
namespace bitpacker {
class BitPacker {
 public:
  // BitPacker::new(&mut SliceByteWriter): packs into the slice from the writer's position on
  explicit BitPacker(bytewriter::SliceByteWriter& writer) : BitPacker(default_context(), writer) {
    mu_ = &default_context_mutex();
  }
  BitPacker(Context& ctx, bytewriter::SliceByteWriter& writer) : writer_(writer) {
    MaybeLock lk(&ctx == &default_context() ? &default_context_mutex() : nullptr);
    int rc = x3_bitpacker_new(ctx.raw(), writer.data(), writer.capacity(), writer.position(), &bp_);
    if (rc) throw std::runtime_error("x3::bitpacker::BitPacker: x3_bitpacker_new failed");
  }
  BitPacker(const BitPacker&) = delete;
  BitPacker& operator=(const BitPacker&) = delete;
  ~BitPacker() {  // Drop (bitpacker.rs:56-62) flushes
    finish();
    x3_bitpacker_free(bp_);
  }
  X3Error write_bits(size_t value, size_t num_bits) {
    return static_cast<X3Error>(x3_bitpacker_write_bits(bp_, value, (uint32_t)num_bits));
  }
  X3Error write_packed_zeros(size_t num_zeros) {
    return static_cast<X3Error>(x3_bitpacker_write_packed_zeros(bp_, (uint32_t)num_zeros));
  }
  X3Error word_align() { return static_cast<X3Error>(x3_bitpacker_word_align(bp_)); }
  // write_bytes (bitpacker.rs:95-102) and inc_counter_n_bytes (:112-118)
  X3Error write_bytes(const uint8_t* array, size_t n) { return static_cast<X3Error>(x3_bitpacker_write_bytes(bp_, array, n)); }
  X3Error inc_counter_n_bytes(size_t n_bytes) { return static_cast<X3Error>(x3_bitpacker_inc_counter_n_bytes(bp_, n_bytes)); }
  // flush (bitpacker.rs:79-86): a partial byte is zero-padded, the packing kernel runs, the writer moves on
  X3Error finish() {
    MaybeLock lk(mu_);
    uint64_t pos = 0, len = 0;
    uint16_t crc = 0;
    int rc = x3_bitpacker_finish(bp_, &len, &crc, &pos);
    if (rc == 0) writer_.advance_to((size_t)pos);
    return static_cast<X3Error>(rc);
  }
  // len() / crc() (bitpacker.rs:75-90): complete bytes so far and their CRC-16
  size_t len() const {
    uint64_t n = 0;
    x3_bitpacker_peek(bp_, &n, nullptr);
    return (size_t)n;
  }
  uint16_t crc() const {
    MaybeLock lk(mu_);
    uint16_t c = 0;
    if (x3_bitpacker_peek(bp_, nullptr, &c)) throw std::runtime_error("x3::bitpacker::BitPacker::crc: HIP error");
    return c;
  }

 private:
  bytewriter::SliceByteWriter& writer_;
  x3_bitpacker* bp_ = nullptr;
  std::mutex* mu_ = nullptr;
};
}  // namespace bitpacker

// bitreader.rs:51-176
namespace bitreader {
class BitReader {
 public:
  BitReader(const uint8_t* array, size_t len) : BitReader(default_context(), array, len) { mu_ = &default_context_mutex(); }
  BitReader(Context& ctx, const uint8_t* array, size_t len) {
    MaybeLock lk(&ctx == &default_context() ? &default_context_mutex() : nullptr);
    if (x3_bitreader_new(ctx.raw(), array, len, &br_)) throw std::runtime_error("x3::bitreader::BitReader: x3_bitreader_new failed");
  }
  BitReader(const BitReader&) = delete;
  BitReader& operator=(const BitReader&) = delete;
  ~BitReader() { x3_bitreader_free(br_); }
  void inc_bits(size_t n) {
    MaybeLock lk(mu_);
    check(x3_bitreader_inc_bits(br_, (uint32_t)n));
  }
  uint32_t read_nbits(size_t n) {
    MaybeLock lk(mu_);
    uint32_t v = 0;
    check(x3_bitreader_read_nbits(br_, (uint32_t)n, &v));
    return v;
  }
  size_t count_zero_bits() {
    MaybeLock lk(mu_);
    uint32_t v = 0;
    check(x3_bitreader_count_zero_bits(br_, &v));
    return v;
  }
  x3_bitreader* raw() { return br_; }
  std::mutex* mutex() { return mu_; }

 private:
  static void check(int rc) {
    if (rc) throw std::runtime_error(std::string("x3::bitreader::BitReader: ") + x3_strerror(rc));
  }
  x3_bitreader* br_ = nullptr;
  std::mutex* mu_ = nullptr;
};
}  // namespace bitreader

namespace encoder {
using bytewriter::ByteWriter;
using bytewriter::SliceByteWriter;

// encoder.rs:122-162
inline void write_frame_header(size_t num_samples, uint8_t id, size_t payload_len, uint16_t payload_crc,
                               uint8_t (&out)[FrameHeader::LENGTH]) {
  x3_write_frame_header(num_samples, id, payload_len, payload_crc, out);
}

namespace detail {
// the reference prints this block after encode() when built with std (encoder.rs:96-108)
inline void print_stats(const uint64_t (&stats)[6]) {
  float t = (float)(stats[0] + stats[1] + stats[2] + stats[3] + stats[4] + stats[5]);
  std::printf("\nStatistics:\n  Rice-0: %.4f%%\n  Rice-1: %.4f%%\n  Rice-2: %.4f%%\n  Rice-3: %.4f%%\n  BFP: %.4f%%\n"
              "  Pass-through %.4f%%\n\n",
              stats[0] / t * 100.0f, stats[1] / t * 100.0f, stats[2] / t * 100.0f, stats[3] / t * 100.0f,
              stats[4] / t * 100.0f, stats[5] / t * 100.0f);
}

inline X3Error encode_samples(Context& ctx, const int16_t* wav, size_t n, size_t n_channels, const Parameters& params,
                              ByteWriter& writer, bool one_frame, uint64_t (&stats)[6]) {
  x3_params c = params.c_params();
  uint64_t pos = 0;
  if (auto* sw = dynamic_cast<SliceByteWriter*>(&writer)) {  // write straight into the caller's slice
    int rc = one_frame ? x3_encode_frame(ctx.raw(), wav, n, &c, sw->data(), sw->capacity(), sw->position(), &pos, stats)
                       : x3_encode(ctx.raw(), wav, n, (uint32_t)n_channels, &c, sw->data(), sw->capacity(),
                                   sw->position(), &pos, stats);
    // (on ByteWriterInsufficientMemory too: the slice holds every frame that fits and *out_pos stands behind the last of
    // them -- include/x3hip.h, x3_encode -- and so must the writer, as the reference's SliceByteWriter does after the frames
    // it has taken: src/bytewriter.rs:86-99)
    if (rc == 0 || rc == static_cast<int>(X3Error::ByteWriterInsufficientMemory)) sw->advance_to((size_t)pos);
    return static_cast<X3Error>(rc);
  }
  uint64_t start = 0;
  writer.stream_position(&start);
  const uint64_t parity = start & 1;  // only the parity of the absolute position matters (encoder.rs:182)
  std::vector<uint8_t> buf((size_t)(parity + x3_encode_bound(n, &c) + 64 + 3 * (one_frame ? n : 0)));
  int rc = one_frame ? x3_encode_frame(ctx.raw(), wav, n, &c, buf.data(), buf.size(), parity, &pos, stats)
                     : x3_encode(ctx.raw(), wav, n, (uint32_t)n_channels, &c, buf.data(), buf.size(), parity, &pos, stats);
  if (rc) return static_cast<X3Error>(rc);
  return writer.write_all(buf.data() + parity, (size_t)(pos - parity));
}
}  // namespace detail

// encoder::encode over slice channels (README shape): channels.len() > 1 -> MoreThanOneChannel (encoder.rs:55-57)
inline X3Error encode(Context& ctx, const Channel* const* channels, size_t n_channels, ByteWriter& writer,
                      bool print_statistics = false) {
  if (n_channels > 1) return X3Error::MoreThanOneChannel;
  if (n_channels == 0) return X3Error::BadArg;
  uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
  const Channel& ch = *channels[0];
  X3Error e = detail::encode_samples(ctx, ch.wav, ch.len, n_channels, ch.params, writer, false, stats);
  if (e == X3Error::Ok && print_statistics) detail::print_stats(stats);
  return e;
}

// encoder::encode over iterator channels (encoder.rs:51-54): the iterator is collected, dispatched, written
template <class It>
inline X3Error encode(Context& ctx, IterChannel<It>* const* channels, size_t n_channels, ByteWriter& writer,
                      bool print_statistics = false) {
  if (n_channels > 1) return X3Error::MoreThanOneChannel;
  if (n_channels == 0) return X3Error::BadArg;
  std::vector<int16_t> wav(channels[0]->wav, channels[0]->wav_end);
  channels[0]->wav = channels[0]->wav_end;  // the reference consumes the iterator
  uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
  X3Error e = detail::encode_samples(ctx, wav.data(), wav.size(), n_channels, channels[0]->params, writer, false, stats);
  if (e == X3Error::Ok && print_statistics) detail::print_stats(stats);
  return e;
}

// encoder::encode_frame (encoder.rs:175-214)
inline X3Error encode_frame(Context& ctx, const int16_t* wav, size_t n, ByteWriter& writer, const Parameters& params,
                            uint64_t (&stats)[6]) {
  uint64_t st[6] = {0, 0, 0, 0, 0, 0};
  X3Error e = detail::encode_samples(ctx, wav, n, 1, params, writer, true, st);
  if (e == X3Error::Ok)
    for (int i = 0; i < 6; ++i) stats[i] += st[i];
  return e;
}

// the reference's own argument lists (encoder.rs:51, :175): no context, statistics printed as the `std` build does
inline X3Error encode(const Channel* const* channels, size_t n_channels, ByteWriter& writer) {
  X3_WITH_DEFAULT_CONTEXT(encode(ctx, channels, n_channels, writer, true));
}
template <class It>
inline X3Error encode(IterChannel<It>* const* channels, size_t n_channels, ByteWriter& writer) {
  X3_WITH_DEFAULT_CONTEXT(encode(ctx, channels, n_channels, writer, true));
}
inline X3Error encode_frame(const int16_t* wav, size_t n, ByteWriter& writer, const Parameters& params, uint64_t (&stats)[6]) {
  X3_WITH_DEFAULT_CONTEXT(encode_frame(ctx, wav, n, writer, params, stats));
}
}  // namespace encoder

namespace decoder {
// decoder.rs:69-118
inline X3Error read_frame_header(const uint8_t* bytes, size_t len, FrameHeader* out) {
  x3_frame_header h;
  int rc = x3_read_frame_header(bytes, len, &h);
  if (rc) return static_cast<X3Error>(rc);
  out->source_id = h.source_id;
  out->samples = h.samples;
  out->channels = h.channels;
  out->payload_len = h.payload_len;
  out->payload_crc = h.payload_crc;
  return X3Error::Ok;
}

// decoder.rs:36-58: returns the number of samples written in *n_out (the reference's Some(p_wav))
inline X3Error decode_frame(Context& ctx, const uint8_t* x3_bytes, size_t len, int16_t* wav_buf, size_t wav_cap,
                            const Parameters& params, size_t samples, size_t* n_out) {
  x3_params c = params.c_params();
  uint64_t n = 0;
  int rc = x3_decode_frame(ctx.raw(), x3_bytes, len, wav_buf, wav_cap, &c, samples, &n);
  if (n_out) *n_out = (size_t)n;
  return static_cast<X3Error>(rc);
}

inline X3Error decode_frame(const uint8_t* x3_bytes, size_t len, int16_t* wav_buf, size_t wav_cap, const Parameters& params,
                            size_t samples, size_t* n_out) {
  X3_WITH_DEFAULT_CONTEXT(decode_frame(ctx, x3_bytes, len, wav_buf, wav_cap, params, samples, n_out));
}

// Not in the reference: announce the frame stream a loop over decode_frame is about to walk (x3_decode_prefetch): the
// calls are then served from windows decoded ahead instead of one dispatch each.  nullptr drops the announcement; the
// buffer must stay unchanged while it stands.
inline X3Error prefetch(Context& ctx, const uint8_t* x3, size_t len, const Parameters& params) {
  x3_params c = params.c_params();
  return static_cast<X3Error>(x3_decode_prefetch(ctx.raw(), x3, len, &c));
}
inline X3Error prefetch(const uint8_t* x3, size_t len, const Parameters& params) {
  X3_WITH_DEFAULT_CONTEXT(prefetch(ctx, x3, len, params));
}

// decoder.rs:132-145: one block of wav_len samples from the reader's position; *last_wav is read and updated
inline X3Error decode_block(bitreader::BitReader& br, int16_t* wav, size_t wav_len, int16_t* last_wav, const Parameters& params) {
  MaybeLock lk(br.mutex());
  x3_params c = params.c_params();
  return static_cast<X3Error>(x3_decode_block(br.raw(), wav, (uint32_t)wav_len, last_wav, &c));
}

// the X3aReader::decode_next_frame loop (decodefile.rs:105-136, 200-209) over an in-memory frame stream
struct StreamResult {
  uint64_t samples = 0, frames_ok = 0, frame_errors = 0;
};
inline X3Error decode_stream(Context& ctx, const uint8_t* x3, size_t len, const Parameters& params, int16_t* wav,
                             size_t wav_cap, StreamResult* res) {
  x3_params c = params.c_params();
  StreamResult r;
  int rc = x3_decode_stream(ctx.raw(), x3, len, &c, wav, wav_cap, &r.samples, &r.frames_ok, &r.frame_errors);
  if (res) *res = r;
  return static_cast<X3Error>(rc);
}
inline X3Error decode_stream(const uint8_t* x3, size_t len, const Parameters& params, int16_t* wav, size_t wav_cap,
                             StreamResult* res) {
  X3_WITH_DEFAULT_CONTEXT(decode_stream(ctx, x3, len, params, wav, wav_cap, res));
}
// the same walk for a stream that is already in HBM: frame index and decode on the device (x3_index_dev)
inline X3Error decode_stream_dev(Context& ctx, const uint8_t* d_x3, size_t len, const Parameters& params, int16_t* d_wav,
                                 size_t wav_cap, StreamResult* res) {
  x3_params c = params.c_params();
  StreamResult r;
  int rc = x3_decode_stream_dev(ctx.raw(), d_x3, len, &c, d_wav, wav_cap, &r.samples, &r.frames_ok, &r.frame_errors);
  if (res) *res = r;
  return static_cast<X3Error>(rc);
}
}  // namespace decoder

// Device-resident encode / decode (NOT in the crate, which knows no device): samples and stream stay in HBM between the two
// (x3_encode_dev / x3_decode_dev), optionally with the SEGMENT INDEX that lets a short stream decode on as many lanes as fill
// the GPU (x3_encode_dev_seg / x3_decode_dev_seg; the index is a hint the decoder proves entry by entry).
namespace device {
// device memory of a context (x3_dev_alloc): movable, freed with the object
class Buffer {
 public:
  Buffer() = default;
  Buffer(Context& ctx, size_t bytes) : ctx_(&ctx), bytes_(bytes) {
    if (x3_dev_alloc(ctx.raw(), bytes, &p_) != X3_OK) { p_ = nullptr; bytes_ = 0; }
  }
  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
  Buffer(Buffer&& o) noexcept : ctx_(o.ctx_), p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  Buffer& operator=(Buffer&& o) noexcept {
    if (this != &o) { reset(); ctx_ = o.ctx_; p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.bytes_ = 0; }
    return *this;
  }
  ~Buffer() { reset(); }
  void reset() { if (p_) x3_dev_free(ctx_->raw(), p_); p_ = nullptr; bytes_ = 0; }
  bool ok() const { return p_ != nullptr; }
  size_t size() const { return bytes_; }
  void* data() const { return p_; }
  template <typename T> T* as() const { return static_cast<T*>(p_); }
  X3Error upload(const void* src, size_t bytes) {
    return bytes <= bytes_ ? static_cast<X3Error>(x3_dev_upload(ctx_->raw(), p_, src, bytes)) : X3Error::BadArg;
  }
  X3Error download(void* dst, size_t bytes) const {
    return bytes <= bytes_ ? static_cast<X3Error>(x3_dev_download(ctx_->raw(), dst, p_, bytes)) : X3Error::BadArg;
  }

 private:
  Context* ctx_ = nullptr;
  void* p_ = nullptr;
  size_t bytes_ = 0;
};

// an encoded batch in device memory: the stream, where its frames begin, and (seg_blocks != 0) the segment index
struct EncodedStream {
  Buffer bytes, frame_offsets, seg_index;
  size_t len = 0;          // bytes of the stream
  size_t n_frames = 0;
  size_t n_per_clip = 0, n_clips = 1;
  uint32_t seg_blocks = 0;
  uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
};

// n_clips clips of n_per_clip samples each, back to back in d_wav, each encoded like encoder::encode encodes a channel.
// seg_blocks: 0 = no index; a power of two >= 4 (32 for the default 500-block frames) = also leave the segment index.
inline X3Error encode(Context& ctx, const int16_t* d_wav, size_t n_per_clip, size_t n_clips, const Parameters& params,
                      uint32_t seg_blocks, EncodedStream* out) {
  if (!out || !d_wav || n_per_clip == 0 || n_clips == 0) return X3Error::BadArg;
  const x3_params c = params.c_params();
  EncodedStream e;
  e.n_per_clip = n_per_clip;
  e.n_clips = n_clips;
  e.n_frames = static_cast<size_t>(x3_num_frames(n_per_clip, &c)) * n_clips;
  const size_t cap = static_cast<size_t>(x3_encode_bound(n_per_clip, &c)) * n_clips;
  e.bytes = Buffer(ctx, cap + 16);
  e.frame_offsets = Buffer(ctx, 8 * (e.n_frames + 1));
  const uint64_t n_idx = seg_blocks ? x3_seg_index_entries(e.n_frames, &c, seg_blocks) : 0;
  if (n_idx) {
    e.seg_index = Buffer(ctx, 8 * n_idx);
    e.seg_blocks = seg_blocks;
  }
  if (!e.bytes.ok() || !e.frame_offsets.ok() || (n_idx && !e.seg_index.ok())) return X3Error::Hip;
  const x3_batch b{n_per_clip, n_per_clip, n_clips};
  int rc = n_idx ? x3_encode_dev_seg(ctx.raw(), d_wav, &b, &c, e.bytes.as<uint8_t>(), cap, 0, e.frame_offsets.as<uint64_t>(),
                                     e.seg_index.as<uint64_t>(), seg_blocks)
                 : x3_encode_dev(ctx.raw(), d_wav, &b, &c, e.bytes.as<uint8_t>(), cap, 0, e.frame_offsets.as<uint64_t>());
  if (rc != X3_OK) return static_cast<X3Error>(rc);
  uint64_t pos = 0;
  rc = x3_encode_result(ctx.raw(), &pos, e.stats);
  if (rc != X3_OK) return static_cast<X3Error>(rc);
  e.len = static_cast<size_t>(pos);
  *out = std::move(e);
  return X3Error::Ok;
}

// ... and back: the samples of every clip at d_wav + clip * n_per_clip.  By the segment index when the stream has one.
// res->frames_ok = the first frame that failed (n_frames if none), frame_errors = its status.
inline X3Error decode(Context& ctx, const EncodedStream& s, const Parameters& params, int16_t* d_wav, size_t wav_cap,
                      decoder::StreamResult* res) {
  if (!d_wav || !s.bytes.ok() || !s.frame_offsets.ok()) return X3Error::BadArg;
  const x3_params c = params.c_params();
  const x3_batch b{s.n_per_clip, s.n_per_clip, s.n_clips};
  int rc = s.seg_blocks ? x3_decode_dev_seg(ctx.raw(), s.bytes.as<uint8_t>(), s.len, s.frame_offsets.as<uint64_t>(), s.n_frames, &b,
                                            nullptr, &c, d_wav, wav_cap, nullptr, s.seg_index.as<uint64_t>(), s.seg_blocks, 0)
                        : x3_decode_dev(ctx.raw(), s.bytes.as<uint8_t>(), s.len, s.frame_offsets.as<uint64_t>(), s.n_frames, &b,
                                        nullptr, &c, d_wav, wav_cap, nullptr);
  if (rc != X3_OK) return static_cast<X3Error>(rc);
  uint64_t first_bad = 0, before = 0;
  int st = 0;
  rc = x3_decode_result(ctx.raw(), &first_bad, &st, &before);
  if (res) { res->samples = before; res->frames_ok = first_bad; res->frame_errors = static_cast<uint64_t>(st); }
  if (rc != X3_OK) return static_cast<X3Error>(rc);
  return static_cast<X3Error>(st);
}

// Placement (x3_place_buffers; profiles/r6/decoder_modes.txt): the round trip timed on every pair of candidate buffers --
// ms[i * backs.size() + j] for (streams[i], backs[j]).  A pipeline that keeps its buffers calls this once and keeps the
// pair that runs best; what it does not keep it frees.
inline X3Error place_buffers(Context& ctx, const int16_t* d_wav, size_t n, const Parameters& params,
                             const std::vector<uint8_t*>& streams, size_t cap, uint64_t* d_frame_offsets,
                             const std::vector<int16_t*>& backs, std::vector<double>* ms, uint32_t warm = 4, uint32_t steps = 8) {
  if (!ms || streams.empty() || backs.empty()) return X3Error::BadArg;
  const x3_params c = params.c_params();
  ms->assign(streams.size() * backs.size(), 0.0);
  return static_cast<X3Error>(x3_place_buffers(ctx.raw(), d_wav, n, &c, streams.data(), static_cast<uint32_t>(streams.size()), cap,
                                               d_frame_offsets, backs.data(), static_cast<uint32_t>(backs.size()), warm, steps,
                                               ms->data()));
}
}  // namespace device

// Multi-channel extension (x3_mc.h; NOT in the crate, whose encode() returns MoreThanOneChannel for more than one channel
// and whose reader refuses such frames -- as encoder::encode and decoder::decode_stream above do): the layout the frame
// header's <Num Channels> and "pack the data block for each channel" (encoder.rs:197) foresee.  Channels of equal length.
namespace multichannel {
inline X3Error encode(Context& ctx, const Channel* const* channels, size_t n_channels, bytewriter::SliceByteWriter& writer,
                      uint64_t (&stats)[6]) {
  if (n_channels == 0) return X3Error::BadArg;
  std::vector<const int16_t*> wavs(n_channels);
  for (size_t k = 0; k < n_channels; ++k) {
    if (channels[k]->len != channels[0]->len) return X3Error::BadArg;
    wavs[k] = channels[k]->wav;
  }
  x3_params c = channels[0]->params.c_params();
  uint64_t pos = 0;
  int rc = x3_encode_mc(ctx.raw(), wavs.data(), (uint32_t)n_channels, channels[0]->len, &c, writer.data(), writer.capacity(),
                        writer.position(), &pos, stats);
  if (rc == 0) writer.advance_to((size_t)pos);
  return static_cast<X3Error>(rc);
}
inline X3Error decode_stream(Context& ctx, const uint8_t* x3, size_t len, size_t n_channels, const Parameters& params,
                             int16_t* const* wavs, size_t wav_cap, decoder::StreamResult* res) {
  x3_params c = params.c_params();
  decoder::StreamResult r;
  int rc = x3_decode_stream_mc(ctx.raw(), x3, len, (uint32_t)n_channels, &c, wavs, wav_cap, &r.samples, &r.frames_ok,
                               &r.frame_errors);
  if (res) *res = r;
  return static_cast<X3Error>(rc);
}
}  // namespace multichannel

// encodefile.rs / decodefile.rs without the files: the .x3a archive header and the whole-buffer conversions
namespace archive {
struct X3aSpec {  // decodefile.rs:36-43
  uint32_t sample_rate = 0;
  Parameters params;
  uint8_t channels = 0;
};
// create_archive_header (encodefile.rs:82-138); returns the header length through *out_len
inline X3Error create_archive_header(uint32_t sample_rate, const Parameters& params, uint8_t* out, size_t cap,
                                     size_t* out_len) {
  x3_params c = params.c_params();
  uint64_t n = 0;
  int rc = x3_archive_header_write(sample_rate, &c, out, cap, &n);
  if (out_len) *out_len = (size_t)n;
  return static_cast<X3Error>(rc);
}
// read_archive_header (decodefile.rs:142-176): header_size = 20 + XML, audio frames start at 8 + header_size
inline X3Error read_archive_header(const uint8_t* bytes, size_t len, X3aSpec* spec, size_t* header_size) {
  x3_params c;
  uint64_t hs = 0;
  X3aSpec s;
  int rc = x3_archive_header_read(bytes, len, &s.sample_rate, &c, &s.channels, &hs);
  if (rc == X3_OK) {
    s.params = Parameters::from_c(c);
    if (spec) *spec = s;
    if (header_size) *header_size = (size_t)hs;
  }
  return static_cast<X3Error>(rc);
}
// wav_to_x3a (encodefile.rs:48-77) / x3a_to_wav (decodefile.rs:189-212) on buffers
inline X3Error wav_to_x3a(Context& ctx, const int16_t* wav, size_t n, uint32_t sample_rate, uint8_t* out, size_t cap,
                          size_t* out_len) {
  uint64_t len = 0;
  int rc = x3_x3a_encode(ctx.raw(), wav, n, sample_rate, out, cap, &len, nullptr);
  if (out_len) *out_len = (size_t)len;
  return static_cast<X3Error>(rc);
}
inline X3Error x3a_to_wav(Context& ctx, const uint8_t* x3a, size_t len, int16_t* wav, size_t wav_cap,
                          uint32_t* sample_rate, decoder::StreamResult* res) {
  decoder::StreamResult r;
  int rc = x3_x3a_decode(ctx.raw(), x3a, len, wav, wav_cap, &r.samples, sample_rate, &r.frames_ok, &r.frame_errors);
  if (res) *res = r;
  return static_cast<X3Error>(rc);
}
}  // namespace archive

// the file level, under the reference's module names (encodefile.rs:48-77, decodefile.rs:189-227): what
// src/bin/x3.rs:79-80 calls.  Streamed through the GPU in chunks (x3hip.h).
namespace encodefile {
inline X3Error wav_to_x3a(Context& ctx, const char* wav_filename, const char* x3a_filename, bool print_statistics = true) {
  uint64_t stats[6] = {0, 0, 0, 0, 0, 0};
  const int rc = x3_wav_to_x3a(ctx.raw(), wav_filename, x3a_filename, stats);
  if (rc == X3_OK && print_statistics) encoder::detail::print_stats(stats);
  return static_cast<X3Error>(rc);
}
inline X3Error wav_to_x3a(const char* wav_filename, const char* x3a_filename) {  // encodefile.rs:48
  X3_WITH_DEFAULT_CONTEXT(wav_to_x3a(ctx, wav_filename, x3a_filename, true));
}
}  // namespace encodefile
namespace decodefile {
constexpr size_t READ_BUFFER_SIZE = X3_READ_BUFFER_SIZE;       // decodefile.rs:44 (X3_READ_BUFFER_SIZE is the C macro)
constexpr size_t X3_WRITE_BUFFER_SIZE = READ_BUFFER_SIZE * 8;  // decodefile.rs:45

// decodefile.rs:47-136.  open() reads the archive header; decode_next_frame() hands out one frame per call with the
// reference's per-call results: Ok + *some + *n = samples (Ok(Some(n)), n >= 1), Ok + !*some (Ok(None): end of the
// data, a payload that runs past it, or a frame that failed to decode -- counted in frame_errors()), or the error the
// reference returns for that frame (header CRC, key, payload CRC ...), after which a further call goes on behind it.
// Behind it the library decodes windows of frames ahead on the GPU (x3_reader.h).
class X3aReader {
 public:
  static X3Error open(const char* filename, X3aReader* out) { return open(default_context(), filename, out, &default_context_mutex()); }
  static X3Error open(Context& ctx, const char* filename, X3aReader* out, std::mutex* mu = nullptr) {
    MaybeLock lk(mu);
    out->close_locked();
    out->mu_ = mu;
    int rc = x3_reader_open(ctx.raw(), filename, &out->r_);
    if (rc) return static_cast<X3Error>(rc);
    x3_params c;
    x3_reader_spec(out->r_, &out->spec_.sample_rate, &c, &out->spec_.channels);
    out->spec_.params = Parameters::from_c(c);
    return X3Error::Ok;
  }
  X3aReader() = default;
  X3aReader(const X3aReader&) = delete;
  X3aReader& operator=(const X3aReader&) = delete;
  ~X3aReader() {
    MaybeLock lk(mu_);
    close_locked();
  }
  const X3aSpec& spec() const { return spec_; }
  X3Error decode_next_frame(int16_t (&wav_buf)[X3_WRITE_BUFFER_SIZE], size_t* n, bool* some) {
    MaybeLock lk(mu_);
    uint64_t got = 0;
    int rc = x3_reader_next_frame(r_, wav_buf, X3_WRITE_BUFFER_SIZE, &got);
    *some = rc == X3_OK && got != 0;
    *n = (size_t)got;
    return static_cast<X3Error>(rc);
  }
  size_t frame_errors() const { return r_ ? (size_t)x3_reader_frame_errors(r_) : 0; }

 private:
  void close_locked() {
    if (r_) x3_reader_close(r_);
    r_ = nullptr;
  }
  x3_reader* r_ = nullptr;
  std::mutex* mu_ = nullptr;
  X3aSpec spec_{0, Parameters{}, 0};
};

inline X3Error x3a_to_wav(Context& ctx, const char* x3a_filename, const char* wav_filename, uint64_t* samples = nullptr,
                          uint64_t* frame_errors = nullptr) {
  return static_cast<X3Error>(x3_x3a_to_wav(ctx.raw(), x3a_filename, wav_filename, samples, frame_errors));
}
inline X3Error x3a_to_wav(const char* x3a_filename, const char* wav_filename) {  // decodefile.rs:189
  X3_WITH_DEFAULT_CONTEXT(x3a_to_wav(ctx, x3a_filename, wav_filename, nullptr, nullptr));
}
}  // namespace decodefile
}  // namespace x3
