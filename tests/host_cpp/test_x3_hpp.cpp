// Exercises x3-rust_amd/host/x3.hpp (the C++ mirror of the reference's Rust API) the way the
// reference's own unit tests use the crate (src/encoder.rs:462-491, src/crc.rs:78-105), and against
// the CPU oracle on synthetic signals.   usage: test_x3_hpp [--host-only]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <unistd.h>
#include <vector>

#include "../../oracle/x3_oracle.h"
#include "../../x3-rust_amd/host/x3.hpp"

#define CHECK(c)                                                              \
  do {                                                                        \
    if (!(c)) {                                                               \
      std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);       \
      std::exit(1);                                                           \
    }                                                                         \
  } while (0)

static void host_only() {
  // Parameters::new (x3.rs:98-122): only the first two thresholds are validated
  x3::Parameters p;
  const size_t codes[3] = {0, 1, 3};
  const size_t ok[3] = {3, 8, 20}, bad0[3] = {7, 8, 20}, bad2[3] = {3, 8, 99};
  CHECK(x3::Parameters::create(20, 500, codes, ok, &p) == x3::X3Error::Ok);
  CHECK(x3::Parameters::create(20, 500, codes, bad0, &p) == x3::X3Error::InvalidEncodingThresh);
  CHECK(x3::Parameters::create(20, 500, codes, bad2, &p) == x3::X3Error::Ok);
  // crc.rs:78-98: the 16 header bytes -> 0xADDB, round trip through write/read_frame_header
  uint8_t h[20];
  x3::encoder::write_frame_header(0x2710, 1, 0x19d0, 0x6f61, h);
  CHECK(h[0] == 0x78 && h[1] == 0x33 && h[2] == 1 && h[3] == 1 && h[16] == 0xad && h[17] == 0xdb);
  x3::FrameHeader fh;
  CHECK(x3::decoder::read_frame_header(h, 20, &fh) == x3::X3Error::Ok);
  CHECK(fh.samples == 0x2710 && fh.payload_len == 0x19d0 && fh.payload_crc == 0x6f61 && fh.channels == 1);
  h[5] ^= 1;
  CHECK(x3::decoder::read_frame_header(h, 20, &fh) == x3::X3Error::FrameHeaderInvalidHeaderCRC);
  CHECK(x3::decoder::read_frame_header(h, 19, &fh) == x3::X3Error::FrameDecodeUnexpectedEnd);
  // SliceByteWriter (bytewriter.rs:27-100)
  uint8_t buf[8] = {0};
  x3::bytewriter::SliceByteWriter w(buf, sizeof buf);
  const uint8_t abc[3] = {1, 2, 3};
  CHECK(w.write_all(abc, 3) == x3::X3Error::Ok);
  size_t padded = 0;
  CHECK(w.align(2, &padded) == x3::X3Error::Ok && padded == 1);
  uint64_t pos = 0;
  w.stream_position(&pos);
  CHECK(pos == 4);
  CHECK(w.seek(x3::bytewriter::SeekFrom::Current, 5) == x3::X3Error::ByteWriterInsufficientMemory);
  CHECK(w.write_all(buf, 5) == x3::X3Error::ByteWriterInsufficientMemory);
  // create_archive_header / read_archive_header (encodefile.rs:82-138, decodefile.rs:142-176)
  uint8_t arc[512];
  size_t arc_len = 0, hsize = 0;
  CHECK(x3::archive::create_archive_header(192000, x3::Parameters{}, arc, sizeof arc, &arc_len) == x3::X3Error::Ok);
  CHECK(arc_len == 320 && std::memcmp(arc, "X3ARCHIV", 8) == 0);
  x3::archive::X3aSpec spec;
  CHECK(x3::archive::read_archive_header(arc, arc_len, &spec, &hsize) == x3::X3Error::Ok);
  CHECK(spec.sample_rate == 192000 && spec.channels == 0 && hsize == 312 && spec.params.block_len == 20 &&
        spec.params.codes[2] == 3 && spec.params.thresholds[2] == 20);
  arc[0] = 'Y';
  CHECK(x3::archive::read_archive_header(arc, arc_len, &spec, &hsize) == x3::X3Error::ArchiveHeaderXMLInvalidKey);
  // RiceCodes::get / Parameters.rice_codes (x3.rs:187-260): spot values of the reference's literal tables
  CHECK(p.rice_codes[0]->offset == 6 && p.rice_codes[1]->offset == 11 && p.rice_codes[2]->offset == 28);
  CHECK(p.rice_codes[2]->nsubs == 3 && p.rice_codes[2]->inv_len == 60 && p.rice_codes[2]->len == 56);
  CHECK(p.rice_codes[2]->code[0] == 15 && p.rice_codes[2]->num_bits[0] == 10 && p.rice_codes[2]->code[28] == 8 &&
        p.rice_codes[2]->num_bits[28] == 4 && p.rice_codes[0]->num_bits[13] == 15 && p.rice_codes[1]->inv[3] == -2);
  const size_t c2[3] = {1, 2, 3};
  CHECK(x3::RiceCodes::get(c2)[1]->offset == 20 && x3::RiceCodes::get(c2)[1]->code[1] == 5);
  std::printf("host-only checks ok\n");
}

static std::vector<uint8_t> oracle_encode(const std::vector<int16_t>& wav) {
  x3o_params p;
  x3o_params_default(&p);
  std::vector<uint8_t> out(wav.size() * 3 + 1024);
  uint64_t pos = 0, stats[6];
  CHECK(x3o_encode(wav.data(), wav.size(), 1, &p, out.data(), out.size(), 0, &pos, stats) == 0);
  out.resize(pos);
  return out;
}

int main(int argc, char** argv) {
  host_only();
  if (argc > 1 && !std::strcmp(argv[1], "--host-only")) return 0;

  x3::Context ctx(0);
  x3::Parameters params;

  // test_encode_frame_zeros (encoder.rs:462-491)
  {
    std::vector<int16_t> wav(20, 0);
    const uint8_t expected[26] = {'x', '3', 1, 1, 0, 20, 0, 6, 0, 0, 0, 0, 0, 0, 0, 0, 194, 242, 205, 128, 0, 0, 127, 255, 248, 0};
    std::vector<uint8_t> out(0x0eff * 2);
    x3::bytewriter::SliceByteWriter w(out.data(), out.size());
    uint64_t stats[6] = {0};
    CHECK(x3::encoder::encode_frame(ctx, wav.data(), wav.size(), w, params, stats) == x3::X3Error::Ok);
    uint64_t pos = 0;
    w.stream_position(&pos);
    CHECK(pos == 26 && !std::memcmp(out.data(), expected, 26) && stats[0] == 19);
  }
  // README shape: Channel -> encode -> SliceByteWriter; then the iterator shape into a stream
  std::vector<int16_t> wav(123457);
  CHECK(x3_synth(2, 0x5833, 0, wav.size(), wav.data()) == 0);
  const std::vector<uint8_t> ref = oracle_encode(wav);
  {
    x3::Channel ch(0, wav.data(), wav.size(), 44100, params);
    const x3::Channel* chans[1] = {&ch};
    std::vector<uint8_t> out(wav.size() * 2);
    x3::bytewriter::SliceByteWriter w(out.data(), out.size());
    CHECK(x3::encoder::encode(ctx, chans, 1, w) == x3::X3Error::Ok);
    uint64_t pos = 0;
    w.stream_position(&pos);
    CHECK(pos == ref.size() && !std::memcmp(out.data(), ref.data(), ref.size()));
    const x3::Channel* two[2] = {&ch, &ch};
    CHECK(x3::encoder::encode(ctx, two, 2, w) == x3::X3Error::MoreThanOneChannel);
    std::vector<uint8_t> small(1000);
    x3::bytewriter::SliceByteWriter ws(small.data(), small.size());
    CHECK(x3::encoder::encode(ctx, chans, 1, ws) == x3::X3Error::ByteWriterInsufficientMemory);
    // ... and a slice that takes some of the frames: they are there, complete and in place, and the writer stands behind
    // the last of them (bytewriter.rs:86-99: the reference's writer has advanced over everything it took)
    size_t third = 0, frames = 0;
    for (size_t o = 0; o + 20 <= ref.size() && frames < 3; ++frames) { o += 20 + ((size_t)ref[o + 6] << 8 | ref[o + 7]); third = o; }
    std::vector<uint8_t> some(third + 37, 0xEE);
    x3::bytewriter::SliceByteWriter w3(some.data(), some.size());
    CHECK(x3::encoder::encode(ctx, chans, 1, w3) == x3::X3Error::ByteWriterInsufficientMemory);
    uint64_t p3 = 0;
    w3.stream_position(&p3);
    CHECK(p3 == third && !std::memcmp(some.data(), ref.data(), third));
    for (size_t i = third; i < some.size(); ++i) CHECK(some[i] == 0xEE);
  }
  {
    using It = std::vector<int16_t>::const_iterator;
    x3::IterChannel<It> ch(0, wav.begin(), wav.end(), 44100, params);
    x3::IterChannel<It>* chans[1] = {&ch};
    std::ostringstream os;
    const char lead = 'L';  // start the stream at an odd position: the encoder must pad to even
    os.write(&lead, 1);
    x3::bytewriter::StreamByteWriter w(os);
    CHECK(x3::encoder::encode(ctx, chans, 1, w) == x3::X3Error::Ok);
    const std::string s = os.str();
    CHECK(s.size() == ref.size() + 2 && s[1] == 0 && !std::memcmp(s.data() + 2, ref.data(), ref.size()));
  }
  // the multi-channel extension (the crate itself stops at MoreThanOneChannel, checked above): two channels against the
  // oracle's definition, and back
  {
    std::vector<int16_t> left(wav.begin(), wav.begin() + 50000), right(wav.begin() + 60000, wav.begin() + 110000);
    x3::Channel l(0, left.data(), left.size(), 44100, params), r(1, right.data(), right.size(), 44100, params);
    const x3::Channel* chans[2] = {&l, &r};
    std::vector<uint8_t> out(400000), want(400000);
    x3::bytewriter::SliceByteWriter w(out.data(), out.size());
    uint64_t stats[6] = {0}, stats_o[6] = {0}, pos = 0, pos_o = 0;
    CHECK(x3::multichannel::encode(ctx, chans, 2, w, stats) == x3::X3Error::Ok);
    w.stream_position(&pos);
    x3o_params po;
    x3o_params_default(&po);
    const int16_t* planes[2] = {left.data(), right.data()};
    CHECK(x3o_encode_mc(planes, 2, left.size(), &po, want.data(), want.size(), 0, &pos_o, stats_o) == 0);
    CHECK(pos == pos_o && !std::memcmp(out.data(), want.data(), pos) && !std::memcmp(stats, stats_o, sizeof stats) && out[3] == 2);
    std::vector<int16_t> bl(left.size()), br(right.size());
    int16_t* back[2] = {bl.data(), br.data()};
    x3::decoder::StreamResult res;
    CHECK(x3::multichannel::decode_stream(ctx, out.data(), pos, 2, params, back, bl.size(), &res) == x3::X3Error::Ok);
    CHECK(res.samples == left.size() && res.frames_ok == 5 && res.frame_errors == 0 && bl == left && br == right);
    CHECK(x3::decoder::decode_stream(ctx, out.data(), pos, params, bl.data(), bl.size(), &res) == x3::X3Error::MoreThanOneChannel);
  }
  // decode: walk + decode the stream, then one frame through decode_frame
  {
    std::vector<int16_t> back(wav.size());
    x3::decoder::StreamResult r;
    CHECK(x3::decoder::decode_stream(ctx, ref.data(), ref.size(), params, back.data(), back.size(), &r) == x3::X3Error::Ok);
    CHECK(r.samples == wav.size() && r.frames_ok == 13 && r.frame_errors == 0 && back == wav);
    x3::FrameHeader fh;
    CHECK(x3::decoder::read_frame_header(ref.data(), ref.size(), &fh) == x3::X3Error::Ok);
    std::vector<int16_t> one(fh.samples);
    size_t n = 0;
    CHECK(x3::decoder::decode_frame(ctx, ref.data() + 20, fh.payload_len, one.data(), one.size(), params, fh.samples, &n) ==
          x3::X3Error::Ok);
    CHECK(n == fh.samples && !std::memcmp(one.data(), wav.data(), n * 2));
    uint16_t c = 0;
    CHECK(x3::crc::crc16(ctx, ref.data() + 20, fh.payload_len, &c) == x3::X3Error::Ok && c == fh.payload_crc);
    // the reference's per-frame loop (decodefile.rs:105-136 without the file), announced first: same samples
    CHECK(x3::decoder::prefetch(ctx, ref.data(), ref.size(), params) == x3::X3Error::Ok);
    std::vector<int16_t> again;
    for (size_t off = 0; off + 20 <= ref.size();) {
      CHECK(x3::decoder::read_frame_header(ref.data() + off, ref.size() - off, &fh) == x3::X3Error::Ok);
      std::vector<int16_t> fr(fh.samples);
      CHECK(x3::decoder::decode_frame(ctx, ref.data() + off + 20, fh.payload_len, fr.data(), fr.size(), params, fh.samples, &n) ==
            x3::X3Error::Ok);
      again.insert(again.end(), fr.begin(), fr.begin() + n);
      off += 20 + fh.payload_len;
    }
    CHECK(again == wav);
    CHECK(x3::decoder::prefetch(ctx, nullptr, 0, params) == x3::X3Error::Ok);
  }
  // the reference's own argument lists, on the default context: encode -> decode_stream, crc16
  {
    x3::Channel ch(0, wav.data(), 30000, 44100, params);
    const x3::Channel* chans[1] = {&ch};
    std::vector<uint8_t> out(60000);
    x3::bytewriter::SliceByteWriter w(out.data(), out.size());
    std::fflush(stdout);
    CHECK(x3::encoder::encode(chans, 1, w) == x3::X3Error::Ok);
    CHECK(w.position() > 60 && !std::memcmp(out.data(), ref.data(), 20));
    std::vector<int16_t> back(30000);
    x3::decoder::StreamResult r;
    CHECK(x3::decoder::decode_stream(out.data(), w.position(), params, back.data(), back.size(), &r) == x3::X3Error::Ok);
    CHECK(r.samples == 30000 && r.frames_ok == 3 && !std::memcmp(back.data(), wav.data(), 60000));
    x3::FrameHeader fh;
    CHECK(x3::decoder::read_frame_header(out.data(), 20, &fh) == x3::X3Error::Ok);
    CHECK(x3::crc::crc16(out.data() + 20, fh.payload_len) == fh.payload_crc);
  }
  // decoder::decode_block over bitreader::BitReader (decoder.rs:257-277, test_decode_block_ftype_1), and the
  // BitReader's own first test (bitreader.rs:195-202)
  {
    const uint8_t x3_inp[] = {0x01, 0x10, 0x23, 0x18, 0x14, 0x90, 0x40, 0x82, 0x58, 0x41, 0x02, 0x0C, 0x4C};
    const int16_t expected[] = {-375, -372, -374, -374, -376, -376, -373, -374, -373, -372,
                                -375, -372, -375, -374, -375, -375, -373, -376, -373};
    x3::bitreader::BitReader br(x3_inp, sizeof x3_inp);
    br.inc_bits(6);
    int16_t out[19], last = -373;
    CHECK(x3::decoder::decode_block(br, out, 19, &last, params) == x3::X3Error::Ok);
    CHECK(!std::memcmp(out, expected, sizeof expected) && last == -373);
    const uint8_t a4[] = {0x00, 0x00, 0x00, 0x3C};
    x3::bitreader::BitReader b2(ctx, a4, 4);
    CHECK(b2.count_zero_bits() == 26 && b2.read_nbits(4) == 0xF);
  }
  // bitpacker::BitPacker (bitpacker.rs:196-214, test_write_packed_bits): 3 bits then a 16-bit word
  {
    uint8_t out[8] = {0};
    x3::bytewriter::SliceByteWriter w(out, sizeof out);
    {
      x3::bitpacker::BitPacker bp(w);
      CHECK(bp.write_bits(0x3, 2) == x3::X3Error::Ok);
      CHECK(bp.write_packed_zeros(5) == x3::X3Error::Ok);
      CHECK(bp.write_bits(0x1FF, 9) == x3::X3Error::Ok);
      CHECK(bp.word_align() == x3::X3Error::Ok);
      const uint8_t packed[2] = {0xC1, 0xFF};
      CHECK(bp.len() == 2 && bp.crc() == x3::crc::crc16(packed, 2));
    }
    CHECK(out[0] == 0xC1 && out[1] == 0xFF && w.position() == 2);
  }
  // the file level (encodefile.rs:48-77, decodefile.rs:189-227) against the oracle's files, byte for byte
  {
    const char* dir = std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp";
    const std::string base = std::string(dir) + "/x3hpp_" + std::to_string((long)getpid());
    const std::string in = base + ".wav", a = base + "_g.x3a", b = base + "_o.x3a", wa = base + "_g.wav", wb = base + "_o.wav";
    uint8_t hdr[44];
    x3o_wav_header_write(44100, wav.size(), hdr);
    {
      std::ofstream f(in, std::ios::binary);
      f.write(reinterpret_cast<const char*>(hdr), 44);
      f.write(reinterpret_cast<const char*>(wav.data()), (std::streamsize)(wav.size() * 2));
    }
    auto slurp = [](const std::string& path) {
      std::ifstream f(path, std::ios::binary);
      return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    uint64_t so[6], n1 = 0, e1 = 0, n2 = 0, e2 = 0;
    CHECK(x3::encodefile::wav_to_x3a(ctx, in.c_str(), a.c_str(), false) == x3::X3Error::Ok);
    CHECK(x3o_wav_to_x3a(in.c_str(), b.c_str(), so) == 0);
    CHECK(slurp(a) == slurp(b) && slurp(a).size() > 320);
    CHECK(x3::decodefile::x3a_to_wav(ctx, a.c_str(), wa.c_str(), &n1, &e1) == x3::X3Error::Ok);
    CHECK(x3o_x3a_to_wav(b.c_str(), wb.c_str(), &n2, &e2) == 0);
    CHECK(n1 == wav.size() && n2 == n1 && e1 == 0 && e2 == 0 && slurp(wa) == slurp(wb) && slurp(wa) == slurp(in));
    CHECK(x3::encodefile::wav_to_x3a(ctx, (base + "_missing.wav").c_str(), a.c_str(), false) == x3::X3Error::Io);
    // decodefile::X3aReader (decodefile.rs:47-136): the reference's own loop (decodefile.rs:200-209)
    {
      x3::decodefile::X3aReader rd;
      CHECK(x3::decodefile::X3aReader::open(a.c_str(), &rd) == x3::X3Error::Ok);
      CHECK(rd.spec().sample_rate == 44100 && rd.spec().params.block_len == 20);
      static int16_t buf[x3::decodefile::X3_WRITE_BUFFER_SIZE];
      std::vector<int16_t> all;
      for (;;) {
        size_t n = 0;
        bool some = false;
        CHECK(rd.decode_next_frame(buf, &n, &some) == x3::X3Error::Ok);
        if (!some) break;
        all.insert(all.end(), buf, buf + n);
      }
      CHECK(all == wav && rd.frame_errors() == 0);
    }
    for (const std::string& f : {in, a, b, wa, wb}) std::remove(f.c_str());
  }
  // x3::device: samples and stream stay in HBM; a batch of three clips with and without the segment index, a corrupted
  // stream (the index does not hide a bad payload CRC), and a stream too short for an index
  {
    const size_t n_per = 41234, n_clips = 3;
    std::vector<int16_t> clips(n_per * n_clips);
    CHECK(x3_synth(2, 0x5834, 0, clips.size(), clips.data()) == 0);
    std::vector<uint8_t> want;
    for (size_t c = 0; c < n_clips; ++c) {
      const std::vector<uint8_t> one = oracle_encode(std::vector<int16_t>(clips.begin() + c * n_per, clips.begin() + (c + 1) * n_per));
      want.insert(want.end(), one.begin(), one.end());
    }
    x3::device::Buffer d_wav(ctx, clips.size() * 2), d_back(ctx, clips.size() * 2);
    CHECK(d_wav.ok() && d_back.ok());
    CHECK(d_wav.upload(clips.data(), clips.size() * 2) == x3::X3Error::Ok);
    for (uint32_t sb : {0u, 32u, 64u}) {
      x3::device::EncodedStream es;
      CHECK(x3::device::encode(ctx, d_wav.as<int16_t>(), n_per, n_clips, params, sb, &es) == x3::X3Error::Ok);
      CHECK(es.len == want.size() && es.n_frames == 15 && es.seg_blocks == sb);
      std::vector<uint8_t> got(es.len);
      CHECK(es.bytes.download(got.data(), got.size()) == x3::X3Error::Ok && got == want);
      std::vector<int16_t> back(clips.size(), 0x5a5a);
      CHECK(d_back.upload(back.data(), back.size() * 2) == x3::X3Error::Ok);
      x3::decoder::StreamResult res;
      CHECK(x3::device::decode(ctx, es, params, d_back.as<int16_t>(), clips.size(), &res) == x3::X3Error::Ok);
      CHECK(res.samples == clips.size() && res.frames_ok == es.n_frames);
      CHECK(d_back.download(back.data(), back.size() * 2) == x3::X3Error::Ok && back == clips);
      // one payload bit of the second clip's third frame flipped
      std::vector<uint64_t> offs(es.n_frames + 1);
      CHECK(es.frame_offsets.download(offs.data(), offs.size() * 8) == x3::X3Error::Ok);
      got[offs[7] + 20 + 100] ^= 0x10;
      CHECK(es.bytes.upload(got.data(), got.size()) == x3::X3Error::Ok);
      CHECK(x3::device::decode(ctx, es, params, d_back.as<int16_t>(), clips.size(), &res) == x3::X3Error::FrameHeaderInvalidPayloadCRC);
      CHECK(res.frames_ok == 7);
    }
    x3::device::EncodedStream tiny;   // 100 samples: five blocks, no stretch to index
    CHECK(x3::device::encode(ctx, d_wav.as<int16_t>(), 100, 1, params, 32, &tiny) == x3::X3Error::Ok);
    x3::decoder::StreamResult res;
    CHECK(x3::device::decode(ctx, tiny, params, d_back.as<int16_t>(), 100, &res) == x3::X3Error::Ok && res.samples == 100);
    std::vector<int16_t> back(100);
    CHECK(d_back.download(back.data(), 200) == x3::X3Error::Ok && std::equal(back.begin(), back.end(), clips.begin()));
    // x3::device::place_buffers: every pair of two stream buffers and two sample buffers timed; the last pair holds the
    // last round trip; a stream buffer too small for the stream is the writer's error
    {
      const x3_params cp = params.c_params();
      const size_t n = clips.size(), cap = x3_encode_bound(n, &cp), nf = x3_num_frames(n, &cp);
      x3::device::Buffer s0(ctx, cap + 16), s1(ctx, cap + 16), b0(ctx, 2 * n), b1(ctx, 2 * n), fo(ctx, 8 * (nf + 1));
      CHECK(s0.ok() && s1.ok() && b0.ok() && b1.ok() && fo.ok());
      std::vector<double> ms;
      CHECK(x3::device::place_buffers(ctx, d_wav.as<int16_t>(), n, params, {s0.as<uint8_t>(), s1.as<uint8_t>()}, cap, fo.as<uint64_t>(),
                                      {b0.as<int16_t>(), b1.as<int16_t>()}, &ms, 1, 2) == x3::X3Error::Ok);
      CHECK(ms.size() == 4 && ms[0] > 0 && ms[1] > 0 && ms[2] > 0 && ms[3] > 0);
      std::vector<int16_t> rt(n);
      CHECK(b1.download(rt.data(), 2 * n) == x3::X3Error::Ok && rt == clips);
      CHECK(x3::device::place_buffers(ctx, d_wav.as<int16_t>(), n, params, {s0.as<uint8_t>()}, 64, fo.as<uint64_t>(), {b0.as<int16_t>()}, &ms,
                                      1, 1) == x3::X3Error::ByteWriterInsufficientMemory);
    }
  }
  std::printf("x3.hpp checks ok\n");
  return 0;
}
