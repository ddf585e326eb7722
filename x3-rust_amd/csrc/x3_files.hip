// x3_files.hip -- the .x3a archive header, wav <-> x3a in memory and on files (the chunked pipeline), the incremental reader
// (C ABI: include/x3hip.h; units: x3_internal.h).  Host code only: the GPU work goes through x3_encode.hip / x3_decode.hip.
#include "x3_internal.h"

// ------------------------------------------------------------------------------------------------
// .x3a archive header (host arithmetic) and the in-memory wav <-> x3a conversions
// ------------------------------------------------------------------------------------------------
static std::string archive_xml(uint32_t sample_rate, const x3_params* p) {
  // the XML block of create_archive_header (encodefile.rs:93-117), field for field
  char buf[512];
  std::snprintf(buf, sizeof buf,
                "<X3ARCH PROG=\"x3new.m\" VERSION=\"2.0\" />"
                "<CFG ID=\"0\" FTYPE=\"XML\" />"
                "<CFG ID=\"1\" FTYPE=\"WAV\">"
                "<FS UNIT=\"Hz\">%u</FS>"
                "<SUFFIX>wav</SUFFIX>"
                "<CODEC TYPE=\"X3\" VERS=\"2\">"
                "<BLKLEN>%u</BLKLEN>"
                "<CODES N=\"4\">RICE%u,RICE%u,RICE%u,BFP</CODES>"
                "<FILTER>DIFF</FILTER>"
                "<NBITS>16</NBITS>"
                "<T N=\"3\">%u,%u,%u</T>"
                "</CODEC>"
                "</CFG>",
                sample_rate, p->block_len, p->codes[0], p->codes[1], p->codes[2], p->thresholds[0], p->thresholds[1],
                p->thresholds[2]);
  return std::string(buf);
}

extern "C" int x3_archive_header_write(uint32_t sample_rate, const x3_params* p, uint8_t* out, uint64_t out_cap,
                                       uint64_t* out_len) {
  if (!p || (!out && out_cap)) return X3_ERR_BAD_ARG;
  std::string xml = archive_xml(sample_rate, p);
  uint16_t crc = header_crc16_host(reinterpret_cast<const uint8_t*>(xml.data()), xml.size());
  if (xml.size() & 1) {  // align to the nearest word (encodefile.rs:123-128)
    xml.push_back('\0');
    crc = x3_crc16_update(crc, 0);
  }
  const uint64_t total = 8 + 20 + xml.size();
  if (out_len) *out_len = total;
  if (total > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  std::memcpy(out, "X3ARCHIV", 8);
  x3_write_frame_header(0, 0, xml.size(), crc, out + 8);  // id 0, 0 samples (encodefile.rs:134)
  std::memcpy(out + 28, xml.data(), xml.size());
  return X3_OK;
}

// text of the first <NAME ...>text</NAME> element (what quick-xml's Event::Start + read_text yield for
// well-formed input; quick-xml 0.38 is a dependency of the reference that is not in its tree)
static bool xml_first_text(const std::string& xml, const char* name, std::string* text) {
  const std::string open = std::string("<") + name;
  size_t i = 0;
  for (;;) {
    i = xml.find(open, i);
    if (i == std::string::npos) return false;
    const char nx = i + open.size() < xml.size() ? xml[i + open.size()] : '\0';
    if (nx == '>' || nx == ' ' || nx == '\t' || nx == '\n' || nx == '\r') break;
    i += open.size();
  }
  const size_t gt = xml.find('>', i);
  if (gt == std::string::npos || (gt > 0 && xml[gt - 1] == '/')) return false;
  const size_t close = xml.find(std::string("</") + name + ">", gt + 1);
  if (close == std::string::npos) return false;
  std::string t = xml.substr(gt + 1, close - gt - 1);
  const size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n");  // trim_text(true)
  *text = a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
  return true;
}

static bool parse_u32(const std::string& s, uint32_t* v) {  // Rust's str::parse::<u32>
  if (s.empty() || s.size() > 10) return false;
  uint64_t acc = 0;
  size_t i = s[0] == '+' ? 1 : 0;
  if (i == s.size()) return false;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    acc = acc * 10 + (uint64_t)(s[i] - '0');
  }
  if (acc > 0xFFFFFFFFull) return false;
  *v = (uint32_t)acc;
  return true;
}

extern "C" int x3_archive_header_read(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, x3_params* p,
                                      uint8_t* channels, uint64_t* header_size) {
  if ((!bytes && len) || !p) return X3_ERR_BAD_ARG;  // (no bytes at all -- an empty file -- is a read that fails: Io)
  if (len < 8) return X3_ERR_IO;  // read_exact fails
  if (std::memcmp(bytes, "X3ARCHIV", 8) != 0) return X3_ERR_ARCHIVE_HEADER_XML_INVALID_KEY;
  if (len < 28) return X3_ERR_IO;
  x3_frame_header h;
  int rc = x3_read_frame_header(bytes + 8, 20, &h);
  if (rc) return rc;
  if (len - 28 < h.payload_len) return X3_ERR_IO;
  const std::string xml(reinterpret_cast<const char*>(bytes + 28), h.payload_len);
  std::string fs, bl, codes, th;
  // a missing element is an index panic in the reference (fs[0] etc., decodefile.rs:267-272)
  if (!xml_first_text(xml, "FS", &fs) || !xml_first_text(xml, "BLKLEN", &bl) || !xml_first_text(xml, "CODES", &codes) ||
      !xml_first_text(xml, "T", &th))
    return X3_ERR_BAD_ARG;
  uint32_t rate = 0, block_len = 0;
  if (!parse_u32(fs, &rate) || !parse_u32(bl, &block_len)) return X3_ERR_BAD_ARG;  // .unwrap() panics
  std::vector<uint32_t> ids, ths;
  for (size_t i = 0; i <= codes.size();) {
    const size_t j = std::min(codes.find(',', i), codes.size());
    const std::string w = codes.substr(i, j - i);
    if (w == "RICE0") ids.push_back(0);
    else if (w == "RICE1") ids.push_back(1);
    else if (w == "RICE2") ids.push_back(2);
    else if (w == "RICE3") ids.push_back(3);
    else if (w != "BFP") return X3_ERR_ARCHIVE_HEADER_XML_RICE_CODE;
    i = j + 1;
  }
  for (size_t i = 0; i <= th.size();) {
    const size_t j = std::min(th.find(',', i), th.size());
    uint32_t v;
    if (!parse_u32(th.substr(i, j - i), &v)) return X3_ERR_BAD_ARG;
    ths.push_back(v);
    i = j + 1;
  }
  if (ids.size() < 3 || ths.size() < 3) return X3_ERR_BAD_ARG;  // rice_code_ids[i] / thresholds[i] panic
  x3_params q;
  q.block_len = block_len;
  q.blocks_per_frame = 500;  // Parameters::DEFAULT_BLOCKS_PER_FRAME (decodefile.rs:297)
  for (int k = 0; k < 3; ++k) { q.codes[k] = ids[k]; q.thresholds[k] = ths[k]; }
  rc = x3_params_validate(&q);
  if (rc) return rc;
  *p = q;
  if (sample_rate) *sample_rate = rate;
  if (channels) *channels = h.channels;
  if (header_size) *header_size = 20 + (uint64_t)h.payload_len;
  return X3_OK;
}

extern "C" int x3_x3a_encode(x3_ctx* c, const int16_t* wav, uint64_t n, uint32_t sample_rate, uint8_t* out,
                             uint64_t out_cap, uint64_t* out_len, uint64_t stats[6]) {
  if (!c || (!wav && n) || (!out && out_cap)) return X3_ERR_BAD_ARG;
  x3_params p;
  x3_params_default(&p);  // wav_to_x3a always uses the default parameters (encodefile.rs:57)
  uint64_t hlen = 0;
  int rc = x3_archive_header_write(sample_rate, &p, out, out_cap, &hlen);
  if (out_len) *out_len = hlen;
  if (rc) return rc;
  uint64_t pos = hlen;
  rc = x3_encode(c, wav, n, 1, &p, out, out_cap, hlen, &pos, stats);
  if (out_len) *out_len = pos;
  return rc;
}

extern "C" int x3_x3a_decode(x3_ctx* c, const uint8_t* x3a, uint64_t len, int16_t* wav, uint64_t wav_cap,
                             uint64_t* n_out, uint32_t* sample_rate, uint64_t* frames_ok, uint64_t* frame_errors) {
  if (!c || (!x3a && len) || (!wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  x3_params p;
  uint8_t ch = 0;
  uint64_t hsize = 0;
  int rc = x3_archive_header_read(x3a, len, sample_rate, &p, &ch, &hsize);
  if (rc) return rc;
  // X3aReader::open: remaining = file length - header_size, i.e. 8 bytes more than really follow
  const uint64_t start = 8 + hsize;
  return decode_stream_impl(c, x3a + start, len - start, 8, &p, wav, wav_cap, n_out, frames_ok, frame_errors);
}


#include "x3_file_pipeline.h"
#include "x3_reader.h"

int x3_wav_parse_fd_for_tests(int fd, uint64_t file_len, uint32_t* sample_rate, uint16_t* channels, uint16_t* bits,
                              uint64_t* data_off, uint64_t* data_len) {
  WavInfo wi;
  const int rc = wav_parse_fd(fd, file_len, &wi);
  if (sample_rate) *sample_rate = wi.sample_rate;
  if (channels) *channels = wi.channels;
  if (bits) *bits = wi.bits;
  if (data_off) *data_off = wi.data_off;
  if (data_len) *data_len = wi.data_len;
  return rc;
}
