// fuzz_host_parsers.cpp -- the library's HOST code that parses bytes somebody else wrote, under AddressSanitizer and
// UndefinedBehaviorSanitizer, against the oracle.  No GPU: the five translation units of libx3hip are
// compiled host-only (hipcc --cuda-host-only -fsanitize=address,undefined) and linked with this driver; nothing here
// creates a context or launches a kernel.  (GPU sanitizers are not available on the pool; the host side is where
// untrusted files are read: decodefile.rs:142-176,232-303, encodefile.rs:82-138, decoder.rs:69-118, decodefile.rs:93-136.)
//
//   archive headers: every prefix of valid headers, the tests' broken archives, random mutations  -> x3_archive_header_read
//   WAV headers    : PCM, extensible, extra chunks; every prefix, random mutations               -> the parser of x3_wav_to_x3a
//   frame headers  : random and mutated                                                          -> x3_read_frame_header
//   frame walk     : streams with damaged / truncated / lying headers                            -> walk_host
//   shard ranges   : x3_shard_frame_range / sample_range / offsets on extreme arguments
//
// Every status is compared with the oracle's for the same bytes (tests/: the oracle is the checker), every result that is
// returned with X3_OK as well.  Prints "ok <counts>" and exits 0, or the first difference and exits 1.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "x3_internal.h"
extern "C" {
#include "../../oracle/x3_oracle.h"
}

static std::mt19937_64 rng(0x58330007ull);
static uint64_t rnd(uint64_t n) { return n ? rng() % n : 0; }
static int fails = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      std::printf("DIFFERENCE %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                            \
      std::printf("\n");                                   \
      if (++fails > 10) std::exit(1);                      \
    }                                                      \
  } while (0)

typedef std::vector<uint8_t> Bytes;

static Bytes mutate(const Bytes& in) {
  Bytes b = in;
  const int ops = 1 + (int)rnd(4);
  for (int k = 0; k < ops; ++k) {
    switch (rnd(8)) {
      case 0: if (!b.empty()) b[rnd(b.size())] ^= (uint8_t)(1u << rnd(8)); break;
      case 1: if (!b.empty()) b[rnd(b.size())] = (uint8_t)rnd(256); break;
      case 2: b.resize(rnd(b.size() + 1)); break;                                   // truncate
      case 3: { const size_t at = rnd(b.size() + 1); b.insert(b.begin() + at, (uint8_t)rnd(256)); break; }
      case 4: if (!b.empty()) b.erase(b.begin() + rnd(b.size())); break;
      case 5: if (b.size() >= 4) { const size_t at = rnd(b.size() - 3); const uint32_t v = (uint32_t)(rnd(3) == 0 ? 0xFFFFFFFFu : rnd(1u << 20));
                                   std::memcpy(&b[at], &v, 4); } break;             // a length field somewhere
      case 6: { const char* w[] = {"<FS>", "</FS>", "RICE9", "BFP", ",", "<T>", "<CODES>", "<BLKLEN>", "-1", "4294967296", " "};
                const std::string t = w[rnd(11)]; const size_t at = rnd(b.size() + 1); b.insert(b.begin() + at, t.begin(), t.end()); } break;
      default: if (b.size() >= 2) { const size_t a = rnd(b.size()), c = rnd(b.size()); std::swap(b[a], b[c]); } break;
    }
  }
  return b;
}

// ---------------------------------------------------------------- archive headers
static void refresh_archive_crcs(Bytes& b) {   // (most mutations die at the header CRC otherwise)
  if (b.size() < 28) return;
  const uint32_t plen = ((uint32_t)b[14] << 8) | b[15];
  const uint16_t hc = x3o_crc16(&b[8], 16);
  b[24] = (uint8_t)(hc >> 8); b[25] = (uint8_t)hc;
  if (b.size() >= 28 + (size_t)plen) { const uint16_t pc = x3o_crc16(&b[28], plen); b[26] = (uint8_t)(pc >> 8); b[27] = (uint8_t)pc; }
}

static void one_archive(const Bytes& b) {
  uint32_t r1 = 0, r2 = 0; x3_params p1; x3o_params p2; uint8_t c1 = 0, c2 = 0; uint64_t h1 = 0, h2 = 0;
  std::memset(&p1, 0, sizeof p1); std::memset(&p2, 0, sizeof p2);
  // (an exact-size heap copy: a read one byte past the end is ASan's to see)
  uint8_t* q = (uint8_t*)std::malloc(b.size() ? b.size() : 1);
  if (!b.empty()) std::memcpy(q, b.data(), b.size());
  const int s1 = x3_archive_header_read(q, b.size(), &r1, &p1, &c1, &h1);
  const int s2 = x3o_archive_header_read(q, b.size(), &r2, &p2, &c2, &h2);
  std::free(q);
  CHECK(s1 == s2, "archive header: status %d, oracle %d (len %zu)", s1, s2, b.size());
  if (s1 == 0 && s2 == 0)
    CHECK(r1 == r2 && c1 == c2 && h1 == h2 && p1.block_len == p2.block_len && p1.blocks_per_frame == p2.blocks_per_frame &&
              !std::memcmp(p1.codes, p2.codes, 12) && !std::memcmp(p1.thresholds, p2.thresholds, 12),
          "archive header: fields differ (len %zu)", b.size());
}

static long fuzz_archives(long n_mut) {
  long n = 0;
  std::vector<Bytes> seeds;
  for (uint32_t rate : {8000u, 44100u, 192000u, 1000000u, 4000000000u}) {
    x3_params p; x3_params_default(&p);
    for (int v = 0; v < 3; ++v) {
      if (v == 1) { p.codes[0] = 1; p.codes[1] = 2; p.codes[2] = 3; p.thresholds[0] = 5; p.thresholds[1] = 9; p.thresholds[2] = 25; p.block_len = 40; }
      if (v == 2) { x3_params_default(&p); p.block_len = 7; }
      Bytes b(1024); uint64_t len = 0;
      if (x3_archive_header_write(rate, &p, b.data(), b.size(), &len) != 0) continue;
      b.resize(len);
      // the writer against the oracle's
      Bytes o(1024); uint64_t ol = 0; x3o_params op; std::memcpy(&op, &p, sizeof op);
      CHECK(x3o_archive_header_write(rate, &op, o.data(), o.size(), &ol) == 0 && ol == len && !std::memcmp(o.data(), b.data(), len), "archive header write");
      seeds.push_back(b);
      for (size_t k = 0; k <= b.size(); ++k) { one_archive(Bytes(b.begin(), b.begin() + k)); ++n; }   // every prefix
      Bytes tail = b; tail.insert(tail.end(), 100, 0x55); one_archive(tail); ++n;
    }
  }
  for (long i = 0; i < n_mut; ++i) {
    Bytes b = mutate(seeds[rnd(seeds.size())]);
    if (rnd(4)) refresh_archive_crcs(b);
    one_archive(b); ++n;
  }
  return n;
}

// ---------------------------------------------------------------- WAV headers
static int memfd = -1;
static void one_wav(const Bytes& b) {
  if (ftruncate(memfd, 0) != 0 || (b.size() && pwrite(memfd, b.data(), b.size(), 0) != (ssize_t)b.size())) { std::perror("memfd"); std::exit(2); }
  uint32_t r1 = 0, r2 = 0; uint16_t c1 = 0, c2 = 0, b1 = 0, b2 = 0; uint64_t o1 = 0, o2 = 0, l1 = 0, l2 = 0;
  const int s1 = x3_wav_parse_fd_for_tests(memfd, b.size(), &r1, &c1, &b1, &o1, &l1);
  uint8_t* q = (uint8_t*)std::malloc(b.size() ? b.size() : 1);
  if (!b.empty()) std::memcpy(q, b.data(), b.size());
  const int s2 = x3o_wav_parse(q, b.size(), &r2, &c2, &b2, &o2, &l2);
  std::free(q);
  CHECK(s1 == s2, "wav header: status %d, oracle %d (len %zu)", s1, s2, b.size());
  if (s1 == 0 && s2 == 0) CHECK(r1 == r2 && c1 == c2 && b1 == b2 && o1 == o2 && l1 == l2, "wav header: fields differ (len %zu)", b.size());
}
static void le32(Bytes& b, uint32_t v) { for (int k = 0; k < 4; ++k) b.push_back((uint8_t)(v >> (8 * k))); }
static void le16(Bytes& b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
static void tag(Bytes& b, const char* t) { b.insert(b.end(), t, t + 4); }

static long fuzz_wavs(long n_mut) {
  long n = 0;
  std::vector<Bytes> seeds;
  for (int v = 0; v < 4; ++v) {
    Bytes b; tag(b, "RIFF"); le32(b, 0); tag(b, "WAVE");
    if (v == 2) { tag(b, "LIST"); le32(b, 5); b.insert(b.end(), {1, 2, 3, 4, 5, 0}); }     // an odd chunk in front, padded
    tag(b, "fmt ");
    if (v == 1 || v == 3) {   // WAVE_FORMAT_EXTENSIBLE
      le32(b, 40); le16(b, 0xFFFE); le16(b, 1); le32(b, 96000); le32(b, 192000); le16(b, 2); le16(b, 16); le16(b, 22); le16(b, 16); le32(b, 4);
      le16(b, v == 3 ? 3 : 1); b.insert(b.end(), 14, 0);
    } else {
      le32(b, 16); le16(b, 1); le16(b, 1); le32(b, 44100); le32(b, 88200); le16(b, 2); le16(b, 16);
    }
    tag(b, "data"); le32(b, 64); b.insert(b.end(), 64, 7);
    seeds.push_back(b);
    for (size_t k = 0; k <= b.size(); ++k) { one_wav(Bytes(b.begin(), b.begin() + k)); ++n; }
  }
  uint8_t h44[44]; x3o_wav_header_write(48000, 1000, h44);
  seeds.push_back(Bytes(h44, h44 + 44));
  for (long i = 0; i < n_mut; ++i) { one_wav(mutate(seeds[rnd(seeds.size())])); ++n; }
  return n;
}

// ---------------------------------------------------------------- frame headers and the host's frame walk
static void one_frame_header(const uint8_t h[20]) {
  x3_frame_header a; x3o_frame_header b;
  std::memset(&a, 0, sizeof a); std::memset(&b, 0, sizeof b);
  uint8_t* q = (uint8_t*)std::malloc(20); std::memcpy(q, h, 20);
  const int s1 = x3_read_frame_header(q, 20, &a), s2 = x3o_read_frame_header(q, 20, &b);
  std::free(q);
  CHECK(s1 == s2, "frame header: status %d, oracle %d", s1, s2);
  if (s1 == 0 && s2 == 0)
    CHECK(a.source_id == b.source_id && a.samples == b.samples && a.channels == b.channels && a.payload_len == b.payload_len &&
              a.payload_crc == b.payload_crc, "frame header: fields differ");
}

// decodefile.rs:105-136 as the oracle walks it (decode_stream_phantom), headers and lengths only: frame offsets and the
// status the walk ends with if every frame it steps over decodes
static void ref_walk(const Bytes& s, uint64_t phantom, uint64_t wav_cap, std::vector<uint64_t>* offs, int* terminal) {
  uint64_t pos = 0, remaining = s.size() + phantom, nsamp = 0;
  *terminal = 0;
  for (;;) {
    if (remaining <= 20) break;
    if (s.size() - pos < 20) { *terminal = X3O_IO; break; }
    x3o_frame_header h;
    const int rc = x3o_read_frame_header(s.data() + pos, 20, &h);
    if (rc) { *terminal = rc; break; }
    if (remaining - 20 < h.payload_len) break;
    if (h.payload_len > 1024 * 24) { *terminal = X3O_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; }
    if (s.size() - pos - 20 < h.payload_len) { *terminal = X3O_IO; break; }
    offs->push_back(pos);
    if (h.samples == 0 || h.payload_len < 2 || nsamp + h.samples > wav_cap) { *terminal = X3O_BAD_ARG; break; }   // (the reference panics)
    nsamp += h.samples;
    pos += 20 + h.payload_len;
    remaining -= 20 + h.payload_len;
  }
}

static long fuzz_walks(long n_mut) {
  long n = 0;
  x3o_params op; x3o_params_default(&op);
  x3_params p; x3_params_default(&p);
  std::vector<Bytes> seeds;
  for (int v = 0; v < 3; ++v) {
    const size_t ns = v == 0 ? 25000 : (v == 1 ? 10001 : 333);
    std::vector<int16_t> wav(ns);
    for (size_t i = 0; i < ns; ++i) wav[i] = (int16_t)((i * 37 + (rng() % (v == 2 ? 20000 : 7))) & 0x7FFF);
    Bytes out(4 * ns + 4096); uint64_t pos = 0; uint64_t st[6];
    if (x3o_encode(wav.data(), ns, 1, &op, out.data(), out.size(), 0, &pos, st) != 0) { std::printf("oracle encode failed\n"); std::exit(2); }
    out.resize(pos);
    seeds.push_back(out);
  }
  for (long i = 0; i < n_mut; ++i) {
    Bytes s = seeds[rnd(seeds.size())];
    if (rnd(8)) {
      // damage in or near a header (the walk reads nothing else), CRC refreshed half of the time
      std::vector<uint64_t> offs; int t;
      ref_walk(s, 0, ~0ull, &offs, &t);
      if (!offs.empty()) {
        const uint64_t o = offs[rnd(offs.size())];
        const size_t at = o + rnd(20);
        if (rnd(2)) s[at] ^= (uint8_t)(1u << rnd(8)); else s[at] = (uint8_t)rnd(256);
        if (rnd(2)) { const uint16_t hc = x3o_crc16(&s[o], 16); s[o + 16] = (uint8_t)(hc >> 8); s[o + 17] = (uint8_t)hc; }
      }
      if (rnd(3) == 0) s.resize(rnd(s.size() + 1));
      if (rnd(5) == 0) s.insert(s.end(), rnd(40), (uint8_t)rnd(256));
    }
    const uint64_t phantom = rnd(4) == 0 ? 8 : 0;
    const uint64_t wav_cap = rnd(6) == 0 ? rnd(30000) : ~0ull;
    std::vector<uint64_t> want; int want_t;
    ref_walk(s, phantom, wav_cap, &want, &want_t);
    uint8_t* q = (uint8_t*)std::malloc(s.size() ? s.size() : 1);
    if (!s.empty()) std::memcpy(q, s.data(), s.size());
    HostWalk w;
    walk_host(q, s.size(), s.size(), s.size() + phantom, &p, wav_cap, ~0ull, &w, 1u);
    std::free(q);
    CHECK(!w.need_more, "walk: need_more with the whole stream in the window");
    CHECK(w.offs == want, "walk: %zu frames, reference %zu (len %zu)", w.offs.size(), want.size(), s.size());
    CHECK(w.terminal == want_t, "walk: terminal %d, reference %d (len %zu, frames %zu)", w.terminal, want_t, s.size(), want.size());
    // the same stream through a window that ends anywhere: the walk asks for more, never reads past the window
    if (!s.empty() && rnd(2)) {
      const size_t win = rnd(s.size() + 1);
      uint8_t* q2 = (uint8_t*)std::malloc(win ? win : 1);
      if (win) std::memcpy(q2, s.data(), win);
      HostWalk w2;
      walk_host(q2, win, s.size(), s.size() + phantom, &p, wav_cap, ~0ull, &w2, 1u);
      std::free(q2);
      CHECK(w2.offs.size() <= want.size() && std::equal(w2.offs.begin(), w2.offs.end(), want.begin()), "walk in a window: frames");
    }
    ++n;
  }
  return n;
}

static long fuzz_frame_headers(long n_mut) {
  long n = 0;
  uint8_t good[20];
  x3o_write_frame_header(10000, 1, 5000, 0x1234, good);
  for (long i = 0; i < n_mut; ++i) {
    uint8_t h[20];
    if (rnd(3) == 0) for (auto& b : h) b = (uint8_t)rnd(256);
    else {
      std::memcpy(h, good, 20);
      const int k = 1 + (int)rnd(3);
      for (int j = 0; j < k; ++j) h[rnd(20)] = (uint8_t)rnd(256);
      if (rnd(2)) { const uint16_t hc = x3o_crc16(h, 16); h[16] = (uint8_t)(hc >> 8); h[17] = (uint8_t)hc; }
    }
    one_frame_header(h); ++n;
  }
  return n;
}

// ---------------------------------------------------------------- shard arithmetic
static long fuzz_shards(long n_cases) {
  long n = 0;
  for (long i = 0; i < n_cases; ++i) {
    const uint64_t choices[] = {0, 1, 2, 7, 8, 63, 64, 69120, 552960, 1ull << 32, (1ull << 40) + 3, ~0ull >> 1, ~0ull};
    const uint64_t F = rnd(3) ? choices[rnd(13)] : rng();
    const int world = 1 + (int)rnd(rnd(4) ? 8 : 1024);
    uint64_t next = 0;
    for (int r = 0; r < world; ++r) {
      uint64_t first = 1, count = 1;
      x3_shard_frame_range(F, r, world, &first, &count);
      CHECK(first == next, "shard frames: rank %d starts at %llu, want %llu", r, (unsigned long long)first, (unsigned long long)next);
      next = first + count;
    }
    CHECK(next == F, "shard frames: the ranks cover %llu of %llu", (unsigned long long)next, (unsigned long long)F);
    x3_params p; x3_params_default(&p);
    if (rnd(3) == 0) { p.block_len = 1 + (uint32_t)rnd(60); p.blocks_per_frame = 1 + (uint32_t)rnd(1000); }
    const uint64_t N = rnd(3) ? choices[rnd(13)] : rng();
    uint64_t nx = 0;
    for (int r = 0; r < world; ++r) {
      uint64_t first = 1, count = 1;
      x3_shard_sample_range(N, &p, r, world, &first, &count);
      CHECK(first == nx && (first % ((uint64_t)p.block_len * p.blocks_per_frame) == 0 || first == N),
            "shard samples: rank %d first %llu", r, (unsigned long long)first);
      nx = first + count;
    }
    CHECK(nx == N, "shard samples: the ranks cover %llu of %llu", (unsigned long long)nx, (unsigned long long)N);
    std::vector<uint64_t> len(world), st(world + 1, 7);
    for (auto& l : len) l = rnd(1ull << 33) & ~1ull;
    x3_shard_offsets(len.data(), world, st.data());
    uint64_t acc = 0;
    for (int r = 0; r < world; ++r) { CHECK(st[r] == acc, "shard offsets"); acc += len[r]; }
    CHECK(st[world] == acc, "shard offsets: total");
    ++n;
  }
  return n;
}

int main(int argc, char** argv) {
  const long scale = argc > 1 ? std::atol(argv[1]) : 100000;
  memfd = memfd_create("x3fuzz", 0);
  if (memfd < 0) { std::perror("memfd_create"); return 2; }
  x3o_init();
  const long a = fuzz_archives(scale), w = fuzz_wavs(scale), h = fuzz_frame_headers(scale), k = fuzz_walks(scale / 4), s = fuzz_shards(scale / 50);
  if (fails) { std::printf("FAILED: %d differences\n", fails); return 1; }
  std::printf("ok archives=%ld wavs=%ld frame_headers=%ld walks=%ld shards=%ld\n", a, w, h, k, s);
  return 0;
}
