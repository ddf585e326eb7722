/*
 * x3_oracle.c -- CPU restatement of the psiphi75/x3-rust encode/decode hot path.
 * TEST INFRASTRUCTURE ONLY (see x3_oracle.h).  Deliberately structured like the
 * reference -- a byte-at-a-time MSB-first packer with a per-byte CRC table update, a
 * per-frame buffer, table-driven Rice codes, a 32-bit look-ahead bit reader -- so that
 * it also serves as the timed single-thread CPU baseline ("port").
 *
 * Paths in comments are relative to /root/reference.
 */
#include "x3_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

x3o_rice_code X3O_RICE[4];
int16_t X3O_INV_RICE[60];
uint16_t X3O_CRC_TABLE[256];
static int g_init_done = 0;

/* src/encoder.rs:228-231 */
static inline uint32_t count_bits(uint32_t n) { return n ? 32u - (uint32_t)__builtin_clz(n) : 0u; }

/*
 * Tables.  src/x3.rs:200-252 holds them as literals; here they are generated from the
 * closed form and tests/test_oracle_golden.py checks every entry
 * against the literals transcribed into tests/golden/rice_tables.json.
 *   d -> u = d>=0 ? 2d : -2d-1 ; codeword = (u>>k) zeros then (1<<k | (u & (2^k-1))) in k+1 bits
 *   (nsubs, offset, len, inv_len) = (0,6,14,16) (1,11,22,26) (2,20,40,44) (3,28,56,60)
 * CRC table: src/crc.rs:22-42, CRC-16/CCITT-FALSE polynomial 0x1021, MSB first.
 */
void x3o_init(void) {
  if (g_init_done) return;
  static const uint32_t offs[4] = {6, 11, 20, 28}, lens[4] = {14, 22, 40, 56}, invl[4] = {16, 26, 44, 60};
  for (uint32_t k = 0; k < 4; k++) {
    x3o_rice_code* rc = &X3O_RICE[k];
    rc->nsubs = k;
    rc->offset = offs[k];
    rc->len = lens[k];
    rc->inv_len = invl[k];
    for (uint32_t i = 0; i < rc->len; i++) {
      int32_t d = (int32_t)i - (int32_t)rc->offset;
      uint32_t u = d >= 0 ? (uint32_t)(2 * d) : (uint32_t)(-2 * d - 1);
      rc->code[i] = (1u << k) | (u & ((1u << k) - 1u));
      rc->num_bits[i] = (u >> k) + 1u + k;
    }
  }
  for (int i = 0; i < 60; i++) X3O_INV_RICE[i] = (int16_t)((i & 1) ? -((i + 1) / 2) : (i / 2));
  for (uint32_t b = 0; b < 256; b++) {
    uint16_t c = (uint16_t)(b << 8);
    for (int j = 0; j < 8; j++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x1021) : (c << 1));
    X3O_CRC_TABLE[b] = c;
  }
  g_init_done = 1;
}

/* ------------------------------------------------------------------ x3.rs */

/* src/x3.rs:124-134 with the constants of :90-96 */
void x3o_params_default(x3o_params* p) {
  p->block_len = 20;
  p->blocks_per_frame = 500;
  p->codes[0] = 0; p->codes[1] = 1; p->codes[2] = 3;
  p->thresholds[0] = 3; p->thresholds[1] = 8; p->thresholds[2] = 20;
}

/* src/x3.rs:98-122: only the first two thresholds are checked (k in 0..2).
 * RiceCodes::get indexes CODE[4] with the code numbers -> a code > 3 panics there. */
int x3o_params_new(const x3o_params* p) {
  x3o_init();
  for (int k = 0; k < 3; k++)
    if (p->codes[k] > 3) return X3O_BAD_ARG;
  for (int k = 0; k < 2; k++)
    if (p->thresholds[k] > X3O_RICE[p->codes[k]].offset) return X3O_INVALID_ENCODING_THRESH;
  return X3O_OK;
}

/* ----------------------------------------------------------------- crc.rs */

/* src/crc.rs:44-47 */
uint16_t x3o_update_crc16(uint16_t crc, uint8_t data) {
  uint8_t lookup = (uint8_t)(data ^ (uint8_t)(crc >> 8));
  return (uint16_t)((uint16_t)(crc << 8) ^ X3O_CRC_TABLE[lookup]);
}

/* src/crc.rs:49-58 */
uint16_t x3o_crc16(const uint8_t* data, size_t n) {
  x3o_init();
  uint16_t crc = 0xffff;
  for (size_t i = 0; i < n; i++) crc = x3o_update_crc16(crc, data[i]);
  return crc;
}

/* ---------------------------------------------------------- bytewriter.rs */

/* src/bytewriter.rs:33-41 */
void x3o_writer_init(x3o_writer* w, uint8_t* slice, size_t cap) {
  w->slice = slice; w->cap = cap; w->p_byte = 0; w->stream_length = 0;
}

/* src/bytewriter.rs:86-99 */
int x3o_writer_write_all(x3o_writer* w, const uint8_t* v, size_t n) {
  if (n > w->cap - w->p_byte) return X3O_BYTE_WRITER_INSUFFICIENT_MEMORY;
  memcpy(w->slice + w->p_byte, v, n);
  w->p_byte += n;
  if (w->p_byte > w->stream_length) w->stream_length = w->p_byte;
  return X3O_OK;
}

/* src/bytewriter.rs:44-54 */
int x3o_writer_align(x3o_writer* w, size_t n) {
  size_t residual = w->p_byte % n;
  if (residual == 0) return X3O_OK;
  static const uint8_t zeros[16] = {0};
  return x3o_writer_write_all(w, zeros, n - residual);
}

/* src/bytewriter.rs:60-80 (SeekFrom::Start) */
int x3o_writer_seek_start(x3o_writer* w, size_t pos) {
  if (pos > w->cap) return X3O_BYTE_WRITER_INSUFFICIENT_MEMORY;
  w->p_byte = pos;
  if (w->p_byte > w->stream_length) w->stream_length = w->p_byte;
  return X3O_OK;
}

/* src/bytewriter.rs:60-80 (SeekFrom::Current) */
int x3o_writer_seek_current(x3o_writer* w, int64_t off) {
  return x3o_writer_seek_start(w, (size_t)((int64_t)w->p_byte + off));
}

/* ----------------------------------------------------------- bitpacker.rs */

/* src/bitpacker.rs:65-73 */
void x3o_bp_new(x3o_bitpacker* bp, x3o_writer* w) {
  x3o_init();
  bp->writer = w; bp->scratch_byte = 0; bp->p_bit = 0; bp->byte_len = 0; bp->crc = 0xffff;
}

/* src/bitpacker.rs:79-86: CRC and length are updated before the 1-byte write */
static int bp_flush(x3o_bitpacker* bp) {
  bp->crc = x3o_update_crc16(bp->crc, bp->scratch_byte);
  bp->byte_len += 1;
  int rc = x3o_writer_write_all(bp->writer, &bp->scratch_byte, 1);
  if (rc) return rc;
  bp->scratch_byte = 0;
  bp->p_bit = 0;
  return X3O_OK;
}

/* src/bitpacker.rs:143-163 (the reference recurses for the straddling case; so do we) */
int x3o_bp_write_bits(x3o_bitpacker* bp, uint64_t value, size_t num_bits) {
  size_t rem_bit = 8 - bp->p_bit;
  uint64_t mask = (num_bits >= 64) ? ~0ull : ((1ull << num_bits) - 1ull);
  value &= mask;
  if (num_bits == rem_bit) {
    bp->scratch_byte |= (uint8_t)value;
    return bp_flush(bp);
  } else if (num_bits < rem_bit) {
    size_t shift_l = rem_bit - num_bits;
    bp->scratch_byte |= (uint8_t)(value << shift_l);
    bp->p_bit += num_bits;
    return X3O_OK;
  } else {
    size_t shift_r = num_bits - rem_bit;
    bp->scratch_byte |= (uint8_t)(value >> shift_r);
    int rc = bp_flush(bp);
    if (rc) return rc;
    return x3o_bp_write_bits(bp, value, shift_r);
  }
}

/* src/bitpacker.rs:174-176 */
int x3o_bp_write_packed_zeros(x3o_bitpacker* bp, size_t num_zeros) {
  return x3o_bp_write_bits(bp, 0, num_zeros);
}

/* src/bitpacker.rs:124-132: pad to a byte, then zero BYTES until the writer's ABSOLUTE
 * position is even; every pad byte goes through flush() so it is in len() and crc() */
int x3o_bp_word_align(x3o_bitpacker* bp) {
  int rc;
  if (bp->p_bit != 0 && (rc = bp_flush(bp))) return rc;
  while (bp->writer->p_byte % 2 != 0)
    if ((rc = bp_flush(bp))) return rc;
  return X3O_OK;
}

/* src/bitpacker.rs:95-102: length and CRC first, then ONE write_all of the array -- whatever sits in the scratch byte
 * stays there and reaches the writer later */
int x3o_bp_write_bytes(x3o_bitpacker* bp, const uint8_t* array, size_t n) {
  bp->byte_len += n;
  for (size_t i = 0; i < n; i++) bp->crc = x3o_update_crc16(bp->crc, array[i]);
  return x3o_writer_write_all(bp->writer, array, n);
}

/* src/bitpacker.rs:112-118: BitPackError::NotByteAligned off a byte boundary, else writer.seek(SeekFrom::Current(n)) */
int x3o_bp_inc_counter_n_bytes(x3o_bitpacker* bp, size_t n_bytes) {
  if (bp->p_bit != 0) return X3O_BITPACK;
  return x3o_writer_seek_current(bp->writer, (int64_t)n_bytes);
}

/* src/bitpacker.rs:56-62 */
int x3o_bp_drop(x3o_bitpacker* bp) {
  if (bp->p_bit != 0) return bp_flush(bp);
  return X3O_OK;
}

/* ------------------------------------------------------------- encoder.rs */

static inline void be16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }

/* src/encoder.rs:122-162.  Byte 3 ("num channels") is written with `id` (:135); the
 * 8 time bytes stay zero (:148-150); samples and payload_len are truncated `as u16`. */
void x3o_write_frame_header(size_t num_samples, uint8_t id, size_t payload_len, uint16_t payload_crc,
                            uint8_t out[20]) {
  memset(out, 0, 20);
  be16(out + 0, 30771); /* FrameHeader::KEY "x3", src/x3.rs:168 */
  out[2] = id;
  out[3] = id;
  be16(out + 4, (uint16_t)num_samples);
  be16(out + 6, (uint16_t)payload_len);
  be16(out + 16, x3o_crc16(out, 16));
  be16(out + 18, payload_crc);
}

/* src/encoder.rs:233-267 */
static int encode_rice_block(const int32_t* wav_diff, size_t n, x3o_bitpacker* bp, const x3o_params* p,
                             int32_t max_abs, size_t* ftype_out) {
  size_t ftype = 0;
  for (int t = 0; t < 3; t++)
    if (max_abs > (int32_t)p->thresholds[t]) ftype += 1;
  int rc = x3o_bp_write_bits(bp, ftype + 1, 2);
  if (rc) return rc;
  /* ftype <= 2 here because max_abs <= thresholds[2] -- unless the thresholds are not
   * ascending, in which case rice_codes[3] is an index panic in the reference */
  if (ftype > 2) return X3O_BAD_ARG;
  const x3o_rice_code* code = &X3O_RICE[p->codes[ftype]];
  for (size_t i = 0; i < n; i++) {
    int64_t ii = (int64_t)wav_diff[i] + (int64_t)code->offset;
    if (ii < 0 || ii >= (int64_t)code->len) return X3O_BAD_ARG; /* reference: index panic */
    uint32_t c = code->code[ii];
    uint32_t rc_num_bits = code->num_bits[ii];
    uint32_t num_zeros = rc_num_bits - count_bits(c);
    if ((rc = x3o_bp_write_packed_zeros(bp, num_zeros))) return rc;
    if ((rc = x3o_bp_write_bits(bp, c, rc_num_bits - num_zeros))) return rc;
  }
  *ftype_out = code->nsubs;
  return X3O_OK;
}

/* src/encoder.rs:269-276 */
static int encode_bfp_block(const int32_t* wav_diff, size_t n, x3o_bitpacker* bp, size_t num_bits,
                            size_t* ftype_out) {
  int rc = x3o_bp_write_bits(bp, num_bits, 6);
  if (rc) return rc;
  for (size_t i = 0; i < n; i++)
    if ((rc = x3o_bp_write_bits(bp, (uint64_t)(int64_t)wav_diff[i], num_bits + 1))) return rc;
  *ftype_out = 4;
  return X3O_OK;
}

/* src/encoder.rs:278-285: literal blocks carry the raw samples, not the diffs */
static int encode_literal(const int16_t* wav, size_t n, x3o_bitpacker* bp, size_t* ftype_out) {
  int rc = x3o_bp_write_bits(bp, 15, 6);
  if (rc) return rc;
  for (size_t i = 0; i < n; i++)
    if ((rc = x3o_bp_write_bits(bp, (uint64_t)(int64_t)wav[i], 16))) return rc;
  *ftype_out = 5;
  return X3O_OK;
}

/* src/encoder.rs:289-315 (+ diff :222-225).  `prev` is the sample before wav[0]. */
int x3o_encode_block(const int16_t* wav, size_t n, int16_t prev, x3o_bitpacker* bp,
                     const x3o_params* p, size_t* ftype_out) {
  int32_t wav_diff[60]; /* Parameters::MAX_BLOCK_LENGTH, src/x3.rs:90 */
  if (n > 60) return X3O_BAD_ARG; /* reference: index panic at :299 */
  int32_t max_abs = 0;
  int32_t last = prev;
  for (size_t i = 0; i < n; i++) {
    int32_t wd = (int32_t)wav[i] - last;
    last = wav[i];
    wav_diff[i] = wd;
    int32_t a = wd < 0 ? -wd : wd;
    if (a > max_abs) max_abs = a;
  }
  if (max_abs <= (int32_t)p->thresholds[2]) {
    return encode_rice_block(wav_diff, n, bp, p, max_abs, ftype_out);
  } else {
    size_t num_bits = count_bits((uint32_t)max_abs);
    if (num_bits >= 15) return encode_literal(wav, n, bp, ftype_out);
    return encode_bfp_block(wav_diff, n, bp, num_bits, ftype_out);
  }
}

/* src/encoder.rs:175-214 */
int x3o_encode_frame(const int16_t* wav, size_t n, x3o_writer* w, const x3o_params* p, uint64_t stats[6]) {
  int rc;
  x3o_init();
  if (n == 0) return X3O_BAD_ARG; /* reference: wav[0] panics */
  if ((rc = x3o_writer_align(w, 2))) return rc;
  size_t frame_header_pos = w->p_byte;
  if ((rc = x3o_writer_seek_current(w, 20))) return rc;

  x3o_bitpacker bp;
  x3o_bp_new(&bp, w);
  if ((rc = x3o_bp_write_bits(&bp, (uint64_t)(int64_t)wav[0], 16))) return rc; /* <Audio State> */
  if (p->block_len == 0 && n > 1) return X3O_BAD_ARG; /* chunks(0) panics */
  for (size_t s = 1; s < n; s += p->block_len) {
    size_t bl = n - s < p->block_len ? n - s : p->block_len;
    size_t ftype = 0;
    if ((rc = x3o_encode_block(wav + s, bl, wav[s - 1], &bp, p, &ftype))) return rc;
    stats[ftype] += bl;
  }
  if ((rc = x3o_bp_word_align(&bp))) return rc;
  size_t payload_len = bp.byte_len;
  uint16_t payload_crc = bp.crc;

  size_t return_position = w->p_byte;
  if ((rc = x3o_writer_seek_start(w, frame_header_pos))) return rc;
  uint8_t hdr[20];
  x3o_write_frame_header(n, 1, payload_len, payload_crc, hdr);
  if ((rc = x3o_writer_write_all(w, hdr, 20))) return rc;
  return x3o_writer_seek_start(w, return_position);
}

/* src/encoder.rs:51-111, std branch (:65-74): the sample iterator is cut into frames of
 * block_len*blocks_per_frame samples, each copied into its own frame buffer (the
 * reference's Vec<i16> per frame) and handed to encode_frame.  The writer is a
 * SliceByteWriter over out[0..out_cap) positioned at start_pos. */
int x3o_encode(const int16_t* wav, uint64_t n, uint32_t n_channels, const x3o_params* p,
               uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  x3o_init();
  if (n_channels > 1) return X3O_MORE_THAN_ONE_CHANNEL;
  if (n_channels == 0) return X3O_BAD_ARG; /* channels[0] panics */
  uint64_t st[6] = {0, 0, 0, 0, 0, 0};
  x3o_writer w;
  x3o_writer_init(&w, out, out_cap);
  if (start_pos > out_cap) return X3O_BYTE_WRITER_INSUFFICIENT_MEMORY;
  w.p_byte = w.stream_length = start_pos;
  size_t spf = (size_t)p->block_len * (size_t)p->blocks_per_frame;
  int16_t* frame_buffer = (int16_t*)malloc((spf ? spf : 1) * sizeof(int16_t));
  int rc = X3O_OK;
  uint64_t pos = 0;
  for (;;) {
    size_t take = (n - pos) < spf ? (size_t)(n - pos) : spf;
    if (take == 0) break;
    memcpy(frame_buffer, wav + pos, take * sizeof(int16_t));
    pos += take;
    if ((rc = x3o_encode_frame(frame_buffer, take, &w, p, st))) break;
  }
  free(frame_buffer);
  if (out_pos) *out_pos = w.p_byte;
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}

/* ----------------------------------------------------------- bitreader.rs */

/* src/bitreader.rs:29-49 */
static inline uint32_t read_word(const uint8_t* array, size_t len, size_t idx, size_t* nread) {
  if (len - idx >= 4) {
    *nread = 4;
    return ((uint32_t)array[idx] << 24) | ((uint32_t)array[idx + 1] << 16) | ((uint32_t)array[idx + 2] << 8) |
           (uint32_t)array[idx + 3];
  }
  size_t remaining = len - idx;
  uint32_t word = 0;
  if (remaining >= 1) word |= (uint32_t)array[idx] << 24;
  if (remaining >= 2) word |= (uint32_t)array[idx + 1] << 16;
  if (remaining == 3) word |= (uint32_t)array[idx + 2] << 8;
  *nread = remaining;
  return word;
}

/* src/bitreader.rs:65-74 */
void x3o_br_new(x3o_bitreader* br, const uint8_t* array, size_t len) {
  size_t idx;
  br->array = array;
  br->len = len;
  br->leading_word = read_word(array, len, 0, &idx);
  br->idx = idx;
  br->rem_bit = idx * 8;
}

/* src/bitreader.rs:148-175 (peek_next + get_next) */
static inline void br_get_next(x3o_bitreader* br) {
  if (br->idx >= br->len) {
    br->leading_word = 0;
    br->rem_bit = 0;
  } else {
    size_t diff;
    br->leading_word = read_word(br->array, br->len, br->idx, &diff);
    br->idx += diff;
    br->rem_bit = diff * 8;
  }
}

/* src/bitreader.rs:76-92.  Release-mode Rust masks an over-wide shift amount to its low
 * 5 bits (the debug_assert is compiled out); `& 31` reproduces that for n >= 32. */
static inline void br_inc_bits(x3o_bitreader* br, size_t n) {
  if (n < br->rem_bit) {
    br->leading_word <<= (n & 31);
    br->rem_bit -= n;
  } else if (n > br->rem_bit) {
    size_t rem = n - br->rem_bit;
    br_get_next(br);
    br->rem_bit = 32 - rem;
    br->leading_word <<= (rem & 31);
  } else {
    br_get_next(br);
  }
}

/* src/bitreader.rs:105-119 */
uint32_t x3o_br_read_nbits(x3o_bitreader* br, size_t n) {
  if (n <= br->rem_bit) {
    uint32_t result = br->leading_word >> ((32 - n) & 31);
    br_inc_bits(br, n);
    return result;
  } else {
    size_t rem = n - br->rem_bit;
    uint32_t result = br->leading_word >> ((32 - n) & 31);
    br_inc_bits(br, br->rem_bit);
    result |= br->leading_word >> ((32 - rem) & 31);
    br_inc_bits(br, rem);
    return result;
  }
}

/* src/bitreader.rs:128-139: the zero run is extended by at most ONE peeked word */
size_t x3o_br_count_zero_bits(x3o_bitreader* br) {
  size_t count = br->leading_word ? (size_t)__builtin_clz(br->leading_word) : 32;
  if (count > br->rem_bit) {
    if (br->idx >= br->len) {
      count = br->rem_bit;
    } else {
      size_t d;
      uint32_t word = read_word(br->array, br->len, br->idx, &d);
      count = br->rem_bit + (word ? (size_t)__builtin_clz(word) : 32);
    }
  }
  br_inc_bits(br, count);
  return count;
}

/* ------------------------------------------------------------- decoder.rs */

static inline uint16_t rd_be16(const uint8_t* p) { return (uint16_t)(((uint16_t)p[0] << 8) | p[1]); }

/* src/decoder.rs:69-118; check order: length, header CRC, key, channels, payload_len */
int x3o_read_frame_header(const uint8_t* bytes, size_t len, x3o_frame_header* h) {
  x3o_init();
  if (len < 20) return X3O_FRAME_DECODE_UNEXPECTED_END;
  if (rd_be16(bytes + 16) != x3o_crc16(bytes, 16)) return X3O_FRAME_HEADER_INVALID_HEADER_CRC;
  if (rd_be16(bytes) != 30771) return X3O_FRAME_HEADER_INVALID_KEY;
  uint8_t channels = bytes[3];
  if (channels > 1) return X3O_MORE_THAN_ONE_CHANNEL;
  uint32_t payload_len = rd_be16(bytes + 6);
  if (payload_len >= 0x7fe0) return X3O_FRAME_LENGTH; /* Frame::MAX_LENGTH src/x3.rs:145 */
  h->source_id = bytes[2];
  h->samples = rd_be16(bytes + 4);
  h->channels = channels;
  h->payload_len = payload_len;
  h->payload_crc = rd_be16(bytes + 18);
  return X3O_OK;
}

/* src/decoder.rs:147-170; i16 accumulation wraps (release build) */
static int decode_ricecode_block_r1(x3o_bitreader* br, int16_t* wav, size_t n, int16_t* last_wav,
                                    const x3o_params* p, size_t ftype) {
  const x3o_rice_code* code = &X3O_RICE[p->codes[ftype - 1]];
  int16_t lw = *last_wav;
  for (size_t b = 0; b < n; b++) {
    size_t i = x3o_br_count_zero_bits(br);
    x3o_br_read_nbits(br, 1);
    if (i >= code->inv_len) return X3O_OUT_OF_BOUNDS_INVERSE;
    lw = (int16_t)(uint16_t)((uint16_t)lw + (uint16_t)X3O_INV_RICE[i]);
    wav[b] = lw;
  }
  *last_wav = lw;
  return X3O_OK;
}

/* src/decoder.rs:172-196; nb is hard-wired: 2 bits for ftype 2, 4 bits for ftype 3 (:180) */
static int decode_ricecode_block_r2r3(x3o_bitreader* br, int16_t* wav, size_t n, int16_t* last_wav,
                                      const x3o_params* p, size_t ftype) {
  const x3o_rice_code* code = &X3O_RICE[p->codes[ftype - 1]];
  size_t nb = (ftype == 2) ? 2 : 4;
  int16_t level = (int16_t)(1 << code->nsubs);
  int16_t lw = *last_wav;
  for (size_t b = 0; b < n; b++) {
    int16_t nz = (int16_t)x3o_br_count_zero_bits(br);
    int16_t r = (int16_t)x3o_br_read_nbits(br, nb);
    size_t i = (size_t)(int64_t)(int16_t)(r + level * (nz - 1)); /* `as usize` sign-extends */
    if (i >= code->inv_len) return X3O_OUT_OF_BOUNDS_INVERSE;
    lw = (int16_t)(uint16_t)((uint16_t)lw + (uint16_t)X3O_INV_RICE[i]);
    wav[b] = lw;
  }
  *last_wav = lw;
  return X3O_OK;
}

/* src/decoder.rs:198-207 */
static inline int16_t unsigned_to_i16(uint16_t a16, size_t num_bits) {
  int32_t a = a16;
  int32_t neg_thresh = 1 << (num_bits - 1);
  int32_t neg = 1 << num_bits;
  if (a > neg_thresh) a -= neg;
  return (int16_t)a;
}

/* src/decoder.rs:209-235 */
static int decode_bpf_block(x3o_bitreader* br, int16_t* wav, size_t n, int16_t* last_wav) {
  size_t num_bits = (size_t)x3o_br_read_nbits(br, 4) + 1;
  if (num_bits <= 5) return X3O_FRAME_DECODE_INVALID_BPF;
  if (num_bits == 16) {
    for (size_t i = 0; i < n; i++) wav[i] = (int16_t)x3o_br_read_nbits(br, 16);
  } else {
    int16_t value = *last_wav;
    for (size_t i = 0; i < n; i++) {
      uint16_t diff = (uint16_t)x3o_br_read_nbits(br, num_bits);
      value = (int16_t)(uint16_t)((uint16_t)value + (uint16_t)unsigned_to_i16(diff, num_bits));
      wav[i] = value;
    }
  }
  if (n == 0) return X3O_BAD_ARG; /* wav[wav.len()-1] panics */
  *last_wav = wav[n - 1];
  return X3O_OK;
}

/* src/decoder.rs:132-145 */
int x3o_decode_block(x3o_bitreader* br, int16_t* wav, size_t n, int16_t* last_wav, const x3o_params* p) {
  x3o_init();
  size_t ftype = x3o_br_read_nbits(br, 2);
  switch (ftype) {
    case 0: return decode_bpf_block(br, wav, n, last_wav);
    case 1: return decode_ricecode_block_r1(br, wav, n, last_wav, p, ftype);
    case 2:
    case 3: return decode_ricecode_block_r2r3(br, wav, n, last_wav, p, ftype);
    default: return X3O_FRAME_DECODE_INVALID_FTYPE;
  }
}

/* src/decoder.rs:36-58.  CRC is NOT checked here and trailing bits are ignored. */
int x3o_decode_frame(const uint8_t* x3_bytes, size_t len, int16_t* wav_buf, size_t wav_cap,
                     const x3o_params* p, size_t samples, size_t* n_out) {
  x3o_init();
  if (len < 2 || samples == 0 || wav_cap < 1) return X3O_BAD_ARG; /* reference panics */
  int16_t last_wav = (int16_t)rd_be16(x3_bytes);
  size_t p_wav = 0;
  wav_buf[p_wav++] = last_wav;
  x3o_bitreader br;
  x3o_br_new(&br, x3_bytes + 2, len - 2);
  size_t remaining = samples - 1;
  /* block_len == 0 (a damaged archive header can say so): every turn decodes an empty block and `remaining` stays --
   * Rice blocks read their two type bits and nothing else, a BFP block fails (E <= 5) or panics on wav[len - 1]; behind
   * the payload the reader yields zeros, which is a BFP block with E = 1: the loop ends there at the latest.  (The
   * turn limit only guards this restatement.) */
  size_t turns = 0;
  while (remaining > 0) {
    size_t block_len = remaining < p->block_len ? remaining : p->block_len;
    if (p_wav + block_len > wav_cap) return X3O_BAD_ARG; /* slice index panic */
    if (block_len == 0 && ++turns > 4 * len + 64) return X3O_BAD_ARG; /* never returns */
    int rc = x3o_decode_block(&br, wav_buf + p_wav, block_len, &last_wav, p);
    if (rc) return rc;
    remaining -= block_len;
    p_wav += block_len;
  }
  if (n_out) *n_out = p_wav;
  return X3O_OK;
}

/* ---------------------------------------------------------- decodefile.rs */

/* X3aReader::decode_next_frame (src/decodefile.rs:105-136) + read_bytes (:80-86) +
 * read_frame_payload (:93-103), looped as x3a_to_wav does (:200-209), over memory. */
static int decode_stream_phantom(const uint8_t* x3, uint64_t len, uint64_t phantom, const x3o_params* p,
                                 int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                                 uint64_t* frame_errors);

int x3o_decode_stream(const uint8_t* x3, uint64_t len, const x3o_params* p, int16_t* wav,
                      uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  return decode_stream_phantom(x3, len, 0, p, wav, wav_cap, n_out, frames_ok, frame_errors);
}

/* `phantom` = bytes X3aReader believes remain beyond the real data: open() subtracts the archive
 * header WITHOUT its 8-byte id from the file length (decodefile.rs:61-66), and read_bytes clamps only
 * to that count (:80-86), so a read past the real end is a read_exact failure = X3Error::Io. */
static int decode_stream_phantom(const uint8_t* x3, uint64_t len, uint64_t phantom, const x3o_params* p,
                                 int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                                 uint64_t* frame_errors) {
  x3o_init();
  uint64_t pos = 0, remaining = len + phantom, nsamp = 0, nframes = 0, nerr = 0;
  int rc = X3O_OK;
  for (;;) {
    if (remaining <= 20) break;                                   /* :107-109 */
    if (len - pos < 20) { rc = X3O_IO; break; }
    x3o_frame_header h;
    rc = x3o_read_frame_header(x3 + pos, 20, &h);                 /* :112 */
    pos += 20; remaining -= 20;
    if (rc) break;
    if (remaining < h.payload_len) break;                         /* :114-116, Ok(None) */
    if (h.payload_len > 1024 * 24) { rc = X3O_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; } /* :118-121 */
    if (len - pos < h.payload_len) { rc = X3O_IO; break; }
    const uint8_t* payload = x3 + pos;
    pos += h.payload_len; remaining -= h.payload_len;
    if (x3o_crc16(payload, h.payload_len) != h.payload_crc) {     /* :96-100 */
      rc = X3O_FRAME_HEADER_INVALID_PAYLOAD_CRC;
      break;
    }
    size_t got = 0;
    int drc = x3o_decode_frame(payload, h.payload_len, wav + nsamp, (size_t)(wav_cap - nsamp), p,
                               h.samples, &got);                  /* :128 */
    if (drc == X3O_BAD_ARG) { rc = drc; break; }                  /* reference would panic */
    if (drc) { nerr += 1; break; }                                /* :129-135: counted, Ok(None) */
    nsamp += got;
    nframes += 1;
  }
  if (n_out) *n_out = nsamp;
  if (frames_ok) *frames_ok = nframes;
  if (frame_errors) *frame_errors = nerr;
  return rc;
}

/* --------------------------------------------------------------- multi-channel (extension) */

/* NOT in the reference: encoder::encode returns MoreThanOneChannel for more than one channel (encoder.rs:55-57) and
 * read_frame_header rejects a frame whose <Num Channels> is above one (decoder.rs:90-94).  What the format foresees is in
 * the header -- a channel count (x3.rs:155-156, encoder.rs:134) -- and in encode_frame's own comment, "pack the data block
 * for each channel" (encoder.rs:197).  The extension follows that: <Audio State> = the first sample of every channel,
 * 16 bits each, in channel order; then, for every block index, the block of channel 0, 1, ... C-1, each coded exactly
 * like a mono block against ITS channel's previous sample; then word_align.  Header: source id 1, byte 3 = C,
 * `samples` = samples per channel.  With C = 1 every byte is the reference's (tests pin that).  Parity unpinned by the
 * reference for C > 1: there is nothing to compare with. */
int x3o_encode_frame_mc(const int16_t* const* wavs, uint32_t n_ch, size_t n, x3o_writer* w, const x3o_params* p,
                        uint64_t stats[6]) {
  int rc;
  x3o_init();
  if (n == 0 || n_ch == 0 || n_ch > 255) return X3O_BAD_ARG;
  if ((rc = x3o_writer_align(w, 2))) return rc;
  size_t frame_header_pos = w->p_byte;
  if ((rc = x3o_writer_seek_current(w, 20))) return rc;
  x3o_bitpacker bp;
  x3o_bp_new(&bp, w);
  for (uint32_t c = 0; c < n_ch; c++)
    if ((rc = x3o_bp_write_bits(&bp, (uint64_t)(int64_t)wavs[c][0], 16))) return rc; /* <Audio State>, per channel */
  if (p->block_len == 0 && n > 1) return X3O_BAD_ARG;
  for (size_t s = 1; s < n; s += p->block_len) {
    size_t bl = n - s < p->block_len ? n - s : p->block_len;
    for (uint32_t c = 0; c < n_ch; c++) {
      size_t ftype = 0;
      if ((rc = x3o_encode_block(wavs[c] + s, bl, wavs[c][s - 1], &bp, p, &ftype))) return rc;
      stats[ftype] += bl;
    }
  }
  if ((rc = x3o_bp_word_align(&bp))) return rc;
  size_t payload_len = bp.byte_len;
  uint16_t payload_crc = bp.crc;
  if (payload_len > 1024 * 24) return X3O_FRAME_LENGTH; /* no reader would take it (decodefile.rs:118-121) */
  size_t return_position = w->p_byte;
  if ((rc = x3o_writer_seek_start(w, frame_header_pos))) return rc;
  uint8_t hdr[20];
  x3o_write_frame_header(n, 1, payload_len, payload_crc, hdr);
  hdr[3] = (uint8_t)n_ch;
  be16(hdr + 16, x3o_crc16(hdr, 16));
  if ((rc = x3o_writer_write_all(w, hdr, 20))) return rc;
  return x3o_writer_seek_start(w, return_position);
}

int x3o_encode_mc(const int16_t* const* wavs, uint32_t n_ch, uint64_t n, const x3o_params* p, uint8_t* out,
                  uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  x3o_init();
  if (n_ch == 0 || n_ch > 255) return X3O_BAD_ARG;
  uint64_t st[6] = {0, 0, 0, 0, 0, 0};
  x3o_writer w;
  x3o_writer_init(&w, out, out_cap);
  if (start_pos > out_cap) return X3O_BYTE_WRITER_INSUFFICIENT_MEMORY;
  w.p_byte = w.stream_length = start_pos;
  size_t spf = (size_t)p->block_len * (size_t)p->blocks_per_frame;
  const int16_t* at[255];
  int rc = X3O_OK;
  uint64_t pos = 0;
  for (;;) {
    size_t take = (n - pos) < spf ? (size_t)(n - pos) : spf;
    if (take == 0) break;
    for (uint32_t c = 0; c < n_ch; c++) at[c] = wavs[c] + pos;
    pos += take;
    if ((rc = x3o_encode_frame_mc(at, n_ch, take, &w, p, st))) break;
  }
  if (out_pos) *out_pos = w.p_byte;
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}

/* decode_frame (decoder.rs:36-58) with n_ch predictors: wavs[c][0 .. samples) */
int x3o_decode_frame_mc(const uint8_t* x3_bytes, size_t len, int16_t* const* wavs, size_t wav_cap, uint32_t n_ch,
                        const x3o_params* p, size_t samples, size_t* n_out) {
  x3o_init();
  if (n_ch == 0 || n_ch > 255 || len < 2u * n_ch || samples == 0 || wav_cap < 1) return X3O_BAD_ARG;
  int16_t last[255];
  for (uint32_t c = 0; c < n_ch; c++) {
    last[c] = (int16_t)rd_be16(x3_bytes + 2 * c);
    wavs[c][0] = last[c];
  }
  size_t p_wav = 1;
  x3o_bitreader br;
  x3o_br_new(&br, x3_bytes + 2 * n_ch, len - 2 * n_ch);
  size_t remaining = samples - 1;
  size_t turns = 0;
  while (remaining > 0) {
    size_t block_len = remaining < p->block_len ? remaining : p->block_len;
    if (p_wav + block_len > wav_cap) return X3O_BAD_ARG;
    if (block_len == 0 && ++turns > 4 * len + 64) return X3O_BAD_ARG;
    for (uint32_t c = 0; c < n_ch; c++) {
      int rc = x3o_decode_block(&br, wavs[c] + p_wav, block_len, &last[c], p);
      if (rc) return rc;
    }
    remaining -= block_len;
    p_wav += block_len;
  }
  if (n_out) *n_out = p_wav;
  return X3O_OK;
}

/* the frame walk of decode_stream_phantom for frames of n_ch channels: the header's <Num Channels> must say n_ch */
int x3o_decode_stream_mc(const uint8_t* x3, uint64_t len, uint32_t n_ch, const x3o_params* p, int16_t* const* wavs,
                         uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  x3o_init();
  if (n_ch == 0 || n_ch > 255) return X3O_BAD_ARG;
  uint64_t pos = 0, remaining = len, nsamp = 0, nframes = 0, nerr = 0;
  int rc = X3O_OK;
  int16_t* at[255];
  for (;;) {
    if (remaining <= 20) break;
    if (len - pos < 20) { rc = X3O_IO; break; }
    x3o_frame_header h;
    uint8_t hb[20];
    memcpy(hb, x3 + pos, 20);
    /* read_frame_header with the channel test turned into "is it n_ch" (check order kept: length, CRC, key, channels) */
    rc = X3O_OK;
    if (rd_be16(hb + 16) != x3o_crc16(hb, 16)) rc = X3O_FRAME_HEADER_INVALID_HEADER_CRC;
    else if (rd_be16(hb) != 30771) rc = X3O_FRAME_HEADER_INVALID_KEY;
    else if (hb[3] != n_ch) rc = X3O_MORE_THAN_ONE_CHANNEL;
    else if (rd_be16(hb + 6) >= 0x7fe0) rc = X3O_FRAME_LENGTH;
    h.samples = rd_be16(hb + 4);
    h.payload_len = rd_be16(hb + 6);
    h.payload_crc = rd_be16(hb + 18);
    pos += 20; remaining -= 20;
    if (rc) break;
    if (remaining < h.payload_len) break;
    if (h.payload_len > 1024 * 24) { rc = X3O_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; }
    if (len - pos < h.payload_len) { rc = X3O_IO; break; }
    const uint8_t* payload = x3 + pos;
    pos += h.payload_len; remaining -= h.payload_len;
    if (x3o_crc16(payload, h.payload_len) != h.payload_crc) { rc = X3O_FRAME_HEADER_INVALID_PAYLOAD_CRC; break; }
    size_t got = 0;
    for (uint32_t c = 0; c < n_ch; c++) at[c] = wavs[c] + nsamp;
    int drc = x3o_decode_frame_mc(payload, h.payload_len, at, (size_t)(wav_cap - nsamp), n_ch, p, h.samples, &got);
    if (drc == X3O_BAD_ARG) { rc = drc; break; }
    if (drc) { nerr += 1; break; }
    nsamp += got;
    nframes += 1;
  }
  if (n_out) *n_out = nsamp;
  if (frames_ok) *frames_ok = nframes;
  if (frame_errors) *frame_errors = nerr;
  return rc;
}

/* ----------------------------------------------- encodefile.rs / decodefile.rs: archive */

#include <stdio.h>

/* src/encodefile.rs:82-138 */
int x3o_archive_header_write(uint32_t sample_rate, const x3o_params* p, uint8_t* out, uint64_t cap,
                             uint64_t* out_len) {
  x3o_init();
  char xml[600];
  int n = snprintf(xml, sizeof xml,
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
                   sample_rate, p->block_len, p->codes[0], p->codes[1], p->codes[2], p->thresholds[0],
                   p->thresholds[1], p->thresholds[2]);
  size_t payload_len = (size_t)n;
  uint16_t payload_crc = x3o_crc16((const uint8_t*)xml, payload_len);
  if (payload_len % 2 == 1) {                     /* :123-128 */
    xml[payload_len++] = 0;
    payload_crc = x3o_update_crc16(payload_crc, 0);
  }
  uint64_t total = 8 + 20 + payload_len;
  if (out_len) *out_len = total;
  if (total > cap) return X3O_BYTE_WRITER_INSUFFICIENT_MEMORY;
  memcpy(out, "X3ARCHIV", 8);                     /* Archive::ID, src/x3.rs:139 */
  x3o_write_frame_header(0, 0, payload_len, payload_crc, out + 8);   /* :134 */
  memcpy(out + 28, xml, payload_len);
  return X3O_OK;
}

/* -> the trimmed text of the first <NAME ...>text</NAME> element as (pointer into xml, length): the text may hold any
 * byte, NUL included (a Rust &str does), and is as long as the payload lets it be */
static int xml_first_text(const char* xml, size_t len, const char* name, const char** text, size_t* tlen) {
  size_t nl = strlen(name);
  for (size_t i = 0; i + nl + 1 < len; i++) {
    if (xml[i] != '<' || memcmp(xml + i + 1, name, nl) != 0) continue;
    char nx = xml[i + 1 + nl];
    if (!(nx == '>' || nx == ' ' || nx == '\t' || nx == '\n' || nx == '\r')) continue;
    size_t gt = i + 1 + nl;
    while (gt < len && xml[gt] != '>') gt++;
    if (gt >= len || xml[gt - 1] == '/') return 0;
    for (size_t c = gt + 1; c + nl + 3 <= len; c++) {
      if (xml[c] == '<' && xml[c + 1] == '/' && memcmp(xml + c + 2, name, nl) == 0 && xml[c + 2 + nl] == '>') {
        size_t a = gt + 1, b = c;
        while (a < b && (xml[a] == ' ' || xml[a] == '\t' || xml[a] == '\r' || xml[a] == '\n')) a++;
        while (b > a && (xml[b - 1] == ' ' || xml[b - 1] == '\t' || xml[b - 1] == '\r' || xml[b - 1] == '\n')) b--;
        *text = xml + a;
        *tlen = b - a;
        return 1;
      }
    }
    return 0;
  }
  return 0;
}

static int parse_u32(const char* s, size_t n, uint32_t* v) {
  if (n == 0 || n > 10) return 0;
  size_t i = s[0] == '+' ? 1 : 0;
  if (i == n) return 0;
  uint64_t acc = 0;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return 0;
    acc = acc * 10 + (uint64_t)(s[i] - '0');
  }
  if (acc > 0xFFFFFFFFull) return 0;
  *v = (uint32_t)acc;
  return 1;
}

/* src/decodefile.rs:142-176 + parse_xml :232-303 */
int x3o_archive_header_read(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, x3o_params* p,
                            uint8_t* channels, uint64_t* header_size) {
  x3o_init();
  if (len < 8) return X3O_IO;
  if (memcmp(bytes, "X3ARCHIV", 8) != 0) return X3O_ARCHIVE_HEADER_XML_INVALID_KEY;
  if (len < 28) return X3O_IO;
  x3o_frame_header h;
  int rc = x3o_read_frame_header(bytes + 8, 20, &h);
  if (rc) return rc;
  if (len - 28 < h.payload_len) return X3O_IO;
  const char* xml = (const char*)bytes + 28;
  const char *fs, *bl, *codes, *th;
  size_t fsl, bll, codesl, thl;
  if (!xml_first_text(xml, h.payload_len, "FS", &fs, &fsl) ||
      !xml_first_text(xml, h.payload_len, "BLKLEN", &bl, &bll) ||
      !xml_first_text(xml, h.payload_len, "CODES", &codes, &codesl) ||
      !xml_first_text(xml, h.payload_len, "T", &th, &thl))
    return X3O_BAD_ARG; /* fs[0] etc. index panic */
  uint32_t rate, block_len;
  if (!parse_u32(fs, fsl, &rate) || !parse_u32(bl, bll, &block_len)) return X3O_BAD_ARG;
  uint32_t ids[16], ths[16];
  size_t nid = 0, nth = 0;
  for (const char *w = codes, *end = codes + codesl;;) {   /* str::split(','): the words between the commas, all of them */
    const char* e = (const char*)memchr(w, ',', (size_t)(end - w));
    size_t wl = e ? (size_t)(e - w) : (size_t)(end - w);
    if (wl == 5 && !memcmp(w, "RICE", 4) && w[4] >= '0' && w[4] <= '3') { if (nid < 16) ids[nid++] = (uint32_t)(w[4] - '0'); }
    else if (!(wl == 3 && !memcmp(w, "BFP", 3))) return X3O_ARCHIVE_HEADER_XML_RICE_CODE;
    if (!e) break;
    w = e + 1;
  }
  for (const char *w = th, *end = th + thl;;) {
    const char* e = (const char*)memchr(w, ',', (size_t)(end - w));
    size_t wl = e ? (size_t)(e - w) : (size_t)(end - w);
    uint32_t v;
    if (!parse_u32(w, wl, &v)) return X3O_BAD_ARG;
    if (nth < 16) ths[nth++] = v;
    if (!e) break;
    w = e + 1;
  }
  if (nid < 3 || nth < 3) return X3O_BAD_ARG;
  x3o_params q;
  q.block_len = block_len;
  q.blocks_per_frame = 500;
  for (int k = 0; k < 3; k++) { q.codes[k] = ids[k]; q.thresholds[k] = ths[k]; }
  if ((rc = x3o_params_new(&q))) return rc;
  *p = q;
  if (sample_rate) *sample_rate = rate;
  if (channels) *channels = h.channels;
  if (header_size) *header_size = 20 + (uint64_t)h.payload_len;
  return X3O_OK;
}

/* src/encodefile.rs:48-77 without the files */
int x3o_x3a_encode(const int16_t* wav, uint64_t n, uint32_t sample_rate, uint8_t* out, uint64_t cap,
                   uint64_t* out_len, uint64_t stats[6]) {
  x3o_params p;
  x3o_params_default(&p);
  uint64_t hlen = 0;
  int rc = x3o_archive_header_write(sample_rate, &p, out, cap, &hlen);
  if (out_len) *out_len = hlen;
  if (rc) return rc;
  uint64_t pos = hlen;
  rc = x3o_encode(wav, n, 1, &p, out, cap, hlen, &pos, stats);
  if (out_len) *out_len = pos;
  return rc;
}

/* src/decodefile.rs:59-136,189-212 without the files */
int x3o_x3a_decode(const uint8_t* x3a, uint64_t len, int16_t* wav, uint64_t wav_cap, uint64_t* n_out,
                   uint32_t* sample_rate, uint64_t* frames_ok, uint64_t* frame_errors) {
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  x3o_params p;
  uint8_t ch;
  uint64_t hsize;
  int rc = x3o_archive_header_read(x3a, len, sample_rate, &p, &ch, &hsize);
  if (rc) return rc;
  uint64_t start = 8 + hsize;
  return decode_stream_phantom(x3a + start, len - start, 8, &p, wav, wav_cap, n_out, frames_ok, frame_errors);
}

/* ------------------------------------------------------------ files (hound restated, see x3_oracle.h) */

static uint32_t le32(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
static uint16_t le16(const uint8_t* b) { return (uint16_t)(b[0] | (b[1] << 8)); }
static void put_le32(uint8_t* b, uint32_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 24); }
static void put_le16(uint8_t* b, uint16_t v) { b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); }

/* hound::WavReader::new: "RIFF" size "WAVE", then chunks until "data"; fmt must come first */
int x3o_wav_parse(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, uint16_t* channels, uint16_t* bits,
                  uint64_t* data_off, uint64_t* data_len) {
  if (len < 12) return X3O_IO;
  if (memcmp(bytes, "RIFF", 4) != 0 || memcmp(bytes + 8, "WAVE", 4) != 0) return X3O_BAD_ARG;
  uint64_t pos = 12;
  int have_fmt = 0;
  for (;;) {
    if (len - pos < 8) return X3O_IO; /* end of file without a data chunk */
    const uint8_t* h = bytes + pos;
    uint32_t clen = le32(h + 4);
    pos += 8;
    if (memcmp(h, "fmt ", 4) == 0) {
      if (clen < 16) return X3O_BAD_ARG;
      if (len - pos < clen) return X3O_IO;
      uint16_t tag = le16(bytes + pos);
      *channels = le16(bytes + pos + 2);
      *sample_rate = le32(bytes + pos + 4);
      *bits = le16(bytes + pos + 14);
      if (tag == 0xFFFE) { /* WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag */
        if (clen < 40) return X3O_BAD_ARG;
        tag = le16(bytes + pos + 24);
      }
      if (tag != 1) return X3O_BAD_ARG; /* integer PCM only (the reference reads i16 samples) */
      have_fmt = 1;
    } else if (memcmp(h, "data", 4) == 0) {
      if (!have_fmt) return X3O_BAD_ARG;
      *data_off = pos;
      *data_len = clen;
      return X3O_OK;
    }
    uint64_t skip = (uint64_t)clen + (clen & 1u); /* chunks are word aligned */
    if (len - pos < skip) return X3O_IO;
    pos += skip;
  }
}

/* hound::WavWriter for {channels 1, 16 bit, Int}: PCMWAVEFORMAT header, sizes as finalize() leaves them */
void x3o_wav_header_write(uint32_t sample_rate, uint64_t n_samples, uint8_t out[44]) {
  uint32_t data_len = (uint32_t)(n_samples * 2);
  memcpy(out, "RIFF", 4);
  put_le32(out + 4, 36 + data_len);
  memcpy(out + 8, "WAVEfmt ", 8);
  put_le32(out + 16, 16);
  put_le16(out + 20, 1);
  put_le16(out + 22, 1);
  put_le32(out + 24, sample_rate);
  put_le32(out + 28, sample_rate * 2);
  put_le16(out + 32, 2);
  put_le16(out + 34, 16);
  memcpy(out + 36, "data", 4);
  put_le32(out + 40, data_len);
}

static uint8_t* read_whole_file(const char* path, uint64_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* b = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
  if (b && n > 0 && fread(b, 1, (size_t)n, f) != (size_t)n) { free(b); b = NULL; }
  fclose(f);
  *len = n > 0 ? (uint64_t)n : 0;
  return b;
}

/* src/encodefile.rs:48-77 */
int x3o_wav_to_x3a(const char* wav_path, const char* x3a_path, uint64_t stats[6]) {
  uint64_t st_local[6];
  if (!stats) stats = st_local;
  memset(stats, 0, 6 * sizeof(uint64_t));
  uint64_t len = 0;
  uint8_t* file = read_whole_file(wav_path, &len);
  if (!file) return X3O_IO; /* WavReader::open(..).unwrap() */
  uint32_t rate = 0;
  uint16_t ch = 0, bits = 0;
  uint64_t off = 0, dlen = 0;
  int rc = x3o_wav_parse(file, len, &rate, &ch, &bits, &off, &dlen);
  if (rc) { free(file); return rc; }
  if (bits != 16 || ch != 1) { free(file); return X3O_BAD_ARG; } /* assert_eq!, :53,56 */
  if (dlen & 1u) { free(file); return X3O_BAD_ARG; }               /* hound: not a multiple of the sample size */
  FILE* out = fopen(x3a_path, "wb");                               /* File::create(..)?, :66 */
  if (!out) { free(file); return X3O_IO; }
  uint64_t n = dlen / 2, avail = (len - off) / 2;
  int truncated = avail < n; /* samples().map(|x| x.unwrap()) panics at the first missing sample */
  if (truncated) n = avail;
  x3o_params p;
  x3o_params_default(&p);
  uint64_t spf = (uint64_t)p.block_len * p.blocks_per_frame, nf = (n + spf - 1) / spf;
  uint64_t cap = 1024 + nf * (20 + 2 * spf + spf / 8 + 64);
  uint8_t* buf = (uint8_t*)malloc(cap);
  int16_t* wav = (int16_t*)malloc((n ? n : 1) * sizeof(int16_t));
  if (!buf || !wav) { free(buf); free(wav); free(file); fclose(out); return X3O_BAD_ARG; }
  for (uint64_t i = 0; i < n; i++) wav[i] = (int16_t)le16(file + off + 2 * i);
  uint64_t olen = 0;
  rc = x3o_x3a_encode(wav, n, rate, buf, cap, &olen, stats);
  if (!rc && fwrite(buf, 1, olen, out) != olen) rc = X3O_IO;
  if (fclose(out) != 0 && !rc) rc = X3O_IO;
  free(buf); free(wav); free(file);
  if (!rc && truncated) rc = X3O_IO;
  return rc;
}

/* src/decodefile.rs:189-227 */
int x3o_x3a_to_wav(const char* x3a_path, const char* wav_path, uint64_t* n_samples, uint64_t* frame_errors) {
  if (n_samples) *n_samples = 0;
  if (frame_errors) *frame_errors = 0;
  uint64_t len = 0;
  uint8_t* file = read_whole_file(x3a_path, &len);
  if (!file) return X3O_IO; /* File::open(..).unwrap(), :60 */
  x3o_params p;
  uint32_t rate = 0;
  uint8_t ch;
  uint64_t hsize;
  int rc = x3o_archive_header_read(file, len, &rate, &p, &ch, &hsize); /* X3aReader::open before the writer exists */
  if (rc) { free(file); return rc; }
  FILE* out = fopen(wav_path, "wb"); /* WavWriter::create(..)?, :201 */
  if (!out) { free(file); return X3O_HOUND; }
  uint64_t cap = 0; /* samples the headers of the stream promise (an upper bound of what the walk yields) */
  for (uint64_t pos = 8 + hsize; len - pos >= 20;) {
    x3o_frame_header h;
    if (x3o_read_frame_header(file + pos, 20, &h)) break;
    cap += h.samples;
    if (len - pos - 20 < h.payload_len) break;
    pos += 20 + (uint64_t)h.payload_len;
  }
  int16_t* wav = (int16_t*)malloc((cap ? cap : 1) * sizeof(int16_t));
  if (!wav) { free(file); fclose(out); return X3O_BAD_ARG; }
  uint64_t n = 0, fok = 0, ferr = 0;
  uint32_t rate2;
  rc = x3o_x3a_decode(file, len, wav, cap, &n, &rate2, &fok, &ferr);
  /* whatever ended the walk, the writer is dropped and finalised with what it got */
  uint8_t hdr[44];
  x3o_wav_header_write(rate, n, hdr);
  int wrc = X3O_OK;
  if (fwrite(hdr, 1, 44, out) != 44) wrc = X3O_IO;
  for (uint64_t i = 0; i < n && !wrc; i++) {
    uint8_t s[2];
    put_le16(s, (uint16_t)wav[i]);
    if (fwrite(s, 1, 2, out) != 2) wrc = X3O_IO;
  }
  if (fclose(out) != 0 && !wrc) wrc = X3O_IO;
  free(wav); free(file);
  if (n_samples) *n_samples = n;
  if (frame_errors) *frame_errors = ferr;
  return rc ? rc : wrc;
}

/* ------------------------------------------------------------ CPU baseline */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int x3o_time_roundtrip(const int16_t* wav, uint64_t n, const x3o_params* p, int reps, double* enc_s,
                       double* dec_s, uint64_t* stream_len) {
  x3o_init();
  size_t spf = (size_t)p->block_len * p->blocks_per_frame;
  if (spf == 0) return X3O_BAD_ARG;
  uint64_t nframes = (n + spf - 1) / spf;
  uint64_t cap = nframes * (20 + 2 * (uint64_t)spf + spf / 8 + 64) + 64;
  uint8_t* out = (uint8_t*)malloc(cap);
  int16_t* back = (int16_t*)malloc((n ? n : 1) * sizeof(int16_t));
  if (!out || !back) { free(out); free(back); return X3O_BAD_ARG; }
  uint64_t stats[6], pos = 0, nout = 0, fok = 0, ferr = 0;
  int rc = X3O_OK;
  double te = 0, td = 0;
  for (int r = 0; r < reps && !rc; r++) {
    double t0 = now_s();
    rc = x3o_encode(wav, n, 1, p, out, cap, 0, &pos, stats);
    double t1 = now_s();
    if (rc) break;
    rc = x3o_decode_stream(out, pos, p, back, n, &nout, &fok, &ferr);
    double t2 = now_s();
    te += t1 - t0;
    td += t2 - t1;
    if (!rc && (nout != n || memcmp(back, wav, n * sizeof(int16_t)) != 0)) rc = X3O_BAD_ARG;
  }
  if (enc_s) *enc_s = te;
  if (dec_s) *dec_s = td;
  if (stream_len) *stream_len = pos;
  free(out);
  free(back);
  return rc;
}
