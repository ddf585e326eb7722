/*
 * x3_oracle.h -- CPU restatement of the psiphi75/x3-rust encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (x3-rust_amd/csrc) never includes, links or calls anything here.
 *
 * Parity status: PINNED for block coding, bit packing, frame header, CRC-16 and
 * word-align padding by the reference's own known-answer vectors (the JSON files in tests/golden,
 * transcribed from the reference's #[cfg(test)] modules; see tests/test_oracle_golden.py).
 * UNPINNED by any reference test (followed from source only): multi-frame
 * concatenation in encode(), the stream walk, error paths.  The reference itself is
 * Rust and cannot be built in this image (no rustc/cargo), so there is no oracle/_ref.
 *
 * Every function cites the reference file:line (paths relative to /root/reference) it follows.
 */
#ifndef X3_ORACLE_H
#define X3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes: 0 = Ok, then the variant order of `enum X3Error` (src/error.rs:27-62),
 * then two codes for situations in which the reference panics instead of returning. */
enum {
  X3O_OK = 0,
  X3O_IO = 1, /* std::io::Error, e.g. read_exact past the end of the data */
  X3O_HOUND = 2,
  X3O_BITPACK = 3,
  X3O_INVALID_ENCODING_THRESH = 4,
  X3O_OUT_OF_BOUNDS_INVERSE = 5,
  X3O_MORE_THAN_ONE_CHANNEL = 6,
  X3O_ARCHIVE_HEADER_XML_INVALID = 7,
  X3O_ARCHIVE_HEADER_XML_RICE_CODE = 8,
  X3O_ARCHIVE_HEADER_XML_INVALID_KEY = 9,
  X3O_FRAME_LENGTH = 10,
  X3O_FRAME_HEADER_INVALID_KEY = 11,
  X3O_FRAME_HEADER_INVALID_PAYLOAD_LEN = 12,
  X3O_FRAME_HEADER_INVALID_HEADER_CRC = 13,
  X3O_FRAME_HEADER_INVALID_PAYLOAD_CRC = 14,
  X3O_FRAME_DECODE_INVALID_BLOCK_LENGTH = 15,
  X3O_FRAME_DECODE_INVALID_INDEX = 16,
  X3O_FRAME_DECODE_INVALID_NTOGO = 17,
  X3O_FRAME_DECODE_INVALID_FTYPE = 18,
  X3O_FRAME_DECODE_INVALID_RICE_CODE = 19,
  X3O_FRAME_DECODE_INVALID_BPF = 20,
  X3O_FRAME_DECODE_UNEXPECTED_END = 21,
  X3O_BYTE_WRITER_INSUFFICIENT_MEMORY = 22,
  X3O_HIP = 23,     /* unused by the oracle; keeps numbering equal to the product ABI */
  X3O_BAD_ARG = 24  /* the reference would panic here (index out of range, 0 samples, ...) */
};

/* x3::Parameters (src/x3.rs:81-134) without the derived rice_codes pointers. */
typedef struct {
  uint32_t block_len;
  uint32_t blocks_per_frame;
  uint32_t codes[3];
  uint32_t thresholds[3];
} x3o_params;

/* x3::RiceCode (src/x3.rs:187-194).  Tables are built by x3o_init() from the closed
 * form and compared against the reference's literal tables in tests/golden/rice_tables.json. */
typedef struct {
  uint32_t nsubs, offset, len, inv_len;
  uint32_t code[56];
  uint32_t num_bits[56];
} x3o_rice_code;

extern x3o_rice_code X3O_RICE[4];
extern int16_t X3O_INV_RICE[60];
extern uint16_t X3O_CRC_TABLE[256];

void x3o_init(void); /* idempotent; called lazily by every entry point */

/* ---- x3.rs ---- */
void x3o_params_default(x3o_params* p);     /* src/x3.rs:124-134 */
int x3o_params_new(const x3o_params* p);    /* src/x3.rs:98-122: validation only */

/* ---- crc.rs ---- */
uint16_t x3o_update_crc16(uint16_t crc, uint8_t data); /* src/crc.rs:44-47 */
uint16_t x3o_crc16(const uint8_t* data, size_t n);     /* src/crc.rs:49-58 */

/* ---- bytewriter.rs: SliceByteWriter (src/bytewriter.rs:27-100) ---- */
typedef struct {
  uint8_t* slice;
  size_t cap;
  size_t p_byte;
  size_t stream_length;
} x3o_writer;

void x3o_writer_init(x3o_writer* w, uint8_t* slice, size_t cap);
int x3o_writer_align(x3o_writer* w, size_t n);
int x3o_writer_write_all(x3o_writer* w, const uint8_t* v, size_t n);
int x3o_writer_seek_start(x3o_writer* w, size_t pos);
int x3o_writer_seek_current(x3o_writer* w, int64_t off);

/* ---- bitpacker.rs: BitPacker (src/bitpacker.rs:46-177) ---- */
typedef struct {
  x3o_writer* writer;
  uint8_t scratch_byte;
  size_t p_bit;
  size_t byte_len;
  uint16_t crc;
} x3o_bitpacker;

void x3o_bp_new(x3o_bitpacker* bp, x3o_writer* w);
int x3o_bp_write_bits(x3o_bitpacker* bp, uint64_t value, size_t num_bits);
int x3o_bp_write_packed_zeros(x3o_bitpacker* bp, size_t num_zeros);
int x3o_bp_word_align(x3o_bitpacker* bp);
int x3o_bp_write_bytes(x3o_bitpacker* bp, const uint8_t* array, size_t n);   /* bitpacker.rs:95-102 */
int x3o_bp_inc_counter_n_bytes(x3o_bitpacker* bp, size_t n_bytes);           /* bitpacker.rs:112-118 */
int x3o_bp_drop(x3o_bitpacker* bp); /* impl Drop: flush a partial byte */

/* ---- encoder.rs ---- */
void x3o_write_frame_header(size_t num_samples, uint8_t id, size_t payload_len, uint16_t payload_crc,
                            uint8_t out[20]);                                     /* :122-162 */
int x3o_encode_block(const int16_t* wav, size_t n, int16_t prev, x3o_bitpacker* bp,
                     const x3o_params* p, size_t* ftype_out);                     /* :289-315 */
int x3o_encode_frame(const int16_t* wav, size_t n, x3o_writer* w, const x3o_params* p,
                     uint64_t stats[6]);                                          /* :175-214 */
int x3o_encode(const int16_t* wav, uint64_t n, uint32_t n_channels, const x3o_params* p,
               uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos,
               uint64_t stats[6]);                                                /* :51-111 */

/* ---- bitreader.rs: BitReader (src/bitreader.rs:29-176) ---- */
typedef struct {
  const uint8_t* array;
  size_t len;
  size_t idx;
  uint32_t leading_word;
  size_t rem_bit;
} x3o_bitreader;

void x3o_br_new(x3o_bitreader* br, const uint8_t* array, size_t len);
uint32_t x3o_br_read_nbits(x3o_bitreader* br, size_t n);
size_t x3o_br_count_zero_bits(x3o_bitreader* br);

/* ---- decoder.rs ---- */
typedef struct {
  uint8_t source_id;
  uint16_t samples;
  uint8_t channels;
  uint32_t payload_len;
  uint16_t payload_crc;
} x3o_frame_header;

int x3o_read_frame_header(const uint8_t* bytes, size_t len, x3o_frame_header* h);   /* :69-118 */
int x3o_decode_block(x3o_bitreader* br, int16_t* wav, size_t n, int16_t* last_wav,
                     const x3o_params* p);                                          /* :132-145 */
int x3o_decode_frame(const uint8_t* x3_bytes, size_t len, int16_t* wav_buf, size_t wav_cap,
                     const x3o_params* p, size_t samples, size_t* n_out);           /* :36-58 */

/* ---- decodefile.rs: the frame walk of X3aReader::decode_next_frame (:93-136) over an
 * in-memory frame stream (no archive header).  Returns the hard error that ends the walk
 * (0 if it ended at end-of-data or on a counted decode error). */
int x3o_decode_stream(const uint8_t* x3, uint64_t len, const x3o_params* p, int16_t* wav,
                      uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                      uint64_t* frame_errors);

/* ---- multi-channel extension (not in the reference: see x3_oracle.c) ---- */
int x3o_encode_frame_mc(const int16_t* const* wavs, uint32_t n_ch, size_t n, x3o_writer* w, const x3o_params* p,
                        uint64_t stats[6]);
int x3o_encode_mc(const int16_t* const* wavs, uint32_t n_ch, uint64_t n, const x3o_params* p, uint8_t* out,
                  uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]);
int x3o_decode_frame_mc(const uint8_t* x3_bytes, size_t len, int16_t* const* wavs, size_t wav_cap, uint32_t n_ch,
                        const x3o_params* p, size_t samples, size_t* n_out);
int x3o_decode_stream_mc(const uint8_t* x3, uint64_t len, uint32_t n_ch, const x3o_params* p, int16_t* const* wavs,
                         uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors);


/* ---- encodefile.rs / decodefile.rs: the .x3a archive around the frame stream (no file I/O).
 * Unpinned by any reference test (its file tests are commented out): follows the source.
 * quick-xml (Cargo.toml: 0.38, not in the reference tree) is restated as "text of the first
 * <NAME ...>...</NAME> element, trimmed", which is what Event::Start + read_text yield on the
 * well-formed XML the writer produces. */
int x3o_archive_header_write(uint32_t sample_rate, const x3o_params* p, uint8_t* out, uint64_t cap,
                             uint64_t* out_len);                                   /* encodefile.rs:82-138 */
int x3o_archive_header_read(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, x3o_params* p,
                            uint8_t* channels, uint64_t* header_size);             /* decodefile.rs:142-176,232-303 */
int x3o_x3a_encode(const int16_t* wav, uint64_t n, uint32_t sample_rate, uint8_t* out, uint64_t cap,
                   uint64_t* out_len, uint64_t stats[6]);                          /* encodefile.rs:48-77 */
int x3o_x3a_decode(const uint8_t* x3a, uint64_t len, int16_t* wav, uint64_t wav_cap, uint64_t* n_out,
                   uint32_t* sample_rate, uint64_t* frames_ok, uint64_t* frame_errors); /* decodefile.rs:59-136,189-212 */

/* ---- the file level: encodefile::wav_to_x3a (encodefile.rs:48-77), decodefile::x3a_to_wav
 * (decodefile.rs:189-227).  WAV files go through `hound` 3.4.0 in the reference (Cargo.toml:24; not in
 * the reference tree, no Cargo.lock).  Restated here from its documented behaviour for the only case the
 * reference accepts, 16-bit integer PCM, one channel: the reader walks the RIFF chunks up to "data"
 * (fmt tag 1, or 0xFFFE with the PCM sub-format) and yields data_len/2 little-endian samples; the writer
 * emits the canonical 44-byte header (RIFF size, "WAVE", a 16-byte PCMWAVEFORMAT "fmt " chunk, "data")
 * followed by the samples, and fixes both sizes when it is dropped -- also on the error path, so a walk
 * that ends with a hard error leaves a valid WAV of the samples decoded before it.  PARITY UNPINNED: no
 * reference test touches files (they are commented out, encodefile.rs:140-148, decodefile.rs:317-325).
 * Where the reference panics (unwrap / assert_eq: input missing, not 16 bit, not mono, malformed WAV)
 * these return X3O_IO for a file that cannot be opened or read and X3O_BAD_ARG for a format it rejects. */
int x3o_wav_parse(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, uint16_t* channels,
                  uint16_t* bits, uint64_t* data_off, uint64_t* data_len);
void x3o_wav_header_write(uint32_t sample_rate, uint64_t n_samples, uint8_t out[44]);
int x3o_wav_to_x3a(const char* wav_path, const char* x3a_path, uint64_t stats[6]);
int x3o_x3a_to_wav(const char* x3a_path, const char* wav_path, uint64_t* n_samples, uint64_t* frame_errors);

/* Timing helper for bench.py's cpu_baseline leg: encode then decode `n` samples `reps` times,
 * single thread; returns seconds for encode and decode separately. */
int x3o_time_roundtrip(const int16_t* wav, uint64_t n, const x3o_params* p, int reps,
                       double* enc_s, double* dec_s, uint64_t* stream_len);

#ifdef __cplusplus
}
#endif
#endif
