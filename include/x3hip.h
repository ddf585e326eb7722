/*
 * x3hip.h -- C ABI of the MI355X-native X3 encoder/decoder (libx3hip.so).
 *
 * This is the drop-in boundary for the encode/decode hot path of the psiphi75/x3-rust crate
 * (SURVEY.md section 8b).  The reference has no FFI layer of its own: the boundary is its public Rust
 * API, so every entry point below names the Rust item it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the `extern "C"` block + safe wrappers a crate
 * maintainer would add; x3-rust_amd/host/x3.hpp is the same surface for C++ callers and
 * x3-rust_amd/x3hip/ the ctypes binding used by tests and bench.py.
 *
 * Rules of the ABI
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns an int status: 0 = OK, otherwise the 1-based variant index of the
 *     reference's `enum X3Error` (src/error.rs:27-62), or X3_ERR_HIP / X3_ERR_BAD_ARG.
 *     X3_ERR_BAD_ARG is returned where the reference would panic; the library never aborts.
 *   - all bulk work (prediction filter, Rice/BFP/literal coding, bit packing, payload CRC,
 *     stream compaction, decoding) runs in HIP kernels on the context's GPU.  There is no CPU
 *     fallback: without a usable HIP device x3_ctx_create fails with X3_ERR_HIP.
 *   - a context is single-threaded; use one per GPU / per host thread.
 *   - `*_dev` entry points take DEVICE pointers, enqueue on the context's stream and do not
 *     synchronise; x3_ctx_sync() / x3_*_result() wait.
 */
#ifndef X3HIP_H
#define X3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
/* order = `enum X3Error`, src/error.rs:27-62 */
enum x3_status {
  X3_OK = 0,
  X3_ERR_IO = 1,
  X3_ERR_HOUND = 2,
  X3_ERR_BITPACK = 3,
  X3_ERR_INVALID_ENCODING_THRESH = 4,
  X3_ERR_OUT_OF_BOUNDS_INVERSE = 5,
  X3_ERR_MORE_THAN_ONE_CHANNEL = 6,
  X3_ERR_ARCHIVE_HEADER_XML_INVALID = 7,
  X3_ERR_ARCHIVE_HEADER_XML_RICE_CODE = 8,
  X3_ERR_ARCHIVE_HEADER_XML_INVALID_KEY = 9,
  X3_ERR_FRAME_LENGTH = 10,
  X3_ERR_FRAME_HEADER_INVALID_KEY = 11,
  X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN = 12,
  X3_ERR_FRAME_HEADER_INVALID_HEADER_CRC = 13,
  X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC = 14,
  X3_ERR_FRAME_DECODE_INVALID_BLOCK_LENGTH = 15,
  X3_ERR_FRAME_DECODE_INVALID_INDEX = 16,
  X3_ERR_FRAME_DECODE_INVALID_NTOGO = 17,
  X3_ERR_FRAME_DECODE_INVALID_FTYPE = 18,
  X3_ERR_FRAME_DECODE_INVALID_RICE_CODE = 19,
  X3_ERR_FRAME_DECODE_INVALID_BPF = 20,
  X3_ERR_FRAME_DECODE_UNEXPECTED_END = 21,
  X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY = 22,
  X3_ERR_HIP = 23,    /* a HIP call failed; see x3_last_error() */
  X3_ERR_BAD_ARG = 24 /* the reference would panic (index out of range, 0 samples, ...) or the
                         argument is outside what the GPU path supports */
};

const char* x3_strerror(int status);

/* ------------------------------------------------------------------ types */

/* x3::Parameters (src/x3.rs:81-134); rice_codes[] is derived from codes[]. */
typedef struct x3_params {
  uint32_t block_len;        /* DEFAULT_BLOCK_LENGTH 20, MAX_BLOCK_LENGTH 60 */
  uint32_t blocks_per_frame; /* DEFAULT_BLOCKS_PER_FRAME 500 */
  uint32_t codes[3];         /* DEFAULT_RICE_CODES {0,1,3} */
  uint32_t thresholds[3];    /* DEFAULT_THRESHOLDS {3,8,20} */
} x3_params;

/* x3::FrameHeader (src/x3.rs:148-184) */
typedef struct x3_frame_header {
  uint8_t source_id;
  uint8_t channels;
  uint16_t samples;
  uint32_t payload_len;
  uint16_t payload_crc;
} x3_frame_header;

#define X3_FRAME_HEADER_LENGTH 20   /* FrameHeader::LENGTH */
#define X3_FRAME_KEY 0x7833         /* FrameHeader::KEY "x3" */
#define X3_FRAME_MAX_LENGTH 0x7fe0  /* Frame::MAX_LENGTH */
#define X3_READ_BUFFER_SIZE 24576   /* decodefile.rs:44 */

typedef struct x3_ctx x3_ctx;

/* ------------------------------------------------------------------ context */

/* Create a context on HIP device `device` with its own non-blocking stream. */
int x3_ctx_create(int device, x3_ctx** ctx);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int x3_ctx_create_on_stream(int device, void* hip_stream, x3_ctx** ctx);
void x3_ctx_destroy(x3_ctx* ctx);
int x3_ctx_sync(x3_ctx* ctx);
/* Text of the last HIP failure seen by this context ("" if none). */
const char* x3_last_error(const x3_ctx* ctx);

/* Tuning / testing knobs of a context.  Each has an X3HIP_* environment variable that gives its initial value;
 * the environment is read ONCE, in x3_ctx_create -- no later call looks at it.
 *   "two_pass" (X3HIP_TWO_PASS)            1: always encode with the two-pass kernels (no persistent grid)
 *   "stream_wgs" (X3HIP_STREAM_WGS)        workgroups per CU of the single-pass encoder, 0 = derived from occupancy
 *   "decode_single" (X3HIP_DECODE_SINGLE)  1: single-wave decoder kernels only
 *   "decode_blocks" (X3HIP_DECODE_BLOCKS)  1: block length 20 on round 6's block-per-lane decoder (a walker wave finds where the
 *                                          blocks begin, three decoder waves decode a block per lane) where the three-wave
 *                                          kernel would run; same results, 0.78-0.82 against 0.65-0.69 ms on config 3.  Block
 *                                          lengths 10 and 40 take that decoder by default ("decode_blocks_off" = 1: not).
 *                                          Read-only "decode_kernel_in_use": 3 = block per lane, 2 = three waves per 64
 *                                          frames, 1 / 0 = single wave
 *   (environment only) X3HIP_SPIN_WAIT=1   the process's waits for the GPU spin instead of sleeping (hipDeviceScheduleSpin,
 *                                          process-wide, effective when x3_ctx_create is the process's first use of the
 *                                          device): calls that end with a trip to the host come back ~20 us sooner
 *                                          (x3_decode_stream_dev on config 3: 0.82 -> 0.79 ms), a core is busy meanwhile
 *   (environment only) X3HIP_FENCE=16      debugging: every device buffer of the library (its own and x3_dev_alloc's) is
 *                                          mapped with unmapped pages on both sides and ends, rounded up to that many bytes,
 *                                          at the end of its mapping; X3HIP_FENCE_FILL=<byte> fills it.  The kernels read
 *                                          aligned 16-byte chunks, so 16 is the tightest fence (csrc/x3_fence.h)
 *   "wav_offsets_x4"                       1: a promise -- every d_wav_offsets[] passed to x3_decode_dev is a multiple of
 *                                          four samples (rows on 8-byte boundaries): such calls then take the three-wave
 *                                          decoder like the other layouts do; an offset that breaks the promise garbles
 *                                          its frame's samples AND may clobber up to three samples in front of the frame's
 *                                          range (the rows leave in whole 8-byte pieces from the piece their first sample
 *                                          lies in): the promise is the caller's to keep
 *   "host_walk" (X3HIP_HOST_WALK)          frame walk of x3_decode_stream: 1 host, 0 GPU, -1 by stream size
 *   "host_chunk_frames" (X3HIP_HOST_CHUNK_FRAMES)  x3_encode and x3_decode_stream take a long host buffer in chunks of whole
 *                                           frames, downloads beside uploads: 0 = on (x3_encode from 32 Mi samples in chunks of
 *                                           16 Mi; x3_decode_stream from 16 MiB of stream in chunks that grow from 16 Mi to
 *                                           128 Mi samples), N > 0 = chunks of N frames whatever the length (tests), -1 = one
 *                                           piece.  A call that goes in chunks starts two helper threads for its duration
 *                                           (pageable copies hold their caller); the results are the same either way.
 *   "file_chunk_frames" (X3HIP_FILE_CHUNK_FRAMES), "file_workers" (X3HIP_FILE_WORKERS)   x3_wav_to_x3a / x3_x3a_to_wav
 *   "reader_window_frames" (X3HIP_READER_WINDOW_FRAMES)   frames x3_reader decodes ahead per launch set
 *   "check_main" (X3HIP_CHECK_MAIN)        1: the check pass on the context's stream, the decoder on the side stream (experiment)
 *   "check_wgs", "check_prio", "check_first"   grid, queue priority and launch order of the check pass (experiments)
 *   "verbose" (X3HIP_VERBOSE)
 * x3_ctx_get_option also reads "encode_fallbacks" (launches of the single-pass encoder that timed out waiting for
 * a non-resident workgroup and were redone by the two-pass kernels), "stream_wgs_in_use", and "encode_pace" /
 * "decode_pace": what the slowest workgroup of the last encoder / decoder launch achieved, in 10 ns ticks per frame /
 * in shader clocks per 16 blocks -- the next launch paces its waves' priorities by it (x3_encode_stream2_kernel.h,
 * x3_decode_split_kernel.h); reading them synchronizes.
 * Unknown name: X3_ERR_BAD_ARG. */
int x3_ctx_set_option(x3_ctx* ctx, const char* name, long long value);
int x3_ctx_get_option(const x3_ctx* ctx, const char* name, long long* value);

/* HIP-event timing of individual kernels on the context's stream (bench.py's roofline leg).
 * which: 0 = encode kernel, 1 = decode kernel, 2 = frame-size kernel, 3 = scan kernel,
 *        4 = frame check (header + payload CRC) kernel, 5 = the encoder's dense pass (frames the wave encoder's LDS image
 *        does not hold: x3_encode_dev below).  Every timed kernel costs a marker behind its dispatch packet (~5 us each,
 *        five a round trip); option "kernel_timing_mask" (bit k = kernel id k, default all) chooses which ones carry them. */
int x3_ctx_enable_kernel_timing(x3_ctx* ctx, int enable);
int x3_ctx_kernel_time(x3_ctx* ctx, int which, double* total_ms, uint64_t* launches); /* syncs */
int x3_ctx_reset_kernel_time(x3_ctx* ctx);
/* Every timed launch's own time in ms, oldest first (which as above; 5 = the encoder's dense pass): *launches = how many
 * there are, the first min(cap, *launches) are written.  Syncs. */
int x3_ctx_kernel_times(x3_ctx* ctx, int which, double* ms, uint64_t cap, uint64_t* launches);
/* The launch log the kernels keep themselves: the last (up to 256) launches of the decoder (which = 1) or the wave
 * encoder (which = 0), oldest first, four words each: the pace the launch aimed at and the pace its slowest group
 * achieved (10 ns ticks per 16 blocks; decoder only, else 0), the shader clock in kHz that workgroup 0 measured over
 * its life (shader ticks against the constant 100 MHz clock), and that life in 10 ns ticks.  Syncs.  For benchmarks:
 * a reader of the line can tell a slow box (clock) from a controller that has not settled (target vs achieved). */
int x3_ctx_launch_log(x3_ctx* ctx, int which, uint32_t* out /* 4 * cap_entries */, uint64_t cap_entries, uint64_t* n_entries);

/* ------------------------------------------------------------------ x3.rs */

/* `impl Default for Parameters`, src/x3.rs:124-134 */
void x3_params_default(x3_params* p);
/* `Parameters::new`, src/x3.rs:98-122: INVALID_ENCODING_THRESH if thresholds[k] > offset of
 * code k for k = 0,1 (the reference checks only those two); BAD_ARG for a code > 3. */
int x3_params_validate(const x3_params* p);
/* `RiceCode` / `RiceCodes::get`, src/x3.rs:187-260: the code table Parameters.rice_codes[k] points at.  `code` and
 * `num_bits` have `len` entries indexed by (difference + offset); `inv` has 60 entries of which inv_len are in use.  The
 * pointers are to static read-only tables of the library (host memory; the kernels derive the same values arithmetically).
 * BAD_ARG for a code number > 3 (the reference panics on the index). */
typedef struct x3_rice_code {
  uint32_t nsubs, offset, len, inv_len;
  const uint32_t* code;
  const uint32_t* num_bits;
  const int16_t* inv;
} x3_rice_code;
int x3_rice_code_get(uint32_t code_number, x3_rice_code* out);
/* number of frames `encode` cuts n samples into (encoder.rs:61-73) */
uint64_t x3_num_frames(uint64_t n, const x3_params* p);
/* worst-case bytes `encode` can produce for n samples (all-literal frames) */
uint64_t x3_encode_bound(uint64_t n, const x3_params* p);

/* ------------------------------------------------------------------ crc.rs */

/* `crc::crc16`, src/crc.rs:49-58 -- computed on the GPU as a segmented reduction. */
int x3_crc16(x3_ctx* ctx, const uint8_t* data, uint64_t n, uint16_t* crc);
int x3_crc16_dev(x3_ctx* ctx, const uint8_t* d_data, uint64_t n, uint16_t* crc); /* syncs */
/* `crc::update_crc16`, src/crc.rs:44-47 -- one byte, host arithmetic. */
uint16_t x3_crc16_update(uint16_t crc, uint8_t byte);

/* ------------------------------------------------------------------ encoder.rs */

/* `encoder::encode` into a `SliceByteWriter` (src/encoder.rs:51-111, bytewriter.rs:27-100).
 * wav[0..n) is the single channel's samples (`x3::Channel::wav` / a collected `IterChannel`);
 * n_channels is `channels.len()` (>1 -> MORE_THAN_ONE_CHANNEL, 0 -> BAD_ARG).  The writer is
 * out[0..out_cap) positioned at start_pos; frames are appended back to back, each preceded by
 * zero padding to an even ABSOLUTE position.  *out_pos receives `writer.stream_position()`.
 * stats[6] (may be NULL) receives the per-sample block-type counts the reference prints
 * (Rice nsubs 0..3, BFP = 4, literal = 5; encoder.rs:96-108,199).
 * On BYTE_WRITER_INSUFFICIENT_MEMORY the slice holds what the reference's holds (bytewriter.rs:86-99,
 * encoder.rs:67-73): every frame that fits, complete and in place (and the pad byte in front of
 * the first one); *out_pos = the end of the last of them, nothing behind it is touched (the
 * reference goes on into the frame that does not fit and leaves some of its payload bytes without
 * a header there).  x3_ctx_get_option("encode_needed_pos") says where the whole stream would have
 * ended. */
int x3_encode(x3_ctx* ctx, const int16_t* wav, uint64_t n, uint32_t n_channels, const x3_params* p,
              uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]);

/* `encoder::encode_frame` (src/encoder.rs:175-214): exactly one frame from wav[0..n), n >= 1. */
int x3_encode_frame(x3_ctx* ctx, const int16_t* wav, uint64_t n, const x3_params* p, uint8_t* out,
                    uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]);

/* `encoder::write_frame_header` (src/encoder.rs:122-162); 20 bytes of host arithmetic. */
void x3_write_frame_header(uint64_t num_samples, uint8_t id, uint64_t payload_len, uint16_t payload_crc,
                           uint8_t out[X3_FRAME_HEADER_LENGTH]);

/* Batched `encode`: `count` independent clips (BASELINE config 5).  Clip c is wavs[c][0..ns[c]).
 * The clips' streams are written back to back into out[0..out_cap); clip_offsets[count+1]
 * receives where each starts/ends (all even).  Equivalent to `count` calls of x3_encode. */
int x3_encode_batch(x3_ctx* ctx, const int16_t* const* wavs, const uint64_t* ns, uint64_t count,
                    const x3_params* p, uint8_t* out, uint64_t out_cap, uint64_t* clip_offsets,
                    uint64_t stats[6]);

/* ------------------------------------------------------------------ decoder.rs / decodefile.rs */

/* `decoder::read_frame_header` (src/decoder.rs:69-118); 20 bytes of host arithmetic.
 * Check order: length, header CRC, key, channels <= 1, payload_len < Frame::MAX_LENGTH. */
int x3_read_frame_header(const uint8_t* bytes, uint64_t len, x3_frame_header* h);

/* `decoder::decode_frame` (src/decoder.rs:36-58): payload[0..len) -> wav[0..samples).
 * Does not check any CRC (the reference does not either).  wav_cap = `wav_buf.len()`: a buffer shorter than `samples`
 * is the reference's slice-index panic (X3_ERR_BAD_ARG) at the first block that does not fit -- unless a block in
 * front of it fails, whose error comes first, as in the reference (decoder.rs:49).  Payloads and sample counts beyond
 * what a frame header or the walk's read buffer allow (>= 32 736 / > 24 576 bytes, > 65 535 samples) are decoded too,
 * by the reference-exact reader on one GPU thread: correct, slow. */
int x3_decode_frame(x3_ctx* ctx, const uint8_t* payload, uint64_t len, int16_t* wav, uint64_t wav_cap,
                    const x3_params* p, uint64_t samples, uint64_t* n_out);

/* The frame walk of `X3aReader::decode_next_frame` looped as `x3a_to_wav` does
 * (src/decodefile.rs:93-136,200-209) over an in-memory frame stream x3[0..len) (no archive
 * header): stop at end of data, at the first hard error (returned), or at the first frame whose
 * payload fails to decode (counted in *frame_errors, return 0).  wav receives the samples of
 * the frames before the stop; *n_out their count; *frames_ok the number of good frames.
 * (Streams of 4 MiB and more are uploaded first and walked on the GPU, x3_index_dev's way; shorter ones are
 * walked on the host.  Same results either way.)  wav_cap is this API's, not the reference's (x3a_to_wav writes to a
 * file): a frame that does not fit behind the samples so far ends the walk with X3_ERR_BAD_ARG once its CRCs have
 * passed -- also when an early block of that very frame would not have decoded. */
int x3_decode_stream(x3_ctx* ctx, const uint8_t* x3, uint64_t len, const x3_params* p, int16_t* wav,
                     uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors);

/* ------------------------------------------------------------------ multi-channel (extension) */

/* NOT in the reference: `encoder::encode` returns MoreThanOneChannel for more than one channel (src/encoder.rs:55-57) and
 * `read_frame_header` refuses frames that announce more than one (src/decoder.rs:90-94) -- and so do x3_encode /
 * x3_decode_stream and every other entry point above.  These two follow what the format foresees: the frame header's
 * <Num Channels> (src/x3.rs:155-156, src/encoder.rs:134) and "pack the data block for each channel" (src/encoder.rs:197):
 * <Audio State> = the first sample of each channel; then for every block index the block of channel 0 .. n-1, each coded
 * as a mono block against its own channel; header byte 3 = n_channels, `samples` = samples per channel.  With
 * n_channels = 1 the bytes are x3_encode's.  wavs[k] = channel k, n samples each (host memory).  n_channels <= 8; a frame
 * whose payload would pass the 24 KB a reader takes is X3_ERR_FRAME_LENGTH (choose shorter frames).
 * x3_decode_stream_mc walks, checks and decodes such a stream (x3_decode_stream's rules; a frame must announce exactly
 * n_channels) into wavs[k][0 .. *n_samples). */
int x3_encode_mc(x3_ctx* ctx, const int16_t* const* wavs, uint32_t n_channels, uint64_t n, const x3_params* p, uint8_t* out,
                 uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]);
int x3_decode_stream_mc(x3_ctx* ctx, const uint8_t* x3, uint64_t len, uint32_t n_channels, const x3_params* p,
                        int16_t* const* wavs, uint64_t wav_cap, uint64_t* n_samples, uint64_t* frames_ok,
                        uint64_t* frame_errors);

/* ------------------------------------------------------------------ bitreader.rs / bitpacker.rs / decode_block */

/* The reference's small public items, for callers and known-answer tests written against them (x3_bits.h: not how the
 * bulk paths work, and a call per field is not how a GPU should be driven -- but what they compute is computed by
 * this library on the GPU, not by a CPU re-implementation).
 * `BitReader` (src/bitreader.rs:51-176): state and a copy of the array live in device memory; each call runs the
 * reference-exact reader (one thread) and brings the result back.  x3_bitreader_state: the reference's private
 * fields (idx, leading_word, rem_bit), for tests that assert on them. */
typedef struct x3_bitreader x3_bitreader;
int x3_bitreader_new(x3_ctx* ctx, const uint8_t* array, uint64_t len, x3_bitreader** br);
int x3_bitreader_read_nbits(x3_bitreader* br, uint32_t n, uint32_t* value);
int x3_bitreader_count_zero_bits(x3_bitreader* br, uint32_t* count);
int x3_bitreader_inc_bits(x3_bitreader* br, uint32_t n);
int x3_bitreader_state(const x3_bitreader* br, uint64_t* idx, uint32_t* leading_word, uint32_t* rem_bit);
void x3_bitreader_free(x3_bitreader* br);
/* Not in the reference: announce a frame stream in host memory (frames back to back, no archive header) whose frames
 * will be handed to x3_decode_frame one by one, as a loop over `decode_frame` does.  Calls whose payload lies in the
 * announced buffer are then served from windows of frames that are walked, checked and decoded ahead in one launch
 * set (a header parse and a memcpy per call instead of a dispatch per call); every other call, and every frame the
 * window did not decode cleanly, takes the per-call path: the results are those of x3_decode_frame without the
 * announcement.  The buffer must stay unchanged until the next x3_decode_prefetch (x3 = NULL drops it) or
 * x3_ctx_destroy. */
int x3_decode_prefetch(x3_ctx* ctx, const uint8_t* x3, uint64_t len, const x3_params* p);

/* `decoder::decode_block` (src/decoder.rs:132-145): wav[0..n) from the reader's position, *last_wav in and out.
 * n <= 60 (MAX_BLOCK_LENGTH); n == 0 is the reference's empty slice (type bits read; a BFP block then fails or panics). */
int x3_decode_block(x3_bitreader* br, int16_t* wav, uint32_t n, int16_t* last_wav, const x3_params* p);
/* `BitPacker` (src/bitpacker.rs:46-177) over a slice writer at start_pos: write_bits / write_packed_zeros / word_align
 * are recorded; x3_bitpacker_finish (= flush, what Drop does) zero-pads a partial byte, packs all recorded fields on the
 * GPU (scan of the field widths, fields OR-ed into place), writes the bytes not handed over yet behind start_pos and
 * returns the reference's len() and crc() at that point (cumulative since new(); CRC-16 init 0xFFFF) and the writer's
 * position; writing may go on from the next byte.  x3_bitpacker_peek = len() / crc() between writes: complete bytes
 * so far and their CRC, nothing written.  A packer over any other ByteWriter is made with out = NULL, out_cap = 0
 * (start_pos = the writer's position: word_align needs its parity) and flushed with x3_bitpacker_take, which delivers
 * the bytes not handed over yet into dst[0, *n_new) for the caller to pass on. */
typedef struct x3_bitpacker x3_bitpacker;
int x3_bitpacker_new(x3_ctx* ctx, uint8_t* out, uint64_t out_cap, uint64_t start_pos, x3_bitpacker** bp);
int x3_bitpacker_write_bits(x3_bitpacker* bp, uint64_t value, uint32_t num_bits);
int x3_bitpacker_write_packed_zeros(x3_bitpacker* bp, uint32_t num_zeros);
/* BitPacker::write_bytes (bitpacker.rs:95-102): the bytes go to the writer at once -- in front of a partial byte the packer
 * still holds -- and count in len() and crc(); BitPacker::inc_counter_n_bytes (:112-118): the writer skips n bytes, len()
 * and crc() stay (X3_ERR_BITPACK = BitPackError::NotByteAligned off a byte boundary; slice-bound packers only). */
int x3_bitpacker_write_bytes(x3_bitpacker* bp, const uint8_t* array, uint64_t n);
int x3_bitpacker_inc_counter_n_bytes(x3_bitpacker* bp, uint64_t n_bytes);
int x3_bitpacker_word_align(x3_bitpacker* bp);
int x3_bitpacker_finish(x3_bitpacker* bp, uint64_t* len, uint16_t* crc, uint64_t* out_pos);
int x3_bitpacker_peek(const x3_bitpacker* bp, uint64_t* len, uint16_t* crc);
int x3_bitpacker_take(x3_bitpacker* bp, uint8_t* dst, uint64_t dst_cap, uint64_t* n_new, uint64_t* len, uint16_t* crc);
void x3_bitpacker_free(x3_bitpacker* bp);

/* ------------------------------------------------------------------ .x3a archive (encodefile.rs / decodefile.rs) */

/* `create_archive_header` (src/encodefile.rs:82-138): "X3ARCHIV" + a frame header with id 0 / 0 samples
 * + the XML configuration (zero-padded to even length).  Host arithmetic.  *out_len = bytes written. */
int x3_archive_header_write(uint32_t sample_rate, const x3_params* p, uint8_t* out, uint64_t out_cap,
                            uint64_t* out_len);
/* `read_archive_header` + `parse_xml` (src/decodefile.rs:142-176,232-303).  *header_size is what the
 * reference returns (20 + XML payload, NOT counting the 8-byte id); the audio frames start at
 * 8 + *header_size.  blocks_per_frame is the default (the XML does not carry it). */
int x3_archive_header_read(const uint8_t* bytes, uint64_t len, uint32_t* sample_rate, x3_params* p,
                           uint8_t* channels, uint64_t* header_size);
/* `wav_to_x3a` without the file I/O (src/encodefile.rs:48-77): archive header + `encode` of one 16-bit
 * mono channel with default parameters, into out[0..out_cap). */
int x3_x3a_encode(x3_ctx* ctx, const int16_t* wav, uint64_t n, uint32_t sample_rate, uint8_t* out,
                  uint64_t out_cap, uint64_t* out_len, uint64_t stats[6]);
/* `x3a_to_wav` without the file I/O (src/decodefile.rs:189-212 with X3aReader, :59-136): archive header,
 * then the frame walk with the reader's own byte accounting. */
int x3_x3a_decode(x3_ctx* ctx, const uint8_t* x3a, uint64_t len, int16_t* wav, uint64_t wav_cap,
                  uint64_t* n_out, uint32_t* sample_rate, uint64_t* frames_ok, uint64_t* frame_errors);

/* The file level: `encodefile::wav_to_x3a` (src/encodefile.rs:48-77) and `decodefile::x3a_to_wav`
 * (src/decodefile.rs:189-227) -- what the reference's CLI calls (src/bin/x3.rs:79-80).  Streaming: the file
 * moves through the GPU in chunks of whole frames (pread -> pinned staging -> H2D -> kernels -> D2H ->
 * pwrite, neighbouring chunks overlapped on a few worker contexts), so memory is bounded whatever the
 * file size; the bytes written are those of x3_x3a_encode / x3_x3a_decode on the whole file.
 * WAV container as `hound` 3.4.0 handles it for the only format the reference accepts: 16-bit integer PCM,
 * one channel (anything else: X3_ERR_BAD_ARG, where the reference asserts); output is hound's canonical
 * 44-byte header + samples.  A file that cannot be opened or read: X3_ERR_IO (the reference unwraps);
 * an output .wav that cannot be created: X3_ERR_HOUND (`WavWriter::create(..)?`).
 * x3_x3a_to_wav leaves, like the reference's dropped WavWriter, a valid WAV of the samples in front of the
 * frame that ended the walk, also when it returns an error.
 * Tuning (x3_ctx_set_option): "file_chunk_frames" (default 800 frames = 16 MB of samples), "file_workers"
 * (default 4; the workers' contexts share the single-pass encoder through a gate -- one chunk's encode launch at a time,
 * uploads, downloads and file I/O side by side). */
int x3_wav_to_x3a(x3_ctx* ctx, const char* wav_path, const char* x3a_path, uint64_t stats[6]);
int x3_x3a_to_wav(x3_ctx* ctx, const char* x3a_path, const char* wav_path, uint64_t* n_samples,
                  uint64_t* frame_errors);

/* `X3aReader` (src/decodefile.rs:47-137): open / spec / decode_next_frame -- one frame per call, for callers written
 * against the reference's incremental reader.  A call that finds nothing prepared decodes a window of frames ahead in
 * one launch set (option "reader_window_frames", default 4096) and the following calls are a header parse and a
 * memcpy.  x3_reader_next_frame = `decode_next_frame`: *n_out = the frame's sample count (Ok(Some(n))), or 0 with
 * X3_OK for Ok(None) -- end of the data, a payload that runs past it, or a frame that fails to decode (counted:
 * x3_reader_frame_errors) -- or an error status; as in the reference the reader has then consumed the frame and a
 * further call goes on behind it.  wav needs room for the frame (<= 65535 samples).  x3_reader_open_mem reads an
 * archive that is in memory (borrowed: it must outlive the reader). */
typedef struct x3_reader x3_reader;
int x3_reader_open(x3_ctx* ctx, const char* x3a_path, x3_reader** reader);
int x3_reader_open_mem(x3_ctx* ctx, const uint8_t* x3a, uint64_t len, x3_reader** reader);
int x3_reader_spec(const x3_reader* reader, uint32_t* sample_rate, x3_params* p, uint8_t* channels);
int x3_reader_next_frame(x3_reader* reader, int16_t* wav, uint64_t wav_cap, uint64_t* n_out);
uint64_t x3_reader_frame_errors(const x3_reader* reader);
uint64_t x3_reader_position(const x3_reader* reader); /* byte offset in the archive of the next frame header */
void x3_reader_close(x3_reader* reader);

/* ------------------------------------------------------------------ device-resident API */

/* Geometry of a uniform batch resident in HBM: n_clips clips of n_per_clip samples, clip c
 * starting at d_wav + c*clip_stride (samples).  A single stream is n_clips = 1. */
typedef struct x3_batch {
  uint64_t n_per_clip;
  uint64_t clip_stride;
  uint64_t n_clips;
} x3_batch;

/* Encode a device-resident batch into d_out[0..out_cap) starting at start_pos (even or odd; an
 * odd start is zero-padded to even as the reference does).  d_frame_offsets (may be NULL)
 * receives F+1 byte offsets: frame f occupies [d_frame_offsets[f], d_frame_offsets[f+1]).
 * Asynchronous; results via x3_encode_result().
 *   Content: frames whose payload does not fit the wave encoder's LDS image (more than 9 728 bytes: loud or noisy
 * material) are written by a dense pass that follows the encode kernel IN THE SAME STREAM, at the offsets that kernel
 * assigned -- whatever is enqueued on the context's stream behind this call (x3_decode_dev, a copy) finds the whole
 * stream, with or without x3_encode_result() in between; no call is encoded twice for its content (until round 3 the
 * whole call was encoded again inside x3_encode_result).  Options "last_dense_frames", "encode_dense_frames" count such
 * frames, "enc_gen_in_use" says which kernel served the last call -- 3 the wave encoder, 2 the second generation, 1 the
 * general kernel in one pass (any block length: sizes by decoupled look-back), 0 the same kernel in two passes (option
 * "two_pass", or what a launch falls back to whose waits gave up) -- (a call with more than a quarter of dense
 * frames makes the context's next call start on the second-generation kernel: a speed hint, the bytes are the same).
 *   Layout: the results never depend on it, the kernels that serve a call do.  block_len 20, 10 or 40 with frames of at most
 * 10 240 samples (512 blocks of 20), d_wav on a dword boundary and (for n_clips > 1) a clip_stride that is a multiple of four samples take the single-pass
 * encoders; x3_decode_dev takes the three-wave decoder when its output begins on an 8-byte boundary and the stride is a
 * multiple of four samples (rows on 16-byte boundaries leave in 16-byte pieces, on 8-byte ones in 8-byte pieces, always
 * as whole 128-byte lines).  Anything else -- other block lengths or code sets, longer frames, odd strides -- is served by
 * the general kernels, three to eight times slower ("GPU-path limits").
 *   Residency: the default-geometry encoder is a persistent grid whose workgroups wait for each other's frame
 * sizes; on a GPU that this context does not have to itself a launch can find them not all resident, gives up after a
 * bounded wait (15-30 ms: a stall of that length per call on a GPU shared with another process's long kernels), and
 * x3_encode_result() then re-encodes with the general kernels (option "encode_fallbacks" counts
 * these).  That re-run reads d_wav again and rewrites d_out[start_pos ..): d_wav and d_out must stay untouched until
 * x3_encode_result() has returned, and the stream is only trusted once it has returned X3_OK (launching x3_decode_dev
 * on the same context in between is fine -- same stream -- as long as its result is only used after that). */
int x3_encode_dev(x3_ctx* ctx, const int16_t* d_wav, const x3_batch* batch, const x3_params* p,
                  uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets);
/* The same for frames taken from anywhere in a device buffer: frame f is the src_samples[f] samples
 * (1 .. block_len * blocks_per_frame) at d_wav + src_offsets[f]; the stream holds them in this order.  A batch of clips of
 * DIFFERENT lengths is such a list (each clip cut into frames as encoder::encode cuts it, encoder.rs:61-73: full frames and
 * a last short one), one launch set for all of them instead of one per clip; d_frame_offsets[F + 1] then gives every clip's
 * byte range.  src_offsets / src_samples are HOST arrays (checked and copied here); offsets that are all even take the
 * single-pass encoders.  Asynchronous like x3_encode_dev; results via x3_encode_result(). */
int x3_encode_frames_dev(x3_ctx* ctx, const int16_t* d_wav, const uint64_t* src_offsets, const uint32_t* src_samples,
                         uint64_t n_frames, const x3_params* p, uint8_t* d_out, uint64_t out_cap, uint64_t start_pos,
                         uint64_t* d_frame_offsets);
/* Waits for the last x3_encode_dev / x3_encode_frames_dev; status is X3_OK, BYTE_WRITER_INSUFFICIENT_MEMORY or BAD_ARG.
 * On BYTE_WRITER_INSUFFICIENT_MEMORY *out_pos = the position the whole stream would have reached, d_frame_offsets holds
 * every frame's offset as if there had been room, and d_out holds every frame that fits (offset + size <= out_cap),
 * complete and in place -- the prefix the reference's slice keeps; a frame that does not fit is not written at all. */
int x3_encode_result(x3_ctx* ctx, uint64_t* out_pos, uint64_t stats[6]);

/* Decode F frames of a device-resident stream.  d_frame_offsets[f] = byte offset of frame f's
 * header in d_x3 (F entries used).  Frame f's samples go to d_wav + d_wav_offsets[f] when
 * d_wav_offsets != NULL, else to the position implied by `batch` and p (the layout
 * x3_encode_dev consumed).  Every frame's header CRC, key, channel count, length and payload CRC
 * are verified on the GPU; d_status (F int32, may be NULL -> internal) receives a status per frame.
 * Asynchronous; results via x3_decode_result(). */
int x3_decode_dev(x3_ctx* ctx, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                  uint64_t n_frames, const x3_batch* batch, const uint64_t* d_wav_offsets,
                  const x3_params* p, int16_t* d_wav, uint64_t wav_cap, int32_t* d_status);
/* ---- The SEGMENT INDEX: decoding a frame on more than one lane.
 * A frame is one serial bit stream (decoder.rs:36-58), so a stream of few frames cannot be decoded faster than one
 * frame's walk, however many lanes there are.  But a block depends only on the bit position it starts at and on the
 * sample in front of it.  The segment index holds both for every `seg_blocks`-th block of every frame: entry
 * [1 + f * (nseg - 1) + j - 1], j = 1 .. nseg - 1 with nseg = ceil(blocks_per_frame / seg_blocks), is the 64-bit word
 * {bits 0..31: bit offset of block seg_blocks * j's header from the start of frame f's payload; bits 32..47: the
 * sample in front of that block; bit 48: entry valid}.  The encoder knows it (x3_encode_dev_seg), and so does a
 * decoder that has been through the stream once (record = 1).  With it, x3_decode_dev_seg decodes nseg stretches of
 * every frame side by side.  The index is NOT part of the .x3a format and is never trusted: every stretch checks where
 * it ended -- bit position and last sample -- against the next entry, which proves (by induction from the frame's
 * first block) that the frame decoded as the serial walk decodes it; a frame with a missing, implausible or
 * contradicted entry goes through the reference's reader, as frames with decode errors do.  A wrong or stale index
 * costs time, never correctness.  Word 0 is a header (who fills the index says so there, and with what seg_blocks; an
 * index without it decodes frame by frame).  The decoder takes as many stretches per frame as fill the GPU -- every
 * entry for a stream of a few dozen frames, every second or fourth for a few hundred, none when the frames alone fill
 * it (option "seg_stretches" overrides; "last_seg_stretches" reports).  seg_blocks: a multiple of 4;
 * d_seg_index: x3_seg_index_entries() words of 8 bytes, 8-byte aligned; NULL = x3_decode_dev / x3_encode_dev.
 * (Layouts the three-wave decoder does not take -- see x3_encode_dev, "Layout" -- decode frame by frame and record
 * nothing: entries stay invalid.) */
uint64_t x3_seg_index_entries(uint64_t n_frames, const x3_params* p, uint32_t seg_blocks);
/* x3_encode_dev that also fills the segment index (seg_blocks: a power of two >= 4; 32 or 64 for 500-block frames).  Only
 * the default-geometry encoder fills it; any other layout leaves an index that says "none" and decodes frame by frame. */
int x3_encode_dev_seg(x3_ctx* ctx, const int16_t* d_wav, const x3_batch* batch, const x3_params* p,
                      uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets,
                      uint64_t* d_seg_index, uint32_t seg_blocks);
int x3_decode_dev_seg(x3_ctx* ctx, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                      uint64_t n_frames, const x3_batch* batch, const uint64_t* d_wav_offsets,
                      const x3_params* p, int16_t* d_wav, uint64_t wav_cap, int32_t* d_status,
                      uint64_t* d_seg_index, uint32_t seg_blocks, int record);
/* ---- Placement (round 6).  Where the x3 stream and where the decoded samples lie in HBM decides the decode phase's pace by
 * up to 10 % -- per PAIR of buffers, reproducibly within a process, and by nothing their addresses show
 * (profiles/r6/decoder_modes.txt).  A pipeline that keeps its buffers allocates a few candidates once and keeps the pair that
 * runs best; this is the measuring loop: for every pair (d_streams[i], d_backs[j]) `warm` untimed and `steps` timed round
 * trips -- x3_encode_dev of the n samples at d_wav into d_streams[i] (capacity cap each; d_frame_offsets: x3_num_frames + 1
 * words), x3_decode_dev of that stream into d_backs[j] (n samples each) -- and ms_per_step[i * n_backs + j] = host wall time
 * per round trip, synchronised.  The samples at d_wav should be of the kind the pipeline will see (the pace follows the
 * stream's density).  X3_ERR_* if a round trip fails or its stream does not decode; the caller frees what it does not keep.
 * (No counterpart in the reference: a property of the device.) */
int x3_place_buffers(x3_ctx* ctx, const int16_t* d_wav, uint64_t n, const x3_params* p, uint8_t* const* d_streams,
                     uint32_t n_streams, uint64_t cap, uint64_t* d_frame_offsets, int16_t* const* d_backs, uint32_t n_backs,
                     uint32_t warm, uint32_t steps, double* ms_per_step);
/* ---- HIP graphs: a launch-bound sequence of device calls, recorded once and replayed with one host call.
 * A short stream's encode + decode is a dozen launches, memsets and event operations of a few microseconds each around
 * kernels of 40-60 us: the host's share of such a step is a third.  Between x3_graph_begin and x3_graph_end the ASYNCHRONOUS
 * device calls of this context -- x3_encode_dev[_seg], x3_encode_frames_dev, x3_decode_dev[_seg] -- are recorded (stream
 * capture on the context's stream; the check pass's side stream joins through its events) instead of launched; nothing
 * may allocate meanwhile, so the same calls must have been made once before, and no call that waits for the GPU
 * (x3_*_result, x3_ctx_sync, the host-buffer entry points) may be made inside.  x3_graph_launch enqueues the whole
 * sequence on the context's stream; x3_encode_result / x3_decode_result then report on the calls in it as if they had just
 * been made.  The graph holds the pointers and sizes the calls were recorded with: replaying it means the same buffers
 * with new contents.  (The decoder's paced priorities follow launch history through a per-launch tag; replays carry
 * one tag and run as a context's first launch does: a graph is for launches too short to be paced.)
 *   MEASURED (round 5, ROCm 7.0.2, profiles/r5/hip_graph_replay.txt): on this stack a replay is SLOWER than the same calls
 * issued back to back on the stream -- config 2's encode + decode by stretches 0.134 against 0.128 ms a step, a 500-frame
 * stream 0.33 against 0.095 -- the runtime executes a captured graph node by node with a barrier behind each.  The entry
 * points are kept (bit-exact, tested) for stacks where that changes; nothing in the library or bench.py uses them.
 *   EXPERIMENTAL: not part of the drop-in surface (the reference has nothing like it), and may go.
 *   A call that fails inside a recording (X3_ERR_BAD_ARG from a buffer that would have to grow, a HIP error) leaves the
 * capture open: end it with x3_graph_end -- it reports the failure or hands back a graph to destroy -- before the context
 * is used again.  x3_graph_begin changes nothing of the context unless the capture has begun. */
typedef struct x3_graph x3_graph;
int x3_graph_begin(x3_ctx* ctx);
int x3_graph_end(x3_ctx* ctx, x3_graph** graph);
int x3_graph_launch(x3_ctx* ctx, x3_graph* graph);
void x3_graph_destroy(x3_graph* graph);
/* Waits for the last x3_decode_dev: index and status of the first frame whose status != 0
 * (first_bad = n_frames, status 0 if all frames are good) and the total samples of good frames
 * before it. */
int x3_decode_result(x3_ctx* ctx, uint64_t* first_bad, int* first_bad_status, uint64_t* samples_before);

/* GPU-side frame walk of a device-resident stream whose frame offsets are not known (SURVEY 8f.2;
 * X3aReader::decode_next_frame, src/decodefile.rs:105-121 + decoder::read_frame_header, src/decoder.rs:69-118):
 * every byte offset is tested for a valid frame header in parallel, the chain off -> off + 20 + payload_len is
 * resolved by pointer doubling.  d_frame_offsets[0..*n_frames) receives the byte offsets of the frames the walk
 * pushes, d_wav_offsets their exclusive sample offsets (both need room for max_frames entries); *terminal is
 * how the walk ends behind them: X3_OK (data exhausted / payload runs past the end), a header error, or
 * X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN.  Synchronous. */
int x3_index_dev(x3_ctx* ctx, const uint8_t* d_x3, uint64_t len, uint64_t max_frames, uint64_t* d_frame_offsets,
                 uint64_t* d_wav_offsets, uint64_t* n_frames, uint64_t* n_samples, int* terminal);
/* x3_decode_stream for device buffers: index (above) + decode, nothing crosses PCIe but the summary.  Same
 * results and status as x3_decode_stream on the same bytes; samples go to d_wav[0..*n_out).  A stream that is one clean
 * chain of frames (what an encoder writes) takes ONE trip to the host: the decode launches are enqueued behind the walk's
 * and read the frame count from device memory (option "two_trips" = 1: wait for the walk first, as before round 5; a
 * stream the walk objects to, or one with more than a frame per KiB, is done that way by itself). */
int x3_decode_stream_dev(x3_ctx* ctx, const uint8_t* d_x3, uint64_t len, const x3_params* p, int16_t* d_wav,
                         uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors);

/* ------------------------------------------------------------------ multi-GPU (no reference analogue) */

/* Frames are independent and 20 + even bytes long, so GPU g encodes a contiguous range of whole frames into its own
 * sub-stream and the sub-streams concatenate without padding; the only coupling is each sub-stream's byte offset.
 * RCCL over xGMI (librccl is opened at run time, on first use): an all-gather of the lengths (8 bytes per rank), and
 * for the reassembly on one rank a grouped ncclSend / ncclRecv, every peer on its own link to the root.
 *
 * x3_shard: ONE rank of a group -- one process (or thread) per GPU, the way torch.distributed.run starts bench.py.
 * Rank 0 makes the id, the others get it out of band; x3_shard_create blocks until all `world` ranks have joined
 * (ncclCommInitRank) and ties the shard to the context's device and stream. */
typedef struct x3_shard x3_shard;
#define X3_SHARD_ID_BYTES 128
int x3_shard_unique_id(uint8_t id[X3_SHARD_ID_BYTES]);
int x3_shard_create(x3_ctx* ctx, const uint8_t id[X3_SHARD_ID_BYTES], int rank, int world, x3_shard** shard);
void x3_shard_destroy(x3_shard* shard);
int x3_shard_rank(const x3_shard* shard);
int x3_shard_world(const x3_shard* shard);
/* Host arithmetic of the sharding.  Rank r owns frames [first, first + count): contiguous, the remainder spread one
 * frame each over the first ranks; samples accordingly (the stream's last frame may be short); starts[r] = exclusive
 * scan of the lengths, starts[world] = total. */
void x3_shard_frame_range(uint64_t n_frames, int rank, int world, uint64_t* first, uint64_t* count);
void x3_shard_sample_range(uint64_t n_samples, const x3_params* p, int rank, int world, uint64_t* first, uint64_t* count);
void x3_shard_offsets(const uint64_t* lengths, int world, uint64_t* starts /* world + 1 */);
/* Step 1, after x3_encode_dev of this rank's samples: all-gather of the sub-stream lengths.  d_len: DEVICE pointer to
 * this rank's length (the last frame offset x3_encode_dev wrote, for a sub-stream that starts at 0); d_lengths:
 * device array of `world` entries or NULL for the shard's own.  Asynchronous on the context's stream.
 * x3_shard_exchange_length_value: the same for a length the host holds.  x3_shard_lengths waits and copies the
 * shard's own array to the host. */
int x3_shard_exchange_lengths(x3_shard* shard, const uint64_t* d_len, uint64_t* d_lengths);
int x3_shard_exchange_length_value(x3_shard* shard, uint64_t len, uint64_t* d_lengths);
int x3_shard_lengths(x3_shard* shard, uint64_t* lengths /* host, world */);
/* Step 2 (optional: a deployment that writes the file in parallel, or decodes where it encoded, never needs it): the
 * whole stream on `root`, d_dst[starts[r] ..) = rank r's d_sub[0 .. lengths[r]).  lengths: host array, identical on
 * all ranks.  d_dst only counts on the root; dst_cap is the ROOT's capacity: every rank that passes it (non-zero) comes
 * to the same verdict before anything is sent, a rank that passes 0 does not check (and would be left waiting in its
 * send if the root refused: size the destination from the lengths first).  Asynchronous on the context's stream. */
int x3_shard_gather(x3_shard* shard, const uint8_t* d_sub, const uint64_t* lengths, int root, uint8_t* d_dst,
                    uint64_t dst_cap, uint64_t* total);
/* The same reassembly beside the context's work: starts behind everything enqueued on the context's stream so far, runs
 * on the shard's own stream and communicator (ncclCommSplit), and the context may go on with its next batch -- into
 * ANOTHER output buffer: d_sub and d_dst stay untouched until x3_shard_gather_wait (on_stream != 0: the context's
 * stream waits, the host does not; 0: the host waits).  One reassembly in flight per shard.  No reference analogue. */
int x3_shard_gather_async(x3_shard* shard, const uint8_t* d_sub, const uint64_t* lengths, int root, uint8_t* d_dst,
                          uint64_t dst_cap, uint64_t* total);
int x3_shard_gather_wait(x3_shard* shard, int on_stream);
/* Step 2, SHARDED: no rank takes in the whole stream.  `encodefile::wav_to_x3a` writes one file through one BufWriter
 * (src/encodefile.rs:66-74); with the frames on N GPUs the equivalent is N writers into ONE file: rank r's sub-stream
 * goes to byte base + starts[r] (x3_shard_offsets) of `fd` -- every rank passes a descriptor of the same file, `base` =
 * what precedes the frames (the archive header, x3_archive_header_write).  Starts behind everything enqueued on the
 * context's stream so far, brings this rank's bytes down over its own host link in 16 MiB pieces (two pinned buffers, a
 * piece is written while the next one is on its way) and returns when they are in the file (pwrite; no fsync).
 * X3_ERR_IO when a write fails.  The single-root reassembly above is bound by the root's seven xGMI links whatever N is;
 * this form scales with the ranks. */
int x3_shard_write_at(x3_shard* shard, const uint8_t* d_sub, const uint64_t* lengths, int fd, uint64_t base,
                      uint64_t* total);

/* x3_mgpu: all GPUs from ONE process -- a context, a shard and a host thread per device.  x3_mgpu_encode /
 * x3_mgpu_decode_stream take and return the same host buffers, bytes and status as x3_encode / x3_decode_stream
 * (encoder::encode, src/encoder.rs:51-111; the walk of src/decodefile.rs:105-136): the samples are dealt out by frame
 * ranges, every GPU encodes its range, the lengths are exchanged and every device copies its sub-stream straight to its
 * place in the caller's host buffer (no reassembly on one GPU first: the destination is host memory);
 * decoding walks the header chain once, deals the frames out and copies every GPU's samples straight to their place
 * (no collective).  x3_mgpu_ctx / x3_mgpu_shard hand out the per-device objects for device-resident use. */
typedef struct x3_mgpu x3_mgpu;
int x3_mgpu_create(const int* devices, int n, x3_mgpu** m);
void x3_mgpu_destroy(x3_mgpu* m);
int x3_mgpu_devices(const x3_mgpu* m);
x3_ctx* x3_mgpu_ctx(x3_mgpu* m, int g);
x3_shard* x3_mgpu_shard(x3_mgpu* m, int g); /* NULL when the group has one GPU */
const char* x3_mgpu_last_error(const x3_mgpu* m);
int x3_mgpu_encode(x3_mgpu* m, const int16_t* wav, uint64_t n, uint32_t n_channels, const x3_params* p, uint8_t* out,
                   uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]);
int x3_mgpu_decode_stream(x3_mgpu* m, const uint8_t* x3, uint64_t len, const x3_params* p, int16_t* wav,
                          uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors);

/* ------------------------------------------------------------------ synthetic inputs (bench/tests) */

/* Seeded, integer-only signal generators.  kind: 0 zeros, 1 white i16 noise,
 * 2 hydrophone-like noise (coloured noise + swell + sparse clicks), 3 fixed-point sine,
 * 4 +-2 LSB random walk.  Sample i depends only on (kind, seed, start+i): the host and device
 * versions are bit-identical and any sub-range can be generated independently. */
int x3_synth(int kind, uint64_t seed, uint64_t start, uint64_t n, int16_t* out);
int x3_synth_dev(x3_ctx* ctx, int kind, uint64_t seed, uint64_t start, uint64_t n, int16_t* d_out);

/* device memory helpers so that ctypes/FFI callers need no HIP binding of their own */
int x3_dev_alloc(x3_ctx* ctx, uint64_t bytes, void** d_ptr);
int x3_dev_free(x3_ctx* ctx, void* d_ptr);
int x3_dev_upload(x3_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);   /* syncs */
int x3_dev_download(x3_ctx* ctx, void* dst, const void* d_src, uint64_t bytes); /* syncs */

#ifdef __cplusplus
}
#endif
#endif
