// x3_internal.h -- what the translation units of libx3hip.so share: the context, its options and scratch buffers, the
// kernel timers, and the prototypes of the host functions that cross a unit's border.  (Until round 4 the library was ONE
// translation unit, x3_api.hip, that #included sixteen headers: VERDICT r3, hygiene.)
//
//   x3_ctx.hip      context, options, timers, launch log, x3.rs parameters, frame headers, CRC entry points, the kernels'
//                   constant tables (x3_tables.h), synthetic signals, device memory helpers
//   x3_encode.hip   encoder kernels + x3_encode* (device and host buffers, the chunked host pipeline), x3_encode_mc
//   x3_decode.hip   decoder, check and index kernels + x3_decode* / x3_index_dev / the stream walks, x3_decode_stream_mc,
//                   the BitReader / BitPacker / decode_block handles (x3_bits.h)
//   x3_files.hip    .x3a archive header, wav <-> x3a in memory and on files, the incremental reader
//   x3_mgpu.hip     x3_shard_* / x3_mgpu_* (librccl through dlopen)
//
// Kernels live in headers; every NON-template kernel header is included by exactly one unit.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/x3hip.h"
#include "x3_fence.h"        // device allocations (guard pages under X3HIP_FENCE)
#include "x3_device.h"
#include "x3_tables.h"

#define X3_INTERNAL __attribute__((visibility("hidden")))

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct KernelTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> used, pool;
};

// Tuning / testing knobs of a context (x3_ctx_set_option).  The X3HIP_* environment variables give their
// initial values and are read ONCE, when the context is created; no call reads the environment afterwards.
struct X3Opts {
  int two_pass = 0;           // X3HIP_TWO_PASS: always use the two-pass encoder kernels
  int stream_wgs = 0;         // X3HIP_STREAM_WGS: workgroups per CU of the single-pass encoder (0 = derive)
  int enc_gen = 3;            // X3HIP_ENC_GEN: 3 = one wave per frame (x3_encode_wave_kernel.h), 2 = eight waves per frame
  int wave_nwg = 0;           // X3HIP_WAVE_NWG: workgroups of the wave encoder (0 = one per CU, at most 256) -- tests: many generations on small inputs
  int wave_m = 0;             // X3HIP_WAVE_M: frames per workgroup generation (0 = derive, 1..16)
  long long wave_drop = -1;   // tests: the workgroup generation whose total the wave encoder never publishes -- what a workgroup
                              // that is not resident looks like to the others: their bounded waits give up (-1 = none)
  int decode_single = 0;      // X3HIP_DECODE_SINGLE: single-wave decoder kernels only
  int decode_blocks_off = 0;  // 1: block lengths 10 and 40 on the single-wave kernels too (what they took until round 6)
  int decode_blocks = 0;      // X3HIP_DECODE_BLOCKS: round 6's block-per-lane decoder (x3_decode_blocks_kernel.h: a walker wave + three
                              // decoder waves per group) where the three-wave kernel would run frame by frame.  Bit-exact and
                              // balanced over the CUs, but it walks every frame twice: 0.82 against 0.65-0.69 ms on config 3
                              // (profiles/r6/decoder_blocks_kernel.txt), so it is not the default
  int host_walk = -1;         // X3HIP_HOST_WALK: frame walk of x3_decode_stream on the host (1) / GPU (0) / by size (-1)
  long long host_chunk_frames = 0;  // X3HIP_HOST_CHUNK_FRAMES: x3_encode on host buffers takes a long input in chunks of this many
                              // frames, upload / encode / download side by side (0 = chunks of 16 Mi samples for inputs from
                              // 32 Mi samples on, -1 = one piece)
  int verbose = 0;            // X3HIP_VERBOSE
  long long file_chunk_frames = 800;  // X3HIP_FILE_CHUNK_FRAMES: 16 MB of samples per chunk (tools/file_bench.py)
  int file_workers = 4;       // X3HIP_FILE_WORKERS
  int check_main = 0;         // X3HIP_CHECK_MAIN: the check pass on the caller's stream and the decoder on the side stream
  long long reader_window_frames = 4096;  // X3HIP_READER_WINDOW_FRAMES: frames x3_reader decodes ahead per launch set
  int check_prio = 1;         // X3HIP_CHECK_PRIO: queue priority of the side stream the check kernel runs on (-1 low, 0 same, 1 high)
  int check_first = 0;        // X3HIP_CHECK_FIRST: enqueue the check kernel in front of the decoder (1) or behind it (0)
  int index_no_fast = 0;      // 1: x3_index_dev / x3_decode_stream_dev always take the general walk (hash + pointer doubling)
  int wav_offsets_x4 = 0;     // 1: the caller promises that every d_wav_offsets[] given to x3_decode_dev is a multiple of four samples
  int mc_decode_threads = 0;  // multi-channel decode: 1 = one thread per frame (the pre-round-4 kernel) for every frame
  int check_wgs = 4;          // X3HIP_CHECK_WGS: check-kernel workgroups per CU (8 until the kernel got leaner in round 3: gpurun_out sweep in profiles/r3)
#ifdef X3_PROFILING
  // profiling builds only (-DX3_PROFILING): never in the shipped library
  int check_serial = 0;       // X3HIP_CHECK_SERIAL: the check pass in front of the decoder, same stream
  int no_check = 0;           // X3HIP_PROFILE_NO_CHECK: time the decoder alone (payload CRCs NOT verified)
#endif
  int two_trips = 0;          // 1: x3_decode_stream_dev always waits for the frame walk before it launches the decoder (the pre-round-5 path)
  int seg_stretches = 0;      // x3_decode_dev_seg: stretches per frame (0 = as many as fill the chip; 1 = never by stretches)
  long long lb_drop = -1;     // tests: the frame whose look-back descriptor is never published (general one-pass encoder)
  int dyn_lds = 0;            // X3HIP_DECODE_DYN_LDS: extra LDS per decoder group = fewer groups per CU (occupancy experiments)
};

// a recorded sequence of device calls (x3_graph_*): the graph and what x3_encode_result / x3_decode_result need to know
// about the calls in it
struct x3_graph;
// the segment index of a call (x3_decode_split_kernel.h, "STRETCHES"): mode 1 = decode by it, 2 = record it
struct X3SegSpec { uint64_t* d_index; uint32_t seg_blocks; int mode; };

struct x3_ctx {
  X3Opts opt;
  X3SegSpec enc_seg{nullptr, 0, 0};         // x3_encode_dev_seg -> encode_dev_impl: the index the next launch fills
  unsigned long long stream_one_trip = 0;   // x3_decode_stream_dev calls served with one trip to the host
  int last_seg_stretches = 0;               // the last decode launch: stretches per frame (0: none given, -1: recorded)
  int last_decode_kernel = 0;               // ... and which kernel served it (option "decode_kernel_in_use")
  unsigned long long needed_pos = 0;        // the position a host-buffer encode that ran out of room would have reached
  unsigned long long encode_fallbacks = 0;  // launches of the single-pass encoder that timed out (two-pass re-run)
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t stream2 = nullptr;        // side stream: the payload-CRC pass runs beside the decoder
  hipStream_t dl_stream = nullptr, ul_stream = nullptr;  // the down- and uploads of long host buffers taken in chunks
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string last_error;
  // persistent small device state
  uint16_t* d_xpow = nullptr;          // X3_XP_SIZE entries
  uint32_t* d_xk2 = nullptr;           // [X3_K2_MAXC][X3_K2_DWORDS]: per-lane and per-wave multipliers (x3_encode_stream2_kernel.h)
  uint32_t* d_wtab = nullptr;          // X3W_TAB_BYTES: the LDS tables of x3_encode_wave_kernel
  uint16_t* d_crctab = nullptr;        // [6][256]: slicing-by-4 CRC tables + the two x^2048 rows
  uint32_t* d_kx64 = nullptr;          // [64][16] (x3_frame_check_kernel)
  uint16_t* d_chktab = nullptr;        // [18][256] (x3_frame_check_kernel: T[s][k][v] and the x^8192 rows)
  uint16_t* d_xinv8 = nullptr;         // x^(-8k), k < X3_CHECK_XINV_N
  int* d_status = nullptr;             // [0] size/scan pass, [1] encode pass
  unsigned long long* d_stats = nullptr;    // 6
  // The encoders' control block (status, statistics, end position, dense count: 128 bytes) exists twice; a call uses the one
  // the call before did not (ctl_begin, x3_encode.hip), and d_status / d_stats / d_end_pos point into it.  A wave-encoder
  // call's last kernel clears the other one: the next call then starts without a memset in front of its first kernel
  // (4 us of fill + a launch gap on a stream that has nothing else to do: 1 % of config 3's step, 10 % of config 2's).
  bool last_enc_mc = false;   // the pending encode is x3_encode_mc's (its look-back fallback is its own: x3_mc.h)
  int32_t* d_ctl_base = nullptr;
  int ctl_half = 0;
  bool ctl_clean[2] = {false, false};
  unsigned long long* d_end_pos = nullptr;  // 1
  DevBuf lb_desc;                            // the general single-pass encoder's look-back descriptors (u64 per frame)
  uint32_t lb_epoch = 0;
  X3DecodeSummary* d_summary = nullptr;
  struct x3_reader* fcache = nullptr;  // x3_decode_prefetch: the frame stream x3_decode_frame calls are served from
  uint32_t* d_pace = nullptr;          // x3_decode_split_kernel's pace word (see there), dec_epoch its launch count
  uint32_t dec_epoch = 1;
  uint32_t enc_log_epoch = 0;          // launches of the wave encoder (its launch-log entries are indexed by it)
  uint16_t* d_crc = nullptr;
  // pinned mirrors
  int* h_status = nullptr;
  unsigned long long* h_stats = nullptr;  // 6 stats + end_pos
  X3DecodeSummary* h_summary = nullptr;
  X3DecodeSummary* h_summary_init = nullptr;
  int32_t* dec_status_ptr = nullptr;
  uint16_t* h_crc = nullptr;
  void* h_walk = nullptr;  // frame and sample offsets of a host walk on their way to the device
  size_t h_walk_cap = 0;
  // growable scratch
  DevBuf in, out, frame_bytes, frame_off, dec_status, dec_cstatus, dec_meta, wav_off, seg_crc, desc, dense_list;
  DevBuf in_more[2], out_more[2];  // x3_decode_stream on a long host buffer: rings of three buffers on either side of the decoder
  DevBuf idx_cand, idx_keys, idx_vals, idx_J, idx_S, idx_L, idx_sum;  // x3_index_dev scratch
  DevBuf idx_wg, idx_sorted, idx_scan;  // ... of its fast path: candidates per scanning workgroup, in order, the scans
  unsigned long long index_fast = 0, index_general = 0;  // walks that the fast path / the general path have served (options)
  int n_cus = 0;
  bool force_single_wave_decode = false;
  uint32_t desc_epoch = 0;    // tag of the current launch's frame-size descriptors (single-pass encoders)
  int stream_wg_per_cu = -1;  // co-resident workgroups per CU of x3_encode_stream2_kernel (-1 = not queried)
  uint64_t stream_wg_key = 0; // ... of which instantiation with how much LDS (block length, table form, bytes)
  // bookkeeping of the last async calls
  bool encode_pending = false, decode_pending = false;
  bool force_two_pass = false;
  // Contexts that encode concurrently on ONE GPU (the file pipeline's workers) share this gate: the single-pass encoders
  // are persistent grids whose workgroups wait for each other, so only one of them may be in flight on a device.  A
  // context holds the gate from its launch to the end of x3_encode_result; copies and file I/O stay outside.
  std::mutex* enc_gate = nullptr;
  // Dense content.  A frame that does not fit the wave encoder's LDS image (payload > X3_DENSE_PAYLOAD_BYTES) is left
  // to the dense pass that follows the wave kernel in the same stream (x3_encode_stream2_kernel<true>): no call is ever
  // encoded twice.  `prefer_gen2` is a speed hint only: a call in which more than a quarter of the frames were dense
  // (white noise, full-scale music) makes the NEXT call of the context start on the second-generation kernel, which
  // holds worst-case images and saves such content the wave kernel's analysis pass; that kernel counts dense frames too,
  // and below an eighth the context is back on the wave encoder.  Bytes are the same either way.
  bool prefer_gen2 = false;
  int last_enc_gen = 0;       // which kernel generation served the pending / last encode (3 wave + dense pass, 2, 1; 0 two-pass)
  unsigned long long encode_dense_frames = 0;   // frames handed to the dense pass so far (read-only option)
  unsigned long long last_dense_frames = 0;     // of the last call (either generation counts them)
  struct LastEnc {
    const int16_t* d_wav; x3_batch b; x3_params p; uint64_t spf; uint8_t* d_out; uint64_t out_cap, start_pos; uint64_t* d_off;
    const uint64_t* src_off; const uint32_t* src_n; bool src_even;   // x3_encode_frames_dev's frame table (device), or nullptr
    X3SegSpec seg;                                                    // x3_encode_dev_seg's index (a re-run says "none" in it)
  } last_enc;
  // x3_graph_begin .. x3_graph_end: the device calls in between are recorded into a HIP graph instead of launched.  Nothing
  // may allocate meanwhile (ensure() fails instead), and the encoders' descriptor words -- tagged with a per-launch epoch so
  // that last launch's totals read as "not there yet" -- are cleared by a memset node of the graph instead (every replay of
  // the graph carries the same epoch).
  bool capturing = false;
  bool timing_before_capture = false;   // (kernel timing rides on events of the dispatch packets: off while recording)
  DevBuf src_tab;   // that table: F offsets (u64), then F sample counts (u32)
  void* h_src_tab = nullptr;        // ... and its pinned host copy (the caller's arrays are only read inside the call)
  size_t h_src_tab_cap = 0;
  hipEvent_t ev_src_tab = nullptr;  // the copy out of it has been done
  uint64_t enc_start_pos = 0;
  uint64_t dec_frames = 0;
  // kernel timing
  bool timing = false;
  uint32_t timing_mask = 0xFFFFFFFFu;   // option "kernel_timing_mask": which kernels (bit = the id x3_ctx_kernel_time takes) carry events
  KernelTimer timers[6];   // encode, decode, sizes, scan, check, dense pass
};

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      if (ctx) (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);           \
      return X3_ERR_HIP;                                                                        \
    }                                                                                           \
  } while (0)


// ---- kernel timing
// Two ways of timing a launch with HIP events.  The plain one brackets the launch with two hipEventRecord on its stream: the
// events are packets of their own, and what lies between them is the kernel plus a few microseconds of queue (2 % of a
// 0.43 ms kernel: the HIP-event averages of round 2's bench line sat 2-4 % off rocprofv3's).  `attached`: the events ride
// on the kernel's own dispatch packet (hipExtLaunchKernelGGL) and hold its begin and end -- what rocprofv3's kernel
// trace reports; the three kernels of the round trip are launched that way (X3_LAUNCH_TIMED).
struct TimerScope {
  x3_ctx* c;
  int which;
  hipStream_t st;
  bool attached;
  bool on;
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  TimerScope(x3_ctx* c_, int w, hipStream_t s_ = nullptr, bool attached_ = false)
      : c(c_), which(w), st(s_ ? s_ : c_->stream), attached(attached_), on(c_->timing && ((c_->timing_mask >> w) & 1u)) {
    if (!on) return;
    KernelTimer& t = c->timers[which];
    if (!t.pool.empty()) {
      ev = t.pool.back();
      t.pool.pop_back();
    } else {
      (void)hipEventCreate(&ev.first);
      (void)hipEventCreate(&ev.second);
    }
    if (!attached) (void)hipEventRecord(ev.first, st);
  }
  ~TimerScope() {
    if (!on) return;
    if (!attached) (void)hipEventRecord(ev.second, st);
    c->timers[which].used.push_back(ev);
  }
};
// launch `kernel` on `stream` inside the TimerScope `ts` (constructed with attached = true)
#define X3_LAUNCH_TIMED(ts, kernel, grid, block, smem, stream, ...)                                        \
  do {                                                                                                     \
    if ((ts).on) hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, (ts).ev.first, (ts).ev.second, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                               \
  } while (0)


// A stage of the host-buffer pipelines below hands work to the next through one of these (one producer, one consumer).
template <class T>
struct X3Handoff {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<T> q;
  bool closed = false;
  void push(T v) {
    { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(v)); }
    cv.notify_all();
  }
  void close() {
    { std::lock_guard<std::mutex> g(mu); closed = true; }
    cv.notify_all();
  }
  bool pop(T* v) {  // false: closed and empty
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return closed || !q.empty(); });
    if (q.empty()) return false;
    *v = std::move(q.front());
    q.pop_front();
    return true;
  }
};
// a count one thread advances and another waits for
struct X3Progress {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t n = 0;
  bool stop = false;
  void advance() {
    { std::lock_guard<std::mutex> g(mu); ++n; }
    cv.notify_all();
  }
  void halt() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
  }
  bool wait_for(uint64_t want) {  // false: halted first
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return stop || n >= want; });
    return n >= want;
  }
};
#define X3_PIPE_UNAVAILABLE (-1000)  // (internal) the chunked front end could not start its threads


// The host side of X3aReader::decode_next_frame's walk (decodefile.rs:105-121) over `buf`, a window of
// `buf_len` bytes at the head of `real_total` bytes that really exist (a file read in pieces; the same for
// an in-memory stream) of which the reader BELIEVES `believed_total` remain (X3aReader::open subtracts the
// archive header without its 8-byte id, decodefile.rs:62-66: 8 phantom bytes).  Collects the frames the walk
// steps over -- and the one frame the decoder will refuse, where there is one -- and says how it ends:
// `need_more`: the window ran out (or `max_samples` were collected) at *end_pos, the walk goes on from there;
// otherwise *terminal is what the reference's walk returns if every frame before decodes.
struct HostWalk {
  std::vector<uint64_t> offs, woffs;
  uint64_t nsamp = 0, end_pos = 0;
  int terminal = X3_OK;
  bool need_more = false;
};

struct EncPlan {
  X3DevParams dp;
  X3Geom g;
  uint32_t nthr, lds_in_bytes, img_dwords;
  size_t smem;
};

#define X3_FRAME_CACHE_MISS (-1)

// ---- x3_ctx.hip
X3_INTERNAL int ensure(x3_ctx* c, DevBuf& b, size_t bytes);
// an encode call takes the control block the call before did not use (x3_ctx::d_ctl_base), cleared -- by the last kernel of
// the call before where that was a wave-encoder call, by a memset otherwise (and always while a graph is recorded: every
// replay runs the same nodes); x3_encode.hip
X3_INTERNAL int ctl_begin(x3_ctx* c);
// x3_encode_result to x3_encode_mc: the one-pass kernel's look-back gave up, encode again in two passes (never leaves the library)
#define X3_RETRY_TWO_PASS (-1000)
X3_INTERNAL int x3_pipe_streams(x3_ctx* c);
X3_INTERNAL uint64_t spf_of(const x3_params* p);
X3_INTERNAL uint64_t max_payload_bytes(uint64_t n, uint32_t block_len);
X3_INTERNAL int derive(const x3_params* p, uint64_t spf, X3DevParams* d);
X3_INTERNAL uint16_t header_crc16_host(const uint8_t* b, size_t n);
X3_INTERNAL int read_frame_header_ch(const uint8_t* b, uint64_t len, x3_frame_header* h, uint32_t n_ch);
X3_INTERNAL int crc_dev_async(x3_ctx* c, const uint8_t* d_data, uint64_t n);
X3_INTERNAL extern const uint32_t X3_RICE_OFFSET[4], X3_RICE_LEN[4];
// ---- x3_encode.hip
X3_INTERNAL int plan_encode(x3_ctx* c, const x3_batch* b, const x3_params* p, uint64_t spf, EncPlan* pl);
// x3_encode_frames_dev: frame f = src_n[f] samples at d_wav + src_off[f] (device pointers); even: every offset is a multiple of two
struct X3FrameTable { const uint64_t* src_off; const uint32_t* src_n; bool even; };
X3_INTERNAL int encode_dev_impl(x3_ctx* c, const int16_t* d_wav, const x3_batch* b, const x3_params* p, uint64_t spf,
                                uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets,
                                const struct X3FrameTable* tab = nullptr);
X3_INTERNAL int encode_host(x3_ctx* c, const int16_t* const* wavs, uint64_t n_per_clip, uint64_t n_clips,
                            const x3_params* p, uint64_t spf, uint8_t* out, uint64_t out_cap, uint64_t start_pos,
                            uint64_t* out_pos, uint64_t* clip_offsets, uint64_t stats[6]);
// ---- x3_decode.hip
X3_INTERNAL int decode_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                                uint64_t F, const x3_batch* batch, const uint64_t* d_wav_offsets, const x3_params* p,
                                int16_t* d_wav, uint64_t wav_cap, int32_t* d_status, bool wav_off_aligned = false,
                                bool bl0 = false, const X3SegSpec* seg = nullptr,
                                const unsigned long long* d_nf = nullptr);  // d_nf: the real frame count, on the device; F is a bound
X3_INTERNAL int decode_stream_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                       int16_t* d_wav, uint64_t wav_cap, DevBuf* own_out, uint64_t* n_out,
                                       uint64_t* frames_ok, uint64_t* frame_errors);
X3_INTERNAL int decode_stream_impl(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                   int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                                   uint64_t* frame_errors);
X3_INTERNAL void walk_host(const uint8_t* buf, uint64_t buf_len, uint64_t real_total, uint64_t believed_total,
                           const x3_params* p, uint64_t wav_cap, uint64_t max_samples, HostWalk* w, uint32_t n_ch = 1u);
X3_INTERNAL int decode_frames_host(x3_ctx* c, const uint8_t* x3, uint64_t len, const HostWalk& w, const x3_params* p,
                                   int16_t* wav, uint64_t wav_cap, uint64_t* before, uint64_t* first_bad, int* bad_status,
                                   bool download = true,  // !download: the samples stay in c->out (x3_mgpu_decode_stream)
                                   const uint8_t* d_x3 = nullptr);  // the frames' bytes are on the device already
X3_INTERNAL int walk_result(uint64_t F, uint64_t first_bad, int bad_status, int terminal, uint64_t* frame_errors);
// ---- x3_files.hip (x3_reader.h)
// the RIFF/WAVE header parser of x3_wav_to_x3a on an open file, for the sanitised host tests (tests/host_cpp/fuzz_host_parsers.cpp)
X3_INTERNAL int x3_wav_parse_fd_for_tests(int fd, uint64_t file_len, uint32_t* sample_rate, uint16_t* channels, uint16_t* bits,
                                          uint64_t* data_off, uint64_t* data_len);
X3_INTERNAL int frame_cache_serve(x3_ctx* c, const uint8_t* payload, uint64_t len, const x3_params* p, uint64_t samples, int16_t* wav);
X3_INTERNAL void reader_free_internal(struct x3_reader* r);   // (x3_ctx_destroy: the frame cache of x3_decode_prefetch)
