// x3_api.hip -- the C ABI of include/x3hip.h on top of the gfx950 kernels.
//
// Host logic only: argument checks that mirror the reference's (or its panics), derivation of
// the kernel parameters from x3::Parameters, scratch-buffer management, the 20-byte frame
// header helpers, and the host side of the stream walk.  All sample/bit/CRC work over payloads
// is done by the kernels in x3_encode_kernel.h / x3_decode_kernel.h / x3_util_kernels.h.
// There is no CPU fallback: every bulk entry point needs a live x3_ctx (a HIP device).
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
#include "x3_decode_kernel.h"
#include "x3_decode_split_kernel.h"
#include "x3_index_kernels.h"
#include "x3_device.h"
#include "x3_encode_kernel.h"
#include "x3_encode_stream2_kernel.h"
#include "x3_encode_wave_kernel.h"
#include "x3_synth_core.h"
#include "x3_util_kernels.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
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
  int check_wgs = 4;          // X3HIP_CHECK_WGS: check-kernel workgroups per CU (8 until the kernel got leaner in round 3: gpurun_out sweep in profiles/r3)
#ifdef X3_PROFILING
  // profiling builds only (-DX3_PROFILING): never in the shipped library
  int check_serial = 0;       // X3HIP_CHECK_SERIAL: the check pass in front of the decoder, same stream
  int no_check = 0;           // X3HIP_PROFILE_NO_CHECK: time the decoder alone (payload CRCs NOT verified)
  int dyn_lds = 0;            // X3HIP_DECODE_DYN_LDS: extra LDS per decoder group (occupancy experiments)
#endif
};

struct x3_ctx {
  X3Opts opt;
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
  unsigned long long* d_end_pos = nullptr;  // 1
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
  int n_cus = 0;
  bool force_single_wave_decode = false;
  uint32_t desc_epoch = 0;    // tag of the current launch's frame-size descriptors (single-pass encoders)
  int stream_wg_per_cu = -1;  // co-resident workgroups per CU of x3_encode_stream_kernel (-1 = not queried)
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
  struct {
    const int16_t* d_wav; x3_batch b; x3_params p; uint64_t spf; uint8_t* d_out; uint64_t out_cap, start_pos; uint64_t* d_off;
  } last_enc;
  uint64_t enc_start_pos = 0;
  uint64_t dec_frames = 0;
  // kernel timing
  bool timing = false;
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

static int ensure(x3_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return X3_OK;
  if (b.p) HIPCHK(c, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = std::max(bytes, (size_t)4096);
  want = (want + 255) & ~(size_t)255;
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  return X3_OK;
}

// ---- GF(2)[x] mod 0x11021 on the host (for the x^n table only)
static uint32_t gf_mul_host(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 0; i < 16; ++i) {
    if ((a >> i) & 1u) r ^= b;
    b = (b << 1) ^ ((b & 0x8000u) ? 0x11021u : 0u);
  }
  return r & 0xFFFFu;
}
static uint32_t gf_xpow_host(uint64_t e) {  // x^e mod P
  uint32_t result = 1, base = 2;
  while (e) {
    if (e & 1) result = gf_mul_host(result, base);
    base = gf_mul_host(base, base);
    e >>= 1;
  }
  return result;
}

static void opts_from_env(X3Opts* o) {
  auto geti = [](const char* name, long long dflt) -> long long {
    const char* e = std::getenv(name);
    return e && *e ? std::strtoll(e, nullptr, 10) : dflt;
  };
  o->two_pass = std::getenv("X3HIP_TWO_PASS") ? 1 : 0;
  o->stream_wgs = (int)std::max(0ll, geti("X3HIP_STREAM_WGS", 0));
  o->decode_single = std::getenv("X3HIP_DECODE_SINGLE") ? 1 : 0;
  o->enc_gen = (int)geti("X3HIP_ENC_GEN", o->enc_gen) == 2 ? 2 : 3;
  o->wave_nwg = (int)std::max(0ll, std::min(256ll, geti("X3HIP_WAVE_NWG", 0)));
  o->wave_m = (int)std::max(0ll, std::min(16ll, geti("X3HIP_WAVE_M", 0)));
  if (const char* e = std::getenv("X3HIP_HOST_WALK")) o->host_walk = e[0] == '0' ? 0 : 1;
  o->host_chunk_frames = std::max(-1ll, geti("X3HIP_HOST_CHUNK_FRAMES", o->host_chunk_frames));
  o->verbose = std::getenv("X3HIP_VERBOSE") ? 1 : 0;
  o->file_chunk_frames = std::max(1ll, geti("X3HIP_FILE_CHUNK_FRAMES", o->file_chunk_frames));
  o->file_workers = (int)std::max(1ll, std::min(16ll, geti("X3HIP_FILE_WORKERS", o->file_workers)));
  o->reader_window_frames = std::max(1ll, geti("X3HIP_READER_WINDOW_FRAMES", o->reader_window_frames));
  o->check_main = (int)geti("X3HIP_CHECK_MAIN", o->check_main);
  o->check_prio = (int)geti("X3HIP_CHECK_PRIO", o->check_prio);
  o->check_first = (int)geti("X3HIP_CHECK_FIRST", o->check_first);
  o->check_wgs = (int)std::max(1ll, geti("X3HIP_CHECK_WGS", o->check_wgs));
#ifdef X3_PROFILING
  o->check_serial = std::getenv("X3HIP_CHECK_SERIAL") ? 1 : 0;
  o->no_check = std::getenv("X3HIP_PROFILE_NO_CHECK") ? 1 : 0;
  o->dyn_lds = (int)std::max(0ll, geti("X3HIP_DECODE_DYN_LDS", 0));
#endif
}

static int ctx_init(x3_ctx* c, int device, hipStream_t stream, bool own) {
  opts_from_env(&c->opt);
  int count = 0;
  HIPCHK(c, hipGetDeviceCount(&count));
  if (device < 0 || device >= count) {
    c->last_error = "no such HIP device";
    return X3_ERR_HIP;
  }
  HIPCHK(c, hipSetDevice(device));
  c->device = device;
  {
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    c->n_cus = prop.multiProcessorCount;
  }
  if (own) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  } else {
    c->stream = stream;
  }
  {
    // The side stream must not share a hardware queue with the caller's stream, or the check kernel runs behind
    // the decoder instead of beside it (seen once RCCL had created its own streams: HIP maps streams of one
    // priority onto a few hardware queues round robin).  Streams of another priority get queues of their own.
    int lo = 0, hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = numerically greatest = lowest priority
    if (hi < lo && c->opt.check_prio != 0)
      HIPCHK(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, c->opt.check_prio > 0 ? hi : lo));
    else
      HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  }
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIPCHK(c, hipMalloc(&c->d_xpow, X3_XP_SIZE * sizeof(uint16_t)));
  // status (4 ints) and stats (6 + end_pos) share one 128-byte block: one memset, one copy back
  HIPCHK(c, hipMalloc(&c->d_status, 128));
  c->d_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_status) + 32);
  c->d_end_pos = c->d_stats + 6;
  HIPCHK(c, hipMalloc(&c->d_summary, sizeof(X3DecodeSummary)));
  HIPCHK(c, hipMalloc(&c->d_pace, X3_PACE_WORDS * sizeof(uint32_t)));
  HIPCHK(c, hipMemset(c->d_pace, 0, X3_PACE_WORDS * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->d_crc, 16));
  HIPCHK(c, hipHostMalloc(&c->h_status, 128));
  c->h_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_status) + 32);
  HIPCHK(c, hipHostMalloc(&c->h_summary, sizeof(X3DecodeSummary)));
  HIPCHK(c, hipHostMalloc(&c->h_summary_init, 256));  // (also the pinned landing place of X3IndexSummary)
  HIPCHK(c, hipHostMalloc(&c->h_crc, 16));
  std::vector<uint16_t> xp(X3_XP_SIZE);
  for (int j = 0; j < X3_XP_LEVELS; ++j)
    for (int m = 0; m <= X3_XP_M; ++m) xp[j * (X3_XP_M + 1) + m] = (uint16_t)gf_xpow_host(32ull * m * (1ull << j));
  // x^-1 = x^15 + x^11 + x^4 (x * that = x^16 + x^12 + x^5 = P + 1); x^-16 = (x^-1)^16
  uint32_t xi = 0x8810u;
  for (int i = 0; i < 4; ++i) xi = gf_mul_host(xi, xi);
  xp[X3_XINV16_INDEX] = (uint16_t)xi;
  {
    uint32_t xi8 = 0x8810u;  // x^-1
    for (int i = 0; i < 3; ++i) xi8 = gf_mul_host(xi8, xi8);  // x^-8
    uint32_t acc = 1;
    for (int t = 1; t <= 3; ++t) {
      acc = gf_mul_host(acc, xi8);
      xp[X3_XINV8_INDEX(t)] = (uint16_t)acc;
    }
  }
  HIPCHK(c, hipMemcpy(c->d_xpow, xp.data(), X3_XP_SIZE * sizeof(uint16_t), hipMemcpyHostToDevice));
  {
    {
      // x3_encode_stream2_kernel: KN[l] = nibble tables of x^(32*c*(63-l)), KA[w] = x^(32*c*64*(7-w)) as its sixteen shifts
      std::vector<uint32_t> k2((size_t)X3_K2_MAXC * X3_K2_DWORDS, 0u);
      for (int cd = 1; cd <= (int)X3_K2_MAXC; ++cd) {
        uint32_t* blk = k2.data() + (size_t)(cd - 1) * X3_K2_DWORDS;
        for (int l = 0; l < 64; ++l) {
          const uint32_t k = gf_xpow_host(32ull * cd * (63 - l));
          uint16_t* row = reinterpret_cast<uint16_t*>(blk + l * X3_K2_ROW);
          for (int j = 0; j < 4; ++j)
            for (uint32_t v = 0; v < 16; ++v) row[j * 16 + v] = (uint16_t)gf_mul_host(v << (4 * j), k);
        }
        for (int w = 0; w < 8; ++w) {
          uint32_t k = gf_xpow_host(32ull * cd * 64 * (7 - w));
          for (int b = 0; b < 16; ++b) {
            blk[X3_K2_KA + w * 16 + b] = k;
            k = ((k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u)) & 0xFFFFu;
          }
        }
      }
      HIPCHK(c, hipMalloc(&c->d_xk2, k2.size() * sizeof(uint32_t)));
      HIPCHK(c, hipMemcpy(c->d_xk2, k2.data(), k2.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // T[j][v] = crc0 of byte v followed by j zero bytes = v * x^(8j+16) mod P
    // rows 4 and 5 (x3_frame_check_kernel): (v << 8) * x^2048 and v * x^2048
    std::vector<uint16_t> tab(6 * 256);
    for (int j = 0; j < 4; ++j) {
      const uint32_t sh = gf_xpow_host(8ull * j + 16);
      for (int v = 0; v < 256; ++v) tab[j * 256 + v] = (uint16_t)gf_mul_host((uint32_t)v, sh);
    }
    for (int v = 0; v < 256; ++v) {
      tab[4 * 256 + v] = (uint16_t)gf_mul_host((uint32_t)v, gf_xpow_host(2048 + 8));
      tab[5 * 256 + v] = (uint16_t)gf_mul_host((uint32_t)v, gf_xpow_host(2048));
    }
    {
      // per-lane constants of x3_frame_check_kernel: x^(32*(63-t)) * x^b, b = 0..15
      std::vector<uint32_t> kx((size_t)64 * 16);
      for (int t = 0; t < 64; ++t) {
        uint32_t k = gf_xpow_host(32ull * (63 - t));
        for (int b = 0; b < 16; ++b) {
          kx[(size_t)t * 16 + b] = k;
          k = ((k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u)) & 0xFFFFu;
        }
      }
      HIPCHK(c, hipMalloc(&c->d_kx64, kx.size() * sizeof(uint32_t)));
      HIPCHK(c, hipMemcpy(c->d_kx64, kx.data(), kx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    {
      // x3_frame_check_kernel: T0[k][v] = v * x^(8k + 16), M2[k][v] = v * x^(8k + 2048), M4[k][v] = v * x^(8k + 4096)
      std::vector<uint16_t> ct(X3_CHECK_TAB_U16);
      const uint64_t shifts[3] = {16, 2048, 4096};
      for (int t = 0; t < 3; ++t)
        for (int k = 0; k < 4; ++k) {
          const uint32_t sh = gf_xpow_host(8ull * k + shifts[t]);
          for (int v = 0; v < 256; ++v) ct[((size_t)t * 4 + k) * 256 + v] = (uint16_t)gf_mul_host((uint32_t)v, sh);
        }
      HIPCHK(c, hipMalloc(&c->d_chktab, ct.size() * sizeof(uint16_t)));
      HIPCHK(c, hipMemcpy(c->d_chktab, ct.data(), ct.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      // x^(-8k): x has order 32767 modulo P (P = (x + 1) * a primitive polynomial of degree 15)
      std::vector<uint16_t> xi(X3_CHECK_XINV_N);
      for (uint32_t k = 0; k < X3_CHECK_XINV_N; ++k) xi[k] = (uint16_t)gf_xpow_host((32767ull * 8 - 8ull * k) % 32767ull);
      HIPCHK(c, hipMalloc(&c->d_xinv8, xi.size() * sizeof(uint16_t)));
      HIPCHK(c, hipMemcpy(c->d_xinv8, xi.data(), xi.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    {
      // x3_encode_wave_kernel: M[k][v] = v * x^(8k + 4096) (byte k of a 32-bit chain state, two rows of 64 dwords on), the
      // two "16-bit state times x^2048" rows above, then per lane the sixteen shifts of its weight as uint16, then x^(-16k).
      // A lane's folded column is held times x^4096; the weight x^(32*(63-t)) carries the x^(16 - 4096) that makes it a CRC.
      std::vector<uint16_t> wt(X3W_TAB_BYTES / 2);
      for (int k = 0; k < 4; ++k) {
        const uint32_t sh = gf_xpow_host(8ull * k + 4096);
        for (int v = 0; v < 256; ++v) wt[(size_t)k * 256 + v] = (uint16_t)gf_mul_host((uint32_t)v, sh);
      }
      for (size_t i = 4 * 256; i < 6 * 256; ++i) wt[i] = tab[i];
      for (int t = 0; t < 64; ++t) {
        uint32_t k = gf_xpow_host((32ull * (63 - t) + 16 + 32767ull - 4096) % 32767ull);
        for (int b = 0; b < 16; ++b) {
          wt[1536 + (size_t)t * 16 + b] = (uint16_t)k;
          k = ((k << 1) ^ ((k & 0x8000u) ? 0x11021u : 0u)) & 0xFFFFu;
        }
      }
      // x^(-16k), k < 128 (x has order 32767 modulo P): undoes the zero bytes behind a payload in its last image row
      for (uint64_t k = 0; k < 128; ++k) wt[2560 + k] = (uint16_t)gf_xpow_host((32767ull * 16 - 16ull * k) % 32767ull);
      HIPCHK(c, hipMalloc(&c->d_wtab, X3W_TAB_BYTES));
      HIPCHK(c, hipMemcpy(c->d_wtab, wt.data(), X3W_TAB_BYTES, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(&c->d_crctab, tab.size() * sizeof(uint16_t)));
    HIPCHK(c, hipMemcpy(c->d_crctab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  return X3_OK;
}

extern "C" int x3_ctx_create(int device, x3_ctx** ctx) {
  if (!ctx) return X3_ERR_BAD_ARG;
  *ctx = nullptr;
  x3_ctx* c = new x3_ctx();
  int rc = ctx_init(c, device, nullptr, true);
  if (rc) {
    std::fprintf(stderr, "x3hip: cannot create context on device %d: %s\n", device, c->last_error.c_str());
    delete c;
    return rc;
  }
  *ctx = c;
  return X3_OK;
}

extern "C" int x3_ctx_create_on_stream(int device, void* hip_stream, x3_ctx** ctx) {
  if (!ctx) return X3_ERR_BAD_ARG;
  *ctx = nullptr;
  x3_ctx* c = new x3_ctx();
  int rc = ctx_init(c, device, (hipStream_t)hip_stream, false);
  if (rc) {
    std::fprintf(stderr, "x3hip: cannot create context on device %d: %s\n", device, c->last_error.c_str());
    delete c;
    return rc;
  }
  *ctx = c;
  return X3_OK;
}

extern "C" void x3_ctx_destroy(x3_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->fcache) x3_reader_close(c->fcache);
  for (DevBuf* b : {&c->in, &c->out, &c->in_more[0], &c->in_more[1], &c->out_more[0], &c->out_more[1], &c->frame_bytes, &c->frame_off, &c->dec_status, &c->dec_cstatus, &c->dec_meta, &c->wav_off,
                    &c->seg_crc, &c->desc, &c->idx_cand, &c->idx_keys, &c->idx_vals, &c->idx_J, &c->idx_S,
                    &c->idx_L, &c->idx_sum})
    if (b->p) (void)hipFree(b->p);
  for (auto& t : c->timers) {
    for (auto& e : t.used) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : t.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
  (void)hipFree(c->d_xpow);
  (void)hipFree(c->d_xk2);
  (void)hipFree(c->d_wtab);
  (void)hipFree(c->d_crctab);
  (void)hipFree(c->d_kx64);
  (void)hipFree(c->d_chktab);
  (void)hipFree(c->d_xinv8);
  (void)hipFree(c->d_status);
  (void)hipFree(c->d_summary);
  (void)hipFree(c->d_pace);
  (void)hipFree(c->d_crc);
  (void)hipHostFree(c->h_status);
  if (c->h_walk) (void)hipHostFree(c->h_walk);
  (void)hipHostFree(c->h_summary);
  (void)hipHostFree(c->h_summary_init);
  (void)hipHostFree(c->h_crc);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
  if (c->ul_stream) (void)hipStreamDestroy(c->ul_stream);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int x3_ctx_sync(x3_ctx* c) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

extern "C" const char* x3_last_error(const x3_ctx* c) { return c ? c->last_error.c_str() : ""; }

extern "C" int x3_ctx_set_option(x3_ctx* c, const char* name, long long value) {
  if (!c || !name) return X3_ERR_BAD_ARG;
  const std::string n(name);
  if (n == "two_pass") c->opt.two_pass = value != 0;
  else if (n == "stream_wgs") { c->opt.stream_wgs = (int)std::max(0ll, value); c->stream_wg_per_cu = -1; }
  else if (n == "decode_single") c->opt.decode_single = value != 0;
  else if (n == "enc_gen") { c->opt.enc_gen = value == 2 ? 2 : 3; c->prefer_gen2 = false; }
  else if (n == "wave_nwg") c->opt.wave_nwg = (int)std::max(0ll, std::min(256ll, value));
  else if (n == "wave_m") c->opt.wave_m = (int)std::max(0ll, std::min(16ll, value));
  else if (n == "wave_drop") c->opt.wave_drop = value;
  else if (n == "host_walk") c->opt.host_walk = value < 0 ? -1 : (value != 0);
  else if (n == "host_chunk_frames") c->opt.host_chunk_frames = std::max(-1ll, value);
  else if (n == "verbose") c->opt.verbose = value != 0;
  else if (n == "file_chunk_frames") c->opt.file_chunk_frames = std::max(1ll, value);
  else if (n == "file_workers") c->opt.file_workers = (int)std::max(1ll, std::min(16ll, value));
  else if (n == "reader_window_frames") c->opt.reader_window_frames = std::max(1ll, value);
  else if (n == "check_main") c->opt.check_main = value != 0;
  else if (n == "check_first") c->opt.check_first = value != 0;
  else if (n == "check_wgs") c->opt.check_wgs = (int)std::max(1ll, value);
  else return X3_ERR_BAD_ARG;
  return X3_OK;
}

extern "C" int x3_ctx_get_option(const x3_ctx* c, const char* name, long long* value) {
  if (!c || !name || !value) return X3_ERR_BAD_ARG;
  const std::string n(name);
  if (n == "two_pass") *value = c->opt.two_pass;
  else if (n == "stream_wgs") *value = c->opt.stream_wgs;
  else if (n == "decode_single") *value = c->opt.decode_single;
  else if (n == "enc_gen") *value = c->opt.enc_gen;
  else if (n == "encode_dense_reruns") *value = 0;  // (rounds 2-3: whole calls encoded again for a dense frame; no longer happens)
  else if (n == "encode_dense_frames") *value = (long long)c->encode_dense_frames;  // read-only: frames the dense pass has written
  else if (n == "last_dense_frames") *value = (long long)c->last_dense_frames;      // read-only: of the last call (after x3_encode_result)
  else if (n == "enc_gen_in_use") *value = c->last_enc_gen;                          // read-only: 3, 2, 1, or 0 = two-pass kernels
  else if (n == "host_walk") *value = c->opt.host_walk;
  else if (n == "host_chunk_frames") *value = c->opt.host_chunk_frames;
  else if (n == "verbose") *value = c->opt.verbose;
  else if (n == "file_chunk_frames") *value = c->opt.file_chunk_frames;
  else if (n == "file_workers") *value = c->opt.file_workers;
  else if (n == "reader_window_frames") *value = c->opt.reader_window_frames;
  else if (n == "check_main") *value = c->opt.check_main;
  else if (n == "decode_pace" || n == "encode_pace") {  // (read-only, syncs) the pace words: 10 ns ticks per 16 blocks / per frame
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(w, c->d_pace, sizeof w, hipMemcpyDeviceToHost) != hipSuccess)
      return X3_ERR_HIP;
    // (the decoder keeps one word per launch parity: the newer one carries the larger epoch tag)
    *value = (long long)((n == "encode_pace" ? w[4] : std::max(w[0], w[1])) & 0xFFFFFu);
  }
  else if (n == "check_first") *value = c->opt.check_first;
  else if (n == "check_wgs") *value = c->opt.check_wgs;
  else if (n == "check_prio") *value = c->opt.check_prio;
  else if (n == "encode_fallbacks") *value = (long long)c->encode_fallbacks;  // read-only counter
  else if (n == "stream_wgs_in_use") *value = c->stream_wg_per_cu;            // read-only, -1 before the first launch
  else return X3_ERR_BAD_ARG;
  return X3_OK;
}

extern "C" const char* x3_strerror(int s) {
  static const char* names[] = {"Ok", "Io", "Hound", "BitPack", "InvalidEncodingThresh", "OutOfBoundsInverse",
                                "MoreThanOneChannel", "ArchiveHeaderXMLInvalid", "ArchiveHeaderXMLRiceCode",
                                "ArchiveHeaderXMLInvalidKey", "FrameLength", "FrameHeaderInvalidKey",
                                "FrameHeaderInvalidPayloadLen", "FrameHeaderInvalidHeaderCRC",
                                "FrameHeaderInvalidPayloadCRC", "FrameDecodeInvalidBlockLength",
                                "FrameDecodeInvalidIndex", "FrameDecodeInvalidNTOGO", "FrameDecodeInvalidFType",
                                "FrameDecodeInvalidRiceCode", "FrameDecodeInvalidBPF", "FrameDecodeUnexpectedEnd",
                                "ByteWriterInsufficientMemory", "Hip", "BadArg"};
  return (s >= 0 && s <= 24) ? names[s] : "Unknown";
}

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
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  TimerScope(x3_ctx* c_, int w, hipStream_t s_ = nullptr, bool attached_ = false)
      : c(c_), which(w), st(s_ ? s_ : c_->stream), attached(attached_) {
    if (!c->timing) return;
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
    if (!c->timing) return;
    if (!attached) (void)hipEventRecord(ev.second, st);
    c->timers[which].used.push_back(ev);
  }
};
// launch `kernel` on `stream` inside the TimerScope `ts` (constructed with attached = true)
#define X3_LAUNCH_TIMED(ts, kernel, grid, block, smem, stream, ...)                                        \
  do {                                                                                                     \
    if ((ts).c->timing) hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, (ts).ev.first, (ts).ev.second, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                               \
  } while (0)

extern "C" int x3_ctx_enable_kernel_timing(x3_ctx* c, int enable) {
  if (!c) return X3_ERR_BAD_ARG;
  c->timing = enable != 0;
  return X3_OK;
}

extern "C" int x3_ctx_reset_kernel_time(x3_ctx* c) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& t : c->timers) {
    for (auto& e : t.used) t.pool.push_back(e);
    t.used.clear();
  }
  return X3_OK;
}

// every timed launch's own time, oldest first (bench.py: minimum, median, p90 of a kernel over the timed steps)
extern "C" int x3_ctx_kernel_times(x3_ctx* c, int which, double* ms, uint64_t cap, uint64_t* launches) {
  if (!c || which < 0 || which > 5 || (!ms && cap)) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  uint64_t k = 0;
  for (auto& e : c->timers[which].used) {
    float t = 0;
    HIPCHK(c, hipEventElapsedTime(&t, e.first, e.second));
    if (k < cap) ms[k] = t;
    ++k;
  }
  if (launches) *launches = k;
  return X3_OK;
}

// The launch log: the last X3_LOG_ENTRIES launches of the decoder (which = 1) or the wave encoder (which = 0), newest
// last.  Per launch four values: ticks of 10 ns per 16 blocks the launch aimed at and its slowest group achieved
// (decoder; 0 for the encoder), and the shader clock in kHz that workgroup 0 measured over its life.  Syncs.
extern "C" int x3_ctx_launch_log(x3_ctx* c, int which, uint32_t* out, uint64_t cap_entries, uint64_t* n_entries) {
  if (!c || (which != 0 && which != 1) || (!out && cap_entries)) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<uint32_t> w(X3_LOG_ENTRIES * X3_LOG_WORDS);
  HIPCHK(c, hipMemcpy(w.data(), c->d_pace + (which ? X3_LOG_BASE : X3_LOG_ENC_BASE), w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  // the newest launch: the decoder's epoch counter has been incremented behind its last launch
  const uint32_t last = which ? (c->dec_epoch - 1u) & 0xFFFu : c->enc_log_epoch & 0xFFFu;
  uint64_t n = 0;
  for (uint32_t back = X3_LOG_ENTRIES; back-- > 0;) {
    const uint32_t ep = (last - back) & 0xFFFu;
    const uint32_t* e = &w[(size_t)(ep & (X3_LOG_ENTRIES - 1u)) * X3_LOG_WORDS];
    if ((e[0] >> 20) != ep || e[3] == 0) continue;   // (not this epoch's entry: never written, or older)
    if (n < cap_entries) {
      out[4 * n + 0] = which ? e[1] & 0xFFFFFu : 0u;
      out[4 * n + 1] = which ? e[0] & 0xFFFFFu : 0u;
      out[4 * n + 2] = (uint32_t)((unsigned long long)e[2] * 100000ull / e[3]);   // shader ticks per 10 ns tick -> kHz
      out[4 * n + 3] = e[3];
    }
    ++n;
  }
  if (n_entries) *n_entries = n;
  return X3_OK;
}

extern "C" int x3_ctx_kernel_time(x3_ctx* c, int which, double* total_ms, uint64_t* launches) {
  if (!c || which < 0 || which > 5) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double tot = 0;
  for (auto& e : c->timers[which].used) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, e.first, e.second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = c->timers[which].used.size();
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// x3.rs: parameters
// ------------------------------------------------------------------------------------------------
static const uint32_t RICE_OFFSET[4] = {6, 11, 20, 28};   // src/x3.rs:209,216,223,236
static const uint32_t RICE_LEN[4] = {14, 22, 40, 56};     // table lengths, src/x3.rs:210-249
static const uint32_t RICE_INV_LEN[4] = {16, 26, 44, 60}; // src/x3.rs:213,220,233,250

extern "C" void x3_params_default(x3_params* p) {
  if (!p) return;
  p->block_len = 20;
  p->blocks_per_frame = 500;
  p->codes[0] = 0; p->codes[1] = 1; p->codes[2] = 3;
  p->thresholds[0] = 3; p->thresholds[1] = 8; p->thresholds[2] = 20;
}

extern "C" int x3_params_validate(const x3_params* p) {
  if (!p) return X3_ERR_BAD_ARG;
  for (int k = 0; k < 3; ++k)
    if (p->codes[k] > 3) return X3_ERR_BAD_ARG;  // RiceCodes::get indexes CODE[4] -> panic
  for (int k = 0; k < 2; ++k)                    // only k = 0,1 are checked (x3.rs:107-112)
    if (p->thresholds[k] > RICE_OFFSET[p->codes[k]]) return X3_ERR_INVALID_ENCODING_THRESH;
  return X3_OK;
}

// RiceCodes::CODE (src/x3.rs:206-252) as arithmetic: entry i stands for the difference d = i - offset, folded to
// u = 2d (d >= 0) or -2d - 1 (d < 0); the codeword is u >> k zeros, a one, then the k low bits of u.
extern "C" int x3_rice_code_get(uint32_t code_number, x3_rice_code* out) {
  struct Tables {
    uint32_t code[4][56], num_bits[4][56];
    int16_t inv[60];
    Tables() {
      for (uint32_t k = 0; k < 4; ++k)
        for (uint32_t i = 0; i < RICE_LEN[k]; ++i) {
          const int32_t d = (int32_t)i - (int32_t)RICE_OFFSET[k];
          const uint32_t u = d >= 0 ? 2u * (uint32_t)d : 2u * (uint32_t)(-d) - 1u;
          code[k][i] = (1u << k) | (u & ((1u << k) - 1u));
          num_bits[k][i] = (u >> k) + 1u + k;
        }
      for (uint32_t i = 0; i < 60; ++i) inv[i] = (i & 1u) ? (int16_t)-(int32_t)((i + 1u) >> 1) : (int16_t)(i >> 1);
    }
  };
  static const Tables t;
  if (!out || code_number > 3) return X3_ERR_BAD_ARG;
  out->nsubs = code_number;
  out->offset = RICE_OFFSET[code_number];
  out->len = RICE_LEN[code_number];
  out->inv_len = RICE_INV_LEN[code_number];
  out->code = t.code[code_number];
  out->num_bits = t.num_bits[code_number];
  out->inv = t.inv;
  return X3_OK;
}

static uint64_t spf_of(const x3_params* p) { return (uint64_t)p->block_len * (uint64_t)p->blocks_per_frame; }

// worst-case payload bytes of a frame of n samples: every block literal
static uint64_t max_payload_bytes(uint64_t n, uint32_t block_len) {
  if (n == 0) return 0;
  uint64_t nblocks = block_len ? (n - 1 + block_len - 1) / block_len : 0;
  uint64_t bits = 16 + nblocks * 6 + 16 * (n - 1);
  return (((bits + 7) >> 3) + 1) & ~1ull;
}

extern "C" uint64_t x3_num_frames(uint64_t n, const x3_params* p) {
  uint64_t spf = p ? spf_of(p) : 0;
  return spf ? (n + spf - 1) / spf : 0;
}

extern "C" uint64_t x3_encode_bound(uint64_t n, const x3_params* p) {
  if (!p) return 0;
  uint64_t spf = spf_of(p);
  if (!spf || !n) return 1;
  uint64_t full = n / spf, tail = n % spf;
  return full * (20 + max_payload_bytes(spf, p->block_len)) + (tail ? 20 + max_payload_bytes(tail, p->block_len) : 0) + 1;
}

static int derive(const x3_params* p, uint64_t spf, X3DevParams* d) {
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;  // a threshold violation is only an error in Parameters::new
  for (int k = 0; k < 3; ++k)
    if (p->thresholds[k] > 0x7FFFFFFFu) return X3_ERR_BAD_ARG;
  d->block_len = p->block_len;
  d->blocks_per_frame = p->blocks_per_frame;
  d->spf = (uint32_t)spf;
  for (int k = 0; k < 3; ++k) {
    uint32_t c = p->codes[k];
    d->thr[k] = p->thresholds[k];
    d->k[k] = c;
    d->dmin[k] = -(int32_t)RICE_OFFSET[c];
    d->dmax[k] = (int32_t)RICE_LEN[c] - (int32_t)RICE_OFFSET[c] - 1;
    d->inv_len[k] = RICE_INV_LEN[c];
  }
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// crc.rs / frame header helpers (20-byte host arithmetic)
// ------------------------------------------------------------------------------------------------
extern "C" uint16_t x3_crc16_update(uint16_t crc, uint8_t byte) {
  uint32_t t = ((crc >> 8) ^ byte) & 0xFFu;
  t ^= t >> 4;
  return (uint16_t)(((uint32_t)crc << 8) ^ (t << 12) ^ (t << 5) ^ t);
}

static uint16_t header_crc16_host(const uint8_t* b, size_t n) {  // headers only (16 bytes)
  uint16_t crc = 0xFFFF;
  for (size_t i = 0; i < n; ++i) crc = x3_crc16_update(crc, b[i]);
  return crc;
}

extern "C" void x3_write_frame_header(uint64_t num_samples, uint8_t id, uint64_t payload_len, uint16_t payload_crc,
                                      uint8_t out[X3_FRAME_HEADER_LENGTH]) {
  std::memset(out, 0, 20);
  out[0] = 0x78; out[1] = 0x33;
  out[2] = id;
  out[3] = id;  // the reference writes `id` here too (encoder.rs:135)
  out[4] = (uint8_t)(num_samples >> 8); out[5] = (uint8_t)num_samples;
  out[6] = (uint8_t)(payload_len >> 8); out[7] = (uint8_t)payload_len;
  uint16_t hc = header_crc16_host(out, 16);
  out[16] = (uint8_t)(hc >> 8); out[17] = (uint8_t)hc;
  out[18] = (uint8_t)(payload_crc >> 8); out[19] = (uint8_t)payload_crc;
}

// n_ch == 1: the reference's test (a channel count above one is refused, decoder.rs:90-94); n_ch > 1 (the multi-channel
// extension): the frame must say exactly n_ch
static int read_frame_header_ch(const uint8_t* b, uint64_t len, x3_frame_header* h, uint32_t n_ch);
extern "C" int x3_read_frame_header(const uint8_t* b, uint64_t len, x3_frame_header* h) {
  return read_frame_header_ch(b, len, h, 1u);
}
static int read_frame_header_ch(const uint8_t* b, uint64_t len, x3_frame_header* h, uint32_t n_ch) {
  if (!b || !h) return X3_ERR_BAD_ARG;
  if (len < 20) return X3_ERR_FRAME_DECODE_UNEXPECTED_END;
  if ((((uint16_t)b[16] << 8) | b[17]) != header_crc16_host(b, 16)) return X3_ERR_FRAME_HEADER_INVALID_HEADER_CRC;
  if (b[0] != 0x78 || b[1] != 0x33) return X3_ERR_FRAME_HEADER_INVALID_KEY;
  if (n_ch == 1u ? b[3] > 1 : b[3] != n_ch) return X3_ERR_MORE_THAN_ONE_CHANNEL;
  uint32_t plen = ((uint32_t)b[6] << 8) | b[7];
  if (plen >= X3_FRAME_MAX_LENGTH) return X3_ERR_FRAME_LENGTH;
  h->source_id = b[2];
  h->channels = b[3];
  h->samples = (uint16_t)(((uint16_t)b[4] << 8) | b[5]);
  h->payload_len = plen;
  h->payload_crc = (uint16_t)(((uint16_t)b[18] << 8) | b[19]);
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// buffer CRC on the GPU
// ------------------------------------------------------------------------------------------------
static int crc_dev_async(x3_ctx* c, const uint8_t* d_data, uint64_t n) {
  if (reinterpret_cast<uintptr_t>(d_data) & 3u) return X3_ERR_BAD_ARG;
  const uint64_t n_dw = n >> 2;
  const uint64_t n_seg = (n_dw + X3_CRC_SEG_DW - 1) / X3_CRC_SEG_DW;
  if (n_seg > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  int rc = ensure(c, c->seg_crc, (n_seg + 1) * sizeof(uint16_t));
  if (rc) return rc;
  if (n_seg)
    hipLaunchKernelGGL(x3_crc_segments_kernel, dim3((unsigned)n_seg), dim3(64), 0, c->stream,
                       reinterpret_cast<const uint32_t*>(d_data), n_dw, n_seg, c->d_xpow, (uint16_t*)c->seg_crc.p);
  hipLaunchKernelGGL(x3_crc_combine_kernel, dim3(1), dim3(64), 0, c->stream, d_data, n, n_seg,
                     (const uint16_t*)c->seg_crc.p, c->d_xpow, c->d_crc);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_crc, c->d_crc, sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
  return X3_OK;
}

extern "C" int x3_crc16_dev(x3_ctx* c, const uint8_t* d_data, uint64_t n, uint16_t* crc) {
  if (!c || !crc || (!d_data && n)) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = crc_dev_async(c, d_data, n);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *crc = *c->h_crc;
  return X3_OK;
}

extern "C" int x3_crc16(x3_ctx* c, const uint8_t* data, uint64_t n, uint16_t* crc) {
  if (!c || !crc || (!data && n)) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->in, n + 16);
  if (rc) return rc;
  if (n) HIPCHK(c, hipMemcpyAsync(c->in.p, data, n, hipMemcpyHostToDevice, c->stream));
  return x3_crc16_dev(c, (const uint8_t*)c->in.p, n, crc);
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// x3_encode_stream2_kernel has no "diff outside the reference's Rice table" test (the reference panics there:
// X3_ERR_BAD_ARG from the two-pass kernels): true when no block can need it.  A block with max|d| = m <= thr[2] is
// coded with code[ft(m)], ft = [m > thr0] + [m > thr1] (encoder.rs:241-247); it is inside that code's table when
// m <= min(offset, len - offset - 1).
static bool stream_safe_thresholds(const x3_params* p) {
  uint32_t mmax[3] = {0, 0, 0};
  bool used[3] = {false, false, false};
  const uint32_t top = std::min<uint32_t>(p->thresholds[2], 70000u);
  for (uint32_t m = 0; m <= top; ++m) {
    const uint32_t ft = (m > p->thresholds[0] ? 1u : 0u) + (m > p->thresholds[1] ? 1u : 0u);
    used[ft] = true;
    mmax[ft] = m;
  }
  for (int ft = 0; ft < 3; ++ft) {
    if (!used[ft]) continue;
    const uint32_t cde = p->codes[ft];
    if (cde > 3) return false;
    const uint32_t inside = std::min(RICE_OFFSET[cde], RICE_LEN[cde] - RICE_OFFSET[cde] - 1u);
    if (mmax[ft] > inside) return false;
  }
  return true;
}

struct EncPlan {
  X3DevParams dp;
  X3Geom g;
  uint32_t nthr, lds_in_bytes, img_dwords;
  size_t smem;
};

static int plan_encode(x3_ctx* c, const x3_batch* b, const x3_params* p, uint64_t spf, EncPlan* pl) {
  int rc = derive(p, spf, &pl->dp);
  if (rc) return rc;
  if (spf == 0 || spf > 0xFFFFFFFFull || p->block_len == 0) return X3_ERR_BAD_ARG;
  const uint64_t fpc = (b->n_per_clip + spf - 1) / spf;
  if (fpc == 0 || fpc > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
  const uint64_t F = fpc * b->n_clips;
  if (F == 0 || F > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  pl->g.n_per_clip = b->n_per_clip;
  pl->g.clip_stride = b->clip_stride;
  pl->g.fpc = (uint32_t)fpc;
  pl->g.n_frames = F;
  const uint64_t nmax = std::min<uint64_t>(spf, b->n_per_clip);   // samples in the largest frame
  // a block longer than MAX_BLOCK_LENGTH = 60 samples overruns the reference's diff array (encoder.rs:296-299)
  if (std::min<uint64_t>(p->block_len, nmax - 1) > 60) return X3_ERR_BAD_ARG;
  const uint64_t nblocks = (nmax - 1 + p->block_len - 1) / p->block_len;
  uint32_t nthr = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(64, (nblocks + 63) & ~63ull));
  pl->nthr = nthr;
  const uint64_t in_bytes = (2 * nmax + 4 + 15) & ~15ull;  // + the dword read behind the last sample
  const uint64_t img_dw = ((5 + (max_payload_bytes(nmax, p->block_len) + 3) / 4 + 4) + 3) & ~3ull;
  const uint64_t smem = X3_ENC_SMEM_HDR + in_bytes + img_dw * 4;
  if (smem > 160 * 1024) {
    c->last_error = "frame too large for the LDS-resident encoder (block_len*blocks_per_frame)";
    return X3_ERR_BAD_ARG;
  }
  pl->lds_in_bytes = (uint32_t)in_bytes;
  pl->img_dwords = (uint32_t)img_dw;
  pl->smem = (size_t)smem;
  return X3_OK;
}

static int encode_dev_impl(x3_ctx* c, const int16_t* d_wav, const x3_batch* b, const x3_params* p, uint64_t spf,
                           uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets) {
  EncPlan pl;
  int rc = plan_encode(c, b, p, spf, &pl);
  if (rc) return rc;
  if (reinterpret_cast<uintptr_t>(d_out) & 1u) return X3_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d_wav) & 1u) return X3_ERR_BAD_ARG;
  const uint64_t F = pl.g.n_frames;
  if ((rc = ensure(c, c->frame_bytes, F * sizeof(uint32_t)))) return rc;
  uint64_t* d_off = d_frame_offsets;
  if (!d_off) {
    if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
    d_off = (uint64_t*)c->frame_off.p;
  }
  HIPCHK(c, hipMemsetAsync(c->d_status, 0, 128, c->stream));
  // ---- single-pass path: default block length, 16-byte aligned frames (see x3_encode_stream2_kernel.h)
  const bool stream_path = p->block_len == 20 && pl.nthr == 512 && (std::min<uint64_t>(spf, b->n_per_clip) + 18) / 20 <= 512 &&
                           (spf % 8) == 0 &&
                           (b->n_clips == 1 || (b->clip_stride % 8) == 0) &&
                           (reinterpret_cast<uintptr_t>(d_wav) & 15u) == 0 && !c->force_two_pass && !c->opt.two_pass;
  c->last_enc = {d_wav, *b, *p, spf, d_out, out_cap, start_pos, d_frame_offsets};
  // part + two worst-case frame images + CRC tables + the multipliers of one chunk size (x3_encode_stream2_kernel.h)
  const size_t smem2 = X3_ENC_SMEM_HDR + 2 * (size_t)pl.img_dwords * 4 + 2048 + X3_K2_DWORDS * 4;
  c->last_enc_gen = 0;
  if (stream_path && c->opt.enc_gen == 3 && c->opt.stream_wgs == 0 &&
      stream_safe_thresholds(p) && smem2 <= 160 * 1024) {
    if (c->prefer_gen2) {
      // the last call's content was mostly dense: the second generation, until it counts few dense frames again
    } else {
      // third generation (x3_encode_wave_kernel.h): one wave per frame, sixteen waves per CU, one workgroup per CU
      static_assert(X3W_SMEM <= 160 * 1024, "LDS");
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_wave_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)X3W_SMEM));
      uint64_t nwg_max = std::min<uint64_t>((uint64_t)c->n_cus, X3W_MAX_NWG);
      if (c->opt.wave_nwg > 0) nwg_max = std::min<uint64_t>(nwg_max, (uint64_t)c->opt.wave_nwg);
      X3WaveArgs wa;
      wa.m = (uint32_t)std::min<uint64_t>(X3W_WAVES, (F + nwg_max - 1) / nwg_max);
      if (c->opt.wave_m > 0) wa.m = (uint32_t)c->opt.wave_m;
      const uint64_t n_wggen = (F + wa.m - 1) / wa.m;
      wa.nwg = (uint32_t)std::min<uint64_t>(nwg_max, n_wggen);
      wa.n_wggen = (uint32_t)n_wggen;
      const uint64_t step = (uint64_t)wa.nwg * wa.m;
      wa.step_clip = (uint32_t)(step / pl.g.fpc);
      wa.step_idx = (uint32_t)(step % pl.g.fpc);
      const size_t desc_bytes = (n_wggen + X3W_DESC_PAD) * sizeof(uint32_t);
      const bool fresh = c->desc.cap < desc_bytes;
      if ((rc = ensure(c, c->desc, desc_bytes))) return rc;
      if (fresh || ++c->desc_epoch > 0xFFFu) {
        HIPCHK(c, hipMemsetAsync(c->desc.p, 0, c->desc.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + 4, 0, 16, c->stream));
        c->desc_epoch = 1;
      }
      wa.wav = d_wav;
      wa.out = d_out;
      wa.frame_off = d_off;
      wa.desc = (uint32_t*)c->desc.p + X3W_DESC_PAD;
      wa.ctl = reinterpret_cast<unsigned char*>(c->d_status);
      wa.tabs = c->d_wtab;
      wa.log = c->d_pace + X3_LOG_ENC_BASE;
      wa.log_epoch = ++c->enc_log_epoch & 0xFFFu;
      if ((rc = ensure(c, c->dense_list, F * sizeof(uint32_t)))) return rc;
      wa.dense_list = (uint32_t*)c->dense_list.p;
      wa.out_cap = out_cap;
      wa.start_pos = start_pos;
      wa.n_per_clip = pl.g.n_per_clip;
      wa.clip_stride = pl.g.clip_stride;
      wa.n_frames = F;
      wa.fpc = pl.g.fpc;
      wa.spf = pl.dp.spf;
      wa.epoch = c->desc_epoch;
      wa.thr0 = pl.dp.thr[0];
      wa.thr1 = pl.dp.thr[1];
      wa.thr2 = pl.dp.thr[2];
      wa.kpack = pl.dp.k[0] | (pl.dp.k[1] << 8) | (pl.dp.k[2] << 16);
      wa.drop_wgi = c->opt.wave_drop >= 0 ? (uint32_t)c->opt.wave_drop : 0xFFFFFFFFu;
      {
        TimerScope ts(c, 0, nullptr, true);
        X3_LAUNCH_TIMED(ts, x3_encode_wave_kernel, dim3(wa.nwg), dim3(X3W_THREADS), X3W_SMEM, c->stream, wa);
      }
      {
        // The dense pass, always: the frames the wave kernel listed (none, in most recordings: the workgroups read a zero
        // count and leave, ~2 us of queue) written at the offsets it assigned.  In the stream, not in x3_encode_result:
        // whatever the caller enqueues behind this call -- x3_decode_dev, a copy -- finds the whole stream.
        if (smem2 > 64 * 1024)
          HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_stream2_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
        const uint64_t per_cu = std::max<uint64_t>(1, (160 * 1024) / (smem2 + 256));
        const uint64_t grid = std::min<uint64_t>(F, (uint64_t)c->n_cus * std::min<uint64_t>(per_cu, 3));
        TimerScope ts(c, 5, nullptr, true);
        X3_LAUNCH_TIMED(ts, x3_encode_stream2_kernel<true>, dim3((unsigned)grid), dim3(X3_STREAM2_THREADS), smem2, c->stream,
                        d_wav, pl.g, pl.dp, d_off, d_out, out_cap, start_pos, (uint32_t*)nullptr, 0u,
                        reinterpret_cast<unsigned char*>(c->d_status), (const uint32_t*)c->d_xk2,
                        (const uint16_t*)c->d_crctab, pl.img_dwords, (uint32_t*)nullptr, (const uint32_t*)c->dense_list.p);
      }
      HIPCHK(c, hipGetLastError());
      c->last_enc_gen = 3;
      c->encode_pending = true;
      c->enc_start_pos = start_pos;
      return X3_OK;
    }
  }
  if (stream_path && stream_safe_thresholds(p)) {
    // second generation (x3_encode_stream2_kernel.h): eight waves, no sample tile in LDS
    if (c->stream_wg_per_cu < 0) {
      // Offsets wait on the other workgroups' frame sizes, so EVERY workgroup of the grid must be resident.  The
      // occupancy API can over-report by one block per CU (MI355X_MICROARCH.md, "Residency"), so it is capped
      // by the kernel's own register/LDS footprint: eight waves are two per SIMD, whatever the placement.
      if (smem2 > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_stream2_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
      int nb = 0;
      HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, x3_encode_stream2_kernel<false>, X3_STREAM2_THREADS, smem2));
      hipFuncAttributes fa;
      HIPCHK(c, hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&x3_encode_stream2_kernel<false>)));
      const int alloc = ((fa.numRegs + 7) / 8) * 8;
      const int wps = std::min(8, 512 / std::max(alloc, 8));
      const int by_regs = (4 * wps) / 8;
      const int by_lds = (int)((160 * 1024) / (smem2 + 256));  // (+ allocation granularity)
      c->stream_wg_per_cu = std::max(0, std::min(std::min(nb, 4), std::min(by_regs, by_lds)));
      // experiments and the fallback test: force a grid (one that is too large cannot be resident: the size
      // waits time out and x3_encode_result re-encodes with the two-pass kernels)
      if (c->opt.stream_wgs > 0) c->stream_wg_per_cu = c->opt.stream_wgs;
      if (c->opt.verbose)
        std::fprintf(stderr, "x3hip: stream encoder v2 %d VGPRs, %zu B LDS, occupancy API %d, by_regs %d, by_lds %d -> %d workgroups/CU\n",
                     fa.numRegs, smem2, nb, by_regs, by_lds, c->stream_wg_per_cu);
    }
    if (c->stream_wg_per_cu >= 1 && smem2 <= 160 * 1024) {
      // frame-size descriptors {epoch:12 | bytes:20}: the epoch makes last launch's words "not ready"
      // without clearing the array (cleared when it is (re)allocated and when the epoch wraps)
      const uint64_t grid = std::min<uint64_t>(std::min<uint64_t>(F, X3_STREAM2_MAX_GRID), (uint64_t)c->n_cus * c->stream_wg_per_cu);
      const size_t desc_pad = 1024 + 64;  // words in front of desc[0]: the windows of the first frames reach below frame 0
      const size_t desc_bytes = (F + desc_pad) * sizeof(uint32_t);
      const bool fresh = c->desc.cap < desc_bytes;
      if ((rc = ensure(c, c->desc, desc_bytes))) return rc;
      if (fresh || ++c->desc_epoch > 0xFFFu) {
        HIPCHK(c, hipMemsetAsync(c->desc.p, 0, c->desc.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_pace + 4, 0, 16, c->stream));  // (the encoder's pace words carry the same epoch)
        c->desc_epoch = 1;
      }
      {
        TimerScope ts(c, 0);
        hipLaunchKernelGGL(x3_encode_stream2_kernel<false>, dim3((unsigned)grid), dim3(X3_STREAM2_THREADS), smem2, c->stream,
                           d_wav, pl.g, pl.dp, d_off, d_out, out_cap, start_pos, (uint32_t*)c->desc.p + desc_pad, c->desc_epoch,
                           reinterpret_cast<unsigned char*>(c->d_status), (const uint32_t*)c->d_xk2,
                           (const uint16_t*)c->d_crctab, pl.img_dwords, c->d_pace + 4, (const uint32_t*)nullptr);
      }
      HIPCHK(c, hipGetLastError());
      c->last_enc_gen = 2;
      c->encode_pending = true;
      c->enc_start_pos = start_pos;
      return X3_OK;
    }
  }
  if (pl.smem > 64 * 1024) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&x3_encode_frames_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
  }
  {
    TimerScope ts(c, 2);
    hipLaunchKernelGGL(x3_encode_frames_kernel<true>, dim3((unsigned)F), dim3(pl.nthr),
                       X3_ENC_SMEM_HDR + pl.lds_in_bytes, c->stream, d_wav, pl.g, pl.dp, (const uint64_t*)nullptr,
                       (uint32_t*)c->frame_bytes.p, (uint8_t*)nullptr, start_pos, c->d_stats, c->d_status,
                       (const uint16_t*)c->d_xpow, pl.lds_in_bytes, 0u, 1u, (uint64_t)0);
  }
  {
    TimerScope ts(c, 3);
    hipLaunchKernelGGL(x3_scan_frame_offsets_kernel, dim3(1), dim3(1024), 0, c->stream,
                       (const uint32_t*)c->frame_bytes.p, F, start_pos, out_cap, d_off, c->d_end_pos, c->d_status);
  }
  {
    TimerScope ts(c, 0);
    hipLaunchKernelGGL(x3_encode_frames_kernel<false>, dim3((unsigned)F), dim3(pl.nthr), pl.smem, c->stream, d_wav,
                       pl.g, pl.dp, (const uint64_t*)d_off, (uint32_t*)nullptr, d_out, start_pos, c->d_stats,
                       c->d_status, (const uint16_t*)c->d_xpow, pl.lds_in_bytes, pl.img_dwords, 1u, (uint64_t)0);
  }
  HIPCHK(c, hipGetLastError());
  c->encode_pending = true;
  c->enc_start_pos = start_pos;
  return X3_OK;
}

extern "C" int x3_encode_dev(x3_ctx* c, const int16_t* d_wav, const x3_batch* batch, const x3_params* p,
                             uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets) {
  if (!c || !d_wav || !batch || !p || !d_out) return X3_ERR_BAD_ARG;
  if (batch->n_per_clip == 0 || batch->n_clips == 0) return X3_ERR_BAD_ARG;
  if (batch->n_clips > 1 && batch->clip_stride < batch->n_per_clip) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return encode_dev_impl(c, d_wav, batch, p, spf_of(p), d_out, out_cap, start_pos, d_frame_offsets);
}

static int encode_dev_impl(x3_ctx* c, const int16_t* d_wav, const x3_batch* b, const x3_params* p, uint64_t spf,
                           uint8_t* d_out, uint64_t out_cap, uint64_t start_pos, uint64_t* d_frame_offsets);

extern "C" int x3_encode_result(x3_ctx* c, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c) return X3_ERR_BAD_ARG;
  if (!c->encode_pending) return X3_ERR_BAD_ARG;
  // status and statistics are fetched here, not behind every launch: a small copy is a packet of its own in
  // the queue (~8 us), and a pipeline that launches encode and decode back to back asks once per batch
  HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status, 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->encode_pending = false;
  if (c->h_status[1] != X3D_SIZE_WAIT_TIMEOUT && (c->last_enc_gen == 3 || c->last_enc_gen == 2)) {
    // how dense the call's content was (both single-pass generations count the frames beyond the wave encoder's image):
    // the hint for the context's next call, nothing else
    uint32_t dense;
    std::memcpy(&dense, reinterpret_cast<const unsigned char*>(c->h_status) + X3_CTL_DENSE_COUNT, sizeof dense);
    const uint64_t frames = ((c->last_enc.b.n_per_clip + c->last_enc.spf - 1) / c->last_enc.spf) * c->last_enc.b.n_clips;
    c->last_dense_frames = dense;
    if (c->last_enc_gen == 3) {
      c->encode_dense_frames += dense;
      if ((uint64_t)dense * 4u > frames) c->prefer_gen2 = true;
    } else if (c->prefer_gen2 && (uint64_t)dense * 8u <= frames) {
      c->prefer_gen2 = false;
    }
  }
  if (c->h_status[1] == X3D_SIZE_WAIT_TIMEOUT) {
    if (c->opt.verbose)
      std::fprintf(stderr, "x3hip: size wait gave up: kind %d generation %d wave %d gen %d masks %08x %08x %06x\n", c->h_status[2],
                   c->h_status[3], c->h_status[4] & 0xFF, c->h_status[4] >> 8, c->h_status[5], c->h_status[6], c->h_status[7]);
    // the single-pass kernel's workgroups were not all resident (GPU shared with other work): its
    // bounded wait for frame sizes gave up.  Encode again with the two-pass kernels, which need no residency.
    ++c->encode_fallbacks;
    if (c->opt.verbose)
      std::fprintf(stderr, "x3hip: stream encoder gave up waiting for frame sizes (grid not co-resident): two-pass fallback\n");
    c->force_two_pass = true;
    auto a = c->last_enc;
    int rc = encode_dev_impl(c, a.d_wav, &a.b, &a.p, a.spf, a.d_out, a.out_cap, a.start_pos, a.d_off);
    c->force_two_pass = false;
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status, 128, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->encode_pending = false;
  }
  if (out_pos) *out_pos = c->h_stats[6];
  if (stats)
    for (int i = 0; i < 6; ++i) stats[i] = c->h_stats[i];
  return std::max(c->h_status[0], c->h_status[1]);
}

// host-buffer front end shared by x3_encode / x3_encode_frame / x3_encode_batch
static int encode_host(x3_ctx* c, const int16_t* const* wavs, uint64_t n_per_clip, uint64_t n_clips,
                       const x3_params* p, uint64_t spf, uint8_t* out, uint64_t out_cap, uint64_t start_pos,
                       uint64_t* out_pos, uint64_t* clip_offsets, uint64_t stats[6]) {
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t total = n_per_clip * n_clips;
  int rc = ensure(c, c->in, total * sizeof(int16_t) + 16);
  if (rc) return rc;
  for (uint64_t k = 0; k < n_clips; ++k)
    HIPCHK(c, hipMemcpyAsync((int16_t*)c->in.p + k * n_per_clip, wavs[k], n_per_clip * sizeof(int16_t),
                             hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  x3_batch b{n_per_clip, n_per_clip, n_clips};
  uint64_t bound;
  {
    uint64_t full = n_per_clip / spf, tail = n_per_clip % spf;
    bound = n_clips * (full * (20 + max_payload_bytes(spf, p->block_len)) +
                       (tail ? 20 + max_payload_bytes(tail, p->block_len) : 0));
  }
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  const uint64_t dev_cap = std::min<uint64_t>(out_cap, start_pos + 1 + bound);
  if ((rc = ensure(c, c->out, dev_cap + 16))) return rc;
  uint64_t pos = 0;
  {
    std::unique_lock<std::mutex> gate;
    if (c->enc_gate) {
      HIPCHK(c, hipStreamSynchronize(c->stream));  // the upload is not the gate's business
      gate = std::unique_lock<std::mutex>(*c->enc_gate);
    }
    if ((rc = encode_dev_impl(c, (const int16_t*)c->in.p, &b, &pp, spf, (uint8_t*)c->out.p, out_cap, start_pos, nullptr)))
      return rc;
    rc = x3_encode_result(c, &pos, stats);
  }
  if (out_pos) *out_pos = pos;
  if (rc) return rc;
  if (pos > start_pos)
    HIPCHK(c, hipMemcpyAsync(out + start_pos, (uint8_t*)c->out.p + start_pos, pos - start_pos, hipMemcpyDeviceToHost,
                             c->stream));
  if (clip_offsets) {
    const uint64_t fpc = (n_per_clip + spf - 1) / spf;
    std::vector<uint64_t> offs(fpc * n_clips + 1);
    HIPCHK(c, hipMemcpyAsync(offs.data(), c->frame_off.p, offs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t k = 0; k <= n_clips; ++k) clip_offsets[k] = offs[k * fpc];
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

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
static int x3_pipe_streams(x3_ctx* c) {
  if (!c->dl_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
  if (!c->ul_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->ul_stream, hipStreamNonBlocking));
  return X3_OK;
}

// x3_encode on a LONG host buffer: the same bytes as encode_host, in chunks of whole frames.  The link to the host is
// the whole cost of this entry point (config 3: 1.38 GB up, 0.36 GB down, 0.4 ms of kernel) and it carries both
// directions at once (tools/ubench/pcie_duplex.hip: 24.7 ms for both against 31.0 one after the other), so three host
// threads work side by side -- pageable copies hold their caller: one sends chunk i+2 up, this one encodes chunk i+1
// where chunk i ended (frames do not depend on each other; x3_encode_result's position is all a chunk waits for),
// one brings the bytes of chunk i down.
static int encode_host_chunked(x3_ctx* c, const int16_t* wav, uint64_t n, const x3_params* p, uint64_t spf,
                               uint64_t chunk, uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos,
                               uint64_t stats[6]) {
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->in, n * sizeof(int16_t) + 16);
  if (rc) return rc;
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  {
    const uint64_t full = n / spf, tail = n % spf;
    const uint64_t bound = full * (20 + max_payload_bytes(spf, p->block_len)) + (tail ? 20 + max_payload_bytes(tail, p->block_len) : 0);
    if ((rc = ensure(c, c->out, std::min<uint64_t>(out_cap, start_pos + 1 + bound) + 16))) return rc;
  }
  if ((rc = x3_pipe_streams(c))) return rc;
  struct Piece { uint64_t lo, hi; };
  X3Handoff<Piece> down;
  X3Progress up;
  hipError_t up_err = hipSuccess, dl_err = hipSuccess;
  int16_t* d_in = (int16_t*)c->in.p;
  std::thread uploader, downloader;
  try {
  // (the helper threads never let an exception out -- that would be std::terminate, and x3hip.h promises that the library
  // does not abort: whatever is thrown in them ends the pipeline with an error -- ADVICE r3)
  uploader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      for (uint64_t s0 = 0; s0 < n && e == hipSuccess; s0 += chunk) {
        { std::lock_guard<std::mutex> g(up.mu); if (up.stop) return; }
        e = hipMemcpyAsync(d_in + s0, wav + s0, std::min<uint64_t>(chunk, n - s0) * sizeof(int16_t), hipMemcpyHostToDevice, c->ul_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->ul_stream);
        if (e == hipSuccess) up.advance();
      }
    } catch (...) { e = hipErrorOutOfMemory; }
    if (e != hipSuccess) { up_err = e; up.halt(); }
  });
  downloader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      Piece pc;
      while (down.pop(&pc)) {
        if (e != hipSuccess) continue;  // (drain)
        e = hipMemcpyAsync(out + pc.lo, (const uint8_t*)c->out.p + pc.lo, pc.hi - pc.lo, hipMemcpyDeviceToHost, c->dl_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->dl_stream);
      }
    } catch (...) { e = hipErrorOutOfMemory; }
    dl_err = e;
  });
  } catch (const std::system_error&) {  // no thread to be had: the call goes through in one piece
    up.halt();
    down.close();
    if (uploader.joinable()) uploader.join();
    return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
  }
  x3_params pp = *p;
  uint64_t pos = start_pos, k = 0;
  bool halted = false;
  // (nothing thrown between here and the joins may leave this frame: a joinable std::thread that is destroyed calls
  // std::terminate.  The hand-off's push allocates; everything else reports through return codes.)
  try {
    for (uint64_t s0 = 0; s0 < n && rc == X3_OK; s0 += chunk, ++k) {
      if (!up.wait_for(k + 1)) { halted = true; break; }
      const uint64_t cnt = std::min<uint64_t>(chunk, n - s0);
      x3_batch b{cnt, cnt, 1};
      uint64_t st[6] = {0, 0, 0, 0, 0, 0}, end = pos;
      {
        std::unique_lock<std::mutex> gate;
        if (c->enc_gate) gate = std::unique_lock<std::mutex>(*c->enc_gate);
        rc = encode_dev_impl(c, d_in + s0, &b, &pp, spf, (uint8_t*)c->out.p, out_cap, pos, nullptr);
        if (rc == X3_OK) rc = x3_encode_result(c, &end, st);
      }
      if (rc != X3_OK) break;
      if (stats)
        for (int i = 0; i < 6; ++i) stats[i] += st[i];
      if (end > pos) down.push({pos, end});
      pos = end;
    }
  } catch (...) {
    c->last_error = "x3_encode: out of host memory in the chunked pipeline";
    rc = X3_ERR_HIP;
  }
  up.halt();
  down.close();
  uploader.join();
  downloader.join();
  if (halted) HIPCHK(c, up_err);
  if (rc == X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY) {
    // the position that WOULD have been reached is part of the contract (include/x3hip.h): the sizes of all frames,
    // in one piece (an error path; nobody times it)
    if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
    return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
  }
  if (out_pos) *out_pos = pos;
  if (rc) return rc;
  HIPCHK(c, dl_err);
  return X3_OK;
}

extern "C" int x3_encode(x3_ctx* c, const int16_t* wav, uint64_t n, uint32_t n_channels, const x3_params* p,
                         uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c || !p || (!wav && n) || (!out && out_cap)) return X3_ERR_BAD_ARG;
  if (n_channels > 1) return X3_ERR_MORE_THAN_ONE_CHANNEL;  // encoder.rs:55-57
  if (n_channels == 0) return X3_ERR_BAD_ARG;                // channels[0] panics
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0 || n == 0) return X3_OK;  // take(0) / empty iterator: nothing is written (encoder.rs:67-73)
  if (c->opt.host_chunk_frames >= 0) {
    // chunks of whole frames, a multiple of eight of them (16-byte aligned chunk starts on the device)
    uint64_t frames = c->opt.host_chunk_frames ? (uint64_t)c->opt.host_chunk_frames : (16ull << 20) / spf;
    frames = std::max<uint64_t>(8, (frames + 7) & ~7ull);
    if (frames <= (~0ull >> 1) / spf && n / spf >= 2 * frames)
      return encode_host_chunked(c, wav, n, p, spf, frames * spf, out, out_cap, start_pos, out_pos, stats);
  }
  return encode_host(c, &wav, n, 1, p, spf, out, out_cap, start_pos, out_pos, nullptr, stats);
}

extern "C" int x3_encode_frame(x3_ctx* c, const int16_t* wav, uint64_t n, const x3_params* p, uint8_t* out,
                               uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!c || !p || !wav || (!out && out_cap)) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  if (n == 0) return X3_ERR_BAD_ARG;  // wav[0] panics (encoder.rs:189)
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  if (p->block_len == 0 && n > 1) return X3_ERR_BAD_ARG;  // chunks(0) panics
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;
  // the whole slice is ONE frame, whatever blocks_per_frame says
  return encode_host(c, &wav, n, 1, &pp, n, out, out_cap, start_pos, out_pos, nullptr, stats);
}

extern "C" int x3_encode_batch(x3_ctx* c, const int16_t* const* wavs, const uint64_t* ns, uint64_t count,
                               const x3_params* p, uint8_t* out, uint64_t out_cap, uint64_t* clip_offsets,
                               uint64_t stats[6]) {
  if (!c || !p || !wavs || !ns || !count || !clip_offsets) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0) return X3_ERR_BAD_ARG;
  bool uniform = true;
  for (uint64_t k = 1; k < count; ++k) uniform = uniform && ns[k] == ns[0];
  if (uniform && ns[0] > 0) {
    uint64_t pos = 0;
    return encode_host(c, wavs, ns[0], count, p, spf, out, out_cap, 0, &pos, clip_offsets, stats);
  }
  // ragged batch: one launch set per clip, streams appended back to back
  uint64_t pos = 0;
  clip_offsets[0] = 0;
  for (uint64_t k = 0; k < count; ++k) {
    uint64_t st[6] = {0, 0, 0, 0, 0, 0};
    if (ns[k]) {
      rc = encode_host(c, &wavs[k], ns[k], 1, p, spf, out, out_cap, pos, &pos, nullptr, st);
      if (rc) return rc;
    }
    clip_offsets[k + 1] = pos;
    if (stats)
      for (int i = 0; i < 6; ++i) stats[i] += st[i];
  }
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
// wav_off_aligned: the caller knows that every d_wav_offsets[f] is a multiple of eight samples (the two-wave decoder
// writes 16-byte aligned rows); without that knowledge caller-supplied offsets go to the single-wave kernels
static int decode_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                           uint64_t F, const x3_batch* batch, const uint64_t* d_wav_offsets, const x3_params* p,
                           int16_t* d_wav, uint64_t wav_cap, int32_t* d_status, bool wav_off_aligned = false,
                           bool bl0 = false) {  // bl0: the caller's block_len is 0 and p carries 1 (x3_decode_merge_kernel)
  if (reinterpret_cast<uintptr_t>(d_x3) & 3u) return X3_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(d_wav) & 1u) return X3_ERR_BAD_ARG;
  if (F == 0 || F > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  const uint64_t spf = spf_of(p);
  X3DevParams dp;
  int rc = derive(p, spf > 0xFFFFFFFFull ? 0 : spf, &dp);
  if (rc) return rc;
  X3Geom g{0, 0, 1, F};
  if (!d_wav_offsets) {
    if (!batch || spf == 0) return X3_ERR_BAD_ARG;
    const uint64_t fpc = (batch->n_per_clip + spf - 1) / spf;
    if (fpc == 0 || fpc > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
    g.n_per_clip = batch->n_per_clip;
    g.clip_stride = batch->clip_stride;
    g.fpc = (uint32_t)fpc;
  }
  if ((rc = ensure(c, c->dec_meta, F * sizeof(X3FrameMeta)))) return rc;
  if (!d_status) {
    if ((rc = ensure(c, c->dec_status, F * sizeof(int32_t)))) return rc;
    d_status = (int32_t*)c->dec_status.p;
  }
  if ((rc = ensure(c, c->dec_cstatus, F * sizeof(int32_t)))) return rc;
  // fork: header + payload-CRC pass on the side stream, decoder on the main stream (independent;
  // the decoder's one-wave-per-SIMD dependency chains leave the CUs mostly idle)
#ifdef X3_PROFILING
  hipStream_t check_stream = c->opt.check_serial ? c->stream : c->stream2;
  const bool no_check = c->opt.no_check != 0;
#else
  // (option check_main: the two passes swap streams -- the one on the side stream starts a cross-queue event later)
  hipStream_t check_stream = c->opt.check_main ? c->stream : c->stream2;
  const bool no_check = false;
#endif
  hipStream_t dec_stream = check_stream == c->stream2 ? c->stream : c->stream2;
  const uint64_t check_wgs_per_cu = (uint64_t)c->opt.check_wgs;
  HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
  // The check pass and the decoder are independent and run side by side; the decoder is enqueued first.  (Neither the
  // order nor the side stream's queue priority selects between the decoder's timing modes -- 0.81 / 0.87 / 0.94 ms
  // per process on one box, 0.98 on another with no check kernel at all: measured, tools/dbg_modes.sh, DESIGN.md.)
  auto launch_check = [&]() -> int {
    if (no_check) {
      // profiling builds only: time the decoder without the check pass beside it (payload CRCs are NOT verified)
      HIPCHK(c, hipMemsetAsync(c->dec_cstatus.p, 0, F * sizeof(int32_t), c->stream2));

    } else {
      TimerScope ts(c, 4, check_stream, true);
      const uint64_t check_grid = std::min<uint64_t>((F + 3) / 4, (uint64_t)c->n_cus * check_wgs_per_cu);
      X3_LAUNCH_TIMED(ts, x3_frame_check_kernel, dim3((unsigned)check_grid), dim3(256), 0, check_stream,
                         reinterpret_cast<const uint32_t*>(d_x3), x3_len, d_frame_offsets, F,
                         (const uint16_t*)c->d_xinv8, (const uint16_t*)c->d_chktab, (const uint32_t*)c->d_kx64,
                         (int32_t*)c->dec_cstatus.p, reinterpret_cast<unsigned long long*>(c->d_summary), 1u);
    }
    if (check_stream == c->stream2) HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
    return X3_OK;
  };
  if (c->opt.check_first && (rc = launch_check())) return rc;
  {
    // the branch-free kernel needs every valid Rice codeword (zeros + terminator + sub-code) to fit 32 bits
    bool fast = true;
    const uint32_t widths[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
      const uint32_t level = k == 0 ? 1u : (1u << dp.k[k]);
      fast = fast && (dp.inv_len[k] / level + 1 + widths[k] <= 32);
    }
    // two waves per group of 64 frames (parser + valuer) when the geometry is the plain one.  Its parser hands
    // over i = (z << k) + r with r the k bits behind the terminating one, which equals the reference's
    // r' + level * (n - 1) (decoder.rs:186, r' = the hard-wired 2 / 4 bits INCLUDING the one) only when the
    // code of ftype 2 has one sub-bit and that of ftype 3 three -- the default codes; the single-wave kernels
    // follow the reference's formula literally and take every other code set.
    const bool split = fast && dp.block_len == X3S_BL && dp.k[1] == 1u && dp.k[2] == 3u &&
                       (!d_wav_offsets || wav_off_aligned) && !c->force_single_wave_decode &&
                       !c->opt.decode_single && (reinterpret_cast<uintptr_t>(d_wav) & 15u) == 0 &&
                       (d_wav_offsets || ((dp.spf % 8u) == 0 && (g.fpc * (uint64_t)1 >= g.n_frames || (g.clip_stride % 8u) == 0)));
#ifdef X3_PROFILING
    const size_t dyn_lds = (size_t)c->opt.dyn_lds;
#else
    const size_t dyn_lds = 0;
#endif
    TimerScope ts(c, 1, dec_stream, split);   // (the split kernel: events on its dispatch packet; the rarer single-wave kernels below: bracketed)
    if (split) {
      // the pace word's 12-bit epoch: launches 1, 2, ... 4095, then the word starts over
      if ((c->dec_epoch & 0xFFFu) == 0u) {
        HIPCHK(c, hipMemsetAsync(c->d_pace, 0, 16, dec_stream));  // (achieved and aimed-at, one word per launch parity)
        HIPCHK(c, hipMemsetAsync(c->d_pace + X3_LOG_BASE, 0, X3_LOG_ENTRIES * X3_LOG_WORDS * sizeof(uint32_t), dec_stream));  // (its atomicMax entries carry the epoch too)
        ++c->dec_epoch;
      }
      X3_LAUNCH_TIMED(ts, x3_decode_split_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64 * X3S_WAVES),
                         dyn_lds, dec_stream, d_x3, x3_len,
                         d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status, (X3FrameMeta*)c->dec_meta.p,
                         c->d_pace, c->dec_epoch & 0xFFFu);
      ++c->dec_epoch;
    }
    else if (fast)
      hipLaunchKernelGGL(x3_decode_fast_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, dec_stream, d_x3, x3_len,
                         d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status,
                         (X3FrameMeta*)c->dec_meta.p);
    else
      hipLaunchKernelGGL((x3_decode_lanes_kernel<false, 64>), dim3((unsigned)((F + 63) / 64)), dim3(64), 0, dec_stream,
                         d_x3, x3_len, d_frame_offsets, F, g, d_wav_offsets, dp, d_wav, wav_cap, d_status,
                         (X3FrameMeta*)c->dec_meta.p);
  }
  if (dec_stream == c->stream2) HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
  if (!c->opt.check_first && (rc = launch_check())) return rc;
  // join, then merge the two status arrays and summarise
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  if (no_check) {  // otherwise the check kernel's first thread does this
    X3DecodeSummary init;
    init.first_bad = F;
    init.samples_before = 0;
    init.first_bad_status = 0;
    init.pad = 0;
    *c->h_summary_init = init;
    HIPCHK(c, hipMemcpyAsync(c->d_summary, c->h_summary_init, sizeof init, hipMemcpyHostToDevice, c->stream));
  }
  hipLaunchKernelGGL(x3_decode_merge_kernel, dim3((unsigned)std::min<uint64_t>((F + 255) / 256, 64)), dim3(256), 0, c->stream,
                     (const int32_t*)c->dec_cstatus.p, d_status, (const X3FrameMeta*)c->dec_meta.p, F, c->d_summary,
                     d_x3, d_frame_offsets, g, d_wav_offsets, dp, d_wav, bl0 ? 1u : 0u);
  c->dec_status_ptr = d_status;
  HIPCHK(c, hipGetLastError());
  c->decode_pending = true;
  c->dec_frames = F;
  return X3_OK;
}

extern "C" int x3_decode_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t x3_len, const uint64_t* d_frame_offsets,
                             uint64_t n_frames, const x3_batch* batch, const uint64_t* d_wav_offsets,
                             const x3_params* p, int16_t* d_wav, uint64_t wav_cap, int32_t* d_status) {
  if (!c || !d_x3 || !d_frame_offsets || !p || !d_wav) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return decode_dev_impl(c, d_x3, x3_len, d_frame_offsets, n_frames, batch, d_wav_offsets, p, d_wav, wav_cap, d_status);
}

extern "C" int x3_decode_result(x3_ctx* c, uint64_t* first_bad, int* first_bad_status, uint64_t* samples_before) {
  if (!c || !c->decode_pending) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(X3DecodeSummary), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->decode_pending = false;
  if (c->h_summary->first_bad < c->dec_frames) {
    // rare: a frame is bad -- its status and the samples of the good frames before it
    hipLaunchKernelGGL(x3_decode_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, (const int32_t*)c->dec_status_ptr,
                       (const X3FrameMeta*)c->dec_meta.p, c->dec_frames, c->d_summary);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_summary, c->d_summary, sizeof(X3DecodeSummary), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (first_bad) *first_bad = c->h_summary->first_bad;
  if (first_bad_status) *first_bad_status = c->h_summary->first_bad_status;
  if (samples_before) *samples_before = c->h_summary->samples_before;
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------
// GPU-side frame index of a device-resident stream (x3_index_kernels.h)
// ------------------------------------------------------------------------------------------------
static int index_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, uint32_t bl0, uint64_t wav_cap,
                          uint64_t max_frames, uint64_t* d_frame_offsets, uint64_t* d_wav_offsets,
                          X3IndexSummary* result) {
  if (reinterpret_cast<uintptr_t>(d_x3) & 3u) return X3_ERR_BAD_ARG;
  const uint32_t* xw = reinterpret_cast<const uint32_t*>(d_x3);
  int rc;
  if ((rc = ensure(c, c->idx_sum, 256))) return rc;
  unsigned int* d_count = reinterpret_cast<unsigned int*>((char*)c->idx_sum.p + 128);
  X3IndexSummary* d_sum = reinterpret_cast<X3IndexSummary*>(c->idx_sum.p);
  hipLaunchKernelGGL(x3_index_init_kernel, dim3(1), dim3(64), 0, c->stream, d_sum, d_count);
  const uint64_t chunks = (len + 15) >> 4;
  const unsigned grid = (unsigned)std::min<uint64_t>((chunks + 255) / 256, (uint64_t)c->n_cus * 16);
  // ONE pass over the stream: the candidates go into a buffer sized for a frame every 256 bytes (the context keeps
  // it; typical streams hold one every few kilobytes); only a stream with more than that is scanned a second time.
  unsigned int n_cand = 0;
  if (grid) {
    const uint64_t guess = std::max<uint64_t>(4096, len / 256 + 1024);
    if ((rc = ensure(c, c->idx_cand, (size_t)std::min<uint64_t>(guess, 0x7FFFFFFFull) * sizeof(X3Cand)))) return rc;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(c->idx_cand.cap / sizeof(X3Cand), 0x7FFFFFFFull);
    hipLaunchKernelGGL(x3_index_candidates_kernel, dim3(grid), dim3(256), 0, c->stream, xw, len, len + phantom, bl0,
                       (X3Cand*)c->idx_cand.p, cap, d_count);
    HIPCHK(c, hipMemcpyAsync(c->h_crc, d_count, sizeof n_cand, hipMemcpyDeviceToHost, c->stream));  // (pinned)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(&n_cand, c->h_crc, sizeof n_cand);
    if (n_cand > cap) {  // rare: denser than one frame per 256 bytes
      if ((rc = ensure(c, c->idx_cand, (size_t)n_cand * sizeof(X3Cand)))) return rc;
      HIPCHK(c, hipMemsetAsync(d_count, 0, sizeof(unsigned int), c->stream));
      hipLaunchKernelGGL(x3_index_candidates_kernel, dim3(grid), dim3(256), 0, c->stream, xw, len, len + phantom, bl0,
                         (X3Cand*)c->idx_cand.p, n_cand, d_count);
    }
  }
  if (n_cand) {
    uint32_t tsize = 1024;
    while (tsize < 2u * n_cand && tsize < 0x80000000u) tsize <<= 1;
    uint32_t levels = 1;
    while ((1ull << (levels - 1)) < n_cand) ++levels;  // the top level spans 2^(levels-1) >= n_cand >= any chain
    if ((rc = ensure(c, c->idx_keys, (size_t)tsize * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, c->idx_vals, (size_t)tsize * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->idx_J, (size_t)levels * n_cand * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->idx_S, (size_t)levels * n_cand * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, c->idx_L, (size_t)levels * n_cand * sizeof(uint32_t)))) return rc;
    X3Cand* cand = (X3Cand*)c->idx_cand.p;
    unsigned long long* keys = (unsigned long long*)c->idx_keys.p;
    uint32_t* vals = (uint32_t*)c->idx_vals.p;
    uint32_t* J = (uint32_t*)c->idx_J.p;
    unsigned long long* S = (unsigned long long*)c->idx_S.p;
    uint32_t* L = (uint32_t*)c->idx_L.p;
    HIPCHK(c, hipMemsetAsync(keys, 0, (size_t)tsize * sizeof(unsigned long long), c->stream));
    const unsigned cg = (n_cand + 255) / 256;
    hipLaunchKernelGGL(x3_index_hash_insert_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand,
                       keys, vals, tsize - 1);
    hipLaunchKernelGGL(x3_index_succ_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand,
                       (const unsigned long long*)keys, (const uint32_t*)vals, tsize - 1, J, S, L);
    uint32_t r = 1;
    for (; r + 4 <= levels; r += 4)  // four levels a launch
      hipLaunchKernelGGL(x3_index_double4_kernel, dim3(cg), dim3(256), 0, c->stream, n_cand,
                         (const uint32_t*)(J + (size_t)(r - 1) * n_cand),
                         (const unsigned long long*)(S + (size_t)(r - 1) * n_cand),
                         (const uint32_t*)(L + (size_t)(r - 1) * n_cand), J + (size_t)r * n_cand,
                         S + (size_t)r * n_cand, L + (size_t)r * n_cand);
    for (; r < levels; ++r)
      hipLaunchKernelGGL(x3_index_double_kernel, dim3(cg), dim3(256), 0, c->stream, n_cand,
                         (const uint32_t*)(J + (size_t)(r - 1) * n_cand),
                         (const unsigned long long*)(S + (size_t)(r - 1) * n_cand),
                         (const uint32_t*)(L + (size_t)(r - 1) * n_cand), J + (size_t)r * n_cand,
                         S + (size_t)r * n_cand, L + (size_t)r * n_cand);
    // start node and chain length stay on the device: the emit grid covers every candidate
    hipLaunchKernelGGL(x3_index_emit_kernel, dim3(cg), dim3(256), 0, c->stream, (const X3Cand*)cand, n_cand, levels,
                       (const uint32_t*)J, (const unsigned long long*)S, (unsigned long long)max_frames,
                       (unsigned long long)wav_cap, (unsigned long long*)d_frame_offsets,
                       (unsigned long long*)d_wav_offsets, d_sum, (const unsigned long long*)keys, (const uint32_t*)vals,
                       tsize - 1, (const uint32_t*)(L + (size_t)(levels - 1) * n_cand));
    hipLaunchKernelGGL(x3_index_finalize_kernel, dim3(1), dim3(64), 0, c->stream, xw, len, len + phantom, bl0,
                       (const X3Cand*)cand, (const unsigned long long*)d_wav_offsets, d_sum);
  } else {
    hipLaunchKernelGGL(x3_index_finalize_kernel, dim3(1), dim3(64), 0, c->stream, xw, len, len + phantom, bl0,
                       (const X3Cand*)nullptr, (const unsigned long long*)nullptr, d_sum);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_summary_init, d_sum, sizeof *result, hipMemcpyDeviceToHost, c->stream));  // (pinned)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(result, c->h_summary_init, sizeof *result);
  if (result->pad) {
    c->last_error = "x3_index_dev: more frames in the stream than max_frames";
    return X3_ERR_BAD_ARG;
  }
  return X3_OK;
}

extern "C" int x3_index_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t max_frames,
                            uint64_t* d_frame_offsets, uint64_t* d_wav_offsets, uint64_t* n_frames,
                            uint64_t* n_samples, int* terminal) {
  if (!c || (!d_x3 && len) || !d_frame_offsets || !d_wav_offsets) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  X3IndexSummary r;
  int rc = index_dev_impl(c, d_x3, len, 0, 0, ~0ull, max_frames, d_frame_offsets, d_wav_offsets, &r);
  if (rc) return rc;
  if (n_frames) *n_frames = r.n_frames;
  if (n_samples) *n_samples = r.n_samples;
  if (terminal) *terminal = r.terminal;
  return X3_OK;
}

static int decode_stream_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                  int16_t* d_wav, uint64_t wav_cap, DevBuf* own_out, uint64_t* n_out,
                                  uint64_t* frames_ok, uint64_t* frame_errors);

extern "C" int x3_decode_stream_dev(x3_ctx* c, const uint8_t* d_x3, uint64_t len, const x3_params* p, int16_t* d_wav,
                                    uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  if (!c || !p || (!d_x3 && len) || (!d_wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  return decode_stream_dev_impl(c, d_x3, len, 0, p, d_wav, wav_cap, nullptr, n_out, frames_ok, frame_errors);
}

// the walk on the GPU (x3_index_kernels.h), then one decode launch; `phantom` as in walk_host.  own_out: decode
// into this scratch buffer, sized once the index knows the sample count, instead of d_wav.
static int decode_stream_dev_impl(x3_ctx* c, const uint8_t* d_x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                  int16_t* d_wav, uint64_t wav_cap, DevBuf* own_out, uint64_t* n_out,
                                  uint64_t* frames_ok, uint64_t* frame_errors) {
  // every frame is at least 22 bytes; index into internal buffers sized for the real count
  const uint64_t max_frames = len / 22 + 1;
  int rc;
  X3IndexSummary r;
  // sizes are not known before the candidate count: index_dev_impl sizes its own scratch, the two output arrays
  // are sized here from an upper bound that is refined by a first call when it is large
  uint64_t cap_frames = std::min<uint64_t>(max_frames, 1u << 20);
  for (;;) {
    if ((rc = ensure(c, c->frame_off, (cap_frames + 1) * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(c, c->wav_off, cap_frames * sizeof(uint64_t)))) return rc;
    rc = index_dev_impl(c, d_x3, len, phantom, p->block_len == 0 ? 1u : 0u, wav_cap, cap_frames, (uint64_t*)c->frame_off.p,
                        (uint64_t*)c->wav_off.p, &r);
    if (rc == X3_ERR_BAD_ARG && cap_frames < max_frames) { cap_frames = max_frames; continue; }
    break;
  }
  if (rc) return rc;
  const uint64_t F = r.n_frames;
  const int terminal = r.terminal;
  if (F == 0) return terminal;
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;  // frames that need block_len are BAD_ARG frames of the index
  if (own_out) {
    if ((rc = ensure(c, *own_out, (r.n_samples + 65536) * sizeof(int16_t)))) return rc;
    d_wav = (int16_t*)own_out->p;
    wav_cap = std::min<uint64_t>(wav_cap, r.n_samples + 65535);
  }
  if ((rc = decode_dev_impl(c, d_x3, len, (const uint64_t*)c->frame_off.p, F, nullptr, (const uint64_t*)c->wav_off.p,
                            &pp, d_wav, wav_cap, nullptr, r.unaligned == 0, p->block_len == 0)))
    return rc;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
  if (n_out) *n_out = before;
  if (frames_ok) *frames_ok = first_bad;
  if (first_bad < F) {
    if (bad_status == X3_ERR_OUT_OF_BOUNDS_INVERSE || bad_status == X3_ERR_FRAME_DECODE_INVALID_BPF) {
      if (frame_errors) *frame_errors = 1;  // counted, the walk ends quietly (decodefile.rs:129-135)
      return X3_OK;
    }
    return bad_status;
  }
  return terminal;
}

static int decode_stream_impl(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                              int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                              uint64_t* frame_errors);

extern "C" int x3_decode_stream(x3_ctx* c, const uint8_t* x3, uint64_t len, const x3_params* p, int16_t* wav,
                                uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  return decode_stream_impl(c, x3, len, 0, p, wav, wav_cap, n_out, frames_ok, frame_errors);
}

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
static void walk_host(const uint8_t* buf, uint64_t buf_len, uint64_t real_total, uint64_t believed_total,
                      const x3_params* p, uint64_t wav_cap, uint64_t max_samples, HostWalk* w, uint32_t n_ch = 1u) {
  uint64_t pos = 0, remaining = believed_total, nsamp = 0;
  for (;;) {
    if (remaining <= 20) break;
    if (real_total - pos < 20) { w->terminal = X3_ERR_IO; break; }  // read_exact past the real end of the data
    if (buf_len - pos < 20) { w->need_more = true; break; }
    x3_frame_header h;
    int rc = read_frame_header_ch(buf + pos, 20, &h, n_ch);
    if (rc) { w->terminal = rc; break; }
    if (remaining - 20 < h.payload_len) break;
    // the buffer-size test comes before the payload is read (decodefile.rs:118-124): a payload that is both too
    // long and cut off by the real end of the data is FrameHeaderInvalidPayloadLen, not Io
    if (h.payload_len > X3_READ_BUFFER_SIZE) { w->terminal = X3_ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN; break; }
    if (real_total - pos - 20 < h.payload_len) { w->terminal = X3_ERR_IO; break; }
    if (buf_len - pos - 20 < h.payload_len) { w->need_more = true; break; }
    if (h.samples == 0 || h.payload_len < 2 * n_ch || nsamp + h.samples > wav_cap || (p->block_len == 0 && h.samples > 1)) {
      // payload CRC is checked before decode_frame runs, so let the GPU look at this frame too:
      // it reports the CRC error if there is one, BAD_ARG (reference panic) otherwise
      w->offs.push_back(pos);
      w->woffs.push_back(nsamp);
      w->terminal = X3_ERR_BAD_ARG;
      pos += 20 + (uint64_t)h.payload_len;
      break;
    }
    if (nsamp + h.samples > max_samples && !w->offs.empty()) { w->need_more = true; break; }
    w->offs.push_back(pos);
    w->woffs.push_back(nsamp);
    nsamp += h.samples;
    pos += 20 + h.payload_len;
    remaining -= 20 + h.payload_len;
  }
  w->nsamp = nsamp;
  w->end_pos = pos;
}

// decode the frames a walk collected from host memory into host memory: H2D, one decode launch, D2H of the
// samples in front of the first frame that fails.  *first_bad == F: all of them decoded.
static int decode_frames_host(x3_ctx* c, const uint8_t* x3, uint64_t len, const HostWalk& w, const x3_params* p,
                              int16_t* wav, uint64_t wav_cap, uint64_t* before, uint64_t* first_bad, int* bad_status,
                              bool download = true,  // !download: the samples stay in c->out (x3_mgpu_decode_stream)
                              const uint8_t* d_x3 = nullptr) {  // the frames' bytes are on the device already
  const uint64_t F = w.offs.size();
  *before = 0;
  *first_bad = 0;
  *bad_status = 0;
  if (F == 0) return X3_OK;
  int rc;
  if (!d_x3) {
    if ((rc = ensure(c, c->in, len + 16))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, x3, len, hipMemcpyHostToDevice, c->stream));
    d_x3 = (const uint8_t*)c->in.p;
  }
  if ((rc = ensure(c, c->frame_off, (F + 1) * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, F * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, (w.nsamp + 65536) * sizeof(int16_t)))) return rc;
  // (through pinned memory: a copy from pageable memory is staged by the runtime under a lock it shares with the large
  // pageable copies the chunked front ends have in flight on other threads -- 0.2-0.4 ms per call when they collide)
  if (c->h_walk_cap < 2 * F * sizeof(uint64_t)) {
    if (c->h_walk) HIPCHK(c, hipHostFree(c->h_walk));
    c->h_walk = nullptr;
    c->h_walk_cap = 0;
    const size_t want = (2 * F * sizeof(uint64_t) * 5 / 4 + 4095) & ~(size_t)4095;
    HIPCHK(c, hipHostMalloc(&c->h_walk, want));
    c->h_walk_cap = want;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (the copy before this one has left the pinned block)
  std::memcpy(c->h_walk, w.offs.data(), F * sizeof(uint64_t));
  std::memcpy((uint64_t*)c->h_walk + F, w.woffs.data(), F * sizeof(uint64_t));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, c->h_walk, F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, (uint64_t*)c->h_walk + F, F * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;  // frames that need block_len were routed to BAD_ARG by the walk
  const uint64_t dev_wav_cap = std::min<uint64_t>(wav_cap, w.nsamp + 65535);
  bool aligned = true;
  for (uint64_t v : w.woffs) aligned = aligned && (v & 7ull) == 0;
  if ((rc = decode_dev_impl(c, d_x3, len, (const uint64_t*)c->frame_off.p, F, nullptr,
                            (const uint64_t*)c->wav_off.p, &pp, (int16_t*)c->out.p, dev_wav_cap, nullptr, aligned,
                            p->block_len == 0)))
    return rc;
  if ((rc = x3_decode_result(c, first_bad, bad_status, before))) return rc;
  if (download && *before)
    HIPCHK(c, hipMemcpyAsync(wav, c->out.p, *before * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

// how the walk ends when frame `first_bad` of F failed with `bad_status` (decodefile.rs:96-100, 128-135)
static int walk_result(uint64_t F, uint64_t first_bad, int bad_status, int terminal, uint64_t* frame_errors) {
  if (first_bad < F) {
    if (bad_status == X3_ERR_OUT_OF_BOUNDS_INVERSE || bad_status == X3_ERR_FRAME_DECODE_INVALID_BPF) {
      if (frame_errors) *frame_errors = 1;  // counted, the walk ends quietly (decodefile.rs:129-135)
      return X3_OK;
    }
    return bad_status;  // payload CRC mismatch (hard error) or BAD_ARG (reference panic)
  }
  return terminal;
}

// x3_decode_stream on a LONG stream in host memory, in chunks of whole frames (the file pipeline's scheme in one
// context).  The samples are 79 % of the bytes this entry point moves (config 3) and the link carries both directions at
// once, so: one host thread walks the headers of chunk i+2 (decodefile.rs:105-121, one dependent cache miss per frame)
// and sends its bytes up, this one decodes chunk i+1 (a launch of a few thousand frames lasts as long as one group of 64
// does: 0.6 ms, whatever the GPU could do beside it), a third brings the samples of chunk i down.  Two device buffers
// take turns on either side of the decoder.
static int decode_stream_host_chunked(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                                      uint64_t chunk_samples, bool grow, int16_t* wav, uint64_t wav_cap, uint64_t* n_out,
                                      uint64_t* frames_ok, uint64_t* frame_errors) {
  int rc = x3_pipe_streams(c);
  if (rc) return rc;
  struct Chunk { HostWalk hw; uint64_t a, sample_off; const uint8_t* d_x3; };
  struct Piece { const void* src; uint64_t sample_off, count; };
  X3Handoff<Chunk> ready;
  X3Handoff<Piece> down;
  X3Progress decoded, landed;
  hipError_t up_err = hipSuccess, dl_err = hipSuccess;
  std::thread uploader, downloader;
  try {
  uploader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
    e = hipSetDevice(c->device);
    uint64_t a = 0, sample_off = 0;
    for (uint64_t k = 0; e == hipSuccess; ++k) {
      { std::lock_guard<std::mutex> g(decoded.mu); if (decoded.stop) break; }
      Chunk ck;
      const uint64_t real_total = len - a;
      // grow: a short first chunk so that the downloads start early, then longer ones (x1.5 up to x8: a chunk's walk and
      // upload take 0.7 of the time its predecessor's samples need to come down) -- every chunk costs a launch set and a
      // handful of runtime calls whatever its size, and those calls now and then stall for milliseconds beside the
      // pageable copies of the other two threads
      uint64_t budget = chunk_samples;
      if (grow)
        for (uint64_t g = 0; g < k && budget < 8 * chunk_samples; ++g) budget += budget >> 1;
      walk_host(x3 + a, real_total, real_total, real_total + phantom, p, wav_cap - sample_off, budget, &ck.hw);
      ck.a = a;
      ck.sample_off = sample_off;
      ck.d_x3 = nullptr;
      const bool last = !ck.hw.need_more;
      if (!ck.hw.offs.empty()) {
        // this chunk's bytes go where those of chunk k-3 were: not before that chunk has been decoded
        if (k >= 3 && !decoded.wait_for(k - 2)) break;
        DevBuf& buf = (k % 3) ? c->in_more[k % 3 - 1] : c->in;
        if (buf.cap < ck.hw.end_pos + 16) {
          if (buf.p) e = hipFree(buf.p);
          buf.p = nullptr;
          buf.cap = 0;
          const size_t want = (size_t)((ck.hw.end_pos * (grow ? 2 : 1) + 16 + (ck.hw.end_pos >> 3) + 255) & ~255ull);
          if (e == hipSuccess) e = hipMalloc(&buf.p, want);
          if (e == hipSuccess) buf.cap = want;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(buf.p, x3 + a, ck.hw.end_pos, hipMemcpyHostToDevice, c->ul_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->ul_stream);
        if (e != hipSuccess) break;
        ck.d_x3 = static_cast<const uint8_t*>(buf.p);
      }
      a += ck.hw.end_pos;
      sample_off += ck.hw.nsamp;
      ready.push(std::move(ck));
      if (last) break;
    }
    } catch (...) { e = hipErrorOutOfMemory; }   // (walk_host's vectors, the hand-off: never std::terminate -- ADVICE r3)
    up_err = e;
    ready.close();
  });
  downloader = std::thread([&] {
    hipError_t e = hipSuccess;
    try {
      e = hipSetDevice(c->device);
      Piece pc;
      while (down.pop(&pc)) {
        if (e == hipSuccess && pc.count) {
          e = hipMemcpyAsync(wav + pc.sample_off, pc.src, pc.count * sizeof(int16_t), hipMemcpyDeviceToHost, c->dl_stream);
          if (e == hipSuccess) e = hipStreamSynchronize(c->dl_stream);
        }
        landed.advance();
      }
    } catch (...) { e = hipErrorOutOfMemory; landed.halt(); }
    dl_err = e;
  });
  } catch (const std::system_error&) {  // no thread to be had: the caller takes the stream in one piece
    decoded.halt();
    if (uploader.joinable()) uploader.join();
    return X3_PIPE_UNAVAILABLE;
  }
  uint64_t total = 0, frames = 0, ferr = 0, k = 0;
  int result = X3_OK;
  bool ended = false;
  Chunk ck;
  try {   // (as in encode_host_chunked: nothing thrown may pass the joins below)
  while (!ended && ready.pop(&ck)) {
    const uint64_t F = ck.hw.offs.size();
    uint64_t before = 0, first_bad = 0;
    int bad_status = 0;
    if (F) {
      // the buffer this chunk decodes into was the source of the download three chunks back
      if (k >= 3) landed.wait_for(k - 2);
      DevBuf* turn = (k % 3) ? &c->out_more[k % 3 - 1] : nullptr;
      if (turn) std::swap(c->out, *turn);
      if (grow && c->out.cap < (ck.hw.nsamp + 65536) * sizeof(int16_t))  // (room for the longer chunks that follow)
        rc = ensure(c, c->out, (2 * ck.hw.nsamp + 65536) * sizeof(int16_t));
      if (rc == X3_OK)
        rc = decode_frames_host(c, nullptr, ck.hw.end_pos, ck.hw, p, nullptr, wav_cap - ck.sample_off, &before, &first_bad,
                                &bad_status, false, ck.d_x3);
      const void* src = c->out.p;
      if (turn) std::swap(c->out, *turn);
      if (rc) break;
      down.push({src, ck.sample_off, before});
      decoded.advance();
      ++k;
    }
    uint64_t fe = 0;
    result = walk_result(F, first_bad, bad_status, ck.hw.need_more ? X3_OK : ck.hw.terminal, &fe);
    ferr += fe;
    frames += first_bad < F ? first_bad : F;
    total = ck.sample_off + before;
    ended = first_bad < F || !ck.hw.need_more;
  }
  } catch (...) {
    c->last_error = "x3_decode_stream: out of host memory in the chunked pipeline";
    rc = X3_ERR_HIP;
  }
  decoded.halt();  // (an uploader that waits for a decode that will not come)
  try { while (ready.pop(&ck)) {} } catch (...) {}
  down.close();
  uploader.join();
  downloader.join();
  if (rc) return rc;
  if (!ended) HIPCHK(c, up_err);
  HIPCHK(c, dl_err);
  if (n_out) *n_out = total;
  if (frames_ok) *frames_ok = frames;
  if (frame_errors) *frame_errors = ferr;
  return result;
}

// `phantom`: bytes the reader BELIEVES remain beyond the real data; a read that runs past the real end is
// X3Error::Io.
static int decode_stream_impl(x3_ctx* c, const uint8_t* x3, uint64_t len, uint64_t phantom, const x3_params* p,
                              int16_t* wav, uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok,
                              uint64_t* frame_errors) {
  if (!c || !p || (!x3 && len) || (!wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  HIPCHK(c, hipSetDevice(c->device));
  // Long streams: the header chain is one dependent cache miss per frame on the host (10 ms for the 69 120
  // frames of config 3), and a few launches on the GPU once the bytes are there anyway.  Short ones: the
  // other way round.  Option "host_walk" = 0/1 forces one or the other (tests run both).
  bool gpu_walk = len >= (4u << 20);
  if (c->opt.host_walk >= 0) gpu_walk = c->opt.host_walk == 0;
  // long streams in chunks, downloads beside uploads (unless a test pins the walk to the GPU)
  if (c->opt.host_walk != 0 && c->opt.host_chunk_frames >= 0 && len > 20 &&
      (c->opt.host_chunk_frames > 0 || len >= (16u << 20))) {
    const uint64_t spf = std::max<uint64_t>(spf_of(p), 1);
    const uint64_t chunk = c->opt.host_chunk_frames > 0 ? (uint64_t)c->opt.host_chunk_frames * spf : 16ull << 20;
    const int rc = decode_stream_host_chunked(c, x3, len, phantom, p, std::max<uint64_t>(chunk, 1), c->opt.host_chunk_frames == 0,
                                              wav, wav_cap, n_out, frames_ok, frame_errors);
    if (rc != X3_PIPE_UNAVAILABLE) return rc;
  }
  if (gpu_walk && len > 0) {
    int rc = ensure(c, c->in, len + 16);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, x3, len, hipMemcpyHostToDevice, c->stream));
    uint64_t before = 0;
    rc = decode_stream_dev_impl(c, (const uint8_t*)c->in.p, len, phantom, p, nullptr, wav_cap, &c->out, &before,
                                frames_ok, frame_errors);
    if (before) HIPCHK(c, hipMemcpyAsync(wav, c->out.p, before * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_out) *n_out = before;
    return rc;
  }
  HostWalk w;
  walk_host(x3, len, len, len + phantom, p, wav_cap, ~0ull, &w);
  const uint64_t F = w.offs.size();
  if (F == 0) return w.terminal;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  int rc = decode_frames_host(c, x3, len, w, p, wav, wav_cap, &before, &first_bad, &bad_status);
  if (rc) return rc;
  if (n_out) *n_out = before;
  if (frames_ok) *frames_ok = first_bad;
  return walk_result(F, first_bad, bad_status, w.terminal, frame_errors);
}

// single bare payload (decoder::decode_frame): wrap it in a frame header so that the one decode
// kernel serves both paths; decode_frame itself checks no CRC, so a correct one is supplied.
#define X3_FRAME_CACHE_MISS (-1)
static int frame_cache_serve(x3_ctx* c, const uint8_t* payload, uint64_t len, const x3_params* p, uint64_t samples, int16_t* wav);

static int decode_frame_impl(x3_ctx* c, const uint8_t* payload, uint64_t len, int16_t* wav, uint64_t wav_cap,
                             const x3_params* p, uint64_t samples, uint64_t* n_out, bool use_cache) {
  if (!c || !p || !payload || !wav) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (len < 2 || samples == 0 || wav_cap < 1) return X3_ERR_BAD_ARG;  // reference panics
  // block_len == 0: every block is empty, the frame's fate is in its block type bits (x3_replay_frame) and no block
  // ever fails to fit
  const bool bl0 = p->block_len == 0 && samples > 1;
  if (samples > wav_cap && !bl0) {
    // decode_frame slices wav block by block (decoder.rs:49) and panics at the first block that does not fit -- but an
    // error in a block in front of that one is returned first.  The blocks in front of it are a frame of their own:
    const uint64_t bl = p->block_len ? p->block_len : 1;
    const uint64_t n_fit = 1 + ((wav_cap - 1) / bl) * bl;  // the first sample and the whole blocks that fit
    if (n_fit > 1) {
      std::vector<int16_t> tmp(n_fit);
      const int rc_fit = decode_frame_impl(c, payload, len, tmp.data(), n_fit, p, n_fit, nullptr, false);
      if (rc_fit != X3_OK) return rc_fit;
    }
    return X3_ERR_BAD_ARG;  // slice index panic
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if (bl0 || samples > 0xFFFF || len >= X3_FRAME_MAX_LENGTH || len > X3_READ_BUFFER_SIZE) {
    // not a frame the walk would hand over (decodefile.rs:118-121, x3.rs:145) and not one a header can describe, but
    // decode_frame itself has no such limits: the reference's reader, one thread (x3_decode_replay.h)
    if (samples > 0xFFFFFFFFull || len > 0xFFFFFFFFull) return X3_ERR_BAD_ARG;
    X3DevParams dpr;
    x3_params pr = *p;
    if (pr.block_len == 0) pr.block_len = 1;
    if ((rc = derive(&pr, 0, &dpr))) return rc;
    if (bl0) dpr.block_len = 0;
    if ((rc = ensure(c, c->in, len + 16))) return rc;
    if ((rc = ensure(c, c->out, (samples + 16) * sizeof(int16_t)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->in.p, payload, len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(x3_replay_one_kernel, dim3(1), dim3(64), 0, c->stream, (const uint8_t*)c->in.p, (uint32_t)len,
                       (uint32_t)samples, dpr, (int16_t*)c->out.p, (int32_t*)c->d_crc);
    HIPCHK(c, hipGetLastError());
    int32_t st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, c->d_crc, sizeof st, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st != X3D_OK) return st;
    HIPCHK(c, hipMemcpyAsync(wav, c->out.p, samples * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_out) *n_out = samples;
    return X3_OK;
  }
  if (c->fcache && use_cache) {  // a frame of the stream announced with x3_decode_prefetch: decoded ahead, a window at a time
    rc = frame_cache_serve(c, payload, len, p, samples, wav);
    if (rc == X3_OK) {
      if (n_out) *n_out = samples;
      return X3_OK;
    }
    if (rc != X3_FRAME_CACHE_MISS) return rc;
  }
  if ((rc = ensure(c, c->in, 20 + len + 16))) return rc;
  if ((rc = ensure(c, c->frame_off, 2 * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->wav_off, sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->out, (samples + 16) * sizeof(int16_t)))) return rc;
  HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in.p + 20, payload, len, hipMemcpyHostToDevice, c->stream));
  uint16_t pcrc = 0;
  if ((rc = crc_dev_async(c, (const uint8_t*)c->in.p + 20, len))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  pcrc = *c->h_crc;
  uint8_t hdr[20];
  x3_write_frame_header(samples, 1, len, pcrc, hdr);
  const uint64_t zero = 0;
  HIPCHK(c, hipMemcpyAsync(c->in.p, hdr, 20, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->frame_off.p, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->wav_off.p, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
  x3_params pp = *p;
  if (pp.block_len == 0) pp.block_len = 1;
  if ((rc = decode_dev_impl(c, (const uint8_t*)c->in.p, 20 + len, (const uint64_t*)c->frame_off.p, 1, nullptr,
                            (const uint64_t*)c->wav_off.p, &pp, (int16_t*)c->out.p, samples, nullptr)))
    return rc;
  uint64_t first_bad = 0, before = 0;
  int bad_status = 0;
  if ((rc = x3_decode_result(c, &first_bad, &bad_status, &before))) return rc;
  if (first_bad == 0) return bad_status;
  HIPCHK(c, hipMemcpyAsync(wav, c->out.p, samples * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_out) *n_out = samples;
  return X3_OK;
}

extern "C" int x3_decode_frame(x3_ctx* c, const uint8_t* payload, uint64_t len, int16_t* wav, uint64_t wav_cap,
                               const x3_params* p, uint64_t samples, uint64_t* n_out) {
  return decode_frame_impl(c, payload, len, wav, wav_cap, p, samples, n_out, true);
}

// ------------------------------------------------------------------------------------------------
// synthetic inputs + device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int x3_synth(int kind, uint64_t seed, uint64_t start, uint64_t n, int16_t* out) {
  if (kind < 0 || kind > 4 || (!out && n)) return X3_ERR_BAD_ARG;
  const uint64_t end = start + n;
  for (uint64_t seg = start / X3_SYNTH_SEG; seg * X3_SYNTH_SEG < end; ++seg) {
    const uint64_t seg_lo = seg * X3_SYNTH_SEG, seg_hi = seg_lo + X3_SYNTH_SEG;
    const uint32_t lo = start > seg_lo ? (uint32_t)(start - seg_lo) : 0u;
    const uint32_t hi = end < seg_hi ? (uint32_t)(end - seg_lo) : X3_SYNTH_SEG;
    int16_t* o = out + (seg_lo + lo - start);
    x3_synth_segment(kind, seed, seg, lo, hi, [o](uint32_t i, int16_t v) { o[i] = v; });
  }
  return X3_OK;
}

extern "C" int x3_synth_dev(x3_ctx* c, int kind, uint64_t seed, uint64_t start, uint64_t n, int16_t* d_out) {
  if (!c || kind < 0 || kind > 4 || (!d_out && n)) return X3_ERR_BAD_ARG;
  if (n == 0) return X3_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t first_seg = start / X3_SYNTH_SEG, last_seg = (start + n - 1) / X3_SYNTH_SEG;
  const uint64_t nseg = last_seg - first_seg + 1;
  hipLaunchKernelGGL(x3_synth_kernel, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, c->stream, kind, seed, start,
                     n, d_out);
  HIPCHK(c, hipGetLastError());
  return X3_OK;
}

extern "C" int x3_dev_alloc(x3_ctx* c, uint64_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(d_ptr, bytes ? bytes : 256));
  return X3_OK;
}
extern "C" int x3_dev_free(x3_ctx* c, void* d_ptr) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipFree(d_ptr));
  return X3_OK;
}
extern "C" int x3_dev_upload(x3_ctx* c, void* d_dst, const void* src, uint64_t bytes) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}
extern "C" int x3_dev_download(x3_ctx* c, void* dst, const void* d_src, uint64_t bytes) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

#ifdef X3_DBG_STAMPS
extern "C" int x3_dbg_read(x3_ctx* c, unsigned long long* out, uint64_t n) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_dbg), n * sizeof(unsigned long long)));
  return X3_OK;
}
#endif

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
#include "x3_bits.h"
#include "x3_mgpu.h"
#include "x3_mc.h"
