// x3_ctx.hip -- context, options, kernel timers and launch log, x3.rs parameters, frame headers, CRC-16 entry points,
// synthetic signals and device memory helpers of libx3hip.so (C ABI: include/x3hip.h; units: x3_internal.h).
#include "x3_internal.h"
#include "x3_tables.h"        // sizes and layouts of the kernels' constant tables (built here, at context creation)
#include "x3_synth_core.h"
#include "x3_util_kernels.h"

int ensure(x3_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return X3_OK;
  if (c->capturing) {   // (no allocation inside a stream capture: the same calls once outside it size everything)
    c->last_error = "x3_graph: a device buffer would have to grow while the calls are being recorded -- make the same calls once before x3_graph_begin";
    return X3_ERR_BAD_ARG;
  }
  if (b.p) HIPCHK(c, x3_dfree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = std::max(bytes, (size_t)4096);
  want = (want + 255) & ~(size_t)255;
  HIPCHK(c, x3_dmalloc(&b.p, want));
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
  o->decode_blocks = std::getenv("X3HIP_DECODE_BLOCKS") ? 1 : 0;
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
#endif
  o->dyn_lds = (int)std::max(0ll, geti("X3HIP_DECODE_DYN_LDS", 0));
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
  // X3HIP_SPIN_WAIT=1: the process's waits for the GPU spin instead of sleeping (hipDeviceScheduleSpin; process-wide, and
  // only when this is the process's first use of the device): calls that end with a trip to the host -- x3_*_result,
  // x3_decode_stream_dev -- come back a few microseconds sooner, at the price of a busy core while they wait
  // (the value is parsed like every other X3HIP_* option: "0" and "" leave the default; a refusal -- the device is already
  // in use with other flags -- is cleared, so that it does not surface as the error of the next kernel launch)
  if (const char* sw = std::getenv("X3HIP_SPIN_WAIT"); sw && std::atoi(sw) != 0) {
    if (hipSetDeviceFlags(hipDeviceScheduleSpin) != hipSuccess) {
      (void)hipGetLastError();
      if (c->opt.verbose) std::fprintf(stderr, "x3hip: X3HIP_SPIN_WAIT: hipSetDeviceFlags(hipDeviceScheduleSpin) refused\n");
    }
  }
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
  HIPCHK(c, x3_dmalloc(&c->d_xpow, X3_XP_SIZE * sizeof(uint16_t)));
  // status (4 ints) and stats (6 + end_pos) share one 128-byte block: one memset, one copy back
  HIPCHK(c, x3_dmalloc(&c->d_ctl_base, 256));
  HIPCHK(c, hipMemset(c->d_ctl_base, 0, 256));
  c->d_status = c->d_ctl_base;
  c->d_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_status) + 32);
  c->d_end_pos = c->d_stats + 6;
  HIPCHK(c, x3_dmalloc(&c->d_summary, sizeof(X3DecodeSummary)));
  HIPCHK(c, x3_dmalloc(&c->d_pace, X3_PACE_WORDS * sizeof(uint32_t)));
  HIPCHK(c, hipMemset(c->d_pace, 0, X3_PACE_WORDS * sizeof(uint32_t)));
  HIPCHK(c, x3_dmalloc(&c->d_crc, 16));
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
      HIPCHK(c, x3_dmalloc(&c->d_xk2, k2.size() * sizeof(uint32_t)));
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
      HIPCHK(c, x3_dmalloc(&c->d_kx64, kx.size() * sizeof(uint32_t)));
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
      HIPCHK(c, x3_dmalloc(&c->d_chktab, ct.size() * sizeof(uint16_t)));
      HIPCHK(c, hipMemcpy(c->d_chktab, ct.data(), ct.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      // x^(-8k): x has order 32767 modulo P (P = (x + 1) * a primitive polynomial of degree 15)
      std::vector<uint16_t> xi(X3_CHECK_XINV_N);
      for (uint32_t k = 0; k < X3_CHECK_XINV_N; ++k) xi[k] = (uint16_t)gf_xpow_host((32767ull * 8 - 8ull * k) % 32767ull);
      HIPCHK(c, x3_dmalloc(&c->d_xinv8, xi.size() * sizeof(uint16_t)));
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
      HIPCHK(c, x3_dmalloc(&c->d_wtab, X3W_TAB_BYTES));
      HIPCHK(c, hipMemcpy(c->d_wtab, wt.data(), X3W_TAB_BYTES, hipMemcpyHostToDevice));
    }
    HIPCHK(c, x3_dmalloc(&c->d_crctab, tab.size() * sizeof(uint16_t)));
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
                    &c->idx_L, &c->idx_sum, &c->idx_wg, &c->idx_sorted, &c->idx_scan, &c->dense_list, &c->lb_desc, &c->src_tab})
    if (b->p) (void)x3_dfree(b->p);
  for (auto& t : c->timers) {
    for (auto& e : t.used) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : t.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
  (void)x3_dfree(c->d_xpow);
  (void)x3_dfree(c->d_xk2);
  (void)x3_dfree(c->d_wtab);
  (void)x3_dfree(c->d_crctab);
  (void)x3_dfree(c->d_kx64);
  (void)x3_dfree(c->d_chktab);
  (void)x3_dfree(c->d_xinv8);
  (void)x3_dfree(c->d_ctl_base);
  (void)x3_dfree(c->d_summary);
  (void)x3_dfree(c->d_pace);
  (void)x3_dfree(c->d_crc);
  (void)hipHostFree(c->h_status);
  if (c->h_walk) (void)hipHostFree(c->h_walk);
  (void)hipHostFree(c->h_summary);
  (void)hipHostFree(c->h_summary_init);
  if (c->h_src_tab) (void)hipHostFree(c->h_src_tab);
  if (c->ev_src_tab) (void)hipEventDestroy(c->ev_src_tab);
  (void)hipHostFree(c->h_crc);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
  if (c->ul_stream) (void)hipStreamDestroy(c->ul_stream);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ------------------------------------------------------------------------------------------------
// HIP graphs: a launch-bound sequence of device calls recorded once, replayed with one host call
// ------------------------------------------------------------------------------------------------
struct x3_graph {
  x3_ctx* c = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // what the result calls need to know about the recorded calls
  bool encode_pending = false, decode_pending = false;
  uint64_t dec_frames = 0, enc_start_pos = 0;
  int32_t* dec_status_ptr = nullptr;
  int last_enc_gen = 0, last_seg_stretches = 0;
  int ctl_half = 0;   // the control block the recorded encode uses (x3_encode_result reads it)
  x3_ctx::LastEnc last_enc{};
};

extern "C" int x3_graph_begin(x3_ctx* c) {
  if (!c || c->capturing) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  // (nothing of the context changes unless the capture has begun: ADVICE r5)
  HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  c->encode_pending = c->decode_pending = false;
  c->timing_before_capture = c->timing;
  c->timing = false;
  c->capturing = true;
  return X3_OK;
}

extern "C" int x3_graph_end(x3_ctx* c, x3_graph** out) {
  if (!c || !c->capturing || !out) return X3_ERR_BAD_ARG;
  *out = nullptr;
  c->capturing = false;
  c->timing = c->timing_before_capture;
  hipGraph_t g = nullptr;
  HIPCHK(c, hipStreamEndCapture(c->stream, &g));
  if (!g) { c->last_error = "x3_graph_end: the capture was invalidated"; return X3_ERR_HIP; }
  hipGraphExec_t ex = nullptr;
  const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    c->last_error = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
    return X3_ERR_HIP;
  }
  x3_graph* r = new (std::nothrow) x3_graph;
  if (!r) { (void)hipGraphExecDestroy(ex); (void)hipGraphDestroy(g); return X3_ERR_HIP; }
  r->c = c;
  r->graph = g;
  r->exec = ex;
  r->encode_pending = c->encode_pending;
  r->decode_pending = c->decode_pending;
  r->dec_frames = c->dec_frames;
  r->enc_start_pos = c->enc_start_pos;
  r->dec_status_ptr = c->dec_status_ptr;
  r->last_enc_gen = c->last_enc_gen;
  r->last_seg_stretches = c->last_seg_stretches;
  r->last_enc = c->last_enc;
  r->ctl_half = c->ctl_half;
  c->encode_pending = c->decode_pending = false;   // (nothing has run yet: x3_graph_launch makes them pending)
  *out = r;
  return X3_OK;
}

extern "C" int x3_graph_launch(x3_ctx* c, x3_graph* g) {
  if (!c || !g || g->c != c || c->capturing) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipGraphLaunch(g->exec, c->stream));
  c->ctl_clean[0] = c->ctl_clean[1] = false;   // (the recorded encoders wrote into the control blocks of their recording)
  c->encode_pending = g->encode_pending;
  c->decode_pending = g->decode_pending;
  c->dec_frames = g->dec_frames;
  c->enc_start_pos = g->enc_start_pos;
  c->dec_status_ptr = g->dec_status_ptr;
  c->last_enc_gen = g->last_enc_gen;
  c->last_seg_stretches = g->last_seg_stretches;
  c->last_enc = g->last_enc;
  c->ctl_half = g->ctl_half;
  c->d_status = c->d_ctl_base + 32 * c->ctl_half;
  c->d_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_status) + 32);
  c->d_end_pos = c->d_stats + 6;
  return X3_OK;
}

extern "C" void x3_graph_destroy(x3_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
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
  else if (n == "decode_blocks") c->opt.decode_blocks = value != 0;
  else if (n == "decode_blocks_off") c->opt.decode_blocks_off = value != 0;
  else if (n == "enc_gen") { c->opt.enc_gen = value == 2 ? 2 : 3; c->prefer_gen2 = false; }
  else if (n == "wave_nwg") c->opt.wave_nwg = (int)std::max(0ll, std::min(256ll, value));
  else if (n == "wave_m") c->opt.wave_m = (int)std::max(0ll, std::min(16ll, value));
  else if (n == "wave_drop") c->opt.wave_drop = value;
  else if (n == "host_walk") c->opt.host_walk = value < 0 ? -1 : (value != 0);
  else if (n == "host_chunk_frames") c->opt.host_chunk_frames = std::max(-1ll, value);
  else if (n == "lb_drop") c->opt.lb_drop = value;
  else if (n == "verbose") c->opt.verbose = value != 0;
  else if (n == "file_chunk_frames") c->opt.file_chunk_frames = std::max(1ll, value);
  else if (n == "file_workers") c->opt.file_workers = (int)std::max(1ll, std::min(16ll, value));
  else if (n == "reader_window_frames") c->opt.reader_window_frames = std::max(1ll, value);
  else if (n == "kernel_timing_mask") c->timing_mask = (uint32_t)value;   // bit k: kernel id k carries events while timing is enabled (default: all)
  else if (n == "check_main") c->opt.check_main = value != 0;
  else if (n == "check_first") c->opt.check_first = value != 0;
  else if (n == "check_wgs") c->opt.check_wgs = (int)std::max(1ll, value);
  else if (n == "seg_stretches") c->opt.seg_stretches = (int)std::max(0ll, value);
  else if (n == "two_trips") c->opt.two_trips = value != 0;
  else if (n == "mc_decode_threads") c->opt.mc_decode_threads = value != 0;
  else if (n == "index_no_fast") c->opt.index_no_fast = value != 0;
  else if (n == "wav_offsets_x4") c->opt.wav_offsets_x4 = value != 0;
  else return X3_ERR_BAD_ARG;
  return X3_OK;
}

extern "C" int x3_ctx_get_option(const x3_ctx* c, const char* name, long long* value) {
  if (!c || !name || !value) return X3_ERR_BAD_ARG;
  const std::string n(name);
  if (n == "two_pass") *value = c->opt.two_pass;
  else if (n == "stream_wgs") *value = c->opt.stream_wgs;
  else if (n == "decode_single") *value = c->opt.decode_single;
  else if (n == "decode_blocks") *value = c->opt.decode_blocks;
  else if (n == "decode_blocks_off") *value = c->opt.decode_blocks_off;
  else if (n == "decode_kernel_in_use") *value = c->last_decode_kernel;   // read-only: 3 = block per lane (round 6), 2 = three waves per 64 frames, 1 = single wave (fast), 0 = single wave (general)
  else if (n == "enc_gen") *value = c->opt.enc_gen;
  else if (n == "encode_dense_reruns") *value = 0;  // (rounds 2-3: whole calls encoded again for a dense frame; no longer happens)
  else if (n == "encode_dense_frames") *value = (long long)c->encode_dense_frames;  // read-only: frames the dense pass has written
  else if (n == "last_dense_frames") *value = (long long)c->last_dense_frames;      // read-only: of the last call (after x3_encode_result)
  else if (n == "enc_gen_in_use") *value = c->last_enc_gen;                          // read-only: 3 = wave encoder, 2 = second generation, 1 = the general kernel in one pass (look-back), 0 = two passes
  else if (n == "host_walk") *value = c->opt.host_walk;
  else if (n == "host_chunk_frames") *value = c->opt.host_chunk_frames;
  else if (n == "verbose") *value = c->opt.verbose;
  else if (n == "file_chunk_frames") *value = c->opt.file_chunk_frames;
  else if (n == "file_workers") *value = c->opt.file_workers;
  else if (n == "reader_window_frames") *value = c->opt.reader_window_frames;
  else if (n == "check_main") *value = c->opt.check_main;
  else if (n == "decode_pace" || n == "encode_pace") {  // (read-only, syncs) the pace words: the decoder's in shader clocks per 16 blocks, the encoder's in 10 ns ticks per frame
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(w, c->d_pace, sizeof w, hipMemcpyDeviceToHost) != hipSuccess)
      return X3_ERR_HIP;
    // (the decoder keeps one word per launch parity: the newer one carries the larger epoch tag)
    *value = (long long)((n == "encode_pace" ? w[4] : std::max(w[0], w[1])) & 0xFFFFFu);
  }
  else if (n == "check_first") *value = c->opt.check_first;
  else if (n == "check_wgs") *value = c->opt.check_wgs;
  else if (n == "seg_stretches") *value = c->opt.seg_stretches;
  else if (n == "two_trips") *value = c->opt.two_trips;
  else if (n == "stream_one_trip") *value = (long long)c->stream_one_trip;   // read-only counter
  else if (n == "last_seg_stretches") *value = c->last_seg_stretches;   // read-only: how the last decode launch used its segment index
  else if (n == "mc_decode_threads") *value = c->opt.mc_decode_threads;
  else if (n == "index_no_fast") *value = c->opt.index_no_fast;
  else if (n == "wav_offsets_x4") *value = c->opt.wav_offsets_x4;
  else if (n == "index_fast_walks") *value = (long long)c->index_fast;        // read-only counters
  else if (n == "index_general_walks") *value = (long long)c->index_general;
  else if (n == "check_prio") *value = c->opt.check_prio;
  else if (n == "encode_fallbacks") *value = (long long)c->encode_fallbacks;  // read-only counter
  else if (n == "encode_needed_pos") *value = (long long)c->needed_pos;       // read-only: where the last host-buffer encode that ran out of room would have ended
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
      // (the decoder's pace words are in shader clocks since round 5; the log hands out 10 ns ticks as before, converted
      // with the clock the launch itself measured: e[2] shader clocks in e[3] ticks)
      out[4 * n + 0] = which ? (uint32_t)((unsigned long long)(e[1] & 0xFFFFFu) * e[3] / std::max<uint32_t>(e[2], 1u)) : 0u;
      out[4 * n + 1] = which ? (uint32_t)((unsigned long long)(e[0] & 0xFFFFFu) * e[3] / std::max<uint32_t>(e[2], 1u)) : 0u;
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
extern const uint32_t X3_RICE_OFFSET[4] = {6, 11, 20, 28};
extern const uint32_t X3_RICE_LEN[4] = {14, 22, 40, 56};
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

uint64_t spf_of(const x3_params* p) { return (uint64_t)p->block_len * (uint64_t)p->blocks_per_frame; }

// worst-case payload bytes of a frame of n samples: every block literal
uint64_t max_payload_bytes(uint64_t n, uint32_t block_len) {
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

int derive(const x3_params* p, uint64_t spf, X3DevParams* d) {
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

uint16_t header_crc16_host(const uint8_t* b, size_t n) {  // headers only (16 bytes)
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
extern "C" int x3_read_frame_header(const uint8_t* b, uint64_t len, x3_frame_header* h) {
  return read_frame_header_ch(b, len, h, 1u);
}
int read_frame_header_ch(const uint8_t* b, uint64_t len, x3_frame_header* h, uint32_t n_ch) {
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
int crc_dev_async(x3_ctx* c, const uint8_t* d_data, uint64_t n) {
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

int x3_pipe_streams(x3_ctx* c) {
  if (!c->dl_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
  if (!c->ul_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->ul_stream, hipStreamNonBlocking));
  return X3_OK;
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
  HIPCHK(c, x3_dmalloc(d_ptr, bytes ? bytes : 256));
  return X3_OK;
}
extern "C" int x3_dev_free(x3_ctx* c, void* d_ptr) {
  if (!c) return X3_ERR_BAD_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, x3_dfree(d_ptr));
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


