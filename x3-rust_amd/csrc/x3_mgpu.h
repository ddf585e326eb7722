// x3_mgpu.h -- frames sharded over the GPUs of one node, RCCL over xGMI (no reference analogue:
// the crate is single-threaded).
//
// Frames are independent in both directions (each re-seeds the predictor with a raw sample and carries its
// own CRCs: encoder.rs:189, decoder.rs:42-46) and every frame is 20 + even bytes, so GPU g encodes a contiguous
// range of whole frames into its own sub-stream and the sub-streams concatenate without padding.  The only
// coupling is each sub-stream's byte offset = the exclusive scan of the sub-stream lengths:
//   1. ncclAllGather of the lengths (8 bytes per rank) -> every rank knows every offset;
//   2. optional reassembly on one rank: ncclGroupStart; root: ncclRecv(dst + off_r, len_r) from every peer,
//      peers: ncclSend(sub, len, root); ncclGroupEnd -- each peer uses its own xGMI link to the root
//      (7 links x ~153 GB/s), where a ring all-gather would be bound by one link and move world x the data.
// Decoding shards the same way by frame index once the header chain has been walked; samples stay where
// they were decoded (or go straight to the caller's host buffer): no collective.
//
// Two forms behind the C ABI (include/x3hip.h):
//   x3_shard_*  one rank of a group (one process or thread per GPU, ncclCommInitRank with a shared id):
//               what bench.py drives under torch.distributed.run, everything device-resident;
//   x3_mgpu_*   all GPUs from one process: one context + one host thread per device, host buffers in and
//               out, bytes identical to x3_encode / x3_decode_stream.
// librccl is opened at run time (dlopen): single-GPU users never load it, and a process that already holds
// PyTorch's copy (same SONAME) keeps exactly one RCCL.
#pragma once
#include <dlfcn.h>
#include <unistd.h>
#include <cerrno>
#include <pthread.h>
#include <rccl/rccl.h>  // types and prototypes only; the entry points come from dlsym

#include <atomic>
#include <mutex>
#include <thread>

struct X3Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommSplit) CommSplit = nullptr;  // optional (RCCL >= 2.18): a communicator of its own for the reassembly
  decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: teardown of a half-built group
  std::string err;
};
static X3Rccl& x3_rccl_state() {
  static X3Rccl r;
  return r;
}
static X3Rccl* x3_rccl();
// why librccl could not be used (set once, inside x3_rccl's call_once; only read afterwards: no race between the
// x3_mgpu_parallel threads -- ADVICE r3)
static const std::string& x3_rccl_error() {
  (void)x3_rccl();
  return x3_rccl_state().err;
}

static X3Rccl* x3_rccl() {
  X3Rccl& r = x3_rccl_state();
  static std::once_flag once;
  std::call_once(once, [&r] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) {
      r.err = std::string("cannot open librccl: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(r.h, n);
      if (!p) { ok = false; r.err = std::string("librccl lacks ") + n; }
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.h); r.h = nullptr; }
    if (r.h) {
      r.CommSplit = reinterpret_cast<decltype(r.CommSplit)>(dlsym(r.h, "ncclCommSplit"));
      r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.h, "ncclCommAbort"));
    }
  });
  return r.h ? &r : nullptr;
}

struct x3_shard {
  x3_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  unsigned long long* d_mine = nullptr;     // this rank's length, when it is not already on the device
  unsigned long long* d_lengths = nullptr;  // [world]
  unsigned long long* h_lengths = nullptr;  // pinned mirror
  // the overlapped reassembly (x3_shard_gather_async): a stream and a communicator of its own, so that neither the
  // context's kernels nor its length exchanges queue behind a 360 MB transfer
  ncclComm_t gcomm = nullptr;               // ncclCommSplit of comm; comm itself where RCCL has no split
  hipStream_t gstream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  bool gather_pending = false;
  // x3_shard_write_at: two pinned staging buffers (a chunk goes to the file while the next one comes down)
  uint8_t* h_stage[2] = {nullptr, nullptr};
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
};
#define X3_SHARD_STAGE_BYTES (16ull << 20)

#define RCCLCHK(ctx, R, call)                                                                          \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      if (ctx) (ctx)->last_error = std::string(#call) + ": " + ((R)->GetErrorString ? (R)->GetErrorString(r_) : "?"); \
      return X3_ERR_HIP;                                                                               \
    }                                                                                                  \
  } while (0)

// ------------------------------------------------------------------------------------------------
// host arithmetic of the sharding (no GPU involved; tests/test_host_logic.py, tests/host_cpp/test_shard_logic.cpp)
// ------------------------------------------------------------------------------------------------
extern "C" void x3_shard_frame_range(uint64_t n_frames, int rank, int world, uint64_t* first, uint64_t* count) {
  // contiguous ranges of whole frames; the remainder goes to the first ranks, one frame each
  const uint64_t w = world > 0 ? (uint64_t)world : 1, r = rank > 0 ? (uint64_t)rank : 0;
  const uint64_t base = n_frames / w, rem = n_frames % w;
  if (first) *first = r * base + std::min(r, rem);
  if (count) *count = r < w ? base + (r < rem ? 1 : 0) : 0;
}

extern "C" void x3_shard_sample_range(uint64_t n_samples, const x3_params* p, int rank, int world, uint64_t* first,
                                      uint64_t* count) {
  const uint64_t spf = p ? spf_of(p) : 0;
  uint64_t f0 = 0, fc = 0;
  // (frames without the sum n + spf - 1, which wraps for sample counts near 2^64; a rank behind the last frame starts at
  // the end of the samples, so that the ranks' ranges tile [0, n_samples) whatever the arguments -- found by
  // tests/host_cpp/fuzz_host_parsers.cpp)
  const uint64_t F = spf ? n_samples / spf + (n_samples % spf != 0) : 0;
  x3_shard_frame_range(F, rank, world, &f0, &fc);
  const uint64_t lo = f0 >= F ? n_samples : f0 * spf;
  const uint64_t hi = f0 + fc >= F ? n_samples : (f0 + fc) * spf;
  if (first) *first = lo;
  if (count) *count = hi > lo ? hi - lo : 0;
}

// exclusive scan of the sub-stream lengths: starts[r] = byte offset of rank r's sub-stream, starts[world] = total
extern "C" void x3_shard_offsets(const uint64_t* lengths, int world, uint64_t* starts) {
  uint64_t acc = 0;
  for (int r = 0; r < world; ++r) {
    starts[r] = acc;
    acc += lengths[r];
  }
  starts[world] = acc;
}

// ------------------------------------------------------------------------------------------------
// one rank
// ------------------------------------------------------------------------------------------------
extern "C" int x3_shard_unique_id(uint8_t id[X3_SHARD_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == X3_SHARD_ID_BYTES, "ncclUniqueId is 128 bytes");
  X3Rccl* R = x3_rccl();
  if (!R || !id) return X3_ERR_HIP;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return X3_ERR_HIP;
  std::memcpy(id, &u, sizeof u);
  return X3_OK;
}

extern "C" void x3_shard_destroy(x3_shard* s) {
  if (!s) return;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
  }
  X3Rccl* R = x3_rccl();
  if (s->gstream) (void)hipStreamSynchronize(s->gstream);
  if (s->gcomm && s->gcomm != s->comm && R) (void)R->CommDestroy(s->gcomm);
  if (s->comm && R) (void)R->CommDestroy(s->comm);
  if (s->gstream) (void)hipStreamDestroy(s->gstream);
  if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
  if (s->ev_done) (void)hipEventDestroy(s->ev_done);
  for (int k = 0; k < 2; ++k) {
    if (s->h_stage[k]) (void)hipHostFree(s->h_stage[k]);
    if (s->ev_stage[k]) (void)hipEventDestroy(s->ev_stage[k]);
  }
  if (s->d_mine) (void)x3_dfree(s->d_mine);
  if (s->d_lengths) (void)x3_dfree(s->d_lengths);
  if (s->h_lengths) (void)hipHostFree(s->h_lengths);
  delete s;
}

extern "C" int x3_shard_create(x3_ctx* c, const uint8_t id[X3_SHARD_ID_BYTES], int rank, int world, x3_shard** out) {
  if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return X3_ERR_BAD_ARG;
  *out = nullptr;
  X3Rccl* R = x3_rccl();
  if (!R) {
    c->last_error = "librccl is not available: " + x3_rccl_error();
    return X3_ERR_HIP;
  }
  HIPCHK(c, hipSetDevice(c->device));
  x3_shard* s = new x3_shard();
  s->ctx = c;
  s->rank = rank;
  s->world = world;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = R->CommInitRank(&s->comm, world, u, rank);  // blocks until every rank has joined
  if (r != ncclSuccess) {
    c->last_error = std::string("ncclCommInitRank: ") + R->GetErrorString(r);
    s->comm = nullptr;
    x3_shard_destroy(s);
    return X3_ERR_HIP;
  }
  if (x3_dmalloc(&s->d_mine, 16) != hipSuccess || x3_dmalloc(&s->d_lengths, sizeof(uint64_t) * (size_t)world) != hipSuccess ||
      hipHostMalloc(&s->h_lengths, sizeof(uint64_t) * (size_t)world) != hipSuccess ||
      hipStreamCreateWithFlags(&s->gstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming) != hipSuccess) {
    c->last_error = "x3_shard_create: out of memory";
    x3_shard_destroy(s);
    return X3_ERR_HIP;
  }
  // (collective: every rank splits; the same members in the same order).  The ranks then AGREE on the outcome -- one
  // all-gather of a flag on `comm` -- and either all use the split communicator or all fall back to `comm`: a rank that
  // fell back alone would post its sends on another communicator than its peers' receives and hang.
  s->gcomm = s->comm;
  if (R->CommSplit && world > 1) {
    ncclComm_t g2 = nullptr;
    const bool mine = R->CommSplit(s->comm, 0, rank, &g2, nullptr) == ncclSuccess && g2;
    bool all = false;
    s->h_lengths[rank] = mine ? 1u : 0u;
    if (hipMemcpyAsync(s->d_mine, &s->h_lengths[rank], sizeof(uint64_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
        R->AllGather(s->d_mine, s->d_lengths, 1, ncclUint64, s->comm, c->stream) == ncclSuccess &&
        hipMemcpyAsync(s->h_lengths, s->d_lengths, sizeof(uint64_t) * (size_t)world, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
        hipStreamSynchronize(c->stream) == hipSuccess) {
      all = true;
      for (int r2 = 0; r2 < world; ++r2) all = all && s->h_lengths[r2] == 1u;
    }
    if (all) s->gcomm = g2;
    else if (g2) (void)R->CommDestroy(g2);
  }
  *out = s;
  return X3_OK;
}

extern "C" int x3_shard_rank(const x3_shard* s) { return s ? s->rank : -1; }
extern "C" int x3_shard_world(const x3_shard* s) { return s ? s->world : 0; }

// Step 1: all-gather of the sub-stream lengths.  d_len: DEVICE pointer to this rank's length (e.g. the last
// entry of the frame offsets x3_encode_dev wrote, when the sub-stream starts at 0); d_lengths: device array of
// `world` entries, or NULL for the shard's own (then read with x3_shard_lengths).  Enqueued on the context's
// stream behind the encoder; does not synchronise.
extern "C" int x3_shard_exchange_lengths(x3_shard* s, const uint64_t* d_len, uint64_t* d_lengths) {
  if (!s || !d_len) return X3_ERR_BAD_ARG;
  X3Rccl* R = x3_rccl();
  x3_ctx* c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  RCCLCHK(c, R, R->AllGather(d_len, d_lengths ? (void*)d_lengths : (void*)s->d_lengths, 1, ncclUint64, s->comm, c->stream));
  return X3_OK;
}

// the same for a length the host holds
extern "C" int x3_shard_exchange_length_value(x3_shard* s, uint64_t len, uint64_t* d_lengths) {
  if (!s) return X3_ERR_BAD_ARG;
  x3_ctx* c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  s->h_lengths[s->rank] = len;  // pinned: stays valid until the copy has run
  HIPCHK(c, hipMemcpyAsync(s->d_mine, &s->h_lengths[s->rank], sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  return x3_shard_exchange_lengths(s, reinterpret_cast<const uint64_t*>(s->d_mine), d_lengths);
}

// waits for the last exchange into the shard's own array and hands the lengths to the host
extern "C" int x3_shard_lengths(x3_shard* s, uint64_t* lengths) {
  if (!s || !lengths) return X3_ERR_BAD_ARG;
  x3_ctx* c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(s->h_lengths, s->d_lengths, sizeof(uint64_t) * (size_t)s->world, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int r = 0; r < s->world; ++r) lengths[r] = s->h_lengths[r];
  return X3_OK;
}

// Step 2: reassemble the whole stream on `root`: d_dst[starts[r] .. starts[r] + lengths[r]) = rank r's d_sub.
// lengths: HOST array of `world` entries, the same on every rank (x3_shard_lengths).  d_dst only counts on the root;
// dst_cap is the ROOT's capacity and every rank that passes it (non-zero) comes to the same verdict -- a rank that passes
// 0 does not check, and is left waiting in its send if the root then refuses: callers size the destination from the
// lengths before they get here (bench.py, x3_mgpu_encode).  Enqueued on the context's stream; does not synchronise.
// *total = the stream's length.
static int x3_shard_gather_on(x3_shard* s, const uint8_t* d_sub, const uint64_t* lengths, int root, uint8_t* d_dst,
                              uint64_t dst_cap, uint64_t* total, ncclComm_t comm, hipStream_t stream) {
  X3Rccl* R = x3_rccl();
  x3_ctx* c = s->ctx;
  std::vector<uint64_t> starts((size_t)s->world + 1);
  x3_shard_offsets(lengths, s->world, starts.data());
  if (total) *total = starts[s->world];
  if ((s->rank == root || dst_cap) && starts[s->world] > dst_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  if (s->rank == root) {
    if (!d_dst && starts[s->world]) return X3_ERR_BAD_ARG;
    if (lengths[root] && d_sub != d_dst + starts[root])
      HIPCHK(c, hipMemcpyAsync(d_dst + starts[root], d_sub, lengths[root], hipMemcpyDeviceToDevice, stream));
    if (s->world > 1) {
      RCCLCHK(c, R, R->GroupStart());
      for (int r = 0; r < s->world; ++r)
        if (r != root && lengths[r]) {
          ncclResult_t e = R->Recv(d_dst + starts[r], lengths[r], ncclUint8, r, comm, stream);
          if (e != ncclSuccess) { (void)R->GroupEnd(); RCCLCHK(c, R, e); }
        }
      RCCLCHK(c, R, R->GroupEnd());
    }
  } else if (lengths[s->rank]) {
    if (!d_sub) return X3_ERR_BAD_ARG;
    RCCLCHK(c, R, R->GroupStart());
    ncclResult_t e = R->Send(d_sub, lengths[s->rank], ncclUint8, root, comm, stream);
    if (e != ncclSuccess) { (void)R->GroupEnd(); RCCLCHK(c, R, e); }
    RCCLCHK(c, R, R->GroupEnd());
  }
  return X3_OK;
}

extern "C" int x3_shard_gather(x3_shard* s, const uint8_t* d_sub, const uint64_t* lengths, int root, uint8_t* d_dst,
                               uint64_t dst_cap, uint64_t* total) {
  if (!s || !lengths || root < 0 || root >= s->world) return X3_ERR_BAD_ARG;
  HIPCHK(s->ctx, hipSetDevice(s->ctx->device));
  return x3_shard_gather_on(s, d_sub, lengths, root, d_dst, dst_cap, total, s->comm, s->ctx->stream);
}

// The same reassembly beside the context's work: it starts when everything enqueued on the context's stream SO FAR has
// run (the sub-stream is complete) and runs on the shard's own stream and communicator, while the context goes on with
// its next batch -- into another output buffer: d_sub (and the root's d_dst) must stay untouched until
// x3_shard_gather_wait.  One reassembly in flight per shard.
extern "C" int x3_shard_gather_async(x3_shard* s, const uint8_t* d_sub, const uint64_t* lengths, int root, uint8_t* d_dst,
                                     uint64_t dst_cap, uint64_t* total) {
  if (!s || !lengths || root < 0 || root >= s->world) return X3_ERR_BAD_ARG;
  x3_ctx* c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  if (s->gather_pending) {
    c->last_error = "x3_shard_gather_async: the reassembly before this one has not been waited for";
    return X3_ERR_BAD_ARG;
  }
  // (no communicator of its own -- RCCL without ncclCommSplit, or a split that did not succeed on every rank: one
  // communicator must not be driven from two streams at once, so the reassembly then runs on the context's stream,
  // in order with its length exchanges -- ADVICE r3)
  const bool own = s->gcomm != s->comm;
  hipStream_t st = own ? s->gstream : c->stream;
  if (own) {
    HIPCHK(c, hipEventRecord(s->ev_ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(s->gstream, s->ev_ready, 0));
  }
  int rc = x3_shard_gather_on(s, d_sub, lengths, root, d_dst, dst_cap, total, s->gcomm, st);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(s->ev_done, st));
  s->gather_pending = true;
  return X3_OK;
}

// on_stream != 0: the context's stream waits for the reassembly in flight (what is enqueued behind this call may reuse
// the buffers; the host does not block); on_stream == 0: the host waits.  No reassembly in flight: nothing happens.
extern "C" int x3_shard_gather_wait(x3_shard* s, int on_stream) {
  if (!s) return X3_ERR_BAD_ARG;
  x3_ctx* c = s->ctx;
  if (!s->gather_pending) return X3_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (on_stream) HIPCHK(c, hipStreamWaitEvent(c->stream, s->ev_done, 0));
  else HIPCHK(c, hipEventSynchronize(s->ev_done));
  s->gather_pending = false;
  return X3_OK;
}

// Step 2, SHARDED (round 4): nobody takes in the whole stream.  The reference writes one .x3a file through one BufWriter
// (encodefile.rs:66-74); with the frames on N GPUs the equivalent is N writers into ONE file, rank r's sub-stream at byte
// base + starts[r] (x3_shard_offsets; `base` = what precedes the frames, e.g. the archive header).  Every rank passes a
// descriptor of the same file (one rank creates it, the others open it without O_TRUNC).  The root of x3_shard_gather
// takes 2.5 GB per step over seven xGMI links at N = 8 -- five times the ranks' compute; here every rank's bytes leave
// over its OWN host link, side by side.  Starts when everything enqueued on the context's stream so far has run; the
// sub-stream comes down in 16 MiB pieces through two pinned buffers on the shard's own stream, each piece written
// (pwrite) while the next one is on its way.  Returns when this rank's bytes are in the file (not fsync'ed).
extern "C" int x3_shard_write_at(x3_shard* s, const uint8_t* d_sub, const uint64_t* lengths, int fd, uint64_t base,
                                 uint64_t* total) {
  if (!s || !lengths || fd < 0) return X3_ERR_BAD_ARG;
  x3_ctx* c = s->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<uint64_t> starts((size_t)s->world + 1);
  x3_shard_offsets(lengths, s->world, starts.data());
  if (total) *total = starts[s->world];
  const uint64_t len = lengths[s->rank];
  if (len == 0) return X3_OK;
  if (!d_sub) return X3_ERR_BAD_ARG;
  for (int k = 0; k < 2; ++k) {
    if (!s->h_stage[k]) HIPCHK(c, hipHostMalloc(&s->h_stage[k], X3_SHARD_STAGE_BYTES));
    if (!s->ev_stage[k]) HIPCHK(c, hipEventCreateWithFlags(&s->ev_stage[k], hipEventDisableTiming));
  }
  HIPCHK(c, hipEventRecord(s->ev_ready, c->stream));
  HIPCHK(c, hipStreamWaitEvent(s->gstream, s->ev_ready, 0));
  const uint64_t at = base + starts[s->rank];
  const uint64_t pieces = (len + X3_SHARD_STAGE_BYTES - 1) / X3_SHARD_STAGE_BYTES;
  auto piece_len = [&](uint64_t k) { return std::min<uint64_t>(X3_SHARD_STAGE_BYTES, len - k * X3_SHARD_STAGE_BYTES); };
  auto fetch = [&](uint64_t k) -> int {
    HIPCHK(c, hipMemcpyAsync(s->h_stage[k & 1], d_sub + k * X3_SHARD_STAGE_BYTES, piece_len(k), hipMemcpyDeviceToHost, s->gstream));
    HIPCHK(c, hipEventRecord(s->ev_stage[k & 1], s->gstream));
    return X3_OK;
  };
  int rc = fetch(0);
  for (uint64_t k = 0; k < pieces && !rc; ++k) {
    HIPCHK(c, hipEventSynchronize(s->ev_stage[k & 1]));
    if (k + 1 < pieces) rc = fetch(k + 1);  // (into the other buffer, whose last piece has been written)
    const uint8_t* src = s->h_stage[k & 1];
    uint64_t left = piece_len(k), pos = at + k * X3_SHARD_STAGE_BYTES;
    while (left) {
      const ssize_t w = ::pwrite(fd, src, (size_t)left, (off_t)pos);
      if (w < 0) {
        if (errno == EINTR) continue;
        c->last_error = std::string("x3_shard_write_at: pwrite: ") + std::strerror(errno);
        (void)hipStreamSynchronize(s->gstream);
        return X3_ERR_IO;
      }
      src += w; pos += (uint64_t)w; left -= (uint64_t)w;
    }
  }
  if (rc) (void)hipStreamSynchronize(s->gstream);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// all GPUs from one process: one context, one shard and one host thread per device
// ------------------------------------------------------------------------------------------------
struct x3_mgpu {
  std::vector<int> devices;
  std::vector<x3_ctx*> ctx;
  std::vector<x3_shard*> shard;
  DevBuf whole;  // on devices[0]: the reassembled stream
  std::string last_error;
};

template <class Fn>
static void x3_mgpu_parallel(int n, Fn&& fn) {  // fn(g) on n host threads (the caller's is thread 0)
  std::vector<std::thread> th;
  for (int g = 1; g < n; ++g) th.emplace_back([&fn, g] { fn(g); });
  fn(0);
  for (auto& t : th) t.join();
}

extern "C" void x3_mgpu_destroy(x3_mgpu* m) {
  if (!m) return;
  if (!m->ctx.empty() && m->ctx[0] && m->whole.p) {
    (void)hipSetDevice(m->ctx[0]->device);
    (void)x3_dfree(m->whole.p);
  }
  // communicators are torn down by all ranks together
  x3_mgpu_parallel((int)m->shard.size(), [&](int g) { x3_shard_destroy(m->shard[g]); });
  for (x3_ctx* c : m->ctx) x3_ctx_destroy(c);
  delete m;
}

extern "C" int x3_mgpu_create(const int* devices, int n, x3_mgpu** out) {
  if (!devices || n < 1 || n > 64 || !out) return X3_ERR_BAD_ARG;
  *out = nullptr;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (devices[a] == devices[b]) return X3_ERR_BAD_ARG;  // one rank per GPU
  x3_mgpu* m = new x3_mgpu();
  m->devices.assign(devices, devices + n);
  m->ctx.assign((size_t)n, nullptr);
  m->shard.assign((size_t)n, nullptr);
  uint8_t id[X3_SHARD_ID_BYTES];
  int rc = n > 1 ? x3_shard_unique_id(id) : X3_OK;
  if (rc) {
    std::fprintf(stderr, "x3hip: x3_mgpu_create: %s\n", x3_rccl() ? "ncclGetUniqueId failed" : ("librccl is not available: " + x3_rccl_error()).c_str());
    delete m;
    return rc;
  }
  std::vector<int> rcs((size_t)n, X3_OK);
  for (int g = 0; g < n && !rc; ++g) rc = rcs[g] = x3_ctx_create(devices[g], &m->ctx[g]);
  if (!rc && n > 1)  // ncclCommInitRank blocks until every rank has joined: one thread per rank
    x3_mgpu_parallel(n, [&](int g) { rcs[g] = x3_shard_create(m->ctx[g], id, g, n, &m->shard[g]); });
  for (int g = 0; g < n; ++g)
    if (rcs[g]) {
      rc = rcs[g];
      if (m->ctx[g]) std::fprintf(stderr, "x3hip: x3_mgpu_create, device %d: %s\n", devices[g], m->ctx[g]->last_error.c_str());
    }
  if (rc) {
    // a half-built group: the ranks that did join are aborted (a collective destroy would wait for the ones that did
    // not), every rank on its own thread, then their device and pinned buffers go
    X3Rccl* R = x3_rccl();
    x3_mgpu_parallel(n, [&](int g) {
      x3_shard* sh = m->shard[g];
      if (!sh) return;
      if (R && R->CommAbort) {
        if (sh->gcomm && sh->gcomm != sh->comm) (void)R->CommAbort(sh->gcomm);
        if (sh->comm) (void)R->CommAbort(sh->comm);
        sh->gcomm = sh->comm = nullptr;
      }
      x3_shard_destroy(sh);
    });
    m->shard.clear();
    for (x3_ctx* c : m->ctx) if (c) x3_ctx_destroy(c);
    delete m;
    return rc;
  }
  if (n == 1) m->shard.clear();  // a single GPU needs no communicator
  *out = m;
  return X3_OK;
}

extern "C" int x3_mgpu_devices(const x3_mgpu* m) { return m ? (int)m->devices.size() : 0; }
extern "C" x3_ctx* x3_mgpu_ctx(x3_mgpu* m, int g) { return m && g >= 0 && g < (int)m->ctx.size() ? m->ctx[g] : nullptr; }
extern "C" x3_shard* x3_mgpu_shard(x3_mgpu* m, int g) { return m && g >= 0 && g < (int)m->shard.size() ? m->shard[g] : nullptr; }
extern "C" const char* x3_mgpu_last_error(const x3_mgpu* m) { return m ? m->last_error.c_str() : ""; }

// `encoder::encode` (src/encoder.rs:51-111) over all GPUs: same arguments, same bytes, same status as x3_encode.
extern "C" int x3_mgpu_encode(x3_mgpu* m, const int16_t* wav, uint64_t n, uint32_t n_channels, const x3_params* p,
                              uint8_t* out, uint64_t out_cap, uint64_t start_pos, uint64_t* out_pos, uint64_t stats[6]) {
  if (!m || !p || (!wav && n) || (!out && out_cap)) return X3_ERR_BAD_ARG;
  if (n_channels > 1) return X3_ERR_MORE_THAN_ONE_CHANNEL;
  if (n_channels == 0) return X3_ERR_BAD_ARG;
  if (stats) std::memset(stats, 0, 6 * sizeof(uint64_t));
  if (out_pos) *out_pos = start_pos;
  int rc = x3_params_validate(p);
  if (rc == X3_ERR_BAD_ARG) return rc;
  const uint64_t spf = spf_of(p);
  if (spf == 0 || n == 0) return X3_OK;
  if (start_pos > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  const int G = (int)m->ctx.size();
  if (G == 1) return x3_encode(m->ctx[0], wav, n, n_channels, p, out, out_cap, start_pos, out_pos, stats);

  const uint64_t base = (start_pos + 1ull) & ~1ull;  // writer.align::<2>() (encoder.rs:182)
  std::vector<int> rcs((size_t)G, X3_OK);
  std::vector<uint64_t> lens((size_t)G, 0), st((size_t)G * 6, 0);
  std::vector<std::vector<uint64_t>> all((size_t)G, std::vector<uint64_t>((size_t)G, 0));
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, (unsigned)G);
  std::atomic<int> failed{0};
  uint64_t total = 0;
  x3_mgpu_parallel(G, [&](int g) {
    x3_ctx* c = m->ctx[g];
    x3_shard* s = m->shard[g];
    int e = hipSetDevice(c->device) == hipSuccess ? X3_OK : X3_ERR_HIP;
    uint64_t first = 0, cnt = 0;
    x3_shard_sample_range(n, p, g, G, &first, &cnt);
    uint64_t len = 0;
    if (!e && cnt) {
      const uint64_t bound = x3_encode_bound(cnt, p);
      if (!(e = ensure(c, c->in, cnt * sizeof(int16_t) + 16)) && !(e = ensure(c, c->out, bound + 16))) {
        if (hipMemcpyAsync(c->in.p, wav + first, cnt * sizeof(int16_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) e = X3_ERR_HIP;
        x3_batch b{cnt, cnt, 1};
        if (!e) e = encode_dev_impl(c, (const int16_t*)c->in.p, &b, p, spf, (uint8_t*)c->out.p, bound, 0, nullptr);
        if (!e) e = x3_encode_result(c, &len, &st[(size_t)g * 6]);
      }
    }
    if (e) { len = 0; failed.store(1); }
    rcs[g] = e;
    // step 1: every rank learns every length (a rank that failed still takes part: nobody may be left waiting)
    int e2 = x3_shard_exchange_length_value(s, len, nullptr);
    if (!e2) e2 = x3_shard_lengths(s, all[g].data());
    if (e2) { failed.store(1); if (!rcs[g]) rcs[g] = e2; }
    pthread_barrier_wait(&bar);  // B1: every rank has every length -- or has raised `failed`
    uint64_t tot = 0;
    for (int r = 0; r < G; ++r) tot += all[g][r];
    if (g == 0) total = tot;
    const bool over = base + tot > out_cap;  // the same verdict on every rank; reported below
    // step 2, sharded (round 4): the destination is HOST memory, so every device copies its own sub-stream straight to
    // its place in the caller's buffer over its own host link -- no reassembly on devices[0] first (until round 3: all
    // sub-streams over xGMI to one GPU, then one copy down one link).  x3_shard_gather stays for callers who want the
    // whole stream on a GPU.  (Every thread passes BOTH barriers whatever has happened, and looks at `failed` only
    // behind the second: a thread that left between them would leave the others waiting for ever -- ADVICE r2.)
    pthread_barrier_wait(&bar);  // B2
    if (failed.load() || over) return;
    uint64_t at = base;
    for (int r = 0; r < g; ++r) at += all[g][r];
    if (g == 0 && (start_pos & 1ull)) out[start_pos] = 0;
    if (len && hipMemcpy(out + at, c->out.p, len, hipMemcpyDeviceToHost) != hipSuccess) e = X3_ERR_HIP;
    if (e) rcs[g] = e;
  });
  pthread_barrier_destroy(&bar);
  for (int g = 0; g < G; ++g)
    if (rcs[g]) {
      m->last_error = "device " + std::to_string(m->devices[g]) + ": " + m->ctx[g]->last_error;
      return rcs[g];
    }
  if (out_pos) *out_pos = base + total;
  if (base + total > out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  if (stats)
    for (int g = 0; g < G; ++g)
      for (int k = 0; k < 6; ++k) stats[k] += st[(size_t)g * 6 + k];
  return X3_OK;
}

// x3_decode_stream over all GPUs: the header chain is walked once (host), the frames are dealt out in contiguous
// ranges, every GPU decodes its range and copies its samples straight to their place in the caller's buffer.
// Same results and status as x3_decode_stream: nothing behind the first frame that fails is delivered.
extern "C" int x3_mgpu_decode_stream(x3_mgpu* m, const uint8_t* x3, uint64_t len, const x3_params* p, int16_t* wav,
                                     uint64_t wav_cap, uint64_t* n_out, uint64_t* frames_ok, uint64_t* frame_errors) {
  if (!m || !p || (!x3 && len) || (!wav && wav_cap)) return X3_ERR_BAD_ARG;
  if (n_out) *n_out = 0;
  if (frames_ok) *frames_ok = 0;
  if (frame_errors) *frame_errors = 0;
  const int G = (int)m->ctx.size();
  if (G == 1) return x3_decode_stream(m->ctx[0], x3, len, p, wav, wav_cap, n_out, frames_ok, frame_errors);
  HostWalk w;
  walk_host(x3, len, len, len, p, wav_cap, ~0ull, &w);
  const uint64_t F = w.offs.size();
  if (F == 0) return w.terminal;
  struct Part {
    uint64_t a = 0, cnt = 0, before = 0, first_bad = 0;
    int bad_status = 0, rc = X3_OK;
    HostWalk hw;
  };
  std::vector<Part> part((size_t)G);
  x3_mgpu_parallel(G, [&](int g) {
    Part& q = part[g];
    x3_shard_frame_range(F, g, G, &q.a, &q.cnt);
    if (!q.cnt) return;
    x3_ctx* c = m->ctx[g];
    if (hipSetDevice(c->device) != hipSuccess) { q.rc = X3_ERR_HIP; return; }
    const uint64_t b0 = w.offs[q.a], s0 = w.woffs[q.a];
    const uint64_t b1 = q.a + q.cnt < F ? w.offs[q.a + q.cnt] : w.end_pos;
    const uint64_t s1 = q.a + q.cnt < F ? w.woffs[q.a + q.cnt] : w.nsamp;
    for (uint64_t i = 0; i < q.cnt; ++i) {
      q.hw.offs.push_back(w.offs[q.a + i] - b0);
      q.hw.woffs.push_back(w.woffs[q.a + i] - s0);
    }
    q.hw.nsamp = s1 - s0;
    q.rc = decode_frames_host(c, x3 + b0, b1 - b0, q.hw, p, nullptr, ~0ull, &q.before, &q.first_bad, &q.bad_status, false);
  });
  for (int g = 0; g < G; ++g)
    if (part[g].rc) {
      m->last_error = "device " + std::to_string(m->devices[g]) + ": " + m->ctx[g]->last_error;
      return part[g].rc;
    }
  // the first frame that failed, over all ranges, ends the walk
  uint64_t first_bad = F, before = w.nsamp;
  int bad_status = 0, stop = G;
  for (int g = 0; g < G; ++g)
    if (part[g].cnt && part[g].first_bad < part[g].cnt) {
      first_bad = part[g].a + part[g].first_bad;
      before = w.woffs[part[g].a] + part[g].before;
      bad_status = part[g].bad_status;
      stop = g;
      break;
    }
  std::vector<int> rcs((size_t)G, X3_OK);
  x3_mgpu_parallel(G, [&](int g) {
    const Part& q = part[g];
    if (!q.cnt || g > stop) return;
    x3_ctx* c = m->ctx[g];
    const uint64_t take = g == stop ? q.before : q.hw.nsamp;
    if (hipSetDevice(c->device) != hipSuccess) { rcs[g] = X3_ERR_HIP; return; }
    if (take && hipMemcpy(wav + w.woffs[q.a], c->out.p, take * sizeof(int16_t), hipMemcpyDeviceToHost) != hipSuccess) rcs[g] = X3_ERR_HIP;
  });
  for (int g = 0; g < G; ++g)
    if (rcs[g]) return rcs[g];
  if (n_out) *n_out = before;
  if (frames_ok) *frames_ok = first_bad;
  return walk_result(F, first_bad, bad_status, w.terminal, frame_errors);
}
