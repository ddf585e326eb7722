// pair_probe.hip -- does a bare kernel with the decoder's memory pattern see the PAIR of buffers it runs on?
//   hipcc --offload-arch=gfx950 -O3 -o pair_probe pair_probe.hip && ./pair_probe [n_candidates]
// profiles/r6/decoder_modes.txt: the decode phase's pace follows the pair (stream buffer, sample buffer) by up to 10 %.
// Here, without any decoding: 1 080 waves, a frame per lane -- every lane reads its frame's 5 264 bytes of A in 16-byte
// chunks and the wave writes its 64 rows of B (20 000 bytes apart) as whole 128-byte lines, eight rows per store
// instruction (x3_decode_split_kernel.h's flusher), 2.1 chunks read per line written; beside it, optionally, a reader of
// all of A (a wave per frame: the check kernel's pattern) on a second stream.  Variants: lines of 256 / 512 bytes per row
// and store, plain instead of non-temporal stores, reads only, writes only.  Prints ms per (A, B) pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>
#define FRAMES 69120u
#define FB 5264u      // stream bytes per frame (16-byte multiple near config 3's 5 257)
#define ROW 20000u    // sample bytes per frame
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// LINE: bytes a row gets per store (128: the decoder's); NT: non-temporal stores; RD / WR: do the reads / the writes
template <int LINE, bool NT, bool RD, bool WR>
__global__ void __launch_bounds__(64) decoder_like(const uint8_t* __restrict__ A, uint8_t* __restrict__ B, uint32_t delay) {
  const uint32_t lane = threadIdx.x;
  const uint32_t f0 = blockIdx.x * 64u;
  const u32x4* src = reinterpret_cast<const u32x4*>(A + (size_t)(f0 + lane) * FB);
  uint8_t* dst0 = B + (size_t)f0 * ROW;
  constexpr uint32_t LPS = LINE / 16u;           // lanes per row and store
  constexpr uint32_t RPS = 64u / LPS;            // rows per store instruction
  const uint32_t nsteps = (ROW - LINE) / LINE;   // steps of LINE bytes per row
  u32x4 acc = {lane, 1u, 2u, 3u};
  uint32_t c = 0;                                 // chunks read, as 16.16: FB / ROW chunks of 16 B per 16 B written
  const uint32_t cstep = (uint32_t)(((unsigned long long)FB << 16) / ROW) * (LINE / 16u);
  for (uint32_t k = 0; k < nsteps; ++k) {
    if (RD) {
      const uint32_t c1 = c + cstep;
      for (uint32_t i = c >> 16; i < (c1 >> 16); ++i) {
        const u32x4 v = src[i];
        acc ^= v;
      }
      c = c1;
    }
    // (a dependent chain between the steps, as the parser's: the kernel is not bandwidth-bound)
    for (uint32_t d = 0; d < delay * (LINE / 128u); ++d) acc.x = acc.x * 1664525u + 1013904223u;
    if (WR) {
#pragma unroll
      for (uint32_t r0 = 0; r0 < 64u; r0 += RPS) {
        const uint32_t r = r0 + lane / LPS, p = lane % LPS;
        // (whole 128-byte lines, as the flusher writes them: a row's pieces are cut at the destination's line boundaries)
        u32x4* d = reinterpret_cast<u32x4*>(B + ((((size_t)(f0 + r) * ROW) & ~(size_t)127) + (size_t)k * LINE + 16u * p));
        if (NT) __builtin_nontemporal_store(acc, d); else *d = acc;
      }
    }
  }
  if (!WR && acc.x == 0x12345u) B[0] = 1;
}
// a wave per frame reads the whole frame (256 threads = 4 frames a workgroup), grid-stride over the frames
__global__ void __launch_bounds__(256) reader_like(const uint8_t* __restrict__ A, uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  for (uint32_t f = blockIdx.x * 4u + w; f < FRAMES; f += gridDim.x * 4u) {
    const u32x4* src = reinterpret_cast<const u32x4*>(A + (size_t)f * FB);
    for (uint32_t i = lane; i < FB / 16u; i += 64u) acc ^= src[i];
  }
  if (acc.x == 0x12345u) sink[0] = 1;
}

template <int LINE, bool NT, bool RD, bool WR>
static float run(const uint8_t* A, uint8_t* B, uint32_t delay, bool with_reader, uint32_t* sink, hipStream_t s0, hipStream_t s1) {
  hipEvent_t a, b, fork, join;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
  float tot = 0;
  const int reps = 6;
  for (int rep = -2; rep < reps; ++rep) {
    CK(hipEventRecord(fork, s0));
    CK(hipStreamWaitEvent(s1, fork, 0));
    CK(hipEventRecord(a, s0));
    hipLaunchKernelGGL((decoder_like<LINE, NT, RD, WR>), dim3(FRAMES / 64u), dim3(64), 0, s0, A, B, delay);
    if (with_reader) hipLaunchKernelGGL(reader_like, dim3(1024), dim3(256), 0, s1, A, sink);
    CK(hipEventRecord(join, s1));
    CK(hipStreamWaitEvent(s0, join, 0));
    CK(hipEventRecord(b, s0));
    CK(hipStreamSynchronize(s0));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 0) tot += ms;
  }
  return tot / reps;
}

// `chunks` mode: which PIECES of memory are of which kind?  N allocations of S megabytes, the write-only pattern folded into
// each (the same 1 080 waves and 64-row stores, addresses modulo the piece), in allocation order.
__global__ void __launch_bounds__(64) write_folded(uint8_t* __restrict__ B, uint32_t piece_lines) {
  const uint32_t lane = threadIdx.x;
  const uint32_t f0 = blockIdx.x * 64u;
  const uint32_t nsteps = (ROW - 128u) / 128u;
  u32x4 acc = {lane, 1u, 2u, 3u};
  for (uint32_t k = 0; k < nsteps; ++k) {
#pragma unroll
    for (uint32_t r0 = 0; r0 < 64u; r0 += 8u) {
      const uint32_t r = r0 + lane / 8u, p = lane % 8u;
      const size_t line = ((((size_t)(f0 + r) * ROW) >> 7) + k) % piece_lines;
      __builtin_nontemporal_store(acc, reinterpret_cast<u32x4*>(B + line * 128u + 16u * p));
    }
  }
}
static int chunks_mode(int n, size_t mb) {
  const size_t bytes = mb << 20;
  std::vector<uint8_t*> c;
  for (int i = 0; i < n; ++i) {
    uint8_t* p;
    CK(hipMalloc(&p, bytes));
    CK(hipMemset(p, 0, bytes));
    c.push_back(p);
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::printf("%d pieces of %zu MB in allocation order: ms of the folded write-only pattern (1.38 GB written into each)\n", n, mb);
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < n; ++i) {
      float tot = 0;
      for (int r = -1; r < 3; ++r) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(write_folded, dim3(FRAMES / 64u), dim3(64), 0, 0, c[i], (uint32_t)(bytes >> 7));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 0) tot += ms;
      }
      std::printf(" %.3f", tot / 3);
      if (i % 16 == 15) std::printf("\n");
    }
    std::printf("\n");
  }
  for (int i = 0; i < n; ++i) std::printf(" %p", (void*)c[i]);
  std::printf("\n");
  return 0;
}

// `sizes` mode: does the SIZE an allocation is made with decide its kind?  Eight allocations of each size (the 1.38 GB are
// written at their start), the write-only pattern on each.
static int sizes_mode() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  uint32_t* sink;
  uint8_t* a;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&a, (size_t)FRAMES * FB + 4096));
  const size_t need = (size_t)FRAMES * ROW + 4096;
  const size_t sizes[] = {need, (size_t)1536 << 20, (size_t)2048 << 20, (size_t)3072 << 20, (size_t)4096 << 20, need};
  for (size_t sz : sizes) {
    std::printf("allocations of %zu MB:", sz >> 20);
    std::vector<uint8_t*> keep;
    for (int i = 0; i < 8; ++i) {
      uint8_t* b;
      CK(hipMalloc(&b, sz));
      CK(hipMemset(b, 0, need));
      keep.push_back(b);
      std::printf(" %.3f", run<128, true, false, true>(a, b, 0, false, sink, s0, s1));
      std::fflush(stdout);
    }
    std::printf("\n");
    for (uint8_t* b : keep) CK(hipFree(b));
  }
  return 0;
}

// `vmm` mode: a sample buffer PIECED TOGETHER on purpose -- one range of virtual addresses, `parts` separate physical
// allocations (hipMemCreate) mapped into it one behind the other.  Is such a buffer of the fast kind by construction?
static int vmm_mode() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  uint32_t* sink;
  uint8_t* a;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&a, (size_t)FRAMES * FB + 4096));
  int dev = 0;
  CK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t need = (size_t)FRAMES * ROW + 4096;
  std::printf("granularity %zu bytes\n", gran);
  for (int parts : {1, 2, 3, 4, 6, 11, 2, 1}) {
    std::printf("%2d parts:", parts);
    for (int trial = 0; trial < 8; ++trial) {
      const size_t part = ((need + parts - 1) / parts + gran - 1) / gran * gran, total = part * parts;
      hipDeviceptr_t va = nullptr;
      CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs(parts);
      // (something odd-sized between the parts, so that they are not neighbours)
      std::vector<void*> pads;
      for (int k = 0; k < parts; ++k) {
        CK(hipMemCreate(&hs[k], part, &prop, 0));
        void* pd;
        CK(hipMalloc(&pd, (size_t)(37 + 11 * k + trial) << 20));
        pads.push_back(pd);
        CK(hipMemMap((hipDeviceptr_t)((uint8_t*)va + k * part), part, 0, hs[k], 0));
      }
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, total, &acc, 1));
      CK(hipMemset(va, 0, need));
      std::printf(" %.3f", run<128, true, false, true>(a, (uint8_t*)va, 0, false, sink, s0, s1));
      std::fflush(stdout);
      CK(hipMemUnmap(va, total));
      for (auto h : hs) CK(hipMemRelease(h));
      CK(hipMemAddressFree(va, total));
      for (void* pd : pads) CK(hipFree(pd));
    }
    std::printf("\n");
  }
  return 0;
}

// `shift` mode: ONE sample buffer with room behind it; the write-only kernel at growing offsets into it -- is a buffer's
// kind (profiles/r6/decoder_modes.txt, 4c) a matter of where in the allocation it begins?
static int shift_mode() {
  const size_t b_bytes = (size_t)FRAMES * ROW + 4096, room = (size_t)6 << 30;
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  uint32_t* sink;
  CK(hipMalloc(&sink, 64));
  for (int rep = 0; rep < 3; ++rep) {
    uint8_t *a, *b, *pad;
    CK(hipMalloc(&pad, (size_t)(rep + 1) * 1237 * 1024));
    CK(hipMalloc(&a, (size_t)FRAMES * FB + 4096));
    CK(hipMalloc(&b, b_bytes + room));
    CK(hipMemset(b, 0, b_bytes + room));
    std::printf("allocation %d (%p): writes only, 128-byte lines, by the offset the buffer begins at\n ", rep, (void*)b);
    const size_t offs[] = {0, (size_t)1 << 21, (size_t)1 << 24, (size_t)1 << 25, (size_t)1 << 26, (size_t)1 << 27, (size_t)1 << 28, (size_t)3 << 27,
                           (size_t)1 << 29, (size_t)3 << 28, (size_t)1 << 30, (size_t)3 << 29, (size_t)1 << 31, (size_t)5 << 29, (size_t)3 << 30,
                           (size_t)1 << 32, (size_t)5 << 30, ((size_t)1 << 30) + ((size_t)1 << 21) * 77};
    for (size_t o : offs) std::printf(" %zuM:%.3f", o >> 20, run<128, true, false, true>(a, b + o, 0, false, sink, s0, s1));
    std::printf("\n");
    std::fflush(stdout);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "shift") return shift_mode();
  if (argc > 1 && std::string(argv[1]) == "sizes") return sizes_mode();
  if (argc > 1 && std::string(argv[1]) == "vmm") return vmm_mode();
  if (argc > 1 && std::string(argv[1]) == "chunks") return chunks_mode(argc > 2 ? std::atoi(argv[2]) : 48, argc > 3 ? (size_t)std::atoi(argv[3]) : 512);
  const int n = argc > 1 ? std::atoi(argv[1]) : 3;
  const uint32_t delay = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 600u;   // ~4 us a step: the decoder's 0.65 ms for its 156 lines a row
  const size_t a_bytes = (size_t)FRAMES * FB + 4096, b_bytes = (size_t)FRAMES * ROW + 4096;
  std::vector<uint8_t*> As, Bs;
  for (int i = 0; i < n; ++i) {
    uint8_t *a, *b, *pad;
    CK(hipMalloc(&a, a_bytes)); CK(hipMalloc(&b, b_bytes)); CK(hipMalloc(&pad, (size_t)(i + 1) * 1237 * 1024));
    CK(hipMemset(a, 0x5A, a_bytes)); CK(hipMemset(b, 0, b_bytes));
    As.push_back(a); Bs.push_back(b);
  }
  uint32_t* sink;
  CK(hipMalloc(&sink, 64));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  struct V { const char* name; float (*fn)(const uint8_t*, uint8_t*, uint32_t, bool, uint32_t*, hipStream_t, hipStream_t); bool reader; };
  const V vs[] = {
      {"lines of 128 B, nt, + reader of A (the decode phase)", run<128, true, true, true>, true},
      {"lines of 128 B, nt, alone", run<128, true, true, true>, false},
      {"lines of 128 B, plain stores, + reader", run<128, false, true, true>, true},
      {"lines of 256 B, nt, + reader", run<256, true, true, true>, true},
      {"lines of 512 B, nt, + reader", run<512, true, true, true>, true},
      {"writes only (128 B, nt)", run<128, true, false, true>, false},
      {"reads only, + reader", run<128, true, true, false>, true},
  };
  for (const V& v : vs) {
    std::printf("%s   (delay %u)\n", v.name, delay);
    for (int i = 0; i < n; ++i) {
      std::printf("  A[%d]:", i);
      for (int j = 0; j < n; ++j) std::printf("  B[%d] %.3f", j, v.fn(As[i], Bs[j], delay, v.reader, sink, s0, s1));
      std::printf("\n");
      std::fflush(stdout);
    }
  }
  return 0;
}
