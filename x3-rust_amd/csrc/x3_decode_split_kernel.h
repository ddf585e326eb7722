// x3_decode_split_kernel.h -- the lane-per-frame decoder of x3_decode_fast_kernel split over THREE waves.
//
// Why: a frame is one serial bit stream, so decode parallelism is frames (69 120 in config 3 = 1 080
// waves, about one per SIMD), and ONE wave can issue a VALU instruction only every ~5-8 cycles however
// idle its SIMD is (tools/ubench/issue_cost.hip: 8.5 cycles dependent, 5.3 with four independent
// chains; the SIMD itself sustains one every ~3).  The time of x3_decode_fast_kernel is therefore
// (instructions per sample) x (that latency), with more than half of every SIMD unused.  Here each group
// of 64 frames gets a workgroup of three waves:
//
//   wave 0, the PARSER: owns the input ring and the bit window.  Per block it reads the 6 header bits and
//     walks the codewords: for every sample the zero run z and the field v behind it, two samples per
//     32-bit peek.  It does not compute a single sample value; it hands over i = (z << k) + r (the index
//     into the reference's inverse Rice table, or simply the field for BFP/literal blocks),
//     two 16-bit values per dword, through a double-buffered LDS block buffer, and the header bits.
//   wave 1, the VALUER: turns indices into differences (zigzag / unsigned_to_i16) and samples (running sum,
//     in packed 16-bit arithmetic), checks the table bounds, stages the samples in per-row LDS rings that are
//     indexed by the DESTINATION address, and owns status and metadata.
//   wave 2, the FLUSHER: one block behind the valuer it writes every 128-byte line of wav that has been
//     completed in the rings -- whole, aligned lines only, eight per store instruction.  (Groups that are not
//     64 equal frames side by side -- a clip's short last frame, two clips' frames, a frame index: every row with its
//     own phase and length -- make a list of the rows that have completed a line each block, flush_rows.)
//
// One s_barrier per 20-sample block: behind barrier k the parser works on block k+1, the valuer on block k,
// the flusher on what block k-1 completed.  All waves derive the per-lane block sizes from the frame header
// alone, so their loop trip counts agree whatever the payload holds; a lane whose frame fails (BFP exponent,
// table bound) is marked dead by the valuer -- the parser keeps walking its bits (lanes are independent, and
// every read is bounded by the ring), the flusher stops storing for it.
//
// Geometry: block_len = 20 (ten pairs per block), output frames 16-byte aligned (the host launches
// x3_decode_fast_kernel otherwise).  Same results as the fast kernel.
//
// What sets this kernel's time (measured, round 2; tools/dbg_stamps_split.py, tools/ubench/store_rate.hip):
//   * The stores.  Without them the kernel takes 0.70 ms.  A flush of four-block windows -- 160-byte runs at 32-byte
//     alignment, 6.4 rows per store instruction -- costs the memory side 0.64 ms for the 1.38 GB of config 3
//     (non-temporal; 0.42 plain) against 0.26 ms as aligned 128-byte lines, and made the kernel run in two or three
//     timing modes per process (0.81 / 0.87 / 0.94 ms).  With whole lines it is one mode.
//   * Non-temporal stores keep the output out of L2: with plain stores the kernel moved 1.20x its algorithmic bytes
//     (profiles/r1), now ~1.0x (profiles/r2).
//   * The flush costs the valuer ~15 % of its time if it does it itself; a wave of its own does it for free.
//   * Tried and dropped: a fourth wave that keeps the parser's ring filled (its requests, one block ahead, wait ~2 400
//     clocks for HBM beside the write stream, and two blocks ahead needs a 256-byte ring per lane that LDS has no room
//     for: +3 % at best, -5 % with the flusher beside it); touching the stream a line ahead (-3 %; round 6, the same touch from the
//     FLUSHER, which has no loads of its own to wait behind it, at the position the frame's density predicts 16 / 32 / 64
//     blocks on: 0.661 / 0.655 / 0.645 against 0.643 ms, placed buffers, profiles/r6/decoder_modes.txt); s_setprio by
//     dispatch order (the SQ issues oldest-first and the last groups run 20 % longer than the first, but handing
//     the priority to them only moves the tail to the first groups).
#pragma once
#include "x3_decode_kernel.h"

#define X3S_BL 20u             // block length served by this kernel
#define X3S_PAIRS 10u
#define X3S_RING_DW 64u        // staging ring per row: 64 dwords = two 128-byte lines of the destination
#ifndef X3S_PERIOD
#define X3S_PERIOD 2u         // the ring is topped up every X3S_PERIOD blocks (1 or 2)
#endif
#ifndef X3S_AHEAD
#define X3S_AHEAD 3u          // 16-byte chunks per lane requested one service ahead (of up to 6 per service)
#endif
#ifndef X3S_VALUER_FOLD
#define X3S_VALUER_FOLD 0
#endif
// Round 5: the THIN parser.  The parser is the group's critical wave -- its pair loop runs at the speed of a lone wave
// (152 clocks per pair of codewords against 139 alone on a SIMD: profiles/r4/stamps_decode_split_base.txt,
// ubench_pair_cost.txt) -- so what it does not have to do itself is taken off it: it walks the codeword LENGTHS only (peek,
// two v_ffbh / v_mad / the shift between them, the window update: 14 vector instructions per pair instead of 19) and hands
// over the 32-bit PEEK of every pair; the valuer, which idles 29 % of the time at the barrier and whose pairs do not depend
// on each other, repeats the five-instruction walk on the peek and extracts the two fields from it (+10 per pair there).
// Measured (profiles/r5/decoder_thin_parser.txt): 0.649 -> 0.684 ms.  The ten instructions a pair gains in the valuer make
// IT the critical wave.  Kept as a build option.
#ifndef X3S_THIN
#define X3S_THIN 0
#endif
// Round 5: the input ring TRANSPOSED -- word slot s of lane l at dword s * 64 + l instead of l * 32 + s.  With a row of 128
// bytes per lane the 64 lanes' rows start on two banks only (lane * 32 dwords: bank 0 or 32), and lanes that are at the same
// slot of their rows -- they read at similar rates -- pile onto one bank pair: 42 % of the decoder's LDS cycles were bank
// conflicts (profiles/r5/pmc_instruction_mix.txt).  Transposed, lane l only ever touches bank l: the parser's read of the next
// window word and the parks are conflict-free whatever the lanes' positions.  Same instruction count (the byte counter
// steps by 256 instead of 4; a park is four dword rows = two ds_write2st64_b32 instead of one ds_write_b128).
#ifndef X3S_RING_T
#define X3S_RING_T 0
#endif
#if X3S_RING_T
#define X3S_QSH 8            // log2 of the byte distance of two ring words of a lane
#define X3S_QMASK 0x1F00u    // (slot & 31) << 8
#else
#define X3S_QSH 2
#define X3S_QMASK 124u
#endif
#if X3S_RING_T && X3S_SWP
#error "the software-pipelined pair loop (X3S_SWP) is written for the untransposed ring"
#endif
#ifndef X3S_PAIR_ASM2
#define X3S_PAIR_ASM2 1      // the parser's pair block also holds the shift's sign and the ring address (one s_nop per pair less)
#endif
#ifndef X3S_VALUER_ASM
#define X3S_VALUER_ASM 0     // 1: the valuer's pair arithmetic as one asm block per pair (no s_nop padding between its instructions; measured: +-0)
#endif
// TIMING experiments on top of X3S_THIN (results are wrong): the valuer converts only the first X3S_THIN_K of a block's ten
// pairs from peeks to indices (as if another wave had converted the others in place); X3S_THIN_F: the flusher executes
// the conversion's instructions and LDS traffic for the other 10 - K on dummy data
#ifndef X3S_THIN_K
#define X3S_THIN_K 10
#endif
#ifndef X3S_THIN_F
#define X3S_THIN_F 0
#endif
// Round 5: the parser's pair loop SOFTWARE-PIPELINED, one asm block per block of ten pairs.  A lone wave issues an
// instruction every ~4.3 clocks when it does not depend on the one before and every ~8.5 when it does
// (tools/ubench/issue_cost.hip), and in the loop as the compiler laid it out (peek, ffbh, mad, shift, ffbh, mad, add3,
// sign, address, read, bfi, bfi: a dependent instruction behind nearly every one) a pair took 139 clocks for its 19 + 1
// instructions.  Here every instruction of the dependent chain is followed by one that does not depend on it: the
// PREVIOUS pair's index packing and its store to the block buffer, this pair's field extraction, the ring address and
// read.  The window has a third word w2 that is brought up to date half a pair late, so that the ring read has a pair and
// a half to arrive (tools/ubench/pair_cost.hip: 139 -> ~100 clocks per pair for one wave on its SIMD).  Nothing that the
// asm block reads asynchronously is left pending in a register the compiler knows: the block ends with the window settled.
#ifndef X3S_SWP
#define X3S_SWP 0
#endif
#if X3S_SWP && X3S_THIN
#error "X3S_THIN is an option of the compiler-scheduled pair loop (X3S_SWP=0)"
#endif
// pair i of a block: chain instruction, filler, chain instruction, filler ...; CNT = LDS operations that may still be in
// flight when the read of pair i-1 is needed (those issued behind it); STORE = the store of pair i-1's packed indices
#define X3S_SWP_PAIR(WNC, WNP, CNT, STORE)                           \
  "v_alignbit_b32 %[tt], %[w0], %[w1], %[s]\n\t"                     \
  "v_lshl_add_u32 %[xa], %[z1], %[lsh], %[pv1]\n\t"                  \
  "v_ffbh_u32 %[z1], %[tt]\n\t"                                      \
  "v_lshl_add_u32 %[xb], %[z2], %[lsh], %[pv2]\n\t"                  \
  "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"             \
  "v_lshl_add_u32 %[qb], %[mp], 2, %[qb]\n\t"                        \
  "v_alignbit_b32 %[t2], %[tt], 0, %[nn1]\n\t"                       \
  "v_and_or_b32 %[ad], %[qb], %[c124], %[rowb]\n\t"                  \
  "v_ffbh_u32 %[z2], %[t2]\n\t"                                      \
  "ds_read_b32 %[" WNC "], %[ad]\n\t"                                \
  "v_mad_i32_i24 %[nn2], %[z2], %[zmask], %[nwidth]\n\t"             \
  "v_perm_b32 %[xa], %[xb], %[xa], %[sel]\n\t"                       \
  "s_waitcnt lgkmcnt(" CNT ")\n\t"                                   \
  "v_bfi_b32 %[w2], %[mp], %[" WNP "], %[w2]\n\t"                    \
  "v_add3_u32 %[s2], %[s], %[nn1], %[nn2]\n\t"                       \
  STORE                                                              \
  "v_bfe_u32 %[pv1], %[tt], %[nn1], %[fw]\n\t"                       \
  "v_ashrrev_i32 %[mp], 31, %[s2]\n\t"                               \
  "v_bfe_u32 %[pv2], %[t2], %[nn2], %[fw]\n\t"                       \
  "v_bfi_b32 %[w0], %[mp], %[w1], %[w0]\n\t"                         \
  "v_and_b32 %[s], 31, %[s2]\n\t"                                    \
  "v_bfi_b32 %[w1], %[mp], %[w2], %[w1]\n\t"
#define X3S_SWP_STORE(OFF) "ds_write_b32 %[buf], %[xa] offset:" #OFF "\n\t"
// timing experiments only (results are wrong): knock out one role's work to see what the others cost each other.
// 1: the valuer's pair arithmetic and staging; 2: the flusher's loads and stores; 4: the parser's codeword walk;
// 8: the flusher's global stores only (its LDS reads stay); 16: the parser's ring service (no loads, no parks);
// 32: the service's loads only are skipped (parks of stale registers stay)
#ifndef X3S_KO
#define X3S_KO 0
#endif
// (Round 4 also swept the cache policies of the flusher's stores and the parser's requests as buffer instructions --
// plain / nt / sc1 / sc0 sc1 / sc1 nt: nothing beats nontemporal stores and plain loads, plain or sc1 stores cost 7 %
// and slow the ENCODER down by 8 % through what they leave in L2, nt loads cost 56 %: profiles/r4/decoder_cache_policies.txt.)
//  the switches above that give WRONG RESULTS are one -D away from the shipped library: they only
// compile in experiment builds
#if (X3S_KO || X3S_THIN_K != 10 || X3S_THIN_F || defined(X3S_HALF_LINES_WRONG)) && !defined(X3_EXPERIMENT)
#error "X3S_KO / X3S_THIN_K / X3S_THIN_F builds give wrong results: experiment builds only (-DX3_EXPERIMENT)"
#endif
#define X3S_XROWS 11u          // transfer rows per block buffer: 10 pair dwords + the header word
#define X3S_WAVES 3u            // parser, valuer, flusher
// Code placement (round 5).  The kernel's time depends on where its hot loops fall against the instruction fetch windows:
// the same instructions shifted by four bytes decode at 0.76 instead of 0.69 ms (MI355X_MICROARCH.md, "Code-placement
// sensitivity"; profiles/r5/decoder_code_placement.txt) -- and any edit in front of a loop shifts it.  Each role's block
// loop therefore starts at a fixed phase: aligned to 64 bytes plus X3S_PAD_x four-byte s_nops (executed once), chosen by
// measurement.  -1: no alignment (the loop falls where the code in front of it puts it).
#ifndef X3S_PAD_P
#define X3S_PAD_P 0
#endif
#ifndef X3S_PAD_V
#define X3S_PAD_V 0
#endif
#ifndef X3S_PAD_F
#define X3S_PAD_F 0
#endif
#define X3S_STR2(x) #x
#define X3S_STR(x) X3S_STR2(x)
#define X3S_PLACE(n) do { if ((n) >= 0) asm volatile(".p2align 6\n\t.rept " X3S_STR(n) "\n\ts_nop 0\n\t.endr" ::: "memory"); } while (0)

// halfword index of sample j (0..19) of a block in its transfer buffer: pair j/2 is dword (j/2 & 1) of the
// 8-byte slot of this lane in row j/4
__device__ __forceinline__ uint32_t x3s_half_index(uint32_t j, uint32_t lane) {
#if X3S_SWP
  return 2u * ((j >> 1) * 64u + lane) + (j & 1u);   // (software-pipelined parser: pair j/2 is dword `lane` of row j/2)
#else
  return 2u * (((j >> 2) * 64u + lane) * 2u + ((j >> 1) & 1u)) + (j & 1u);
#endif
}

// Pacing.  The SQ issues oldest-first, so the groups dispatched first run ahead of the ones dispatched last (group
// lifetimes 555 / 650 / 680 / 740 us by quartile of blockIdx) and the kernel ends with the stragglers, 20 % behind
// the mean.  Every 8 blocks a wave compares its block index with where the clock says it should be and sets its own
// priority: ahead -> lower, behind -> higher.  The groups then finish within 3 % of each other: 0.76 -> 0.65 ms.
// The target pace comes from the launch before: every group leaves (shader clocks per 16 blocks) in `pace` by
// atomicMax, tagged with the launch's epoch, and the next launch sets its target by that and by what the last one aimed
// at (below).  A target that does not fit (first launch of a context, other data) pins all waves at one priority: the
// unpaced kernel, nothing worse.
#ifndef X3S_PACE_OFF
#define X3S_PACE_OFF 0
#endif
#define X3S_PACE_BAND 6            // blocks ahead / behind that move a wave one priority level
// (Round 5: the pace is counted in SHADER clocks -- s_memtime -- not in 10 ns wall ticks: a box that is still ramping its
// clock, or a launch that runs at 2.0 GHz instead of 2.4, does the same work per clock, and a target in wall time read that
// as "behind" -- all waves at one priority, the controller backing off for launches on end: VERDICT r4, item 2)
#define X3S_PACE_DEFAULT 48000u    // shader clocks per 16 blocks when there is no launch to go by (1.25 us per block at 2.4 GHz)
#define X3S_PACE_EPOCH_SHIFT 20u   // pace word: epoch << 20 | shader clocks per 16 blocks
// Priorities by ROLE (round 4).  The SIMD arbitrates by priority, then age (MI355X_MICROARCH.md, "Two waves per SIMD"):
// with all three waves of a group paced over the same four levels the critical wave -- the parser, whose dependent
// chain sets the group's time per block -- was never preferred over the valuer or flusher of another group on its SIMD.
// X3S_PRIO_MODE: 0 = round 3 (every role 0..3 by pace); 1 = parser {2,3}, valuer / flusher {0,1}; 2 = parser 3,
// the others 0..2 by pace; 3 = parser {1,2,3}, others {0,1}; 4 = parser 0..3, others one level below;
// 5 = no pacing: parser 3, valuer 1, flusher 0; 6 = parser {2,3}, valuer {1,2}, flusher {0,1}; 7 = round 3 with the
// groups beyond 1 024 (four per CU) one level up
#ifndef X3S_PRIO_MODE
#define X3S_PRIO_MODE 0
#endif
#define X3S_ROLE_PARSER 0
#define X3S_ROLE_VALUER 1
#define X3S_ROLE_FLUSHER 2
template <int ROLE>
__device__ __forceinline__ void x3s_set_priority(int32_t d) {
  constexpr int B2 = 2 * X3S_PACE_BAND;
#define X3S_P4(a, b, c, e) { if (d > B2) __builtin_amdgcn_s_setprio(a); else if (d > 0) __builtin_amdgcn_s_setprio(b); else if (d > -B2) __builtin_amdgcn_s_setprio(c); else __builtin_amdgcn_s_setprio(e); }
#define X3S_P2(a, b) { if (d > 0) __builtin_amdgcn_s_setprio(a); else __builtin_amdgcn_s_setprio(b); }
#if X3S_PRIO_MODE == 0
  X3S_P4(0, 1, 2, 3)
#elif X3S_PRIO_MODE == 1
  if (ROLE == X3S_ROLE_PARSER) X3S_P2(2, 3) else X3S_P2(0, 1)
#elif X3S_PRIO_MODE == 2
  if (ROLE == X3S_ROLE_PARSER) __builtin_amdgcn_s_setprio(3); else X3S_P4(0, 1, 2, 2)
#elif X3S_PRIO_MODE == 3
  if (ROLE == X3S_ROLE_PARSER) X3S_P4(1, 2, 2, 3) else X3S_P2(0, 1)
#elif X3S_PRIO_MODE == 4
  if (ROLE == X3S_ROLE_PARSER) X3S_P4(0, 1, 2, 3) else X3S_P4(0, 0, 1, 2)
#elif X3S_PRIO_MODE == 5
  __builtin_amdgcn_s_setprio(ROLE == X3S_ROLE_PARSER ? 3 : (ROLE == X3S_ROLE_VALUER ? 1 : 0));
#elif X3S_PRIO_MODE == 6
  if (ROLE == X3S_ROLE_PARSER) X3S_P2(2, 3) else if (ROLE == X3S_ROLE_VALUER) X3S_P2(1, 2) else X3S_P2(0, 1)
#elif X3S_PRIO_MODE == 7
  // the groups beyond four per CU (the fifth group of 56 CUs in config 3) one level up
  if (blockIdx.x >= 1024u) X3S_P4(1, 2, 3, 3) else X3S_P4(0, 1, 2, 3)
#endif
#undef X3S_P4
#undef X3S_P2
}
#define X3S_PACE_STEP(b, ROLE)                                                                 \
  if (paced && ((b) & 7u) == 0u) {                                                             \
    const unsigned long long el = (unsigned long long)(clock64() - pace_t0); /* shader clocks */ \
    const int32_t d = (int32_t)(b) - (int32_t)((el * pace_inv) >> 24);                           \
    x3s_set_priority<ROLE>(d);                                                                 \
  }

// ---- Round 5: STRETCHES.  A frame is one serial bit stream, so a stream of few frames cannot be decoded faster than
// one frame's walk (0.41 ms for 500 blocks), and 1 080 groups on 256 CUs leave 56 CUs with a fifth group that ends the
// kernel.  A block, though, depends on nothing but the bit position it starts at and the sample in front of it
// (decoder.rs:36-58).  The encoder knows both for every block (its prefix scan of the blocks' bit lengths, its input), and
// so does a serial decode: the SEGMENT INDEX holds, for every frame and every `sb` blocks, {bit offset of block sb*j's
// header from the start of the payload, the sample in front of it | 0x10000}.  With it, lane = (frame, stretch j): blocks
// [sb*j, sb*(j+1)) of the frame, decoded from the index entry into the row wav + wo + 20*sb*j.  The index is a HINT, not
// part of the format and not trusted: a stretch that ends inside its frame compares where it ended -- bit position and
// last sample -- with the entry of the next one; by induction from the frame's first block, the stretches of a frame
// whose comparisons all hold decode what the serial walk decodes.  Any mismatch, any entry that is not plausible, marks
// the frame X3D_REPLAY and the reference's reader decodes it (x3_decode_replay.h), so a wrong index costs time only.
// Layout: word 0 is a header {X3S_SEG_MAGIC, blocks per entry}, written by whoever fills the index (an encoder that cannot --
// the kernels for other layouts -- leaves it zero, and a decode by such an index falls back to whole frames per lane);
// entry (f, k), k = 1 .. pitch, at word 1 + f * pitch + k - 1 is for block isb * k.  A decode may take every mul-th entry
// (stretches of sb = isb * mul blocks): the host picks as many stretches as fill the chip and no more.
// in: the index to decode by (grid = groups * nseg);  out: the index to RECORD while decoding serially (either may be null)
struct X3SegArgs {
  const uint2* in;
  uint2* out;
  uint32_t sb;     // blocks per stretch (a multiple of 4: stretches start on 16-sample = 32-byte boundaries of the frame)
  uint32_t nseg;   // stretches per frame = ceil(blocks per frame / sb)
  uint32_t pitch;  // entries per frame in the index
  uint32_t mul;    // sb / blocks per index entry
};
#define X3S_SEG_VALID 0x10000u
#define X3S_SEG_MAGIC 0x58335347u

// LDS barrier of the group's waves: LDS operations retired, nothing else waited for
#define X3S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// (amdgpu_waves_per_eu(1, 4): LDS holds five groups per CU = fifteen waves = four per SIMD at most, so the register allocator
// may use 128 registers per lane.  Left to aim at eight waves per SIMD it squeezes the kernel into 70 and, depending on the
// code around them, COPIES part of a ring request's destination registers right behind the request -- a wait for a load
// that has just been issued, in every service: 0.69 -> 0.75 ms (profiles/r5/decoder_service_schedule.txt).)
#ifndef X3S_WAVES_PER_EU
#define X3S_WAVES_PER_EU 4
#endif
__global__ void __launch_bounds__(64 * X3S_WAVES) __attribute__((amdgpu_waves_per_eu(1, X3S_WAVES_PER_EU)))
x3_decode_split_kernel(const uint8_t* __restrict__ x3, uint64_t x3_len, const uint64_t* __restrict__ frame_off,
                       uint64_t n_frames_arg, X3Geom g, const uint64_t* __restrict__ wav_off, X3DevParams p,
                       int16_t* __restrict__ wav, uint64_t wav_cap, int32_t* __restrict__ status,
                       X3FrameMeta* __restrict__ meta, uint32_t* __restrict__ pace, uint32_t pace_epoch, X3SegArgs sg,
                       const unsigned long long* __restrict__ d_nf) {
  // d_nf: the frame count is still on its way to the host (x3_decode_stream_dev in one trip: the frame walk's kernels are
  // in front of this one in the stream) -- the launch covers an upper bound, the groups behind the real count leave here
  const uint64_t n_frames = d_nf ? (*d_nf < n_frames_arg ? (uint64_t)*d_nf : n_frames_arg) : n_frames_arg;   // (never beyond the bound the arrays were sized for)
  if (d_nf && (uint64_t)blockIdx.x * 64u >= n_frames) return;
  // input ring, 32 dwords per lane in rows of exactly 128 bytes at 128-byte aligned addresses, stream word j in
  // slot ~j & 31 (descending): the address of a word is then ONE v_and_or_b32 on a byte counter that a shift of
  // the window decrements with one v_lshl_add_u32.  (Lanes are at different places in their rows, so the aligned
  // rows do not line the reads up on one bank.)
#if X3S_RING_T
  __shared__ __attribute__((aligned(8192))) uint32_t ring[64 * X3_DEC_RING_DW];   // [slot][lane]; 8 KB aligned: slot bits 8..12 of the address are the ring's own
#else
  __shared__ __attribute__((aligned(128))) uint32_t ring[64 * X3_DEC_RING_DW];
#endif
  __shared__ __attribute__((aligned(256))) uint32_t outs[64 * X3S_RING_DW];
  __shared__ __attribute__((aligned(16))) uint32_t xfer[2 * X3S_XROWS * 64];
  __shared__ uint32_t s_over[64];  // parser -> valuer: the frame was read beyond its payload (x3_decode_replay.h)
  __shared__ uint32_t s_dead[64];  // valuer -> flusher: the frame failed, no more stores for it
  // flusher, groups that are not 64 equal frames side by side: the rows that have a line to write this block, in the
  // order of their rank among them: {line address lo, hi, row | ring byte of the line << 8 | first piece << 16 | end piece << 20}
  __shared__ __attribute__((aligned(16))) uint32_t s_prm[64 * 4];

  const uint32_t lane = threadIdx.x & 63u;
  const bool parser = threadIdx.x < 64u;
  const bool flusher = (threadIdx.x >> 6) == 2u;
  const bool seg_grid = sg.in != nullptr;   // (uniform) the grid has nseg groups per 64 frames
  const uint32_t seg_j = seg_grid ? blockIdx.x % sg.nseg : 0u;
  // (an index whose header does not say what this launch expects: whole frames per lane, in the groups of stretch 0)
  bool segd = false;
  if (seg_grid) {
    const uint2 h = sg.in[0];
    segd = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.x) == X3S_SEG_MAGIC &&
           (uint32_t)__builtin_amdgcn_readfirstlane((int)h.y) * sg.mul == sg.sb;
    if (!segd && seg_j) return;
  }
  const uint64_t f = (uint64_t)(seg_grid ? blockIdx.x / sg.nseg : blockIdx.x) * 64 + lane;
  if (sg.out && blockIdx.x == 0 && threadIdx.x == 0) sg.out[0] = make_uint2(X3S_SEG_MAGIC, sg.sb);
  const bool paced = !X3S_PACE_OFF && !segd;   // (stretches: many short groups, dispatched as CUs fall free -- nothing to pace)
  const unsigned long long wall_t0 = wall_clock64();
  const unsigned long long clk_t0 = clock64();   // (shader clock: the pace's clock, and the launch log's clock measurement, below)
  const unsigned long long pace_t0 = clk_t0;
  uint32_t pace_inv;        // blocks per shader clock, 8.24 fixed point
  uint32_t pace_target;     // shader clocks per 16 blocks that this launch aims at
  {
    // pace[q], pace[2 + q], q = parity of the launch before: what the slowest group of that launch achieved (P) and what it
    // aimed at (T).  This launch writes the words of ITS parity, so that a group that is dispatched late -- a grid larger
    // than the chip holds -- still reads what the first groups read, not what the first finishers of this launch have
    // left.  At its best the kernel achieves ~4.5 % more than it aims at; a target that is too fast by as
    // little as 3 % throws the gain away (all waves end up "behind", at one priority: P jumps to 1.1 T), one that is too
    // slow is simply met (P = T).  So: met -> aim 1.5 % faster; missed by more than 6 % -> back to 4.5 % under what was
    // achieved; in between -> hold.  (Aiming a fixed fraction under P saw-toothed over the cliff every third or fourth
    // launch: tools/pace_trace.py.)
    const uint32_t mask = (1u << X3S_PACE_EPOCH_SHIFT) - 1u, prev = (pace_epoch - 1u) & 0xFFFu, q = prev & 1u;
    const uint32_t wp = __builtin_amdgcn_readfirstlane(pace[q]), wt = __builtin_amdgcn_readfirstlane(pace[2u + q]);
    // ... and only a launch of the SAME SHAPE counts (round 4): pace[8 + q] holds the number of groups of the launch
    // before.  A context that decodes a stream in chunks of growing size, or a window of 4 096 frames between two whole
    // streams, left pace words that fitted another launch; the next launches of config 3 then took 0.75 ms instead of
    // 0.70 until the controller had found its way back (bench.py's with_frame_walk behind the host-buffer calls).  A
    // launch of another shape starts from the data, like a context's first.
    const uint32_t ws = __builtin_amdgcn_readfirstlane(pace[8u + q]);
    const bool same_shape = (ws >> X3S_PACE_EPOCH_SHIFT) == prev && (ws & mask) == (gridDim.x & mask);
    const uint32_t P = same_shape && (wp >> X3S_PACE_EPOCH_SHIFT) == prev ? (wp & mask) : 0u;
    const uint32_t T = same_shape && (wt >> X3S_PACE_EPOCH_SHIFT) == prev ? (wt & mask) : 0u;
    uint32_t t16;
    if (P == 0u) {
      // No launch to go by (a context's first one): from the DATA.  A group's time per block is linear in the bytes it
      // has to pull in per block -- settled paces of 1.16 / 1.28 / 1.82 / 1.92 us per block at 0.14 / 0.53 / 1.66 / 2.04
      // stream bytes per sample (zeros, hydrophone noise, sine, white noise) are 1.10 us + 0.40 us
      // per byte and sample, the target that settles there is 4.5 % under what is achieved -- and the stream's density is
      // in the frame offsets.
      t16 = X3S_PACE_DEFAULT;
      if (n_frames >= 2u && p.spf) {
        const unsigned long long span = frame_off[n_frames - 1u] - frame_off[0];
        const unsigned long long samples = (unsigned long long)(n_frames - 1u) * p.spf;
        const unsigned long long t = 40416ull + (14688ull * span) / samples;   // shader clocks per 16 blocks (the fit above at 2.4 GHz)
        t16 = t > mask ? mask : (uint32_t)t;
      }
    }
    else if (T == 0u) t16 = P - P / 22u;
    else {
      const uint32_t r = (P << 8) / T;  // 256 = met exactly
      // (met with room to spare: 4 % faster.  Missed: back off, but by no more than 3 % a launch -- a launch that missed by
      // far did so for another reason, e.g. the first one of a process on idle clocks, and the target was right)
      const uint32_t back = P - P / 22u, cap3 = T + T / 32u;
      t16 = r < 259u ? T - T / 24u : (r < 264u ? T - T / 64u : (r <= 272u ? T : (back < cap3 ? back : cap3)));
    }
    if (t16 < 1536u) t16 = 1536u;
    if (t16 > mask) t16 = mask;
    pace_target = t16;
    pace_inv = (16u << 24) / t16;
  }
#ifdef X3_DBG_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_start = wall_clock64();
#endif

  // ---- per-lane frame setup, done by both waves (same checks as the fast kernel)
  bool active = f < n_frames;
  int32_t st = X3D_OK;
  uint32_t samples = 0, plen = 2;
  uint64_t p0 = 0, wo = 0;
  if (active) {
    uint32_t pcrc_unused;
    st = x3_frame_header_check(reinterpret_cast<const uint32_t*>(x3 - (reinterpret_cast<uintptr_t>(x3) & 3u)),
                               (x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u) + 3) >> 2,
                               x3_len + (reinterpret_cast<uintptr_t>(x3) & 3u),
                               frame_off[f] + (reinterpret_cast<uintptr_t>(x3) & 3u), plen, samples, pcrc_unused);
    if ((threadIdx.x >> 6) == 1u) {
      meta[f].payload_len = plen;
      meta[f].samples = samples;
    }
    p0 = frame_off[f] + 20;
    if (st != X3D_OK) {
      active = false;
    } else if (samples == 0 || plen < 2) {
      st = X3D_BAD_ARG;
      active = false;
    } else {
      if (wav_off) {  // (the caller vouches for multiples of four samples: output rows on 8-byte boundaries at least)
        wo = wav_off[f];
      } else {
        const uint64_t clip = f / g.fpc;
        const uint64_t idx = f - clip * g.fpc;
        wo = clip * g.clip_stride + idx * (uint64_t)p.spf;
      }
      if (wo + samples > wav_cap) {
        st = X3D_BAD_ARG;
        active = false;
      }
    }
  }
  // ---- the stretch of the frame that this lane decodes (all of it without an index)
  uint32_t hb = 16u;                    // bit offset of its first block header from the start of the payload
  uint32_t pred = 0;                    // (seg_j > 0) the sample in front of it
  bool mid = false;                     // it ends inside the frame: its last sample is the next stretch's first
  uint32_t exp_bits = 0, exp_pred = 0;  // (mid) the next stretch's index entry
  if (segd && active) {
    const uint32_t nbf = (samples - 1u + X3S_BL - 1u) / X3S_BL;   // blocks of the frame
    const uint32_t b0 = sg.sb * seg_j;
    const uint2* const e = sg.in + 1 + f * (uint64_t)sg.pitch;
    if (seg_j && b0 >= nbf) {
      active = false;                   // the frame has no such stretch (st stays OK)
    } else {
      if (seg_j) {
        const uint2 h = e[seg_j * sg.mul - 1u];
        if (!(h.y & X3S_SEG_VALID) || h.x < 16u || h.x > 8u * plen) {
          st = X3D_REPLAY;              // not an entry to start from: the reference's reader takes the frame
          active = false;
        } else {
          hb = h.x;
          pred = h.y & 0xFFFFu;
        }
      }
      // (the frame's LAST stretch of the launch is never `mid`: a frame with more blocks than nseg * sb -- its header asks for
      // more than the parameters' frame holds -- is decoded to its end by that stretch, and no entry behind the frame's row
      // of the index is ever read: ADVICE r5)
      if (active && b0 + sg.sb < nbf && seg_j + 1u < sg.nseg) {
        const uint2 hn = e[(seg_j + 1u) * sg.mul - 1u];
        mid = true;
        exp_bits = hn.x;
        exp_pred = hn.y;
      }
      if (active) {
        wo += (uint64_t)X3S_BL * b0;
        samples = mid ? X3S_BL * sg.sb + 1u : samples - X3S_BL * b0;
      }
    }
  }
  if (!active) { p0 = 0; plen = 2; wo = 0; samples = 0; hb = 16u; mid = false; }
  // blocks of this lane's frame and of the longest frame of the group: the loop both waves run
  const uint32_t nblk = samples ? (samples - 1u + X3S_BL - 1u) / X3S_BL : 0u;
  const uint32_t nblk_max = __builtin_amdgcn_readfirstlane(x3_wave_max_u32(nblk));
  uint32_t remaining = samples ? samples - 1u : 0u;

  // ---- the input ring of this lane's frame (the parser fills and reads it)
#if X3S_RING_T
  const uint32_t row_base = x3_lds_addr(ring) + 4u * lane;   // LDS byte address of this lane's column (bits 8..12 zero)
#define X3S_RING_WORD(j) ring[((~(j)) & 31u) * 64u + lane]
#else
  uint32_t* const row = ring + lane * X3_DEC_RING_DW;
  const uint32_t row_base = (uint32_t)(uintptr_t)row;  // LDS byte address of the row (low 7 bits zero)
#define X3S_RING_WORD(j) row[~(j) & 31u]
#endif
  // words are parked BIG-ENDIAN
  const uint32_t adj = (uint32_t)(reinterpret_cast<uintptr_t>(x3) & 15u);
  // offsets are relative to this lane's first 16-byte chunk (a frame is < 64 KB): a 64-bit pointer per lane,
  // 32-bit arithmetic on everything else, streams of any length
  const uint64_t abs_bits = (uint64_t)adj + p0 + (hb >> 3);   // (a byte position; hb = 16 without an index: behind the first sample)
  const uint32_t ebits = hb & 7u;                             // ... and the bits of that byte in front of the header
  // The first chunk is the one that holds that byte -- or the payload's LAST byte, where the bit stream starts at the very
  // end of the payload (a frame of one sample; an index entry that points there) on a 16-byte boundary: the chunk behind
  // the payload may be the first one behind the stream (found by the guard pages of x3_fence.h: until round 5 such a lane
  // read its eight chunks from there, 128 bytes that nobody used and that nobody may have mapped).  hb >> 3 <= plen, plen >= 2.
  const uint64_t abs_last = (uint64_t)adj + p0 + plen - 1u;
  const uint64_t abs_base = (abs_bits < abs_last ? abs_bits : abs_last) & ~15ull;
  const uint8_t* __restrict__ const x3b = (x3 - adj) + abs_base;
  const uint32_t v_bits = (uint32_t)(abs_bits - abs_base);   // first block header (0..16)
  const uint32_t v_end = v_bits - (hb >> 3) + plen;          // end of the payload
  const int32_t v_rel = 8 * (int32_t)(hb >> 3) - 8 * (int32_t)v_bits;   // payload bit = ring bit + v_rel
  const uint32_t v_last = (v_end - 1u) & ~15u;               // last 16-byte chunk that holds payload
  uint32_t v_next = 0;
  uint32_t wr_abs = 0;
  constexpr uint32_t SVC_MAX = 3u * X3S_PERIOD;  // chunks a service can park per lane
  constexpr uint32_t SVC_AHEAD = X3S_AHEAD;      // of which requested one service ahead
  auto request = [&](uint32_t v) -> uint4 {
    const uint32_t a = v < v_last ? v : v_last;
    return *reinterpret_cast<const uint4*>(x3b + a);
  };
  // Bytes behind the end of the payload are parked as they come (the next frame's header): a frame that READS
  // beyond its payload is flagged (s_over) and decoded again by the reference's reader, which knows about the
  // zeros there (x3_decode_replay.h); no conforming frame does.
  auto park = [&](uint4 c) {
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
    // words wr_abs .. wr_abs+3 -> slots ~wr_abs & 31 downwards = the aligned 16-byte block at slot ~(wr_abs+3) & 31:
    // byte offset (-4 * wr_abs - 16) & 112 of the 128-byte aligned row
#if X3S_RING_T
    // ... transposed: slots g .. g + 3, g = ~(wr_abs + 3) & 31 (a multiple of four), 256 bytes apart in this lane's column
    uint32_t* const q = ring + ((~(wr_abs + 3u)) & 31u) * 64u + lane;
    q[0] = x3_bswap32(w[3]);
    q[64] = x3_bswap32(w[2]);
    q[128] = x3_bswap32(w[1]);
    q[192] = x3_bswap32(w[0]);
#else
    x3_lds_write_b128(x3_and_or(0u - 4u * wr_abs - 16u, 112u, row_base), x3_bswap32(w[3]), x3_bswap32(w[2]),
                      x3_bswap32(w[1]), x3_bswap32(w[0]));
#endif
    wr_abs += 4;
  };
  // Round 5 (-DX3S_DENSE_AHEAD=0: off): a group that holds a DENSE frame -- a payload beyond what the encoder's image takes,
  // literal and wide BFP blocks: a ship passing the hydrophone -- needs all SVC_MAX chunks in that lane at every service,
  // and asked for the ones beyond SVC_AHEAD only then, with the whole group waiting for memory (1 % loud frames: every
  // second group holds one, 0.82 ms against 0.68).  Such groups request all of them a service ahead.
  // (ONE lambda, ONE array: the register allocator is touchy about the requests' destination registers -- with the two
  // cases in lambdas of their own behind a dispatching one it copied part of a destination right behind the request and
  // waited for it there, 0.69 -> 0.75 ms with the feature compiled out; tools/check_decoder_isa.py looks for that.)
#ifndef X3S_DENSE_AHEAD
#define X3S_DENSE_AHEAD 1
#endif
  const bool dense_grp = X3S_DENSE_AHEAD && __any(active && plen > 9728u);   // (9 728: X3_DENSE_PAYLOAD_BYTES, the encoder's image)
  uint4 ld[SVC_MAX];
  uint32_t v_req = 0;
  auto fill_ring = [&]() {  // the first 128 bytes, and the requests of the first service
    uint4 c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = request(v_next + 16u * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) park(c[k]);
    v_next += 128;
    // the ring is topped up with up to SVC_MAX chunks per service (>= what the blocks in between can consume: at
    // most one word per pair).  SVC_AHEAD of them are requested one service ahead -- most lanes need one or two
    // (0.53 bytes per sample), and a scattered 16-byte-per-lane load costs ~64 cycles of issue -- the others only
    // when some lane does need them.
#pragma unroll
    for (int k = 0; k < (int)SVC_AHEAD; ++k) ld[k] = request(v_next + 16u * k);
    if (dense_grp) {
#pragma unroll
      for (int k = (int)SVC_AHEAD; k < (int)SVC_MAX; ++k) ld[k] = request(v_next + 16u * k);
    }
    v_req = v_next;
  };
  // widx = ring index of the parser's w0 (may be -1): everything in front of it is free
  auto service = [&](uint32_t widx) {
    const uint32_t used = wr_abs - widx;  // dwords from w0 on that the ring still needs
    const uint32_t fit = used >= X3_DEC_RING_DW ? 0u : (X3_DEC_RING_DW - used) >> 2;
#pragma unroll
    for (uint32_t k = 0; k < SVC_AHEAD; ++k) {
      if (fit > k) park(ld[k]);
    }
    if (__any(fit > SVC_AHEAD)) {  // a lane went through more than that since the last service (BFP / literal blocks)
      if (!dense_grp) {
#pragma unroll
        for (uint32_t k = SVC_AHEAD; k < SVC_MAX; ++k) ld[k] = request(v_req + 16u * k);
      }
#pragma unroll
      for (uint32_t k = SVC_AHEAD; k < SVC_MAX; ++k) {
        if (fit > k) park(ld[k]);
      }
    }
    v_next += 16u * (fit > SVC_MAX ? SVC_MAX : fit);
    v_req = v_next;
    if (!(X3S_KO & 32)) {
#pragma unroll
      for (int k = 0; k < (int)SVC_AHEAD; ++k) ld[k] = request(v_req + 16u * k);
      if (dense_grp) {
#pragma unroll
        for (int k = (int)SVC_AHEAD; k < (int)SVC_MAX; ++k) ld[k] = request(v_req + 16u * k);
      }
    }
  };

  // the usual group: 64 frames of the same size, one behind the other in wav.  Rows r, r + 4, r + 8 ... then have
  // the same phase against the 128-byte lines (S0 a multiple of 16 samples), and a CLASS of 16 rows (r & 3 == c)
  // completes its next line in the same block: two store instructions of eight whole lines each.  Frames of 8 (mod 16)
  // samples -- an even number of blocks that is not a multiple of four -- have eight phases: classes of eight rows
  // (r & 7 == c), one store instruction each (`cls8`, round 4; such frames went row by row through the valuer before).
  const uint32_t S0 = __builtin_amdgcn_readfirstlane(samples);
  // the ROW a lane writes: its samples without the last one of a stretch that ends inside the frame (that one is the
  // next stretch's); rows of a regular group are R0 samples long and ST samples apart (frames: R0 = ST = S0)
  const uint32_t rowlen = samples - (mid ? 1u : 0u);
  const uint32_t R0 = __builtin_amdgcn_readfirstlane(rowlen);
  // (the builtin returns int: without the casts a low word with bit 31 set sign-extends over the high one, and every
  // group whose sample offset has that bit stops being "regular" -- half of all groups beyond 2^31 samples)
  const uint64_t wo0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wo >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wo);
  // (rows on 8-byte boundaries only -- an output or a clip stride of 4 (mod 8) samples, frames of an odd number of
  // blocks --: the flusher's list with 8-byte pieces, flush_rows)
  const bool p8 = __any(active && (reinterpret_cast<uintptr_t>(wav + wo) & 15u) != 0u);
  const uint64_t wo1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wo >> 32), 1) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wo, 1);
  const uint32_t ST = segd ? (uint32_t)(wo1 - wo0) : S0;
  bool regular = __all(active && samples == S0 && rowlen == R0 && wo == wo0 + (uint64_t)lane * ST) && (R0 & 7u) == 0 &&
                 (ST & 7u) == 0 && ST >= R0 && ST < 0x10000u && !p8;
  const bool cls8 = (ST & 15u) != 0u;
  const uint64_t B0 = (uint64_t)(uintptr_t)(wav + wo0);  // destination byte address of the group (16-byte aligned)
  uint8_t* const line0 = reinterpret_cast<uint8_t*>(B0 & ~127ull);  // its first line
  // this lane's part in the flush of class c, store i: piece `pc` of the current line of row rr[c][i]
  const uint32_t pc = lane & 7u;
  uint32_t f_src[4][2], f_dst[4][2];  // LDS byte address / byte offset from line0, for the row's FIRST line
  uint32_t ph[8], nfl[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // per class: dwords of the first line in front of the row; lines flushed
#pragma unroll
  for (uint32_t c = 0; c < 8; ++c) ph[c] = (uint32_t)(((B0 + (uint64_t)c * 2u * ST) & 127u) >> 2);
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) {
#pragma unroll
    for (uint32_t i = 0; i < 2; ++i) {
      // (eight classes: entry [c][i] is class c + 4 i)
      const uint32_t rr = cls8 ? 8u * (lane >> 3) + (c + 4u * i) : 4u * ((lane >> 3) + 8u * i) + c;
      const uint32_t off = (uint32_t)(B0 & 127u) + rr * 2u * ST;     // bytes from line0 to the row (< 2^32: 64 frames)
      const uint32_t rrot = 16u * ((rr >> 2) & 15u);
      f_dst[c][i] = (off & ~127u) + 16u * pc;
      f_src[c][i] = x3_lds_addr(outs) + rr * (4u * X3S_RING_DW) +
                    ((((uint32_t)B0 & ~127u) + (off & ~127u) + 16u * pc + rrot) & 255u);
    }
  }
  // one line of every row of class c: line n of the row (n = 0: its first, possibly partial line), pieces
  // [p_lo, p_hi) of it
  auto flush_class = [&](uint32_t c, uint32_t c_src0, uint32_t c_src1, uint32_t c_dst0, uint32_t c_dst1, uint32_t n,
                         uint32_t p_lo, uint32_t p_hi) {
    X3_WAVE_LDS_ORDER();
    const uint32_t flip = (n & 1u) << 7;
    const x3_u32x4 v0 = x3_lds_read_b128(c_src0 ^ flip);
    const x3_u32x4 v1 = x3_lds_read_b128(c_src1 ^ flip);
    // a frame that failed is not written any further (the valuer says which)
    const bool ok0 = s_dead[4u * (lane >> 3) + c] == 0u, ok1 = s_dead[4u * ((lane >> 3) + 8u) + c] == 0u;
    if (!(X3S_KO & 8) && pc >= p_lo && pc < p_hi) {
      if (ok0) x3_store_stream16(line0 + (c_dst0 + 128u * n), v0);
      if (ok1) x3_store_stream16(line0 + (c_dst1 + 128u * n), v1);
    }
    if (X3S_KO & 8) asm volatile("" :: "v"(v0.x), "v"(v1.x), "v"(v0.w), "v"(v1.w));
    X3_WAVE_LDS_ORDER();
  };
  // the same for one of EIGHT classes: one line of each of its eight rows
  auto flush_row = [&](uint32_t c, uint32_t c_src, uint32_t c_dst, uint32_t n, uint32_t p_lo, uint32_t p_hi) {
    X3_WAVE_LDS_ORDER();
    const x3_u32x4 v = x3_lds_read_b128(c_src ^ ((n & 1u) << 7));
    const bool ok = s_dead[8u * (lane >> 3) + c] == 0u;
    if (pc >= p_lo && pc < p_hi && ok) x3_store_stream16(line0 + (c_dst + 128u * n), v);
    X3_WAVE_LDS_ORDER();
  };
  // every class whose rows have completed a line: k_done = dwords of each row that are staged
#ifndef X3S_HALF_LINES
#define X3S_FLUSH_CLASS(c)                                                                                       \
  {                                                                                                              \
    const uint32_t ld = (ph[c] + k_done) >> 5;                                                                   \
    if (ld > nfl[c]) {                                                                                           \
      flush_class(c, f_src[c][0], f_src[c][1], f_dst[c][0], f_dst[c][1], nfl[c], nfl[c] ? 0u : ph[c] >> 2, 8u);  \
      nfl[c] = ld;                                                                                               \
    }                                                                                                            \
  }
#define X3S_FLUSH_TAILS()                                                                                        \
  _Pragma("unroll") for (uint32_t c = 0; c < 4; ++c) {                                                           \
    const uint32_t tail = ((ph[c] + (R0 >> 1)) & 31u) >> 2; /* pieces of the last, partial line */               \
    if (tail) flush_class(c, f_src[c][0], f_src[c][1], f_dst[c][0], f_dst[c][1], nfl[c], nfl[c] ? 0u : ph[c] >> 2, tail); \
  }
#define X3S_FLUSH_CLASS8(c)                                                                                      \
  {                                                                                                              \
    const uint32_t ld = (ph[c] + k_done) >> 5;                                                                   \
    if (ld > nfl[c]) {                                                                                           \
      flush_row(c, f_src[(c) & 3][(c) >> 2], f_dst[(c) & 3][(c) >> 2], nfl[c], nfl[c] ? 0u : ph[c] >> 2, 8u);    \
      nfl[c] = ld;                                                                                               \
    }                                                                                                            \
  }
#define X3S_FLUSH_TAILS8()                                                                                       \
  _Pragma("unroll") for (uint32_t c = 0; c < 8; ++c) {                                                           \
    const uint32_t tail = ((ph[c] + (R0 >> 1)) & 31u) >> 2;                                                      \
    if (tail) flush_row(c, f_src[c & 3][c >> 2], f_dst[c & 3][c >> 2], nfl[c], nfl[c] ? 0u : ph[c] >> 2, tail);   \
  }
#else
  // EXPERIMENT (-DX3S_HALF_LINES; never shipped): the same flush in aligned 64-byte HALF lines, each as soon as it is
  // complete -- what a group with 128-byte staging rows would do.  nfl counts half lines here.  Bit-exact
  // (tools/scratch/half_check.py) and 0.94-1.00 ms against 0.67-0.70 on the same box: whole lines are not negotiable
  // (DESIGN.md section 8).
#define X3S_FLUSH_CLASS(c)                                                                                       \
  {                                                                                                              \
    const uint32_t hd = (ph[c] + k_done) >> 4;                                                                   \
    if (hd > nfl[c]) {                                                                                           \
      const uint32_t lo_ = 4u * (nfl[c] & 1u), first_ = ph[c] >> 2;                                              \
      flush_class(c, f_src[c][0], f_src[c][1], f_dst[c][0], f_dst[c][1], nfl[c] >> 1,                            \
                  (nfl[c] >> 1) == 0u && first_ > lo_ ? first_ : lo_, lo_ + 4u);                                 \
      nfl[c] += 1u;                                                                                              \
    }                                                                                                            \
  }
#define X3S_FLUSH_TAILS()                                                                                        \
  _Pragma("unroll") for (uint32_t c = 0; c < 4; ++c) {                                                           \
    const uint32_t tail = ((ph[c] + (R0 >> 1)) & 31u) >> 2; /* pieces of the last, partial line */               \
    const uint32_t tline = (ph[c] + (R0 >> 1)) >> 5;        /* that line's number */                             \
    if (tail) flush_class(c, f_src[c][0], f_src[c][1], f_dst[c][0], f_dst[c][1], tline, tline ? 0u : ph[c] >> 2, tail); \
  }
#endif


  if (flusher) {
    // ================================================================= last wave: flusher
    // Writes the lines that the valuer has completed in the staging rings of a regular group, one block behind it
    // (behind barrier b the valuer's block b - 1 is staged; a line is overwritten five blocks after it was completed).
    X3S_BARRIER();
    uint32_t rem = S0 ? S0 - 1u : 0u, have = 0;
    // Any other group (a clip's short last frame, the frames of two clips, offsets from a frame index): every row has its
    // own phase against the lines and its own length.  This lane OWNS row `lane`: it knows what the valuer has staged of
    // it (g_have), and when that completes a line the row joins the block's list (s_prm, by rank among the rows that
    // do); then eight lanes take one listed row each round, as in the classes' flush.  (Until round 4 the valuer moved
    // such rows itself, sixteen bytes per lane and store: a batch of clips with ragged ends decoded at half the speed.)
    uint32_t g_rem = samples ? samples - 1u : 0u, g_have = 0, g_nfl = 0;
    const uint32_t g_sh = p8 ? 3u : 4u;                                // pieces of 8 or 16 bytes (the group's rows allow)
    const uint32_t g_pl = 128u >> g_sh;                                // pieces per line = lanes per listed row
    const uint64_t g_B = (uint64_t)(uintptr_t)(wav + wo);              // the row's first byte
    const uint32_t g_p0 = (uint32_t)(g_B & 127u) >> g_sh;              // pieces of its first line in front of it
    const uint32_t g_end = g_p0 + (rowlen >> (g_sh - 1u));             // end of its whole pieces, counted from that line
    const uint32_t g_rot = 16u * ((lane >> 2) & 15u);
    auto flush_rows = [&]() {
      X3_WAVE_LDS_ORDER();
      const uint32_t n = g_nfl;
      // a line that the staged dwords reach the end of -- or the row's last, partial line as soon as everything is
      // staged: a lane whose frame has ended goes on through the group's blocks, and what its valuer lane stages from
      // then on is not the row's (six blocks on it is over the row's last samples)
      const bool whole = ((((uint32_t)(g_B & 127u) >> 2) + (g_have >> 1)) >> 5) > n && g_pl * n + g_pl <= g_end;
      const bool rest = !whole && g_rem == 0u && g_pl * n < g_end;
      uint32_t p_hi = whole ? g_pl : g_end - g_pl * n;
      if (p_hi > g_pl) p_hi = g_pl;
      const uint32_t p_lo = n ? 0u : g_p0;
      const bool ready = active && s_dead[lane] == 0u && (whole || rest) && p_hi > p_lo;
      const unsigned long long mask = __ballot(ready);
      if (mask == 0ull) return;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      const uint64_t la = (g_B & ~127ull) + 128ull * n;
      if (ready) {
        x3_lds_write_b128(x3_lds_addr(s_prm) + 16u * rank, (uint32_t)la, (uint32_t)(la >> 32),
                          lane | ((((uint32_t)la + g_rot) & 255u) << 8) | (p_lo << 16) | (p_hi << 21), 0u);
        g_nfl = n + 1u;
      }
      X3_WAVE_LDS_ORDER();
      const uint32_t nready = (uint32_t)__builtin_popcountll(mask);
      const uint32_t per_round = 64u >> (7u - g_sh);   // listed rows a round takes: 8 (16-byte pieces) or 4
      const uint32_t px = lane & (g_pl - 1u);          // this lane's piece of its row's line
      for (uint32_t i = 0; per_round * i < nready; ++i) {
        const uint32_t idx = per_round * i + (lane >> (7u - g_sh));
        if (idx < nready) {
          const x3_u32x4 e = x3_lds_read_b128(x3_lds_addr(s_prm) + 16u * idx);
          const uint32_t row = e.z & 63u, rb = (e.z >> 8) & 255u, lo = (e.z >> 16) & 31u, hi = (e.z >> 21) & 31u;
          if (px >= lo && px < hi) {
            uint8_t* const dst = reinterpret_cast<uint8_t*>(((uint64_t)e.y << 32) | e.x);
            const uint32_t src = x3_lds_addr(outs) + row * (4u * X3S_RING_DW) + ((rb + (px << g_sh)) & 255u);
            if (p8) x3_store_stream8(dst + 8u * px, x3_lds_read_b64(src));
            else x3_store_stream16(dst + 16u * px, x3_lds_read_b128(src));
          }
        }
      }
      X3_WAVE_LDS_ORDER();
    };
    X3S_PLACE(X3S_PAD_F);
    for (uint32_t b = 0; b < nblk_max; ++b) {
      X3S_PACE_STEP(b, X3S_ROLE_FLUSHER)
      X3_STAMP(0);
      X3S_BARRIER();
      X3_STAMP(4);
      if (!(X3S_KO & 2) && regular && b) {
        const uint32_t k_done = have >> 1;
#ifndef X3S_HALF_LINES
        if (cls8) {
          X3S_FLUSH_CLASS8(0) X3S_FLUSH_CLASS8(1) X3S_FLUSH_CLASS8(2) X3S_FLUSH_CLASS8(3)
          X3S_FLUSH_CLASS8(4) X3S_FLUSH_CLASS8(5) X3S_FLUSH_CLASS8(6) X3S_FLUSH_CLASS8(7)
        } else
#endif
        {
          X3S_FLUSH_CLASS(0) X3S_FLUSH_CLASS(1) X3S_FLUSH_CLASS(2) X3S_FLUSH_CLASS(3)
        }
      }
      if (!regular && b) flush_rows();
#if X3S_THIN && X3S_THIN_F
      {   // TIMING ONLY: what converting 10 - K pairs of the block in place would cost this wave
        const uint32_t* const tb = xfer + ((b + 1u) & 1u) * (X3S_XROWS * 64u);
        uint32_t dz = lane, dn = 0u - 2u, dfw = 1u, dl = 1u, acc = 0;
#pragma unroll
        for (uint32_t q = X3S_THIN_K; q < 10u; ++q) {
          const uint32_t t = tb[(q >> 1) * 128u + 2u * lane + (q & 1u)];
          uint32_t z1, z2, v1, v2, t2, nn1, nn2;
          asm volatile("v_ffbh_u32 %0, %7\n\tv_mad_i32_i24 %4, %0, %8, %9\n\tv_alignbit_b32 %6, %7, 0, %4\n\tv_bfe_u32 %1, %7, %4, %10\n\t"
              "v_ffbh_u32 %2, %6\n\tv_mad_i32_i24 %5, %2, %8, %9\n\tv_bfe_u32 %3, %6, %5, %10"
              : "=&v"(z1), "=&v"(v1), "=&v"(z2), "=&v"(v2), "=&v"(nn1), "=&v"(nn2), "=&v"(t2) : "v"(t), "v"(dz), "v"(dn), "v"(dfw));
          acc = x3_pack_lo16((z1 << dl) + v1, (z2 << dl) + v2);
          s_prm[(q & 3u) * 64u + lane] = acc;     // (a write of the same size, somewhere harmless)
        }
      }
#endif
      const uint32_t cnt = rem < X3S_BL ? rem : X3S_BL;
      rem -= cnt;
      have = 1u + X3S_BL * b + cnt - ((cnt && rem == 0u) ? 0u : 1u);  // staged once the valuer is through block b
      {
        const uint32_t gcnt = g_rem < X3S_BL ? g_rem : X3S_BL;
        g_rem -= gcnt;
        if (gcnt) g_have = 1u + X3S_BL * b + gcnt - (g_rem == 0u ? 0u : 1u);
      }
      X3_STAMP(1);
    }
    X3S_BARRIER();
    if (!regular) {
      flush_rows();
      flush_rows();
    }
    if (regular) {
      const uint32_t k_done = have >> 1;
#ifndef X3S_HALF_LINES
      if (cls8) {
        X3S_FLUSH_CLASS8(0) X3S_FLUSH_CLASS8(1) X3S_FLUSH_CLASS8(2) X3S_FLUSH_CLASS8(3)
        X3S_FLUSH_CLASS8(4) X3S_FLUSH_CLASS8(5) X3S_FLUSH_CLASS8(6) X3S_FLUSH_CLASS8(7)
        X3S_FLUSH_TAILS8()
      } else
#endif
      {
        X3S_FLUSH_CLASS(0) X3S_FLUSH_CLASS(1) X3S_FLUSH_CLASS(2) X3S_FLUSH_CLASS(3)
        X3S_FLUSH_TAILS()
      }
    }
  } else
  if (parser) {
    // ================================================================= wave 0: parser
    fill_ring();
    X3S_BARRIER();  // the valuer has cleared s_dead
    // window: w0 holds `s` unconsumed bits (its low s bits), then w1; widx = ring index of w0.  A pair of
    // codewords is at most 32 bits, so one peek never reaches beyond w1; wn is the word behind w1, re-read from
    // the ring after every consume (the read has a whole pair's time to arrive before the next shift needs it)
    const uint32_t skip = v_bits;                               // (0..16: see abs_base)
    const uint32_t a0 = 8u * (skip & 3u) + ebits;               // bits of the first word in front of the header
    const uint32_t widx0 = (skip >> 2) - (a0 == 0 ? 1u : 0u);  // a0 == 0: start with a fully consumed w0
    uint32_t s = (32u - a0) & 31u;
    uint32_t w0 = X3S_RING_WORD(widx0), w1 = X3S_RING_WORD(widx0 + 1u);
#if X3S_SWP
    // (software-pipelined pair loop: a window of three words, wn is the word behind w2)
    uint32_t w2 = X3S_RING_WORD(widx0 + 2u);
    uint32_t wn = X3S_RING_WORD(widx0 + 3u);
    uint32_t qb = 4u * ~(widx0 + 3u);
#else
    uint32_t wn = X3S_RING_WORD(widx0 + 2u);
    // qb = (4 or 256) * ~(widx + 2): the byte offset of wn's slot before masking; widx itself is only needed by service()
    uint32_t qb = (1u << X3S_QSH) * ~(widx0 + 2u);
#endif
    // consume -nn (<= 32) bits, given as the NEGATIVE count (that is what the codeword walk below has at hand);
    // the word shift is v_bfi with a VGPR mask (see x3_decode_fast_kernel)
    auto consume_to = [&](int32_t s2) {  // s2 = s - bits consumed (>= -32)
      const uint32_t m = (uint32_t)(s2 >> 31);
      s = (uint32_t)s2 & 31u;
      w0 = x3_bfi(m, w1, w0);
#if X3S_SWP
      w1 = x3_bfi(m, w2, w1);
      w2 = x3_bfi(m, wn, w2);
#else
      w1 = x3_bfi(m, wn, w1);
#endif
      uint32_t addr;
      asm("v_lshl_add_u32 %0, %2, " X3S_STR(X3S_QSH) ", %0\n\t"   // widx += 1 on a shift: qb -= 4 (256: transposed ring)
          "v_and_or_b32 %1, %0, %3, %4"
          : "+v"(qb), "=v"(addr) : "v"(m), "v"(X3S_QMASK), "v"(row_base));
      wn = x3_lds_read_b32(addr);
      // keep the read HERE: left to itself the scheduler sinks it to just in front of the next shift, where its
      // whole LDS latency is waited for
      __builtin_amdgcn_sched_barrier(0);
    };
    auto consume_neg = [&](uint32_t nn) { consume_to((int32_t)(s + nn)); };
#if X3S_SWP
    auto ring_index = [&]() -> uint32_t { return ~((uint32_t)((int32_t)qb >> 2)) - 3u; };  // of w0
#else
    auto ring_index = [&]() -> uint32_t { return ~((uint32_t)((int32_t)qb >> X3S_QSH)) - 2u; };  // of w0
#endif

    const uint32_t k_tab = (p.k[1] << 16) | (p.k[2] << 24);  // log2(level) by ftype
    uint32_t over = 0;
    X3S_PLACE(X3S_PAD_P);
    for (uint32_t b = 0; b < nblk_max; ++b) {
      X3S_PACE_STEP(b, X3S_ROLE_PARSER)
      const uint32_t cnt = remaining < X3S_BL ? remaining : X3S_BL;
      remaining -= cnt;
      X3_STAMP(0);
      // a serial decode leaves the segment index behind: where block sb * j begins (the valuer adds the sample in front)
      // (only the entries the index has room for: a frame whose HEADER asks for more blocks than the parameters' frame --
      // a stream encoded with a larger blocks_per_frame, a crafted sample count -- must not write into the next frames' rows
      // or behind the index: ADVICE r5)
      if (sg.out && b && (b % sg.sb) == 0u && cnt && b / sg.sb <= sg.pitch)
        sg.out[1 + f * (uint64_t)sg.pitch + (b / sg.sb - 1u)].x = (uint32_t)((int32_t)(32u * ring_index() + 32u - s) + v_rel);
      if (!(X3S_KO & 16) && (b % X3S_PERIOD) == 0) service(ring_index());
      X3_STAMP(1);
      // block header: 2 bits ftype; ftype 0 -> 4 more bits E-1 (decoder.rs:138-144, 209-216)
      const uint32_t hdr = __builtin_amdgcn_alignbit(w0, w1, s) >> 26;
      // all of it as arithmetic on the 6 bits (no compares: selects on a stale VCC are slow on gfx950)
      const uint32_t ftype = hdr >> 4;
      const uint32_t zmask = (uint32_t)((int32_t)(15u - hdr) >> 31);     // all ones for Rice (hdr >= 16)
      consume_neg((cnt ? 0xFFFFFFFFu : 0u) & ((zmask & 4u) - 6u));       // 6 header bits for BFP, 2 for Rice
      const uint32_t width = x3_bfi(zmask, (1u << ftype) >> 1, (hdr & 15u) + 1u);  // Rice 1,2,4; BFP E
      const uint32_t kk = (k_tab >> (8u * ftype)) & 0xFFu;               // 0, 0, k1, k2
      // what is handed over is the index into the reference's inverse table, i = (z << k) + r with r the k bits
      // BEHIND the terminating one (decoder.rs:186 computes the same as r' + level * (n - 1) with the one
      // included in r'), and the whole field in BFP/literal blocks: the last fw bits of the codeword, and a
      // zero run shifted by 31 there, out of the 16 bits that go across
      const uint32_t fw = x3_bfi(zmask, kk, width);
      const uint32_t lsh = x3_bfi(zmask, kk, 31u);
      const uint32_t nwidth = 0u - width;
      // block buffer: five rows of 64 x 8 bytes (two pair dwords per lane per row), then the 64 header words
      uint32_t* const buf = xfer + (b & 1u) * (X3S_XROWS * 64u);
      buf[X3S_PAIRS * 64u + lane] = hdr;
      X3_STAMP(2);
      if ((X3S_KO & 4)) {
      } else if (__all(cnt == X3S_BL || cnt == 0u)) {
#if X3S_SWP
        {
          // ten pairs, software-pipelined (see X3S_SWP_PAIR).  Even pairs read the ring into wnb and use wn, odd pairs the
          // other way round; pair i packs and stores the indices of pair i - 1 (row i - 1 of the block buffer: one dword per
          // lane), the tail does so for pair 9 and settles the window: w2 up to date, the last read (into wn) arrived.
          uint32_t tt, t2, nn1, nn2, ad, xa, xb, z1, z2, pv1, pv2, wnb, mp;
          int32_t s2;
          const uint32_t bufa = x3_lds_addr(buf) + 4u * lane;
          asm volatile(
              "v_mov_b32 %[mp], 0\n\t"
              X3S_SWP_PAIR("wnb", "wn", "1", "")
              X3S_SWP_PAIR("wn", "wnb", "1", X3S_SWP_STORE(0))
              X3S_SWP_PAIR("wnb", "wn", "2", X3S_SWP_STORE(256))
              X3S_SWP_PAIR("wn", "wnb", "2", X3S_SWP_STORE(512))
              X3S_SWP_PAIR("wnb", "wn", "2", X3S_SWP_STORE(768))
              X3S_SWP_PAIR("wn", "wnb", "2", X3S_SWP_STORE(1024))
              X3S_SWP_PAIR("wnb", "wn", "2", X3S_SWP_STORE(1280))
              X3S_SWP_PAIR("wn", "wnb", "2", X3S_SWP_STORE(1536))
              X3S_SWP_PAIR("wnb", "wn", "2", X3S_SWP_STORE(1792))
              X3S_SWP_PAIR("wn", "wnb", "2", X3S_SWP_STORE(2048))
              "v_lshl_add_u32 %[xa], %[z1], %[lsh], %[pv1]\n\t"
              "v_lshl_add_u32 %[xb], %[z2], %[lsh], %[pv2]\n\t"
              "v_lshl_add_u32 %[qb], %[mp], 2, %[qb]\n\t"
              "v_perm_b32 %[xa], %[xb], %[xa], %[sel]\n\t"
              "ds_write_b32 %[buf], %[xa] offset:2304\n\t"
              "s_waitcnt lgkmcnt(2)\n\t"
              "v_bfi_b32 %[w2], %[mp], %[wn], %[w2]\n\t"
              : [tt] "=&v"(tt), [t2] "=&v"(t2), [nn1] "=&v"(nn1), [nn2] "=&v"(nn2), [ad] "=&v"(ad), [xa] "=&v"(xa),
                [xb] "=&v"(xb), [z1] "=&v"(z1), [z2] "=&v"(z2), [pv1] "=&v"(pv1), [pv2] "=&v"(pv2), [wnb] "=&v"(wnb),
                [mp] "=&v"(mp), [s2] "=&v"(s2), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [wn] "+v"(wn), [s] "+v"(s),
                [qb] "+v"(qb)
              : [zmask] "v"(zmask), [nwidth] "v"(nwidth), [fw] "v"(fw), [lsh] "v"(lsh), [rowb] "v"(row_base), [buf] "v"(bufa),
                [c124] "s"(124u), [sel] "s"(0x05040100u)
              : "memory");
          // the word behind the settled window, as consume_to leaves it (the compiler tracks this read)
          wn = x3_lds_read_b32(x3_and_or(qb, 124u, row_base));
        }
#else
        // two samples per 32-bit peek and per window update (two valid codewords are <= 32 bits)
        uint2* const b2 = reinterpret_cast<uint2*>(buf) + lane;
#pragma unroll
        for (uint32_t j = 0; j < X3S_PAIRS; j += 2) {
          uint32_t X[2];
#pragma unroll
          for (uint32_t e = 0; e < 2; ++e) {
            // a codeword = z zeros + `width` bits (z only counts in Rice blocks): n = z + width bits in all, and
            // its field v = the top n bits of the peek (the zeros in front do not change the value).  With
            // nn = -n = z * zmask - width (one v_mad_i32_i24, zmask being -1 or 0), both "drop n bits"
            // (alignbit by 32 - n) and "the fw bits that end n bits in" (v_bfe_u32 at 32 - n) take nn as their
            // shift count (the hardware uses its low five bits): 4 instructions per codeword.  One asm block, so that the
            // compiler neither pads the dependent chain with s_nop nor reorders it.
            const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
#if X3S_THIN
            // the lengths only: the valuer takes z and the fields from the peek (x3s_fields)
            uint32_t z1, z2, t2, nn1, nn2;
            int32_t s2;
            asm("v_ffbh_u32 %0, %6\n\t"
                "v_mad_i32_i24 %2, %0, %7, %8\n\t"
                "v_alignbit_b32 %4, %6, 0, %2\n\t"
                "v_ffbh_u32 %1, %4\n\t"
                "v_mad_i32_i24 %3, %1, %7, %8\n\t"
                "v_add3_u32 %5, %9, %2, %3"
                : "=&v"(z1), "=&v"(z2), "=&v"(nn1), "=&v"(nn2), "=&v"(t2), "=&v"(s2)
                : "v"(t), "v"(zmask), "v"(nwidth), "v"(s));
            consume_to(s2);
            X[e] = t;
#elif X3S_PAIR_ASM2 && !X3S_SWP
            // (round 5) ... and the sign of the bit counter, the ring byte counter and the address of the next ring word
            // in the same block: a compiler-made instruction that reads what an asm block wrote gets an `s_nop 0` in
            // front (the dst_sel forwarding hazard, assumed for any asm) -- one per pair on the parser's path
            uint32_t z1, z2, v1, v2, t2, nn1, nn2, m, addr;
            int32_t s2;
            asm("v_ffbh_u32 %[z1], %[t]\n\t"
                "v_mad_i32_i24 %[nn1], %[z1], %[zmask], %[nwidth]\n\t"
                "v_alignbit_b32 %[t2], %[t], 0, %[nn1]\n\t"
                "v_bfe_u32 %[v1], %[t], %[nn1], %[fw]\n\t"
                "v_ffbh_u32 %[z2], %[t2]\n\t"
                "v_mad_i32_i24 %[nn2], %[z2], %[zmask], %[nwidth]\n\t"
                "v_bfe_u32 %[v2], %[t2], %[nn2], %[fw]\n\t"
                "v_add3_u32 %[s2], %[s], %[nn1], %[nn2]\n\t"
                "v_ashrrev_i32 %[m], 31, %[s2]\n\t"
                "v_lshl_add_u32 %[qb], %[m], " X3S_STR(X3S_QSH) ", %[qb]\n\t"
                "v_and_or_b32 %[addr], %[qb], %[c124], %[rowb]"
                : [z1] "=&v"(z1), [v1] "=&v"(v1), [z2] "=&v"(z2), [v2] "=&v"(v2), [nn1] "=&v"(nn1), [nn2] "=&v"(nn2),
                  [t2] "=&v"(t2), [s2] "=&v"(s2), [m] "=&v"(m), [addr] "=&v"(addr), [qb] "+v"(qb)
                : [t] "v"(t), [zmask] "v"(zmask), [nwidth] "v"(nwidth), [s] "v"(s), [fw] "v"(fw), [c124] "v"(X3S_QMASK), [rowb] "v"(row_base));
            {
              const uint32_t wn_new = x3_lds_read_b32(addr);
              __builtin_amdgcn_sched_barrier(0);   // (the read stays here: see consume_to)
              s = (uint32_t)s2 & 31u;
              w0 = x3_bfi(m, w1, w0);
              w1 = x3_bfi(m, wn, w1);
              wn = wn_new;
            }
            X[e] = x3_pack_lo16((z1 << lsh) + v1, (z2 << lsh) + v2);
#else
            uint32_t z1, z2, v1, v2, t2, nn1, nn2;
            int32_t s2;
            asm("v_ffbh_u32 %0, %8\n\t"
                "v_mad_i32_i24 %4, %0, %9, %10\n\t"
                "v_alignbit_b32 %6, %8, 0, %4\n\t"
                "v_bfe_u32 %1, %8, %4, %12\n\t"
                "v_ffbh_u32 %2, %6\n\t"
                "v_mad_i32_i24 %5, %2, %9, %10\n\t"
                "v_bfe_u32 %3, %6, %5, %12\n\t"
                "v_add3_u32 %7, %11, %4, %5"
                : "=&v"(z1), "=&v"(v1), "=&v"(z2), "=&v"(v2), "=&v"(nn1), "=&v"(nn2), "=&v"(t2), "=&v"(s2)
                : "v"(t), "v"(zmask), "v"(nwidth), "v"(s), "v"(fw));
            consume_to(s2);
            X[e] = x3_pack_lo16((z1 << lsh) + v1, (z2 << lsh) + v2);
#endif
          }
          b2[(j >> 1) * 64u] = make_uint2(X[0], X[1]);
        }
#endif
      } else {
        // a block that is short in some lane (the last block of a frame): one sample at a time
        uint16_t* const h = reinterpret_cast<uint16_t*>(buf);
        for (uint32_t j = 0; j < X3S_BL; ++j) {
          const uint32_t t = __builtin_amdgcn_alignbit(w0, w1, s);
          const uint32_t z = x3_ffbh(t) & zmask;
          const uint32_t nn = nwidth - z;
          const uint32_t v = (t >> (nn & 31u)) & ((1u << fw) - 1u);
          consume_neg(j < cnt ? nn : 0u);
          h[x3s_half_index(j, lane)] = (uint16_t)((z << lsh) + v);
        }
      }
      // A frame that was READ beyond its payload is decoded again by the reference's reader (s_over): the read position
      // (bits from the ring's first chunk on) against the end of the payload, taken where the lane's frame ENDS.  (Until
      // round 4 it was taken behind the group's last block -- and a lane whose frame is shorter than its neighbours'
      // walks on through what follows its payload until then: every clip's short last frame in a batch of clips was
      // replayed, one thread each, 2.7 ms a step for a thousand clips.)
      if (__any(cnt != 0u && remaining == 0u)) {
        const int32_t widx = (int32_t)ring_index();
        // (a stretch that ends inside the frame: it must have ended where the index says the next one begins)
        if (cnt != 0u && remaining == 0u)
          over = (mid ? (32 * widx + 32 - (int32_t)s + v_rel != (int32_t)exp_bits)
                      : (32 * widx + 32 - (int32_t)s > (int32_t)(8u * v_end))) ? 1u : 0u;
      }
      X3_STAMP(3);
      X3S_BARRIER();
      X3_STAMP(4);
    }
    s_over[lane] = over;
    X3S_BARRIER();
  } else {
    // ================================================================= wave 1: valuer
    // Staging: per lane (= frame = row of the output) a ring of 256 bytes that is indexed by the DESTINATION address:
    // the sample pair that goes to byte address A of wav sits at ring byte (A + rot) & 255, rot a per-lane rotation
    // (a multiple of 16 bytes) that spreads the rows over the LDS banks.  A 128-byte line of HBM is then 128
    // contiguous bytes of the ring, and what leaves the CU are whole, aligned lines: 8 lanes x 16 bytes per line,
    // 8 lines per store instruction.  (Why: the memory side retires a store instruction of eight aligned lines in
    // ~60 % less time than one of 6.4 runs of 160 bytes at 32-byte alignment, which is what windows of four blocks
    // gave -- tools/ubench/store_rate.hip: 0.26 against 0.64 ms for this kernel's 1.38 GB -- and with the runs the
    // stores, not the arithmetic, set this kernel's time and its two timing modes.)
    uint32_t* const orow = outs + lane * X3S_RING_DW;
    const uint32_t orow_b = x3_lds_addr(orow);  // LDS byte address of the row (256-byte aligned)
    int16_t* __restrict__ const o = wav + wo;
    const uint32_t rot = 16u * ((lane >> 2) & 15u);
    const uint32_t pos0 = ((uint32_t)(uintptr_t)o + rot) & 255u;  // ring byte of sample 0 (a multiple of 16)
    bool alive = active, seg_bad = false;
    uint32_t prevP = 0;  // the previous pair; its high half is the last sample so far (pending: even index)
    if (active) {
      const uint32_t first = seg_j ? pred : (((uint32_t)x3[p0] << 8) | x3[p0 + 1]);
      prevP = first << 16;
      if (samples == 1u) o[0] = (int16_t)first;
    }
    s_dead[lane] = 0u;
    X3_WAVE_LDS_ORDER();
    X3S_BARRIER();  // (s_dead is cleared)

    const uint32_t bound_tab = (p.inv_len[0] << 8) | (p.inv_len[1] << 16) | (p.inv_len[2] << 24);  // by ftype (< 256)
#if X3S_THIN
    const uint32_t k_tab_v = (p.k[1] << 16) | (p.k[2] << 24);  // log2(level) by ftype (the parser's k_tab)
    // a pair's peek -> its two indices (z << k) + r / fields, packed: exactly what the parser computed until round 4
    auto x3s_fields = [&](uint32_t t, uint32_t zmask_, uint32_t nwidth_, uint32_t fw_, uint32_t lsh_) __attribute__((always_inline)) -> uint32_t {
      uint32_t z1, z2, v1, v2, t2, nn1, nn2;
      asm("v_ffbh_u32 %0, %7\n\t"
          "v_mad_i32_i24 %4, %0, %8, %9\n\t"
          "v_alignbit_b32 %6, %7, 0, %4\n\t"
          "v_bfe_u32 %1, %7, %4, %10\n\t"
          "v_ffbh_u32 %2, %6\n\t"
          "v_mad_i32_i24 %5, %2, %8, %9\n\t"
          "v_bfe_u32 %3, %6, %5, %10"
          : "=&v"(z1), "=&v"(v1), "=&v"(z2), "=&v"(v2), "=&v"(nn1), "=&v"(nn2), "=&v"(t2)
          : "v"(t), "v"(zmask_), "v"(nwidth_), "v"(fw_));
      return x3_pack_lo16((z1 << lsh_) + v1, (z2 << lsh_) + v2);
    };
#endif
    uint32_t posb = pos0;  // ring byte of the block's first pair, unmasked
    X3S_PLACE(X3S_PAD_V);
    for (uint32_t b = 0; b < nblk_max; ++b, posb += 2u * X3S_BL) {
      X3S_PACE_STEP(b, X3S_ROLE_VALUER)
      X3_STAMP(0);
      X3S_BARRIER();
      X3_STAMP(4);
      const uint32_t cnt = remaining < X3S_BL ? remaining : X3S_BL;
      remaining -= cnt;
      const uint32_t* const buf = xfer + (b & 1u) * (X3S_XROWS * 64u);
      const uint32_t hdr = buf[X3S_PAIRS * 64u + lane];
      if (sg.out && b && (b % sg.sb) == 0u && cnt && b / sg.sb <= sg.pitch)   // (the segment index of a serial decode: the sample in front of block b)
        sg.out[1 + f * (uint64_t)sg.pitch + (b / sg.sb - 1u)].y = (prevP >> 16) | X3S_SEG_VALID;
      // block parameters from the header bits, as arithmetic (see the parser)
      const uint32_t ftype = hdr >> 4;
      const uint32_t E = (hdr & 15u) + 1u;
      const uint32_t zmask = (uint32_t)((int32_t)(15u - hdr) >> 31);      // all ones for Rice
      const bool bfp = zmask == 0u;
      const uint32_t litmask = ~zmask & (uint32_t)((int32_t)(14u - (hdr & 15u)) >> 31);  // BFP with E == 16
      const uint32_t neg_thresh = ~zmask & (1u << (E - 1u));
      const uint32_t neg2 = (neg_thresh << 1) & ~litmask;
      const uint32_t bound = x3_bfi(zmask, (bound_tab >> (8u * ftype)) & 0xFFu, 0xFFFFFFFFu);
      if (cnt && alive && bfp && E <= 5u) {  // decoder.rs:209-216
        st = X3D_FRAME_DECODE_INVALID_BPF;
        alive = false;
      }
#if X3S_THIN
      // the codeword geometry of this lane's block, as the parser derives it (thin parser: the walk is repeated here)
      const uint32_t t_width = x3_bfi(zmask, (1u << ftype) >> 1, E);                 // Rice 1,2,4; BFP E
      const uint32_t t_kk = (k_tab_v >> (8u * ftype)) & 0xFFu;                       // 0, 0, k1, k2
      const uint32_t t_fw = x3_bfi(zmask, t_kk, t_width);
      const uint32_t t_lsh = x3_bfi(zmask, t_kk, 31u);
      const uint32_t t_nwidth = 0u - t_width;
#endif
      const uint32_t tm12 = ((neg_thresh - 1u) & 0xFFFFu) * 0x10001u;  // thresh - 1 in each half
      const uint32_t neg22 = (neg2 & 0xFFFFu) * 0x10001u;              // 2 * thresh = 2^E (0 for a literal block)
      uint32_t maxii2 = 0;
      X3_STAMP(2);

      if ((X3S_KO & 1)) {
      } else if (__all(cnt == X3S_BL || cnt == 0u)) {
        // the whole block's indices at once (five 8-byte reads in flight), then ten pairs from registers
        uint2 XX[5];
#if X3S_SWP
#pragma unroll
        for (uint32_t r = 0; r < 5u; ++r) XX[r] = make_uint2(buf[(2u * r) * 64u + lane], buf[(2u * r + 1u) * 64u + lane]);
#else
        const uint2* const b2 = reinterpret_cast<const uint2*>(buf) + lane;
#pragma unroll
        for (uint32_t r = 0; r < 5u; ++r) XX[r] = b2[r * 64u];
#endif
#if X3S_VALUER_FOLD
        // Round 4: the Rice / BFP choice is in the CONSTANTS, not in a select behind both computations.  A Rice lane's
        // thresh is 0 (B = X - ((X + 0xFFFF) & 0) = X: the BFP step is the identity there); a BFP lane shifts by 0 and
        // takes no sign bit (zsh2 = 0: the zigzag step is the identity there).  One instruction per pair less than
        // R, B and v_bfi -- and the literal select (0.1 % of config 3's blocks) only runs in blocks in which some lane
        // of the wave HAS a literal block (6 % of them).
        const uint32_t zsh2 = zmask & 0x00010001u;   // per half: shift and sign-bit mask of the zigzag step
        auto pairs = [&](auto lit_tag) __attribute__((always_inline)) {
          constexpr bool LIT = decltype(lit_tag)::value;
#pragma unroll
          for (uint32_t r = 0; r < 5u; ++r) {
            uint32_t W[2];
#pragma unroll
            for (uint32_t e = 0; e < 2; ++e) {
#if X3S_THIN
              const uint32_t X = (2u * r + e < X3S_THIN_K) ? x3s_fields(e ? XX[r].y : XX[r].x, zmask, t_nwidth, t_fw, t_lsh) : (e ? XX[r].y : XX[r].x);
#else
              const uint32_t X = e ? XX[r].y : XX[r].x;
#endif
              maxii2 = x3_pk_max_u16(maxii2, X);   // (Rice: X = i, the index into the inverse table, decoder.rs:186)
              // BFP: unsigned_to_i16 (decoder.rs:198-207): v - (v > thresh ? 2*thresh : 0), strict compare.
              // v < 2^E and thresh = 2^(E-1), so bit E of v + thresh - 1 says v > thresh.
              const uint32_t B = x3_pk_sub_u16(X, x3_pk_add_u16(X, tm12) & neg22);
              // Rice: the inverse table is a zigzag (x3.rs:200-204)
              const uint32_t D = x3_pk_lshr_b16(B, zsh2) ^ x3_pk_sub_u16(0u, B & zsh2);   // (d1, d2)
              // (last + d1, last + d1 + d2), last = the high half of the previous pair: both halves of D plus last,
              // then d1 once more onto the high half
              uint32_t P = x3_pk_mad_u16_alo(D, 0x00010000u, x3_pk_add_u16_bhi(D, prevP));
              if (LIT) P = x3_bfi(litmask, X, P);                          // literal: field = sample
              W[e] = __builtin_amdgcn_alignbit(P, prevP, 16);                // (pending sample, la)
              prevP = P;
            }
            // samples 20 b + 4 r .. + 3: eight bytes of the ring (positions are multiples of 8: no wrap inside)
            x3_lds_write_b64(x3_and_or(posb + 8u * r, 248u, orow_b), W[0], W[1]);
          }
        };
        if (__any(litmask != 0u)) pairs(std::true_type{}); else pairs(std::false_type{});
#elif X3S_VALUER_ASM && !X3S_THIN
        // Round 5: the thirteen instructions of a pair as ONE asm block.  As single-instruction helpers the compiler puts
        // an `s_nop 0` behind most of them (it cannot see into an asm and assumes that one which feeds the next may have
        // written half a register: the dst_sel forwarding hazard of gfx94x -- packed instructions write whole registers);
        // forty issue slots per block in this wave, a quarter more than its arithmetic.  Same arithmetic, same order.
        {
          const uint32_t c10001 = 0x00010001u, c10000 = 0x00010000u;
#pragma unroll
          for (uint32_t r = 0; r < 5u; ++r) {
            uint32_t W[2];
#pragma unroll
            for (uint32_t e = 0; e < 2; ++e) {
              const uint32_t X = e ? XX[r].y : XX[r].x;
              uint32_t t0, t1, t2, P;
              asm("v_pk_max_u16 %[mx], %[mx], %[X]\n\t"
                  "v_and_b32 %[t0], %[c10001], %[X]\n\t"
                  "v_pk_lshrrev_b16 %[t1], 1, %[X] op_sel_hi:[0,1]\n\t"
                  "v_pk_sub_u16 %[t0], 0, %[t0]\n\t"
                  "v_pk_add_u16 %[t2], %[X], %[tm12]\n\t"
                  "v_xor_b32 %[t1], %[t0], %[t1]\n\t"
                  "v_and_b32 %[t2], %[t2], %[neg22]\n\t"
                  "v_pk_sub_u16 %[t2], %[X], %[t2]\n\t"
                  "v_bfi_b32 %[t1], %[zmask], %[t1], %[t2]\n\t"
                  "v_pk_add_u16 %[t2], %[t1], %[prev] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                  "v_pk_mad_u16 %[t1], %[t1], %[c10000], %[t2] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                  "v_bfi_b32 %[P], %[litmask], %[X], %[t1]\n\t"
                  "v_alignbit_b32 %[W], %[P], %[prev], 16"
                  : [mx] "+v"(maxii2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [P] "=&v"(P), [W] "=&v"(W[e])
                  : [X] "v"(X), [tm12] "v"(tm12), [neg22] "v"(neg22), [zmask] "v"(zmask), [litmask] "v"(litmask), [prev] "v"(prevP),
                    [c10001] "s"(c10001), [c10000] "v"(c10000));
              prevP = P;
            }
            x3_lds_write_b64(x3_and_or(posb + 8u * r, 248u, orow_b), W[0], W[1]);
          }
        }
#else
#pragma unroll
        for (uint32_t r = 0; r < 5u; ++r) {
          uint32_t W[2];
#pragma unroll
          for (uint32_t e = 0; e < 2; ++e) {
#if X3S_THIN
            const uint32_t X = (2u * r + e < X3S_THIN_K) ? x3s_fields(e ? XX[r].y : XX[r].x, zmask, t_nwidth, t_fw, t_lsh) : (e ? XX[r].y : XX[r].x);
#else
            const uint32_t X = e ? XX[r].y : XX[r].x;
#endif
            // Rice: X = i, the index into the inverse table (decoder.rs:186), which is a zigzag (x3.rs:200-204)
            maxii2 = x3_pk_max_u16(maxii2, X);
            const uint32_t R = x3_pk_lshr_b16_1(X) ^ x3_pk_sub_u16(0u, X & 0x00010001u);
            // BFP: unsigned_to_i16 (decoder.rs:198-207): v - (v > thresh ? 2*thresh : 0), strict compare.
            // v < 2^E and thresh = 2^(E-1), so bit E of v + thresh - 1 says v > thresh.
            const uint32_t B = x3_pk_sub_u16(X, x3_pk_add_u16(X, tm12) & neg22);
            const uint32_t D = x3_bfi(zmask, R, B);                        // (d1, d2)
            // (last + d1, last + d1 + d2), last = the high half of the previous pair: both halves of D plus last,
            // then d1 once more onto the high half
            uint32_t P = x3_pk_mad_u16_alo(D, 0x00010000u, x3_pk_add_u16_bhi(D, prevP));
            P = x3_bfi(litmask, X, P);                                     // literal: field = sample
            W[e] = __builtin_amdgcn_alignbit(P, prevP, 16);                // (pending sample, la)
            prevP = P;
          }
          // samples 20 b + 4 r .. + 3: eight bytes of the ring (positions are multiples of 8: no wrap inside)
          x3_lds_write_b64(x3_and_or(posb + 8u * r, 248u, orow_b), W[0], W[1]);
        }
#endif
      } else {
        // short block somewhere in the group: one sample at a time, staged as halfwords
        const uint16_t* const h = reinterpret_cast<const uint16_t*>(buf);
        if (cnt) x3_lds_write_u16(x3_and_or(posb, 254u, orow_b), prevP >> 16);
        uint32_t last = prevP >> 16, maxii = 0;
        for (uint32_t j = 0; j < X3S_BL; ++j) {
          if (j < cnt) {
            const uint32_t x = h[x3s_half_index(j, lane)];
            const uint32_t ii = x;
            const uint32_t d_rice = (ii >> 1) ^ (0u - (ii & 1u));
            const uint32_t d_bfp = x - (x > neg_thresh ? neg2 : 0u);
            const uint32_t d = bfp ? d_bfp : d_rice;
            last = litmask ? x : ((last + d) & 0xFFFFu);
            maxii = bfp ? maxii : (ii > maxii ? ii : maxii);
            x3_lds_write_u16(x3_and_or(posb + 2u * (j + 1u), 254u, orow_b), last);
          }
        }
        maxii2 = maxii;
        prevP = last << 16;
      }
      X3_STAMP(3);
      // OutOfBoundsInverse (decoder.rs:160,187): the frame stops here
      if (cnt && alive && max(maxii2 & 0xFFFFu, maxii2 >> 16) >= bound) {
        st = X3D_OUT_OF_BOUNDS_INVERSE;
        alive = false;
      }
      const bool finished = cnt && remaining == 0;  // this lane's frame is complete
      if (finished && alive && (samples & 1u) && !mid)      // its last sample is a pending one: stage it
        x3_lds_write_u16(x3_and_or(pos0 + 2u * (samples - 1u), 254u, orow_b), prevP >> 16);
      // (the last sample of a stretch that ends inside the frame is the next stretch's first: not written, compared)
      if (finished && mid && (prevP >> 16) != (exp_pred ^ X3S_SEG_VALID)) seg_bad = true;
      X3_STAMP(1);
      if (!alive) s_dead[lane] = 1u;  // (visible to the flusher behind the next barrier)
      // (the flusher moves whole 16-byte pieces.)  The ragged end of a frame, fewer than eight samples:
      if (!regular && finished && alive)
        for (uint32_t sx = rowlen & (p8 ? ~3u : ~7u); sx < rowlen; ++sx)
          o[sx] = (int16_t)x3_lds_read_u16(x3_and_or(pos0 + 2u * sx, 254u, orow_b), 0u);
      X3_STAMP(5);
    }
    X3S_BARRIER();
    if (active && (st == X3D_OUT_OF_BOUNDS_INVERSE || st == X3D_FRAME_DECODE_INVALID_BPF || s_over[lane] || seg_bad))
      st = X3D_REPLAY;  // the reference's reader decides (x3_decode_replay.h; x3_decode_merge_kernel)
    if (f < n_frames) {
      // (stretches: the host has cleared the array, the stretches of a frame are in different workgroups; X3D_REPLAY is the
      // largest status, and a frame's other statuses -- its header's -- are the same in all of its stretches)
      if (!seg_grid) status[f] = st;
      else if (st != X3D_OK) atomicMax(&status[f], st);
    }
    if (lane == 0 && (nblk_max >= 64u || (segd && nblk_max))) {  // this group's pace, for the next launch
      uint64_t t16 = ((unsigned long long)(clock64() - pace_t0) * 16u) / nblk_max;
      if (t16 >= (1u << X3S_PACE_EPOCH_SHIFT)) t16 = (1u << X3S_PACE_EPOCH_SHIFT) - 1u;
      if (!segd) {   // (launches by stretches are not paced and leave no pace behind)
        atomicMax(pace + (pace_epoch & 1u), ((pace_epoch & 0xFFFu) << X3S_PACE_EPOCH_SHIFT) | (uint32_t)t16);
        if (blockIdx.x == 0) {
          pace[2u + (pace_epoch & 1u)] = ((pace_epoch & 0xFFFu) << X3S_PACE_EPOCH_SHIFT) | pace_target;
          pace[8u + (pace_epoch & 1u)] = ((pace_epoch & 0xFFFu) << X3S_PACE_EPOCH_SHIFT) | (gridDim.x & ((1u << X3S_PACE_EPOCH_SHIFT) - 1u));
        }
      }
      // The launch log (x3_ctx_launch_log; bench.py's per-step list): what this launch aimed at, what its slowest group
      // achieved, and the shader clock it ran at -- group 0's shader ticks against the 100 MHz clock over its whole
      // life (MI355X_MICROARCH.md, DVFS: a box that runs this kernel at 2.0 GHz instead of 2.3 shows HERE, and a line
      // with the clock in it lets a reader tell a slow box from a controller that has not settled).  Nothing in the
      // kernel reads these words.
      uint32_t* const lg = pace + X3_LOG_BASE + X3_LOG_WORDS * (pace_epoch & (X3_LOG_ENTRIES - 1u));
      atomicMax(lg, ((pace_epoch & 0xFFFu) << X3S_PACE_EPOCH_SHIFT) | (uint32_t)t16);
      if (blockIdx.x == 0) {
        lg[1] = ((pace_epoch & 0xFFFu) << X3S_PACE_EPOCH_SHIFT) | pace_target;
        lg[2] = (uint32_t)(clock64() - clk_t0);
        lg[3] = (uint32_t)(wall_clock64() - wall_t0);
      }
    }
  }
#ifdef X3_DBG_STAMPS
  dbg_acc[6] = dbg_start;          // constant-rate clock: which groups were resident together
  dbg_acc[7] = wall_clock64();
  {
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const unsigned long long where = ((unsigned long long)xcc_id << 32) | hw_id;
    if (parser) dbg_acc[5] = where; else dbg_acc[6] = where;
  }
  if (lane == 0 && blockIdx.x < 2048)
    for (int k = 0; k < 8; ++k) x3_dbg[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + k] = dbg_acc[k];
#endif
}
