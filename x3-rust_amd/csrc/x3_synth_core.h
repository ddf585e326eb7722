// x3_synth_core.h -- seeded, integer-only synthetic audio, shared host/device.
//
// Every sample is a pure function of (kind, seed, absolute sample index): signals are generated
// per SEGMENT of X3_SYNTH_SEG samples from a hash of (seed, segment index), so the host and the
// GPU produce bit-identical data and any sub-range can be generated on its own.  No libm.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define X3_HD __host__ __device__ inline
#else
#define X3_HD static inline
#endif

#define X3_SYNTH_SEG 4096u

enum { X3_SYNTH_ZEROS = 0, X3_SYNTH_WHITE = 1, X3_SYNTH_HYDROPHONE = 2, X3_SYNTH_SINE = 3, X3_SYNTH_WALK = 4 };

X3_HD uint64_t x3_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

X3_HD int32_t x3_sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// triangle "swell": period 384000 samples (2 s at 192 kHz), amplitude 2000
X3_HD int32_t x3_swell(uint64_t n) {
  const uint32_t P = 384000u, H = P / 2;
  uint32_t t = (uint32_t)(n % P);
  return t < H ? (-2000 + (int32_t)((4000u * (uint64_t)t) / H)) : (2000 - (int32_t)((4000u * (uint64_t)(t - H)) / H));
}

// parabolic fixed-point "sine": 16-bit phase accumulator, amplitude amp
X3_HD int32_t x3_parasine(uint64_t n, uint32_t step, int32_t amp) {
  uint32_t ph = (uint32_t)((n * (uint64_t)step) & 0xFFFFu);
  int32_t half = ph < 32768u ? 1 : -1;
  int32_t x = (int32_t)(ph & 32767u) - 16384;                      // [-16384, 16383]
  int64_t y = (int64_t)amp - (((int64_t)amp * x * x) >> 28);       // amp * (1 - (x/16384)^2)
  return half * (int32_t)y;
}

// Generate samples [seg*SEG + lo, seg*SEG + hi) of segment `seg` into out[0 .. hi-lo).
// `emit(i, v)` is called for every in-range sample index i (relative to lo).
template <class Emit>
X3_HD void x3_synth_segment(int kind, uint64_t seed, uint64_t seg, uint32_t lo, uint32_t hi, Emit emit) {
  const uint64_t base = seg * (uint64_t)X3_SYNTH_SEG;
  if (kind == X3_SYNTH_ZEROS) {
    for (uint32_t i = lo; i < hi; ++i) emit(i - lo, (int16_t)0);
    return;
  }
  const uint64_t sk = x3_splitmix64(seed ^ (seg * 0xD1B54A32D192ED03ull + 0x5833u));
  if (kind == X3_SYNTH_WHITE) {
    for (uint32_t i = lo; i < hi; ++i) {
      uint64_t r = x3_splitmix64(sk + (i >> 2));
      emit(i - lo, (int16_t)(uint16_t)(r >> (16 * (i & 3))));
    }
    return;
  }
  if (kind == X3_SYNTH_SINE) {
    // 440 Hz-ish at 16 kHz: step = 65536*440/16000 = 1802; amplitude 12000 (mostly BFP blocks)
    for (uint32_t i = lo; i < hi; ++i) emit(i - lo, (int16_t)x3_parasine(base + i, 1802u, 12000));
    return;
  }
  if (kind == X3_SYNTH_WALK) {
    int32_t v = (int32_t)(sk % 16385u) - 8192;
    for (uint32_t i = 0; i < hi; ++i) {
      uint64_t r = x3_splitmix64(sk + 1 + (i >> 4));
      int32_t st = (int32_t)((r >> (4 * (i & 15))) & 15u) % 5 - 2;  // -2..2
      v = x3_sat16(v + st);
      if (i >= lo) emit(i - lo, (int16_t)v);
    }
    return;
  }
  // X3_SYNTH_HYDROPHONE: coloured noise whose level changes per segment + swell + sparse clicks
  // per-segment noise level, one of {1,2,2,3,5,8,14,40} (packed bytes: no static table in device code)
  const int32_t A = (int32_t)((0x280E080503020201ull >> (8 * ((sk >> 8) & 7))) & 0xFFu);
  const bool has_click = ((sk >> 20) & 7u) == 0;                   // p = 1/8 per segment ~ 2^-15 per sample
  const uint32_t cpos = (uint32_t)((sk >> 24) % X3_SYNTH_SEG);
  const int32_t camp = (int32_t)(6000u + (uint32_t)((sk >> 40) % 24000u)) * (((sk >> 60) & 1) ? 1 : -1);
  int32_t x = 0;  // IIR state, Q3
  for (uint32_t i = 0; i < hi; ++i) {
    uint64_t r = x3_splitmix64(sk + 1 + (i >> 2));
    int32_t u16v = (int32_t)((r >> (16 * (i & 3))) & 0xFFFFu);
    int32_t u = (u16v % (2 * 8 * A + 1)) - 8 * A;                  // uniform in [-8A, 8A] (Q3)
    x = x - (x >> 2) + u;                                          // one-pole low-pass
    if (i < lo) continue;
    int32_t v = (x >> 3) + x3_swell(base + i);
    if (has_click && i >= cpos && i < cpos + 64u) {
      int32_t k = (int32_t)(i - cpos);
      int32_t c = (camp * (64 - k)) / 64;
      v += (k & 1) ? -c : c;
    }
    emit(i - lo, (int16_t)x3_sat16(v));
  }
}
