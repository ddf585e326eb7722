/* The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; built and run by
 * tests/test_oracle_sanitized.py): random signals through x3o_encode / x3o_decode_stream / x3o_x3a_*, tampered and
 * truncated streams, exact-size buffers -- the checker itself must not read or write out of bounds on any of it.
 * (A restated UB of the reference, e.g. shifts that release-mode Rust masks, must be restated as defined arithmetic.) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/x3_oracle.h"

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); }

static void fill(int16_t* w, size_t n) {
  size_t i = 0;
  while (i < n) {
    size_t seg = 1 + rnd() % 3000;
    if (seg > n - i) seg = n - i;
    int kind = rnd() % 6, acc = 0;
    int amp = (int[]){1, 3, 8, 20, 300, 20000}[rnd() % 6];
    for (size_t k = 0; k < seg; ++k) {
      int v;
      switch (kind) {
        case 0: v = 0; break;
        case 1: v = (int)(rnd() % 65536) - 32768; break;
        case 2: acc += (int)(rnd() % (2 * amp + 1)) - amp; if (acc > 32767) acc = 32767; if (acc < -32768) acc = -32768; v = acc; break;
        case 3: v = (k & 1) ? 32767 : -32768; break;
        case 4: v = (int)(rnd() % 7) - 3; break;
        default: v = (int)(rnd() % (2 * amp + 1)) - amp; break;
      }
      w[i + k] = (int16_t)v;
    }
    i += seg;
  }
}

int main(int argc, char** argv) {
  int trials = argc > 1 ? atoi(argv[1]) : 300;
  for (int t = 0; t < trials; ++t) {
    x3o_params p;
    x3o_params_default(&p);
    if (t % 3 == 1) { p.block_len = 1 + rnd() % 60; p.blocks_per_frame = 1 + rnd() % 80; }
    if (t % 7 == 3) { p.block_len = 20; p.blocks_per_frame = 1 + rnd() % 600; }
    size_t n = 1 + rnd() % 60000;
    int16_t* wav = malloc(n * sizeof *wav);
    fill(wav, n);
    size_t spf = (size_t)p.block_len * p.blocks_per_frame, nf = (n + spf - 1) / spf;
    uint64_t cap = 64 + nf * 84 + 3 * n + (rnd() % 3);
    if (t % 11 == 5) cap = rnd() % (cap + 1);                /* a writer that runs out */
    uint8_t* out = malloc(cap ? cap : 1);                     /* exact size: ASan sees the first byte beyond it */
    uint64_t pos = 0, stats[6];
    uint64_t start = rnd() % 4;
    if (start > cap) start = cap;
    int rc = x3o_encode(wav, n, 1, &p, out, cap, start, &pos, stats);
    if (rc == 0) {
      uint64_t body = (start + 1) & ~1ull, len = pos - body;
      uint8_t* s = malloc(len ? len : 1);
      memcpy(s, out + body, len);
      int dmg = rnd() % 4;
      for (int d = 0; d < dmg && len > 24; ++d) {
        uint64_t q = rnd() % len;
        switch (rnd() % 3) {
          case 0: s[q] ^= (uint8_t)(1u << (rnd() % 8)); break;
          case 1: memset(s + q, 0, (len - q) < 9 ? (size_t)(len - q) : 9); break;
          default: s[q] = (uint8_t)rnd(); break;
        }
      }
      if (dmg) {  /* refresh the CRCs of every frame the (possibly damaged) chain still reaches */
        uint64_t o = 0;
        while (o + 20 <= len) {
          uint32_t plen = (uint32_t)s[o + 6] << 8 | s[o + 7];
          if (o + 20 + plen > len) break;
          if (rnd() % 2) {
            uint16_t pc = x3o_crc16(s + o + 20, plen), hc;
            s[o + 18] = (uint8_t)(pc >> 8); s[o + 19] = (uint8_t)pc;
            hc = x3o_crc16(s + o, 16);
            s[o + 16] = (uint8_t)(hc >> 8); s[o + 17] = (uint8_t)hc;
          }
          o += 20 + plen;
        }
      }
      uint64_t cut = (rnd() % 4 == 0) ? rnd() % (len + 1) : len;
      uint8_t* s2 = malloc(cut ? cut : 1);
      memcpy(s2, s, cut);
      uint64_t wcap = (rnd() % 5 == 0) ? rnd() % (n + 1) : n + 70000;
      int16_t* back = malloc((wcap ? wcap : 1) * sizeof *back);
      uint64_t got = 0, fok = 0, ferr = 0;
      int drc = x3o_decode_stream(s2, cut, &p, back, wcap, &got, &fok, &ferr);
      if (!dmg && cut == len && wcap >= n && p.codes[0] == 0 && p.thresholds[2] >= 16) {
        if (drc != 0 || got != n || memcmp(back, wav, n * sizeof *wav)) { fprintf(stderr, "round trip failed at trial %d\n", t); return 1; }
      }
      free(back); free(s2); free(s);
    }
    if (t % 5 == 0) {  /* the archive level, default parameters */
      uint64_t acap = 1024 + 64 + ((n + 9999) / 10000) * 84 + 3 * n, alen = 0;
      uint8_t* a = malloc(acap);
      if (x3o_x3a_encode(wav, n, 1 + rnd() % 400000, a, acap, &alen, stats) == 0) {
        if (rnd() % 2 && alen > 40) a[rnd() % alen] ^= (uint8_t)(1u << (rnd() % 8));
        uint64_t alen2 = (rnd() % 3 == 0) ? rnd() % (alen + 1) : alen;
        uint8_t* a2 = malloc(alen2 ? alen2 : 1);
        memcpy(a2, a, alen2);
        int16_t* back = malloc((n + 70000) * sizeof *back);
        uint64_t got = 0, fok = 0, ferr = 0; uint32_t rate = 0;
        (void)x3o_x3a_decode(a2, alen2, back, n + 70000, &got, &rate, &fok, &ferr);
        free(back); free(a2);
      }
      free(a);
    }
    free(out); free(wav);
  }
  printf("oracle_sanitize: %d trials clean\n", trials);
  return 0;
}
