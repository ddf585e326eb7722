#!/usr/bin/env python3
"""soak seed 741 trial 10315: blocks of 40, codes (2,1,1), thresholds (8,6,11), a capacity both encoders run out of --
the GPU said BAD_ARG (24) where the oracle says ByteWriterInsufficientMemory (22)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_lib as O, x3hip, fuzz_parity as fz
rng = np.random.default_rng([741, 10315])
k = "nng"[int(rng.integers(0, 3))]
bl = int(rng.choice([10, 40])); per = 20 // bl if bl < 20 else 1
bpf = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 50, 100, 125, 250, 255, 256])) * (2 if bl == 10 else 1) if rng.random() < 0.7 else int(rng.integers(1, 257 * max(per, 1)))
codes, thr = (0, 1, 3), (3, 8, 20)
if rng.random() < 0.2:
    offsets = [6, 11, 20, 28]
    codes = tuple(int(c) for c in rng.integers(0, 4, size=3))
    thr = tuple(int(rng.integers(0, offsets[c] + 1)) for c in codes)
p = x3hip.Params.make(bl, bpf, codes, thr); po = O.Params.make(bl, bpf, codes, thr)
spf = bl * bpf
frames = int(rng.choice([1, 2, 3, 5, 40, 300, 1500])) if spf <= 400 else int(rng.choice([1, 2, 3, 7, 30, 90]))
n = max(1, spf * frames - int(rng.integers(0, spf)) + int(rng.integers(0, 3)))
wav = fz.content(rng, n)
sp = int(rng.choice([0, 0, 1, 2, 3, 18]))
g2 = rng.random() < 0.4
cut = float(rng.uniform(0.0, 0.3)) if rng.random() < 0.1 else None
print(k, bl, bpf, codes, thr, n, sp, g2, cut)
cap_full = sp + 64 + ((wav.size + spf - 1) // spf) * 84 + 3 * wav.size
ctx = x3hip.Context(0)
for c in (cut, None, 0.5, 0.1):
    cap = int(c * cap_full) if c else cap_full
    ro = O.encode(wav, po, start_pos=sp, cap=cap)
    for gen in (3, 2, 1):
        for chunk in (8, -1):
            ctx.set_option("enc_gen", gen); ctx.set_option("host_chunk_frames", chunk)
            rg = ctx.encode(wav, p, start_pos=sp, cap=cap)
            print("cut", c, "gen", gen, "in use", ctx.get_option("enc_gen_in_use"), "chunk", chunk, "oracle rc", ro[0], ro[1].size, "gpu rc", rg[0], rg[1].size,
                  "same" if rg[0] == ro[0] and np.array_equal(rg[1][sp:], ro[1][sp:]) else "DIFFERENT", ctx.last_error())
