#!/usr/bin/env python3
"""three or four codewords per 32-bit peek in the three-wave decoder's parser.  How often do the
codewords of an aligned quad (samples 4q .. 4q+3 of a block) or triple of ALL 64 lanes of a group fit 32 bits, on bench.py's
content?  (CPU only: code lengths from the signal, the encoder's block types by its thresholds.)  The lanes of a group walk
in lockstep, so one lane whose quad is longer sends the whole wave down the pair path for that quad."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "x3-rust_amd"))
import x3hip

F = 1024
for kind, name in ((2, "hydrophone noise (bench.py)"), (4, "kind 4"), (1, "kind 1")):
    wav = x3hip.synth(kind, 0x58330003, 0, 10000 * F).astype(np.int64).reshape(F, 10000)
    d = np.diff(wav, axis=1)[:, :9980].reshape(F, 499, 20)
    mx = np.abs(d).max(axis=2)
    ft = np.where(mx <= 3, 1, np.where(mx <= 8, 2, np.where(mx <= 20, 3, 0)))      # x3.rs thresholds 3, 8, 20
    k = np.array([0, 0, 1, 3])[ft]
    i = np.where(d > 0, 2 * d - 1, -2 * d)                                           # index into the inverse (zigzag) table
    rice = (i >> k[..., None]) + 1 + k[..., None]
    E = np.maximum(np.ceil(np.log2(mx + 1)).astype(int) + 1, 6)
    E = np.minimum(E, 16)
    ln = np.where(ft[..., None] > 0, rice, E[..., None])
    q = ln.reshape(F, 499, 5, 4).sum(axis=3)
    t = ln[..., :18].reshape(F, 499, 6, 3).sum(axis=3)
    anyq = (q > 32).reshape(F // 64, 64, 499, 5).any(axis=1)
    anyt = (t > 32).reshape(F // 64, 64, 499, 6).any(axis=1)
    pq, pt = anyq.mean(), anyt.mean()
    print("%-28s %.2f bits/sample; block types BFP/R0/R1/R3 = %s" % (name, ln.mean(), [round(float((ft == x).mean()), 3) for x in range(4)]))
    print("    a lane's quad > 32 bits: %.4f; some lane of the 64: %.3f of the quad slots; triples: %.3f" % ((q > 32).mean(), pq, pt))
    # walk cost per four samples: pairs 38; quad path 27 + 2 (compare + branch), and the pair path behind a failed quad
    print("    walk instructions per four samples: 38 now -> %.1f with quads (27 + 2, + 38 where a lane does not fit)" % (29 + 38 * pq))
