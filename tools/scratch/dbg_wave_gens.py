#!/usr/bin/env python3
"""Many generations on small inputs (wave_nwg / wave_m) against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import x3hip, oracle_lib as O
ctx = x3hip.Context(0); ctx.set_option("verbose", 1)
bad = 0
for nwg, m in ((1, 1), (1, 2), (2, 1), (3, 5), (4, 16), (7, 3), (16, 16), (64, 2), (256, 1)):
    ctx.set_option("wave_nwg", nwg); ctx.set_option("wave_m", m)
    for n in (10000 * 37 + 123, 10000 * 200, 10000 * 513 + 1):
        wav = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 0x58330002, 0, n)
        rc_o, s_o, st_o = O.encode(wav)
        for rep in range(3):
            f0 = ctx.get_option("encode_fallbacks")
            rc, s, st = ctx.encode(wav)
            f1 = ctx.get_option("encode_fallbacks")
            ok = rc == rc_o and len(s) == len(s_o) and np.array_equal(s, s_o) and st.tolist() == st_o.tolist()
            if not ok or f1 > f0: bad += 1
            print("%s nwg %d m %d n %d rep %d%s" % ("ok  " if ok else "FAIL", nwg, m, n, rep, "  [TIMEOUT fallback]" if f1 > f0 else ""), flush=True)
print("FAILED %d" % bad if bad else "ALL OK")
