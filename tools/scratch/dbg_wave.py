#!/usr/bin/env python3
"""Differential check of the wave-per-frame encoder (enc_gen 3) against the oracle, with a diagnosis of the first
difference.  usage: dbg_wave.py [quick]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import x3hip, oracle_lib as O

ctx = x3hip.Context(0)
ctx.set_option("enc_gen", 3)
bad = 0

def frames_of(stream):
    out, off = [], 0
    while off + 20 <= len(stream):
        plen = (int(stream[off + 6]) << 8) | int(stream[off + 7])
        out.append((off, plen))
        off += 20 + plen
    return out

def check(tag, wav, start_pos=0):
    global bad
    rc_o, s_o, st_o = O.encode(wav, start_pos=start_pos)
    r0 = ctx.get_option("encode_dense_reruns"); f0 = ctx.get_option("encode_fallbacks")
    rc, s, st = ctx.encode(wav, start_pos=start_pos)
    r1 = ctx.get_option("encode_dense_reruns"); f1 = ctx.get_option("encode_fallbacks")
    note = ("  [dense rerun]" if r1 > r0 else "") + ("  [TIMEOUT fallback]" if f1 > f0 else "")
    if rc != rc_o:
        print("FAIL %-28s rc %d vs oracle %d %s %s" % (tag, rc, rc_o, ctx.last_error(), note)); bad += 1; return
    if rc != 0:
        print("ok   %-28s rc %d (both)%s" % (tag, rc, note)); return
    if len(s) == len(s_o) and np.array_equal(s, s_o) and st.tolist() == st_o.tolist():
        print("ok   %-28s %9d samples -> %9d bytes%s" % (tag, wav.size, len(s), note)); return
    bad += 1
    print("FAIL %-28s len %d vs %d; stats %s vs %s%s" % (tag, len(s), len(s_o), st.tolist(), st_o.tolist(), note))
    m = min(len(s), len(s_o))
    d = np.nonzero(s[:m] != s_o[:m])[0]
    if d.size:
        first = int(d[0])
        fr = frames_of(s_o)
        k = max(i for i, (off, _) in enumerate(fr) if off <= first)
        off, plen = fr[k]
        print("     first difference at byte %d = frame %d (+%d; header 20, payload %d); %d differing bytes in all; frames with differences: %s"
              % (first, k, first - off, plen, d.size, sorted(set(int(np.searchsorted([o for o, _ in fr], x, side='right') - 1) for x in d[:2000]))[:12]))
        lo = max(off, first - 8)
        print("     gpu   ", s[lo:lo + 32].tobytes().hex())
        print("     oracle", s_o[lo:lo + 32].tobytes().hex())
        print("     gpu header   ", s[off:off + 20].tobytes().hex())
        print("     oracle header", s_o[off:off + 20].tobytes().hex())

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
sizes = [1, 2, 3, 20, 21, 22, 41, 81, 82, 100, 1000, 5120, 5121, 5122, 5141, 9999, 10000, 10001, 10002, 10021, 16000, 20000, 25121,
         100000, 160001, 1000000]
if not quick:
    sizes += [2560000 + 7, 26460000]
for kind, name in ((x3hip.SYNTH_HYDROPHONE, "hydro"), (x3hip.SYNTH_ZEROS, "zeros"), (x3hip.SYNTH_WALK, "walk"),
                   (x3hip.SYNTH_SINE, "sine"), (x3hip.SYNTH_WHITE, "white")):
    for n in sizes:
        if kind != x3hip.SYNTH_HYDROPHONE and n > 1000000:
            continue
        check("%s n=%d" % (name, n), x3hip.synth(kind, 0x58330002, 0, n))
for sp in (1, 2, 3, 17, 320):
    check("hydro n=30001 start_pos=%d" % sp, x3hip.synth(x3hip.SYNTH_HYDROPHONE, 0x58330002, 0, 30001), start_pos=sp)
# mixtures: a loud stretch in quiet content (frames that do and do not fit the image in one call)
rng = np.random.default_rng(7)
w = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 0x58330002, 0, 400000).copy()
w[123456:123456 + 30000] = rng.integers(-32768, 32767, 30000, dtype=np.int16)
check("hydro + white stretch", w)
w = x3hip.synth(x3hip.SYNTH_HYDROPHONE, 0x58330002, 0, 400000).copy()
w[5000:5400] = rng.integers(-32768, 32767, 400, dtype=np.int16)
w[77777:77777 + 2000] = rng.integers(-2000, 2000, 2000, dtype=np.int16)
check("hydro + short bursts", w)
print("dense reruns %d, timeouts %d" % (ctx.get_option("encode_dense_reruns"), ctx.get_option("encode_fallbacks")))
print("FAILED: %d" % bad if bad else "ALL OK")
sys.exit(1 if bad else 0)
