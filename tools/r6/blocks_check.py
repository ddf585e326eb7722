#!/usr/bin/env python3
"""Round-trip check of the decoder kernels on a few sizes and signal kinds (identity with the input, and the
three-wave kernel's output): quick, prints the first difference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip

ctx = x3hip.Context(0)
p = x3hip.Params.default()
bad = 0
for kind, n in ((2, 10_000), (2, 16_000), (2, 1), (2, 2), (2, 21), (2, 22), (2, 41), (0, 20_001), (1, 30_000), (3, 50_000), (4, 123_457),
                (2, 640_000), (2, 3_000_017), (1, 700_000), (2, 26_460_000)):
    wav = x3hip.synth(kind, 77 + kind, 0, n)
    rc, stream, stats = ctx.encode(wav, p)
    assert rc == 0
    for three in (0, 1):
        ctx.set_option("decode_blocks", 1 - three)
        rc, back, frames_ok, frame_errors = ctx.decode_stream(stream, p, wav_cap=wav.size)
        k = ctx.get_option("decode_kernel_in_use")
        ok = rc == 0 and back.size == wav.size and np.array_equal(back, wav)
        print("kind %d n %9d three_wave %d kernel %d rc %d frames %d  %s" % (kind, n, three, k, rc, frames_ok, "ok" if ok else "DIFF"), flush=True)
        if not ok:
            bad += 1
            m = min(back.size, wav.size)
            d = np.nonzero(back[:m] != wav[:m])[0]
            print("   sizes", back.size, wav.size, "first diffs at", d[:10], "count", d.size)
            if d.size:
                i = int(d[0])
                print("   got ", back[max(0, i - 4):i + 8].tolist())
                print("   want", wav[max(0, i - 4):i + 8].tolist())
    ctx.set_option("decode_blocks", 0)
print("BAD" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
