#!/usr/bin/env python3
"""Time the GPU-side frame walk (x3_index_dev) and x3_decode_stream_dev on config 3's stream."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
d_fo = ctx.alloc(8 * (F + 8)); d_wo = ctx.alloc(8 * (F + 8))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
rc, pos, _ = ctx.encode_result(); assert rc == 0
for name, fn in (("x3_index_dev", lambda: ctx.index_dev(d_out, pos, F + 8, d_fo, d_wo)),
                 ("x3_decode_stream_dev", lambda: ctx.decode_stream_dev(d_out, pos, p, d_back, n))):
    ts = []
    for _ in range(8):
        ctx.sync(); t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, r, "ms:", " ".join("%.3f" % t for t in ts))
