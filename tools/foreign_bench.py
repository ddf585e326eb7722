#!/usr/bin/env python3
"""x3_decode_stream_dev on config 3's stream (no frame index: the walk on the GPU, then check + decode), N calls.
Under `rocprofv3 --kernel-trace --stats` this gives the walk's kernels one by one (profiles/r4/foreign_stream_kernels.csv)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 691_200_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = x3hip.Context(0); p = x3hip.Params.default(); L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
rc, pos, st = ctx.encode_result(); assert rc == 0
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    r = ctx.decode_stream_dev(d_out, pos, p, d_back, n)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert r == (0, n, F, 0), r
ts.sort()
print("x3_decode_stream_dev: min %.3f median %.3f max %.3f ms over %d calls (%d frames, %d bytes)" % (ts[0], ts[len(ts) // 2], ts[-1], reps, F, pos))
