python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
python3 - <<'PY'
import sys, os, time, ctypes as C
sys.path.insert(0, "x3-rust_amd")
import numpy as np, x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); L = x3hip.lib()
for n in (691_200_000, 26_460_000):
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
    d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
    ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    rc, pos, _ = ctx.encode_result(); assert rc == 0
    for two in (1, 0, 1, 0):
        ctx.set_option("two_trips", two)
        ts = []
        for i in range(25):
            ctx.sync(); t0 = time.perf_counter()
            r = ctx.decode_stream_dev(d_out, pos, p, d_back, n)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert r == (0, n, F, 0), r
        ts = sorted(ts[5:])
        print("n=%d two_trips=%d: median %.3f ms  min %.3f   one-trip calls so far %d" % (n, two, ts[len(ts)//2], ts[0], ctx.get_option("stream_one_trip")))
    for d in (d_wav, d_out, d_off, d_back): ctx.free(d)
PY
