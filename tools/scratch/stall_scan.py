"""does a long series of small decode calls on one context stall now and then?  (host time per call, outliers by index)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 33_550_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
wav, out, off, back = ctx.alloc(2 * n), ctx.alloc(cap + 16), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, wav)
assert ctx.encode_dev(wav, n, p, out, cap, 0, off) == 0
ctx.encode_result()
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    t0 = time.perf_counter()
    assert ctx.decode_dev(out, cap, off, F, p, back, n, n_per_clip=n) == 0
    r = ctx.decode_result()
    ts.append(time.perf_counter() - t0)
    assert r[:3] == (0, F, 0)
ts_ms = [t * 1e3 for t in ts]
med = sorted(ts_ms)[len(ts_ms) // 2]
print("median %.3f ms; calls above 2x median: %s" % (med, [(i, round(t, 2)) for i, t in enumerate(ts_ms) if t > 2 * med]))
