"""config 2's round trip (encode with the segment index + decode by stretches) call by call and as a replayed HIP graph:
host wall time per step over back-to-back steps between two synchronisations"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); L = x3hip.lib()
for n in (26_460_000, 5_000_000, 100_000_000):
    sb = 32
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p)); ne = L.x3_seg_index_entries(F, C.byref(p), sb)
    d_wav = ctx.alloc(2*n + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n); d_seg = ctx.alloc(8*ne)
    ctx.synth_dev(2, 0x58330002, 0, n, d_wav)
    def calls():
        assert ctx.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, sb, 0, d_off) == 0
        assert ctx.decode_dev_seg(d_out, cap, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n) == 0
    calls(); assert ctx.encode_result()[0] == 0 and ctx.decode_result() == (0, F, 0, n)
    ctx.graph_begin(); calls(); g = ctx.graph_end()
    def timed(fn, k=200):
        for _ in range(20): fn()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(k): fn()
        ctx.sync(); return (time.perf_counter() - t0) / k * 1e3
    a = timed(calls); b = timed(lambda: ctx.graph_launch(g)); a2 = timed(calls); b2 = timed(lambda: ctx.graph_launch(g))
    assert ctx.encode_result()[0] == 0 and ctx.decode_result() == (0, F, 0, n)
    print("n=%d (%d frames): calls %.4f / %.4f ms per step, graph %.4f / %.4f ms  -> %.0f vs %.0f Gsamples/s" % (n, F, a, a2, b, b2, n / min(a, a2) / 1e6, n / min(b, b2) / 1e6))
    ctx.graph_destroy(g)
    for d in (d_wav, d_out, d_off, d_back, d_seg): ctx.free(d)
