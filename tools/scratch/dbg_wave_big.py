#!/usr/bin/env python3
"""The wave encoder on config-3-sized input, launch by launch, with the size-wait diagnosis (X3HIP_VERBOSE=1)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 691_200_000
ctx = x3hip.Context(0)
ctx.set_option("verbose", 1)
for o in sys.argv[2:]:
    k, v = o.split("="); ctx.set_option(k, int(v))
p = x3hip.Params.default(); L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for i in range(8):
    ctx.reset_kernel_time()
    t0 = time.perf_counter()
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    rc, pos, st = ctx.encode_result()
    t1 = time.perf_counter()
    print("launch %d: rc %d pos %d wall %.2f ms kernel %.3f ms fallbacks %d dense %d" % (
        i, rc, pos, (t1 - t0) * 1e3, ctx.kernel_time(0)[0], ctx.get_option("encode_fallbacks"), ctx.get_option("encode_dense_reruns")), flush=True)
