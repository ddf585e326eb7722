"""x3_decode_stream_dev (GPU frame walk + check + decode) a few times, for a kernel timeline (rocprofv3 --kernel-trace)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
wav, out, off, back = ctx.alloc(2 * n), ctx.alloc(cap + 16), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, wav)
assert ctx.encode_dev(wav, n, p, out, cap, 0, off) == 0
rc, pos, _ = ctx.encode_result()
assert rc == 0
for i in range(8):
    t0 = time.perf_counter()
    r = ctx.decode_stream_dev(out, pos, p, back, n)
    dt = time.perf_counter() - t0
    assert r == (0, n, F, 0), r
    print("call %d: %.3f ms" % (i, dt * 1e3))
# the decoder kernel's own time in the two entry points, same buffers, alternating
ctx.enable_kernel_timing(True)
for rep in range(3):
    ctx.reset_kernel_time()
    for i in range(10):
        assert ctx.decode_dev(out, cap, off, F, p, back, n, n_per_clip=n) == 0
        ctx.decode_result()
    a = ctx.kernel_time(1)
    ctx.reset_kernel_time()
    for i in range(10):
        assert ctx.decode_stream_dev(out, pos, p, back, n) == (0, n, F, 0)
    b = ctx.kernel_time(1)
    print("decoder kernel: decode_dev %.4f ms, decode_stream_dev %.4f ms" % (a[0] / a[1], b[0] / b[1]))
