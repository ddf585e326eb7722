"""the decoder kernel's time behind the encoder, with and without the frame walk between them (bench.py's two loops)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
wav, out, off, back = ctx.alloc(2 * n), ctx.alloc(cap + 16), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, wav)
assert ctx.encode_dev(wav, n, p, out, cap, 0, off) == 0
rc, pos, _ = ctx.encode_result(); assert rc == 0
ctx.enable_kernel_timing(True)
for rep in range(3):
    res = []
    for walk in (0, 1):
        ctx.reset_kernel_time()
        for i in range(12):
            assert ctx.encode_dev(wav, n, p, out, cap, 0, off) == 0
            if walk:
                assert ctx.decode_stream_dev(out, pos, p, back, n) == (0, n, F, 0)
            else:
                assert ctx.decode_dev(out, cap, off, F, p, back, n, n_per_clip=n) == 0
                ctx.decode_result()
        ctx.encode_result()
        res.append([ctx.kernel_time(i)[0] / max(1, ctx.kernel_time(i)[1]) for i in (0, 1, 4)])
    print("encode+decode_dev: enc %.4f dec %.4f chk %.4f | encode+decode_stream_dev: enc %.4f dec %.4f chk %.4f" % tuple(res[0] + res[1]))
