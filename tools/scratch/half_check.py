"""round trip through the device API with the library named by X3HIP_LIB: decode(encode(x)) == x for a few sizes"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np
import x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); L = x3hip.lib()
for n, kind in ((20_000_000, 2), (6_400_000, 4), (640_000 * 3 + 10_000 * 7, 2), (1_234_567, 1), (64 * 10_000 * 5, 0)):
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
    wav, out, off, back = ctx.alloc(2 * n), ctx.alloc(cap + 16), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n + 64)
    ctx.synth_dev(kind, 77, 0, n, wav)
    assert ctx.encode_dev(wav, n, p, out, cap, 0, off) == 0 and ctx.encode_result()[0] == 0
    for shift in (0, 8, 24):   # the output 0, 16 and 48 bytes into a line
        assert ctx.decode_dev(out, cap, off, F, p, back + 2 * shift, n, n_per_clip=n) == 0
        assert ctx.decode_result()[:3] == (0, F, 0)
        a = ctx.download(wav, 2 * n, np.int16); b = ctx.download(back + 2 * shift, 2 * n, np.int16)
        assert np.array_equal(a, b), (n, kind, shift, int(np.nonzero(a != b)[0][0]))
    for d in (wav, out, off, back):
        ctx.free(d)
print("round trips identical:", os.environ.get("X3HIP_LIB", "default library"))
