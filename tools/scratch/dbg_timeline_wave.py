"""Timeline of a few workgroups of x3_encode_wave_kernel (stamps build): when sizes go out, when offsets are waited for."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
for _ in range(6):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    ctx.encode_result()
out = np.zeros(8*8192, dtype=np.uint64)
L.x3_dbg_read_enc.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read_enc(ctx._h, out.ctypes.data, out.size)
t = out[49152:49152 + 5*16*20*4].reshape(5, 16, 20, 4).astype(np.float64)
t0 = t[:, :, 0, 0][t[:, :, 0, 0] > 0].min()
names = ["wg 0", "wg 64", "wg 128", "wg 192", "wg 255"]
print("times in us from the first size of generation 0; per generation: size out (first .. last wave) | offset wait begins (first .. last) | ends (first .. last) | polls")
for g in range(17):
    for k in range(5):
        f1 = (t[k, :, g, 0] - t0) / 100.0; b = (t[k, :, g, 1] - t0) / 100.0; e = (t[k, :, g, 2] - t0) / 100.0
        ok = t[k, :, g, 0] > 0
        if not ok.any(): continue
        print("gen %2d %-6s size %7.1f .. %7.1f | wait from %7.1f .. %7.1f | to %7.1f .. %7.1f | waited mean %5.1f max %5.1f | polls %s" % (
            g, names[k], f1[ok].min(), f1[ok].max(), b[ok].min(), b[ok].max(), e[ok].min(), e[ok].max(), (e - b)[ok].mean(), (e - b)[ok].max(),
            t[k, :, g, 3][ok].astype(int).tolist()))
