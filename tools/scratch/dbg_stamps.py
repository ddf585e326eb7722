import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for _ in range(2):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    print(ctx.encode_result()[0], ctx.decode_result())
NW = int(os.environ.get('X3_STAMP_WGS','1080'))
out = np.zeros(8*NW, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
print(L.x3_dbg_read(ctx._h, out.ctypes.data, out.size))
a = out.reshape(-1, 8).astype(np.float64)
names = os.environ.get("X3_STAMP_NAMES","setup,blockhdr-pre,service,header,samples,flush,tail,-").split(",")
print("mean cycles per wave (x100MHz clock64 ticks?)")
for k in range(8): print("%-14s mean %12.0f  min %12.0f max %12.0f" % (names[k], a[:,k].mean(), a[:,k].min(), a[:,k].max()))
print("total", a.sum(axis=1).mean())
tot = a.sum(axis=1)
print("total percentiles 0/10/50/90/95/99/100:", np.percentile(tot, [0, 10, 50, 90, 95, 99, 100]).round(0))
print("waves with total > 1.15 x median:", int((tot > 1.15 * np.median(tot)).sum()), "of", len(tot))
print("kernel ms/launches (encode, decode, check):", ctx.kernel_time(0), ctx.kernel_time(1), ctx.kernel_time(4))
