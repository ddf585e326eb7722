"""Per-wave phase times of x3_decode_blocks_kernel (library built with -DX3_DBG_STAMPS, X3HIP_LIB=<that .so>)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); ctx.set_option("decode_blocks", 1); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
print(ctx.encode_result()[0])
for _ in range(3):
    ctx.reset_kernel_time()
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    print(ctx.decode_result(), "decode ms", ctx.kernel_time(1)[0])
NW = 1280
out = np.zeros(32*NW, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
print(L.x3_dbg_read(ctx._h, out.ctypes.data, out.size))
a = out.reshape(NW, 4, 8)
W = a[:, 0, :].astype(np.float64)
print("WALKER (shader clocks per wave, whole launch):")
for k, nm in [(0, "loop/desc"), (1, "service"), (2, "header"), (3, "pairs"), (5, "record"), (4, "barrier wait")]:
    print("  %-14s mean %10.0f  p10 %10.0f  p90 %10.0f" % (nm, W[:, k].mean(), np.percentile(W[:, k], 10), np.percentile(W[:, k], 90)))
print("  total          mean %10.0f" % W[:, 0:6].sum(axis=1).mean())
for wv in (1, 2, 3):
    D = a[:, wv, :].astype(np.float64)
    print("DECODER %d:" % wv)
    for k, nm in [(0, "rec/staging"), (1, "init/header"), (2, "pairs"), (3, "scan/adjust"), (5, "flush"), (4, "barrier wait")]:
        print("  %-14s mean %10.0f  p10 %10.0f  p90 %10.0f" % (nm, D[:, k].mean(), np.percentile(D[:, k], 10), np.percentile(D[:, k], 90)))
    print("  total          mean %10.0f" % D[:, 0:6].sum(axis=1).mean())
life = (a[:, 0, 7] - a[:, 0, 6]).astype(np.float64)
t0 = a[:, 0, 6].min()
print("group lifetime (us): mean %.1f p10 %.1f p90 %.1f max %.1f ; start spread max %.1f us ; end max %.1f us" % (
    life.mean() / 100, np.percentile(life, 10) / 100, np.percentile(life, 90) / 100, life.max() / 100,
    (a[:, 0, 6] - t0).max() / 100.0, (a[:, 0, 7] - t0).max() / 100.0))
