import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
print(ctx.encode_result()[0])
for _ in range(2):
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    print(ctx.decode_result())
out = np.zeros(8*2*1080, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
print(L.x3_dbg_read(ctx._h, out.ctypes.data, out.size))
a = out.reshape(-1, 8).astype(np.float64)
for who, sl, names in (("parser", a[0::2], "loophead,service,header,pairs,barrier wait,-,-,-"), ("valuer", a[1::2], "tail of block,block end+complete,params,pairs,barrier wait,flush,-,-")):
    print(who, "total mean %.0f  p50 %.0f  max %.0f" % (sl.sum(axis=1).mean(), np.median(sl.sum(axis=1)), sl.sum(axis=1).max()))
    for k, nm in enumerate(names.split(",")):
        if nm != "-": print("   %-22s mean %10.0f   (per block %.0f)" % (nm, sl[:, k].mean(), sl[:, k].mean() / 500.0))
print("kernel ms (decode, check):", ctx.kernel_time(1), ctx.kernel_time(4))
tot = a[0::2].sum(axis=1)
print("parser total percentiles 0/10/50/90/95/99/100:", np.percentile(tot, [0, 10, 50, 90, 95, 99, 100]).round(0))
print("WGs with total > 1.1 x median:", int((tot > 1.1 * np.median(tot)).sum()), "of", len(tot))
