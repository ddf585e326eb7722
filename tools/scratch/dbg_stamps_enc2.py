"""Per-wave phase times of x3_encode_stream2_kernel (build with -DX3_DBG_STAMPS, X3HIP_LIB=...libx3hip_stamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
for o in sys.argv[1:]:
    k, v = o.split("="); ctx.set_option(k, int(v))
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for _ in range(8):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    ctx.encode_result()
print("kernel ms (incl. stamp overhead):", ctx.kernel_time(0)[0] / 8, "wgs/CU", ctx.get_option("stream_wgs_in_use"))
G = 256 * ctx.get_option("stream_wgs_in_use")
fpw = F / G
NW = 400
out = np.zeros(8*4096, dtype=np.uint64)
L.x3_dbg_read_enc.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read_enc(ctx._h, out.ctypes.data, out.size)
a = out[:8*8*NW].reshape(NW, 8, 8).astype(np.float64) / fpw
sp = out[8*8*NW:8*8*NW + NW*8*2].reshape(NW, 8, 2).astype(np.float64) / fpw
print("settle(): polls per frame and wave, mean %.2f; share of frames with a size missing at first look %.2f; by wave:" % (sp[:, :, 0].mean(), sp[:, :, 1].mean()),
      sp[:, :, 1].mean(axis=0).round(2))
names = "analysis+tail,wait B1,emit,loads+settle,wait B3,resolve+copyout,crc,wait B4".split(",")
print("ticks per frame, mean over workgroups 0..399; rows = waves")
print("wave  " + "  ".join("%15s" % nm for nm in names) + "       total")
for w in range(8):
    row = a[:, w, :].mean(axis=0)
    print("%4d  " % w + "  ".join("%15.0f" % row[k] for k in range(8)) + "  %10.0f" % row.sum())
tot = a.sum(axis=2).mean(axis=1)
print("frame time per workgroup percentiles 0/25/50/75/100:", np.percentile(tot, [0, 25, 50, 75, 100]).round(0))
# who waits least (the stragglers that everybody else waits for)?
w = (a[:, :, 3] + a[:, :, 4]).min(axis=1)       # per workgroup: the wave that spent least in loads+settle+wait B3
busy = a.sum(axis=2).mean(axis=1) - (a[:, :, 3] + a[:, :, 4]).mean(axis=1)
print("loads+settle+wait-B3 per workgroup (least-waiting wave), percentiles 0/5/25/50/75/100:", np.percentile(w, [0, 5, 25, 50, 75, 100]).round(0))
print("busy time outside of it, percentiles 0/25/50/75/95/100:", np.percentile(busy, [0, 25, 50, 75, 95, 100]).round(0))
order = np.argsort(w)[:16]
print("least-waiting workgroups (blockIdx: wait, busy):", [(int(i), int(w[i]), int(busy[i])) for i in order])
for lo in range(0, NW, 100):
    print("  blockIdx %3d..%3d: wait %.0f busy %.0f" % (lo, lo + 99, w[lo:lo + 100].mean(), busy[lo:lo + 100].mean()))
