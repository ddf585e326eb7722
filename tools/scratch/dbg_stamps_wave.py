"""Per-wave phase times of x3_encode_wave_kernel (build with -DX3_DBG_STAMPS, X3HIP_LIB=...libx3hip_stamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
kind = 2
for o in sys.argv[1:]:
    k, v = o.split("=")
    if k == "kind": kind = int(v)
    else: ctx.set_option(k, int(v))
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1))
ctx.synth_dev(kind, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for _ in range(6):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    ctx.encode_result()
print("kernel ms (incl. stamp overhead):", ctx.kernel_time(0)[0] / 6)
out = np.zeros(8*8192, dtype=np.uint64)
L.x3_dbg_read_enc.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read_enc(ctx._h, out.ctypes.data, out.size)
a = out[:256*16*8].reshape(256, 16, 8).astype(np.float64)
fpw = F / 4096.0
names = "wait samples,analysis,scan+size+req,emission,crc,offset waits,copy-out,clear+book".split(",")
m = a.mean(axis=(0, 1)) / fpw
print("clock64 ticks per frame and wave (mean over 4096 waves):")
for k in range(8): print("  %-16s %9.0f  %5.1f %%" % (names[k], m[k], 100 * m[k] / m.sum()))
print("  total %.0f ticks per frame; per wave total min/median/max: %s" % (m.sum(), np.percentile(a.sum(axis=2), [0, 50, 100]).round(0)))
print("by wave index (total ticks per frame):", (a.sum(axis=2).mean(axis=0) / fpw).round(0))
print("offset waits by wave index:", (a[:, :, 5].mean(axis=0) / fpw).round(0))
c = out[32768:32768 + 256*16*4].reshape(256, 16, 4).astype(np.float64) / fpw
print("per frame and wave: descriptor polls %.3f, trips to the descriptors %.3f, ticks there %.0f, ticks waiting for another wave's base %.0f" % tuple(c.mean(axis=(0, 1))))
print("ticks per trip: %.0f, polls per trip %.2f" % (c[:, :, 2].sum() / max(c[:, :, 1].sum(), 1e-9), c[:, :, 0].sum() / max(c[:, :, 1].sum(), 1e-9)))
print("trips to the descriptors by wave index:", c[:, :, 1].mean(axis=0).round(3))
print("ticks there by wave index:", c[:, :, 2].mean(axis=0).round(0))
print("base-wait ticks by wave index:", c[:, :, 3].mean(axis=0).round(0))
tot = a.sum(axis=2); busy = tot - a[:, :, 5]
wg_busy = busy.mean(axis=1); wg_tot = tot.mean(axis=1); wg_wait = a[:, :, 5].mean(axis=1)
gens = np.where(np.arange(256) < (F // 16) - 16 * 256, 17, 16) if F // 16 > 16 * 256 else np.full(256, F / 4096.0)
print("per workgroup: busy ticks per frame percentiles 0/5/50/95/100:", np.percentile(wg_busy / gens, [0, 5, 50, 95, 100]).round(0))
print("per workgroup: wait ticks per frame percentiles 0/5/50/95/100:", np.percentile(wg_wait / gens, [0, 5, 50, 95, 100]).round(0))
print("busy per frame by XCD (blockIdx % 8):", np.array([(wg_busy / gens)[x::8].mean() for x in range(8)]).round(0))
print("wait per frame by XCD (blockIdx % 8):", np.array([(wg_wait / gens)[x::8].mean() for x in range(8)]).round(0))
order = np.argsort(wg_wait / gens)
print("workgroups that wait least (blockIdx, wait, busy per frame):", [(int(i), int((wg_wait / gens)[i]), int((wg_busy / gens)[i])) for i in order[:12]])
print("workgroups that wait most:", [(int(i), int((wg_wait / gens)[i]), int((wg_busy / gens)[i])) for i in order[-6:]])
