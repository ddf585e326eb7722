import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for _ in range(3):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    print(ctx.encode_result()[0])
out = np.zeros(8*1024, dtype=np.uint64)
L.x3_dbg_read_enc.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
print(L.x3_dbg_read_enc(ctx._h, out.ctypes.data, out.size))
a = out.reshape(-1, 8).astype(np.float64)
names="analyze|geom,scan,barrier waits,zero|publish+dma,emit|hdrcrc,crc|resolve,copy-out|request,-".split(",")
for who,sl in (("compute wave0",a[0::2]),("helper wave8",a[1::2])):
    sl=sl[sl.sum(axis=1)>0]
    print(who, "WGs", len(sl), "total", sl.sum(axis=1).mean(), "max", sl.sum(axis=1).max())
    for k in range(8): print("   %-18s mean %10.0f" % (names[k], sl[:,k].mean()))
print("kernel ms/launches (encode):", ctx.kernel_time(0))
h = a[1::2]; c = a[0::2]
lb = h[:, 5]
print("helper resolve per WG percentiles 0/5/25/50/75/95/100:", np.percentile(lb, [0, 5, 25, 50, 75, 95, 100]).round(0))
work = c[:, 0] + c[:, 1] + c[:, 3] + c[:, 4] + c[:, 5] + c[:, 6]
print("compute wave0 non-barrier time percentiles:", np.percentile(work, [0, 5, 25, 50, 75, 95, 100]).round(0))
raw = out.reshape(-1, 8)[1::2, 7]
print("helper: polls per frame %.2f, windows with a missing size per frame %.2f, missing sizes per frame %.1f" % (
    (raw & 0xFFFFFFFF).mean() / 135.0, ((raw >> 32) & 0xFFFF).mean() / 135.0, (raw >> 48).mean() / 135.0))
print("helper per frame: windows examined %.0f, reduce %.0f, bookkeeping(slot5) %.0f" % (h[:, 1].mean() / 135.0, h[:,0].mean()/135.0, h[:,5].mean()/135.0))
polls = (raw & 0xFFFFFFFF).astype(np.float64) / 135.0
print("polls per frame by WG percentiles 0/5/25/50/75/95/100:", np.percentile(polls, [0, 5, 25, 50, 75, 95, 100]).round(2))
print("polls per frame, mean over blockIdx ranges of 64:", [round(float(polls[i:i+64].mean()), 2) for i in range(0, 512, 64)])
print("resolve cycles per frame, mean over blockIdx ranges of 64:", [int(lb[i:i+64].mean() / 135) for i in range(0, 512, 64)])
tot = h.sum(axis=1) - h[:, 7]
print("helper total cycles percentiles:", np.percentile(h[:, :7].sum(axis=1), [0, 50, 100]).round(0))
