"""Where do the parser and valuer waves of the split decoder land (XCD, SE, CU, SIMD), and does a group's
duration depend on what shares its SIMDs?  (stamps build: X3HIP_LIB=...libx3hip_stamps.so)"""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
ctx.encode_result()
for _ in range(2):
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    ctx.decode_result()
out = np.zeros(8*4096, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read(ctx._h, out.ctypes.data, out.size)
a = out.reshape(-1, 8)[: 2 * 1080]
P, V = a[0::2], a[1::2]
def where(w):
    hw, xcc = int(w) & 0xFFFFFFFF, int(w) >> 32
    return (xcc & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF, (hw >> 4) & 3)   # xcc, se, sh, cu, simd
dur = (P[:, 7].astype(np.int64) - P[:, 6].astype(np.int64)) / 100.0
pw = [where(x) for x in P[:, 5]]; vw = [where(x) for x in V[:, 6]]
print("first groups (xcc, se, sh, cu, simd) parser / valuer:", list(zip(pw[:10], vw[:10])))
cus = collections.Counter(w[:4] for w in pw)
print("distinct CUs:", len(cus), " groups per CU histogram:", sorted(collections.Counter(cus.values()).items()))
simd_p = collections.Counter(pw); simd_v = collections.Counter(vw)
print("parser SIMD ids:", sorted(collections.Counter(w[4] for w in pw).items()), " valuer SIMD ids:", sorted(collections.Counter(w[4] for w in vw).items()))
rows = collections.defaultdict(list)
for g in range(1080):
    key = (cus[pw[g][:4]], simd_p[pw[g]], simd_v[pw[g]], simd_p[vw[g]], simd_v[vw[g]])
    rows[key].append(dur[g])
print("groups/CU, parsers on P's SIMD, valuers on P's SIMD, parsers on V's SIMD, valuers on V's SIMD -> n, mean us, max us")
for k in sorted(rows):
    print("  ", k, len(rows[k]), round(float(np.mean(rows[k])), 1), round(float(np.max(rows[k])), 1))
print("duration percentiles:", np.percentile(dur, [0, 10, 50, 90, 100]).round(1))
