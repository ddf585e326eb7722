"""Per-wave phase times of x3_decode_split_kernel (library built with -DX3_DBG_STAMPS, X3HIP_LIB=<that .so>):
parser and valuer rows separately, plus a histogram of group lifetimes by the number of groups that share the CU."""
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
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
print(ctx.encode_result()[0])
for _ in range(3):
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    print(ctx.decode_result())
NW = 1080
out = np.zeros(32*NW, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
print(L.x3_dbg_read(ctx._h, out.ctypes.data, out.size))
a = out.reshape(NW, 4, 8)
P = a[:, 0, :].astype(np.float64); V = a[:, 1, :].astype(np.float64)
print("PARSER  (shader clocks per wave):")
for k, nm in [(1, "service"), (2, "header"), (3, "samples"), (4, "barrier wait"), (0, "loop top")]:
    print("  %-14s mean %10.0f  p10 %10.0f  p90 %10.0f" % (nm, P[:, k].mean(), np.percentile(P[:, k], 10), np.percentile(P[:, k], 90)))
print("  total         mean %10.0f" % P[:, 0:5].sum(axis=1).mean())
for wv in (2, 3):
  Fd = a[:, wv, :].astype(np.float64)
  if Fd[:, 4].sum() > 0:
    print("WAVE %d (feeder, then flusher, if built):" % wv)
    for k, nm in [(4, "barrier wait"), (3, "svc: pos read"), (5, "svc: load wait"), (2, "svc: parks"), (1, "svc: requests / flush"), (0, "loop top")]:
        print("  %-14s mean %10.0f  p10 %10.0f  p90 %10.0f" % (nm, Fd[:, k].mean(), np.percentile(Fd[:, k], 10), np.percentile(Fd[:, k], 90)))
print("VALUER:")
for k, nm in [(4, "barrier wait"), (2, "header"), (3, "samples"), (1, "tail/bounds"), (5, "flush"), (0, "loop top")]:
    print("  %-14s mean %10.0f  p10 %10.0f  p90 %10.0f" % (nm, V[:, k].mean(), np.percentile(V[:, k], 10), np.percentile(V[:, k], 90)))
print("  total         mean %10.0f" % V[:, 0:6].sum(axis=1).mean())
# where: parser acc[5], valuer acc[6] = xcc<<32 | hw_id ; hw_id: [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se
whereP = a[:, 0, 5]; whereV = a[:, 1, 6]
def cu_of(w):
    hw = w & 0xFFFFFFFF; xcc = w >> 32
    return (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 15)
cu = cu_of(whereP)
simdP = (whereP >> 4) & 3; simdV = (whereV >> 4) & 3
uniq, cnt = np.unique(cu, return_counts=True)
print("CUs used:", len(uniq), "groups per CU histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
life = (a[:, 0, 7] - a[:, 0, 6]).astype(np.float64)   # 100 MHz ticks
per = dict(zip(uniq, cnt))
for c in sorted(set(cnt)):
    sel = np.array([per[x] == c for x in cu])
    print("  groups on CUs with %d groups: lifetime mean %.1f us (n=%d), parser total %0.f, parser barrier wait %.0f, valuer barrier wait %.0f" %
          (c, life[sel].mean() / 100.0, sel.sum(), P[sel, 0:5].sum(axis=1).mean(), P[sel, 4].mean(), V[sel, 4].mean()))
print("same SIMD for parser and valuer:", int((simdP == simdV).sum()), "of", NW)
t0 = a[:, 0, 6].min()
print("start spread (us): p50 %.1f p99 %.1f max %.1f ; end (us): p50 %.1f max %.1f" % tuple(
    x / 100.0 for x in (np.percentile(a[:, 0, 6] - t0, 50), np.percentile(a[:, 0, 6] - t0, 99), (a[:, 0, 6] - t0).max(),
                        np.percentile(a[:, 0, 7] - t0, 50), (a[:, 0, 7] - t0).max())))
# per-SIMD wave counts on each CU
key = np.concatenate([cu * 4 + simdP, cu * 4 + simdV])
u2, c2 = np.unique(key, return_counts=True)
print("waves per SIMD histogram:", dict(zip(*np.unique(c2, return_counts=True))))
# parsers per SIMD: how evenly the dispatcher spreads the critical waves of a CU's groups
u3, c3 = np.unique(cu * 4 + simdP, return_counts=True)
print("parsers per SIMD histogram (SIMDs that hold at least one):", dict(zip(*np.unique(c3, return_counts=True))))
for c in sorted(set(cnt)):
    cus = [x for x in uniq if per[x] == c]
    hh = {}
    for x in cus:
        k = tuple(sorted(int(((cu * 4 + simdP) == x * 4 + s_).sum()) for s_ in range(4)))
        hh[k] = hh.get(k, 0) + 1
    print("  CUs with %d groups: parsers on SIMD 0..3 (sorted) -> CUs:" % c, hh)
print("kernel ms/launches decode:", ctx.kernel_time(1))
# ---- who is slow?
xcc = (whereP >> 32).astype(np.int64)
se = ((whereP & 0xFFFFFFFF) >> 13) & 7
print("lifetime (us) by XCC:", {int(x): round(float(life[xcc == x].mean()) / 100, 1) for x in np.unique(xcc)})
print("lifetime (us) by SE :", {int(x): round(float(life[se == x].mean()) / 100, 1) for x in np.unique(se)})
wps = dict(zip(u2, c2))
pw = np.array([wps[k] for k in (cu * 4 + simdP)]); vw = np.array([wps[k] for k in (cu * 4 + simdV)])
for a_, b_ in [(2, 2), (2, 3), (3, 2), (3, 3)]:
    sel = (pw == a_) & (vw == b_)
    if sel.sum(): print("  parser on a SIMD with %d waves, valuer with %d: n=%4d lifetime %.1f us  parser busy %.0f  valuer busy %.0f" % (
        a_, b_, sel.sum(), life[sel].mean() / 100, (P[sel, 0:4].sum(axis=1)).mean(), (V[sel][:, [0, 1, 2, 3, 5]].sum(axis=1)).mean()))
print("lifetime percentiles (us) 0/10/50/90/99/100:", (np.percentile(life, [0, 10, 50, 90, 99, 100]) / 100).round(1))
order = np.argsort(life)[-12:]
print("slowest groups: blockIdx, xcc, se, cu-groups, parser-simd-waves, valuer-simd-waves, lifetime")
for i in order: print("   ", int(i), int(xcc[i]), int(se[i]), per[cu[i]], pw[i], vw[i], round(life[i] / 100, 1))
bi = np.arange(NW)
print("lifetime by blockIdx quartile:", [round(float(life[(bi >= q * NW // 4) & (bi < (q + 1) * NW // 4)].mean()) / 100, 1) for q in range(4)])
