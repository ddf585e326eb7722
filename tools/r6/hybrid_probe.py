#!/usr/bin/env python3
"""Probe: the two decoder kernels side by side on disjoint frame ranges of config 3 (two contexts = two streams).
frames [0, F1) -> the three-wave kernel, [F1, F) -> the block-per-lane kernel; wall time per decode of the whole stream."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
A = x3hip.Context(0); B = x3hip.Context(0)
p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p)); spf = 10000
d_wav = A.alloc(2*n); d_out = A.alloc(cap+16); d_off = A.alloc(8*(F+1)); d_back = A.alloc(2*n)
A.synth_dev(2, 0x58330003, 0, n, d_wav)
assert A.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
print(A.encode_result()[0])
A.set_option("decode_blocks", 0)
def run(F1, steps=20, three_b=0):
    B.set_option("decode_blocks", 1 - three_b)
    def step():
        if F1 > 0:
            assert A.decode_dev(d_out, cap, d_off, F1, p, d_back, F1 * spf, n_per_clip=F1 * spf) == 0
        if F1 < F:
            assert B.decode_dev(d_out, cap, d_off + 8 * F1, F - F1, p, d_back + 2 * F1 * spf, (F - F1) * spf, n_per_clip=(F - F1) * spf) == 0
        ra = A.decode_result() if F1 > 0 else None
        rb = B.decode_result() if F1 < F else None
        return ra, rb
    for _ in range(5): r = step()
    t0 = time.perf_counter()
    for _ in range(steps): r = step()
    t1 = time.perf_counter()
    return (t1 - t0) / steps * 1e3, r
for F1 in (F, 0, 768 * 64, 896 * 64, 1024 * 64, 640 * 64, 512 * 64):
    for three_b in ((0, 1) if 0 < F1 < F else (0,)):
        ms, r = run(F1, three_b=three_b)
        print("old kernel frames %6d (%4d groups) | other context (%s) frames %6d : %.3f ms per whole decode (host wall, incl. sync)  %s" % (
            F1, F1 // 64, "three-wave" if three_b else "blocks", F - F1, ms, (r[0] or (0,))[0:1] + (r[1] or (0,))[0:1]))
back = np.zeros(1, dtype=np.int16)
