#!/usr/bin/env python3
"""The decode phase's two paces against WHERE IN THEIR ALLOCATIONS the stream and the decoded samples begin -- one process,
one set of allocations (so the physical placement is held), config 3.  usage: modes_shift_sweep.py [--pad KiB] [--steps K]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ap = argparse.ArgumentParser()
ap.add_argument("--pad", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--samples", type=int, default=691_200_000)
a = ap.parse_args()
ctx = x3hip.Context(0)
p = x3hip.Params.default()
L = x3hip.lib()
n = a.samples
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
SL = 16 << 20
if a.pad:
    ctx.alloc(a.pad * 1024)
w0 = ctx.alloc(2 * n + 64); o0 = ctx.alloc(cap + 16 + SL); d_off = ctx.alloc(8 * (F + 1)); b0 = ctx.alloc(2 * n + SL)
ctx.synth_dev(2, 0x58330003, 0, n, w0)
print("wav=%x out=%x back=%x" % (w0, o0, b0))
def measure(osh, bsh):
    d_out, d_back = o0 + osh, b0 + bsh
    ctx.enable_kernel_timing(False)
    for _ in range(3):
        assert ctx.encode_dev(w0, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n, n_clips=1, clip_stride=n) == 0
    assert ctx.encode_result()[0] == 0 and ctx.decode_result()[:3] == (0, F, 0)
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(a.steps):
        assert ctx.encode_dev(w0, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n, n_clips=1, clip_stride=n) == 0
    ctx.encode_result(); ctx.decode_result()
    return [ctx.kernel_time(i)[0] / a.steps for i in (0, 1, 4)]
shifts = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23]
print("out shift   (back shift 0):")
for s in shifts:
    e, d, c = measure(s, 0)
    print("  %9d  encode %.3f decode %.3f check %.3f" % (s, e, d, c), flush=True)
print("back shift  (out shift 0):")
for s in shifts:
    e, d, c = measure(0, s)
    print("  %9d  encode %.3f decode %.3f check %.3f" % (s, e, d, c), flush=True)
print("both:")
for s in (4096, 65536, 1 << 20, 1 << 21):
    e, d, c = measure(s, s)
    print("  %9d  encode %.3f decode %.3f check %.3f" % (s, e, d, c), flush=True)
e, d, c = measure(0, 0)
print("  again 0/0  encode %.3f decode %.3f check %.3f" % (e, d, c))
