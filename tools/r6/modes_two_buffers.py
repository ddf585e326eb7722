#!/usr/bin/env python3
"""Is the decode phase's pace a property of the PROCESS or of WHERE THE BUFFERS LIE?  One process, several stream buffers and
several output buffers (allocated with odd-sized allocations between them, so that they lie differently), every combination
timed, twice.  usage: modes_two_buffers.py [--steps K]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--n", type=int, default=4)
a = ap.parse_args()
ctx = x3hip.Context(0)
p = x3hip.Params.default()
L = x3hip.lib()
n = 691_200_000
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
w0 = ctx.alloc(2 * n + 64); d_off = ctx.alloc(8 * (F + 1))
ctx.synth_dev(2, 0x58330003, 0, n, w0)
outs, backs = [], []
for i in range(a.n):
    outs.append(ctx.alloc(cap + 16))
    backs.append(ctx.alloc(2 * n))
    ctx.alloc((i + 1) * 1237 * 1024)
print("wav=%x" % w0, "outs", ["%x" % o for o in outs], "backs", ["%x" % b for b in backs])
def measure(d_out, d_back):
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
for rnd in range(2):
    for i, o in enumerate(outs):
        print("round %d out[%d]:" % (rnd, i), "  ".join("back[%d] %.3f/%.3f" % ((j,) + tuple(measure(o, b)[1:])) for j, b in enumerate(backs)), flush=True)
