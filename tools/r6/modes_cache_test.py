#!/usr/bin/env python3
"""Does the pair effect (modes_two_buffers.py) live in what the ENCODER leaves in the caches?  One process, 3 x 3 pairs, each
timed three ways: (a) encode + decode per step, as the bench does; (b) the timed steps decode only; (c) encode, then a kernel
that writes 1 GB elsewhere (whatever the caches held of the stream is gone), then decode."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--n", type=int, default=3)
a = ap.parse_args()
ctx = x3hip.Context(0)
p = x3hip.Params.default()
L = x3hip.lib()
n = 691_200_000
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
w0 = ctx.alloc(2 * n + 64); d_off = ctx.alloc(8 * (F + 1)); junk = ctx.alloc(1 << 30)
ctx.synth_dev(2, 0x58330003, 0, n, w0)
outs, backs = [], []
for i in range(a.n):
    outs.append(ctx.alloc(cap + 16)); backs.append(ctx.alloc(2 * n)); ctx.alloc((i + 1) * 1237 * 1024)
def dec(o, b): assert ctx.decode_dev(o, cap, d_off, F, p, b, n, n_per_clip=n, n_clips=1, clip_stride=n) == 0
def enc(o): assert ctx.encode_dev(w0, n, p, o, cap, 0, d_off) == 0
def measure(o, b, mode):
    def step():
        if mode != "decode-only": enc(o)
        if mode == "flushed": ctx.synth_dev(0, 1, 0, 1 << 29, junk)
        dec(o, b)
    ctx.enable_kernel_timing(False)
    enc(o)
    for _ in range(4): step()
    ctx.encode_result(); assert ctx.decode_result()[:3] == (0, F, 0)
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(a.steps): step()
    if mode != "decode-only": ctx.encode_result()
    ctx.decode_result()
    return ctx.kernel_time(1)[0] / a.steps, ctx.kernel_time(4)[0] / a.steps
for mode in ("round-trip", "decode-only", "flushed", "round-trip"):
    print(mode)
    for i, o in enumerate(outs):
        print("  out[%d]:" % i, "  ".join("back[%d] %.3f/%.3f" % ((j,) + measure(o, b, mode)) for j, b in enumerate(backs)), flush=True)
