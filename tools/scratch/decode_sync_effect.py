"""does the decoder run slower when every launch is followed by a trip to the host (x3_decode_result), as in
x3_decode_stream_dev, or when it is given explicit sample offsets?  kernel times from HIP events, clocks from the launch log"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
d_fo = ctx.alloc(8 * (F + 8)); d_wo = ctx.alloc(8 * (F + 8))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
rc, pos, _ = ctx.encode_result(); assert rc == 0
assert ctx.index_dev(d_out, pos, F + 8, d_fo, d_wo)[0] == 0
def run(tag, fn, steps=40):
    for _ in range(5): fn()
    ctx.decode_result()
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(steps): fn()
    ctx.decode_result()
    ts = sorted(ctx.kernel_times(1)); log = ctx.launch_log(1)[-steps:]
    ctx.enable_kernel_timing(False)
    print("%-44s decode median %.4f min %.4f ms; clock median %.0f MHz" % (tag, ts[len(ts) // 2], ts[0], sorted(e["clock_mhz"] for e in log)[len(log) // 2]))
def plain(): assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
def synced(): plain(); ctx.decode_result()
def stream_dev(): assert ctx.decode_stream_dev(d_out, pos, p, d_back, n)[0] == 0
for rep in range(2):
    run("back to back (the bench's step)", plain)
    run("a host trip behind every launch", synced)
    run("x3_decode_stream_dev", stream_dev)
