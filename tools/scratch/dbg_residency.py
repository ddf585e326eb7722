"""How many split-decoder workgroups are resident at once?  (stamps build: X3HIP_LIB=...libx3hip_stamps.so)
Counts the groups that started before the first one finished, for a grid larger than the chip holds."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 2 * 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
print(ctx.encode_result()[0])
for _ in range(2):
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    print(ctx.decode_result())
out = np.zeros(8*4096, dtype=np.uint64)
L.x3_dbg_read.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read(ctx._h, out.ctypes.data, out.size)
a = out.reshape(-1, 8)[0::2]
start, end = a[:, 6].astype(np.int64), a[:, 7].astype(np.int64)
t0 = start.min()
first_end = end.min()
print("groups in the buffer:", len(a), " started before the first one finished:", int((start < first_end).sum()))
print("start offsets (us at 100 MHz) percentiles:", np.percentile((start - t0) / 100.0, [0, 25, 50, 60, 70, 80, 90, 100]).round(1))
print("durations (us) percentiles:", np.percentile((end - start) / 100.0, [0, 25, 50, 75, 100]).round(1))
