"""decoder / encoder time against the placement of their buffers (offsets inside one big allocation)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
SL = 64 << 20
big_w = ctx.alloc(2 * n + SL); big_o = ctx.alloc(cap + 16 + SL); big_b = ctx.alloc(2 * n + SL); d_off = ctx.alloc(8 * (F + 1))
def run(ow, oo, ob, steps=25):
    d_wav, d_out, d_back = big_w + ow, big_o + oo, big_b + ob
    ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
    ctx.enable_kernel_timing(False)
    for _ in range(12):
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    ctx.encode_result(); ctx.decode_result()
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(steps):
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    ctx.encode_result(); r = ctx.decode_result(); assert r[:3] == (0, F, 0)
    return [ctx.kernel_time(i)[0] / steps for i in (0, 1, 4)]
print("bases: wav %x out %x back %x" % (big_w, big_o, big_b))
offs = [0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20]
for ob in offs:
    e, d, c = run(0, 0, ob)
    print("back +%9d: encode %.3f decode %.3f check %.3f" % (ob, e, d, c), flush=True)
for oo in offs[1:]:
    e, d, c = run(0, oo, 0)
    print("out  +%9d: encode %.3f decode %.3f check %.3f" % (oo, e, d, c), flush=True)
for ow in offs[1:8]:
    e, d, c = run(ow, 0, 0)
    print("wav  +%9d: encode %.3f decode %.3f check %.3f" % (ow, e, d, c), flush=True)
