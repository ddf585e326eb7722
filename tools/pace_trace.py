"""How the pace words settle: one line per step (tools; X3HIP_LIB to test a build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
if os.environ.get("X3HIP_LIB"): x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1)); d_back = ctx.alloc(2*n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
ctx.enable_kernel_timing(True)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    ctx.reset_kernel_time()
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    ctx.encode_result(); ctx.decode_result()
    print("step %2d: encode %.3f ms (pace %d ticks/frame)  decode %.3f ms (pace %d ticks/16 blocks)  check %.3f" % (
        i, ctx.kernel_time(0)[0], ctx.get_option("encode_pace"), ctx.kernel_time(1)[0], ctx.get_option("decode_pace"), ctx.kernel_time(4)[0]))
