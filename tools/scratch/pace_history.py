"""the decoder's pace controller against launch history: config 3's decode launches right behind launches of another size
(a window of 4 096 frames), per launch the HIP-event time and the pace the launch aimed at"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
assert ctx.encode_result()[0] == 0
def big(): assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
def small(): assert ctx.decode_dev(d_out, cap, d_off, 4096, p, d_back, 40_960_000, n_per_clip=40_960_000) == 0
for _ in range(30): big()
ctx.decode_result()
for rnd in range(2):
    for _ in range(6): small()
    ctx.decode_result()
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(12): big()
    ctx.decode_result()
    ts = ctx.kernel_times(1); log = ctx.launch_log(1)[-12:]
    ctx.enable_kernel_timing(False)
    print("config 3's launches behind six launches of 4 096 frames: ms    ", " ".join("%.3f" % t for t in ts))
    print("                                                     target us/blk", " ".join("%.3f" % (e["target_ticks16"] / 1600.0) for e in log))
