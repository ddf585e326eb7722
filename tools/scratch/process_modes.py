"""One process: the kernels of config 3's round trip (HIP-event times, 40 steps) and the shader clock the decoder logged.
Run several times in one gpurun call: does the decoder's time move from process to process on one box?  (It does not:
profiles/r4/decoder_experiments_part2.txt; between boxes it moves by 8 % at one shader clock.)"""
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
def step():
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
for _ in range(10): step()
ctx.decode_result()
ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
for _ in range(40): step()
ctx.encode_result(); ctx.decode_result()
t = [ctx.kernel_time(i)[0] / 40 for i in range(6)]
log = ctx.launch_log(1)[-40:]
mhz = sorted(e.get("clock_mhz", 0.0) for e in log)
ctx.enable_kernel_timing(False)
print("encode %.3f decode %.3f check %.3f   decode clock median %.0f MHz   out=%x back=%x" % (
    t[0], t[1], t[4], mhz[len(mhz) // 2] if mhz else 0, d_out, d_back), flush=True)
