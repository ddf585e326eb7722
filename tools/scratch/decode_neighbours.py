"""The decoder's time against what ran in front of it: (A) config 3's step, encode then decode; (B) decode after decode;
(C) with a profiling build (-DX3_PROFILING) and X3HIP_CHECK_SERIAL=1, the check kernel in front of the decoder on one stream.
Per mode: HIP-event time of the decode kernel (mean of 40), the shader clock the kernel logged, its pace."""
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
def enc(): assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
def dec(): assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
enc(); assert ctx.encode_result()[0] == 0
def measure(name, step, k=40):
    for _ in range(12): step()
    ctx.decode_result()
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(k): step()
    ctx.decode_result()
    t = [ctx.kernel_time(i) for i in range(6)]
    log = ctx.launch_log(1)[-k:]
    ctx.enable_kernel_timing(False)
    mhz = sorted(e["clock_mhz"] for e in log); tg = sorted(e["target_ticks16"] for e in log); ac = sorted(e["achieved_ticks16"] for e in log)
    print("%-28s decode %.3f ms  check %.3f  encode %.3f   clock median %.0f MHz (min %.0f)   pace target %.3f achieved %.3f us/block" % (
        name, t[1][0] / max(1, t[1][1]), t[4][0] / max(1, t[4][1]), t[0][0] / max(1, t[0][1]), mhz[k // 2], mhz[0], tg[k // 2] / 1600.0, ac[k // 2] / 1600.0), flush=True)
def both(): enc(); dec()
measure("encode, decode, ...", both)
measure("decode, decode, ...", dec)
measure("encode, decode, ... (again)", both)
