"""A fresh context's first launches (no host trips between the steps: as bench.py issues them), then the launch log: decode
time per launch is not available without trips, so the log's life of group 0 (us), the clock and target / achieved pace."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
if os.environ.get("X3HIP_LIB"): x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
ctx0 = x3hip.Context(0)
d_wav = ctx0.alloc(2*n); d_out = ctx0.alloc(cap+16); d_off = ctx0.alloc(8*(F+1)); d_back = ctx0.alloc(2*n)
ctx0.synth_dev(2, 0x58330003, 0, n, d_wav); ctx0.sync()
import time
for rep in range(3):
    time.sleep(1.0)     # (the GPU's clock drops while the host sleeps)
    ctx = x3hip.Context(0)
    ctx.enable_kernel_timing(True)
    K = 24
    for i in range(K):
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
    assert ctx.decode_result()[:3] == (0, F, 0)
    ts = ctx.kernel_times(1)
    lg = ctx.launch_log(1)[-K:]
    print("fresh context %d: decode ms per launch: %s" % (rep, " ".join("%.3f" % t for t in ts)))
    print("   clock MHz: %s" % " ".join("%d" % e["clock_mhz"] for e in lg))
    print("   target/achieved us per block: %s" % " ".join("%.3f/%.3f" % (e["target_ticks16"] / 1600.0, e["achieved_ticks16"] / 1600.0) for e in lg))
    ctx.close()
