"""Time the host-buffer entry points (x3_encode / x3_decode_stream) on caller-owned pageable memory.

The buffers are allocated and touched before the timed calls (a caller that streams audio owns them
already); the first call pays the context's device allocations, the second is the steady state."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np
import x3hip

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=691_200_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--chunk-frames", type=int, default=0, help="option host_chunk_frames: 0 = default chunks, -1 = one piece")
ap.add_argument("--kernel-times", action="store_true", help="HIP-event times of the kernels inside each call")
ap.add_argument("--membind", type=int, default=-1, help="bind this process's memory to one NUMA node (set_mempolicy)")
a = ap.parse_args()
if a.membind >= 0:
    mask = C.c_ulong(1 << a.membind)
    rc = C.CDLL(None, use_errno=True).syscall(238, 2, C.byref(mask), 64)   # set_mempolicy(MPOL_BIND, ...)
    print("set_mempolicy(MPOL_BIND, node %d) -> %d" % (a.membind, rc))
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
L = x3hip.lib(); ctx = x3hip.Context(0); ctx.set_option("host_chunk_frames", a.chunk_frames); p = x3hip.Params.default(); n = a.samples
d = ctx.alloc(2 * n)
ctx.synth_dev(2, 0x58330003, 0, n, d)
wav = np.empty(n, dtype=np.int16)
L.x3_dev_download(ctx._h, wav.ctypes.data, d, 2 * n)
ctx.free(d)
cap = L.x3_encode_bound(n, C.byref(p))
out = np.zeros(cap, dtype=np.uint8); out[::4096] = 1
back = np.zeros(n, dtype=np.int16); back[::2048] = 1
pos = C.c_uint64(0); stats = np.zeros(6, dtype=np.uint64)
nn, fok, ferr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
if a.kernel_times:
    ctx.enable_kernel_timing(True)
for r in range(a.reps):
    t0 = time.perf_counter()
    rc = L.x3_encode(ctx._h, wav.ctypes.data, n, 1, C.byref(p), out.ctypes.data, cap, 0, C.byref(pos), stats.ctypes.data)
    t1 = time.perf_counter()
    rc2 = L.x3_decode_stream(ctx._h, out.ctypes.data, pos.value, C.byref(p), back.ctypes.data, n, C.byref(nn), C.byref(fok), C.byref(ferr))
    t2 = time.perf_counter()
    assert rc == 0 and rc2 == 0 and nn.value == n, (rc, rc2, nn.value)
    if a.kernel_times:
        kt = [ctx.kernel_time(i) for i in (0, 1, 4)]
        print("   kernels: " + "  ".join("%s %.3f ms in %d launches" % (nm, ms, cnt) for nm, (ms, cnt) in zip(("encode", "decode", "check"), kt)))
        ctx.reset_kernel_time()
    print("call %d: encode %.1f ms (%.0f Msamples/s, %.1f GB/s in+out)  decode %.1f ms (%.0f Msamples/s)  stream %d B" % (
        r, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, (2 * n + pos.value) / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n / (t2 - t1) / 1e6, pos.value), flush=True)
def numa_of(arr):
    """pages per NUMA node of the mapping that holds arr (from /proc/self/numa_maps)"""
    try:
        addr = arr.ctypes.data
        best = None
        for line in open("/proc/self/numa_maps"):
            f = line.split()
            a0 = int(f[0], 16)
            if a0 <= addr and (best is None or a0 > best[0]):
                best = (a0, " ".join(x for x in f[1:] if x[0] == "N" or x.startswith("kernelpagesize")))
        return best[1] if best else "?"
    except Exception as e:  # noqa
        return "? (%s)" % e
print("cpu %d; pages by NUMA node: wav {%s} out {%s} back {%s}" % (C.CDLL(None).sched_getcpu(), numa_of(wav), numa_of(out), numa_of(back)))
assert np.array_equal(back, wav)
print("round trip identical")
