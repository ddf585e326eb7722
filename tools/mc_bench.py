"""Throughput of the multi-channel extension (x3_encode_mc / x3_decode_stream_mc, host buffers in and out) beside the
oracle's single thread, on hydrophone-like noise:   python tools/mc_bench.py [--channels 2] [--samples 69120000]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import x3hip
ap = argparse.ArgumentParser()
ap.add_argument("--channels", type=int, default=2)
ap.add_argument("--samples", type=int, default=69_120_000, help="per channel (6 min at 192 kHz)")
ap.add_argument("--cpu-samples", type=int, default=10_000_000)
a = ap.parse_args()
ctx = x3hip.Context(0)
wavs = [x3hip.synth(2, 0x58330010 + k, 0, a.samples) for k in range(a.channels)]
ap2 = None
tot = a.samples * a.channels
for threads in (0, 1):   # the lane-per-frame decoder (round 4), then the thread-per-frame one it replaced
    ctx.set_option("mc_decode_threads", threads)
    best_e = best_d = 1e9
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for rep in range(3):
        t0 = time.perf_counter(); rc, x, st = ctx.encode_mc(wavs); t1 = time.perf_counter()
        assert rc == 0, (rc, ctx.last_error())
        rc, back, fok, ferr = ctx.decode_stream_mc(x, a.channels, wav_cap=a.samples + 64); t2 = time.perf_counter()
        assert (rc, ferr) == (0, 0) and all(np.array_equal(b, w) for b, w in zip(back, wavs))
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    ms, cnt = ctx.kernel_time(1)
    ctx.enable_kernel_timing(False)
    print("mc_decode_threads=%d: decode kernels %.3f ms per call on the device (%.1f Gsamples/s device-resident), %.1f ms host to host"
          % (threads, ms / max(cnt, 1), tot / (ms / max(cnt, 1)) / 1e6, best_d * 1e3))
ctx.set_option("mc_decode_threads", 0)
print("%d channels x %d samples, %d stream bytes (%.3f B/sample): encode %.1f ms (%.0f Msamples/s), decode %.1f ms (%.0f Msamples/s), round trip bit-exact"
      % (a.channels, a.samples, x.size, x.size / tot, best_e * 1e3, tot / best_e / 1e6, best_d * 1e3, tot / best_d / 1e6))
if a.cpu_samples:
    import oracle_lib as O
    m = min(a.samples, a.cpu_samples)
    cut = [w[:m] for w in wavs]
    t0 = time.perf_counter(); rc, xo, _ = O.encode_mc(cut); t1 = time.perf_counter()
    rc2, bo, _, _ = O.decode_stream_mc(xo, a.channels, wav_cap=m + 64); t2 = time.perf_counter()
    assert rc == 0 and rc2 == 0
    rcg, xg, _ = ctx.encode_mc(cut)
    assert rcg == 0 and np.array_equal(xg, xo), "GPU stream differs from the oracle's"
    print("oracle (1 thread, %d samples per channel): encode %.0f Msamples/s, decode %.0f Msamples/s; GPU stream identical"
          % (m, m * a.channels / (t1 - t0) / 1e6, m * a.channels / (t2 - t1) / 1e6))
