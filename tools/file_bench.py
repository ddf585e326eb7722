"""Throughput of the file level (x3_wav_to_x3a / x3_x3a_to_wav) on a synthetic hydrophone recording,
beside the oracle's single-thread restatement of the reference's file functions on a bounded prefix.

    python tools/file_bench.py [--samples N] [--dir /dev/shm] [--workers 1,2,3,4] [--chunk-frames 3200]
"""
import argparse, ctypes as C, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import x3hip

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=691_200_000)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--workers", default="1,2,3,4")
ap.add_argument("--chunk-frames", default="3200")
ap.add_argument("--cpu-samples", type=int, default=100_000_000)
a = ap.parse_args()
n = a.samples
ctx = x3hip.Context(0); L = x3hip.lib()
d = ctx.alloc(2 * n); ctx.synth_dev(2, 0x58330003, 0, n, d)
wav = np.empty(n, dtype=np.int16); L.x3_dev_download(ctx._h, wav.ctypes.data, d, 2 * n); ctx.free(d)
src, x3a, back = (os.path.join(a.dir, s) for s in ("x3bench.wav", "x3bench.x3a", "x3bench_back.wav"))
hdr = b"RIFF" + struct.pack("<I", 36 + 2 * n) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 192000, 384000, 2, 16) + b"data" + struct.pack("<I", 2 * n)
with open(src, "wb") as f:
    f.write(hdr); f.write(wav.tobytes())
try:
    for cf in a.chunk_frames.split(","):
        for w in a.workers.split(","):
            ctx.set_option("file_workers", int(w)); ctx.set_option("file_chunk_frames", int(cf))
            for rep in range(2):
                t0 = time.perf_counter(); rc, stats = ctx.wav_to_x3a(src, x3a); t1 = time.perf_counter()
                rc2, ns, ferr = ctx.x3a_to_wav(x3a, back); t2 = time.perf_counter()
                assert rc == 0 and rc2 == 0 and ns == n, (rc, rc2, ns, ctx.last_error())
            print("chunk %5s frames, %s workers: wav->x3a %7.1f ms (%6.0f Msamples/s)   x3a->wav %7.1f ms (%6.0f Msamples/s)   x3a %d B" % (
                cf, w, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, (t2 - t1) * 1e3, n / (t2 - t1) / 1e6, os.path.getsize(x3a)), flush=True)
    with open(back, "rb") as f:
        assert f.read() == hdr + wav.tobytes()
    print("round trip identical; encode_fallbacks %d, dense reruns %d" % (ctx.get_option("encode_fallbacks"), ctx.get_option("encode_dense_reruns")))
    if a.cpu_samples:
        import oracle_lib as O
        m = min(n, a.cpu_samples)
        with open(src, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + 2 * m) + hdr[8:40] + struct.pack("<I", 2 * m)); f.write(wav[:m].tobytes())
        O.lib(native=True)
        Ln = O.lib(native=True)
        t0 = time.perf_counter(); rc = Ln.x3o_wav_to_x3a(os.fsencode(src), os.fsencode(x3a), None); t1 = time.perf_counter()
        ns, fe = C.c_uint64(0), C.c_uint64(0)
        rc2 = Ln.x3o_x3a_to_wav(os.fsencode(x3a), os.fsencode(back), C.byref(ns), C.byref(fe)); t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0 and ns.value == m
        print("oracle (1 thread, %d samples): wav->x3a %.0f Msamples/s   x3a->wav %.0f Msamples/s" % (m, m / (t1 - t0) / 1e6, m / (t2 - t1) / 1e6))
finally:
    for p in (src, x3a, back):
        if os.path.exists(p):
            os.remove(p)
