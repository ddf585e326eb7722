"""x3_encode_batch on clips of DIFFERENT lengths (host buffers): wall time for 2 000 clips of 1-5 s at 44.1 kHz.
Round 4: one launch set for all of them (x3_encode_frames_dev) instead of one per clip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0)
rng = np.random.default_rng(3)
base = x3hip.synth(2, 11, 0, 5 * 44100 + 2000)
clips = [base[o:o + n] for o, n in zip(rng.integers(0, 2000, size=2000), rng.integers(44100, 5 * 44100, size=2000))]
clips = [np.ascontiguousarray(c) for c in clips]
tot = sum(c.size for c in clips)
for _ in range(2):
    t0 = time.perf_counter()
    rc, out, offs, stats = ctx.encode_batch(clips)
    dt = time.perf_counter() - t0
    assert rc == 0
    print("2 000 clips, %.1f M samples: %.1f ms (%.2f Gsamples/s host to host), %d bytes" % (tot / 1e6, dt * 1e3, tot / dt / 1e9, offs[-1]))

# what the ragged path did until round 4: one launch set, one wait and two copies per clip
t0 = time.perf_counter()
pos = 0
for c in clips:
    rc, o, st = ctx.encode(c)
    assert rc == 0
    pos += o.size
dt = time.perf_counter() - t0
print("the same clips one x3_encode each: %.1f ms (%.2f Gsamples/s), %d bytes" % (dt * 1e3, tot / dt / 1e9, pos))
