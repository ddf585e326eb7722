#!/usr/bin/env python3
"""Kernel micro-bench: time the individual kernels of the round trip on config-3-like data.
usage: kbench.py [--samples N] [--steps K] [--kind 2]     (X3HIP_LIB=/path/to/variant.so to test a build)"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=691_200_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--kind", type=int, default=2)
ap.add_argument("--clips", type=int, default=1, help="config-5 style batch: CLIPS clips of SAMPLES/CLIPS samples")
ap.add_argument("--pad", type=int, default=0, help="allocate PAD KiB first (shifts the addresses of everything behind it)")
ap.add_argument("--repeat", type=int, default=1, help="re-allocate and re-measure REPEAT times in one process")
ap.add_argument("--loud", type=float, default=0.0, help="this share of the frames is full-scale noise (frames that do not fit the wave encoder's image)")
ap.add_argument("--stride", type=int, default=0, help="--clips: samples from one clip's start to the next (default: the clip length)")
ap.add_argument("--shift", type=int, default=0, help="the samples begin this many bytes into a 16-byte unit")
ap.add_argument("--bpf", type=int, default=500, help="blocks per frame")
ap.add_argument("--bl", type=int, default=20, help="block length (samples)")
ap.add_argument("--opt", action="append", default=[], help="context option name=value (x3_ctx_set_option), repeatable")
ap.add_argument("--order", default="wofb", help="the order the buffers are allocated in: w(av) o(ut) f(rame offsets) b(ack)")
ap.add_argument("--out-shift", type=int, default=0, help="the stream begins this many bytes into its allocation")
ap.add_argument("--back-shift", type=int, default=0, help="the decoded samples begin this many bytes into their allocation")
ap.add_argument("--place", type=int, default=1, help="candidates per buffer for x3hip.place_buffers (the best pair of stream / sample buffers is used; 1 = the first allocation)")
ap.add_argument("--decode-only", action="store_true", help="the timed steps decode only (the stream of the first encode)")
ap.add_argument("--seg", type=int, default=0, help="decode by a segment index of SEG blocks per stretch (recorded by the first decode)")
a = ap.parse_args()
ctx = x3hip.Context(0)
for o in a.opt:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
p = x3hip.Params.make(a.bl, a.bpf)
n = a.samples
L = x3hip.lib()
npc = n // a.clips
stride = a.stride or npc
n = npc * a.clips
F = L.x3_num_frames(npc, C.byref(p)) * a.clips; cap = L.x3_encode_bound(npc, C.byref(p)) * a.clips
def run_once(tag):
    bufs = {}
    for ch in a.order:
        bufs[ch] = {"w": lambda: ctx.alloc(2 * stride * a.clips + 64) + a.shift, "o": lambda: ctx.alloc(cap + 16 + a.out_shift) + a.out_shift,
                    "f": lambda: ctx.alloc(8 * (F + 1)), "b": lambda: ctx.alloc(2 * stride * a.clips + a.back_shift) + a.back_shift}[ch]()
    d_wav, d_out, d_off, d_back = bufs["w"], bufs["o"], bufs["f"], bufs["b"]
    if a.place > 1 and a.clips == 1:
        ctx.synth_dev(a.kind, 0x58330003, 0, stride * a.clips, d_wav)
        co, cb = [d_out], [d_back]
        for k in range(a.place - 1):
            ctx.alloc((k + 1) * 1237 * 1024); co.append(ctx.alloc(cap + 16)); cb.append(ctx.alloc(2 * stride * a.clips))
        ms = x3hip.place_buffers(ctx, p, d_wav, npc, co, cap, d_off, cb)
        best = min((ms[i][j], i, j) for i in range(a.place) for j in range(a.place))
        d_out, d_back = co[best[1]], cb[best[2]]
        print("placement: first %.4f best %.4f worst %.4f ms per step" % (ms[0][0], best[0], max(max(r) for r in ms)), end="; ")
        if os.environ.get("X3_PLACE_MATRIX"):
            print("\n" + "\n".join("  stream[%d]: " % i + " ".join("%.3f" % v for v in r) for i, r in enumerate(ms)))
    ctx.synth_dev(a.kind, 0x58330003, 0, stride * a.clips, d_wav)
    if a.loud > 0:   # every k-th frame loud
        k = max(1, int(round(1.0 / a.loud)))
        for f in range(k // 2, F, k):
            lo = f * p.spf
            ctx.synth_dev(1, 0x58330003 + f, lo, min(p.spf, n - lo), d_wav + 2 * lo)
    d_seg = ctx.alloc(8 * L.x3_seg_index_entries(F, C.byref(p), a.seg) + 8) if a.seg else None
    enc_seg = a.seg >= 4 and (a.seg & (a.seg - 1)) == 0   # (the encoder's index: a power of two; else the one the first decode records)
    def step():
        if a.decode_only:
            pass
        elif enc_seg:
            assert ctx.encode_dev_seg(d_wav, npc, p, d_out, cap, d_seg, a.seg, 0, d_off, n_clips=a.clips, clip_stride=stride) == 0
        else:
            assert ctx.encode_dev(d_wav, npc, p, d_out, cap, 0, d_off, n_clips=a.clips, clip_stride=stride) == 0
        if a.seg:
            assert ctx.decode_dev_seg(d_out, cap, d_off, F, p, d_back, stride * a.clips, d_seg, a.seg, n_per_clip=npc, n_clips=a.clips, clip_stride=stride) == 0
        else:
            assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, stride * a.clips, n_per_clip=npc, n_clips=a.clips, clip_stride=stride) == 0
    ctx.enable_kernel_timing(False)
    # (the first call: the encoder's result first -- dense content is encoded again inside x3_encode_result, the stream is
    # not valid before it -- then the decode)
    assert ctx.encode_dev(d_wav, npc, p, d_out, cap, 0, d_off, n_clips=a.clips, clip_stride=stride) == 0
    rc, pos, st = ctx.encode_result(); assert rc == 0
    if a.seg:
        assert ctx.decode_dev_seg(d_out, cap, d_off, F, p, d_back, stride * a.clips, d_seg, a.seg, record=True, n_per_clip=npc, n_clips=a.clips, clip_stride=stride) == 0
    else:
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, stride * a.clips, n_per_clip=npc, n_clips=a.clips, clip_stride=stride) == 0
    r = ctx.decode_result(); assert os.environ.get("X3_NOCHECK") or r[:3] == (0, F, 0), r
    ctx.enable_kernel_timing(not os.environ.get('X3_NOTIMING')); ctx.reset_kernel_time()
    for _ in range(a.steps): step()
    rc, pos2, st2 = (0, pos, st) if a.decode_only else ctx.encode_result(); r2 = ctx.decode_result()
    if os.environ.get("X3_WALL"):   # the whole step, without the timers' events in the queues
        import time
        ctx.enable_kernel_timing(False)
        for _ in range(10): step()
        ctx.decode_result(); t0 = time.perf_counter()
        for _ in range(a.steps): step()
        ctx.decode_result(); t1 = time.perf_counter()
        print("step wall %.4f ms" % ((t1 - t0) / a.steps * 1e3), end="  ")
    assert rc == 0 and pos2 == pos and (os.environ.get("X3_NOCHECK") or r2[:3] == (0, F, 0)), (rc, pos2, r2)
    names = ["encode", "decode", "sizes", "scan", "check", "dense"]
    print(" ".join("%s=%.3f" % (names[i], ctx.kernel_time(i)[0] / a.steps) for i in range(6)), "gen=%d dense_frames=%d" % (ctx.get_option("enc_gen_in_use"), ctx.get_option("last_dense_frames")), "ms; stream B/sample=%.4f" % (pos / n),
          "; wgs/CU=%d fallbacks=%d" % (ctx.get_option("stream_wgs_in_use"), ctx.get_option("encode_fallbacks")), a.opt, tag,
          "wav=%x out=%x back=%x" % (d_wav, d_out, d_back), flush=True)
    return d_wav, d_out, d_off, d_back
pads = []
if a.pad:
    pads.append(ctx.alloc(a.pad * 1024))
for rep in range(a.repeat):
    bufs = run_once("rep %d" % rep)
    if rep + 1 < a.repeat:
        for d in bufs: ctx.free(d - a.shift if d is bufs[0] else d)
        pads.append(ctx.alloc((rep + 1) * 1234 * 1024))
