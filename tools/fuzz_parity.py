#!/usr/bin/env python3
"""Parity soak: the HIP path (through the C ABI) against the CPU oracle on randomly drawn inputs, for a time budget.
Not part of the test suite (it runs as long as it is told to); a failure prints the seed and the drawn case, and
`--seed S --only T` replays trial T of seed S.
   python tools/fuzz_parity.py --minutes 10 [--seed 1]
Families: (e) the single-pass encoders on block_len 20 with mixed content, many frames and ragged tails;
          (g) arbitrary geometry / codes / thresholds; (d) decode of tampered streams with refreshed CRCs,
          truncations and header damage; (b) batches of clips through the device API, of equal and of different lengths
          (x3_encode_frames_dev); (a) .x3a archives in memory and
          the incremental reader; (f) decode_frame frame by frame, with and without x3_decode_prefetch; (w) WAV and .x3a FILES through the
          chunked pipeline (not in the default family set: file I/O); (s) the segment index: the encoder's against the one a
          decode records, decode by it with the stream and the index intact or damaged."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import x3hip
if os.environ.get("X3HIP_LIB"):
    x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
import oracle_lib as O

ctx = None   # the x3hip.Context under test (set by run())
START_TRIAL = 0   # --start: the sequence of trials begins here (a fresh context; for getting near a failing trial quickly)


def oparams(p):
    return O.Params.make(p.block_len, p.blocks_per_frame, tuple(p.codes), tuple(p.thresholds))


def content(rng, n):
    """a signal stitched from segments of different character, so that frames mix block types"""
    out = np.zeros(n, dtype=np.int64)
    i = 0
    while i < n:
        seg = int(rng.choice([1, 7, 19, 20, 21, 40, 100, 400, 1000, 5000, 20000]))
        seg = min(seg, n - i)
        kind = int(rng.integers(0, 9))
        if kind == 0:
            v = np.zeros(seg)
        elif kind == 1:
            v = np.full(seg, int(rng.choice([-32768, 32767, 1, -1, 12345])))
        elif kind == 2:
            v = rng.integers(-32768, 32768, size=seg)
        elif kind == 3:
            amp = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 10, 20, 21, 30, 64, 500, 8000, 16383, 16384]))
            v = np.cumsum(rng.integers(-amp, amp + 1, size=seg))
        elif kind == 4:
            v = np.round(float(rng.choice([50, 1000, 20000, 32767])) * np.sin(np.arange(seg) * float(rng.uniform(0.001, 1.5))))
        elif kind == 5:
            v = np.tile(np.array([32767, -32768]), seg // 2 + 1)[:seg]
        elif kind == 6:
            amp = int(rng.choice([3, 4, 8, 9, 20, 21]))  # exactly at the Rice thresholds
            v = np.cumsum(rng.choice([-amp, 0, amp], size=seg))
        elif kind == 7:
            v = rng.integers(-3, 4, size=seg) + (rng.random(seg) < 0.02) * rng.integers(-30000, 30000, size=seg)
        else:
            v = x3hip.synth(int(rng.choice([0, 1, 2, 3, 4])), int(rng.integers(0, 1 << 30)), 0, seg).astype(np.int64)
        v = np.asarray(v, dtype=np.int64)
        # fold back into the i16 range (cumulative sums wander)
        v = ((v + 32768) % 65536) - 32768 if kind in (3, 6) and rng.random() < 0.3 else np.clip(v, -32768, 32767)
        out[i:i + seg] = v
        i += seg
    return out.astype(np.int16)


def frame_offsets(stream):
    offs, pos = [], 0
    while pos + 20 <= stream.size:
        plen = int(stream[pos + 6]) << 8 | int(stream[pos + 7])
        if pos + 20 + plen > stream.size:
            break
        offs.append(pos)
        pos += 20 + plen
    return offs


def refresh_crcs(s, off):
    plen = min(int(s[off + 6]) << 8 | int(s[off + 7]), s.size - off - 20)
    hc = O.crc16(s[off:off + 16])
    s[off + 16], s[off + 17] = hc >> 8, hc & 0xFF
    pc = O.crc16(s[off + 20:off + 20 + plen])
    s[off + 18], s[off + 19] = pc >> 8, pc & 0xFF


def cmp_encode(wav, p, sp, tag, cap_cut=None):
    spf = p.block_len * p.blocks_per_frame
    cap = sp + 64 + ((wav.size + spf - 1) // spf) * 84 + 3 * wav.size   # the same, generous, capacity for both
    if cap_cut is not None:
        cap = int(cap_cut * cap)                                       # ... or one both run out of
    rc_o, out_o, st_o = O.encode(wav, oparams(p), start_pos=sp, cap=cap)
    # (one input in three: the host front end takes it in chunks of eight frames, if it has sixteen)
    ctx.set_option("host_chunk_frames", 8 if wav.size % 3 == 0 else -1)
    try:
        rc_g, out_g, st_g = ctx.encode(wav, p, start_pos=sp, cap=cap)
    finally:
        ctx.set_option("host_chunk_frames", 0)
    assert rc_g == rc_o, (tag, "status", rc_g, rc_o, ctx.last_error())
    if rc_o == 0:
        assert out_g.size == out_o.size, (tag, "size", out_g.size, out_o.size)
        if not np.array_equal(out_g[sp:], out_o[sp:]):
            bad = np.nonzero(out_g[sp:] != out_o[sp:])[0]
            raise AssertionError((tag, "bytes differ", int(bad[0]) + sp, int(bad.size)))
        assert st_g.tolist() == st_o.tolist(), (tag, "stats")
    return rc_o, out_o


def cmp_decode(stream, p, cap, tag):
    r_o = O.decode_stream(stream, oparams(p), wav_cap=cap)
    # the walk on the host, on the GPU, and the stream taken in chunks of 1 + (its length mod 5) frames
    # (the GPU walk twice: its fast path for clean chains with the general walk behind it, and the general walk alone)
    for host_walk, chunk, no_fast in ((1, -1, 0), (0, -1, 0), (0, -1, 1), (-1, 1 + len(stream) % 5, 0)):
        ctx.set_option("host_walk", host_walk)
        ctx.set_option("host_chunk_frames", chunk)
        ctx.set_option("index_no_fast", no_fast)
        try:
            r_g = ctx.decode_stream(stream, p, wav_cap=cap)
        finally:
            ctx.set_option("host_walk", -1)
            ctx.set_option("host_chunk_frames", 0)
            ctx.set_option("index_no_fast", 0)
        ok = (r_g[0], r_g[2], r_g[3]) == (r_o[0], r_o[2], r_o[3]) and np.array_equal(r_g[1], r_o[1])
        if not ok and os.environ.get("X3_FUZZ_DUMP"):   # what went in and what came out, for a post-mortem on the CPU
            os.makedirs(os.environ["X3_FUZZ_DUMP"], exist_ok=True)
            np.savez(os.path.join(os.environ["X3_FUZZ_DUMP"], "fail_%s.npz" % "_".join(str(x) for x in tag[0])), stream=stream,
                     got=r_g[1], want=r_o[1], got_rc=np.array([r_g[0], r_g[2], r_g[3]]), want_rc=np.array([r_o[0], r_o[2], r_o[3]]),
                     params=np.array([p.block_len, p.blocks_per_frame]), mode=np.array([host_walk, chunk, no_fast]))
            # the same call again, twice, and on the other decoder kernels: is it the data or the moment?
            for name, opts in (("again", {}), ("again2", {}), ("three_wave", {"decode_blocks": 0}), ("single", {"decode_blocks": 0, "decode_single": 1})):
                old = {k: ctx.get_option(k) for k in opts}
                for k, v in opts.items():
                    ctx.set_option(k, v)
                ctx.set_option("host_walk", host_walk); ctx.set_option("host_chunk_frames", chunk); ctx.set_option("index_no_fast", no_fast)
                r2 = ctx.decode_stream(stream, p, wav_cap=cap)
                ctx.set_option("host_walk", -1); ctx.set_option("host_chunk_frames", 0); ctx.set_option("index_no_fast", 0)
                for k, v in old.items():
                    ctx.set_option(k, v)
                d = np.nonzero(r2[1][:min(r2[1].size, r_o[1].size)] != r_o[1][:min(r2[1].size, r_o[1].size)])[0]
                print("   %-10s rc %s sizes %d/%d first diffs %s" % (name, (r2[0], r2[2], r2[3]), r2[1].size, r_o[1].size, d[:8].tolist()), flush=True)
        assert (r_g[0], r_g[2], r_g[3]) == (r_o[0], r_o[2], r_o[3]), (tag, host_walk, chunk, r_g[0], r_g[2:], r_o[0], r_o[2:])
        assert np.array_equal(r_g[1], r_o[1]), (tag, host_walk, chunk, "samples")
    return r_o


def fam_e(rng, tag):
    bpf = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 50, 100, 250, 500, 502, 504, 510, 511, 512])) if rng.random() < 0.7 else int(rng.integers(1, 513))
    p = x3hip.Params.make(20, bpf)
    spf = 20 * bpf
    frames = int(rng.choice([1, 2, 3, 5, 40, 300, 1500, 4000])) if bpf <= 16 else int(rng.choice([1, 2, 3, 7, 30, 90]))
    n = max(1, spf * frames - int(rng.integers(0, spf)) + int(rng.integers(0, 3)))
    wav = content(rng, n)
    sp = int(rng.choice([0, 0, 1, 2, 3, 18]))
    g2 = rng.random() < 0.25   # (a quarter of the trials on the second-generation kernel: it also serves the dense pass)
    ctx.set_option("enc_gen", 2 if g2 else 3)
    try:
        cut = float(rng.uniform(0.0, 0.3)) if rng.random() < 0.1 else None
        rc, out = cmp_encode(wav, p, sp, (tag, "e", bpf, n, sp, g2, cut), cut)
    finally:
        ctx.set_option("enc_gen", 3)
    if rc == 0:
        r = cmp_decode(out[(sp + 1) & ~1:], p, n, (tag, "e-dec", bpf, n))
        assert r[0] == 0 and np.array_equal(r[1], wav), (tag, "round trip")


def fam_n(rng, tag):
    """round 6: block lengths 10 and 40 on the single-pass encoders (wave encoder and second generation: a lane's run of 20
    samples is two blocks of 10 or half a block of 40), whole and ragged frames, other codes and thresholds now and then"""
    bl = int(rng.choice([10, 40]))
    per = 20 // bl if bl < 20 else 1
    bpf = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 50, 100, 125, 250, 255, 256])) * (2 if bl == 10 else 1) if rng.random() < 0.7 else int(rng.integers(1, 257 * max(per, 1)))
    codes, thr = (0, 1, 3), (3, 8, 20)
    if rng.random() < 0.2:
        offsets = [6, 11, 20, 28]
        codes = tuple(int(c) for c in rng.integers(0, 4, size=3))
        thr = tuple(int(rng.integers(0, offsets[c] + 1)) for c in codes)
    p = x3hip.Params.make(bl, bpf, codes, thr)
    if x3hip.lib().x3_params_validate(C.byref(p)) != 0:
        return
    spf = bl * bpf
    frames = int(rng.choice([1, 2, 3, 5, 40, 300, 1500])) if spf <= 400 else int(rng.choice([1, 2, 3, 7, 30, 90]))
    n = max(1, spf * frames - int(rng.integers(0, spf)) + int(rng.integers(0, 3)))
    wav = content(rng, n)
    sp = int(rng.choice([0, 0, 1, 2, 3, 18]))
    g2 = rng.random() < 0.4
    ctx.set_option("enc_gen", 2 if g2 else 3)
    try:
        cut = float(rng.uniform(0.0, 0.3)) if rng.random() < 0.1 else None
        if thr != (3, 8, 20):
            cut = None   # (a parameter set the reference can panic on AND no room: which of the two it meets first is not modelled -- INTEGRATION.md, "Limits")
        rc, out = cmp_encode(wav, p, sp, (tag, "n", bl, bpf, codes, thr, n, sp, g2, cut), cut)
    finally:
        ctx.set_option("enc_gen", 3)
    if rc == 0 and codes == (0, 1, 3):
        r = cmp_decode(out[(sp + 1) & ~1:], p, n, (tag, "n-dec", bl, bpf, n))
        # (with other thresholds the reference's own round trip is not always the identity -- BFP blocks of very few bits)
        assert thr != (3, 8, 20) or (r[0] == 0 and np.array_equal(r[1], wav)), (tag, "round trip")


def fam_g(rng, tag):
    offsets = [6, 11, 20, 28]
    bl = int(rng.integers(1, 61))
    bpf = int(rng.integers(1, 60)) if rng.random() < 0.7 else int(rng.choice([100, 333, 500, 1000]))
    codes = (0, 1, 3) if rng.random() < 0.5 else tuple(int(c) for c in rng.integers(0, 4, size=3))
    thr = tuple(int(rng.integers(0, offsets[c] + 1)) for c in codes)
    if bl * bpf > 38000:   # the library's frames are LDS-resident: <= ~40 000 samples ("Limits")
        bpf = 38000 // bl
    p = x3hip.Params.make(bl, bpf, codes, thr)
    if x3hip.lib().x3_params_validate(C.byref(p)) != 0:
        return
    n = int(rng.integers(1, 5 * bl * bpf + 40))
    wav = content(rng, n)
    sp = int(rng.integers(0, 4))
    rc, out = cmp_encode(wav, p, sp, (tag, "g", bl, bpf, codes, thr, n, sp))
    if rc == 0 and codes == (0, 1, 3):
        cmp_decode(out[(sp + 1) & ~1:], p, n, (tag, "g-dec", bl, bpf, thr, n))


def fam_k(rng, tag):
    """round 6's block-per-lane decoder: block lengths 10 and 40 (its units of 10 / 20 samples, two units per block of 40;
    the default decoder there) and 20 (option decode_blocks), frames of few and many blocks, damaged streams"""
    bl = int(rng.choice([10, 40, 20]))
    bpf = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 31, 32, 33, 50, 64, 100, 250, 500])) if rng.random() < 0.8 else int(rng.integers(1, 600))
    if bl * bpf > 20000:
        bpf = 20000 // bl
    p = x3hip.Params.make(bl, bpf)
    spf = bl * bpf
    n = spf * int(rng.integers(1, 9 if spf > 2000 else 200)) + int(rng.integers(0, spf))
    wav = content(rng, n)
    stream = O.encode(wav, oparams(p))[1]
    s = damage(rng, stream, frame_offsets(stream)) if rng.random() < 0.6 else stream
    ctx.set_option("decode_blocks", 1)
    try:
        cmp_decode(s, p, n + 70000, (tag, "k", bl, bpf, n))
    finally:
        ctx.set_option("decode_blocks", 0)


def damage(rng, stream, offs):
    """a copy of `stream` with one to three of its frames (byte offsets `offs`) tampered with, perhaps truncated"""
    s = stream.copy()
    for _ in range(int(rng.integers(1, 4))):
        fi = int(rng.integers(0, len(offs)))
        off = offs[fi]
        plen = int(stream[off + 6]) << 8 | int(stream[off + 7])    # (of the intact stream: headers get damaged too)
        kind = int(rng.integers(0, 8))
        if kind == 0 and plen > 2:       # one bit, CRCs refreshed
            s[off + 20 + int(rng.integers(0, plen))] ^= 1 << int(rng.integers(0, 8)); refresh_crcs(s, off)
        elif kind == 1 and plen > 12:    # zero run
            q = off + 20 + int(rng.integers(0, plen - 10)); s[q:q + int(rng.integers(1, 10))] = 0; refresh_crcs(s, off)
        elif kind == 2 and plen > 12:    # random bytes
            q = off + 20 + int(rng.integers(0, plen - 10)); k = int(rng.integers(1, 10))
            s[q:q + k] = rng.integers(0, 256, size=k, dtype=np.uint8); refresh_crcs(s, off)
        elif kind == 3:                  # sample count in the header raised or lowered
            ns = int(s[off + 4]) << 8 | int(s[off + 5])
            ns2 = max(0, min(65535, ns + int(rng.choice([-200, -20, -1, 1, 19, 20, 21, 300, 5000]))))
            s[off + 4], s[off + 5] = ns2 >> 8, ns2 & 0xFF; refresh_crcs(s, off)
        elif kind == 4:                  # payload bit without CRC refresh
            s[off + 20 + int(rng.integers(0, max(1, plen)))] ^= 0x40
        elif kind == 5:                  # header byte, header CRC refreshed or not
            s[off + int(rng.integers(0, 16))] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.5:
                hc = O.crc16(s[off:off + 16]); s[off + 16], s[off + 17] = hc >> 8, hc & 0xFF
        elif kind == 6 and plen > 40:    # all ones
            q = off + 20 + int(rng.integers(0, plen - 10)); s[q:q + int(rng.integers(1, 10))] = 0xFF; refresh_crcs(s, off)
        else:                            # the tail of the payload cleared (codes running off the end)
            k = int(rng.integers(1, min(40, max(2, plen)))); s[off + 20 + plen - k:off + 20 + plen] = 0; refresh_crcs(s, off)
    if rng.random() < 0.3:
        s = s[:int(rng.integers(0, s.size + 1))].copy()
    return s


def fam_d(rng, tag):
    bpf = int(rng.choice([3, 10, 50, 500]))
    p = x3hip.Params.make(20, bpf)
    n = 20 * bpf * int(rng.integers(2, 9)) + int(rng.integers(0, 20 * bpf))
    wav = content(rng, n)
    stream = O.encode(wav, oparams(p))[1]
    s = damage(rng, stream, frame_offsets(stream))
    cmp_decode(s, p, n + 70000, (tag, "d", bpf, n))


def fam_a(rng, tag):
    """.x3a archives in memory: encode == oracle; decode of intact and damaged archives through x3_x3a_decode and
    through the incremental reader (random window) == the oracle's x3a_to_wav"""
    rate = int(rng.choice([8000, 44100, 48000, 96000, 192000, 384000, 1, 999999, 1000000]))
    n = int(rng.integers(1, 60000))
    wav = content(rng, n)
    rc_o, x_o, st_o = O.x3a_encode(wav, rate)
    rc_g, x_g, st_g = ctx.x3a_encode(wav, rate)
    assert rc_g == rc_o and np.array_equal(x_g, x_o) and st_g.tolist() == st_o.tolist(), (tag, "a-enc", rate, n)
    arch = x_o
    if rng.random() < 0.7:
        hdr = 28 + (int(arch[8 + 6]) << 8 | int(arch[8 + 7]))
        offs = [hdr + o for o in frame_offsets(arch[hdr:])]
        if offs:
            arch = damage(rng, arch, offs)
        if rng.random() < 0.15 and arch.size > 40:   # the archive header itself
            arch = arch.copy(); arch[int(rng.integers(0, min(arch.size, hdr)))] ^= 1 << int(rng.integers(0, 8))
    cap = n + 70000
    r_o = O.x3a_decode(arch, wav_cap=cap)
    r_g = ctx.x3a_decode(arch, wav_cap=cap)
    assert (r_g[0],) + tuple(r_g[2:]) == (r_o[0],) + tuple(r_o[2:]), (tag, "a-dec", rate, n, r_g[0], r_g[2:], r_o[0], r_o[2:])
    assert np.array_equal(r_g[1], r_o[1]), (tag, "a-dec samples")
    # the incremental reader, as x3a_to_wav drives it
    ctx.set_option("reader_window_frames", int(rng.choice([1, 2, 3, 7, 64, 4096])))
    r = x3hip.Reader(ctx, arch)
    try:
        if r.rc:
            assert r.rc == r_o[0], (tag, "a-reader open", r.rc, r_o[0])
        else:
            out, rc = [], 0
            while True:
                rc, smp = r.next_frame()
                if rc or smp is None:
                    break
                out.append(smp)
            got = np.concatenate(out) if out else np.zeros(0, dtype=np.int16)
            assert (rc, r.frame_errors()) == (r_o[0], r_o[4]), (tag, "a-reader rc", rc, r.frame_errors(), r_o[0], r_o[4])
            assert np.array_equal(got, r_o[1]) and r.spec()[0] == r_o[2], (tag, "a-reader samples")
    finally:
        r.close()


def fam_f(rng, tag):
    """decoder::decode_frame frame by frame (with and without x3_decode_prefetch) on intact and damaged streams"""
    bpf = int(rng.choice([1, 3, 10, 50, 500]))
    p = x3hip.Params.make(20, bpf)
    n = 20 * bpf * int(rng.integers(1, 7)) + int(rng.integers(0, 20 * bpf))
    wav = content(rng, n)
    stream = O.encode(wav, oparams(p))[1]
    offs = frame_offsets(stream)
    s = damage(rng, stream, offs) if rng.random() < 0.6 else stream
    s = np.ascontiguousarray(s)
    pre = rng.random() < 0.5
    if pre:
        assert ctx.decode_prefetch(s, p) == 0
    try:
        for off in frame_offsets(s):
            plen = int(s[off + 6]) << 8 | int(s[off + 7])
            ns = int(s[off + 4]) << 8 | int(s[off + 5])
            if ns == 0:
                continue
            payload = s[off + 20:off + 20 + plen]
            cap = int(rng.choice([ns, ns, ns + 5, max(0, ns - 1)]))
            r_o = O.decode_frame(payload, ns, oparams(p), wav_cap=cap)
            r_g = ctx.decode_frame(payload, ns, p, wav_cap=cap)
            assert r_g[0] == r_o[0], (tag, "f rc", bpf, off, ns, plen, cap, r_g[0], r_o[0])
            if r_o[0] == 0:
                assert np.array_equal(r_g[1], r_o[1]), (tag, "f samples", bpf, off)
    finally:
        if pre:
            ctx.decode_prefetch(None)


def fam_b_ragged(rng, tag):
    """clips of DIFFERENT lengths: x3_encode_batch on host buffers (one launch set through x3_encode_frames_dev) and the
    device entry itself with clips at arbitrary sample offsets, against per-clip oracle streams; decoded back by frame
    index and per-frame sample offsets (with and without the promise that they are multiples of four)"""
    bpf = int(rng.choice([1, 2, 7, 8, 100, 500, 501, 512]))
    bl = 20 if rng.random() < 0.6 else int(rng.choice([7, 19, 33, 10, 40, 10, 40]))   # (10 and 40: the single-pass encoders' table forms since round 6)
    p = x3hip.Params.make(bl, bpf)
    spf = bl * bpf
    n_clips = int(rng.integers(1, 12))
    clips = [content(rng, int(rng.integers(1, 3 * spf + 40)) if rng.random() < 0.9 else 1) for _ in range(n_clips)]
    exp = [O.encode(c, oparams(p)) for c in clips]
    if any(e[0] != 0 for e in exp):
        return
    rc, out, offs, stats = ctx.encode_batch(clips, p)
    assert rc == 0, (tag, "b-ragged host", rc)
    for i, e in enumerate(exp):
        assert np.array_equal(out[offs[i]:offs[i + 1]], e[1]), (tag, "b-ragged host", i, bl, bpf, clips[i].size)
    assert stats.tolist() == np.sum([e[2] for e in exp], axis=0).tolist(), (tag, "b-ragged stats")
    # the device entry: clips anywhere (even / multiple-of-four / arbitrary offsets)
    mode = int(rng.integers(0, 3))
    step = [1, 2, 4][mode]
    starts, pos = [], 0
    for c in clips:
        pos += step * int(rng.integers(0, 9))
        starts.append(pos)
        pos += c.size
        pos += (-pos) % step
    total = pos + 16
    buf = np.zeros(total, dtype=np.int16)
    so, sn = [], []
    for s0, c in zip(starts, clips):
        buf[s0:s0 + c.size] = c
        for k in range(0, c.size, spf):
            so.append(s0 + k); sn.append(min(spf, c.size - k))
    F = len(so)
    L = x3hip.lib()
    cap = sum(L.x3_encode_bound(int(c.size), C.byref(p)) for c in clips) + 64
    d_wav = ctx.alloc(2 * total); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_wo = ctx.alloc(8 * F); d_back = ctx.alloc(2 * total)
    try:
        ctx.upload(d_wav, buf)
        sp = int(rng.choice([0, 0, 1, 6]))
        assert ctx.encode_frames_dev(d_wav, so, sn, p, d_out, cap, sp, d_off) == 0
        rc, end, st = ctx.encode_result()
        assert rc == 0, (tag, "b-ragged dev", rc)
        body0 = (sp + 1) & ~1
        expect = np.concatenate([e[1] for e in exp])
        got = ctx.download(d_out, end)
        assert end == body0 + expect.size and np.array_equal(got[body0:], expect), (tag, "b-ragged dev", bl, bpf, mode, sp)
        # (frames of 20 x 512 samples at most: a payload beyond the walk's 24 KB buffer is the reference's FrameLength)
        if tuple(p.codes) == (0, 1, 3) and step >= 2 and bl == 20:
            ctx.upload(d_wo, np.array(so, dtype=np.uint64)); ctx.upload(d_back, np.zeros(total, dtype=np.int16))
            ctx.set_option("wav_offsets_x4", 1 if (step == 4 and spf % 4 == 0) else 0)
            try:
                assert ctx.decode_dev(d_out, end, d_off, F, p, d_back, total, d_wav_offsets=d_wo) == 0
                r = ctx.decode_result()
            finally:
                ctx.set_option("wav_offsets_x4", 0)
            assert r[:3] == (0, F, 0), (tag, "b-ragged decode", r)
            assert np.array_equal(ctx.download(d_back, 2 * total).view(np.int16), buf), (tag, "b-ragged round trip", bl, bpf, mode)
    finally:
        for d in (d_wav, d_out, d_off, d_wo, d_back):
            ctx.free(d)


def fam_b(rng, tag):
    """clips of equal length through the device batch API against per-clip oracle streams"""
    if rng.random() < 0.5:
        return fam_b_ragged(rng, tag)
    bpf = int(rng.choice([2, 8, 100, 500]))
    p = x3hip.Params.make(20, bpf)
    spf = 20 * bpf
    n_clips = int(rng.integers(1, 9))
    npc = 8 * int(rng.integers(1, max(2, spf * 5 // 8)))
    wav = content(rng, npc * n_clips)
    L = x3hip.lib()
    fpc = L.x3_num_frames(npc, C.byref(p))
    F = fpc * n_clips
    cap = L.x3_encode_bound(npc, C.byref(p)) * n_clips
    d_wav = ctx.alloc(2 * wav.size); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * wav.size)
    try:
        ctx.upload(d_wav, wav)
        assert ctx.encode_dev(d_wav, npc, p, d_out, cap, 0, d_off, n_clips=n_clips, clip_stride=npc) == 0
        rc, pos, st = ctx.encode_result()
        assert rc == 0, (tag, rc)
        expect = np.concatenate([O.encode(wav[c * npc:(c + 1) * npc], oparams(p))[1] for c in range(n_clips)])
        got = ctx.download(d_out, pos)
        assert pos == expect.size and np.array_equal(got, expect), (tag, "b", bpf, n_clips, npc)
        assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, wav.size, n_per_clip=npc, n_clips=n_clips, clip_stride=npc) == 0
        r = ctx.decode_result()
        assert r[:3] == (0, F, 0), (tag, r)
        assert np.array_equal(ctx.download(d_back, 2 * wav.size).view(np.int16), wav), (tag, "b round trip")
    finally:
        for d in (d_wav, d_out, d_off, d_back):
            ctx.free(d)


def _write_wav(path, wav, rate, extra=()):
    import struct
    data = np.ascontiguousarray(wav, dtype="<i2").tobytes()
    fmt = struct.pack("<HHIIHH", 1, 1, rate, rate * 2, 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    for cid, payload in extra:
        body += cid + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    body += b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def fam_w(rng, tag):
    """files: x3_wav_to_x3a / x3_x3a_to_wav (chunked pipeline, random chunk sizes and worker counts) == the oracle's
    wav_to_x3a / x3a_to_wav, byte for byte; damaged archives too"""
    import tempfile
    d = tempfile.mkdtemp(prefix="x3fz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rate = int(rng.choice([8000, 44100, 96000, 192000]))
        n = int(rng.integers(0, 400000)) if rng.random() < 0.8 else int(rng.integers(0, 30))
        wav = content(rng, n) if n else np.zeros(0, dtype=np.int16)
        a, b_o, b_g, c_o, c_g = (os.path.join(d, x) for x in ("in.wav", "o.x3a", "g.x3a", "o.wav", "g.wav"))
        extra = [(b"LIST", bytes(int(rng.integers(0, 9))))] if rng.random() < 0.3 else []
        _write_wav(a, wav, rate, extra)
        ctx.set_option("file_chunk_frames", int(rng.choice([1, 2, 3, 5, 16, 64])))
        ctx.set_option("file_workers", int(rng.choice([1, 2, 3, 4])))
        rc_o, st_o = O.wav_to_x3a(a, b_o)
        rc_g, st_g = ctx.wav_to_x3a(a, b_g)
        assert rc_g == rc_o, (tag, "w enc rc", rc_g, rc_o)
        if rc_o == 0:
            x_o, x_g = open(b_o, "rb").read(), open(b_g, "rb").read()
            assert x_o == x_g and st_g.tolist() == st_o.tolist(), (tag, "w enc bytes", n, rate)
            if rng.random() < 0.5 and len(x_o) > 400:
                arch = np.frombuffer(x_o, dtype=np.uint8)
                hdr = 28 + (int(arch[14]) << 8 | int(arch[15]))
                offs = [hdr + o for o in frame_offsets(arch[hdr:])]
                if offs:
                    arch = damage(rng, arch, offs)
                    open(b_o, "wb").write(bytes(arch))
            r_o = O.x3a_to_wav(b_o, c_o)
            r_g = ctx.x3a_to_wav(b_o, c_g)
            assert tuple(r_g) == tuple(r_o), (tag, "w dec", r_g, r_o)
            if os.path.exists(c_o) or os.path.exists(c_g):
                assert open(c_o, "rb").read() == open(c_g, "rb").read(), (tag, "w dec bytes")
    finally:
        ctx.set_option("file_chunk_frames", 800)
        ctx.set_option("file_workers", 4)
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def fam_m(rng, tag):
    """multi-channel extension: x3_encode_mc / x3_decode_stream_mc == the oracle's x3o_encode_mc / x3o_decode_stream_mc over
    channels x geometry x content x start position, intact, damaged and truncated streams, wrong channel counts"""
    n_ch = int(rng.integers(1, 9))
    bl = int(rng.choice([20, 20, 20, 7, 33, 60, 1]))
    bpf = int(rng.integers(1, 200))
    n = int(rng.integers(1, 5 * bl * bpf + 50)) if rng.random() < 0.9 else int(rng.integers(1, 4))
    p, po = x3hip.Params.default(), O.Params.default()
    for q in (p, po):
        q.block_len, q.blocks_per_frame = bl, bpf
    wavs = [content(rng, n) for _ in range(n_ch)]
    start = int(rng.integers(0, 5))
    cap = n_ch * O.encode_bound(n, po) + start + 64   # (the same capacity on both sides: running out of it is part of the result)
    rc_o, x_o, st_o = O.encode_mc(wavs, po, start_pos=start, cap=cap)
    rc_g, x_g, st_g = ctx.encode_mc(wavs, p, start_pos=start, cap=cap)
    assert rc_g == rc_o, (tag, "m enc rc", rc_g, rc_o, n_ch, bl, bpf, n)
    if rc_o:
        return
    assert np.array_equal(x_g[start:], x_o[start:]) and st_g.tolist() == st_o.tolist(), (tag, "m enc bytes", n_ch, bl, bpf, n)
    s = x_o[start + (start & 1):].copy()
    what = int(rng.integers(0, 4))
    if what == 1 and s.size > 40:
        s[int(rng.integers(0, s.size))] ^= 1 << int(rng.integers(0, 8))
    elif what == 2 and s.size > 40:
        s = s[: int(rng.integers(1, s.size))]
    ask = n_ch if what != 3 else int(rng.integers(1, 9))
    got = ctx.decode_stream_mc(s, ask, p, wav_cap=n + 8)
    want = O.decode_stream_mc(s, ask, po, wav_cap=n + 8)
    assert (got[0], got[2], got[3]) == (want[0], want[2], want[3]), (tag, "m dec", got[0], got[2:], want[0], want[2:])
    for k in range(ask):
        assert np.array_equal(got[1][k], want[1][k]), (tag, "m samples", k)


fams = {"w": fam_w, "e": fam_e, "g": fam_g, "d": fam_d, "b": fam_b, "a": fam_a, "f": fam_f, "m": fam_m, "k": fam_k, "n": fam_n}


def fam_s(rng, tag):
    """the segment index (x3_encode_dev_seg / x3_decode_dev_seg): mixed content in frames of up to 512 blocks, one clip or a
    batch; the encoder's index = the one a frame-by-frame decode records; decoding by it -- every entry, every second one, as
    the library picks -- gives the oracle's samples, with the stream intact or damaged (CRCs refreshed or not) and with the
    index intact, partly overwritten or random"""
    L = x3hip.lib()
    bpf = int(rng.choice([8, 16, 33, 64, 100, 128, 250, 256, 500, 501, 512]))
    p = x3hip.Params.make(20, bpf)
    spf = 20 * bpf
    n_clips = int(rng.choice([1, 1, 1, 2, 5]))
    n_per = spf * int(rng.integers(1, 6)) + int(rng.choice([0, 0, 1, 19, 20, 21, spf // 2, spf - 1]))
    if n_clips > 1:
        n_per = (n_per + 3) & ~3
    sb = int(rng.choice([4, 8, 16, 32, 64, 128]))
    wavs = [content(rng, n_per) for _ in range(n_clips)]
    wav = np.concatenate(wavs)
    n = wav.size
    F = L.x3_num_frames(n_per, C.byref(p)) * n_clips
    cap = L.x3_encode_bound(n_per, C.byref(p)) * n_clips
    ne = max(1, L.x3_seg_index_entries(F, C.byref(p), sb))
    d = [ctx.alloc(2 * n + 64), ctx.alloc(cap + 64), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n + 64), ctx.alloc(8 * ne + 8), ctx.alloc(8 * ne + 8)]
    d_wav, d_out, d_off, d_back, d_seg, d_seg2 = d
    try:
        ctx.upload(d_wav, wav)
        ctx.upload(d_seg, rng.integers(0, 1 << 62, ne, dtype=np.uint64))
        assert ctx.encode_dev_seg(d_wav, n_per, p, d_out, cap, d_seg, sb, 0, d_off, n_clips=n_clips) == 0
        rc, pos, _ = ctx.encode_result()
        assert rc == 0, (tag, rc)
        stream = ctx.download(d_out, pos)
        ref = np.concatenate([O.encode(w, oparams(p))[1] for w in wavs])
        assert np.array_equal(stream, ref), (tag, "stream")
        nseg = (bpf + sb - 1) // sb
        have_index = nseg >= 2 and ctx.get_option("enc_gen_in_use") == 3
        if nseg >= 2:
            assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg2, sb, record=True, n_per_clip=n_per, n_clips=n_clips) == 0
            assert ctx.decode_result() == (0, F, 0, n), (tag, "record")
            if have_index:
                a, b = ctx.download(d_seg, 8 * ne, np.uint64), ctx.download(d_seg2, 8 * ne, np.uint64)
                assert np.array_equal(a, b), (tag, "index", np.flatnonzero(a != b)[:8].tolist())
        # damage: the stream, the index, both, neither
        offs = frame_offsets(stream)
        bad = stream
        if rng.random() < 0.6:
            # payload damage only (the device API places frames by the layout, the oracle's walk by the headers it reads:
            # the two agree as long as the headers do), CRCs refreshed or not
            bad = stream.copy()
            for _ in range(int(rng.integers(1, 4))):
                off = offs[int(rng.integers(0, len(offs)))]
                plen = int(stream[off + 6]) << 8 | int(stream[off + 7])
                if plen <= 2:
                    continue
                q = off + 20 + int(rng.integers(2, plen))
                k = min(int(rng.integers(1, 10)), off + 20 + plen - q)
                kind = int(rng.integers(0, 4))
                if kind == 0: bad[q] ^= np.uint8(1 << int(rng.integers(0, 8)))
                elif kind == 1: bad[q:q + k] = 0
                elif kind == 2: bad[q:q + k] = rng.integers(0, 256, size=k, dtype=np.uint8)
                else: bad[q:q + k] = 0xFF
                if rng.random() < 0.7:
                    refresh_crcs(bad, off)
        idx = ctx.download(d_seg2 if nseg >= 2 else d_seg, 8 * ne, np.uint64)
        r = rng.random()
        if r < 0.25 and ne > 1:
            k = int(rng.integers(1, ne)); idx[k] ^= np.uint64(1) << np.uint64(int(rng.integers(0, 49)))
        elif r < 0.4 and ne > 1:
            sel = rng.random(ne) < 0.2; sel[0] = False
            idx[sel] = rng.integers(0, 1 << 49, int(sel.sum()), dtype=np.uint64)
        elif r < 0.45:
            idx[0] = 0
        ctx.upload(d_out, bad); ctx.upload(d_seg, idx)
        # the oracle's walk over the same frames (frame f's bytes at offs[f]; a bad frame stops it)
        want = O.decode_stream(bad, oparams(p), wav_cap=n + 70000)
        for want_st in (0, nseg, 2):
            ctx.set_option("seg_stretches", want_st)
            ctx.upload(d_back, np.zeros(n, dtype=np.int16))
            assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n_per, n_clips=n_clips) == 0
            rc_d, first_bad, st, before = ctx.decode_result()
            if n_clips == 1:
                assert rc_d == 0 and first_bad == want[2] and before == want[1].size, (tag, want_st, first_bad, st, before, want[2], want[1].size)
                assert np.array_equal(ctx.download(d_back, 2 * before, np.int16), want[1]), (tag, want_st, "samples")
            else:   # (a batch: the oracle's walk knows nothing about clips; intact streams only)
                if bad is stream:
                    assert (rc_d, first_bad, st, before) == (0, F, 0, n), (tag, want_st)
                    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), (tag, want_st, "samples")
    finally:
        ctx.set_option("seg_stretches", 0)
        for q in d:
            ctx.free(q)


fams["s"] = fam_s


def run(seed=1, minutes=None, trials=None, families="egdbaf", only=-1, context=None):
    """draw and check cases until the time or the trial budget is used up -> {family: trials}"""
    global ctx
    own = context is None
    ctx = context if context is not None else x3hip.Context(0)
    t_end = time.time() + 60 * minutes if minutes is not None else None
    trace = open(os.environ["X3_FUZZ_TRACE"], "w") if os.environ.get("X3_FUZZ_TRACE") else None
    if os.environ.get("X3_FUZZ_TRACE_CALLS"):   # every Context call of the trial(s), appended before it is made
        calls = open(os.environ["X3_FUZZ_TRACE_CALLS"], "a")

        def brief(v):
            if isinstance(v, np.ndarray):
                return "ndarray(%s,%d)" % (v.dtype, v.size)
            if isinstance(v, (list, tuple)) and len(v) > 8:
                return "%s[%d]" % (type(v).__name__, len(v))
            if isinstance(v, int) and v > 1 << 32:
                return hex(v)
            return repr(v)[:80]

        def wrap(name, fn):
            def g(*a, **k):
                calls.write("%s(%s)\n" % (name, ", ".join([brief(v) for v in a] + ["%s=%s" % (q, brief(v)) for q, v in k.items()])))
                calls.flush()
                r = fn(*a, **k)
                if name == "alloc":
                    calls.write("  -> %s\n" % brief(r)); calls.flush()
                return r
            return g
        for name in dir(ctx):
            if not name.startswith("_") and callable(getattr(ctx, name)):
                setattr(ctx, name, wrap(name, getattr(ctx, name)))
    trial = START_TRIAL
    counts = {k: 0 for k in fams}
    try:
        while (t_end is None or time.time() < t_end) and (trials is None or trial < trials):
            if only >= 0:
                trial = only
            rng = np.random.default_rng([seed, trial])
            k = families[int(rng.integers(0, len(families)))]
            if trace is not None:     # (a trial that kills the process -- a GPU memory fault -- leaves its number here)
                trace.seek(0); trace.write("%d %d %s\n" % (seed, trial, k)); trace.flush()
            try:
                fams[k](rng, (seed, trial))
            except Exception:
                print("FAILED: seed %d trial %d family %s" % (seed, trial, k), flush=True)
                raise
            counts[k] += 1
            trial += 1
            if only >= 0:
                break
    finally:
        if own:
            ctx.close()
        else:  # a borrowed context goes back with the options the families touch at their defaults
            for name, value in (("reader_window_frames", 4096), ("enc_gen", 3), ("host_walk", -1), ("host_chunk_frames", 0), ("index_no_fast", 0),
                                ("file_chunk_frames", 800), ("file_workers", 4), ("seg_stretches", 0)):
                ctx.set_option(name, value)
        ctx = None
    return counts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--families", default="egdbaf")
    ap.add_argument("--start", type=int, default=0, help="begin the sequence of trials at this trial number (a fresh context)")
    a = ap.parse_args()
    START_TRIAL = a.start
    c = run(a.seed, a.minutes, None, a.families, a.only)
    print("fuzz_parity: seed %d, %d trials OK in %.1f min %s" % (a.seed, sum(c.values()), a.minutes, c), flush=True)
