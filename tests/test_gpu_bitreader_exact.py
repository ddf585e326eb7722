"""GPU parity on SHORT and CORRUPT payloads: the reference's BitReader is not "a bit string that is zero behind its
end" there (src/bitreader.rs:128-139: a zero run is extended by at most one peeked word; :148-163 + :76-92: behind
the last byte zero runs come back as phantom counts), so a frame whose header asks for more samples than its payload
encodes decodes "successfully" in the reference, to values that depend on that state machine.  The GPU decoders defer
such frames to an exact replay (x3_decode_replay.h); these tests feed thousands of them -- payloads cut inside
Rice0 / Rice1 / Rice3 / BFP / literal blocks, sample counts beyond the payload, zero runs of 32 bits and more (also
with codes[0] in {2, 3}, where such runs are valid indices), plain garbage -- and compare status and samples of every
frame with the oracle's decode_frame, through each of the three decoder kernels.  `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

STRIDE = 65544  # samples between the frames' output ranges (a header can ask for up to 65535 samples)


@pytest.fixture(scope="module")
def x3():
    import x3hip
    return x3hip


def oparams(p):
    return O.Params.make(p.block_len, p.blocks_per_frame, tuple(p.codes), tuple(p.thresholds))


def signals(x3, rng):
    """short clips whose blocks are mostly Rice0, Rice1, Rice3, BFP and literal"""
    out = []
    for amp in (1, 3, 8, 20, 300, 9000, 32000):
        n = int(rng.integers(30, 260))
        d = rng.integers(-amp, amp + 1, size=n)
        w = np.clip(np.cumsum(d), -32768, 32767).astype(np.int16) if amp < 9000 else \
            rng.integers(-amp, amp + 1, size=n).astype(np.int16)
        out.append(w)
    out.append(np.zeros(61, dtype=np.int16))
    out.append(x3.synth(2, 4242, 0, 241))
    return out


def crafted_frames(x3, rng, params, count):
    """-> list of (payload bytes, samples)"""
    op = oparams(params)
    base = []
    for w in signals(x3, rng):
        p1 = O.Params.make(params.block_len, 4000, tuple(params.codes), tuple(params.thresholds))  # one frame
        rc, s, _ = O.encode(w, p1)
        assert rc == 0
        base.append((s[20:].copy(), w.size))
    frames = []
    while len(frames) < count:
        pay, n = base[int(rng.integers(0, len(base)))]
        kind = int(rng.integers(0, 7))
        pay = pay.copy()
        if kind == 0:      # payload cut anywhere (odd lengths too), header samples unchanged
            pay = pay[: int(rng.integers(2, pay.size + 1))]
        elif kind == 1:    # more samples than the payload encodes
            n = n + int(rng.choice([1, 2, 3, 7, 19, 20, 21, 40, 41, 64, 333]))
        elif kind == 2:    # both
            pay = pay[: int(rng.integers(2, pay.size + 1))]
            n = n + int(rng.integers(0, 100))
        elif kind == 3:    # a long zero run somewhere (4..12 zero bytes), sometimes at the very end
            k = int(rng.integers(4, 13))
            at = int(rng.integers(2, max(3, pay.size - k + 1)))
            pay[at:at + k] = 0
            if rng.random() < 0.3:
                n += int(rng.integers(0, 50))
        elif kind == 4:    # garbage
            pay = rng.integers(0, 256, size=int(rng.integers(2, 120)), dtype=np.uint8)
            n = int(rng.integers(1, 400))
        elif kind == 5:    # sparse garbage: long zero runs with a few ones (phantom counts, one-word peeks)
            pay = np.zeros(int(rng.integers(3, 90)), dtype=np.uint8)
            for _ in range(int(rng.integers(0, 6))):
                pay[int(rng.integers(0, pay.size))] = 1 << int(rng.integers(0, 8))
            pay[2] |= int(rng.choice([0x40, 0x80, 0xC0]))   # a Rice block header up front
            n = int(rng.integers(1, 300))
        else:              # the untouched frame
            pass
        frames.append((pay, n))
    return frames


def run_batch(x3, ctx, params, frames, mode):
    """decode all frames in one launch; -> (status[F], list of sample arrays)"""
    F = len(frames)
    offs, chunks, pos = [], [], 0
    for pay, n in frames:
        hdr = x3.write_frame_header(n, 1, pay.size, O.crc16(pay))
        offs.append(pos)
        chunks += [hdr, pay]
        pos += 20 + pay.size
        if pos & 1:
            chunks.append(np.zeros(1, dtype=np.uint8))
            pos += 1
    stream = np.concatenate(chunks + [np.zeros(64, dtype=np.uint8)])
    d_x3 = ctx.alloc(stream.size)
    ctx.upload(d_x3, stream)
    d_off = ctx.alloc(8 * (F + 1))
    ctx.upload(d_off, np.array(offs + [pos], dtype=np.uint64))
    d_wav = ctx.alloc(2 * STRIDE * F)
    ctx.upload(d_wav, np.full(STRIDE * F, 0x5A5A, dtype=np.int16))
    d_st = ctx.alloc(4 * F)
    spf = params.block_len * params.blocks_per_frame
    if mode in ("offsets", "offsets_x4"):
        # caller-supplied sample offsets: the single-wave kernels -- or, with the caller's promise that they are multiples
        # of four samples (option wav_offsets_x4), the three-wave decoder and its list of rows (every row its own length)
        d_wo = ctx.alloc(8 * F)
        ctx.upload(d_wo, (np.arange(F, dtype=np.uint64) * STRIDE))
        ctx.set_option("wav_offsets_x4", 1 if mode == "offsets_x4" else 0)
        try:
            rc = ctx.decode_dev(d_x3, pos, d_off, F, params, d_wav, STRIDE * F, d_wav_offsets=d_wo, d_status=d_st)
        finally:
            ctx.set_option("wav_offsets_x4", 0)
    else:                     # a batch of F one-frame clips: the two-wave kernel for block_len 20
        d_wo = None
        rc = ctx.decode_dev(d_x3, pos, d_off, F, params, d_wav, STRIDE * F, n_per_clip=spf, n_clips=F,
                            clip_stride=STRIDE, d_status=d_st)
    assert rc == 0, ctx.last_error()
    rc, first_bad, st0, before = ctx.decode_result()
    assert rc == 0
    status = ctx.download(d_st, 4 * F, np.int32)
    wav = ctx.download(d_wav, 2 * STRIDE * F, np.int16).reshape(F, STRIDE)
    for d in (d_x3, d_off, d_wav, d_st) + ((d_wo,) if d_wo else ()):
        ctx.free(d)
    return status, wav, first_bad


def compare(x3, ctx, params, frames, mode):
    status, wav, first_bad = run_batch(x3, ctx, params, frames, mode)
    op = oparams(params)
    seen = {}
    exp_first_bad = len(frames)
    for i, (pay, n) in enumerate(frames):
        rc_o, w_o = O.decode_frame(pay, n, op)
        assert status[i] == rc_o, (mode, i, int(status[i]), rc_o, pay.size, n)
        if rc_o == 0:
            assert np.array_equal(wav[i, :n], w_o), (mode, i, pay.size, n)
            assert (wav[i, n:n + 8] == 0x5A5A).all()
        elif exp_first_bad == len(frames):
            exp_first_bad = i
        seen[rc_o] = seen.get(rc_o, 0) + 1
    assert first_bad == exp_first_bad
    return seen


@pytest.mark.parametrize("mode", ["batch", "offsets", "offsets_x4"])
def test_short_and_corrupt_payloads_default_params(x3, mode):
    rng = np.random.default_rng(20261004)
    p = x3.Params.default()
    frames = crafted_frames(x3, rng, p, 3000)
    ctx = x3.Context(0)
    try:
        seen = compare(x3, ctx, p, frames, mode)
    finally:
        ctx.close()
    # all three outcomes occur: phantom decodes that succeed, OutOfBoundsInverse, InvalidBPF
    assert seen.get(0, 0) > 300 and seen.get(5, 0) > 100 and seen.get(20, 0) > 100, seen


def test_single_wave_kernel_forced(x3):
    rng = np.random.default_rng(77)
    p = x3.Params.default()
    frames = crafted_frames(x3, rng, p, 1500)
    ctx = x3.Context(0)
    try:
        ctx.set_option("decode_single", 1)
        compare(x3, ctx, p, frames, "batch")
    finally:
        ctx.close()


@pytest.mark.parametrize("codes,thr,bl", [((2, 1, 3), (3, 8, 20), 20), ((3, 1, 3), (3, 8, 20), 20),
                                          ((3, 3, 3), (2, 9, 27), 20), ((0, 1, 3), (3, 8, 20), 7),
                                          ((2, 2, 2), (3, 8, 18), 33), ((0, 0, 0), (1, 2, 5), 20)])
def test_zero_runs_general_codes(x3, codes, thr, bl):
    """codes[0] in {2,3}: the r1 path bounds the run by inv_len 44 / 60, so runs of 32 and more are valid indices
    and the reference's one-word peek decides what they decode to; other block lengths take the single-wave kernels"""
    rng = np.random.default_rng(sum(codes) * 100 + bl)
    p = x3.Params.make(bl, 500, codes, thr)
    frames = crafted_frames(x3, rng, p, 1200)
    ctx = x3.Context(0)
    try:
        for mode in ("batch", "offsets"):
            compare(x3, ctx, p, frames, mode)
    finally:
        ctx.close()


def test_stream_walk_continues_over_phantom_frames(x3):
    """a frame that only decodes through phantom reads is a GOOD frame to the reference's walk: the frames behind it
    are decoded too, and the sample count includes it (x3_decode_stream, both walks, and x3_decode_frame)"""
    rng = np.random.default_rng(5)
    p = x3.Params.default()
    op = oparams(p)
    good = O.encode(x3.synth(2, 9, 0, 20000))[1]
    frames = crafted_frames(x3, rng, p, 400)
    ctx = x3.Context(0)
    try:
        tried = 0
        for pay, n in frames:
            rc_o, w_o = O.decode_frame(pay, n, op)
            stream = np.concatenate([good, x3.write_frame_header(n, 1, pay.size, O.crc16(pay)), pay, good])
            r_o = O.decode_stream(stream, op, wav_cap=200000)
            for host_walk in (1, 0):
                ctx.set_option("host_walk", host_walk)
                r_g = ctx.decode_stream(stream, p, wav_cap=200000)
                ctx.set_option("host_walk", -1)
                assert (r_g[0], r_g[2], r_g[3]) == (r_o[0], r_o[2], r_o[3]), (host_walk, pay.size, n, r_g[0], r_g[2:], r_o[0], r_o[2:])
                assert np.array_equal(r_g[1], r_o[1])
            rc_g, w_g = ctx.decode_frame(pay, n)
            assert rc_g == rc_o and (rc_o != 0 or np.array_equal(w_g, w_o))
            tried += 1
            if tried >= 60:
                break
        assert tried >= 40
    finally:
        ctx.close()
