"""The segment index (include/x3hip.h, "The SEGMENT INDEX"): decoding a frame on more than one lane.

Every test holds the segmented decode against the frame-by-frame decode of the same stream (which the rest of the suite
holds against the oracle) and against the oracle itself; the index is a hint that is never trusted, so every way of
breaking it must still give the oracle's samples."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def x3():
    import x3hip
    return x3hip


@pytest.fixture()
def ctx(x3):
    c = x3.Context(0)
    yield c
    c.close()


def _encode_dev(ctx, x3, wav, p, n_clips=1):
    n_per = wav.size // n_clips
    F = x3.lib().x3_num_frames(n_per, C.byref(p)) * n_clips
    cap = x3.lib().x3_encode_bound(n_per, C.byref(p)) * n_clips
    d_wav = ctx.alloc(2 * wav.size + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1))
    ctx.upload(d_wav, wav)
    assert ctx.encode_dev(d_wav, n_per, p, d_out, cap, 0, d_off, n_clips=n_clips) == 0
    rc, pos, _ = ctx.encode_result()
    assert rc == 0
    return d_wav, d_out, d_off, F, cap, pos


@pytest.mark.parametrize("sb", [64, 128, 4, 252])
@pytest.mark.parametrize("kind,n", [(2, 640_000), (2, 1_000_003), (4, 333_333), (1, 200_000), (0, 130_000), (3, 70_001)])
def test_segmented_decode_equals_the_serial_decode_and_the_oracle(ctx, x3, sb, kind, n):
    """record the index during a frame-by-frame decode, decode by it, compare: samples, statuses, and the index itself
    against the positions the oracle's encoder went through (block boundaries of its own stream)"""
    p = x3.Params.default()
    wav = x3.synth(kind, 1000 + sb, 0, n)
    d_wav, d_out, d_off, F, cap, pos = _encode_dev(ctx, x3, wav, p)
    rc_o, ref, _ = O.encode(wav)
    assert rc_o == 0 and np.array_equal(ctx.download(d_out, pos), ref)
    ne = x3.lib().x3_seg_index_entries(F, C.byref(p), sb)
    nseg = (500 + sb - 1) // sb
    assert ne == 1 + F * (nseg - 1)   # (a header word, then nseg - 1 entries per frame)
    d_seg = ctx.alloc(8 * ne + 8)
    d_back = ctx.alloc(2 * n + 64)
    ctx.upload(d_back, np.zeros(n, dtype=np.int16))
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, record=True, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
    seg = ctx.download(d_seg, 8 * ne, np.uint64)
    assert int(seg[0]) == (sb << 32) | 0x58335347
    seg = seg[1:].reshape(F, nseg - 1)
    # every entry of a block the frame has: valid, its sample = the input sample in front of the block
    for f in range(F):
        ns = min(10000, n - 10000 * f)
        nb = (ns - 1 + 19) // 20
        for j in range(1, nseg):
            e = int(seg[f, j - 1])
            if sb * j < nb:
                assert (e >> 48) & 1, (f, j)
                assert ((e >> 32) & 0xFFFF) == (int(wav[10000 * f + 20 * sb * j]) & 0xFFFF), (f, j)
                assert 16 <= (e & 0xFFFFFFFF) <= 8 * 20376
            else:
                assert e == 0, (f, j, hex(e))
    # by the index: as many stretches as the library picks, all of them, two
    for want in (0, nseg, 2):
        ctx.set_option("seg_stretches", want)
        ctx.upload(d_back, np.zeros(n, dtype=np.int16))
        assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, record=False, n_per_clip=n) == 0
        assert ctx.decode_result() == (0, F, 0, n)
        assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), want
        used = ctx.get_option("last_seg_stretches")
        assert 2 <= used <= nseg and (want == 0 or used <= max(want, 2)), (want, used)
    ctx.set_option("seg_stretches", 0)
    for d in (d_wav, d_out, d_off, d_seg, d_back):
        ctx.free(d)


def test_a_broken_index_costs_time_not_correctness(ctx, x3):
    """entries zeroed, shifted by a bit, pointing behind the payload, carrying the wrong sample, swapped between frames,
    random: the frames they belong to go through the reference's reader and the samples are the oracle's"""
    p = x3.Params.default()
    n, sb = 400_000, 64
    wav = x3.synth(2, 4242, 0, n)
    d_wav, d_out, d_off, F, cap, pos = _encode_dev(ctx, x3, wav, p)
    ne = x3.lib().x3_seg_index_entries(F, C.byref(p), sb)
    d_seg = ctx.alloc(8 * ne); d_back = ctx.alloc(2 * n)
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, record=True, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    good = ctx.download(d_seg, 8 * ne, np.uint64)
    rng = np.random.default_rng(5)
    cases = []
    a = good.copy(); a[:] = 0; cases.append(("all zero", a))
    a = good.copy(); a[0] = 0; cases.append(("no header: whole frames per lane", a))
    a = good.copy(); a[0] = (np.uint64(128) << np.uint64(32)) | np.uint64(0x58335347); cases.append(("header of another granularity", a))
    a = good.copy(); a[1:] = 0; cases.append(("header but no entry", a))
    a = good.copy(); a[3] += 1; cases.append(("one bit late", a))
    a = good.copy(); a[10] -= 1; cases.append(("one bit early", a))
    a = good.copy(); a[17] = (a[17] & ~np.uint64(0xFFFFFFFF)) | np.uint64(8 * 30000); cases.append(("behind the payload", a))
    a = good.copy(); a[20] ^= np.uint64(1 << 32); cases.append(("wrong sample", a))
    a = good.copy(); a[8:15] = good[15:22]; cases.append(("another frame's entries", a))
    a = good.copy(); a[::3] = rng.integers(0, 1 << 49, a[::3].size, dtype=np.uint64); cases.append(("random thirds", a))
    a = rng.integers(0, 1 << 63, good.size, dtype=np.uint64); a[0] = good[0]; cases.append(("all random", a))
    a = good.copy(); a[-1] = 0; cases.append(("last entry missing", a))
    for name, idx in cases:
        ctx.upload(d_seg, idx)
        ctx.upload(d_back, np.zeros(n, dtype=np.int16))
        assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n) == 0, name
        assert ctx.decode_result() == (0, F, 0, n), name
        assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), name


def test_segmented_decode_of_corrupt_streams_matches_the_oracle(ctx, x3):
    """damage in the payload (decode errors, wrong CRCs) with a GOOD index recorded before the damage: the same first bad
    frame, status and samples as the oracle's walk"""
    p = x3.Params.default()
    n, sb = 200_000, 64
    wav = x3.synth(2, 99, 0, n)
    rc, stream, _ = O.encode(wav)
    F = 20
    offs = [0]
    while offs[-1] < stream.size:
        offs.append(offs[-1] + 20 + ((int(stream[offs[-1] + 6]) << 8) | int(stream[offs[-1] + 7])))
    d_x3 = ctx.alloc(stream.size + 64); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
    ne = x3.lib().x3_seg_index_entries(F, C.byref(p), sb)
    d_seg = ctx.alloc(8 * ne)
    ctx.upload(d_x3, stream); ctx.upload(d_off, np.array(offs, dtype=np.uint64))
    assert ctx.decode_dev_seg(d_x3, stream.size, d_off, F, p, d_back, n, d_seg, sb, record=True, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    rng = np.random.default_rng(11)
    for trial in range(12):
        bad = stream.copy()
        f = int(rng.integers(0, F))
        where = offs[f] + 20 + int(rng.integers(2, offs[f + 1] - offs[f] - 20))
        bad[where] ^= np.uint8(1 << int(rng.integers(0, 8)))
        want = O.decode_stream(bad, wav_cap=n)
        ctx.upload(d_x3, bad)
        ctx.upload(d_back, np.zeros(n, dtype=np.int16))
        assert ctx.decode_dev_seg(d_x3, bad.size, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n) == 0
        rc_d, first_bad, st, before = ctx.decode_result()
        assert rc_d == 0 and first_bad == want[2] and before == want[1].size, (trial, first_bad, st, want[2])
        assert np.array_equal(ctx.download(d_back, 2 * before, np.int16), want[1])


def test_segmented_decode_of_a_batch_of_clips(ctx, x3):
    """clips side by side (config 5's layout) with a short last frame each: groups that span clips, frames without the
    later stretches"""
    p = x3.Params.default()
    n_clips, n_per, sb = 7, 47_000, 128
    wav = np.concatenate([x3.synth(2 if c % 2 else 4, 300 + c, 0, n_per) for c in range(n_clips)])
    d_wav, d_out, d_off, F, cap, pos = _encode_dev(ctx, x3, wav, p, n_clips=n_clips)
    n = wav.size
    ne = x3.lib().x3_seg_index_entries(F, C.byref(p), sb)
    d_seg = ctx.alloc(8 * ne); d_back = ctx.alloc(2 * n)
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, record=True, n_per_clip=n_per, n_clips=n_clips) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    ctx.upload(d_back, np.zeros(n, dtype=np.int16))
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n_per, n_clips=n_clips) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
    # clips SHORTER than a frame (6 000 samples: 300 of the parameters' 500 blocks): the index's pitch follows the parameters,
    # so x3_seg_index_entries, the encoder and the decoder agree on the layout whatever the clips' lengths are
    n_clips, n_per, sb = 9, 6_000, 32
    wav = np.concatenate([x3.synth(2, 900 + c, 0, n_per) for c in range(n_clips)])
    n = wav.size
    L = x3.lib()
    F = L.x3_num_frames(n_per, C.byref(p)) * n_clips
    assert F == n_clips
    cap = L.x3_encode_bound(n_per, C.byref(p)) * n_clips
    ne = L.x3_seg_index_entries(F, C.byref(p), sb)
    assert ne == 1 + F * 15
    d_wav2 = ctx.alloc(2 * n + 64); d_out2 = ctx.alloc(cap + 16); d_off2 = ctx.alloc(8 * (F + 1)); d_back2 = ctx.alloc(2 * n)
    d_seg2 = ctx.alloc(8 * ne); d_seg3 = ctx.alloc(8 * ne)
    ctx.upload(d_wav2, wav)
    assert ctx.encode_dev_seg(d_wav2, n_per, p, d_out2, cap, d_seg2, sb, 0, d_off2, n_clips=n_clips) == 0
    rc, pos2, _ = ctx.encode_result()
    assert rc == 0
    assert ctx.decode_dev_seg(d_out2, pos2, d_off2, F, p, d_back2, n, d_seg3, sb, record=True, n_per_clip=n_per, n_clips=n_clips) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    a, b = ctx.download(d_seg2, 8 * ne, np.uint64), ctx.download(d_seg3, 8 * ne, np.uint64)
    assert np.array_equal(a, b) and int((a[1:].reshape(F, 15) != 0).sum(axis=1).max()) == 9   # (blocks 32 .. 288 of 300)
    ctx.upload(d_back2, np.zeros(n, dtype=np.int16))
    assert ctx.decode_dev_seg(d_out2, pos2, d_off2, F, p, d_back2, n, d_seg2, sb, n_per_clip=n_per, n_clips=n_clips) == 0
    assert ctx.decode_result() == (0, F, 0, n) and ctx.get_option("last_seg_stretches") >= 2
    assert np.array_equal(ctx.download(d_back2, 2 * n, np.int16), wav)


@pytest.mark.parametrize("sb", [32, 64, 128])
def test_the_encoders_index_is_the_one_a_serial_decode_records(ctx, x3, sb):
    """x3_encode_dev_seg: the wave encoder's prefix scan and its input give the same index, word for word, as a frame-by-
    frame decode of the stream records -- quiet, mixed (every seventh frame loud: the dense pass writes those, the wave
    encoder still sizes and indexes them) and with a short last frame; decode by it = the input"""
    p = x3.Params.default()
    n = 1_234_567
    wav = x3.synth(2, 77 + sb, 0, n)
    for f in range(3, n // 10000, 7):
        wav[10000 * f:10000 * (f + 1)] = x3.synth(1, 500 + f, 0, 10000)
    L = x3.lib()
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
    ne = L.x3_seg_index_entries(F, C.byref(p), sb)
    d_wav = ctx.alloc(2 * n + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
    d_seg = ctx.alloc(8 * ne); d_seg2 = ctx.alloc(8 * ne)
    ctx.upload(d_wav, wav)
    ctx.upload(d_seg, np.full(ne, 0xDEADBEEFDEADBEEF, dtype=np.uint64))
    assert ctx.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, sb, 0, d_off) == 0
    rc, pos, _ = ctx.encode_result()
    assert rc == 0 and ctx.get_option("enc_gen_in_use") == 3 and ctx.get_option("last_dense_frames") > 10
    assert np.array_equal(ctx.download(d_out, pos), O.encode(wav)[1])
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg2, sb, record=True, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    a, b = ctx.download(d_seg, 8 * ne, np.uint64), ctx.download(d_seg2, 8 * ne, np.uint64)
    assert np.array_equal(a, b), np.flatnonzero(a != b)[:10]
    ctx.set_option("seg_stretches", 500 // sb + 1)
    ctx.upload(d_back, np.zeros(n, dtype=np.int16))
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n) and ctx.get_option("last_seg_stretches") >= 2
    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
    ctx.set_option("seg_stretches", 0)


def test_an_encoder_that_cannot_index_says_so(ctx, x3):
    """other layouts (here: block_len 10, the general encoder) leave a header that says "no index"; decoding by it is the
    frame-by-frame decode"""
    p = x3.Params.make(10, 300)
    n = 100_000
    wav = x3.synth(2, 5, 0, n)
    L = x3.lib()
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
    ne = L.x3_seg_index_entries(F, C.byref(p), 64)
    d_wav = ctx.alloc(2 * n + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
    d_seg = ctx.alloc(8 * ne)
    ctx.upload(d_wav, wav)
    ctx.upload(d_seg, np.full(ne, 0xDEADBEEFDEADBEEF, dtype=np.uint64))
    assert ctx.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, 64, 0, d_off) == 0
    rc, pos, _ = ctx.encode_result()
    assert rc == 0 and int(ctx.download(d_seg, 8, np.uint64)[0]) == 0
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, 64, n_per_clip=n) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
    # bad arguments
    assert ctx.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, 48, 0, d_off) == x3.ERR_BAD_ARG
    assert ctx.decode_dev_seg(d_out, pos, d_off, F, p, d_back, n, d_seg, 6, n_per_clip=n) == x3.ERR_BAD_ARG


def test_hip_graph_replays_encode_and_decode(ctx, x3):
    """x3_graph_*: a short stream's encode + decode by stretches recorded once and replayed on NEW contents of the same
    buffers -- stream and samples equal the oracle's every time; recording without a first pass outside the capture (the
    buffers would have to grow) fails cleanly; a graph of one context is refused by another"""
    p = x3.Params.default()
    L = x3.lib()
    n, sb = 250_000, 32
    F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
    ne = L.x3_seg_index_entries(F, C.byref(p), sb)
    d_wav = ctx.alloc(2 * n + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
    d_seg = ctx.alloc(8 * ne)

    def calls():
        assert ctx.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, sb, 0, d_off) == 0
        assert ctx.decode_dev_seg(d_out, cap, d_off, F, p, d_back, n, d_seg, sb, n_per_clip=n) == 0
    # a fresh context has nothing allocated: recording must refuse, and leave the context usable
    c2 = x3.Context(0)
    try:
        c2.graph_begin()
        assert c2.encode_dev_seg(d_wav, n, p, d_out, cap, d_seg, sb, 0, d_off) == x3.ERR_BAD_ARG
        assert "grow" in c2.last_error()
        try:
            c2.graph_destroy(c2.graph_end())    # (whatever was recorded up to there: nothing that runs a kernel)
        except x3.X3Error:
            pass
        # ... and the context works as before
        w0 = x3.synth(2, 77, 0, 30_000)
        rc0, s0, _ = c2.encode(w0)
        assert rc0 == 0 and np.array_equal(s0, O.encode(w0)[1])
    finally:
        c2.close()
    wav = x3.synth(2, 1, 0, n)
    ctx.upload(d_wav, wav)
    calls()
    assert ctx.encode_result()[0] == 0 and ctx.decode_result() == (0, F, 0, n)
    ctx.graph_begin()
    calls()
    g = ctx.graph_end()
    try:
        for seed in (2, 3, 4, 5):
            wav = x3.synth(2 if seed % 2 else 4, seed, 0, n)
            if seed == 4:
                wav[30_000:40_000] = x3.synth(1, 9, 0, 10_000)     # a loud frame: the dense pass is part of the graph
            ctx.upload(d_wav, wav)
            ctx.upload(d_back, np.zeros(n, dtype=np.int16))
            ctx.graph_launch(g)
            rc, pos, stats = ctx.encode_result()
            assert rc == 0
            rc_o, ref, st_o = O.encode(wav)
            assert pos == ref.size and np.array_equal(ctx.download(d_out, pos), ref) and list(stats) == st_o.tolist(), seed
            assert ctx.decode_result() == (0, F, 0, n)
            assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), seed
        # ordinary calls still work behind replays (the encoders' epochs go on)
        calls()
        assert ctx.encode_result()[0] == 0 and ctx.decode_result() == (0, F, 0, n)
        c3 = x3.Context(0)
        try:
            assert L.x3_graph_launch(c3._h, g) == x3.ERR_BAD_ARG
        finally:
            c3.close()
    finally:
        ctx.graph_destroy(g)
