"""Block lengths 10 and 40 on the single-pass encoder (x3-rust_amd/csrc/x3_encode_stream2_kernel.h, BL).
A lane of the second-generation kernel holds 20 samples whatever a block is: two blocks of 10 (two filters, two headers in
one run of bits) or half a block of 40 (the pair of lanes shares the largest difference by DPP; the even lane writes the
header).  The wave encoder takes a block of 40 as two of a lane's runs of 20 and a run as two blocks of 10
(x3_encode_wave_kernel.h, x3w_analyse40 / x3w_analyse10).
Streams and statistics against the oracle (encoder.rs:170-315 -- the reference's encoder has no special case for any
block length), on whole frames, ragged tails, batches of clips, frames from a table, every filter next to every other,
and the generation in use read back (2 on frames of a multiple of four samples, the general kernel elsewhere)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def x3():
    import x3hip
    return x3hip


@pytest.fixture(scope="module")
def ctx(x3):
    c = x3.Context(0)
    yield c
    c.close()


def patchwork(seed, n):
    """runs of 3..70 samples whose differences stay inside one of the encoder's classes: silence, each Rice code's range and
    its edges, BFP widths, literals, saturating jumps -- so that blocks of every type and lanes of every mix sit side by side"""
    rng = np.random.default_rng(seed)
    out = np.zeros(n, dtype=np.int64)
    pos, level = 0, 0
    amps = (0, 1, 2, 3, 4, 7, 8, 9, 19, 20, 21, 31, 32, 100, 1000, 8191, 8192, 16383, 16384, 30000, 65535)
    while pos < n:
        ln = int(rng.integers(3, 71))
        a = amps[int(rng.integers(0, len(amps)))]
        d = rng.integers(-a, a + 1, size=ln)
        if a and rng.integers(0, 4) == 0:
            d[int(rng.integers(0, ln))] = a if rng.integers(0, 2) else -a   # the class's edge itself
        seg = level + np.cumsum(d)
        seg = np.clip(seg, -32768, 32767)
        m = min(ln, n - pos)
        out[pos:pos + m] = seg[:m]
        level = int(seg[m - 1])
        pos += m
    return out.astype(np.int16)


GEOMS = [(10, 1000), (10, 1024), (10, 500), (10, 2), (10, 6), (10, 64), (40, 250), (40, 256), (40, 37), (40, 1), (40, 3), (40, 500)]


@pytest.mark.parametrize("bl,bpf", GEOMS)
def test_streams_and_statistics_equal_the_oracles(ctx, x3, bl, bpf):
    p, po = x3.Params.make(bl, bpf), O.Params.make(bl, bpf)
    spf = bl * bpf
    for k, (nfr, tail) in enumerate(((1, 0), (1, 1), (2, 2), (3, spf // 2), (5, spf - 1), (7, 21), (40, 0), (129, 39), (300, 11))):
        n = min(nfr * spf + tail, 2_500_000 // spf * spf + tail)
        for kind in ("patch", 0, 1, 2, 4):
            if kind != "patch" and k not in (1, 4, 7):
                continue
            wav = patchwork(97 * bl + bpf + k, n) if kind == "patch" else x3.synth(kind, 4400 + bl + bpf + k, 0, n)
            for start_pos, gen in ((0, 3), (7, 3), (0, 2)):
                ctx.set_option("enc_gen", gen)   # (also forgets what earlier calls said about dense content)
                rc_o, so, st_o = O.encode(wav, po, start_pos=start_pos)
                rc, s, st = ctx.encode(wav, p, start_pos=start_pos)
                assert rc == rc_o == 0, (rc, rc_o, ctx.last_error())
                want = 1 if spf % 4 or (min(spf, n) + 18) // 20 > 512 else gen   # (frames of at most 512 runs of 20 samples)
                assert ctx.get_option("enc_gen_in_use") == want, (bl, bpf, n, ctx.get_option("enc_gen_in_use"))
                assert s.size == so.size and np.array_equal(s[start_pos:], so[start_pos:]), (bl, bpf, n, kind, start_pos, gen,
                                                                                             int(np.argmax(s[:so.size] != so)))
                assert st.tolist() == st_o.tolist(), (bl, bpf, n, kind, gen)
            ctx.set_option("enc_gen", 3)
            if k in (1, 7) and kind == "patch":
                # (back again, as the reference's decoder has it: a loud frame of 20 000 samples is beyond its payload limit)
                r, o = ctx.decode_stream(s, p, wav_cap=n + 8), O.decode_stream(s, po, wav_cap=n + 8)
                assert (r[0], r[2], r[3]) == (o[0], o[2], o[3]) and np.array_equal(r[1], o[1])
                assert o[0] != 0 or np.array_equal(r[1], wav)


@pytest.mark.parametrize("bl,bpf", [(10, 1000), (40, 250), (10, 36), (40, 9)])
def test_batches_of_clips(ctx, x3, bl, bpf):
    """x3_encode_batch: equally long clips in one launch (clip stride a multiple of four samples: the single-pass kernel) and
    ragged ones (a launch per group of lengths)"""
    p, po = x3.Params.make(bl, bpf), O.Params.make(bl, bpf)
    spf = bl * bpf
    for lens in ([3 * spf + 8] * 9, [spf] * 70, [2 * spf + 1, 5, spf, 1, 3 * spf + 2 * bl + 3, 2 * spf + 1], [4 * bl] * 200):
        clips = [patchwork(31 * i + bl, n) for i, n in enumerate(lens)]
        rc, out, offs, st = ctx.encode_batch(clips, p)
        assert rc == 0, ctx.last_error()
        tot = np.zeros(6, dtype=np.uint64)
        for i, cl in enumerate(clips):
            rc_o, so, st_o = O.encode(cl, po)
            assert rc_o == 0
            got = out[offs[i]:offs[i + 1]]
            assert got.size >= so.size and np.array_equal(got[:so.size], so), (bl, bpf, i, len(cl))
            tot += st_o
        assert st.tolist() == tot.tolist()


@pytest.mark.parametrize("bl,bpf", [(10, 1000), (40, 250)])
def test_frames_from_a_table(ctx, x3, bl, bpf):
    """x3_encode_frames_dev: every frame its own source offset and sample count (the TAB instantiations)"""
    p, po = x3.Params.make(bl, bpf), O.Params.make(bl, bpf)
    spf = bl * bpf
    rng = np.random.default_rng(bl)
    wav = patchwork(5 + bl, 40 * spf)
    F = 90
    src_n = rng.integers(1, spf + 1, size=F).astype(np.uint32)
    src_n[::7] = spf
    src_off = (rng.integers(0, wav.size - spf, size=F) & ~1).astype(np.uint64)   # (dword-aligned sources: the table's even form)
    d_wav = ctx.alloc(2 * wav.size)
    cap = int(sum(20 + 2 * int(n) + (int(n) // bl + 1) + 4 for n in src_n)) + 64
    d_out = ctx.alloc(cap)
    d_off = ctx.alloc(8 * (F + 1))
    try:
        ctx.upload(d_wav, wav)
        rc = ctx.encode_frames_dev(d_wav, src_off, src_n, p, d_out, cap, d_frame_offsets=d_off)
        assert rc == 0, ctx.last_error()
        rc, pos, _ = ctx.encode_result()
        assert rc == 0
        assert ctx.get_option("enc_gen_in_use") in (2, 3)
        got = ctx.download(d_out, pos, np.uint8)
        offs = ctx.download(d_off, 8 * (F + 1), np.uint64)
        for f in range(F):
            fr = wav[int(src_off[f]):int(src_off[f]) + int(src_n[f])]
            rc_o, so, _ = O.encode(fr, po)
            assert rc_o == 0
            a = int(offs[f])
            assert np.array_equal(got[a:a + so.size], so), (bl, f, int(src_n[f]))
    finally:
        for d in (d_wav, d_out, d_off):
            ctx.free(d)


def test_other_block_lengths_stay_on_the_general_kernel(ctx, x3):
    for bl, bpf in ((30, 100), (60, 50), (8, 100), (12, 250)):
        p, po = x3.Params.make(bl, bpf), O.Params.make(bl, bpf)
        wav = patchwork(bl, 7 * bl * bpf + 5)
        rc, s, st = ctx.encode(wav, p)
        rc_o, so, st_o = O.encode(wav, po)
        assert rc == rc_o == 0 and np.array_equal(s, so) and st.tolist() == st_o.tolist()
        assert ctx.get_option("enc_gen_in_use") == 1
