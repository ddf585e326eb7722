"""Multi-channel extension (x3_mc.h): x3_encode_mc / x3_decode_stream_mc through the C ABI against the
oracle's x3o_encode_mc / x3o_decode_stream_mc.  The reference itself stops at MoreThanOneChannel (encoder.rs:55-57,
decoder.rs:90-94), so for more than one channel the oracle IS the definition (parity unpinned by the reference); with ONE
channel the extension must reproduce the reference-pinned mono path byte for byte, which is what ties it down."""
import numpy as np
import pytest

import oracle_lib as O
import x3hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = x3hip.Context(0)
    yield c
    c.close()


def _signals(n, k, seed):
    kinds = [2, 4, 3, 0, 2, 4, 3, 2]
    return [x3hip.synth(kinds[i % len(kinds)], seed + i, 0, n) for i in range(k)]


@pytest.mark.parametrize("n", [1, 2, 21, 9999, 10000, 10001, 30001, 70003])
def test_one_channel_is_the_mono_stream(ctx, n):
    """C = 1: the extension's bytes are x3_encode's (and the oracle's, which the reference's KATs pin)"""
    wav = x3hip.synth(2, 5, 0, n)
    rc, mono, st = ctx.encode(wav)
    rc_o, mono_o, st_o = O.encode(wav)
    rcm, mc, stm = ctx.encode_mc([wav])
    rcm_o, mc_o, stm_o = O.encode_mc([wav])
    assert rc == rc_o == rcm == rcm_o == 0
    assert np.array_equal(mono, mono_o) and np.array_equal(mc, mono) and np.array_equal(mc_o, mono)
    assert st.tolist() == stm.tolist() == stm_o.tolist()
    rcd, back, fok, ferr = ctx.decode_stream_mc(mc, 1, wav_cap=n + 16)
    assert (rcd, ferr) == (0, 0) and np.array_equal(back[0], wav)


def test_one_channel_with_long_dense_frames(ctx):
    """C = 1 with frames whose payload passes the 24 KB that only counts for several channels (blocks_per_frame 1 000 of
    full-scale noise: ~40 KB): x3_encode's bytes, not BAD_ARG"""
    p = x3hip.Params.default()
    p.blocks_per_frame = 1000
    po = O.Params.default()
    po.blocks_per_frame = 1000
    wav = x3hip.synth(1, 9, 0, 20000 * 3 + 777)
    rc, mono, st = ctx.encode(wav, p)
    rc_o, mono_o, st_o = O.encode(wav, po)
    rcm, mc, stm = ctx.encode_mc([wav], p)
    assert rc == rc_o == rcm == 0
    assert np.array_equal(mono, mono_o) and np.array_equal(mc, mono) and st.tolist() == stm.tolist() == st_o.tolist()


@pytest.mark.parametrize("n_ch", [2, 3, 4, 8])
@pytest.mark.parametrize("n,bpf", [(30001, 500), (4000, 100), (1, 500), (20, 50), (250123, 250)])
def test_channels_match_the_oracle(ctx, n_ch, n, bpf):
    p = x3hip.Params.default()
    p.blocks_per_frame = bpf
    po = O.Params.default()
    po.blocks_per_frame = bpf
    wavs = _signals(n, n_ch, 100 * n_ch + bpf)
    rc_o, x_o, st_o = O.encode_mc(wavs, po)
    rc_g, x_g, st_g = ctx.encode_mc(wavs, p)
    assert rc_g == rc_o, (rc_g, rc_o, ctx.last_error())
    if rc_o != 0:
        return
    assert np.array_equal(x_g, x_o), (x_g.size, x_o.size, np.nonzero(x_g[: min(x_g.size, x_o.size)] != x_o[: min(x_g.size, x_o.size)])[0][:4])
    assert st_g.tolist() == st_o.tolist()
    assert x_g[3] == n_ch  # <Num Channels> of the first frame
    rc_d, back, fok, ferr = ctx.decode_stream_mc(x_g, n_ch, p, wav_cap=n + 64)
    rc_do, back_o, fok_o, ferr_o = O.decode_stream_mc(x_o, n_ch, po, wav_cap=n + 64)
    assert (rc_d, fok, ferr) == (rc_do, fok_o, ferr_o) == (0, (n + 20 * bpf - 1) // (20 * bpf), 0)
    for k in range(n_ch):
        assert np.array_equal(back[k], wavs[k]) and np.array_equal(back_o[k], wavs[k]), k


def test_a_frame_no_reader_takes_is_frame_length(ctx):
    """three channels of full-scale noise in default frames: 60 KB of payload per frame -> FrameLength on both sides"""
    rng = np.random.default_rng(3)
    wavs = [rng.integers(-32768, 32768, 25000).astype(np.int16) for _ in range(3)]
    rc_o, _, _ = O.encode_mc(wavs)
    rc_g, _, _ = ctx.encode_mc(wavs)
    assert rc_o == rc_g == 10  # X3Error::FrameLength
    p = x3hip.Params.default(); p.blocks_per_frame = 100
    po = O.Params.default(); po.blocks_per_frame = 100
    rc_o, x_o, _ = O.encode_mc(wavs, po)
    rc_g, x_g, _ = ctx.encode_mc(wavs, p)
    assert rc_o == rc_g == 0 and np.array_equal(x_g, x_o)   # 2 000-sample frames: 12 KB of literal blocks, fine


def test_damaged_and_mismatched_streams(ctx):
    wavs = _signals(45000, 2, 77)
    rc, x, _ = ctx.encode_mc(wavs)
    assert rc == 0
    cases = {"intact": x}
    y = x.copy(); y[20 + 300] ^= 0x10; cases["payload bit"] = y                 # payload CRC of frame 0
    y = x.copy(); y[5] ^= 1; cases["header bit"] = y                            # header CRC
    cases["truncated"] = x[: x.size - 7]
    cases["half"] = x[: x.size // 2]
    for tag, s in cases.items():
        got = ctx.decode_stream_mc(s, 2, wav_cap=46000)
        want = O.decode_stream_mc(s, 2, wav_cap=46000)
        assert (got[0], got[2], got[3]) == (want[0], want[2], want[3]), (tag, got[0], got[2:], want[0], want[2:])
        for k in range(2):
            assert np.array_equal(got[1][k], want[1][k]), (tag, k)
    # a two-channel stream is not a one-channel stream, nor a three-channel one -- for the extension and for the reference
    for n_ch in (1, 3):
        got = ctx.decode_stream_mc(x, n_ch, wav_cap=46000)
        want = O.decode_stream_mc(x, n_ch, wav_cap=46000)
        assert got[0] == want[0] == 6 and got[2] == want[2] == 0, (n_ch, got[0], want[0])   # X3Error::MoreThanOneChannel
    rc, _, _, _ = ctx.decode_stream(x, wav_cap=46000)
    assert rc == 6   # the reference's reader (x3_decode_stream) refuses the frame, as the crate does
    # arguments
    assert ctx.encode_mc([wavs[0]] * 9)[0] == x3hip.ERR_BAD_ARG


def test_random_multichannel_sweep(ctx):
    """seeded sweep: channels x geometry x codes/thresholds x content x damage, GPU == oracle (status, bytes, samples)"""
    rng = np.random.default_rng(2026)
    for trial in range(60):
        n_ch = int(rng.integers(1, 9))
        bl = int(rng.choice([20, 20, 20, 7, 33, 60]))
        bpf = int(rng.integers(1, 120))
        n = int(rng.integers(1, 4 * bl * bpf + 50))
        p = x3hip.Params.default(); po = O.Params.default()
        for q in (p, po):
            q.block_len, q.blocks_per_frame = bl, bpf
        wavs = []
        for k in range(n_ch):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                w = np.cumsum(rng.integers(-4, 5, n)).astype(np.int16)
            elif kind == 1:
                w = rng.integers(-25, 26, n).astype(np.int16)
            elif kind == 2:
                w = (rng.integers(-2000, 2000, n) * (rng.random(n) < 0.1)).astype(np.int16)
            else:
                w = rng.integers(-32768, 32768, n).astype(np.int16)
            wavs.append(w)
        start = int(rng.integers(0, 4))
        cap = n_ch * O.encode_bound(n, po) + start + 64
        rc_o, x_o, st_o = O.encode_mc(wavs, po, start_pos=start, cap=cap)
        rc_g, x_g, st_g = ctx.encode_mc(wavs, p, start_pos=start, cap=cap)
        assert rc_g == rc_o, (trial, rc_g, rc_o, n_ch, bl, bpf, n, ctx.last_error())
        if rc_o:
            continue
        assert np.array_equal(x_g[start:], x_o[start:]) and st_g.tolist() == st_o.tolist(), (trial, n_ch, bl, bpf, n)
        s = x_o[start + (start & 1):].copy()
        if trial % 3 == 1 and s.size > 40:
            s[int(rng.integers(0, s.size))] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 2 and s.size > 40:
            s = s[: int(rng.integers(1, s.size))]
        got = ctx.decode_stream_mc(s, n_ch, p, wav_cap=n + 8)
        want = O.decode_stream_mc(s, n_ch, po, wav_cap=n + 8)
        assert (got[0], got[2], got[3]) == (want[0], want[2], want[3]), (trial, got[0], got[2:], want[0], want[2:])
        for k in range(n_ch):
            assert np.array_equal(got[1][k], want[1][k]), (trial, k)


def test_encode_is_one_pass_and_falls_back_to_two(ctx):
    """x3_encode_mc runs the general encoder's one-pass kernel (sizes by decoupled look-back: enc_gen_in_use 1) since round 5;
    the two-pass kernels give the same bytes (option two_pass), and a launch whose look-back gives up -- test hook lb_drop:
    a frame that never publishes its size -- is encoded again in two passes by x3_encode_mc itself (several channels are
    not the mono call x3_encode_result re-runs)."""
    p = x3hip.Params.default()
    p.blocks_per_frame = 100
    po = O.Params.default()
    po.blocks_per_frame = 100
    wavs = _signals(2000 * 37 + 5, 3, 4242)
    rc_o, x_o, st_o = O.encode_mc(wavs, po)
    assert rc_o == 0
    rc, x, st = ctx.encode_mc(wavs, p)
    assert rc == 0 and np.array_equal(x, x_o) and st.tolist() == st_o.tolist()
    assert ctx.get_option("enc_gen_in_use") == 1
    ctx.set_option("two_pass", 1)
    try:
        rc, x, st = ctx.encode_mc(wavs, p)
        assert rc == 0 and np.array_equal(x, x_o) and st.tolist() == st_o.tolist() and ctx.get_option("enc_gen_in_use") == 0
    finally:
        ctx.set_option("two_pass", 0)
    before = ctx.get_option("encode_fallbacks")
    ctx.set_option("lb_drop", 11)
    try:
        rc, x, st = ctx.encode_mc(wavs, p, start_pos=7)
        rc_o7, x_o7, st_o7 = O.encode_mc(wavs, po, start_pos=7)
        assert rc == 0 and np.array_equal(x[8:], x_o7[8:]) and st.tolist() == st_o7.tolist()
        assert ctx.get_option("encode_fallbacks") == before + 1 and ctx.get_option("enc_gen_in_use") == 0
        # the mono general path (block length 19: the one-pass kernel) falls back the same way, through x3_encode_result
        pm = x3hip.Params.make(19, 64)
        wav = x3hip.synth(2, 99, 0, 19 * 64 * 40 + 3)
        rc_m, x_m, st_m = ctx.encode(wav, pm)
        rc_om, x_om, st_om = O.encode(wav, O.Params.make(19, 64))
        assert rc_m == rc_om == 0 and np.array_equal(x_m, x_om) and st_m.tolist() == st_om.tolist()
        assert ctx.get_option("encode_fallbacks") == before + 2
    finally:
        ctx.set_option("lb_drop", -1)
    rc, x, st = ctx.encode_mc(wavs, p)
    assert rc == 0 and np.array_equal(x, x_o) and ctx.get_option("enc_gen_in_use") == 1
