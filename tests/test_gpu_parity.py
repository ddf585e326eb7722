"""GPU parity tests: the HIP path (through the C ABI, libx3hip.so) against the CPU oracle and the
reference's golden vectors.  Everything here needs a real MI355X: `pytest -m gpu`.
Bar: bit-exact (integer/byte work)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(G, name)))


@pytest.fixture(scope="module")
def x3():
    import x3hip
    return x3hip


@pytest.fixture(scope="module")
def ctx(x3):
    c = x3.Context(0)
    yield c
    c.close()


def oparams(p):
    return O.Params.make(p.block_len, p.blocks_per_frame, tuple(p.codes), tuple(p.thresholds))


def check_encode(ctx, x3, wav, params=None, start_pos=0):
    params = params or x3.Params.default()
    rc_o, out_o, st_o = O.encode(wav, oparams(params), start_pos=start_pos)
    rc_g, out_g, st_g = ctx.encode(wav, params, start_pos=start_pos)
    assert rc_g == rc_o, (rc_g, rc_o, ctx.last_error())
    if rc_o == 0:
        assert out_g.size == out_o.size
        assert np.array_equal(out_g[start_pos:], out_o[start_pos:])
        assert st_g.tolist() == st_o.tolist()
    return out_o


def check_decode(ctx, x3, stream, params=None, wav_cap=None):
    params = params or x3.Params.default()
    r_o = O.decode_stream(stream, oparams(params), wav_cap=wav_cap)
    # x3_decode_stream walks the frame headers on the host for short streams and on the GPU for long ones, and takes
    # long streams in chunks of whole frames (downloads beside uploads): all three against the oracle, whatever the size
    chunk = 3 if len(stream) < (4 << 20) else 800
    for host_walk, chunk_frames in ((1, -1), (0, -1), (-1, chunk)):
        ctx.set_option("host_walk", host_walk)
        ctx.set_option("host_chunk_frames", chunk_frames)
        try:
            r_g = ctx.decode_stream(stream, params, wav_cap=wav_cap)
        finally:
            ctx.set_option("host_walk", -1)
            ctx.set_option("host_chunk_frames", 0)
        assert (r_g[0], r_g[2], r_g[3]) == (r_o[0], r_o[2], r_o[3]), (host_walk, chunk_frames, r_g[0], r_g[2:], r_o[0], r_o[2:])
        assert np.array_equal(r_g[1], r_o[1]), (host_walk, chunk_frames)
    return r_o


# ------------------------------------------------------------------ golden vectors through the C ABI

def test_golden_encode_frame(ctx, x3):
    for f in load("encoder_kat.json")["frames"]:
        wav = np.array(f["wav"], dtype=np.int16)
        rc, out, stats = ctx.encode_frame(wav)
        assert rc == 0, ctx.last_error()
        assert out.tolist() == f["expected"], f["name"]
        rc, out, _ = ctx.encode(wav)
        assert rc == 0 and out.tolist() == f["expected"]


def test_golden_encode_blocks(ctx, x3):
    """encode_frame([prev, block...]) = 16 raw bits + the block; compare the block's bits."""
    for b in load("encoder_kat.json")["blocks"]:
        wav = np.array(b["wav"], dtype=np.int16)
        rc, out, _ = ctx.encode_frame(wav)
        assert rc == 0
        bits = np.unpackbits(out[20:])[16:]
        exp = np.unpackbits(np.array(b["expected"], dtype=np.uint8))[b["prepad_zero_bits"]:]
        n = min(bits.size, exp.size)  # both end with alignment zeros of different lengths
        assert np.array_equal(bits[:n], exp[:n]), b["name"]
        assert not bits[n:].any() and not exp[n:].any()


def test_golden_decode(ctx, x3):
    for f in load("encoder_kat.json")["frames"]:
        rc, wav, fok, ferr = ctx.decode_stream(np.array(f["expected"], dtype=np.uint8))
        assert (rc, fok, ferr) == (0, 1, 0)
        assert wav.tolist() == f["wav"]


def test_golden_decode_blocks(ctx, x3):
    for b in load("decoder_kat.json")["blocks"]:
        x = np.array(b["x3_inp"], dtype=np.uint8)
        if b["first_sample_in_stream"]:
            payload = x
        else:  # re-pack: 16-bit predecessor + the bits after the skipped prefix
            bits = np.unpackbits(x)[b["skip_bits"]:]
            first = np.unpackbits(np.array([b["last_wav"]], dtype=">i2").view(np.uint8))
            payload = np.packbits(np.concatenate([first, bits]))
        n = len(b["expected_wav"])
        payload = np.concatenate([payload, np.zeros(8, dtype=np.uint8)])
        rc, wav = ctx.decode_frame(payload, 1 + n)
        assert rc == 0, (b["name"], rc)
        assert wav[1:].tolist() == b["expected_wav"], b["name"]
        rc_o, wav_o = O.decode_frame(payload, 1 + n)
        assert rc_o == 0 and np.array_equal(wav, wav_o)


def test_golden_crc(ctx, x3):
    for c in load("crc_kat.json")["cases"]:
        assert ctx.crc16(bytes(c["bytes"])) == c["crc"]


def test_crc_lengths(ctx, x3):
    rng = np.random.default_rng(7)
    big = rng.integers(0, 256, size=300001, dtype=np.uint8)
    for n in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 255, 256, 257, 4095, 4096, 4097, 32767, 32768, 32769,
              65535, 65536, 65537, 100000, 300001]:
        assert ctx.crc16(big[:n]) == O.crc16(big[:n]), n
    zeros = np.zeros(70000, dtype=np.uint8)
    assert ctx.crc16(zeros) == O.crc16(zeros)


# ------------------------------------------------------------------ encode parity on synthetic signals

SIZES = [1, 2, 3, 20, 21, 22, 41, 9999, 10000, 10001, 10002, 16000, 20000, 123457]


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_encode_parity_kinds(ctx, x3, kind):
    for n in SIZES:
        wav = x3.synth(kind, 0x58330000 + kind, 1000, n)
        check_encode(ctx, x3, wav)


def test_encode_start_pos(ctx, x3):
    wav = x3.synth(2, 11, 0, 25000)
    for sp in [0, 1, 2, 3, 17, 100, 101]:
        check_encode(ctx, x3, wav, start_pos=sp)


def test_encode_threshold_edges(ctx, x3):
    """max|diff| exactly at 3/4, 8/9, 20/21, 16383/16384, +-32767 swings."""
    rng = np.random.default_rng(3)
    chunks = []
    for m in [0, 1, 3, 4, 8, 9, 20, 21, 31, 32, 63, 64, 16383, 16384, 16385, 32767, 40000, 65535]:
        for sign in (1, -1):
            d = rng.integers(-min(m, 20000), min(m, 20000) + 1, size=40)
            d[rng.integers(0, 40)] = sign * m
            chunks.append(d)
    diffs = np.concatenate(chunks).astype(np.int64)
    wav = np.zeros(diffs.size, dtype=np.int64)
    acc = 0
    for i, d in enumerate(diffs):  # keep the walk inside i16 by reflecting
        nxt = acc + d
        if nxt > 32767 or nxt < -32768:
            nxt = acc - d
        if nxt > 32767 or nxt < -32768:
            nxt = acc
        acc = nxt
        wav[i] = acc
    wav = wav.astype(np.int16)
    check_encode(ctx, x3, wav)
    alt = np.tile(np.array([32767, -32768], dtype=np.int16), 5000)
    check_encode(ctx, x3, alt)
    for p in [x3.Params.make(20, 5), x3.Params.make(20, 1)]:
        check_encode(ctx, x3, wav, p)


@pytest.mark.parametrize("bl,bpf", [(1, 10), (2, 33), (7, 100), (19, 500), (20, 500), (20, 1024), (33, 64),
                                    (60, 100), (60, 600), (5, 2000)])
def test_encode_generic_geometry(ctx, x3, bl, bpf):
    p = x3.Params.make(bl, bpf)
    for kind in (2, 1, 4):
        wav = x3.synth(kind, 99 + bl, 0, 3 * bl * bpf + 17)
        out = check_encode(ctx, x3, wav, p)
        check_decode(ctx, x3, out, p)


@pytest.mark.parametrize("bpf", [2, 6, 50, 100, 256, 448, 512, 1, 3, 51, 255, 499, 501, 511])
def test_short_frames_take_the_single_pass_encoder(ctx, x3, bpf):
    """Frames of any number of blocks of 20 up to 512 (x3.rs:81-113 lets blocks_per_frame be anything) go through the
    single-pass encoders -- an odd number puts every other frame on an 8-byte boundary --: both generations against
    the oracle, ragged last frame included, and the decoders on what they wrote."""
    p = x3.Params.make(20, bpf)
    n = 20 * bpf * 37 + 20 * bpf // 2 + 3
    for kind in (2, 1, 4):
        wav = x3.synth(kind, 4100 + bpf, 0, n)
        out = check_encode(ctx, x3, wav, p)
        # (noise in frames of 5 120 samples and more is dense content: the context moves to the second generation)
        assert ctx.get_option("enc_gen_in_use") in ((3,) if kind == 2 else (3, 2)), ctx.get_option("enc_gen_in_use")
        check_decode(ctx, x3, out, p)
        with _opt(ctx, enc_gen=2):
            check_encode(ctx, x3, wav, p)
            assert ctx.get_option("enc_gen_in_use") == 2
        ctx.set_option("enc_gen", 3)
        check_encode(ctx, x3, wav, p, start_pos=2)


@pytest.mark.parametrize("shift", [4, 8, 12, 2])
def test_encode_dev_from_a_buffer_that_is_not_16_byte_aligned(ctx, x3, shift):
    """x3_encode_dev on samples that begin 4, 8 or 12 bytes into a 16-byte unit (a slice of somebody's tensor): still the
    single-pass encoder (its loads are dword buffer loads); 2 bytes in: the general kernel.  Same stream either way."""
    p = x3.Params.default()
    n = 10000 * 70 + 4321
    lib = x3.lib()
    F = lib.x3_num_frames(n, C.byref(p)); cap = lib.x3_encode_bound(n, C.byref(p))
    d_buf = ctx.alloc(2 * n + 64); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1))
    try:
        for kind in (2, 1):
            wav = x3.synth(kind, 6100 + shift, 0, n)
            rc_o, out_o, st_o = O.encode(wav, oparams(p))
            assert rc_o == 0
            ctx.upload(d_buf + shift, wav)
            assert ctx.encode_dev(d_buf + shift, n, p, d_out, cap, 0, d_off) == 0
            rc, pos, st = ctx.encode_result()
            assert rc == 0 and pos == out_o.size and list(st) == st_o.tolist()
            assert ctx.get_option("enc_gen_in_use") == (1 if shift == 2 else (3 if kind == 2 else ctx.get_option("enc_gen_in_use")))
            assert np.array_equal(ctx.download(d_out, pos), out_o)
            offs = ctx.download(d_off, 8 * (F + 1), np.uint64)
            assert offs[F] == pos and offs[0] == 0
        ctx.set_option("enc_gen", 3)
    finally:
        for d in (d_buf, d_out, d_off):
            ctx.free(d)


@pytest.mark.parametrize("bpf", [2, 6, 50, 250, 502, 4, 100, 1, 3, 51, 501])
def test_decode_whole_groups_of_other_frame_lengths(ctx, x3, bpf):
    """The split decoder's whole-line flush on frames that are 8 (mod 16) samples long (an even number of blocks that is
    not a multiple of four: eight row phases against the 128-byte lines instead of four) and, for comparison, on
    multiples of sixteen: three full groups of 64 frames, a ragged fourth and a short last frame, decoded into a buffer
    that starts on a 16-byte boundary but not on a line -- and on an 8-byte one; an odd number of blocks puts every
    other row on an 8-byte boundary anyway (the flusher's 8-byte pieces)."""
    p = x3.Params.make(20, bpf)
    spf = 20 * bpf
    n = spf * (3 * 64 + 17) + spf // 2 + 5
    for kind in (2, 1):
        wav = x3.synth(kind, 5200 + bpf, 0, n)
        out = check_encode(ctx, x3, wav, p)
        check_decode(ctx, x3, out, p)
    # the device API with an output that begins 16, 48, 80 and 8, 72 bytes into a line
    wav = x3.synth(2, 5300 + bpf, 0, n)
    F = (n + spf - 1) // spf
    lib = x3.lib()
    cap = lib.x3_encode_bound(n, C.byref(p))
    d_wav = ctx.alloc(2 * n); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n + 256)
    try:
        ctx.upload(d_wav, wav)
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        assert ctx.encode_result()[0] == 0
        for shift in (16, 48, 80, 8, 72):
            ctx.upload(d_back, np.zeros(n + 128, dtype=np.int16))
            assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back + shift, n, n_per_clip=n) == 0
            r = ctx.decode_result()
            assert r[:3] == (0, F, 0), r
            back = ctx.download(d_back, 2 * n + 256, np.int16)
            assert np.array_equal(back[shift // 2: shift // 2 + n], wav), (bpf, shift)
            assert not back[:shift // 2].any() and not back[shift // 2 + n:].any(), (bpf, shift)
    finally:
        for d in (d_wav, d_out, d_off, d_back):
            ctx.free(d)


@pytest.mark.parametrize("bpf", [500, 502, 504, 510])
def test_encode_largest_single_pass_frames(ctx, x3, bpf):
    """The longest payloads the single-pass encoder sees: full-scale noise (every block a literal) in frames of up
    to 10 200 samples -- 20.8 KB, eleven payload dwords per lane in its CRC pass -- at both stream alignments."""
    p = x3.Params.make(20, bpf)
    rng = np.random.default_rng(bpf)
    n = 20 * bpf * 6 + 1234
    wav = rng.integers(-32768, 32768, size=n).astype(np.int16)
    wav[20 * bpf * 2: 20 * bpf * 3] >>= 9       # one frame of short codes: the chunk size changes and changes back
    for sp in (0, 2):
        out = check_encode(ctx, x3, wav, p, start_pos=sp)
    check_decode(ctx, x3, out[2:], p)
    with _opt(ctx, enc_gen=2):                   # the second-generation single-pass kernel has the same pass
        check_encode(ctx, x3, wav, p)
    ctx.set_option("enc_gen", 3)


@pytest.mark.parametrize("codes,thr", [((0, 1, 3), (3, 8, 20)), ((0, 1, 2), (3, 8, 18)), ((1, 2, 3), (5, 10, 25)),
                                       ((0, 2, 3), (2, 12, 27)), ((3, 3, 3), (1, 2, 3)), ((0, 0, 0), (1, 2, 5)),
                                       ((0, 1, 3), (6, 10, 27)), ((0, 1, 3), (0, 0, 0)), ((0, 1, 3), (8, 3, 20))])
def test_encode_generic_codes(ctx, x3, codes, thr):
    p = x3.Params.make(20, 500, codes, thr)
    for kind in (2, 4):
        wav = x3.synth(kind, 5, 0, 30011)
        check_encode(ctx, x3, wav, p)


def test_encode_out_of_table_is_bad_arg(ctx, x3):
    """thresholds above a Rice table's range: the reference indexes out of bounds (panics)."""
    p = x3.Params.make(20, 500, (0, 1, 3), (3, 8, 40))   # Rice3 table covers -28..27 only
    wav = x3.synth(2, 5, 0, 50000)
    rc_o, _, _ = O.encode(wav, oparams(p))
    rc_g, _, _ = ctx.encode(wav, p)
    assert rc_g == rc_o == x3.ERR_BAD_ARG


def test_encode_errors(ctx, x3):
    wav = x3.synth(2, 1, 0, 30000)
    rc, _, _ = ctx.encode(wav, n_channels=2)
    assert rc == x3.ERR_MORE_THAN_ONE_CHANNEL
    rc, _, _ = ctx.encode(wav, n_channels=0)
    assert rc == x3.ERR_BAD_ARG
    full = O.encode(wav)[1]
    for cap in [0, 10, 20, full.size - 1000, full.size - 2, full.size - 1]:
        rc, _, _ = ctx.encode(wav, cap=cap)
        assert rc == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY, cap
    rc, out, _ = ctx.encode(wav, cap=full.size)
    assert rc == 0 and np.array_equal(out, full)
    # empty input / zero-sized frames write nothing
    rc, out, _ = ctx.encode(np.zeros(0, dtype=np.int16))
    assert rc == 0 and out.size == 0
    rc, out, _ = ctx.encode(wav, x3.Params.make(0, 500))
    assert rc == 0 and out.size == 0
    assert x3.lib().x3_params_validate(C.byref(x3.Params.make(20, 500, (0, 1, 3), (7, 8, 20)))) == x3.ERR_INVALID_ENCODING_THRESH
    assert x3.lib().x3_params_validate(C.byref(x3.Params.make(20, 500, (0, 1, 3), (3, 8, 99)))) == 0
    assert x3.lib().x3_params_validate(C.byref(x3.Params.make(20, 500, (0, 4, 3), (3, 8, 20)))) == x3.ERR_BAD_ARG


def test_insufficient_memory_matches_oracle(ctx, x3):
    """ByteWriterInsufficientMemory keeps the reference's prefix guarantee (bytewriter.rs:86-99, encoder.rs:67-73): every
    frame that fits is in the slice, complete and in place, *out_pos is the end of the last of them, nothing behind it is
    touched -- in one piece and through the chunked pipeline, at even and odd start positions"""
    wav = x3.synth(2, 1, 0, 30000)
    L = x3.lib()
    p = x3.Params.default()
    try:
        for start in (0, 7):
            full = O.encode(wav, start_pos=start)[1]
            ends = [start + (start & 1)]
            while ends[-1] < full.size:
                ends.append(ends[-1] + 20 + ((int(full[ends[-1] + 6]) << 8) | int(full[ends[-1] + 7])))
            assert ends[-1] == full.size and len(ends) == 4
            for cap in [start, start + 1, start + 19, start + 21, 5000, ends[1], ends[1] + 1, ends[2] - 1, ends[2] + 20, full.size - 2]:
                rc_o, got_o, _ = O.encode(wav, cap=cap, start_pos=start)
                assert rc_o == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY
                want_end = max([e for e in ends if e <= cap] + [start])
                for frames in (-1, 1):
                    ctx.set_option("host_chunk_frames", frames)
                    out = np.full(max(cap, 1) + 64, 0xAA, dtype=np.uint8)
                    pos = C.c_uint64(0)
                    rc = L.x3_encode(ctx._h, wav.ctypes.data, wav.size, 1, C.byref(p), out.ctypes.data, cap, start, C.byref(pos), None)
                    assert rc == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY, (start, cap, rc)
                    assert pos.value == want_end, (start, cap, frames, pos.value, want_end)
                    assert np.array_equal(out[start:want_end], full[start:want_end]), (start, cap, frames)
                    # ... which is what the reference's slice holds there (it goes on into the frame that does not fit)
                    assert got_o.size >= want_end and np.array_equal(out[start:want_end], got_o[start:want_end])
                    assert np.all(out[:start] == 0xAA) and np.all(out[want_end:] == 0xAA), (start, cap, frames)
                    assert ctx.get_option("encode_needed_pos") == full.size
    finally:
        ctx.set_option("host_chunk_frames", 0)


def test_encode_host_buffers_in_chunks(ctx, x3):
    """x3_encode takes a long host buffer in chunks of whole frames, downloads beside uploads (option
    host_chunk_frames; by default only from 64 MB of samples): same bytes, position, statistics and errors as in one
    piece and as the oracle, for chunk sizes that do and do not divide the stream, odd start positions, short tails,
    dense content (the wave encoder's re-run inside a chunk) and an output that ends inside a later chunk"""
    rng = np.random.default_rng(77)
    try:
        for (bl, bpf), n, frames in [((20, 500), 1_234_567, 8), ((20, 500), 400_000, 16), ((20, 25), 100_003, 24),
                                      ((7, 33), 55_555, 8), ((20, 500), 160_000, 8), ((20, 500), 170_001, 8)]:
            p = x3.Params.make(bl, bpf, (0, 1, 3), (3, 8, 20))
            op = O.Params.make(bl, bpf, (0, 1, 3), (3, 8, 20))
            wav = x3.synth(2, 5, 0, n)
            wav[n // 3: n // 3 + 30_000] = rng.integers(-32768, 32768, 30_000)  # dense frames in the middle
            for start in (0, 7):
                want = O.encode(wav, op, start_pos=start)
                ctx.set_option("host_chunk_frames", -1)
                one = ctx.encode(wav, p, start_pos=start)
                ctx.set_option("host_chunk_frames", frames)
                got = ctx.encode(wav, p, start_pos=start)
                assert want[0] == 0 and one[0] == 0 and got[0] == 0
                assert np.array_equal(got[1][start:], want[1][start:]) and np.array_equal(one[1][start:], want[1][start:])
                assert np.array_equal(got[2], want[2]) and np.array_equal(one[2], want[2])
            # the output ends inside the third chunk: the same status and position as in one piece
            full = O.encode(wav, op)[1].size
            for cap in (full - 2, full * 2 // 3, 21):
                ctx.set_option("host_chunk_frames", -1)
                one = ctx.encode(wav, p, cap=cap); pos_one = ctx.out_pos
                ctx.set_option("host_chunk_frames", frames)
                got = ctx.encode(wav, p, cap=cap)
                assert got[0] == one[0] == O.encode(wav, op, cap=cap)[0] == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY
                assert ctx.out_pos == pos_one
    finally:
        ctx.set_option("host_chunk_frames", 0)


# ------------------------------------------------------------------ decode parity incl. corrupt streams

def refresh_crcs(x3, stream, off):
    """recompute payload CRC + header CRC of the frame at `off` after tampering"""
    rc, h = x3.read_frame_header(stream[off:off + 20])
    plen = int(stream[off + 6]) << 8 | int(stream[off + 7])
    pcrc = O.crc16(stream[off + 20: off + 20 + plen])
    stream[off + 18] = pcrc >> 8
    stream[off + 19] = pcrc & 0xFF
    hcrc = O.crc16(stream[off: off + 16])
    stream[off + 16] = hcrc >> 8
    stream[off + 17] = hcrc & 0xFF


def frame_offsets(stream):
    offs, pos = [], 0
    while pos + 20 < stream.size:
        offs.append(pos)
        pos += 20 + (int(stream[pos + 6]) << 8 | int(stream[pos + 7]))
    return offs


def test_decode_roundtrip_kinds(ctx, x3):
    for kind in range(5):
        wav = x3.synth(kind, 77, 0, 54321)
        rc, stream, _ = O.encode(wav)
        r = check_decode(ctx, x3, stream, wav_cap=wav.size)
        assert r[0] == 0 and np.array_equal(r[1], wav)


def test_decode_corrupt_payload_crc(ctx, x3):
    wav = x3.synth(2, 78, 0, 60000)
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    for fi in (0, 2, len(offs) - 1):
        s = stream.copy()
        s[offs[fi] + 20 + 100] ^= 0x10
        r = check_decode(ctx, x3, s, wav_cap=wav.size)
        assert r[0] == x3.ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC and r[2] == fi


def test_decode_corrupt_payload_with_valid_crc(ctx, x3):
    """tamper with payload bits, then fix both CRCs: the block decoders must agree on what they make of it
    (wrong samples, OutOfBoundsInverse, or InvalidBPF)."""
    rng = np.random.default_rng(5)
    wav = x3.synth(2, 79, 0, 40000)
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    seen = set()
    for trial in range(60):
        s = stream.copy()
        fi = int(rng.integers(0, len(offs)))
        plen = int(s[offs[fi] + 6]) << 8 | int(s[offs[fi] + 7])
        kind = trial % 3
        pos = offs[fi] + 22 + int(rng.integers(0, plen - 12))
        if kind == 0:
            s[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            s[pos:pos + 8] = 0          # long zero run -> OutOfBoundsInverse or BFP with E <= 5
        else:
            s[pos:pos + 6] = rng.integers(0, 256, size=6, dtype=np.uint8)
        refresh_crcs(x3, s, offs[fi])
        r = check_decode(ctx, x3, s, wav_cap=wav.size + 70000)
        seen.add((r[0], r[3]))
    assert (0, 1) in seen  # at least one counted frame error was exercised


def test_decode_header_errors(ctx, x3):
    wav = x3.synth(2, 80, 0, 35000)
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    o2 = offs[2]
    # header CRC
    s = stream.copy(); s[o2 + 5] ^= 1
    assert check_decode(ctx, x3, s, wav_cap=wav.size)[0] == x3.ERR_FRAME_HEADER_INVALID_HEADER_CRC
    # key (with a valid header CRC)
    s = stream.copy(); s[o2] = 0x79; refresh_crcs(x3, s, o2)
    assert check_decode(ctx, x3, s, wav_cap=wav.size)[0] == x3.ERR_FRAME_HEADER_INVALID_KEY
    # channels > 1
    s = stream.copy(); s[o2 + 3] = 2; refresh_crcs(x3, s, o2)
    assert check_decode(ctx, x3, s, wav_cap=wav.size)[0] == x3.ERR_MORE_THAN_ONE_CHANNEL
    # payload_len >= Frame::MAX_LENGTH
    s = stream.copy(); s[o2 + 6] = 0x7F; s[o2 + 7] = 0xE0; refresh_crcs(x3, s, o2)
    assert check_decode(ctx, x3, s, wav_cap=wav.size)[0] == x3.ERR_FRAME_LENGTH
    # payload_len beyond the data: quiet end
    s = stream.copy(); s[o2 + 6] = 0x70; hc = O.crc16(s[o2:o2 + 16]); s[o2 + 16] = hc >> 8; s[o2 + 17] = hc & 0xFF
    r = check_decode(ctx, x3, s, wav_cap=wav.size)
    assert r[0] == 0 and r[2] == 2
    # truncated streams
    for cut in [0, 1, 19, 20, 21, offs[1] - 1, offs[1], offs[1] + 20, offs[1] + 21, stream.size - 1, stream.size - 2]:
        check_decode(ctx, x3, stream[:cut].copy(), wav_cap=wav.size)
    # trailing garbage shorter than a header is ignored
    check_decode(ctx, x3, np.concatenate([stream, np.zeros(20, dtype=np.uint8)]), wav_cap=wav.size)
    # samples field larger than what the payload encodes: reads zeros past the end
    s = stream.copy(); s[offs[-1] + 4] = 0x27; s[offs[-1] + 5] = 0x10; refresh_crcs(x3, s, offs[-1])
    check_decode(ctx, x3, s, wav_cap=wav.size + 20000)


def test_decode_wav_cap_too_small(ctx, x3):
    wav = x3.synth(2, 81, 0, 35000)
    stream = O.encode(wav)[1]
    for cap in [0, 1, 9999, 10000, 10001, 34999]:
        check_decode(ctx, x3, stream, wav_cap=cap)


# ------------------------------------------------------------------ batch + device-resident API

def test_encode_batch(ctx, x3):
    clips = [x3.synth(2, 1000 + i, 0, 57600) for i in range(7)]
    rc, out, offs, stats = ctx.encode_batch(clips)
    assert rc == 0
    tot = np.zeros(6, dtype=np.uint64)
    for i, c in enumerate(clips):
        rc_o, o, st = O.encode(c)
        assert np.array_equal(out[offs[i]:offs[i + 1]], o)
        tot += st
    assert stats.tolist() == tot.tolist()
    # a uniform batch whose clip length is odd: side by side on the device at a padded stride, the wave encoder all the same
    odd = [x3.synth(2, 1500 + i, 0, 57601) for i in range(70)]
    rc, out, offs, stats = ctx.encode_batch(odd)
    assert rc == 0 and ctx.get_option("enc_gen_in_use") == 3
    for i, c in enumerate(odd):
        assert np.array_equal(out[offs[i]:offs[i + 1]], O.encode(c)[1]), i
    ragged = [x3.synth(4, 2000 + i, 0, n) for i, n in enumerate([1, 10000, 25001, 3])]
    rc, out, offs, stats = ctx.encode_batch(ragged)
    assert rc == 0
    for i, c in enumerate(ragged):
        assert np.array_equal(out[offs[i]:offs[i + 1]], O.encode(c)[1])


def test_encode_frames_dev_clips_of_different_lengths(ctx, x3):
    """x3_encode_frames_dev: a batch of clips of DIFFERENT lengths as one list of frames (one launch set): each clip's
    bytes equal the oracle's encoding of that clip; clips at even and (second round) at odd sample offsets -- the wave
    encoder and the general kernel --, loud clips (the dense pass) among them, the frames decoded back through the frame
    index and per-frame sample offsets."""
    p = x3.Params.default()
    spf = 10000
    rng = np.random.default_rng(77)
    lens = [1, 7, 9999, 10000, 10001, 25000, 31234, 40000, 57601, 3, 20000, 12345] + [int(v) for v in rng.integers(1, 70000, size=60)]
    lib = x3.lib()
    for odd in (False, True):
        ctx.set_option("enc_gen", 3)   # (forget what earlier calls said about dense content)
        starts, pos = [], (1 if odd else 0)
        clips = []
        for i, n in enumerate(lens):
            starts.append(pos)
            clips.append(x3.synth(1 if i % 9 == 4 else 2, 9000 + i, 0, n))
            pos += n + (int(rng.integers(0, 5)) * 2 if not odd else int(rng.integers(0, 7)))
            if not odd and pos % 2:
                pos += 1
        total = pos + 16
        buf = np.zeros(total, dtype=np.int16)
        so, sn, first = [], [], []
        for s0, c in zip(starts, clips):
            buf[s0:s0 + c.size] = c
            first.append(len(so))
            for k in range(0, c.size, spf):
                so.append(s0 + k)
                sn.append(min(spf, c.size - k))
        F = len(so)
        cap = sum(lib.x3_encode_bound(int(c.size), C.byref(p)) for c in clips) + 64
        d_wav = ctx.alloc(2 * total); d_out = ctx.alloc(cap + 16); d_off = ctx.alloc(8 * (F + 1))
        try:
            ctx.upload(d_wav, buf)
            assert ctx.encode_frames_dev(d_wav, so, sn, p, d_out, cap, 0, d_off) == 0
            rc, end, st = ctx.encode_result()
            assert rc == 0
            assert ctx.get_option("enc_gen_in_use") == (1 if odd else 3)
            offs = ctx.download(d_off, 8 * (F + 1), np.uint64)
            out = ctx.download(d_out, end)
            assert offs[F] == end
            tot = np.zeros(6, dtype=np.uint64)
            for i, c in enumerate(clips):
                rc_o, o, st_o = O.encode(c)
                lo = int(offs[first[i]]); hi = int(offs[first[i + 1]]) if i + 1 < len(clips) else int(end)
                assert np.array_equal(out[lo:hi], o), (odd, i, c.size)
                tot += st_o
            assert list(st) == tot.tolist()
            # too little room for the stream: the reference's ByteWriterInsufficientMemory, and the position it would have needed
            assert ctx.encode_frames_dev(d_wav, so, sn, p, d_out, int(end) - 2, 0, d_off) == 0
            rc2, end2, _ = ctx.encode_result()
            assert rc2 == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY and end2 == end, (rc2, end2, end)
            # one frame, one sample
            assert ctx.encode_frames_dev(d_wav, so[:1], [1], p, d_out, cap, 0, d_off) == 0
            rc3, end3, _ = ctx.encode_result()
            assert rc3 == 0 and np.array_equal(ctx.download(d_out, end3), O.encode(buf[so[0]:so[0] + 1])[1])
            assert ctx.encode_frames_dev(d_wav, so[:1], [spf + 1], p, d_out, cap, 0, d_off) == x3.ERR_BAD_ARG
            assert ctx.encode_frames_dev(d_wav, so[:1], [0], p, d_out, cap, 0, d_off) == x3.ERR_BAD_ARG
            if not odd:
                # back through the frame index, every frame to where it came from (offsets that are multiples of two
                # only: the single-wave decoder; clips padded to multiples of four: option wav_offsets_x4, see below)
                d_wo = ctx.alloc(8 * F); d_back = ctx.alloc(2 * total)
                try:
                    ctx.upload(d_wo, np.array(so, dtype=np.uint64))
                    ctx.upload(d_back, np.zeros(total, dtype=np.int16))
                    assert ctx.decode_dev(d_out, end, d_off, F, p, d_back, total, d_wav_offsets=d_wo) == 0
                    assert ctx.decode_result()[:3] == (0, F, 0)
                    assert np.array_equal(ctx.download(d_back, 2 * total, np.int16), buf)
                finally:
                    ctx.free(d_wo); ctx.free(d_back)
        finally:
            for d in (d_wav, d_out, d_off):
                ctx.free(d)


def test_device_batch_roundtrip(ctx, x3):
    """BASELINE config 5 in miniature: a uniform batch of clips resident in HBM (with padding between
    clips), encoded by one launch set and decoded from the encoder's frame index."""
    n_clips, n_per, stride = 9, 57_603, 57_608
    p = x3.Params.default()
    clips = [x3.synth(2 + (i % 3 == 2) * 2, 500 + i, 0, n_per) for i in range(n_clips)]
    host = np.zeros(n_clips * stride, dtype=np.int16)
    for i, c in enumerate(clips):
        host[i * stride:i * stride + n_per] = c
    d_wav = ctx.alloc(host.nbytes)
    ctx.upload(d_wav, host)
    fpc = (n_per + p.spf - 1) // p.spf
    F = fpc * n_clips
    cap = sum(x3.lib().x3_encode_bound(n_per, C.byref(p)) for _ in range(n_clips)) + 16
    d_out = ctx.alloc(cap)
    d_off = ctx.alloc(8 * (F + 1))
    assert ctx.encode_dev(d_wav, n_per, p, d_out, cap, 0, d_off, n_clips=n_clips, clip_stride=stride) == 0
    rc, pos, stats = ctx.encode_result()
    assert rc == 0
    offs = ctx.download(d_off, 8 * (F + 1), np.uint64)
    stream = ctx.download(d_out, (pos + 3) & ~3)[:pos]
    tot = np.zeros(6, dtype=np.uint64)
    for i, c in enumerate(clips):
        rc_o, o, st = O.encode(c)
        assert np.array_equal(stream[int(offs[i * fpc]):int(offs[(i + 1) * fpc])], o), i
        tot += st
    assert stats.tolist() == tot.tolist()
    d_back = ctx.alloc(host.nbytes)
    ctx.upload(d_back, np.full(host.size, 12345, dtype=np.int16))
    assert ctx.decode_dev(d_out, pos, d_off, F, p, d_back, host.size, n_per_clip=n_per, n_clips=n_clips,
                          clip_stride=stride) == 0
    assert ctx.decode_result() == (0, F, 0, n_per * n_clips)
    back = ctx.download(d_back, host.nbytes, np.int16)
    for i, c in enumerate(clips):
        assert np.array_equal(back[i * stride:i * stride + n_per], c)
        assert (back[i * stride + n_per:(i + 1) * stride] == 12345).all()  # the padding is untouched
    for d in (d_wav, d_out, d_off, d_back):
        ctx.free(d)


def test_synth_host_equals_device(ctx, x3):
    n = 100000
    d = ctx.alloc(2 * n)
    for kind in range(5):
        ctx.synth_dev(kind, 0x5833, 12345, n, d)
        ctx.sync()
        dev = ctx.download(d, 2 * n, np.int16)
        assert np.array_equal(dev, x3.synth(kind, 0x5833, 12345, n))
    ctx.free(d)


def test_device_api_config2(ctx, x3):
    """BASELINE config 2: 10 min @ 44.1 kHz hydrophone noise, encode on the GPU, bit-exact vs CPU."""
    n = 26_460_000
    p = x3.Params.default()
    d_wav = ctx.alloc(2 * n)
    ctx.synth_dev(2, 0x58330002, 0, n, d_wav)
    cap = x3.lib().x3_encode_bound(n, C.byref(p))
    d_out = ctx.alloc(cap)
    F = x3.lib().x3_num_frames(n, C.byref(p))
    d_off = ctx.alloc(8 * (F + 1))
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    rc, pos, stats = ctx.encode_result()
    assert rc == 0
    wav = x3.synth(2, 0x58330002, 0, n)
    rc_o, out_o, st_o = O.encode(wav)
    assert pos == out_o.size
    got = ctx.download(d_out, (pos + 3) & ~3)[:pos]
    assert np.array_equal(got, out_o)
    assert stats.tolist() == st_o.tolist()
    offs = ctx.download(d_off, 8 * (F + 1), np.uint64)
    assert offs[0] == 0 and offs[-1] == pos and frame_offsets(out_o) == offs[:-1].tolist()
    # decode on the device from the encoder's own frame index
    d_back = ctx.alloc(2 * n)
    assert ctx.decode_dev(d_out, pos, d_off, F, p, d_back, n, n_per_clip=n) == 0
    rc, first_bad, st, before = ctx.decode_result()
    assert (rc, first_bad, st, before) == (0, F, 0, n)
    assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
    for d in (d_wav, d_out, d_off, d_back):
        ctx.free(d)


def test_full_size_config3_roundtrip(ctx, x3):
    """BASELINE config 3 at full size (1 h @ 192 kHz = 691.2 M samples): size-independent
    properties -- encode -> decode is the identity, the frame index is a consistent header chain,
    and sampled frames equal the CPU oracle's encoding of the same samples."""
    torch = pytest.importorskip("torch")
    import time
    tt = [time.perf_counter()]

    def lap(what):
        tt.append(time.perf_counter())
        print("  [config 3] %-34s %.2f s" % (what, tt[-1] - tt[-2]), flush=True)
    n = 691_200_000
    p = x3.Params.default()
    F = x3.lib().x3_num_frames(n, C.byref(p))
    dev = torch.device("cuda:0")
    wav = torch.empty(n, dtype=torch.int16, device=dev)
    ctx.synth_dev(2, 0x58330003, 0, n, wav.data_ptr())
    torch.cuda.synchronize(dev)
    lap("signal in HBM")
    cap = int(n * 1.2)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    off = torch.empty(F + 1, dtype=torch.int64, device=dev)
    assert ctx.encode_dev(wav.data_ptr(), n, p, out.data_ptr(), cap, 0, off.data_ptr()) == 0
    rc, pos, stats = ctx.encode_result()
    assert rc == 0 and int(stats.sum()) == n - F
    lap("encode")
    back = torch.zeros(n, dtype=torch.int16, device=dev)
    torch.cuda.synchronize(dev)  # the fill ran on torch's stream, the decoder runs on the context's
    assert ctx.decode_dev(out.data_ptr(), pos, off.data_ptr(), F, p, back.data_ptr(), n, n_per_clip=n) == 0
    rc, first_bad, st, before = ctx.decode_result()
    assert (rc, first_bad, st, before) == (0, F, 0, n)
    lap("decode")
    assert torch.equal(back, wav)
    offs = off.cpu().numpy()
    lap("compare on the device")
    assert offs[0] == 0 and offs[-1] == pos and np.all(np.diff(offs) >= 22) and np.all(offs % 2 == 0)
    # the WHOLE stream against the CPU oracle, all 69 120 frames: chunks of whole frames on a thread pool (the
    # oracle is C behind ctypes, which releases the GIL).  A round trip alone would also pass an encoder and a
    # decoder that are wrong in the same way.
    t0 = time.perf_counter()
    host_wav = wav.cpu().numpy()
    host_out = out[:pos].cpu().numpy()
    t1 = time.perf_counter()
    _compare_stream_with_oracle(host_wav, host_out, offs, 10000, 1, n)
    print("config 3: device -> host %.2f s, oracle compare of %d frames %.2f s on %d cpus (load %s)" % (
        t1 - t0, F, time.perf_counter() - t1, os.cpu_count(), os.getloadavg()))
    # the GPU-side frame walk finds the same 69 120 frames in the bare byte stream
    fo = torch.empty(F + 8, dtype=torch.int64, device=dev)
    wo = torch.empty(F + 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    rc, nf, ns, term = ctx.index_dev(out.data_ptr(), pos, F + 8, fo.data_ptr(), wo.data_ptr())
    dt = time.perf_counter() - t0
    assert (rc, nf, ns, term) == (0, F, n, 0)
    assert torch.equal(fo[:F], off[:F]) and torch.equal(wo[:F], torch.arange(F, device=dev, dtype=torch.int64) * 10000)
    print("x3_index_dev: %d frames in %.0f MB indexed in %.2f ms" % (F, pos / 1e6, dt * 1e3))
    back.zero_()
    torch.cuda.synchronize(dev)
    assert ctx.decode_stream_dev(out.data_ptr(), pos, p, back.data_ptr(), n) == (0, n, F, 0)
    assert torch.equal(back, wav)
    lap("index + foreign-stream decode")


def _compare_stream_with_oracle(host_wav, host_out, offs, spf, n_clips, n_per_clip, clip_stride=None, chunk_frames=256,
                                clips=None):
    """every frame of the device-encoded stream == the oracle's encoding of the same samples.  offs: F+1 byte
    offsets; frames never cross clips, so a chunk of whole frames of one clip encodes to a contiguous byte range."""
    import concurrent.futures as cf
    clip_stride = n_per_clip if clip_stride is None else clip_stride
    fpc = (n_per_clip + spf - 1) // spf
    jobs = []
    for c in (range(n_clips) if clips is None else clips):
        for f0 in range(0, fpc, chunk_frames):
            jobs.append((c, f0, min(fpc, f0 + chunk_frames)))

    def work(job):
        c, f0, f1 = job
        a = c * clip_stride + f0 * spf
        b = c * clip_stride + min(n_per_clip, f1 * spf)
        rc, enc, _ = O.encode(host_wav[a:b])
        lo, hi = int(offs[c * fpc + f0]), int(offs[c * fpc + f1])
        return rc == 0 and enc.size == hi - lo and np.array_equal(enc, host_out[lo:hi]), job

    bad = []
    with cf.ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        for ok, job in ex.map(work, jobs):
            if not ok:
                bad.append(job)
    assert not bad, "frames differ from the CPU oracle in (clip, first frame, last frame): %s" % bad[:5]
    return len(jobs)


def test_full_size_config5_batch(ctx, x3):
    """BASELINE config 5 at FULL size: 1000 clips x 1 min @ 96 kHz = 5.76 G samples (> 2^32: 64-bit sample and byte
    indexing), 576 000 frames, encoded by one launch set and decoded from the encoder's frame index, all in HBM.
    Size-independent properties over everything (identity, header chain, statistics) and the CPU oracle on whole
    sampled clips -- the first ones, the last one (sample indices beyond 2^32) and a spread in between."""
    torch = pytest.importorskip("torch")
    n_clips, n_per = 1000, 5_760_000
    n = n_clips * n_per
    p = x3.Params.default()
    fpc = n_per // 10000
    F = fpc * n_clips
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 34 * (1 << 30):
        pytest.skip("needs ~30 GB of HBM")
    wav = torch.empty(n, dtype=torch.int16, device=dev)
    for c in range(n_clips):
        ctx.synth_dev(2 if c % 7 else 4, 0x58330005 + c, 0, n_per, wav.data_ptr() + 2 * c * n_per)
    ctx.sync()
    cap = int(n * 0.75)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    off = torch.empty(F + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    assert ctx.encode_dev(wav.data_ptr(), n_per, p, out.data_ptr(), cap, 0, off.data_ptr(), n_clips=n_clips) == 0
    rc, pos, stats = ctx.encode_result()
    assert rc == 0 and int(stats.sum()) == n - F, (rc, pos, ctx.last_error())
    assert ctx.get_option("encode_fallbacks") == 0
    back = torch.zeros(n, dtype=torch.int16, device=dev)
    torch.cuda.synchronize(dev)
    assert ctx.decode_dev(out.data_ptr(), pos, off.data_ptr(), F, p, back.data_ptr(), n, n_per_clip=n_per,
                          n_clips=n_clips) == 0
    assert ctx.decode_result() == (0, F, 0, n)
    assert torch.equal(back, wav)
    del back
    # header chain: every frame's header says what the index says
    offs = off.cpu().numpy()
    d = np.diff(offs)
    assert offs[0] == 0 and offs[-1] == pos and np.all(d >= 22) and np.all(offs % 2 == 0)
    o_t = off[:F]
    hdr = torch.stack([out[o_t + k] for k in (0, 1, 4, 5, 6, 7)], dim=1).cpu().numpy().astype(np.int64)
    assert np.all(hdr[:, 0] == 0x78) and np.all(hdr[:, 1] == 0x33)
    assert np.all(hdr[:, 2] * 256 + hdr[:, 3] == 10000)
    assert np.array_equal(hdr[:, 4] * 256 + hdr[:, 5], d - 20)
    # the oracle on whole clips: 24 of them, the last one included
    clips = sorted(set([0, 1, 2, 6, 7, 499, 500, 998, 999] + list(range(13, 1000, 67))))
    assert len(clips) >= 20 and clips[-1] == n_clips - 1
    host_wav = np.empty(n, dtype=np.int16)      # only the sampled clips are filled in
    host_out = np.empty(pos, dtype=np.uint8)
    for c in clips:
        host_wav[c * n_per:(c + 1) * n_per] = wav[c * n_per:(c + 1) * n_per].cpu().numpy()
        lo, hi = int(offs[c * fpc]), int(offs[(c + 1) * fpc])
        host_out[lo:hi] = out[lo:hi].cpu().numpy()
    assert (clips[-1] + 1) * n_per > 2 ** 32
    jobs = _compare_stream_with_oracle(host_wav, host_out, offs, 10000, n_clips, n_per, clips=clips, chunk_frames=96)
    assert jobs == len(clips) * 6
    # the GPU-side frame walk over the 3 GB stream finds the same 576 000 frames
    fo = torch.empty(F + 8, dtype=torch.int64, device=dev)
    wo = torch.empty(F + 8, dtype=torch.int64, device=dev)
    rc, nf, ns, term = ctx.index_dev(out.data_ptr(), pos, F + 8, fo.data_ptr(), wo.data_ptr())
    assert (rc, nf, ns, term) == (0, F, n, 0)
    assert torch.equal(fo[:F], off[:F])


class _opt:
    """set an option of the context (x3_ctx_set_option) for a with-block"""

    def __init__(self, ctx, **kv):
        self.ctx, self.kv = ctx, kv

    def __enter__(self):
        self.old = {k: self.ctx.get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            self.ctx.set_option(k, v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            self.ctx.set_option(k, v)


def test_decoder_kernels_agree(ctx, x3):
    """the two-wave decoder (default for the plain geometry) and the single-wave kernel it falls back to
    must make the same of good and of tampered streams: status, frame counts, samples"""
    rng = np.random.default_rng(11)
    wav = np.concatenate([x3.synth(k, 90 + k, 0, 30011 + 977 * k) for k in range(5)])
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    cases = [stream]
    for trial in range(24):
        s = stream.copy()
        fi = int(rng.integers(0, len(offs)))
        plen = int(s[offs[fi] + 6]) << 8 | int(s[offs[fi] + 7])
        pos = offs[fi] + 22 + int(rng.integers(0, plen - 12))
        if trial % 3 == 0:
            s[pos] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            s[pos:pos + 8] = 0
        else:
            s[pos:pos + 6] = rng.integers(0, 256, size=6, dtype=np.uint8)
        refresh_crcs(x3, s, offs[fi])
        cases.append(s)
    for s in cases:
        a = ctx.decode_stream(s, x3.Params.default(), wav_cap=wav.size + 70000)
        with _opt(ctx, decode_single=1):
            b = ctx.decode_stream(s, x3.Params.default(), wav_cap=wav.size + 70000)
        assert a[0] == b[0] and a[2:] == b[2:] and np.array_equal(a[1], b[1])
        o = O.decode_stream(s, O.Params.default(), wav_cap=wav.size + 70000)
        assert a[0] == o[0] and a[2:] == o[2:] and np.array_equal(a[1], o[1])


def test_encoder_kernels_agree(ctx, x3):
    """single-pass persistent encoder (default) vs the two-pass kernels it falls back to"""
    p = x3.Params.default()
    for kind, n in ((2, 1_234_567), (4, 10_000 * 700 + 1), (0, 20_001)):
        wav = x3.synth(kind, 120 + kind, 0, n)
        a = ctx.encode(wav, p)
        with _opt(ctx, two_pass=1):
            b = ctx.encode(wav, p)
        assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]) and a[2].tolist() == b[2].tolist()


# ------------------------------------------------------------------ GPU-side frame index

def _dev_stream(ctx, stream):
    d = ctx.alloc(stream.size + 64)
    ctx.upload(d, np.concatenate([stream, np.zeros(64, dtype=np.uint8)]))
    return d


def test_index_dev_matches_host_walk(ctx, x3):
    wav = np.concatenate([x3.synth(k, 300 + k, 0, 20000 + 1111 * k) for k in range(5)])
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    # a few short frames in front: many frames per kilobyte, sample offsets that are not multiples of anything
    small = np.concatenate([O.encode(x3.synth(4, 400 + i, 0, 1 + 7 * i))[1] for i in range(40)])
    # every stream through BOTH walks: the fast path for clean chains (round 4) with the general walk behind it, and the
    # general walk alone (option index_no_fast)
    for no_fast, s in [(nf_, s_) for nf_ in (0, 1) for s_ in (stream, np.concatenate([small, stream]), small[:-1],
                                                              stream[:offs[3] + 7], stream[:offs[3] + 21])]:
        ctx.set_option("index_no_fast", no_fast)
        d = _dev_stream(ctx, s)
        d_fo = ctx.alloc(8 * 4096)
        d_wo = ctx.alloc(8 * 4096)
        rc, nf, ns, term = ctx.index_dev(d, s.size, 4000, d_fo, d_wo)
        ctx.set_option("index_no_fast", 0)
        assert rc == 0
        # the host walk on the same bytes
        exp_off, exp_wo, pos, nsamp = [], [], 0, 0
        while s.size - pos > 20:
            rch, h = x3.read_frame_header(s[pos:pos + 20])
            if rch or s.size - pos - 20 < h.payload_len:
                break
            exp_off.append(pos); exp_wo.append(nsamp)
            nsamp += h.samples
            pos += 20 + h.payload_len
        assert nf == len(exp_off) and ns == nsamp
        assert ctx.download(d_fo, 8 * nf, np.uint64).tolist() == exp_off
        assert ctx.download(d_wo, 8 * nf, np.uint64).tolist() == exp_wo
        for x in (d, d_fo, d_wo):
            ctx.free(x)


def test_decode_stream_dev_matches_host_api(ctx, x3):
    """x3_decode_stream_dev (GPU index + decode, device buffers) == x3_decode_stream == oracle on good streams,
    truncations, trailing bytes, broken headers and tampered payloads"""
    rng = np.random.default_rng(21)
    wav = np.concatenate([x3.synth(k, 500 + k, 0, 30011 + 977 * k) for k in range(5)])
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    cases = [stream, stream[:-1], stream[:-13], stream[:offs[5] + 20], stream[:offs[5] + 19], stream[:19],
             np.concatenate([stream, np.zeros(12, dtype=np.uint8)]), np.concatenate([stream, np.zeros(40, dtype=np.uint8)]),
             np.concatenate([stream, stream[:offs[2]]])]
    for fi, byte in ((0, 0), (3, 1), (7, 2), (9, 5), (11, 16), (len(offs) - 1, 18)):
        s = stream.copy(); s[offs[fi] + byte] ^= 0x41; cases.append(s)
    for trial in range(9):
        s = stream.copy()
        fi = int(rng.integers(0, len(offs)))
        plen = int(s[offs[fi] + 6]) << 8 | int(s[offs[fi] + 7])
        pos = offs[fi] + 22 + int(rng.integers(0, plen - 12))
        if trial % 3 == 0:
            s[pos] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            s[pos:pos + 8] = 0
        else:
            s[pos:pos + 6] = rng.integers(0, 256, size=6, dtype=np.uint8)
        refresh_crcs(x3, s, offs[fi])
        cases.append(s)
    p = x3.Params.default()
    for cap, no_fast in ((wav.size + 70000, 0), (55555, 0), (wav.size + 70000, 1)):
        d_wav = ctx.alloc(2 * (wav.size + 70000))
        ctx.set_option("index_no_fast", no_fast)   # (1: the general walk alone; 0: the fast path first)
        for s in cases:
            d = _dev_stream(ctx, s)
            a = ctx.decode_stream(s, p, wav_cap=cap)
            b = ctx.decode_stream_dev(d, s.size, p, d_wav, cap)
            assert (a[0], a[1].size, a[2], a[3]) == b, (a[0], a[1].size, a[2:], b, s.size)
            assert np.array_equal(ctx.download(d_wav, 2 * b[1], np.int16), a[1])
            ctx.free(d)
        ctx.set_option("index_no_fast", 0)
        ctx.free(d_wav)


def test_index_fast_path_serves_clean_chains_only(x3):
    """the walk's fast path (ordered candidates, two scans, one check kernel) takes a stream that is one clean chain from
    offset 0; a valid frame that no frame leads to (a false candidate, here: a frame behind six bytes of junk) or a stream
    that does not start with a frame sends the stream through the general walk -- with the host walk's results either way"""
    c = x3.Context(0)
    try:
        p = x3.Params.default()
        wav = x3.synth(2, 77, 0, 250000)
        stream = O.encode(wav)[1]
        extra = O.encode(x3.synth(4, 78, 0, 5000))[1]
        junk = np.arange(6, dtype=np.uint8) + 1
        d_wav = c.alloc(2 * 400000)
        for s, fast in ((stream, True), (np.concatenate([stream, extra]), True), (np.concatenate([stream, junk, extra]), False),
                        (np.concatenate([junk, stream]), False), (stream[:stream.size - 9], False)):   # (a last frame cut short is no frame the walk steps over: general walk)
            f0, g0 = c.get_option("index_fast_walks"), c.get_option("index_general_walks")
            d = c.alloc(s.size + 64); c.upload(d, s)
            a = c.decode_stream(s, p, wav_cap=400000)
            b = c.decode_stream_dev(d, s.size, p, d_wav, 400000)
            assert (a[0], a[1].size, a[2], a[3]) == b, (a[0], a[1].size, a[2:], b)
            assert np.array_equal(c.download(d_wav, 2 * b[1], np.int16), a[1])
            assert (c.get_option("index_fast_walks") - f0, c.get_option("index_general_walks") - g0) == ((1, 0) if fast else (0, 1)), (s.size, fast)
            c.free(d)
        c.free(d_wav)
    finally:
        c.close()


@pytest.mark.gpu
def test_index_max_frames_equal_to_the_chain_with_a_stray_candidate(x3):
    """a stream of N good frames plus one stray CANDIDATE (a last frame cut short, a frame behind junk) walked with
    max_frames = N.  The fast path used to compare the number of candidates with max_frames and fail the call; it now leaves
    such streams to the general walk, which counts the chain.  Same answers with the fast path on and off."""
    torch = pytest.importorskip("torch")
    c = x3.Context(0)
    try:
        p = x3.Params.default()
        wav = x3.synth(2, 79, 0, 65000)
        stream = O.encode(wav)[1]                   # 7 frames
        hdr_offs = []
        pos = 0
        while pos + 20 <= stream.size:
            hdr_offs.append(pos)
            pos += 20 + ((int(stream[pos + 6]) << 8) | int(stream[pos + 7]))
        assert len(hdr_offs) == 7
        extra = O.encode(x3.synth(4, 80, 0, 5000))[1]
        junk = np.arange(6, dtype=np.uint8) + 1
        cases = [(stream[:hdr_offs[3] + 21], 3),                      # three frames + a header whose payload is cut short
                 (np.concatenate([stream, junk, extra]), 7),          # seven frames, junk, a frame nothing leads to
                 (stream, 7)]                                         # (the clean chain itself, max_frames = its length)
        for s, nfr in cases:
            d = c.alloc(s.size + 64); c.upload(d, s)
            fo = torch.zeros(nfr + 8, dtype=torch.int64, device="cuda:0")
            wo = torch.zeros(nfr + 8, dtype=torch.int64, device="cuda:0")
            res = []
            for no_fast in (0, 1):
                c.set_option("index_no_fast", no_fast)
                r = c.index_dev(d, s.size, nfr, fo.data_ptr(), wo.data_ptr())
                res.append((r, fo[:nfr].cpu().tolist()))
            c.set_option("index_no_fast", 0)
            assert res[0] == res[1], (nfr, res)
            assert res[0][0][0] == 0 and res[0][0][1] == nfr, (nfr, res[0][0])
            assert res[0][1] == hdr_offs[:nfr]
            c.free(d)
    finally:
        c.close()


def test_decode_stream_dev_in_one_trip_and_in_two(x3):
    """round 5: x3_decode_stream_dev enqueues the decode behind the frame walk's fast path and reads both summaries in ONE
    trip (the kernels take the frame count from device memory, the grid covers a frame per KiB of stream).  Same answers as
    the two-trip path (option two_trips) and as the oracle on: a clean stream, a stream of frames shorter than the bound
    allows for (falls back), damaged frames (decode error, CRC), a truncated tail, junk in front."""
    c = x3.Context(0)
    try:
        cases = []
        p = x3.Params.default()
        wav = x3.synth(2, 4711, 0, 123_457)
        stream = O.encode(wav)[1]
        cases.append(("clean", p, stream, True))
        ps = x3.Params.make(20, 8)                      # 160-sample frames of ~110 bytes: more frames than one per KiB
        cases.append(("short frames", ps, O.encode(wav[:40_000], O.Params.make(20, 8))[1], False))
        bad = stream.copy(); bad[5000] ^= 0x10
        cases.append(("crc", p, bad, True))
        offs = []
        pos = 0
        while pos + 20 <= stream.size:
            offs.append(pos); pos += 20 + ((int(stream[pos + 6]) << 8) | int(stream[pos + 7]))
        bad = stream.copy(); q = offs[3] + 20 + 40; bad[q:q + 12] = 0
        pc = O.crc16(bad[offs[3] + 20:offs[4]]); bad[offs[3] + 18], bad[offs[3] + 19] = pc >> 8, pc & 0xFF
        cases.append(("decode error", p, bad, True))
        cases.append(("truncated", p, stream[:offs[5] + 300].copy(), False))
        cases.append(("junk in front", p, np.concatenate([np.arange(6, dtype=np.uint8), stream]), False))
        d_wav = c.alloc(2 * 200_000)
        for name, pp, s, one in cases:
            op = O.Params.make(pp.block_len, pp.blocks_per_frame)
            want = O.decode_stream(s, op, wav_cap=200_000)
            d = c.alloc(s.size + 64); c.upload(d, s)
            got = []
            for two in (0, 1):
                c.set_option("two_trips", two)
                n1 = c.get_option("stream_one_trip")
                c.upload(d_wav, np.zeros(200_000, dtype=np.int16))
                r = c.decode_stream_dev(d, s.size, pp, d_wav, 200_000)
                got.append((r, c.download(d_wav, 2 * r[1], np.int16)))
                assert c.get_option("stream_one_trip") - n1 == (1 if (one and not two) else 0), (name, two)
            c.set_option("two_trips", 0)
            for r, samples in got:
                assert r == (want[0], want[1].size, want[2], want[3]), (name, r, want[0], want[1].size, want[2:])
                assert np.array_equal(samples, want[1]), name
            c.free(d)
        c.free(d_wav)
    finally:
        c.close()


def _make_encoder_lose_its_grid(c, gen):
    """-> what to undo.  Second generation (eight waves per frame): one workgroup per CU more than fit, so that part of
    the grid is not resident.  Third generation (one wave per frame): one workgroup generation never publishes its total
    (option wave_drop), which is what a workgroup that is not resident looks like to all the others."""
    c.set_option("enc_gen", gen)
    c.set_option("host_chunk_frames", -1)   # one launch over the whole input (the host front end would take it in chunks)
    if gen == 2:
        rc, _, _ = c.encode(np.zeros(30000, dtype=np.int16), None)
        natural = c.get_option("stream_wgs_in_use")
        assert rc == 0 and natural >= 1
        c.set_option("stream_wgs", natural + 1)
    else:
        c.set_option("wave_drop", 300)


@pytest.mark.parametrize("gen", [3, 2])
def test_encoder_falls_back_when_grid_not_resident(x3, gen):
    """the single-pass encoders' workgroups wait for each other's frame sizes; when part of the grid is not there, the
    waits must time out in bounded time and x3_encode_result must hand back the two-pass kernels' bit-exact result"""
    import time
    c = x3.Context(0)
    try:
        # (long enough that every resident workgroup has a second frame behind frames of workgroups that are not)
        wav = x3.synth(2, 321, 0, 10000 * 6400 + 17)
        _make_encoder_lose_its_grid(c, gen)
        t0 = time.perf_counter()
        rc, out, stats = c.encode(wav, x3.Params.default())
        dt = time.perf_counter() - t0
        assert c.get_option("encode_fallbacks") == 1
    finally:
        c.close()
    rco, oo, so = O.encode(wav)
    assert rc == rco == 0 and np.array_equal(out, oo) and stats.tolist() == so.tolist()
    assert dt < 20.0, dt


@pytest.mark.parametrize("gen", [3, 2])
def test_encoder_fallback_leaves_the_prefix_alone(x3, gen):
    """x3_encode_dev with start_pos > 0 promises output at d_out[start_pos..) only.  A wave or workgroup whose size
    wait times out has no offset for its frames: none of them may be written (they used to land at d_out + 0), and the
    two-pass re-run rewrites start_pos.. only -- a sentinel-filled prefix (an archive header, an earlier
    sub-stream) must survive the fallback."""
    c = x3.Context(0)
    try:
        p = x3.Params.default()
        _make_encoder_lose_its_grid(c, gen)
        n = 10000 * 6400 + 17
        wav = x3.synth(2, 654, 0, n)
        for start_pos in (4096, 321):
            d_wav = c.alloc(2 * n + 64)
            c.upload(d_wav, wav)
            cap = start_pos + x3.lib().x3_encode_bound(n, C.byref(p)) + 16
            d_out = c.alloc(cap)
            c.upload(d_out, np.full(cap, 0xA5, dtype=np.uint8))
            before = c.get_option("encode_fallbacks")
            assert c.encode_dev(d_wav, n, p, d_out, cap, start_pos) == 0
            rc, pos, stats = c.encode_result()
            assert rc == 0 and c.get_option("encode_fallbacks") == before + 1
            got = c.download(d_out, (pos + 3) & ~3)[:pos]
            rco, oo, so = O.encode(wav, start_pos=start_pos)
            assert pos == oo.size and np.array_equal(got[(start_pos + 1) & ~1:], oo[(start_pos + 1) & ~1:])
            assert (got[:start_pos] == 0xA5).all(), "bytes in front of start_pos were overwritten"
            c.free(d_wav); c.free(d_out)
    finally:
        c.close()


@pytest.mark.parametrize("nwg,m", [(1, 1), (1, 16), (3, 5), (7, 3), (16, 16), (64, 2), (256, 1)])
def test_wave_encoder_many_generations(x3, nwg, m):
    """the wave-per-frame encoder on small inputs with few workgroups / few waves (options wave_nwg, wave_m): dozens to
    hundreds of generations per workgroup, every one with the size exchange in LDS, the generation totals and bases,
    the last generation short -- byte for byte against the oracle, with no fallback"""
    c = x3.Context(0)
    try:
        c.set_option("wave_nwg", nwg)
        c.set_option("wave_m", m)
        for n in (10000 * 37 + 123, 10000 * 200, 10000 * 513 + 1):
            wav = x3.synth(x3.SYNTH_HYDROPHONE, 0x58330002, 0, n)
            rc_o, s_o, st_o = O.encode(wav)
            for rep in range(2):
                rc, s, st = c.encode(wav)
                assert rc == rc_o == 0 and np.array_equal(s, s_o) and st.tolist() == st_o.tolist(), (nwg, m, n, rep)
        assert c.get_option("encode_fallbacks") == 0 and c.get_option("encode_dense_reruns") == 0
    finally:
        c.close()


def _with_loud_frames(x3, n, loud_frames, seed):
    """hydrophone-like content of n samples in which the default frames (10 000 samples) listed in loud_frames are
    full-scale noise: their payloads (~20 KB) do not fit the wave encoder's 9 728-byte LDS image"""
    wav = x3.synth(x3.SYNTH_HYDROPHONE, seed, 0, n).copy()
    rng = np.random.default_rng(seed)
    for f in loud_frames:
        lo, hi = 10000 * f, min(n, 10000 * (f + 1))
        wav[lo:hi] = rng.integers(-32768, 32767, hi - lo, dtype=np.int16)
    return wav


@pytest.mark.parametrize("dense", ["none", "one", "three", "half", "all", "last_ragged"])
def test_wave_encoder_dense_frames_take_the_dense_pass(x3, dense):
    """Frames whose payload does not fit the wave encoder's LDS image (more than 9 728 bytes: loud or noisy content) are
    ordinary content to the reference (encoder.rs:289-315).  The wave kernel sizes and places them, the dense pass behind
    it in the same stream writes them: the stream equals the oracle's whatever the share of such frames, NO call is
    encoded twice (round 3 re-encoded the whole call for one such frame), and the count of dense frames is exact."""
    F = 40
    n = 10000 * F - (3777 if dense == "last_ragged" else 0)
    loud = {"none": [], "one": [17], "three": [0, 18, 39], "half": list(range(0, F, 2)), "all": list(range(F)),
            "last_ragged": [5, F - 1]}[dense]
    wav = _with_loud_frames(x3, n, loud, 7)
    rc_o, s_o, st_o = O.encode(wav)
    assert rc_o == 0
    c = x3.Context(0)
    try:
        for start in (0, 321):   # (an odd start position: the pad byte and every frame offset move)
            rc_o, s_o, st_o = O.encode(wav, start_pos=start)
            rc, s, st = c.encode(wav, start_pos=start)
            assert rc == rc_o == 0 and np.array_equal(s, s_o) and st.tolist() == st_o.tolist(), (dense, start)
            if c.get_option("enc_gen_in_use") == 3:
                assert c.get_option("last_dense_frames") == len(loud), (dense, c.get_option("last_dense_frames"))
        assert c.get_option("encode_dense_reruns") == 0 and c.get_option("encode_fallbacks") == 0
    finally:
        c.close()


def test_wave_encoder_dense_hint_is_only_a_hint(x3):
    """A call in which more than a quarter of the frames were dense makes the context's NEXT call start on the
    second-generation kernel (which holds worst-case images); that kernel counts dense frames as well, and below an
    eighth the wave encoder is back.  Bytes and statistics equal the oracle's throughout, and one loud frame among many
    never moves the context off the wave encoder."""
    c = x3.Context(0)
    try:
        quiet = x3.synth(x3.SYNTH_HYDROPHONE, 5, 0, 400000)
        one_loud = _with_loud_frames(x3, 400000, [12], 8)
        white = x3.synth(x3.SYNTH_WHITE, 6, 0, 250000)
        # (content, generation expected to serve it)
        for wav, gen in ((quiet, 3), (one_loud, 3), (one_loud, 3), (white, 3), (white, 2), (quiet, 2), (quiet, 3), (one_loud, 3)):
            rc_o, s_o, st_o = O.encode(wav)
            rc, s, st = c.encode(wav)
            assert rc == rc_o == 0 and np.array_equal(s, s_o) and st.tolist() == st_o.tolist()
            assert c.get_option("enc_gen_in_use") == gen, (gen, c.get_option("enc_gen_in_use"))
        assert c.get_option("encode_dense_reruns") == 0 and c.get_option("encode_dense_frames") == 1 + 1 + 25 + 1
    finally:
        c.close()


def test_dense_frames_stream_is_complete_without_encode_result(x3):
    """x3_encode_dev followed on the same stream by x3_decode_dev, with x3_encode_result only afterwards: the dense pass
    is part of the encode's work in the stream, so the decoder finds every frame (with the round-3 rerun inside
    x3_encode_result a chained consumer met CRC-bad frames)."""
    import ctypes as C
    c = x3.Context(0)
    try:
        n = 10000 * 64
        wav = _with_loud_frames(x3, n, [3, 31, 63], 11)
        p = x3.Params.default()
        L = x3.lib()
        F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
        d_wav = c.alloc(2 * n); d_out = c.alloc(cap + 16); d_off = c.alloc(8 * (F + 1)); d_back = c.alloc(2 * n)
        c.upload(d_wav, wav)
        assert c.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        assert c.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
        rc, pos, st = c.encode_result()
        assert rc == 0 and c.get_option("last_dense_frames") == 3
        r = c.decode_result()
        assert r[:3] == (0, F, 0), r
        back = c.download(d_back, 2 * n, np.int16)
        assert np.array_equal(back, wav)
        for d in (d_wav, d_out, d_off, d_back): c.free(d)
    finally:
        c.close()


@pytest.mark.parametrize("n", [1, 2, 3, 21, 22, 41, 82, 5121, 5122, 5141, 10002, 10018, 25121, 160001])
@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_wave_encoder_ragged_frames(ctx, x3, kind, n):
    """frames whose last block has 1..18 samples take the wave encoder's generic path (sample by sample from memory);
    all five signal kinds, lengths around the half-frame and frame boundaries"""
    wav = x3.synth(kind, 77, 0, n)
    rc_o, s_o, st_o = O.encode(wav)
    rc, s, st = ctx.encode(wav)
    assert rc == rc_o == 0 and np.array_equal(s, s_o) and st.tolist() == st_o.tolist()


def test_random_parameter_sweep(ctx, x3):
    """seeded sweep over geometry x codes x thresholds x signal scale x start position: encode equals the oracle
    byte for byte (or fails with the same status), and streams of the default codes decode back (GPU == oracle)"""
    rng = np.random.default_rng(20261003)
    offsets = [6, 11, 20, 28]  # Rice table offsets by code (x3.rs:200-252): thresholds 0,1 must not exceed them
    done = 0
    for trial in range(48):
        bl = int(rng.choice([1, 2, 3, 5, 8, 13, 19, 20, 21, 32, 47, 60]))
        bpf = int(rng.integers(1, 40)) if trial % 3 else int(rng.choice([100, 500, 777]))
        codes = (0, 1, 3) if trial % 2 else tuple(int(c) for c in rng.integers(0, 4, size=3))
        thr = tuple(int(rng.integers(0, offsets[c] + 1)) for c in codes)
        p = x3.Params.make(bl, bpf, codes, thr)
        assert x3.lib().x3_params_validate(C.byref(p)) == 0
        n = int(rng.integers(1, 4 * bl * bpf + 40))
        base = x3.synth(int(rng.choice([1, 2, 4])), 700 + trial, 0, n).astype(np.int32)
        scale = int(rng.choice([1, 1, 2, 7, 64, 4096]))   # widen the differences: Rice -> BFP -> literal
        if rng.random() < 0.5:
            wav = np.clip(base // max(1, 4096 // scale), -32768, 32767).astype(np.int16)
        else:
            wav = np.clip(base * scale // 64, -32768, 32767).astype(np.int16)
        sp = int(rng.integers(0, 3))
        rc_o, out_o, st_o = O.encode(wav, oparams(p), start_pos=sp)
        rc_g, out_g, st_g = ctx.encode(wav, p, start_pos=sp)
        assert rc_g == rc_o, (trial, bl, bpf, codes, thr, rc_g, rc_o)
        if rc_o == 0:
            assert np.array_equal(out_g, out_o) and st_g.tolist() == st_o.tolist(), (trial, bl, bpf, codes, thr)
            body = out_o[(sp + 1) & ~1:]
            if codes == (0, 1, 3):
                # the reference DECODER is hard-wired to codes 0,1,3 (decoder.rs:180): only such streams round-trip,
                # in the reference as here
                r = check_decode(ctx, x3, body, p, wav_cap=n)
                # ... and only when BFP blocks have nb >= 5: the reference decoder rejects E = nb + 1 <= 5
                # (decoder.rs:213), which its own encoder emits for a third threshold below 16
                if thr[2] >= 16:
                    assert r[0] == 0 and np.array_equal(r[1], wav), (trial, bl, bpf, codes, thr)
            done += 1
    assert done >= 24


def test_decode_dev_survives_wild_frame_offsets(ctx, x3):
    """x3_decode_dev takes the frame index from the caller: offsets that point into payloads, past the end of the
    stream or nowhere near a frame must end as per-frame errors, never as reads outside the stream buffer."""
    rng = np.random.default_rng(11)
    n = 64 * 10000 * 3
    p = x3.Params.default()
    wav = x3.synth(2, 91, 0, n)
    rc, stream, _ = ctx.encode(wav, p)
    assert rc == 0
    F = n // 10000
    good = np.array(frame_offsets(stream) + [stream.size], dtype=np.uint64)
    assert good.size == F + 1
    d_x3 = ctx.alloc(stream.size + 16); d_off = ctx.alloc(8 * (F + 1)); d_wav = ctx.alloc(2 * n)
    L = x3.lib()
    L.x3_dev_upload(ctx._h, d_x3, stream.ctypes.data, stream.size)
    try:
        for trial in range(6):
            offs = good.copy()
            k = rng.integers(0, F, size=F // 3)
            if trial == 0:
                offs[k] = rng.integers(0, stream.size, size=k.size)                 # anywhere inside
            elif trial == 1:
                offs[k] = stream.size - rng.integers(0, 40, size=k.size)            # the last bytes
            elif trial == 2:
                offs[k] = stream.size + rng.integers(0, 1 << 20, size=k.size)       # beyond the end
            elif trial == 3:
                offs[k] = np.uint64(1) << np.uint64(40)                             # far beyond
            elif trial == 4:
                offs[k] = offs[k] + np.uint64(2) * rng.integers(1, 9, size=k.size).astype(np.uint64)  # a few bytes off
            else:
                offs[:] = 0                                                         # every lane on frame 0
            L.x3_dev_upload(ctx._h, d_off, offs.ctypes.data, offs.size * 8)
            rc = ctx.decode_dev(d_x3, stream.size, d_off, F, p, d_wav, n, n_per_clip=n)
            assert rc == 0
            rc, first_bad, st, before = ctx.decode_result()
            assert rc == 0 and first_bad <= F
            if trial == 5:
                assert first_bad == F  # sixty-four copies of frame 0 per group: all valid
            else:
                assert first_bad < F and st != 0 and first_bad == int(np.min(k)) or trial == 0
    finally:
        ctx.free(d_x3); ctx.free(d_off); ctx.free(d_wav)


# ------------------------------------------------------------------ multi-GPU entry points on the one GPU of a test box

def test_shard_world_of_one_over_rccl(ctx, x3):
    """x3_shard_* with world = 1: librccl is opened, a communicator is built, the all-gather and the gather run --
    the same calls every rank of an 8-GPU job makes (the offset arithmetic for more ranks is covered on the CPU:
    tests/host_cpp/test_shard_logic.cpp, tests/test_sharding_gloo.py; more ranks on a box that has them:
    test_mgpu_over_all_visible_devices, test_bench_strong_scaling_over_all_visible_devices)"""
    p = x3.Params.default()
    n = 1_234_567
    wav = x3.synth(2, 4711, 0, n)
    F = x3.lib().x3_num_frames(n, C.byref(p))
    cap = x3.lib().x3_encode_bound(n, C.byref(p))
    d_wav, d_out, d_off, d_whole, d_len = ctx.alloc(2 * n + 64), ctx.alloc(cap), ctx.alloc(8 * (F + 1)), ctx.alloc(cap), ctx.alloc(64)
    ctx.upload(d_wav, wav)
    sh = x3.Shard(ctx, x3.shard_unique_id(), 0, 1)
    try:
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        sh.exchange_lengths(d_off + 8 * F)            # the shard's own array
        sh.exchange_lengths(d_off + 8 * F, d_len)     # a caller's array
        rc, pos, _ = ctx.encode_result()
        assert rc == 0
        lens = sh.lengths()
        assert lens == [pos] and ctx.download(d_len, 8, np.uint64).tolist() == [pos]
        assert sh.gather(d_out, lens, 0, d_whole, cap) == pos
        ctx.sync()
        ref = O.encode(wav)[1]
        assert np.array_equal(ctx.download(d_whole, (pos + 3) & ~3)[:pos], ref)
        with pytest.raises(x3.X3Error):
            sh.gather(d_out, lens, 0, d_whole, pos - 2)  # destination too small
        # the overlapped form: on the shard's own stream, behind what the context has enqueued; one in flight
        ctx.upload(d_whole, np.zeros(cap, dtype=np.uint8))
        assert sh.gather(d_out, lens, 0, d_whole, cap, overlapped=True) == pos
        with pytest.raises(x3.X3Error):
            sh.gather(d_out, lens, 0, d_whole, cap, overlapped=True)   # the first one has not been waited for
        sh.gather_wait(on_stream=False)
        assert np.array_equal(ctx.download(d_whole, (pos + 3) & ~3)[:pos], ref)
        assert sh.gather(d_out, lens, 0, d_whole, cap, overlapped=True) == pos
        sh.gather_wait(on_stream=True)
        ctx.sync()
    finally:
        sh.close()
        for d in (d_wav, d_out, d_off, d_whole, d_len):
            ctx.free(d)


def _visible_devices():
    import torch
    return max(1, torch.cuda.device_count())   # (counting devices does not initialise the GPU)


def test_mgpu_over_all_visible_devices(x3):
    """x3_mgpu_* (all GPUs from one process: a context, a shard and a host thread per device; lengths by ncclAllGather,
    sub-streams to device 0 by grouped send/recv) over EVERY device the box shows -- one on the single-GPU boxes, eight
    on a node: same bytes and status as the single-context calls and as the oracle"""
    G = _visible_devices()
    m = x3.MultiGpu(list(range(G)))
    try:
        # (enough frames for every device to get some, and sizes that leave the last devices with none or a ragged tail)
        for kind, n, sp in ((2, 345_678, 0), (4, 10_000, 3), (1, 25_001, 0), (0, 1, 1), (2, 10_000 * (3 * G + 1) + 17, 0),
                            (3, 10_000 * G, 2)):
            wav = x3.synth(kind, 31 + kind, 0, n)
            rc, out, stats = m.encode(wav, start_pos=sp)
            rco, oo, so = O.encode(wav, start_pos=sp)
            assert rc == rco == 0 and np.array_equal(out[sp:], oo[sp:]) and stats.tolist() == so.tolist(), (G, kind, n, sp)
            body = oo[(sp + 1) & ~1:]
            r = m.decode_stream(body, wav_cap=n)
            assert r[0] == 0 and np.array_equal(r[1], wav) and r[3] == 0, (G, kind, n, sp)
        assert m.encode(wav, n_channels=2)[0] == x3.ERR_MORE_THAN_ONE_CHANNEL
        assert m.encode(x3.synth(2, 1, 0, 30000 * G), cap=100)[0] == x3.ERR_BYTE_WRITER_INSUFFICIENT_MEMORY
        # a damaged frame in the middle: nothing behind it is delivered, whichever device it falls to
        wav = x3.synth(2, 77, 0, 10_000 * (2 * G + 3))
        good = O.encode(wav)[1].copy()
        bad = good.copy()
        bad[len(bad) // 2] ^= 0x40
        r_g = m.decode_stream(bad, wav_cap=wav.size)
        r_o = O.decode_stream(bad, wav_cap=wav.size)
        assert r_g[0] == r_o[0] and r_g[2] == r_o[2] and np.array_equal(r_g[1], r_o[1])
    finally:
        m.close()
    with pytest.raises(x3.X3Error):
        x3.MultiGpu([0, 0])   # one rank per GPU


def test_bench_strong_scaling_over_all_visible_devices():
    """bench.py --strong under torch.distributed.run with one rank per visible device (skipped on a single-GPU box): the
    ranks encode their frame ranges, exchange the lengths over RCCL and reassemble the stream on rank 0 in all three
    gather modes; rank 0 compares the reassembled stream with the oracle's encoding of the whole signal, byte for byte"""
    import subprocess
    import sys
    G = _visible_devices()
    if G < 2:
        pytest.skip("one GPU: the one-rank form of this run is test_bench_distributed_path_with_one_rank")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket

    def free_port():   # (two or three launches back to back: a fixed port is still in TIME_WAIT for the next one)
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        return port
    for mode in ("in-step", "overlapped", "sharded"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G),
                            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", str(G),
                            "--steps", "3", "--warmup", "2", "--strong", "--total-samples", str(10_000 * (37 * G + 5) + 123),
                            "--verify-gather", "--gather", mode, "--mode-steps", "3"],
                           capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert r.returncode == 0, (mode, r.stdout[-1500:], r.stderr[-3000:])
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert j["n_gpus"] == G and j["rccl_ranks"] == G and j["scaling"] == "strong"
        assert j["gather"]["whole_stream_verified_vs_oracle"] is True and j["gather"]["mode"] == mode
        assert j["gather"]["sharded_file"]["whole_file_verified_vs_oracle"] is True
        assert set(j["gather_modes"]) == {"in-step", "overlapped", "sharded", "none"}


def test_decode_frame_loop_with_prefetch(ctx):
    """x3_decode_prefetch: a loop over decode_frame (decoder.rs:36-58) served from windows decoded ahead gives what the
    per-call path gives -- also for a frame whose payload CRC is wrong (decode_frame does not look at it), a frame
    called with another sample count than its header's, and payloads outside the announced buffer"""
    import time
    import x3hip
    n = 10000 * 300 + 123
    wav = x3hip.synth(2, 99, 0, n)
    rc, x3, _ = ctx.encode(wav)
    assert rc == 0
    x3 = np.ascontiguousarray(x3)
    bad = x3.copy()
    frames = []
    off = 0
    while off + 20 <= x3.size:
        h = x3hip.read_frame_header(x3[off:off + 20])
        assert h[0] == 0
        frames.append((off, h[1].payload_len, h[1].samples))
        off += 20 + h[1].payload_len
    assert len(frames) == 301
    o7, l7, s7 = frames[7]
    bad[o7 + 20 + l7 // 2] ^= 0x10   # payload of frame 7 tampered, CRC left alone

    def loop(buf):
        out = []
        for (o, ln, ns) in frames:
            out.append(ctx.decode_frame(buf[o + 20:o + 20 + ln], ns))
        return out

    for buf in (x3, bad):
        ctx.decode_prefetch(None)
        t0 = time.time(); plain = loop(buf); t1 = time.time()
        assert ctx.decode_prefetch(buf) == 0
        cached = loop(buf); t2 = time.time()
        for (a, b) in zip(plain, cached):
            assert a[0] == b[0] and np.array_equal(a[1], b[1])
        print("decode_frame loop: %.1f ms per-call, %.1f ms with prefetch" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    good = np.concatenate([r[1] for r in loop(x3)])
    assert np.array_equal(good, wav)
    # another sample count than the header's, and a payload that is not in the announced buffer: the per-call path
    o, ln, ns = frames[3]
    a = ctx.decode_frame(x3[o + 20:o + 20 + ln], ns - 20)
    ctx.decode_prefetch(None)
    b = ctx.decode_frame(x3[o + 20:o + 20 + ln], ns - 20)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    ctx.decode_prefetch(x3)
    c = ctx.decode_frame(x3[o + 20:o + 20 + ln].copy(), ns)
    assert c[0] == 0 and np.array_equal(c[1], wav[3 * 10000:3 * 10000 + ns])
    ctx.decode_prefetch(None)


@pytest.mark.parametrize("bpf", [1, 2, 3, 4, 7, 8, 12, 16, 28, 64, 100, 500, 801])
def test_split_decoder_frame_sizes(ctx, x3, bpf):
    """the three-wave decoder over frame sizes on both sides of its "regular group" rule (64 equal frames side by
    side, a multiple of 16 samples each: whole 128-byte lines by the flusher wave; anything else: row by row by the
    valuer) -- 200 full frames and a ragged tail, all signal kinds incl. all-literal, decoded from the device API
    with and without explicit sample offsets; every sample against the oracle's decode of the same stream"""
    p = x3.Params.make(20, bpf)
    spf = 20 * bpf
    for kind, seed in ((2, 11), (1, 12), (0, 13), (4, 14)):
        n = spf * 200 + (spf // 3 if bpf > 2 else 1)
        wav = x3.synth(kind, seed + bpf, 0, n)
        rc, stream, _ = ctx.encode(wav, p)
        assert rc == 0
        r = check_decode(ctx, x3, stream, p, wav_cap=n)
        if r[0] != 0:   # (801 blocks of white noise: a payload beyond the reader's 24 576 bytes, decodefile.rs:118-121)
            assert r[0] == x3.ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN and kind == 1
            continue
        assert np.array_equal(r[1], wav)
        # the device API with the encoder's frame index: the decoder writes straight into the caller's geometry
        d_wav = ctx.alloc(2 * n + 64)
        ctx.upload(d_wav, wav)
        cap = int(x3.lib().x3_encode_bound(n, C.byref(p)))
        F = int(x3.lib().x3_num_frames(n, C.byref(p)))
        d_out, d_off, d_back = ctx.alloc(cap + 16), ctx.alloc(8 * (F + 1)), ctx.alloc(2 * n + 64)
        assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
        rc, pos, _ = ctx.encode_result()
        assert rc == 0 and pos == stream.size
        assert ctx.decode_dev(d_out, pos, d_off, F, p, d_back, n, n_per_clip=n) == 0
        assert ctx.decode_result() == (0, F, 0, n)
        assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), (bpf, kind)
        for d in (d_wav, d_out, d_off, d_back):
            ctx.free(d)


def test_parity_soak_sample(ctx):
    """a fixed slice of tools/fuzz_parity.py (random content x geometry x damage, HIP path == oracle): 420 trials of
    seed 7; the tool itself runs for as long as it is given"""
    import importlib.util
    import sys
    path = os.path.join(os.path.dirname(G.rstrip("/")), "..", "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.abspath(path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fuzz_parity"] = mod
    spec.loader.exec_module(mod)
    counts = mod.run(seed=7, trials=420, families="egdbaf", context=ctx)
    assert sum(counts.values()) == 420 and all(counts[k] > 40 for k in "egdbaf"), counts


def test_decode_frame_short_buffer_reports_the_earlier_error(ctx, x3):
    """decode_frame slices wav per block (decoder.rs:49): with a buffer shorter than `samples` the reference panics at
    the first block that does not fit, but a decode error in a block in front of it is returned first"""
    p = x3.Params.default()
    wav = x3.synth(2, 4242, 0, 10000)
    stream = O.encode(wav)[1]
    payload = stream[20:].copy()
    for cap in (10000, 9999, 9981, 9980, 5000, 21, 20, 2, 1):
        r_o = O.decode_frame(payload, 10000, oparams(p), wav_cap=cap)
        r_g = ctx.decode_frame(payload, 10000, p, wav_cap=cap)
        assert r_g[0] == r_o[0] == (0 if cap >= 10000 else x3.ERR_BAD_ARG), (cap, r_g[0], r_o[0])
    bad = payload.copy()
    bad[700:712] = 0                      # a zero run in the payload's first quarter: OutOfBoundsInverse / InvalidBPF there
    full = O.decode_frame(bad, 10000, oparams(p), wav_cap=10000)
    assert full[0] != 0
    seen = set()
    for cap in (10000, 9999, 6000, 3000, 2500, 2000, 1500, 1000, 500, 100, 21, 1):
        r_o = O.decode_frame(bad, 10000, oparams(p), wav_cap=cap)
        r_g = ctx.decode_frame(bad, 10000, p, wav_cap=cap)
        assert r_g[0] == r_o[0], (cap, r_g[0], r_o[0])
        seen.add(r_o[0])
    assert full[0] in seen and x3.ERR_BAD_ARG in seen


def test_decode_frame_beyond_the_walk_limits(ctx, x3):
    """decoder::decode_frame has no payload or sample limit of its own (the 24 KB read buffer and Frame::MAX_LENGTH belong
    to the walk): payloads of 30 KB and 70 KB, 70 000 samples -- what no frame header can describe -- equal the oracle"""
    rng = np.random.default_rng(77)
    p = x3.Params.default()
    for n, amp in ((14000, 32768), (33000, 32768), (70000, 40), (70000, 32768)):
        wav = rng.integers(-amp, amp, size=n).astype(np.int16)
        pp = x3.Params.make(20, (n + 19) // 20)            # one frame
        rc, stream, _ = O.encode(wav, oparams(pp))
        assert rc == 0
        payload = stream[20:].copy()                      # (the header's 16-bit fields have wrapped; the payload is whole)
        r_o = O.decode_frame(payload, n, oparams(p))
        r_g = ctx.decode_frame(payload, n, p)
        assert r_o[0] == 0 and r_g[0] == 0 and np.array_equal(r_g[1], r_o[1]) and np.array_equal(r_g[1], wav), (n, amp)
        bad = payload.copy(); bad[len(bad) // 2: len(bad) // 2 + 12] = 0
        r_o = O.decode_frame(bad, n, oparams(p))
        r_g = ctx.decode_frame(bad, n, p)
        assert r_g[0] == r_o[0] and (r_o[0] != 0 or np.array_equal(r_g[1], r_o[1])), (n, amp, r_g[0], r_o[0])


def test_block_len_zero_from_a_damaged_archive_header(ctx, x3):
    """<BLKLEN>00</BLKLEN> (one flipped bit in the XML, which no CRC protects in the reference): decode_frame then walks
    empty blocks -- Rice blocks read only their type bits, the first BFP block is FrameDecodeInvalidBPF (E <= 5) or the
    reference's panic.  x3_x3a_decode, the reader and x3_decode_frame follow the oracle through both endings."""
    rng = np.random.default_rng(99)
    seen = set()
    for trial in range(40):
        n = int(rng.integers(2, 30000))
        kind = trial % 4
        if kind == 0:
            wav = rng.integers(-32768, 32768, size=n).astype(np.int16)
        elif kind == 1:
            wav = np.cumsum(rng.integers(-40, 41, size=n)).astype(np.int16)
        elif kind == 2:
            wav = np.cumsum(rng.integers(-2, 3, size=n)).astype(np.int16)
        else:
            wav = x3.synth(2, 500 + trial, 0, n)
        arch = O.x3a_encode(wav, 48000)[1].copy()
        at = bytes(arch).find(b"<BLKLEN>20")
        assert at > 0
        arch[at + 8] = ord("0")
        r_o = O.x3a_decode(arch, wav_cap=n + 70000)
        r_g = ctx.x3a_decode(arch, wav_cap=n + 70000)
        assert (r_g[0],) + tuple(r_g[2:]) == (r_o[0],) + tuple(r_o[2:]), (trial, r_g[0], r_g[2:], r_o[0], r_o[2:])
        assert np.array_equal(r_g[1], r_o[1])
        seen.add((r_o[0], r_o[4]))
        rd = x3.Reader(ctx, arch)
        assert rd.rc == 0
        rc, smp = rd.next_frame()
        assert (rc, rd.frame_errors()) == (r_o[0], r_o[4]), (trial, rc, rd.frame_errors(), r_o[0], r_o[4])
        rd.close()
        p0 = x3.Params.make(0, 500)
        hdr = 28 + (int(arch[14]) << 8 | int(arch[15]))
        plen = int(arch[hdr + 6]) << 8 | int(arch[hdr + 7])
        ns = int(arch[hdr + 4]) << 8 | int(arch[hdr + 5])
        payload = arch[hdr + 20: hdr + 20 + plen]
        f_o = O.decode_frame(payload, ns, O.Params.make(0, 500))
        f_g = ctx.decode_frame(payload, ns, p0)
        assert f_g[0] == f_o[0] and f_o[0] in (x3.ERR_BAD_ARG, x3.ERR_FRAME_DECODE_INVALID_BPF), (trial, f_g[0], f_o[0])
    assert (0, 1) in seen and (x3.ERR_BAD_ARG, 0) in seen, seen      # a counted frame error, and the panic
