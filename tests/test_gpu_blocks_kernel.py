"""Round 6's block-per-lane decoder (x3-rust_amd/csrc/x3_decode_blocks_kernel.h, option "decode_blocks" /
X3HIP_DECODE_BLOCKS): a walker wave that finds where the blocks begin + three decoder waves that decode a block per lane.
It is not the default kernel (it walks every frame twice and is slower on config 3), so it gets its own tests: the same
answers as the oracle, as the three-wave kernel and as the single-wave kernel on good and on damaged streams, on the
layouts it takes (any number of blocks of 20 per frame, rows on 8-byte boundaries, batches of clips), and the whole
decoder parity suite once more in a child process that starts with X3HIP_DECODE_BLOCKS=1 under guard pages.
Reference: src/decoder.rs:36-58,132-235, src/bitreader.rs:105-139, src/decodefile.rs:93-136."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def x3():
    import x3hip
    return x3hip


@pytest.fixture(scope="module")
def ctx(x3):
    c = x3.Context(0)
    c.set_option("decode_blocks", 1)
    yield c
    c.close()


def frame_offsets(stream):
    offs, pos = [], 0
    while pos + 20 <= len(stream):
        offs.append(pos)
        pos += 20 + (int(stream[pos + 6]) << 8 | int(stream[pos + 7]))
    return offs


def refresh_crcs(x3, s, off):
    """payload and header CRC of the frame at `off`, after its payload was tampered with"""
    plen = int(s[off + 6]) << 8 | int(s[off + 7])
    pc = O.crc16(s[off + 20:off + 20 + plen])
    s[off + 18], s[off + 19] = pc >> 8, pc & 0xFF
    hc = O.crc16(s[off:off + 16])
    s[off + 16], s[off + 17] = hc >> 8, hc & 0xFF


def test_round_trips_and_kernel_in_use(ctx, x3):
    p = x3.Params.default()
    for kind, n in ((2, 1), (2, 2), (2, 21), (2, 22), (2, 41), (2, 10_000), (2, 10_001), (2, 16_000), (0, 20_001), (1, 30_000),
                    (3, 50_000), (4, 123_457), (2, 640_000), (1, 700_000), (2, 3_000_017)):
        wav = x3.synth(kind, 770 + kind, 0, n)
        rc_o, stream, _ = O.encode(wav)
        assert rc_o == 0
        r = ctx.decode_stream(stream, p, wav_cap=n)
        assert ctx.get_option("decode_kernel_in_use") == 3, (kind, n)
        o = O.decode_stream(stream, O.Params.default(), wav_cap=n)
        assert (r[0], r[2], r[3]) == (o[0], o[2], o[3]) and np.array_equal(r[1], o[1]) and np.array_equal(r[1], wav), (kind, n)


def test_three_decoder_kernels_agree_on_damaged_streams(ctx, x3):
    """block-per-lane, three-wave and single-wave kernels against the oracle: status, frame counts, samples
    (damage inside payloads with the CRCs made good again: decode errors, long zero runs, reads behind the payload)"""
    rng = np.random.default_rng(611)
    wav = np.concatenate([x3.synth(k, 190 + k, 0, 30011 + 977 * k) for k in range(5)])
    stream = O.encode(wav)[1]
    offs = frame_offsets(stream)
    cases = [stream]
    for trial in range(36):
        s = stream.copy()
        fi = int(rng.integers(0, len(offs)))
        plen = int(s[offs[fi] + 6]) << 8 | int(s[offs[fi] + 7])
        pos = offs[fi] + 22 + int(rng.integers(0, plen - 12))
        if trial % 4 == 0:
            s[pos] ^= 1 << int(rng.integers(0, 8))
        elif trial % 4 == 1:
            s[pos:pos + 8] = 0
        elif trial % 4 == 2:
            s[pos:pos + 6] = rng.integers(0, 256, size=6, dtype=np.uint8)
        else:   # the header asks for more samples than the payload holds: reads behind the payload
            ns = min(65535, (int(s[offs[fi] + 4]) << 8 | int(s[offs[fi] + 5])) + int(rng.integers(1, 400)))
            s[offs[fi] + 4], s[offs[fi] + 5] = ns >> 8, ns & 0xFF
        refresh_crcs(x3, s, offs[fi])
        cases.append(s)
    cap = wav.size + 70000
    on_blocks = 0
    for s in cases:
        o = O.decode_stream(s, O.Params.default(), wav_cap=cap)
        got = {}
        for name, opts in (("blocks", {"decode_blocks": 1}), ("three", {"decode_blocks": 0}), ("single", {"decode_blocks": 0, "decode_single": 1})):
            for k, v in opts.items():
                ctx.set_option(k, v)
            try:
                got[name] = ctx.decode_stream(s, x3.Params.default(), wav_cap=cap)
                # (a damaged sample count moves the rows behind it off the 8-byte grid: such a stream goes to the single-wave kernels)
                if name == "blocks" and ctx.get_option("decode_kernel_in_use") == 3:
                    on_blocks += 1
            finally:
                ctx.set_option("decode_single", 0)
                ctx.set_option("decode_blocks", 1)
        for name, a in got.items():
            assert a[0] == o[0] and a[2:] == o[2:] and np.array_equal(a[1], o[1]), name
    assert on_blocks >= 24, on_blocks


@pytest.mark.parametrize("bpf", [1, 3, 7, 16, 17, 31, 32, 33, 100, 256, 499, 500, 501, 502, 1000, 3200])
def test_blocks_per_frame(ctx, x3, bpf):
    """any number of blocks of 20 samples per frame (the first batch of a frame is cut so that it ends on a 128-byte line
    of the output; frames of an odd number of blocks put rows on 8-byte boundaries)"""
    p = x3.Params.make(20, bpf)
    po = O.Params.make(20, bpf)
    spf = 20 * bpf
    for nfr, tail in ((1, 0), (3, 1), (70, 0), (131, spf // 2 + 3)):
        n = nfr * spf + tail
        if n > 4_000_000:
            n = 4_000_000 // spf * spf + tail
        wav = x3.synth(2, 5100 + bpf + nfr, 0, n)
        rc, stream, _ = O.encode(wav, po)
        assert rc == 0
        r = ctx.decode_stream(stream, p, wav_cap=n)
        o = O.decode_stream(stream, po, wav_cap=n)
        assert (r[0], r[2], r[3]) == (o[0], o[2], o[3]) and np.array_equal(r[1], o[1]), (bpf, nfr, tail)
        assert o[0] != 0 or np.array_equal(r[1], wav)


def test_batch_of_ragged_clips_on_the_device(ctx, x3):
    """config-5 style batches: clips whose last frame is short, clip strides that put rows on 8-byte boundaries only, and an
    output buffer that begins 8 bytes into a 16-byte unit"""
    L = x3.lib()
    p = x3.Params.default()
    for npc, clips, stride_extra, shift in ((57_600, 40, 0, 0), (25_013, 33, 3, 0), (10_001, 130, 7, 8), (9_999, 77, 1, 8), (19, 500, 1, 0)):
        stride = npc + stride_extra
        stride += (-stride) % 4          # rows on 8-byte boundaries: what the kernel takes
        F = L.x3_num_frames(npc, C.byref(p)) * clips
        cap = L.x3_encode_bound(npc, C.byref(p)) * clips
        d_wav = ctx.alloc(2 * stride * clips + 64)
        d_out = ctx.alloc(cap + 16)
        d_off = ctx.alloc(8 * (F + 1))
        d_back0 = ctx.alloc(2 * stride * clips + 64)
        d_back = d_back0 + shift
        try:
            ctx.synth_dev(2, 0x5833 + npc, 0, stride * clips, d_wav)
            assert ctx.encode_dev(d_wav, npc, p, d_out, cap, 0, d_off, n_clips=clips, clip_stride=stride) == 0
            assert ctx.encode_result()[0] == 0
            ctx.upload(d_back0, np.zeros(stride * clips + 32, dtype=np.int16))
            assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, stride * clips, n_per_clip=npc, n_clips=clips, clip_stride=stride) == 0
            assert ctx.decode_result()[:3] == (0, F, 0)
            assert ctx.get_option("decode_kernel_in_use") == 3
            src = ctx.download(d_wav, 2 * stride * clips, np.int16).reshape(clips, stride)
            got = ctx.download(d_back, 2 * stride * clips, np.int16).reshape(clips, stride)
            assert np.array_equal(got[:, :npc], src[:, :npc]), (npc, clips, stride, shift)
            assert not got[:, npc:].any(), "samples written between the clips"
        finally:
            for d in (d_wav, d_out, d_off, d_back0):
                ctx.free(d)


def test_device_resident_stream_in_one_trip(ctx, x3):
    """x3_decode_stream_dev: the frame walk's kernels in front, the frame count read from device memory (d_nf)"""
    p = x3.Params.default()
    for n in (10_000 * 40 + 17, 10_000 * 700, 1_234_567):
        wav = x3.synth(2, 9000 + n % 97, 0, n)
        stream = O.encode(wav)[1]
        d_x3 = ctx.alloc(stream.size + 64)
        d_wav = ctx.alloc(2 * n + 64)
        try:
            ctx.upload(d_x3, np.concatenate([stream, np.zeros(64, dtype=np.uint8)]))
            before = ctx.get_option("stream_one_trip")
            rc, n_out, fok, ferr = ctx.decode_stream_dev(d_x3, stream.size, p, d_wav, n)
            assert (rc, n_out, ferr) == (0, n, 0), (rc, n_out, fok, ferr)
            assert ctx.get_option("decode_kernel_in_use") == 3
            assert np.array_equal(ctx.download(d_wav, 2 * n, np.int16), wav)
            assert ctx.get_option("stream_one_trip") == before + 1
        finally:
            ctx.free(d_x3)
            ctx.free(d_wav)


def test_whole_decoder_parity_suite_with_the_blocks_kernel_under_guard_pages():
    """every GPU parity test again, in a child that starts with X3HIP_DECODE_BLOCKS=1 (every context takes the block-per-lane
    kernel wherever the three-wave kernel would run) and X3HIP_FENCE=16 (guard pages behind every device buffer)"""
    env = dict(os.environ, X3HIP_DECODE_BLOCKS="1", X3HIP_FENCE="16", X3HIP_FENCE_FILL="165")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_gpu_bitreader_exact.py"),
                        "-k", "not config5"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, "the parity suite with the block-per-lane decoder ended with %d:\n%s" % (r.returncode, tail)


def test_codeword_that_begins_in_the_payloads_last_bits(ctx, x3):
    """Found by round 6's soak (seed 701, trial 120377; tools/r6/repro_overread.py): a frame whose header asks for one sample
    more than it holds; the extra codeword's zero run begins in the payload's last bit.  Walker and decoders must parse it on
    the SAME bytes (the stream as it comes, its last chunk repeating), see that it reads behind the payload, and leave the
    frame to the reference's reader -- whatever stands behind the stream in the buffer."""
    stream = np.frombuffer(bytes.fromhex(
        "78330101003c000c0000000000000000259eac0900007ffffdfffff7ffff800078330101003d002c00000000000000005a2b611700007ffffcb0"
        "0000000000000000000000000000000000000000000000000000000082d826984f3b7104f704"), dtype=np.uint8)
    assert stream.size == 96
    p = x3.Params.make(20, 3)
    o = O.decode_stream(stream, O.Params.make(20, 3), wav_cap=200)
    assert (o[0], o[1].size) == (0, 121) and o[1][116:121].tolist() == [-2004, -2006, -2006, -2002, -2002]
    d_off = ctx.alloc(8 * 3); d_wo = ctx.alloc(16); d_back = ctx.alloc(2 * 256); d_x3 = ctx.alloc(256)
    ctx.upload(d_off, np.array([0, 32, 96], dtype=np.uint64)); ctx.upload(d_wo, np.array([0, 60], dtype=np.uint64))
    ctx.set_option("wav_offsets_x4", 1)
    try:
        for fill in (0x00, 0xFF, 0x80, 0x55, 0x78, 0x01, 0xA5):
            buf = np.full(256, fill, dtype=np.uint8); buf[:96] = stream
            ctx.upload(d_x3, buf)
            for blocks in (1, 0):
                ctx.set_option("decode_blocks", blocks)
                ctx.upload(d_back, np.zeros(256, dtype=np.int16))
                assert ctx.decode_dev(d_x3, 96, d_off, 2, p, d_back, 200, d_wav_offsets=d_wo) == 0
                assert ctx.decode_result() == (0, 2, 0, 121)
                assert ctx.get_option("decode_kernel_in_use") == (3 if blocks else 2)
                assert np.array_equal(ctx.download(d_back, 2 * 121, np.int16), o[1]), (fill, blocks)
    finally:
        ctx.set_option("decode_blocks", 1)
        ctx.set_option("wav_offsets_x4", 0)
        for d in (d_off, d_wo, d_back, d_x3):
            ctx.free(d)


@pytest.mark.parametrize("bl,bpf", [(40, 500), (40, 250), (40, 1), (40, 3), (40, 37), (10, 500), (10, 1000), (10, 1), (10, 7), (10, 64)])
def test_block_lengths_10_and_40_take_the_blocks_kernel(x3, bl, bpf):
    """block lengths 10 and 40: a lane's unit is 10 or 20 samples, a block of 40 is two units -- the
    block-per-lane kernel is the DEFAULT decoder of such streams (rows on 16-byte boundaries).  Against the oracle and the
    single-wave kernels (option decode_blocks_off), on good and damaged streams."""
    c = x3.Context(0)
    try:
        p = x3.Params.make(bl, bpf)
        po = O.Params.make(bl, bpf)
        spf = bl * bpf
        rng = np.random.default_rng(1000 * bl + bpf)
        for nfr, tail, kind in ((1, 0, 2), (2, 1, 2), (9, spf // 2 + 1, 4), (70, 0, 2), (131, 3, 1), (260, 0, 3)):
            n = nfr * spf + tail
            if n > 3_000_000:
                n = 3_000_000 // spf * spf + tail
            wav = x3.synth(kind, 6100 + bl + bpf + nfr, 0, n)
            rc, stream, _ = O.encode(wav, po)
            assert rc == 0
            cases = [stream]
            offs = frame_offsets(stream)
            for trial in range(4):
                s2 = stream.copy()
                fi = int(rng.integers(0, len(offs)))
                plen = int(s2[offs[fi] + 6]) << 8 | int(s2[offs[fi] + 7])
                if plen > 14:
                    pos = offs[fi] + 22 + int(rng.integers(0, plen - 12))
                    if trial % 2:
                        s2[pos] ^= 1 << int(rng.integers(0, 8))
                    else:
                        s2[pos:pos + 4] = rng.integers(0, 256, size=4, dtype=np.uint8)
                    refresh_crcs(x3, s2, offs[fi])
                    cases.append(s2)
            for s2 in cases:
                o = O.decode_stream(s2, po, wav_cap=n + 100)
                r = c.decode_stream(s2, p, wav_cap=n + 100)
                used = c.get_option("decode_kernel_in_use")
                c.set_option("decode_blocks_off", 1)
                r1 = c.decode_stream(s2, p, wav_cap=n + 100)
                assert c.get_option("decode_kernel_in_use") != 3
                c.set_option("decode_blocks_off", 0)
                for a in (r, r1):
                    assert (a[0], a[2], a[3]) == (o[0], o[2], o[3]) and np.array_equal(a[1], o[1]), (bl, bpf, nfr, tail)
                # (rows on 8-byte boundaries -- frames of a multiple of 4 samples: always for blocks of 40, for blocks of 10
                # when the frame has an even number of blocks)
                # (a stream whose first frame is refused -- white noise in frames of 20 000 samples: payloads beyond the walk's
                # 24 KB -- launches no decoder at all)
                if s2 is stream and (spf % 4) == 0 and o[2] > 0:
                    assert used == 3, (bl, bpf, used)
    finally:
        c.close()
