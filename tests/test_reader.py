"""x3_reader_* = the reference's incremental `X3aReader` (src/decodefile.rs:47-137): open / spec / decode_next_frame.
A loop over next_frame until the first Ok(None) / Err is what `x3a_to_wav` does (decodefile.rs:200-209), so its result
must equal the oracle's x3a_to_wav on the same bytes -- good archives, broken ones, every window size -- and the
reader must go on behind a frame that failed, as the reference's does.  `pytest -m gpu`."""
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


def read_like_x3a_to_wav(ctx, source):
    r = x3hip.Reader(ctx, source)
    if r.rc:
        return r.rc, np.zeros(0, dtype=np.int16), 0, 0
    out, frames = [], 0
    while True:
        rc, s = r.next_frame()
        if rc or s is None:
            break
        out.append(s)
        frames += 1
    ferr = r.frame_errors()
    rate = r.spec()[0]
    r.close()
    return rc, (np.concatenate(out) if out else np.zeros(0, dtype=np.int16)), ferr, rate


def refresh(b, off):
    plen = b[off + 6] << 8 | b[off + 7]
    c = O.crc16(bytes(b[off + 20:off + 20 + plen])); b[off + 18] = c >> 8; b[off + 19] = c & 0xFF
    c = O.crc16(bytes(b[off:off + 16])); b[off + 16] = c >> 8; b[off + 17] = c & 0xFF


def frame_offsets(b, start):
    offs, pos = [], start
    while pos + 20 < len(b):
        offs.append(pos)
        pos += 20 + (b[pos + 6] << 8 | b[pos + 7])
    return offs


@pytest.mark.parametrize("window", [1, 3, 4096])
def test_reader_equals_x3a_to_wav_on_good_archives(ctx, window, tmp_path):
    ctx.set_option("reader_window_frames", window)
    try:
        for kind, n, rate in ((2, 123457, 192000), (4, 10000, 44100), (1, 25001, 8000), (0, 1, 96000), (3, 0, 48000)):
            wav = x3hip.synth(kind, 900 + kind, 0, n) if n else np.zeros(0, dtype=np.int16)
            rco, x3a, _ = O.x3a_encode(wav, rate)
            assert rco == 0
            rc, got, ferr, r = read_like_x3a_to_wav(ctx, x3a)
            ro = O.x3a_decode(x3a, wav_cap=max(n, 1))
            assert (rc, ferr) == (ro[0], ro[4]) and np.array_equal(got, ro[1]) and np.array_equal(got, wav) and r == rate
            p = tmp_path / ("a%d_%d.x3a" % (kind, n))
            p.write_bytes(bytes(x3a))
            rc2, got2, ferr2, _ = read_like_x3a_to_wav(ctx, str(p))
            assert (rc2, ferr2) == (rc, ferr) and np.array_equal(got2, got)
    finally:
        ctx.set_option("reader_window_frames", 4096)


@pytest.mark.parametrize("window", [2, 4096])
def test_reader_equals_x3a_to_wav_on_broken_archives(ctx, window, tmp_path):
    wav = x3hip.synth(2, 77, 0, 150000)
    good = bytearray(bytes(O.x3a_encode(wav, 192000)[1]))
    offs = frame_offsets(good, 320)
    assert len(offs) == 15
    cases = {}
    for fi in (0, 4, 14):
        b = bytearray(good); b[offs[fi] + 20 + 50] ^= 0x20; cases["pcrc%d" % fi] = b
        b = bytearray(good); b[offs[fi] + 30:offs[fi] + 38] = bytes(8); refresh(b, offs[fi]); cases["dec%d" % fi] = b
        b = bytearray(good); b[offs[fi] + 1] ^= 1; cases["hdr%d" % fi] = b
        b = bytearray(good); b[offs[fi]] = 0x79; refresh(b, offs[fi]); cases["key%d" % fi] = b
    for cut in (300, 11, 0, 28, 20, 12):
        cases["cut%d" % cut] = good[: offs[9] + cut]
    cases["tail_garbage"] = good + bytes(range(50))
    cases["tail_short"] = good + bytes(13)
    cases["header_only"] = good[: offs[0]]
    cases["empty"] = b""                       # (soak seed 44: no bytes at all is a read that fails, not a bad argument)
    cases["cut_in_id"] = good[:5]
    cases["cut_in_header"] = good[:20]
    cases["cut_in_xml"] = good[:100]
    b = bytearray(good); b[offs[3] + 4] = 0; b[offs[3] + 5] = 0; refresh(b, offs[3]); cases["zero_samples"] = b
    b = bytearray(good); b[offs[5] + 6] = 0x70; c = O.crc16(bytes(b[offs[5]:offs[5] + 16])); b[offs[5] + 16] = c >> 8; b[offs[5] + 17] = c & 0xFF
    cases["long_payload"] = b
    ctx.set_option("reader_window_frames", window)
    try:
        seen = set()
        for name, b in cases.items():
            arr = np.frombuffer(bytes(b), dtype=np.uint8)
            rc, got, ferr, _ = read_like_x3a_to_wav(ctx, arr)
            ro = O.x3a_decode(arr, wav_cap=wav.size + 70000)
            assert (rc, ferr) == (ro[0], ro[4]), (name, rc, ferr, ro[0], ro[4])
            assert np.array_equal(got, ro[1]), name
            seen.add(rc)
            if name in ("empty", "cut_in_id", "cut_in_xml"):   # the same from a file (x3_reader_open)
                fp = tmp_path / (name + ".x3a")
                fp.write_bytes(bytes(b))
                rc2, got2, ferr2, _ = read_like_x3a_to_wav(ctx, str(fp))
                assert (rc2, ferr2, got2.size) == (rc, ferr, got.size), (name, rc2, rc)
        assert {0, 1, x3hip.ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC, x3hip.ERR_FRAME_HEADER_INVALID_HEADER_CRC} <= seen
    finally:
        ctx.set_option("reader_window_frames", 4096)


def test_reader_goes_on_behind_a_failed_frame(ctx):
    """decode_next_frame returns Ok(None) for a frame that does not decode and Err for a payload CRC mismatch, but
    in both cases it has consumed the frame: the next call yields the frame behind it (decodefile.rs:93-135)"""
    wav = x3hip.synth(2, 78, 0, 80000)
    good = bytearray(bytes(O.x3a_encode(wav, 48000)[1]))
    offs = frame_offsets(good, 320)
    b = bytearray(good)
    b[offs[2] + 30:offs[2] + 38] = bytes(8); refresh(b, offs[2])     # frame 2: decode error
    b[offs[5] + 20 + 9] ^= 0x04                                        # frame 5: payload CRC
    for window in (1, 3, 4096):
        ctx.set_option("reader_window_frames", window)
        r = x3hip.Reader(ctx, np.frombuffer(bytes(b), dtype=np.uint8))
        assert r.rc == 0
        got = []
        for i in range(8):
            rc, s = r.next_frame()
            got.append((rc, None if s is None else s.copy()))
        assert r.position() == len(b)
        assert r.next_frame() == (0, None)                           # end of the file
        assert r.frame_errors() == 1
        r.close()
        for i, (rc, s) in enumerate(got):
            if i == 2:
                assert (rc, s) == (0, None)
            elif i == 5:
                assert rc == x3hip.ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC
            else:
                assert rc == 0 and np.array_equal(s, wav[i * 10000:(i + 1) * 10000]), i
    ctx.set_option("reader_window_frames", 4096)
