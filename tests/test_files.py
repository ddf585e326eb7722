"""The file level: encodefile::wav_to_x3a / decodefile::x3a_to_wav.

CPU part: the oracle's restatement against itself and against the formats the source spells out (the
reference's own file tests are commented out, so nothing pins these bytes).  GPU part: the streaming
pipeline of libx3hip.so against the oracle, file for file and byte for byte, with chunk sizes small enough
that every file crosses many chunk boundaries on several workers."""
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
import x3hip


def write_wav(path, wav, rate, channels=1, bits=16, extra_chunks=(), extensible=False, data_len=None, fmt_tag=1):
    data = np.ascontiguousarray(wav, dtype="<i2").tobytes()
    if extensible:
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits,
                          22, bits, 4) + struct.pack("<H", fmt_tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xAA\x00\x38\x9B\x71"
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    for cid, payload in extra_chunks:
        body += cid + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    body += b"data" + struct.pack("<I", len(data) if data_len is None else data_len) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read(path):
    with open(path, "rb") as f:
        return f.read()


def test_oracle_wav_header_is_canonical():
    h = bytes(O.wav_header(48000, 1000))
    assert h == (b"RIFF" + struct.pack("<I", 36 + 2000) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 48000, 96000, 2, 16)
                 + b"data" + struct.pack("<I", 2000))
    assert O.wav_parse(np.frombuffer(h + bytes(2000), dtype=np.uint8)) == (0, 48000, 1, 16, 44, 2000)


def test_oracle_wav_parse_variants(tmp_path):
    wav = np.arange(100, dtype=np.int16)
    p = str(tmp_path / "a.wav")
    write_wav(p, wav, 8000, extra_chunks=[(b"LIST", b"abc"), (b"fact", struct.pack("<I", 100))])
    rc, rate, ch, bits, off, dlen = O.wav_parse(np.frombuffer(read(p), dtype=np.uint8))
    assert (rc, rate, ch, bits, dlen) == (0, 8000, 1, 16, 200) and off == 12 + 24 + 12 + 12 + 8
    write_wav(p, wav, 8000, extensible=True)
    assert O.wav_parse(np.frombuffer(read(p), dtype=np.uint8))[:4] == (0, 8000, 1, 16)
    write_wav(p, wav, 8000, fmt_tag=3)  # IEEE float
    assert O.wav_parse(np.frombuffer(read(p), dtype=np.uint8))[0] == 24
    assert O.wav_parse(np.frombuffer(b"RIFX" + read(p)[4:], dtype=np.uint8))[0] == 24
    assert O.wav_parse(np.frombuffer(read(p)[:30], dtype=np.uint8))[0] == 1


def test_oracle_file_roundtrip(tmp_path):
    for kind, n, rate in ((2, 54321, 192000), (1, 10000, 8000), (4, 1, 44100), (0, 0, 96000)):
        wav = x3hip.synth(kind, 40 + kind, 0, n) if n else np.zeros(0, dtype=np.int16)
        a, b, c = (str(tmp_path / s) for s in ("in.wav", "mid.x3a", "out.wav"))
        write_wav(a, wav, rate)
        rc, stats = O.wav_to_x3a(a, b)
        assert rc == 0 and int(stats.sum()) == max(n - (n + 9999) // 10000, 0)
        rc2, x3a, _ = O.x3a_encode(wav, rate)
        assert rc2 == 0 and read(b) == bytes(x3a)
        rc, ns, ferr = O.x3a_to_wav(b, c)
        assert (rc, ns, ferr) == (0, n, 0)
        assert read(c) == bytes(O.wav_header(rate, n)) + wav.astype("<i2").tobytes()


def test_oracle_file_errors(tmp_path):
    a, b = str(tmp_path / "in.wav"), str(tmp_path / "out.x3a")
    assert O.wav_to_x3a(str(tmp_path / "missing.wav"), b)[0] == 1
    wav = np.arange(50, dtype=np.int16)
    write_wav(a, wav, 8000, channels=2)
    assert O.wav_to_x3a(a, b)[0] == 24
    write_wav(a, wav.astype(np.int8), 8000, bits=8)
    assert O.wav_to_x3a(a, b)[0] == 24
    write_wav(a, wav, 8000)
    assert O.wav_to_x3a(a, str(tmp_path / "nodir" / "x.x3a"))[0] == 1
    assert O.x3a_to_wav(str(tmp_path / "missing.x3a"), a)[0] == 1
    with open(b, "wb") as f:
        f.write(b"NOTANX3A" + bytes(100))
    assert O.x3a_to_wav(b, str(tmp_path / "o.wav"))[0] == 9 and not os.path.exists(str(tmp_path / "o.wav"))


# ------------------------------------------------------------------ GPU: the streaming pipeline vs the oracle

class _opt:
    """options of the context (x3_ctx_set_option) for a with-block"""

    def __init__(self, ctx, **kv):
        self.ctx, self.kv = ctx, kv

    def __enter__(self):
        self.old = {k: self.ctx.get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            self.ctx.set_option(k, v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            self.ctx.set_option(k, v)


@pytest.fixture(scope="module")
def ctx():
    c = x3hip.Context(0)
    yield c
    c.close()


def both_ways(ctx, tmp_path, wav_bytes_or_array, rate, tag, expect_rc=0):
    a = str(tmp_path / (tag + ".wav"))
    if isinstance(wav_bytes_or_array, np.ndarray):
        write_wav(a, wav_bytes_or_array, rate)
    else:
        with open(a, "wb") as f:
            f.write(wav_bytes_or_array)
    bo, bg = str(tmp_path / (tag + "_o.x3a")), str(tmp_path / (tag + "_g.x3a"))
    rco, so = O.wav_to_x3a(a, bo)
    rcg, sg = ctx.wav_to_x3a(a, bg)
    assert rcg == rco == expect_rc, (rcg, rco, ctx.last_error())
    if os.path.exists(bo):
        assert read(bg) == read(bo) and sg.tolist() == so.tolist()
    else:
        assert not os.path.exists(bg)
    return bo


def decode_both(ctx, tmp_path, x3a_path, tag):
    co, cg = str(tmp_path / (tag + "_o.wav")), str(tmp_path / (tag + "_g.wav"))
    ro = O.x3a_to_wav(x3a_path, co)
    rg = ctx.x3a_to_wav(x3a_path, cg)
    assert rg == ro, (rg, ro, ctx.last_error())
    assert os.path.exists(cg) == os.path.exists(co)
    if os.path.exists(co):
        assert read(cg) == read(co)
    return ro


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_frames,workers", [(1, 3), (3, 2), (7, 4), (3200, 3), (2, 1)])
def test_files_match_oracle(ctx, tmp_path, chunk_frames, workers):
    with _opt(ctx, file_chunk_frames=chunk_frames, file_workers=workers):
        for kind, n, rate in ((2, 234567, 192000), (1, 70001, 8000), (4, 30000, 44100), (3, 10000, 96000), (0, 1, 1000),
                              (0, 0, 48000)):
            wav = x3hip.synth(kind, 50 + kind, 0, n) if n else np.zeros(0, dtype=np.int16)
            tag = "k%d_%d" % (kind, n)
            x3a = both_ways(ctx, tmp_path, wav, rate, tag)
            assert decode_both(ctx, tmp_path, x3a, tag) == (0, n, 0)
            assert read(str(tmp_path / (tag + "_g.wav")))[44:] == wav.astype("<i2").tobytes()


@pytest.mark.gpu
def test_file_workers_share_the_single_pass_encoder(ctx, tmp_path):
    """wav -> x3a over four workers: the workers' persistent-grid encoders take turns (one gate per pipeline, round 3)
    instead of colliding -- no launch gives up its size wait (encode_fallbacks stays) -- and the archive is the oracle's"""
    wav = x3hip.synth(2, 77, 0, 1_500_000)
    before = ctx.get_option("encode_fallbacks")
    with _opt(ctx, file_chunk_frames=16, file_workers=4):   # 10 chunks of 160 000 samples
        both_ways(ctx, tmp_path, wav, 192000, "gate")
    assert ctx.get_option("encode_fallbacks") == before


@pytest.mark.gpu
def test_wav_container_variants(ctx, tmp_path):
    wav = x3hip.synth(2, 61, 0, 25000)
    a = str(tmp_path / "v.wav")
    with _opt(ctx, file_chunk_frames=1):
        write_wav(a, wav, 22050, extra_chunks=[(b"LIST", b"odd"), (b"bext", bytes(40))])
        both_ways(ctx, tmp_path, read(a), 22050, "chunks")
        write_wav(a, wav, 22050, extensible=True)
        both_ways(ctx, tmp_path, read(a), 22050, "ext")
        # rejected formats and broken files: same code, no output where the reference never creates one
        write_wav(a, wav, 22050, channels=2)
        both_ways(ctx, tmp_path, read(a), 22050, "stereo", expect_rc=x3hip.ERR_BAD_ARG)
        write_wav(a, wav.astype(np.int8), 22050, bits=8)
        both_ways(ctx, tmp_path, read(a), 22050, "u8", expect_rc=x3hip.ERR_BAD_ARG)
        write_wav(a, wav, 22050, fmt_tag=3)
        both_ways(ctx, tmp_path, read(a), 22050, "float", expect_rc=x3hip.ERR_BAD_ARG)
        both_ways(ctx, tmp_path, b"RIFF\x04\x00\x00\x00WAVE", 0, "nodata", expect_rc=1)
        both_ways(ctx, tmp_path, b"junk", 0, "junk", expect_rc=1)
        # the data chunk promises more samples than the file holds: what is there is encoded, then Io
        write_wav(a, wav, 22050, data_len=2 * wav.size + 4000)
        both_ways(ctx, tmp_path, read(a), 22050, "short", expect_rc=1)
        write_wav(a, wav, 22050, data_len=2 * wav.size - 4001)
        both_ways(ctx, tmp_path, read(a), 22050, "oddlen", expect_rc=x3hip.ERR_BAD_ARG)
    assert ctx.wav_to_x3a(str(tmp_path / "missing.wav"), str(tmp_path / "m.x3a"))[0] == 1
    write_wav(a, wav, 22050)
    assert ctx.wav_to_x3a(a, str(tmp_path / "nodir" / "m.x3a"))[0] == 1
    assert ctx.x3a_to_wav(str(tmp_path / "missing.x3a"), a)[0] == 1


def _frames(x3a):
    """frame offsets of the audio frames of an archive (bytes)"""
    hlen = 28 + (x3a[14] << 8 | x3a[15])
    offs, pos = [], hlen
    while pos + 20 < len(x3a):
        offs.append(pos)
        pos += 20 + (x3a[pos + 6] << 8 | x3a[pos + 7])
    return offs


def _refresh(b, off):
    plen = b[off + 6] << 8 | b[off + 7]
    pcrc = O.crc16(np.frombuffer(bytes(b[off + 20: off + 20 + plen]), dtype=np.uint8))
    b[off + 18], b[off + 19] = pcrc >> 8, pcrc & 0xFF
    hcrc = O.crc16(np.frombuffer(bytes(b[off: off + 16]), dtype=np.uint8))
    b[off + 16], b[off + 17] = hcrc >> 8, hcrc & 0xFF


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_frames,workers", [(1, 3), (4, 2), (3200, 3)])
def test_broken_archives_match_oracle(ctx, tmp_path, chunk_frames, workers):
    wav = x3hip.synth(2, 62, 0, 173000)
    rc, x3a, _ = O.x3a_encode(wav, 96000)
    good = bytearray(bytes(x3a))
    offs = _frames(good)
    assert len(offs) == 18
    cases = {}
    for fi in (0, 5, 17):
        b = bytearray(good); b[offs[fi] + 20 + 77] ^= 4; cases["crc%d" % fi] = b          # payload CRC: hard error
        b = bytearray(good); b[offs[fi] + 30: offs[fi] + 44] = bytes(14); _refresh(b, offs[fi])
        cases["zeros%d" % fi] = b                                                           # decode error: counted, quiet
        b = bytearray(good); b[offs[fi] + 1] ^= 1; cases["hdr%d" % fi] = b                 # header CRC
        b = bytearray(good); b[offs[fi]] = 0x79; _refresh(b, offs[fi]); cases["key%d" % fi] = b
    cases["cut_in_payload"] = good[: offs[9] + 300]
    cases["cut_in_header"] = good[: offs[9] + 11]
    cases["cut_at_frame"] = good[: offs[9]]
    cases["cut_28_after"] = good[: offs[9] + 28]       # the reader's 8 phantom bytes make these differ
    cases["cut_20_after"] = good[: offs[9] + 20]
    cases["cut_12_after"] = good[: offs[9] + 12]
    cases["tail_garbage"] = good + bytes(range(50))
    cases["tail_short"] = good + bytes(13)
    cases["header_only"] = good[: offs[0]]
    cases["header_cut"] = good[:100]
    cases["empty_file"] = b""                  # (soak seed 44: the archive id's read_exact fails -- Io, not a null-pointer BAD_ARG)
    cases["cut_in_archive_id"] = good[:5]
    cases["cut_in_archive_header"] = good[:20]
    b = bytearray(good); b[offs[3] + 4] = 0; b[offs[3] + 5] = 0; _refresh(b, offs[3]); cases["zero_samples"] = b
    b = bytearray(good); b[40] ^= 1; cases["xml_crc"] = b  # XML payload CRC is not checked by the reader
    with _opt(ctx, file_chunk_frames=chunk_frames, file_workers=workers):
        seen = set()
        for name, b in cases.items():
            p = str(tmp_path / (name + ".x3a"))
            with open(p, "wb") as f:
                f.write(bytes(b))
            seen.add(decode_both(ctx, tmp_path, p, name)[0])
        assert {0, 1, x3hip.ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC, x3hip.ERR_FRAME_HEADER_INVALID_HEADER_CRC} <= seen


@pytest.mark.gpu
def test_cli_round_trip(tmp_path):
    """the reference's command line (src/bin/x3.rs): x3 -i a.wav -o a.x3a, then back"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "x3-rust_amd", "bin", "x3")
    assert os.path.exists(exe), "x3-rust_amd/bin/x3 is missing: run python x3-rust_amd/build.py"
    wav = x3hip.synth(2, 63, 0, 48000)
    a, b, c, bo = (str(tmp_path / s) for s in ("a.wav", "a.x3a", "back.wav", "o.x3a"))
    write_wav(a, wav, 48000)
    r = subprocess.run([exe, "-i", a, "-o", b], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Statistics:" in r.stdout and "Rice-0:" in r.stdout and "Pass-through" in r.stdout
    assert O.wav_to_x3a(a, bo)[0] == 0 and read(b) == read(bo)
    r = subprocess.run([exe, "--input", b, "--output", c], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert read(c) == read(a)
    # the reference panics on these (exit status 101)
    assert subprocess.run([exe, "-i", a, "-o", str(tmp_path / "b.wav")], capture_output=True, timeout=60).returncode == 101
    assert subprocess.run([exe, "-i", a, "-o", str(tmp_path / "b.flac")], capture_output=True, timeout=60).returncode == 101
    assert subprocess.run([exe, "-i", str(tmp_path / "none.wav"), "-o", b], capture_output=True, timeout=60).returncode == 101
    assert subprocess.run([exe, "-i", a], capture_output=True, timeout=60).returncode == 2
