"""The .x3a archive around the frame stream: header writer/reader and the in-memory
wav <-> x3a conversions, product library vs the oracle's restatement of encodefile.rs / decodefile.rs.
No reference test pins these bytes (its file tests are commented out): the expected header below is
assembled from the format the source spells out (encodefile.rs:88-138)."""
import numpy as np
import pytest

import oracle_lib as O
import x3hip


def expected_header(rate, p):
    xml = ('<X3ARCH PROG="x3new.m" VERSION="2.0" /><CFG ID="0" FTYPE="XML" /><CFG ID="1" FTYPE="WAV">'
           '<FS UNIT="Hz">%d</FS><SUFFIX>wav</SUFFIX><CODEC TYPE="X3" VERS="2"><BLKLEN>%d</BLKLEN>'
           '<CODES N="4">RICE%d,RICE%d,RICE%d,BFP</CODES><FILTER>DIFF</FILTER><NBITS>16</NBITS>'
           '<T N="3">%d,%d,%d</T></CODEC></CFG>' % ((rate, p.block_len) + tuple(p.codes) + tuple(p.thresholds))).encode()
    if len(xml) % 2:
        xml += b"\0"
    hdr = x3hip.write_frame_header(0, 0, len(xml), O.crc16(xml))
    return b"X3ARCHIV" + bytes(hdr) + xml


def test_archive_header_bytes():
    for rate in (8000, 44100, 96000, 192000, 1234567):
        for p, po in ((x3hip.Params.default(), O.Params.default()),
                      (x3hip.Params.make(40, 500, (1, 2, 3), (5, 10, 25)), O.Params.make(40, 500, (1, 2, 3), (5, 10, 25)))):
            rc, h = x3hip.archive_header_write(rate, p)
            rco, ho = O.archive_header_write(rate, po)
            assert rc == rco == 0 and bytes(h) == bytes(ho) == expected_header(rate, p)
            assert len(h) % 2 == 0
            r = x3hip.archive_header_read(h)
            ro = O.archive_header_read(h)
            assert r[0] == ro[0] == 0 and r[1] == ro[1] == rate and r[3:] == ro[3:] == (0, len(h) - 8)
            assert (r[2].block_len, list(r[2].codes), list(r[2].thresholds)) == (p.block_len, list(p.codes), list(p.thresholds))
    assert len(x3hip.archive_header_write(192000)[1]) == 320  # first audio frame at byte 320


def test_archive_header_errors_match_oracle():
    rc, h = x3hip.archive_header_write(48000)
    cases = [h[:5], h[:20], h[:100], np.concatenate([np.frombuffer(b"X3ARCHIW", dtype=np.uint8), h[8:]])]
    for mutate in (lambda b: b.replace(b"RICE1", b"RICE7"), lambda b: b.replace(b"<FS UNIT", b"<FX UNIT").replace(b"</FS>", b"</FX>"),
                   lambda b: b.replace(b"48000", b"4800x"), lambda b: b.replace(b"3,8,20", b"7,8,20"),
                   lambda b: b.replace(b"3,8,20", b"3,8   "), lambda b: b.replace(b"RICE0,RICE1,RICE3,BFP", b"RICE0,BFP,RICE3,BFP ")):
        xml = mutate(bytes(h[28:]))
        hdr = x3hip.write_frame_header(0, 0, len(xml), O.crc16(xml))
        cases.append(np.frombuffer(b"X3ARCHIV" + bytes(hdr) + xml, dtype=np.uint8))
    bad_crc = h.copy(); bad_crc[10] ^= 1
    cases.append(bad_crc)
    for c in cases:
        assert x3hip.archive_header_read(c)[0] == O.archive_header_read(c)[0], bytes(c[:40])
    assert x3hip.archive_header_read(cases[3])[0] == 9  # ArchiveHeaderXMLInvalidKey


@pytest.mark.gpu
@pytest.mark.parametrize("host_walk", [1, 0])
def test_x3a_roundtrip_matches_oracle(host_walk):
    ctx = x3hip.Context(0)
    ctx.set_option("host_walk", host_walk)  # the frame walk on the host / on the GPU (x3_index_kernels.h)
    try:
        for kind, n, rate in ((2, 123457, 192000), (4, 10000, 44100), (1, 25001, 8000), (0, 1, 96000)):
            wav = x3hip.synth(kind, 900 + kind, 0, n)
            rc, x3a, stats = ctx.x3a_encode(wav, rate)
            rco, x3ao, statso = O.x3a_encode(wav, rate)
            assert rc == rco == 0 and np.array_equal(x3a, x3ao) and stats.tolist() == statso.tolist()
            r = ctx.x3a_decode(x3a, wav_cap=n)
            ro = O.x3a_decode(x3a, wav_cap=n)
            assert r[0] == ro[0] == 0 and np.array_equal(r[1], wav) and np.array_equal(ro[1], wav)
            assert r[2:] == ro[2:] and r[2] == rate
            # the reader's byte accounting: truncations and trailing bytes (Io where a read runs past the end)
            for cut in (x3a.size - 1, x3a.size - 7, x3a.size - 13, 330, 321, 320, 300):
                if 0 < cut < x3a.size:
                    a, b = ctx.x3a_decode(x3a[:cut].copy(), wav_cap=n), O.x3a_decode(x3a[:cut].copy(), wav_cap=n)
                    assert (a[0], a[2:]) == (b[0], b[2:]) and np.array_equal(a[1], b[1]), cut
            for extra in (1, 11, 12, 13, 19, 20, 33):
                t = np.concatenate([x3a, np.zeros(extra, dtype=np.uint8)])
                a, b = ctx.x3a_decode(t, wav_cap=n), O.x3a_decode(t, wav_cap=n)
                assert (a[0], a[2:]) == (b[0], b[2:]) and np.array_equal(a[1], b[1]), extra
    finally:
        ctx.close()
