"""Pin the CPU oracle (oracle/x3_oracle.c) against every known-answer vector the reference's
own unit tests hold for the encode/decode path.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(G, name)))


# ---------------------------------------------------------------- tables (src/x3.rs:200-252)

def test_rice_tables_match_reference_literals():
    L = O.lib()
    g = load("rice_tables.json")
    inv = (C.c_int16 * 60).in_dll(L, "X3O_INV_RICE")
    assert list(inv) == g["inv"]
    rice = (O.RiceCode * 4).in_dll(L, "X3O_RICE")
    for k, t in enumerate(g["tables"]):
        rc = rice[k]
        assert (rc.nsubs, rc.offset, rc.inv_len, rc.len) == (t["nsubs"], t["offset"], t["inv_len"], len(t["code"]))
        assert list(rc.code)[: rc.len] == t["code"]
        assert list(rc.num_bits)[: rc.len] == t["num_bits"]


# ---------------------------------------------------------------- crc (src/crc.rs:78-105)

def test_crc_kat():
    for c in load("crc_kat.json")["cases"]:
        assert O.crc16(bytes(c["bytes"])) == c["crc"]


def test_crc_header_roundtrip():
    L = O.lib()
    h = bytes(load("crc_kat.json")["header20"])
    fh = O.FrameHeader()
    buf = np.frombuffer(h, dtype=np.uint8).copy()
    assert L.x3o_read_frame_header(buf.ctypes.data, 20, C.byref(fh)) == 0
    assert (fh.source_id, fh.channels, fh.samples, fh.payload_len, fh.payload_crc) == (1, 1, 0x2710, 0x19D0, 0x6F61)
    out = np.zeros(20, dtype=np.uint8)
    L.x3o_write_frame_header(0x2710, 1, 0x19D0, 0x6F61, out.ctypes.data)
    assert bytes(out) == h


# ---------------------------------------------------------------- bitpacker (src/bitpacker.rs:196-289)

def test_bitpacker_kat():
    L = O.lib()
    for c in load("bitpacker_kat.json")["cases"]:
        arr = np.array(c["init"], dtype=np.uint8)
        w = O.Writer()
        L.x3o_writer_init(C.byref(w), arr.ctypes.data, arr.size)
        bp = O.BitPacker()
        L.x3o_bp_new(C.byref(bp), C.byref(w))
        for v, n in c["writes"]:
            assert L.x3o_bp_write_bits(C.byref(bp), v, n) == 0
        assert L.x3o_bp_drop(C.byref(bp)) == 0
        assert arr.tolist() == c["expected"]


# ---------------------------------------------------------------- bitreader (src/bitreader.rs:195-303)

def test_bitreader_kat():
    L = O.lib()
    for t in load("bitreader_kat.json")["traces"]:
        arr = np.array(t["bytes"], dtype=np.uint8)
        br = O.BitReader()
        L.x3o_br_new(C.byref(br), arr.ctypes.data, arr.size)
        assert br.rem_bit == t["init"]["rem_bit"]
        assert br.leading_word == t["init"]["leading_word"]
        for op in t["ops"]:
            if op["op"] == "zeros":
                r = L.x3o_br_count_zero_bits(C.byref(br))
            else:
                r = L.x3o_br_read_nbits(C.byref(br), op["n"])
            assert r == op["result"], (t["name"], op)
            if "rem_bit" in op:
                assert br.rem_bit == op["rem_bit"], (t["name"], op)
            assert br.leading_word == op["leading_word"], (t["name"], op)


# ---------------------------------------------------------------- encoder (src/encoder.rs:341-620)

def test_encode_frame_kat():
    L = O.lib()
    for f in load("encoder_kat.json")["frames"]:
        wav = np.array(f["wav"], dtype=np.int16)
        out = np.zeros(0x0EFF * 2, dtype=np.uint8)  # NUM_SAMPLES*2 as in the reference test
        w = O.Writer()
        L.x3o_writer_init(C.byref(w), out.ctypes.data, out.size)
        stats = np.zeros(6, dtype=np.uint64)
        p = O.Params.default()
        assert L.x3o_encode_frame(wav.ctypes.data, wav.size, C.byref(w), C.byref(p), stats.ctypes.data) == 0
        assert out[: w.p_byte].tolist() == f["expected"], f["name"]
        assert int(stats.sum()) == wav.size - 1
        if f["name"] == "test_encode_frame":
            # block-type coverage of the 1000-sample vector: 41 Rice3, 2 Rice1, 7 BFP
            assert stats.tolist() == [0, 2 * 20, 0, 40 * 20 + 19, 7 * 20, 0]
        else:
            assert stats.tolist() == [19, 0, 0, 0, 0, 0]


def test_encode_stream_equals_frame_kat():
    """encode() of <= one frame of samples is exactly encode_frame()."""
    for f in load("encoder_kat.json")["frames"]:
        rc, out, _ = O.encode(np.array(f["wav"], dtype=np.int16))
        assert rc == 0 and out.tolist() == f["expected"]


def test_encode_block_kat():
    L = O.lib()
    for b in load("encoder_kat.json")["blocks"]:
        wav = np.array(b["wav"], dtype=np.int16)
        out = np.zeros(0x0EFF * 2 + 1, dtype=np.uint8)
        w = O.Writer()
        L.x3o_writer_init(C.byref(w), out.ctypes.data, out.size)
        bp = O.BitPacker()
        L.x3o_bp_new(C.byref(bp), C.byref(w))
        if b["prepad_zero_bits"]:
            assert L.x3o_bp_write_packed_zeros(C.byref(bp), b["prepad_zero_bits"]) == 0
        p = O.Params.default()
        ft = C.c_size_t(0)
        blk = wav[1:].copy()
        assert L.x3o_encode_block(blk.ctypes.data, blk.size, int(wav[0]), C.byref(bp), C.byref(p), C.byref(ft)) == 0
        assert L.x3o_bp_word_align(C.byref(bp)) == 0
        assert out[: bp.byte_len].tolist() == b["expected"], b["name"]


# ---------------------------------------------------------------- decoder (src/decoder.rs:257-355)

def test_decode_block_kat():
    L = O.lib()
    for b in load("decoder_kat.json")["blocks"]:
        x3 = np.array(b["x3_inp"], dtype=np.uint8)
        br = O.BitReader()
        if b["first_sample_in_stream"]:
            last = C.c_int16(int(np.frombuffer(x3[:2].tobytes(), dtype=">i2")[0]))
            body = x3[2:].copy()
        else:
            last = C.c_int16(b["last_wav"])
            body = x3
        L.x3o_br_new(C.byref(br), body.ctypes.data, body.size)
        if b["skip_bits"]:
            L.x3o_br_read_nbits(C.byref(br), b["skip_bits"])
        wav = np.zeros(b["block_len"], dtype=np.int16)
        p = O.Params.default()
        assert L.x3o_decode_block(C.byref(br), wav.ctypes.data, wav.size, C.byref(last), C.byref(p)) == 0
        n = len(b["expected_wav"])
        assert wav[:n].tolist() == b["expected_wav"], b["name"]


def test_decode_of_encoder_kat_roundtrips():
    for f in load("encoder_kat.json")["frames"]:
        x3 = np.array(f["expected"], dtype=np.uint8)
        rc, wav, fok, ferr = O.decode_stream(x3)
        assert (rc, fok, ferr) == (0, 1, 0)
        assert wav.tolist() == f["wav"]


def test_oracle_decode_frame_with_block_len_zero():
    """Parameters.block_len == 0 (parse_xml accepts <BLKLEN>0</BLKLEN>; nothing in the reference tests it -- followed
    from source): decode_frame (decoder.rs:36-58) then hands decode_block EMPTY slices and `remaining_samples` never
    moves.  Rice blocks (:147-196) loop over `0..wav.len()` = nothing and return Ok; decode_bpf_block (:209-235) reads
    its exponent, returns FrameDecodeInvalidBPF for E + 1 <= 5 and otherwise panics on `wav[wav.len() - 1]`; behind the
    payload the reader yields zeros, i.e. a BFP block with exponent 0."""
    import numpy as np
    p0 = O.Params.make(0, 500)
    INVALID_BPF, BAD_ARG = 20, 24

    def run(bits, samples=100):
        bits = bits + "0" * (-len(bits) % 8)
        payload = np.array([0x12, 0x34] + [int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)], dtype=np.uint8)
        return O.decode_frame(payload, samples, p0)[0]

    assert run("01" "10" "11" "00" "0011") == INVALID_BPF          # three Rice types, then BFP with E + 1 = 4
    assert run("00" "1111") == BAD_ARG                             # literal block (E + 1 = 16) of no samples: panic
    assert run("11" * 7 + "00" "0101") == BAD_ARG                  # BFP with E + 1 = 6 of no samples: panic
    assert run("01" * 64) == INVALID_BPF                           # Rice types to the end, then the zeros behind it
    assert O.decode_frame(np.array([0x12, 0x34, 0xFF], dtype=np.uint8), 1, p0) [0] == 0   # one sample: no block at all


def test_multichannel_extension_of_the_oracle():
    """x3o_encode_mc / x3o_decode_stream_mc (the extension's definition; the reference stops at MoreThanOneChannel): with ONE
    channel they are the reference-pinned mono functions byte for byte; with more, a round trip, the header's channel
    byte, and the mono reader's refusal"""
    rng = np.random.default_rng(8)
    n = 25003
    walk = np.cumsum(rng.integers(-9, 10, n)).astype(np.int16)
    noise = rng.integers(-300, 300, n).astype(np.int16)
    loud = rng.integers(-32768, 32768, n).astype(np.int16)
    for w in (walk, noise, loud):
        rc1, a, st1 = O.encode(w)
        rc2, b, st2 = O.encode_mc([w])
        assert rc1 == rc2 == 0 and np.array_equal(a, b) and st1.tolist() == st2.tolist()
    p = O.Params.default()
    p.blocks_per_frame = 100   # 2 000-sample frames: three channels of literal blocks stay under the reader's 24 KB
    rc, x, st = O.encode_mc([walk, noise, loud], p)
    assert rc == 0 and x[3] == 3 and int(st.sum()) == 3 * (n - 13)   # 13 frames: their first samples are not in a block
    rc, back, fok, ferr = O.decode_stream_mc(x, 3, p, wav_cap=n + 8)
    assert (rc, fok, ferr) == (0, 13, 0)
    assert all(np.array_equal(a, b) for a, b in zip(back, (walk, noise, loud)))
    assert O.decode_stream(x, p, wav_cap=n + 8)[0] == 6          # X3Error::MoreThanOneChannel, as the crate
    assert O.encode_mc([walk, noise, loud])[0] == 10              # default frames: 60 KB payloads -> FrameLength
