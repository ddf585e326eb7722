"""The reference's known-answer vectors for `BitReader` (src/bitreader.rs:195-303), `BitPacker`
(src/bitpacker.rs:196-289) and `decode_block` (src/decoder.rs:257-355), through the product's own exports
(x3_bitreader_*, x3_bitpacker_*, x3_decode_block: GPU work behind the C ABI), plus randomized comparisons with the
oracle -- reads past the end of the array and zero runs across words included.  `pytest -m gpu`."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import x3hip

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(G, name)))


@pytest.fixture(scope="module")
def ctx():
    c = x3hip.Context(0)
    yield c
    c.close()


class GpuReader:
    def __init__(self, ctx, arr):
        self.L = x3hip.lib()
        self.h = C.c_void_p()
        self.arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert self.L.x3_bitreader_new(ctx._h, self.arr.ctypes.data if self.arr.size else None, self.arr.size, C.byref(self.h)) == 0

    def state(self):
        idx, w, rem = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        self.L.x3_bitreader_state(self.h, C.byref(idx), C.byref(w), C.byref(rem))
        return idx.value, w.value, rem.value

    def read(self, n):
        v = C.c_uint32(0)
        assert self.L.x3_bitreader_read_nbits(self.h, n, C.byref(v)) == 0
        return v.value

    def zeros(self):
        v = C.c_uint32(0)
        assert self.L.x3_bitreader_count_zero_bits(self.h, C.byref(v)) == 0
        return v.value

    def close(self):
        self.L.x3_bitreader_free(self.h)


def test_bitreader_kat(ctx):
    for t in load("bitreader_kat.json")["traces"]:
        br = GpuReader(ctx, t["bytes"])
        _, w, rem = br.state()
        assert rem == t["init"]["rem_bit"] and w == t["init"]["leading_word"], t["name"]
        for op in t["ops"]:
            r = br.zeros() if op["op"] == "zeros" else br.read(op["n"])
            assert r == op["result"], (t["name"], op)
            _, w, rem = br.state()
            if "rem_bit" in op:
                assert rem == op["rem_bit"], (t["name"], op)
            assert w == op["leading_word"], (t["name"], op)
        br.close()


def test_bitreader_random_vs_oracle(ctx):
    """random op sequences over short arrays, far past their end: result and private state equal the oracle's"""
    rng = np.random.default_rng(42)
    OL = O.lib()
    for trial in range(12):
        n = int(rng.integers(0, 14))
        arr = rng.integers(0, 256, size=n, dtype=np.uint8)
        if trial % 3 == 0:
            arr[rng.integers(0, max(n, 1), size=max(n // 2, 0))] = 0   # long zero runs
        br = GpuReader(ctx, arr)
        ob = O.BitReader()
        OL.x3o_br_new(C.byref(ob), arr.ctypes.data if n else None, n)
        assert br.state() == (ob.idx, ob.leading_word, ob.rem_bit)
        for _ in range(40):
            if rng.random() < 0.4:
                a, b = br.zeros(), OL.x3o_br_count_zero_bits(C.byref(ob))
            else:
                k = int(rng.integers(0, 32))
                a, b = br.read(k), OL.x3o_br_read_nbits(C.byref(ob), k)
            assert a == b, (trial, arr.tolist())
            assert br.state() == (ob.idx, ob.leading_word, ob.rem_bit), (trial, arr.tolist())
        br.close()


def test_bitpacker_kat(ctx):
    L = x3hip.lib()
    for c in load("bitpacker_kat.json")["cases"]:
        arr = np.array(c["init"], dtype=np.uint8)
        bp = C.c_void_p()
        assert L.x3_bitpacker_new(ctx._h, arr.ctypes.data, arr.size, 0, C.byref(bp)) == 0
        for v, n in c["writes"]:
            assert L.x3_bitpacker_write_bits(bp, v, n) == 0
        ln, crc, pos = C.c_uint64(0), C.c_uint16(0), C.c_uint64(0)
        assert L.x3_bitpacker_finish(bp, C.byref(ln), C.byref(crc), C.byref(pos)) == 0   # the reference's Drop flushes
        L.x3_bitpacker_free(bp)
        assert arr.tolist() == c["expected"], c
        assert crc.value == O.crc16(arr[:ln.value]) and pos.value == ln.value


def test_bitpacker_random_vs_oracle(ctx):
    """random field lists (widths 0..33, over-wide values, zero runs), odd start positions, word_align: bytes, len()
    and crc() equal the oracle's byte-at-a-time packer"""
    rng = np.random.default_rng(9)
    L, OL = x3hip.lib(), O.lib()
    for trial in range(25):
        start = int(rng.integers(0, 5))
        nf = int(rng.integers(0, 700))
        fields = [(int(rng.integers(0, 1 << 40)), int(rng.integers(0, 34))) for _ in range(nf)]
        cap = start + nf * 5 + 16
        a = np.full(cap, 0, dtype=np.uint8)
        b = np.full(cap, 0, dtype=np.uint8)
        bp = C.c_void_p()
        assert L.x3_bitpacker_new(ctx._h, a.ctypes.data, cap, start, C.byref(bp)) == 0
        w = O.Writer()
        OL.x3o_writer_init(C.byref(w), b.ctypes.data, cap)
        OL.x3o_writer_seek_start(C.byref(w), start)
        ob = O.BitPacker()
        OL.x3o_bp_new(C.byref(ob), C.byref(w))
        ln, crc, pos = C.c_uint64(0), C.c_uint16(0), C.c_uint64(0)
        for i, (v, n) in enumerate(fields):
            if i == nf // 2:   # len() / crc() between writes, then a flush in mid-stream (Drop of a first packer)
                assert L.x3_bitpacker_peek(bp, C.byref(ln), C.byref(crc)) == 0
                assert (ln.value, crc.value) == (ob.byte_len, ob.crc), trial
                if trial % 2:
                    assert L.x3_bitpacker_finish(bp, C.byref(ln), C.byref(crc), C.byref(pos)) == 0
                    OL.x3o_bp_drop(C.byref(ob))
                    assert (ln.value, crc.value, pos.value) == (ob.byte_len, ob.crc, w.p_byte), trial
            if i % 17 == 5:
                assert L.x3_bitpacker_write_packed_zeros(bp, n) == 0
                assert OL.x3o_bp_write_packed_zeros(C.byref(ob), n) == 0
            else:
                assert L.x3_bitpacker_write_bits(bp, v, n) == 0
                assert OL.x3o_bp_write_bits(C.byref(ob), v, n) == 0
        assert L.x3_bitpacker_word_align(bp) == 0
        assert OL.x3o_bp_word_align(C.byref(ob)) == 0
        assert L.x3_bitpacker_finish(bp, C.byref(ln), C.byref(crc), C.byref(pos)) == 0
        L.x3_bitpacker_free(bp)
        assert (ln.value, crc.value, pos.value) == (ob.byte_len, ob.crc, w.p_byte), trial
        assert np.array_equal(a, b), trial


def test_decode_block_kat(ctx):
    L = x3hip.lib()
    p = x3hip.Params.default()
    for b in load("decoder_kat.json")["blocks"]:
        x = np.array(b["x3_inp"], dtype=np.uint8)
        n = len(b["expected_wav"])
        if b["first_sample_in_stream"]:   # the reference's tests read the first sample, then hand the rest to a BitReader
            last = int(np.frombuffer(x[:2].tobytes(), dtype=">i2")[0])
            br = GpuReader(ctx, x[2:])
        else:
            last = b["last_wav"]
            br = GpuReader(ctx, x)
            br.read(b["skip_bits"])
        wav = np.zeros(n, dtype=np.int16)
        lw = C.c_int16(last)
        rc = L.x3_decode_block(br.h, wav.ctypes.data, n, C.byref(lw), C.byref(p))
        assert rc == 0, (b["name"], rc)
        assert wav.tolist() == b["expected_wav"], b["name"]
        assert lw.value == b["expected_wav"][-1]
        br.close()


def test_decode_block_random_and_empty(ctx):
    """decode_block over random bits, n = 0..25 samples (an EMPTY block reads its type bits; a BFP one then fails or hits
    the reference's `wav[wav.len() - 1]` panic): status, samples, last_wav and the reader's state equal the oracle's"""
    rng = np.random.default_rng(31)
    L, OL = x3hip.lib(), O.lib()
    p, op = x3hip.Params.default(), O.Params.default()
    seen = set()
    for trial in range(300):
        x = rng.integers(0, 256, size=int(rng.integers(1, 80)), dtype=np.uint8)
        if trial % 3 == 0:
            x[0] &= 0x3F                                    # ftype 0: BFP / literal
        n = int(rng.choice([0, 0, 1, 2, 19, 20, 25]))
        last = int(rng.integers(-32768, 32768))
        br = GpuReader(ctx, x)
        obr = O.BitReader()
        OL.x3o_br_new(C.byref(obr), x.ctypes.data, x.size)
        wav, owav = np.zeros(max(n, 1), dtype=np.int16), np.zeros(max(n, 1), dtype=np.int16)
        lw, olw = C.c_int16(last), C.c_int16(last)
        rc = L.x3_decode_block(br.h, wav.ctypes.data, n, C.byref(lw), C.byref(p))
        orc = OL.x3o_decode_block(C.byref(obr), owav.ctypes.data, n, C.byref(olw), C.byref(op))
        assert rc == orc, (trial, n, rc, orc, x[:4].tolist())
        seen.add((n == 0, rc))
        if rc == 0:
            assert np.array_equal(wav[:n], owav[:n]) and lw.value == olw.value, (trial, n)
        br.close()
    assert (True, 0) in seen and (True, x3hip.ERR_BAD_ARG) in seen and (True, x3hip.ERR_FRAME_DECODE_INVALID_BPF) in seen, seen


def test_bitpacker_write_bytes_and_inc_counter(ctx):
    """BitPacker::write_bytes (bitpacker.rs:95-102: the array reaches the writer in front of a partial byte still in the
    scratch) and inc_counter_n_bytes (:112-118: the writer skips, len() and crc() stay; NotByteAligned off a boundary)
    mixed into random field lists, against the oracle: the slice (untouched bytes keep their sentinel), len(), crc() and
    the writer's position"""
    rng = np.random.default_rng(31)
    L, OL = x3hip.lib(), O.lib()
    for trial in range(30):
        start = int(rng.integers(0, 5))
        nops = int(rng.integers(1, 120))
        cap = start + nops * 40 + 64
        a = np.full(cap, 0xA5, dtype=np.uint8)
        b = np.full(cap, 0xA5, dtype=np.uint8)
        bp = C.c_void_p()
        assert L.x3_bitpacker_new(ctx._h, a.ctypes.data, cap, start, C.byref(bp)) == 0
        w = O.Writer()
        OL.x3o_writer_init(C.byref(w), b.ctypes.data, cap)
        OL.x3o_writer_seek_start(C.byref(w), start)
        ob = O.BitPacker()
        OL.x3o_bp_new(C.byref(ob), C.byref(w))
        ln, crc, pos = C.c_uint64(0), C.c_uint16(0), C.c_uint64(0)
        for i in range(nops):
            op = int(rng.integers(0, 10))
            if op < 6:
                v, n = int(rng.integers(0, 1 << 40)), int(rng.integers(0, 34))
                assert L.x3_bitpacker_write_bits(bp, v, n) == 0 and OL.x3o_bp_write_bits(C.byref(ob), v, n) == 0
            elif op < 8:
                arr = rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8)
                assert L.x3_bitpacker_write_bytes(bp, arr.ctypes.data, arr.size) == 0
                assert OL.x3o_bp_write_bytes(C.byref(ob), arr.ctypes.data, arr.size) == 0
            elif op == 8:
                k = int(rng.integers(0, 9))
                r_g = L.x3_bitpacker_inc_counter_n_bytes(bp, k)
                r_o = OL.x3o_bp_inc_counter_n_bytes(C.byref(ob), k)
                assert r_g == r_o and r_g in (0, 3), (trial, i, r_g, r_o)   # 3 = BitPack(NotByteAligned)
            else:
                assert L.x3_bitpacker_peek(bp, C.byref(ln), C.byref(crc)) == 0
                assert (ln.value, crc.value) == (ob.byte_len, ob.crc), (trial, i)
                if trial % 3 == 0:   # a flush in mid-stream
                    assert L.x3_bitpacker_finish(bp, C.byref(ln), C.byref(crc), C.byref(pos)) == 0
                    OL.x3o_bp_drop(C.byref(ob))
                    assert (ln.value, crc.value, pos.value) == (ob.byte_len, ob.crc, w.p_byte), (trial, i)
        assert L.x3_bitpacker_word_align(bp) == 0 and OL.x3o_bp_word_align(C.byref(ob)) == 0
        assert L.x3_bitpacker_finish(bp, C.byref(ln), C.byref(crc), C.byref(pos)) == 0
        L.x3_bitpacker_free(bp)
        assert (ln.value, crc.value, pos.value) == (ob.byte_len, ob.crc, w.p_byte), trial
        assert np.array_equal(a, b), (trial, np.nonzero(a != b)[0][:8])
    # a skip beyond the slice is the writer's ByteWriterInsufficientMemory; an unbound packer has no writer to move
    a = np.zeros(16, dtype=np.uint8)
    bp = C.c_void_p()
    assert L.x3_bitpacker_new(ctx._h, a.ctypes.data, 16, 4, C.byref(bp)) == 0
    assert L.x3_bitpacker_inc_counter_n_bytes(bp, 13) == 22 and L.x3_bitpacker_inc_counter_n_bytes(bp, 12) == 0
    L.x3_bitpacker_free(bp)
    assert L.x3_bitpacker_new(ctx._h, None, 0, 0, C.byref(bp)) == 0
    assert L.x3_bitpacker_inc_counter_n_bytes(bp, 1) == 24
    L.x3_bitpacker_free(bp)


def test_bitpacker_unbound_take(ctx):
    """a packer over "any other ByteWriter" (out = NULL): bytes delivered by x3_bitpacker_take in two flushes, at an odd
    writer position, equal the oracle's"""
    rng = np.random.default_rng(77)
    L, OL = x3hip.lib(), O.lib()
    fields = [(int(rng.integers(0, 1 << 32)), int(rng.integers(1, 33))) for _ in range(500)]
    start = 3
    b = np.zeros(4096, dtype=np.uint8)
    w = O.Writer()
    OL.x3o_writer_init(C.byref(w), b.ctypes.data, b.size)
    OL.x3o_writer_seek_start(C.byref(w), start)
    ob = O.BitPacker()
    OL.x3o_bp_new(C.byref(ob), C.byref(w))
    bp = C.c_void_p()
    assert L.x3_bitpacker_new(ctx._h, None, 0, start, C.byref(bp)) == 0
    got = bytearray()
    n_new, ln, crc = C.c_uint64(0), C.c_uint64(0), C.c_uint16(0)
    tmp = np.zeros(4096, dtype=np.uint8)
    for i, (v, n) in enumerate(fields):
        assert L.x3_bitpacker_write_bits(bp, v, n) == 0
        assert OL.x3o_bp_write_bits(C.byref(ob), v, n) == 0
        if i == 200:
            assert L.x3_bitpacker_take(bp, tmp.ctypes.data, 1, C.byref(n_new), None, None) == 22   # ByteWriterInsufficientMemory
            assert L.x3_bitpacker_take(bp, tmp.ctypes.data, tmp.size, C.byref(n_new), C.byref(ln), C.byref(crc)) == 0
            got += tmp[:n_new.value].tobytes()
            OL.x3o_bp_drop(C.byref(ob))
            assert (ln.value, crc.value) == (ob.byte_len, ob.crc) and len(got) == ln.value
    assert L.x3_bitpacker_word_align(bp) == 0
    assert OL.x3o_bp_word_align(C.byref(ob)) == 0
    assert L.x3_bitpacker_take(bp, tmp.ctypes.data, tmp.size, C.byref(n_new), C.byref(ln), C.byref(crc)) == 0
    got += tmp[:n_new.value].tobytes()
    L.x3_bitpacker_free(bp)
    assert (ln.value, crc.value) == (ob.byte_len, ob.crc)
    assert bytes(got) == b[start:w.p_byte].tobytes() and (start + len(got)) % 2 == 0
