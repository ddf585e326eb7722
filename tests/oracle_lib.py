"""ctypes loader for the CPU oracle (oracle/x3_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Params(C.Structure):
    _fields_ = [("block_len", C.c_uint32), ("blocks_per_frame", C.c_uint32),
                ("codes", C.c_uint32 * 3), ("thresholds", C.c_uint32 * 3)]

    @classmethod
    def default(cls):
        return cls(20, 500, (C.c_uint32 * 3)(0, 1, 3), (C.c_uint32 * 3)(3, 8, 20))

    @classmethod
    def make(cls, block_len=20, blocks_per_frame=500, codes=(0, 1, 3), thresholds=(3, 8, 20)):
        return cls(block_len, blocks_per_frame, (C.c_uint32 * 3)(*codes), (C.c_uint32 * 3)(*thresholds))


class Writer(C.Structure):
    _fields_ = [("slice", C.c_void_p), ("cap", C.c_size_t), ("p_byte", C.c_size_t), ("stream_length", C.c_size_t)]


class BitPacker(C.Structure):
    _fields_ = [("writer", C.POINTER(Writer)), ("scratch_byte", C.c_uint8), ("p_bit", C.c_size_t),
                ("byte_len", C.c_size_t), ("crc", C.c_uint16)]


class BitReader(C.Structure):
    _fields_ = [("array", C.c_void_p), ("len", C.c_size_t), ("idx", C.c_size_t),
                ("leading_word", C.c_uint32), ("rem_bit", C.c_size_t)]


class FrameHeader(C.Structure):
    _fields_ = [("source_id", C.c_uint8), ("samples", C.c_uint16), ("channels", C.c_uint8),
                ("payload_len", C.c_uint32), ("payload_crc", C.c_uint16)]


class RiceCode(C.Structure):
    _fields_ = [("nsubs", C.c_uint32), ("offset", C.c_uint32), ("len", C.c_uint32), ("inv_len", C.c_uint32),
                ("code", C.c_uint32 * 56), ("num_bits", C.c_uint32 * 56)]


_lib = None


def build(native=False):
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", ODIR, target], check=True)
    return os.path.join(ODIR, "libx3oracle_native.so" if native else "libx3oracle.so")


def _stale(so):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    return any(os.path.getmtime(os.path.join(ODIR, f)) > t for f in ("x3_oracle.c", "x3_oracle.h"))


def lib(native=False):
    global _lib
    if _lib is not None and not native:
        return _lib
    so = os.path.join(ODIR, "libx3oracle_native.so" if native else "libx3oracle.so")
    if native or _stale(so):
        so = build(native)
    L = C.CDLL(so)
    L.x3o_crc16.restype = C.c_uint16
    L.x3o_crc16.argtypes = [C.c_void_p, C.c_size_t]
    L.x3o_update_crc16.restype = C.c_uint16
    L.x3o_update_crc16.argtypes = [C.c_uint16, C.c_uint8]
    L.x3o_bp_write_bits.argtypes = [C.POINTER(BitPacker), C.c_uint64, C.c_size_t]
    L.x3o_bp_write_packed_zeros.argtypes = [C.POINTER(BitPacker), C.c_size_t]
    L.x3o_writer_init.argtypes = [C.POINTER(Writer), C.c_void_p, C.c_size_t]
    L.x3o_writer_seek_start.argtypes = [C.POINTER(Writer), C.c_size_t]
    L.x3o_bp_new.argtypes = [C.POINTER(BitPacker), C.POINTER(Writer)]
    L.x3o_bp_new.restype = None
    L.x3o_bp_word_align.argtypes = [C.POINTER(BitPacker)]
    L.x3o_bp_write_bytes.argtypes = [C.POINTER(BitPacker), C.c_void_p, C.c_size_t]
    L.x3o_bp_inc_counter_n_bytes.argtypes = [C.POINTER(BitPacker), C.c_size_t]
    L.x3o_bp_drop.argtypes = [C.POINTER(BitPacker)]
    L.x3o_br_count_zero_bits.argtypes = [C.POINTER(BitReader)]
    L.x3o_br_new.argtypes = [C.POINTER(BitReader), C.c_void_p, C.c_size_t]
    L.x3o_br_read_nbits.restype = C.c_uint32
    L.x3o_br_read_nbits.argtypes = [C.POINTER(BitReader), C.c_size_t]
    L.x3o_br_count_zero_bits.restype = C.c_size_t
    L.x3o_encode_block.argtypes = [C.c_void_p, C.c_size_t, C.c_int16, C.POINTER(BitPacker), C.POINTER(Params),
                                   C.POINTER(C.c_size_t)]
    L.x3o_encode_frame.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Writer), C.POINTER(Params), C.c_void_p]
    L.x3o_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_uint64,
                             C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    L.x3o_write_frame_header.argtypes = [C.c_size_t, C.c_uint8, C.c_size_t, C.c_uint16, C.c_void_p]
    L.x3o_read_frame_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FrameHeader)]
    L.x3o_decode_block.argtypes = [C.POINTER(BitReader), C.c_void_p, C.c_size_t, C.POINTER(C.c_int16),
                                   C.POINTER(Params)]
    L.x3o_decode_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Params), C.c_size_t,
                                   C.POINTER(C.c_size_t)]
    L.x3o_decode_stream.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.x3o_archive_header_write.argtypes = [C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.x3o_archive_header_read.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(Params),
                                          C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.x3o_x3a_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                 C.c_void_p]
    L.x3o_x3a_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.x3o_wav_parse.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16),
                                C.POINTER(C.c_uint16), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.x3o_wav_header_write.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p]
    L.x3o_wav_header_write.restype = None
    L.x3o_wav_to_x3a.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p]
    L.x3o_x3a_to_wav.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.x3o_time_roundtrip.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_int, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.x3o_init()
    if not native:
        _lib = L
    return L


# ------------------------------------------------------------------ numpy-level helpers

def crc16(data):
    b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return lib().x3o_crc16(b.ctypes.data, b.size)


def encode_bound(n, params):
    spf = params.block_len * params.blocks_per_frame
    if spf == 0:
        return 64
    nf = (n + spf - 1) // spf
    # per frame: header, 16 bits per sample, 6 header bits per block (block_len may be 1), alignment
    nblocks = (spf + params.block_len - 1) // params.block_len
    return nf * (20 + 2 * spf + (6 * nblocks + 7) // 8 + 64) + 64


def encode(wav, params=None, start_pos=0, cap=None, n_channels=1):
    """-> (rc, bytes produced as np.uint8 incl. the start_pos prefix, stats[6])"""
    params = params or Params.default()
    wav = np.ascontiguousarray(wav, dtype=np.int16)
    cap = encode_bound(wav.size, params) + start_pos if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    pos = C.c_uint64(0)
    stats = np.zeros(6, dtype=np.uint64)
    rc = lib().x3o_encode(wav.ctypes.data, wav.size, n_channels, C.byref(params), out.ctypes.data, cap, start_pos,
                          C.byref(pos), stats.ctypes.data)
    return rc, out[: pos.value].copy(), stats


def decode_stream(x3, params=None, wav_cap=None):
    """-> (rc, samples np.int16, frames_ok, frame_errors)"""
    params = params or Params.default()
    x3 = np.ascontiguousarray(x3, dtype=np.uint8)
    if wav_cap is None:
        wav_cap = max(1, x3.size * 16)
    wav = np.zeros(wav_cap, dtype=np.int16)
    n = C.c_uint64(0); fok = C.c_uint64(0); ferr = C.c_uint64(0)
    rc = lib().x3o_decode_stream(x3.ctypes.data, x3.size, C.byref(params), wav.ctypes.data, wav_cap, C.byref(n),
                                 C.byref(fok), C.byref(ferr))
    return rc, wav[: n.value].copy(), fok.value, ferr.value


def encode_mc(wavs, params=None, start_pos=0, cap=None):
    """multi-channel extension (x3o_encode_mc): wavs = list of equally long int16 arrays -> (rc, bytes, stats[6])"""
    params = params or Params.default()
    wavs = [np.ascontiguousarray(w, dtype=np.int16) for w in wavs]
    n = wavs[0].size
    assert all(w.size == n for w in wavs)
    cap = len(wavs) * encode_bound(n, params) + start_pos + 64 if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    pos = C.c_uint64(0)
    stats = np.zeros(6, dtype=np.uint64)
    ptrs = (C.c_void_p * len(wavs))(*[w.ctypes.data for w in wavs])
    L = lib()
    L.x3o_encode_mc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                C.c_void_p, C.c_void_p]
    rc = L.x3o_encode_mc(ptrs, len(wavs), n, C.byref(params), out.ctypes.data, cap, start_pos, C.byref(pos),
                         stats.ctypes.data)
    return rc, out[: pos.value].copy(), stats


def decode_stream_mc(x3, n_ch, params=None, wav_cap=None):
    """-> (rc, [samples of channel c], frames_ok, frame_errors)"""
    params = params or Params.default()
    x3 = np.ascontiguousarray(x3, dtype=np.uint8)
    if wav_cap is None:
        wav_cap = max(1, x3.size * 16)
    wavs = [np.zeros(wav_cap, dtype=np.int16) for _ in range(n_ch)]
    ptrs = (C.c_void_p * n_ch)(*[w.ctypes.data for w in wavs])
    n = C.c_uint64(0); fok = C.c_uint64(0); ferr = C.c_uint64(0)
    L = lib()
    L.x3o_decode_stream_mc.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.x3o_decode_stream_mc(x3.ctypes.data, x3.size, n_ch, C.byref(params), ptrs, wav_cap, C.byref(n), C.byref(fok),
                                C.byref(ferr))
    return rc, [w[: n.value].copy() for w in wavs], fok.value, ferr.value


def decode_frame(payload, samples, params=None, wav_cap=None):
    params = params or Params.default()
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    wav_cap = samples if wav_cap is None else wav_cap
    wav = np.zeros(max(wav_cap, 1), dtype=np.int16)
    n = C.c_size_t(0)
    rc = lib().x3o_decode_frame(payload.ctypes.data, payload.size, wav.ctypes.data, wav_cap, C.byref(params), samples,
                                C.byref(n))
    return rc, wav[: n.value].copy()


def archive_header_write(sample_rate, params=None, cap=1024):
    params = params or Params.default()
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    rc = lib().x3o_archive_header_write(sample_rate, C.byref(params), out.ctypes.data, cap, C.byref(n))
    return rc, out[: min(n.value, cap)].copy()


def archive_header_read(data):
    b = np.ascontiguousarray(data, dtype=np.uint8)
    rate, p, ch, hs = C.c_uint32(0), Params(), C.c_uint8(0), C.c_uint64(0)
    rc = lib().x3o_archive_header_read(b.ctypes.data, b.size, C.byref(rate), C.byref(p), C.byref(ch), C.byref(hs))
    return rc, rate.value, p, ch.value, hs.value


def x3a_encode(wav, sample_rate, cap=None):
    wav = np.ascontiguousarray(wav, dtype=np.int16)
    cap = 1024 + encode_bound(wav.size, Params.default()) if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64(0)
    stats = np.zeros(6, dtype=np.uint64)
    rc = lib().x3o_x3a_encode(wav.ctypes.data, wav.size, sample_rate, out.ctypes.data, cap, C.byref(n), stats.ctypes.data)
    return rc, out[: min(n.value, cap)].copy(), stats


def x3a_decode(x3a, wav_cap=None):
    x3a = np.ascontiguousarray(x3a, dtype=np.uint8)
    wav_cap = max(1, x3a.size * 16) if wav_cap is None else wav_cap
    wav = np.zeros(wav_cap, dtype=np.int16)
    n, rate, fok, ferr = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib().x3o_x3a_decode(x3a.ctypes.data, x3a.size, wav.ctypes.data, wav_cap, C.byref(n), C.byref(rate),
                              C.byref(fok), C.byref(ferr))
    return rc, wav[: n.value].copy(), rate.value, fok.value, ferr.value


def wav_header(sample_rate, n_samples):
    out = np.zeros(44, dtype=np.uint8)
    lib().x3o_wav_header_write(sample_rate, n_samples, out.ctypes.data)
    return out


def wav_parse(data):
    """-> (rc, sample_rate, channels, bits, data_off, data_len)"""
    b = np.ascontiguousarray(data, dtype=np.uint8)
    rate, ch, bits, off, dlen = C.c_uint32(0), C.c_uint16(0), C.c_uint16(0), C.c_uint64(0), C.c_uint64(0)
    rc = lib().x3o_wav_parse(b.ctypes.data, b.size, C.byref(rate), C.byref(ch), C.byref(bits), C.byref(off), C.byref(dlen))
    return rc, rate.value, ch.value, bits.value, off.value, dlen.value


def wav_to_x3a(wav_path, x3a_path):
    stats = np.zeros(6, dtype=np.uint64)
    rc = lib().x3o_wav_to_x3a(os.fsencode(wav_path), os.fsencode(x3a_path), stats.ctypes.data)
    return rc, stats


def x3a_to_wav(x3a_path, wav_path):
    n, ferr = C.c_uint64(0), C.c_uint64(0)
    rc = lib().x3o_x3a_to_wav(os.fsencode(x3a_path), os.fsencode(wav_path), C.byref(n), C.byref(ferr))
    return rc, n.value, ferr.value
