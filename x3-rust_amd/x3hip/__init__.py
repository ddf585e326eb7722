"""ctypes binding of libx3hip.so (include/x3hip.h): the MI355X-native X3 encoder/decoder.

Thin by design: every call goes straight to the C ABI, which runs the HIP kernels.  There is
no Python or CPU implementation behind it -- if the library or a GPU is missing, loading or
context creation raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libx3hip.so")
if os.environ.get("X3HIP_LIB"):   # (development: an experiment build of the same library, tools/variants.py)
    LIB_PATH = os.environ["X3HIP_LIB"]

OK = 0
ERR_INVALID_ENCODING_THRESH = 4
ERR_OUT_OF_BOUNDS_INVERSE = 5
ERR_MORE_THAN_ONE_CHANNEL = 6
ERR_FRAME_LENGTH = 10
ERR_FRAME_HEADER_INVALID_KEY = 11
ERR_FRAME_HEADER_INVALID_PAYLOAD_LEN = 12
ERR_FRAME_HEADER_INVALID_HEADER_CRC = 13
ERR_FRAME_HEADER_INVALID_PAYLOAD_CRC = 14
ERR_FRAME_DECODE_INVALID_BPF = 20
ERR_FRAME_DECODE_UNEXPECTED_END = 21
ERR_BYTE_WRITER_INSUFFICIENT_MEMORY = 22
ERR_HIP = 23
ERR_BAD_ARG = 24

# every symbol include/x3hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "x3_strerror", "x3_ctx_create", "x3_ctx_create_on_stream", "x3_ctx_destroy", "x3_ctx_sync", "x3_last_error",
    "x3_ctx_set_option", "x3_ctx_get_option",
    "x3_ctx_enable_kernel_timing", "x3_ctx_kernel_time", "x3_ctx_kernel_times", "x3_ctx_launch_log", "x3_ctx_reset_kernel_time",
    "x3_params_default", "x3_params_validate", "x3_rice_code_get", "x3_num_frames", "x3_encode_bound",
    "x3_crc16", "x3_crc16_dev", "x3_crc16_update",
    "x3_encode", "x3_encode_frame", "x3_write_frame_header", "x3_encode_batch",
    "x3_read_frame_header", "x3_decode_frame", "x3_decode_prefetch", "x3_decode_stream",
    "x3_archive_header_write", "x3_archive_header_read", "x3_x3a_encode", "x3_x3a_decode",
    "x3_wav_to_x3a", "x3_x3a_to_wav",
    "x3_bitreader_new", "x3_bitreader_read_nbits", "x3_bitreader_count_zero_bits", "x3_bitreader_inc_bits",
    "x3_bitreader_state", "x3_bitreader_free", "x3_decode_block",
    "x3_bitpacker_new", "x3_bitpacker_write_bits", "x3_bitpacker_write_packed_zeros", "x3_bitpacker_write_bytes", "x3_bitpacker_inc_counter_n_bytes", "x3_bitpacker_word_align",
    "x3_bitpacker_finish", "x3_bitpacker_peek", "x3_bitpacker_take", "x3_bitpacker_free",
    "x3_reader_open", "x3_reader_open_mem", "x3_reader_spec", "x3_reader_next_frame", "x3_reader_frame_errors",
    "x3_reader_position", "x3_reader_close",
    "x3_encode_dev", "x3_encode_frames_dev", "x3_encode_result", "x3_decode_dev", "x3_decode_result", "x3_index_dev", "x3_decode_stream_dev",
    "x3_seg_index_entries", "x3_decode_dev_seg", "x3_encode_dev_seg", "x3_place_buffers",
    "x3_graph_begin", "x3_graph_end", "x3_graph_launch", "x3_graph_destroy",
    "x3_synth", "x3_synth_dev", "x3_dev_alloc", "x3_dev_free", "x3_dev_upload", "x3_dev_download",
    "x3_shard_unique_id", "x3_shard_create", "x3_shard_destroy", "x3_shard_rank", "x3_shard_world",
    "x3_shard_frame_range", "x3_shard_sample_range", "x3_shard_offsets", "x3_shard_exchange_lengths",
    "x3_shard_exchange_length_value", "x3_shard_lengths", "x3_shard_gather", "x3_shard_gather_async", "x3_shard_gather_wait", "x3_shard_write_at",
    "x3_mgpu_create", "x3_mgpu_destroy", "x3_mgpu_devices", "x3_mgpu_ctx", "x3_mgpu_shard", "x3_mgpu_last_error",
    "x3_mgpu_encode", "x3_mgpu_decode_stream",
    "x3_encode_mc", "x3_decode_stream_mc",
]


class RiceCode(C.Structure):
    """x3_rice_code (RiceCode, src/x3.rs:187-194)"""
    _fields_ = [("nsubs", C.c_uint32), ("offset", C.c_uint32), ("len", C.c_uint32), ("inv_len", C.c_uint32),
                ("code", C.POINTER(C.c_uint32)), ("num_bits", C.POINTER(C.c_uint32)), ("inv", C.POINTER(C.c_int16))]


class Params(C.Structure):
    """x3::Parameters (src/x3.rs:81-134)"""
    _fields_ = [("block_len", C.c_uint32), ("blocks_per_frame", C.c_uint32),
                ("codes", C.c_uint32 * 3), ("thresholds", C.c_uint32 * 3)]

    @classmethod
    def default(cls):
        p = cls()
        lib().x3_params_default(C.byref(p))
        return p

    @classmethod
    def make(cls, block_len=20, blocks_per_frame=500, codes=(0, 1, 3), thresholds=(3, 8, 20)):
        return cls(block_len, blocks_per_frame, (C.c_uint32 * 3)(*codes), (C.c_uint32 * 3)(*thresholds))

    @property
    def spf(self):
        return self.block_len * self.blocks_per_frame


class FrameHeader(C.Structure):
    """x3::FrameHeader (src/x3.rs:148-184)"""
    _fields_ = [("source_id", C.c_uint8), ("channels", C.c_uint8), ("samples", C.c_uint16),
                ("payload_len", C.c_uint32), ("payload_crc", C.c_uint16)]


class Batch(C.Structure):
    _fields_ = [("n_per_clip", C.c_uint64), ("clip_stride", C.c_uint64), ("n_clips", C.c_uint64)]


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed its copy is loaded
    first and libx3hip.so's DT_NEEDED libamdhip64.so.7 then binds to it -- in either import order
    bench.py (torch for HBM tensors / streams / RCCL + this library for the kernels) sees one runtime.
    Without torch the system runtime from /opt/rocm is used."""
    if os.environ.get("X3HIP_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libx3hip.so; raises if it has not been built (python x3-rust_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libx3hip.so is missing (%s): run `python x3-rust_amd/build.py`; "
                           "there is no CPU fallback" % LIB_PATH)
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    PP = C.POINTER(Params)
    L.x3_strerror.restype = C.c_char_p
    L.x3_strerror.argtypes = [i32]
    L.x3_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.x3_ctx_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.x3_ctx_destroy.restype = None
    L.x3_ctx_destroy.argtypes = [vp]
    L.x3_ctx_sync.argtypes = [vp]
    L.x3_last_error.restype = C.c_char_p
    L.x3_last_error.argtypes = [vp]
    L.x3_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.x3_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.x3_ctx_enable_kernel_timing.argtypes = [vp, i32]
    L.x3_ctx_kernel_time.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)]
    L.x3_ctx_kernel_times.argtypes = [vp, i32, C.POINTER(C.c_double), u64, C.POINTER(u64)]
    L.x3_ctx_launch_log.argtypes = [vp, i32, C.POINTER(C.c_uint32), u64, C.POINTER(u64)]
    L.x3_ctx_reset_kernel_time.argtypes = [vp]
    L.x3_params_default.restype = None
    L.x3_params_default.argtypes = [PP]
    L.x3_params_validate.argtypes = [PP]
    L.x3_rice_code_get.argtypes = [C.c_uint32, C.POINTER(RiceCode)]
    L.x3_num_frames.restype = u64
    L.x3_num_frames.argtypes = [u64, PP]
    L.x3_encode_bound.restype = u64
    L.x3_encode_bound.argtypes = [u64, PP]
    L.x3_crc16.argtypes = [vp, vp, u64, C.POINTER(C.c_uint16)]
    L.x3_crc16_dev.argtypes = [vp, vp, u64, C.POINTER(C.c_uint16)]
    L.x3_crc16_update.restype = C.c_uint16
    L.x3_crc16_update.argtypes = [C.c_uint16, C.c_uint8]
    L.x3_encode.argtypes = [vp, vp, u64, u32, PP, vp, u64, u64, C.POINTER(u64), vp]
    L.x3_encode_frame.argtypes = [vp, vp, u64, PP, vp, u64, u64, C.POINTER(u64), vp]
    L.x3_write_frame_header.restype = None
    L.x3_write_frame_header.argtypes = [u64, C.c_uint8, u64, C.c_uint16, vp]
    L.x3_encode_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), u64, PP, vp, u64, C.POINTER(u64), vp]
    L.x3_read_frame_header.argtypes = [vp, u64, C.POINTER(FrameHeader)]
    L.x3_decode_frame.argtypes = [vp, vp, u64, vp, u64, PP, u64, C.POINTER(u64)]
    L.x3_decode_prefetch.argtypes = [vp, vp, u64, PP]
    L.x3_decode_stream.argtypes = [vp, vp, u64, PP, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.x3_archive_header_write.argtypes = [u32, PP, vp, u64, C.POINTER(u64)]
    L.x3_archive_header_read.argtypes = [vp, u64, C.POINTER(u32), PP, C.POINTER(C.c_uint8), C.POINTER(u64)]
    L.x3_x3a_encode.argtypes = [vp, vp, u64, u32, vp, u64, C.POINTER(u64), vp]
    L.x3_x3a_decode.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
    L.x3_wav_to_x3a.argtypes = [vp, C.c_char_p, C.c_char_p, vp]
    L.x3_x3a_to_wav.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(u64), C.POINTER(u64)]
    L.x3_encode_dev.argtypes = [vp, vp, C.POINTER(Batch), PP, vp, u64, u64, vp]
    L.x3_encode_frames_dev.argtypes = [vp, vp, vp, vp, u64, PP, vp, u64, u64, vp]
    L.x3_encode_result.argtypes = [vp, C.POINTER(u64), vp]
    L.x3_decode_dev.argtypes = [vp, vp, u64, vp, u64, C.POINTER(Batch), vp, PP, vp, u64, vp]
    L.x3_decode_result.argtypes = [vp, C.POINTER(u64), C.POINTER(i32), C.POINTER(u64)]
    L.x3_encode_dev_seg.argtypes = [vp, vp, C.POINTER(Batch), PP, vp, u64, u64, vp, vp, u32]
    L.x3_graph_begin.argtypes = [vp]
    L.x3_graph_end.argtypes = [vp, C.POINTER(vp)]
    L.x3_graph_launch.argtypes = [vp, vp]
    L.x3_graph_destroy.argtypes = [vp]
    L.x3_graph_destroy.restype = None
    L.x3_seg_index_entries.argtypes = [u64, PP, u32]
    L.x3_seg_index_entries.restype = u64
    L.x3_decode_dev_seg.argtypes = [vp, vp, u64, vp, u64, C.POINTER(Batch), vp, PP, vp, u64, vp, vp, u32, i32]
    L.x3_index_dev.argtypes = [vp, vp, u64, u64, vp, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.x3_decode_stream_dev.argtypes = [vp, vp, u64, PP, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.x3_synth.argtypes = [i32, u64, u64, u64, vp]
    L.x3_synth_dev.argtypes = [vp, i32, u64, u64, u64, vp]
    L.x3_dev_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.x3_dev_free.argtypes = [vp, vp]
    L.x3_dev_upload.argtypes = [vp, vp, vp, u64]
    L.x3_dev_download.argtypes = [vp, vp, vp, u64]
    L.x3_bitreader_new.argtypes = [vp, vp, u64, C.POINTER(vp)]
    L.x3_bitreader_read_nbits.argtypes = [vp, u32, C.POINTER(u32)]
    L.x3_bitreader_count_zero_bits.argtypes = [vp, C.POINTER(u32)]
    L.x3_bitreader_inc_bits.argtypes = [vp, u32]
    L.x3_bitreader_state.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]
    L.x3_bitreader_free.restype = None
    L.x3_bitreader_free.argtypes = [vp]
    L.x3_decode_block.argtypes = [vp, vp, u32, C.POINTER(C.c_int16), PP]
    L.x3_bitpacker_new.argtypes = [vp, vp, u64, u64, C.POINTER(vp)]
    L.x3_bitpacker_write_bits.argtypes = [vp, u64, u32]
    L.x3_bitpacker_write_packed_zeros.argtypes = [vp, u32]
    L.x3_bitpacker_write_bytes.argtypes = [vp, vp, u64]
    L.x3_bitpacker_inc_counter_n_bytes.argtypes = [vp, u64]
    L.x3_bitpacker_word_align.argtypes = [vp]
    L.x3_bitpacker_finish.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_uint16), C.POINTER(u64)]
    L.x3_bitpacker_peek.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_uint16)]
    L.x3_bitpacker_take.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_uint16)]
    L.x3_bitpacker_free.restype = None
    L.x3_bitpacker_free.argtypes = [vp]
    L.x3_reader_open.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.x3_reader_open_mem.argtypes = [vp, vp, u64, C.POINTER(vp)]
    L.x3_reader_spec.argtypes = [vp, C.POINTER(u32), PP, C.POINTER(C.c_uint8)]
    L.x3_reader_next_frame.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.x3_reader_frame_errors.restype = u64
    L.x3_reader_frame_errors.argtypes = [vp]
    L.x3_reader_position.restype = u64
    L.x3_reader_position.argtypes = [vp]
    L.x3_reader_close.restype = None
    L.x3_reader_close.argtypes = [vp]
    L.x3_shard_unique_id.argtypes = [vp]
    L.x3_shard_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.x3_shard_destroy.restype = None
    L.x3_shard_destroy.argtypes = [vp]
    L.x3_shard_rank.argtypes = [vp]
    L.x3_shard_world.argtypes = [vp]
    L.x3_shard_frame_range.restype = None
    L.x3_shard_frame_range.argtypes = [u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.x3_shard_sample_range.restype = None
    L.x3_shard_sample_range.argtypes = [u64, PP, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.x3_shard_offsets.restype = None
    L.x3_shard_offsets.argtypes = [C.POINTER(u64), i32, C.POINTER(u64)]
    L.x3_shard_exchange_lengths.argtypes = [vp, vp, vp]
    L.x3_shard_exchange_length_value.argtypes = [vp, u64, vp]
    L.x3_shard_lengths.argtypes = [vp, C.POINTER(u64)]
    L.x3_shard_gather.argtypes = [vp, vp, C.POINTER(u64), i32, vp, u64, C.POINTER(u64)]
    L.x3_shard_gather_async.argtypes = [vp, vp, C.POINTER(u64), i32, vp, u64, C.POINTER(u64)]
    L.x3_shard_gather_wait.argtypes = [vp, i32]
    L.x3_shard_write_at.argtypes = [vp, vp, C.POINTER(u64), i32, u64, C.POINTER(u64)]
    L.x3_mgpu_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    L.x3_mgpu_destroy.restype = None
    L.x3_mgpu_destroy.argtypes = [vp]
    L.x3_mgpu_devices.argtypes = [vp]
    L.x3_mgpu_ctx.restype = vp
    L.x3_mgpu_ctx.argtypes = [vp, i32]
    L.x3_mgpu_shard.restype = vp
    L.x3_mgpu_shard.argtypes = [vp, i32]
    L.x3_mgpu_last_error.restype = C.c_char_p
    L.x3_mgpu_last_error.argtypes = [vp]
    L.x3_mgpu_encode.argtypes = [vp, vp, u64, u32, PP, vp, u64, u64, C.POINTER(u64), vp]
    L.x3_mgpu_decode_stream.argtypes = [vp, vp, u64, PP, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    _lib = L
    return L


# ---- sharding arithmetic (host only: works without a GPU)

def shard_frame_range(n_frames, rank, world):
    a, c = C.c_uint64(0), C.c_uint64(0)
    lib().x3_shard_frame_range(n_frames, rank, world, C.byref(a), C.byref(c))
    return a.value, c.value


def shard_sample_range(n_samples, params, rank, world):
    a, c = C.c_uint64(0), C.c_uint64(0)
    lib().x3_shard_sample_range(n_samples, C.byref(params), rank, world, C.byref(a), C.byref(c))
    return a.value, c.value


def shard_offsets(lengths):
    n = len(lengths)
    src = (C.c_uint64 * n)(*[int(v) for v in lengths])
    dst = (C.c_uint64 * (n + 1))()
    lib().x3_shard_offsets(src, n, dst)
    return list(dst)


def shard_unique_id():
    """ncclGetUniqueId through the library: 128 bytes, made by rank 0 and handed to the other ranks out of band"""
    buf = (C.c_uint8 * 128)()
    rc = lib().x3_shard_unique_id(buf)
    if rc:
        raise X3Error(rc, "x3_shard_unique_id (librccl not available?)")
    return bytes(buf)


class Reader:
    """X3aReader (decodefile.rs:47-137) over a file path or over archive bytes: spec() and next_frame()"""

    def __init__(self, ctx, source):
        self._h = C.c_void_p()
        self.ctx = ctx
        if isinstance(source, (str, bytes, os.PathLike)) and not isinstance(source, bytes):
            rc = lib().x3_reader_open(ctx._h, os.fsencode(source), C.byref(self._h))
        else:
            self._keep = np.ascontiguousarray(source, dtype=np.uint8)   # borrowed by the reader
            rc = lib().x3_reader_open_mem(ctx._h, self._keep.ctypes.data, self._keep.size, C.byref(self._h))
        self.rc = rc
        self._buf = np.zeros(65536, dtype=np.int16)

    def spec(self):
        rate, p, ch = C.c_uint32(0), Params(), C.c_uint8(0)
        lib().x3_reader_spec(self._h, C.byref(rate), C.byref(p), C.byref(ch))
        return rate.value, p, ch.value

    def next_frame(self):
        """-> (rc, samples or None): None = Ok(None) of the reference"""
        n = C.c_uint64(0)
        rc = lib().x3_reader_next_frame(self._h, self._buf.ctypes.data, self._buf.size, C.byref(n))
        return rc, (self._buf[: n.value].copy() if (rc == 0 and n.value) else None)

    def frame_errors(self):
        return lib().x3_reader_frame_errors(self._h)

    def position(self):
        return lib().x3_reader_position(self._h)

    def close(self):
        if self._h:
            lib().x3_reader_close(self._h)
            self._h = C.c_void_p()


class Shard:
    """one rank of a group of GPUs (x3_shard): RCCL communicator on the context's device and stream"""

    def __init__(self, ctx, unique_id, rank, world):
        self._h = C.c_void_p()
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = lib().x3_shard_create(ctx._h, buf, rank, world, C.byref(self._h))
        if rc:
            raise X3Error(rc, "x3_shard_create: " + ctx.last_error())

    def close(self):
        if self._h:
            lib().x3_shard_destroy(self._h)
            self._h = C.c_void_p()

    def exchange_lengths(self, d_len, d_lengths=None):
        rc = lib().x3_shard_exchange_lengths(self._h, d_len, d_lengths)
        if rc:
            raise X3Error(rc, "x3_shard_exchange_lengths: " + self.ctx.last_error())

    def lengths(self):
        out = (C.c_uint64 * self.world)()
        rc = lib().x3_shard_lengths(self._h, out)
        if rc:
            raise X3Error(rc, "x3_shard_lengths: " + self.ctx.last_error())
        return list(out)

    def gather(self, d_sub, lengths, root, d_dst, dst_cap, overlapped=False):
        """reassembly on `root`; overlapped=True: on the shard's own stream and communicator, beside whatever the context
        does next (x3_shard_gather_async) -- d_sub / d_dst stay untouched until gather_wait()"""
        src = (C.c_uint64 * self.world)(*[int(v) for v in lengths])
        tot = C.c_uint64(0)
        fn = lib().x3_shard_gather_async if overlapped else lib().x3_shard_gather
        rc = fn(self._h, d_sub, src, root, d_dst, dst_cap, C.byref(tot))
        if rc:
            raise X3Error(rc, "x3_shard_gather: " + self.ctx.last_error())
        return tot.value

    def write_at(self, d_sub, lengths, fd, base=0):
        """sharded reassembly (x3_shard_write_at): this rank's sub-stream to byte base + starts[rank] of the file `fd`"""
        src = (C.c_uint64 * self.world)(*[int(v) for v in lengths])
        tot = C.c_uint64(0)
        rc = lib().x3_shard_write_at(self._h, d_sub, src, fd, base, C.byref(tot))
        if rc:
            raise X3Error(rc, "x3_shard_write_at: " + self.ctx.last_error())
        return tot.value

    def gather_wait(self, on_stream=True):
        rc = lib().x3_shard_gather_wait(self._h, 1 if on_stream else 0)
        if rc:
            raise X3Error(rc, "x3_shard_gather_wait: " + self.ctx.last_error())


class MultiGpu:
    """all GPUs from one process (x3_mgpu): host buffers in and out, same bytes as Context.encode / decode_stream"""

    def __init__(self, devices):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = lib().x3_mgpu_create(arr, len(devices), C.byref(self._h))
        if rc:
            raise X3Error(rc, "x3_mgpu_create")

    def close(self):
        if self._h:
            lib().x3_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def last_error(self):
        return lib().x3_mgpu_last_error(self._h).decode()

    def encode(self, wav, params=None, start_pos=0, cap=None, n_channels=1):
        params = params or Params.default()
        wav = np.ascontiguousarray(wav, dtype=np.int16)
        if cap is None:
            cap = start_pos + lib().x3_encode_bound(wav.size, C.byref(params)) + 1
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        pos = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_mgpu_encode(self._h, wav.ctypes.data, wav.size, n_channels, C.byref(params), out.ctypes.data,
                                  cap, start_pos, C.byref(pos), stats.ctypes.data)
        return rc, out[: min(pos.value, cap)].copy(), stats

    def decode_stream(self, x3, params=None, wav_cap=None):
        params = params or Params.default()
        x3 = np.ascontiguousarray(x3, dtype=np.uint8)
        if wav_cap is None:
            wav_cap = max(1, x3.size * 16)
        wav = np.zeros(wav_cap, dtype=np.int16)
        n, fok, ferr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib().x3_mgpu_decode_stream(self._h, x3.ctypes.data, x3.size, C.byref(params), wav.ctypes.data, wav_cap,
                                         C.byref(n), C.byref(fok), C.byref(ferr))
        return rc, wav[: n.value].copy(), fok.value, ferr.value


def place_buffers(ctx, params, d_wav, n, d_streams, cap, d_frame_offsets, d_backs, warm=4, steps=8):
    """x3_place_buffers (include/x3hip.h, "Placement"): where in HBM the stream and the decoded samples lie decides the decode
    phase's pace by up to 10 % -- per PAIR of buffers, reproducibly within a process, and not by anything an address shows
    (profiles/r6/decoder_modes.txt).  Times `steps` round trips (after `warm` untimed ones) on every (stream buffer, sample
    buffer) pair and returns ms_per_step[i][j]; the caller keeps the best pair and frees the rest."""
    ns, nb = len(d_streams), len(d_backs)
    streams = (C.c_void_p * ns)(*d_streams)
    backs = (C.c_void_p * nb)(*d_backs)
    ms = (C.c_double * (ns * nb))()
    L = lib()
    L.x3_place_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.x3_place_buffers.restype = C.c_int
    rc = L.x3_place_buffers(ctx._h, d_wav, n, C.byref(params), streams, ns, cap, d_frame_offsets, backs, nb, warm, steps, ms)
    if rc:
        raise X3Error(rc, "x3_place_buffers: " + ctx.last_error())
    return [[ms[i * nb + j] for j in range(nb)] for i in range(ns)]


def strerror(rc):
    return lib().x3_strerror(rc).decode()


class X3Error(RuntimeError):
    def __init__(self, rc, what=""):
        self.rc = rc
        super().__init__("%s: %s (%d)" % (what, strerror(rc), rc))


SYNTH_ZEROS, SYNTH_WHITE, SYNTH_HYDROPHONE, SYNTH_SINE, SYNTH_WALK = range(5)


def synth(kind, seed, start, n):
    """host-side synthetic samples (bit-identical to Context.synth_dev)"""
    out = np.empty(n, dtype=np.int16)
    rc = lib().x3_synth(kind, seed, start, n, out.ctypes.data)
    if rc:
        raise X3Error(rc, "x3_synth")
    return out


def write_frame_header(num_samples, ident, payload_len, payload_crc):
    out = np.zeros(20, dtype=np.uint8)
    lib().x3_write_frame_header(num_samples, ident, payload_len, payload_crc, out.ctypes.data)
    return out


def archive_header_write(sample_rate, params=None, cap=1024):
    """-> (rc, header bytes)"""
    params = params or Params.default()
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    rc = lib().x3_archive_header_write(sample_rate, C.byref(params), out.ctypes.data, cap, C.byref(n))
    return rc, out[: min(n.value, cap)].copy()


def archive_header_read(data):
    """-> (rc, sample_rate, Params, channels, header_size)"""
    b = np.ascontiguousarray(data, dtype=np.uint8)
    rate, p, ch, hs = C.c_uint32(0), Params(), C.c_uint8(0), C.c_uint64(0)
    rc = lib().x3_archive_header_read(b.ctypes.data, b.size, C.byref(rate), C.byref(p), C.byref(ch), C.byref(hs))
    return rc, rate.value, p, ch.value, hs.value


def read_frame_header(data):
    b = np.ascontiguousarray(data, dtype=np.uint8)
    h = FrameHeader()
    rc = lib().x3_read_frame_header(b.ctypes.data, b.size, C.byref(h))
    return rc, h


class Context:
    """One x3_ctx: a GPU + stream + scratch.  Not thread-safe."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        L = lib()
        if stream is None:
            rc = L.x3_ctx_create(device, C.byref(self._h))
        else:
            rc = L.x3_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(self._h))
        if rc:
            raise X3Error(rc, "x3_ctx_create (no usable HIP device? there is no CPU fallback)")

    def close(self):
        if self._h:
            lib().x3_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return lib().x3_last_error(self._h).decode()

    def set_option(self, name, value):
        """tuning / testing knobs (include/x3hip.h: x3_ctx_set_option)"""
        rc = lib().x3_ctx_set_option(self._h, name.encode(), int(value))
        if rc:
            raise X3Error(rc, "x3_ctx_set_option(%s)" % name)

    def get_option(self, name):
        v = C.c_longlong(0)
        rc = lib().x3_ctx_get_option(self._h, name.encode(), C.byref(v))
        if rc:
            raise X3Error(rc, "x3_ctx_get_option(%s)" % name)
        return v.value

    def sync(self):
        rc = lib().x3_ctx_sync(self._h)
        if rc:
            raise X3Error(rc, "x3_ctx_sync: " + self.last_error())

    # ---- host-buffer API (returns status codes, like the C ABI) ---------------------------
    def encode(self, wav, params=None, start_pos=0, cap=None, n_channels=1):
        """encoder::encode into a slice writer -> (rc, np.uint8 bytes[0:out_pos], stats[6])"""
        params = params or Params.default()
        wav = np.ascontiguousarray(wav, dtype=np.int16)
        if cap is None:
            cap = start_pos + lib().x3_encode_bound(wav.size, C.byref(params)) + 1
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        pos = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_encode(self._h, wav.ctypes.data, wav.size, n_channels, C.byref(params), out.ctypes.data, cap,
                             start_pos, C.byref(pos), stats.ctypes.data)
        self.out_pos = pos.value
        return rc, out[: min(pos.value, cap)].copy(), stats

    def encode_frame(self, wav, params=None, start_pos=0, cap=None):
        params = params or Params.default()
        wav = np.ascontiguousarray(wav, dtype=np.int16)
        if cap is None:
            cap = start_pos + 64 + 3 * wav.size
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        pos = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_encode_frame(self._h, wav.ctypes.data, wav.size, C.byref(params), out.ctypes.data, cap,
                                   start_pos, C.byref(pos), stats.ctypes.data)
        return rc, out[: min(pos.value, cap)].copy(), stats

    def encode_batch(self, clips, params=None, cap=None):
        params = params or Params.default()
        clips = [np.ascontiguousarray(c, dtype=np.int16) for c in clips]
        ptrs = (C.c_void_p * len(clips))(*[c.ctypes.data for c in clips])
        ns = (C.c_uint64 * len(clips))(*[c.size for c in clips])
        if cap is None:
            cap = sum(lib().x3_encode_bound(c.size, C.byref(params)) + 2 for c in clips)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        offs = (C.c_uint64 * (len(clips) + 1))()
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_encode_batch(self._h, ptrs, ns, len(clips), C.byref(params), out.ctypes.data, cap, offs,
                                   stats.ctypes.data)
        return rc, out, list(offs), stats

    def decode_stream(self, x3, params=None, wav_cap=None):
        """-> (rc, samples, frames_ok, frame_errors)"""
        params = params or Params.default()
        x3 = np.ascontiguousarray(x3, dtype=np.uint8)
        if wav_cap is None:
            wav_cap = max(1, x3.size * 16)
        wav = np.zeros(wav_cap, dtype=np.int16)
        n, fok, ferr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib().x3_decode_stream(self._h, x3.ctypes.data, x3.size, C.byref(params), wav.ctypes.data, wav_cap,
                                    C.byref(n), C.byref(fok), C.byref(ferr))
        return rc, wav[: n.value].copy(), fok.value, ferr.value

    def encode_mc(self, wavs, params=None, start_pos=0, cap=None):
        """multi-channel extension (x3_encode_mc): wavs = equally long int16 arrays -> (rc, bytes, stats[6])"""
        params = params or Params.default()
        wavs = [np.ascontiguousarray(w, dtype=np.int16) for w in wavs]
        n = wavs[0].size
        assert all(w.size == n for w in wavs)
        L = lib()
        cap = len(wavs) * L.x3_encode_bound(n, C.byref(params)) + start_pos + 64 if cap is None else cap
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        pos = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        ptrs = (C.c_void_p * len(wavs))(*[w.ctypes.data for w in wavs])
        L.x3_encode_mc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                   C.c_uint64, C.c_void_p, C.c_void_p]
        rc = L.x3_encode_mc(self._h, ptrs, len(wavs), n, C.byref(params), out.ctypes.data, cap, start_pos, C.byref(pos),
                            stats.ctypes.data)
        return rc, out[: pos.value].copy(), stats

    def decode_stream_mc(self, x3, n_ch, params=None, wav_cap=None):
        """-> (rc, [samples of channel k], frames_ok, frame_errors)"""
        params = params or Params.default()
        x3 = np.ascontiguousarray(x3, dtype=np.uint8)
        if wav_cap is None:
            wav_cap = max(1, x3.size * 16)
        wavs = [np.zeros(wav_cap, dtype=np.int16) for _ in range(n_ch)]
        ptrs = (C.c_void_p * n_ch)(*[w.ctypes.data for w in wavs])
        n, fok, ferr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L = lib()
        L.x3_decode_stream_mc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = L.x3_decode_stream_mc(self._h, x3.ctypes.data, x3.size, n_ch, C.byref(params), ptrs, wav_cap, C.byref(n),
                                   C.byref(fok), C.byref(ferr))
        return rc, [w[: n.value].copy() for w in wavs], fok.value, ferr.value

    def decode_prefetch(self, x3=None, params=None):
        """announce a frame stream (a contiguous uint8 array, kept alive here) for decode_frame loops; None drops it"""
        if x3 is None:
            self._prefetched = None
            return lib().x3_decode_prefetch(self._h, None, 0, None)
        params = params or Params.default()
        assert x3.dtype == np.uint8 and x3.flags.c_contiguous
        self._prefetched = x3
        return lib().x3_decode_prefetch(self._h, x3.ctypes.data, x3.size, C.byref(params))

    def decode_frame(self, payload, samples, params=None, wav_cap=None):
        params = params or Params.default()
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        wav_cap = samples if wav_cap is None else wav_cap
        wav = np.zeros(max(wav_cap, 1), dtype=np.int16)
        n = C.c_uint64(0)
        rc = lib().x3_decode_frame(self._h, payload.ctypes.data, payload.size, wav.ctypes.data, wav_cap,
                                   C.byref(params), samples, C.byref(n))
        return rc, wav[: n.value].copy()

    def x3a_encode(self, wav, sample_rate, cap=None):
        """wav_to_x3a in memory -> (rc, .x3a bytes, stats)"""
        wav = np.ascontiguousarray(wav, dtype=np.int16)
        p = Params.default()
        if cap is None:
            cap = 1024 + lib().x3_encode_bound(wav.size, C.byref(p))
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_x3a_encode(self._h, wav.ctypes.data, wav.size, sample_rate, out.ctypes.data, cap, C.byref(n),
                                 stats.ctypes.data)
        return rc, out[: min(n.value, cap)].copy(), stats

    def x3a_decode(self, x3a, wav_cap=None):
        """x3a_to_wav in memory -> (rc, samples, sample_rate, frames_ok, frame_errors)"""
        x3a = np.ascontiguousarray(x3a, dtype=np.uint8)
        if wav_cap is None:
            wav_cap = max(1, x3a.size * 16)
        wav = np.zeros(wav_cap, dtype=np.int16)
        n, rate, fok, ferr = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib().x3_x3a_decode(self._h, x3a.ctypes.data, x3a.size, wav.ctypes.data, wav_cap, C.byref(n),
                                 C.byref(rate), C.byref(fok), C.byref(ferr))
        return rc, wav[: n.value].copy(), rate.value, fok.value, ferr.value


    def wav_to_x3a(self, wav_path, x3a_path):
        """encodefile::wav_to_x3a on files (streamed through the GPU) -> (rc, stats[6])"""
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_wav_to_x3a(self._h, os.fsencode(wav_path), os.fsencode(x3a_path), stats.ctypes.data)
        return rc, stats

    def x3a_to_wav(self, x3a_path, wav_path):
        """decodefile::x3a_to_wav on files -> (rc, samples written, frame_errors)"""
        n, ferr = C.c_uint64(0), C.c_uint64(0)
        rc = lib().x3_x3a_to_wav(self._h, os.fsencode(x3a_path), os.fsencode(wav_path), C.byref(n), C.byref(ferr))
        return rc, n.value, ferr.value

    def crc16(self, data):
        b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) \
            else np.ascontiguousarray(data, dtype=np.uint8)
        crc = C.c_uint16(0)
        rc = lib().x3_crc16(self._h, b.ctypes.data if b.size else None, b.size, C.byref(crc))
        if rc:
            raise X3Error(rc, "x3_crc16: " + self.last_error())
        return crc.value

    # ---- device-resident API (raw device pointers as ints, e.g. torch.Tensor.data_ptr()) -----
    def encode_dev(self, d_wav, n_per_clip, params, d_out, out_cap, start_pos=0, d_frame_offsets=None, n_clips=1,
                   clip_stride=None):
        b = Batch(n_per_clip, n_per_clip if clip_stride is None else clip_stride, n_clips)
        return lib().x3_encode_dev(self._h, d_wav, C.byref(b), C.byref(params), d_out, out_cap, start_pos,
                                   d_frame_offsets)

    def encode_dev_seg(self, d_wav, n_per_clip, params, d_out, out_cap, d_seg_index, seg_blocks, start_pos=0, d_frame_offsets=None,
                       n_clips=1, clip_stride=None):
        """x3_encode_dev_seg: encode and leave the segment index of the stream in d_seg_index"""
        b = Batch(n_per_clip, n_per_clip if clip_stride is None else clip_stride, n_clips)
        return lib().x3_encode_dev_seg(self._h, d_wav, C.byref(b), C.byref(params), d_out, out_cap, start_pos,
                                       d_frame_offsets, d_seg_index, seg_blocks)

    def encode_frames_dev(self, d_wav, src_offsets, src_samples, params, d_out, out_cap, start_pos=0, d_frame_offsets=None):
        """x3_encode_frames_dev: frame f = src_samples[f] samples at d_wav + src_offsets[f] (host arrays)"""
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64)
        sn = np.ascontiguousarray(src_samples, dtype=np.uint32)
        assert so.size == sn.size
        return lib().x3_encode_frames_dev(self._h, d_wav, so.ctypes.data, sn.ctypes.data, so.size, C.byref(params), d_out,
                                          out_cap, start_pos, d_frame_offsets)

    def encode_result(self):
        pos = C.c_uint64(0)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().x3_encode_result(self._h, C.byref(pos), stats.ctypes.data)
        return rc, pos.value, stats

    def decode_dev(self, d_x3, x3_len, d_frame_offsets, n_frames, params, d_wav, wav_cap, n_per_clip=None, n_clips=1,
                   clip_stride=None, d_wav_offsets=None, d_status=None):
        b = None
        if n_per_clip is not None:
            b = C.byref(Batch(n_per_clip, n_per_clip if clip_stride is None else clip_stride, n_clips))
        return lib().x3_decode_dev(self._h, d_x3, x3_len, d_frame_offsets, n_frames, b, d_wav_offsets,
                                   C.byref(params), d_wav, wav_cap, d_status)

    def decode_dev_seg(self, d_x3, x3_len, d_frame_offsets, n_frames, params, d_wav, wav_cap, d_seg_index, seg_blocks,
                       record=False, n_per_clip=None, n_clips=1, clip_stride=None, d_wav_offsets=None, d_status=None):
        """x3_decode_dev_seg: decode by the segment index (record=False) or decode frame by frame and record it"""
        b = None
        if n_per_clip is not None:
            b = C.byref(Batch(n_per_clip, n_per_clip if clip_stride is None else clip_stride, n_clips))
        return lib().x3_decode_dev_seg(self._h, d_x3, x3_len, d_frame_offsets, n_frames, b, d_wav_offsets,
                                       C.byref(params), d_wav, wav_cap, d_status, d_seg_index, seg_blocks, 1 if record else 0)

    # ---- HIP graphs (x3_graph_*): record the device calls made between graph_begin() and graph_end(), replay them
    def graph_begin(self):
        rc = lib().x3_graph_begin(self._h)
        if rc:
            raise X3Error(rc, "x3_graph_begin: " + self.last_error())

    def graph_end(self):
        g = C.c_void_p()
        rc = lib().x3_graph_end(self._h, C.byref(g))
        if rc:
            raise X3Error(rc, "x3_graph_end: " + self.last_error())
        return g

    def graph_launch(self, g):
        rc = lib().x3_graph_launch(self._h, g)
        if rc:
            raise X3Error(rc, "x3_graph_launch: " + self.last_error())

    def graph_destroy(self, g):
        lib().x3_graph_destroy(g)

    def decode_result(self):
        fb, st, nb = C.c_uint64(0), C.c_int(0), C.c_uint64(0)
        rc = lib().x3_decode_result(self._h, C.byref(fb), C.byref(st), C.byref(nb))
        return rc, fb.value, st.value, nb.value

    def index_dev(self, d_x3, x3_len, max_frames, d_frame_offsets, d_wav_offsets):
        """GPU-side frame walk -> (rc, n_frames, n_samples, terminal)"""
        nf, ns, term = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        rc = lib().x3_index_dev(self._h, d_x3, x3_len, max_frames, d_frame_offsets, d_wav_offsets, C.byref(nf),
                                C.byref(ns), C.byref(term))
        return rc, nf.value, ns.value, term.value

    def decode_stream_dev(self, d_x3, x3_len, params, d_wav, wav_cap):
        """x3_decode_stream on device buffers -> (rc, n_samples, frames_ok, frame_errors)"""
        n, fok, ferr = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib().x3_decode_stream_dev(self._h, d_x3, x3_len, C.byref(params), d_wav, wav_cap, C.byref(n),
                                        C.byref(fok), C.byref(ferr))
        return rc, n.value, fok.value, ferr.value

    def synth_dev(self, kind, seed, start, n, d_out):
        rc = lib().x3_synth_dev(self._h, kind, seed, start, n, d_out)
        if rc:
            raise X3Error(rc, "x3_synth_dev: " + self.last_error())

    def enable_kernel_timing(self, on=True):
        lib().x3_ctx_enable_kernel_timing(self._h, 1 if on else 0)

    def reset_kernel_time(self):
        lib().x3_ctx_reset_kernel_time(self._h)

    def kernel_time(self, which):
        ms, cnt = C.c_double(0), C.c_uint64(0)
        rc = lib().x3_ctx_kernel_time(self._h, which, C.byref(ms), C.byref(cnt))
        if rc:
            raise X3Error(rc, "x3_ctx_kernel_time")
        return ms.value, cnt.value

    def kernel_times(self, which):
        """every timed launch's own time in ms, oldest first"""
        n = C.c_uint64(0)
        lib().x3_ctx_kernel_times(self._h, which, None, 0, C.byref(n))
        out = (C.c_double * max(1, n.value))()
        rc = lib().x3_ctx_kernel_times(self._h, which, out, n.value, C.byref(n))
        if rc:
            raise X3Error(rc, "x3_ctx_kernel_times")
        return [out[i] for i in range(n.value)]

    def launch_log(self, which):
        """the kernels' own launch log (x3_ctx_launch_log): list of dicts, oldest first"""
        n = C.c_uint64(0)
        out = (C.c_uint32 * (4 * 256))()
        rc = lib().x3_ctx_launch_log(self._h, which, out, 256, C.byref(n))
        if rc:
            raise X3Error(rc, "x3_ctx_launch_log")
        return [{"target_ticks16": out[4 * i], "achieved_ticks16": out[4 * i + 1], "clock_mhz": out[4 * i + 2] / 1000.0,
                 "life_us": out[4 * i + 3] / 100.0} for i in range(min(n.value, 256))]

    def alloc(self, nbytes):
        p = C.c_void_p()
        rc = lib().x3_dev_alloc(self._h, nbytes, C.byref(p))
        if rc:
            raise X3Error(rc, "x3_dev_alloc: " + self.last_error())
        return p.value

    def free(self, ptr):
        lib().x3_dev_free(self._h, ptr)

    def upload(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        rc = lib().x3_dev_upload(self._h, d_dst, arr.ctypes.data, arr.nbytes)
        if rc:
            raise X3Error(rc, "x3_dev_upload: " + self.last_error())

    def download(self, d_src, nbytes, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        rc = lib().x3_dev_download(self._h, out.ctypes.data, d_src, nbytes)
        if rc:
            raise X3Error(rc, "x3_dev_download: " + self.last_error())
        return out
