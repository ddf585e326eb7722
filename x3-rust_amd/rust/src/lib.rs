//! Drop-in for the encode/decode path of the `x3` crate (psiphi75/x3-rust 0.3.1), backed by libx3hip.so: HIP kernels
//! on an MI355X.  Module names, item names, argument lists and result types are the reference crate's, so a caller
//! switches by changing the dependency; the work behind every item runs on the GPU through the C ABI of
//! include/x3hip.h.  There is no CPU path: without a usable HIP device the first call fails with `X3Error::Hip`
//! (items whose reference signature has no error path panic with the library's message).
//!
//! The reference's functions take no device handle.  They run on a process-wide context ([`gpu::with_default`]:
//! device `$X3HIP_DEVICE` or 0, created on first use, serialized by a mutex); [`gpu::Gpu`] and the `*_on` variants
//! exist for callers that manage devices themselves.
//!
//! SOURCE ONLY -- the build container has no Rust toolchain; host/x3.hpp mirrors these shapes in C++ and
//! tests/host_cpp/test_x3_hpp.cpp compiles and runs them on the GPU.

pub mod ffi {
    //! `extern "C"` declarations of include/x3hip.h (the entry points this crate binds).
    use std::os::raw::{c_char, c_int, c_longlong, c_void};

    /// `x3_batch`: `n_clips` clips of `n_per_clip` samples, `clip_stride` samples apart
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct x3_batch {
        pub n_per_clip: u64,
        pub clip_stride: u64,
        pub n_clips: u64,
    }
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct x3_params {
        pub block_len: u32,
        pub blocks_per_frame: u32,
        pub codes: [u32; 3],
        pub thresholds: [u32; 3],
    }
    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct x3_frame_header {
        pub source_id: u8,
        pub channels: u8,
        pub samples: u16,
        pub payload_len: u32,
        pub payload_crc: u16,
    }
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct x3_rice_code {
        pub nsubs: u32,
        pub offset: u32,
        pub len: u32,
        pub inv_len: u32,
        pub code: *const u32,
        pub num_bits: *const u32,
        pub inv: *const i16,
    }
    macro_rules! opaque {
        ($($n:ident),*) => { $(#[repr(C)] pub struct $n { _private: [u8; 0] })* };
    }
    opaque!(x3_ctx, x3_bitreader, x3_bitpacker, x3_reader);

    extern "C" {
        pub fn x3_strerror(status: c_int) -> *const c_char;
        pub fn x3_ctx_create(device: c_int, ctx: *mut *mut x3_ctx) -> c_int;
        pub fn x3_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, ctx: *mut *mut x3_ctx) -> c_int;
        pub fn x3_ctx_destroy(ctx: *mut x3_ctx);
        pub fn x3_last_error(ctx: *const x3_ctx) -> *const c_char;
        pub fn x3_ctx_set_option(ctx: *mut x3_ctx, name: *const c_char, value: c_longlong) -> c_int;
        pub fn x3_params_validate(p: *const x3_params) -> c_int;
        pub fn x3_rice_code_get(code_number: u32, out: *mut x3_rice_code) -> c_int;
        pub fn x3_encode_bound(n: u64, p: *const x3_params) -> u64;
        pub fn x3_crc16(ctx: *mut x3_ctx, data: *const u8, n: u64, crc: *mut u16) -> c_int;
        pub fn x3_crc16_update(crc: u16, byte: u8) -> u16;
        pub fn x3_encode(ctx: *mut x3_ctx, wav: *const i16, n: u64, n_channels: u32, p: *const x3_params,
                         out: *mut u8, out_cap: u64, start_pos: u64, out_pos: *mut u64, stats: *mut u64) -> c_int;
        pub fn x3_encode_frame(ctx: *mut x3_ctx, wav: *const i16, n: u64, p: *const x3_params, out: *mut u8,
                               out_cap: u64, start_pos: u64, out_pos: *mut u64, stats: *mut u64) -> c_int;
        pub fn x3_write_frame_header(num_samples: u64, id: u8, payload_len: u64, payload_crc: u16, out: *mut u8);
        pub fn x3_read_frame_header(bytes: *const u8, len: u64, h: *mut x3_frame_header) -> c_int;
        pub fn x3_decode_frame(ctx: *mut x3_ctx, payload: *const u8, len: u64, wav: *mut i16, wav_cap: u64,
                               p: *const x3_params, samples: u64, n_out: *mut u64) -> c_int;
        pub fn x3_decode_prefetch(ctx: *mut x3_ctx, x3: *const u8, len: u64, p: *const x3_params) -> c_int;
        pub fn x3_decode_stream(ctx: *mut x3_ctx, x3: *const u8, len: u64, p: *const x3_params, wav: *mut i16,
                                wav_cap: u64, n_out: *mut u64, frames_ok: *mut u64, frame_errors: *mut u64) -> c_int;
        pub fn x3_encode_mc(ctx: *mut x3_ctx, wavs: *const *const i16, n_channels: u32, n: u64, p: *const x3_params,
                            out: *mut u8, out_cap: u64, start_pos: u64, out_pos: *mut u64, stats: *mut u64) -> c_int;
        pub fn x3_decode_stream_mc(ctx: *mut x3_ctx, x3: *const u8, len: u64, n_channels: u32, p: *const x3_params,
                                   wavs: *const *mut i16, wav_cap: u64, n_samples: *mut u64, frames_ok: *mut u64,
                                   frame_errors: *mut u64) -> c_int;
        pub fn x3_bitreader_new(ctx: *mut x3_ctx, array: *const u8, len: u64, br: *mut *mut x3_bitreader) -> c_int;
        pub fn x3_bitreader_read_nbits(br: *mut x3_bitreader, n: u32, value: *mut u32) -> c_int;
        pub fn x3_bitreader_count_zero_bits(br: *mut x3_bitreader, count: *mut u32) -> c_int;
        pub fn x3_bitreader_inc_bits(br: *mut x3_bitreader, n: u32) -> c_int;
        pub fn x3_bitreader_free(br: *mut x3_bitreader);
        pub fn x3_decode_block(br: *mut x3_bitreader, wav: *mut i16, n: u32, last_wav: *mut i16, p: *const x3_params) -> c_int;
        pub fn x3_bitpacker_new(ctx: *mut x3_ctx, out: *mut u8, out_cap: u64, start_pos: u64, bp: *mut *mut x3_bitpacker) -> c_int;
        pub fn x3_bitpacker_write_bits(bp: *mut x3_bitpacker, value: u64, num_bits: u32) -> c_int;
        pub fn x3_bitpacker_write_packed_zeros(bp: *mut x3_bitpacker, num_zeros: u32) -> c_int;
        pub fn x3_bitpacker_write_bytes(bp: *mut x3_bitpacker, array: *const u8, n: u64) -> c_int;
        pub fn x3_bitpacker_inc_counter_n_bytes(bp: *mut x3_bitpacker, n_bytes: u64) -> c_int;
        pub fn x3_bitpacker_word_align(bp: *mut x3_bitpacker) -> c_int;
        pub fn x3_bitpacker_finish(bp: *mut x3_bitpacker, len: *mut u64, crc: *mut u16, out_pos: *mut u64) -> c_int;
        pub fn x3_bitpacker_peek(bp: *const x3_bitpacker, len: *mut u64, crc: *mut u16) -> c_int;
        pub fn x3_bitpacker_take(bp: *mut x3_bitpacker, dst: *mut u8, dst_cap: u64, n_new: *mut u64, len: *mut u64,
                                 crc: *mut u16) -> c_int;
        pub fn x3_bitpacker_free(bp: *mut x3_bitpacker);
        // device-resident buffers (mod device)
        pub fn x3_num_frames(n: u64, p: *const x3_params) -> u64;
        pub fn x3_dev_alloc(ctx: *mut x3_ctx, bytes: u64, d_ptr: *mut *mut c_void) -> c_int;
        pub fn x3_dev_free(ctx: *mut x3_ctx, d_ptr: *mut c_void) -> c_int;
        pub fn x3_dev_upload(ctx: *mut x3_ctx, d_dst: *mut c_void, src: *const c_void, bytes: u64) -> c_int;
        pub fn x3_dev_download(ctx: *mut x3_ctx, dst: *mut c_void, d_src: *const c_void, bytes: u64) -> c_int;
        pub fn x3_seg_index_entries(n_frames: u64, p: *const x3_params, seg_blocks: u32) -> u64;
        pub fn x3_encode_dev(ctx: *mut x3_ctx, d_wav: *const i16, batch: *const x3_batch, p: *const x3_params, d_out: *mut u8,
                             out_cap: u64, start_pos: u64, d_frame_offsets: *mut u64) -> c_int;
        pub fn x3_encode_dev_seg(ctx: *mut x3_ctx, d_wav: *const i16, batch: *const x3_batch, p: *const x3_params, d_out: *mut u8,
                                 out_cap: u64, start_pos: u64, d_frame_offsets: *mut u64, d_seg_index: *mut u64,
                                 seg_blocks: u32) -> c_int;
        pub fn x3_encode_result(ctx: *mut x3_ctx, out_pos: *mut u64, stats: *mut u64) -> c_int;
        pub fn x3_decode_dev(ctx: *mut x3_ctx, d_x3: *const u8, x3_len: u64, d_frame_offsets: *const u64, n_frames: u64,
                             batch: *const x3_batch, d_wav_offsets: *const u64, p: *const x3_params, d_wav: *mut i16,
                             wav_cap: u64, d_status: *mut i32) -> c_int;
        pub fn x3_decode_dev_seg(ctx: *mut x3_ctx, d_x3: *const u8, x3_len: u64, d_frame_offsets: *const u64, n_frames: u64,
                                 batch: *const x3_batch, d_wav_offsets: *const u64, p: *const x3_params, d_wav: *mut i16,
                                 wav_cap: u64, d_status: *mut i32, d_seg_index: *mut u64, seg_blocks: u32, record: c_int) -> c_int;
        pub fn x3_decode_result(ctx: *mut x3_ctx, first_bad: *mut u64, first_bad_status: *mut c_int, samples_before: *mut u64) -> c_int;
        pub fn x3_place_buffers(ctx: *mut x3_ctx, d_wav: *const i16, n: u64, p: *const x3_params, d_streams: *const *mut u8,
                                n_streams: u32, cap: u64, d_frame_offsets: *mut u64, d_backs: *const *mut i16, n_backs: u32,
                                warm: u32, steps: u32, ms_per_step: *mut f64) -> c_int;
        pub fn x3_wav_to_x3a(ctx: *mut x3_ctx, wav_path: *const c_char, x3a_path: *const c_char, stats: *mut u64) -> c_int;
        pub fn x3_x3a_to_wav(ctx: *mut x3_ctx, x3a_path: *const c_char, wav_path: *const c_char, n_samples: *mut u64,
                             frame_errors: *mut u64) -> c_int;
        pub fn x3_reader_open(ctx: *mut x3_ctx, x3a_path: *const c_char, reader: *mut *mut x3_reader) -> c_int;
        pub fn x3_reader_spec(reader: *const x3_reader, sample_rate: *mut u32, p: *mut x3_params, channels: *mut u8) -> c_int;
        pub fn x3_reader_next_frame(reader: *mut x3_reader, wav: *mut i16, wav_cap: u64, n_out: *mut u64) -> c_int;
        pub fn x3_reader_frame_errors(reader: *const x3_reader) -> u64;
        pub fn x3_reader_close(reader: *mut x3_reader);
    }
}

pub mod error {
    //! `error::X3Error` with the reference's variants in the reference's order (src/error.rs:27-62), then the two
    //! codes the C ABI adds: `Hip` (a HIP call failed) and `BadArg` (the reference would have panicked).
    use crate::bitpacker::BitPackError;

    pub type Result<T> = core::result::Result<T, X3Error>;

    #[derive(Debug)]
    pub enum X3Error {
        Io(std::io::Error),
        Hound(hound::Error),
        BitPack(BitPackError),

        InvalidEncodingThresh,
        OutOfBoundsInverse,
        MoreThanOneChannel,

        ArchiveHeaderXMLInvalid,
        ArchiveHeaderXMLRiceCode,
        ArchiveHeaderXMLInvalidKey,

        FrameLength,

        FrameHeaderInvalidKey,
        FrameHeaderInvalidPayloadLen,
        FrameHeaderInvalidHeaderCRC,
        FrameHeaderInvalidPayloadCRC,

        FrameDecodeInvalidBlockLength,
        FrameDecodeInvalidIndex,
        FrameDecodeInvalidNTOGO,
        FrameDecodeInvalidFType,
        FrameDecodeInvalidRiceCode,
        FrameDecodeInvalidBPF,
        FrameDecodeUnexpectedEnd,

        ByteWriterInsufficientMemory,

        Hip,
        BadArg,
    }

    impl From<std::io::Error> for X3Error {
        fn from(err: std::io::Error) -> X3Error {
            X3Error::Io(err)
        }
    }
    impl From<hound::Error> for X3Error {
        fn from(err: hound::Error) -> X3Error {
            X3Error::Hound(err)
        }
    }
    impl From<BitPackError> for X3Error {
        fn from(err: BitPackError) -> X3Error {
            X3Error::BitPack(err)
        }
    }

    /// C ABI status (include/x3hip.h:30-61) -> X3Error; the payload-carrying variants get a generic payload
    pub(crate) fn from_status(rc: i32) -> X3Error {
        use X3Error::*;
        match rc {
            1 => Io(std::io::Error::new(std::io::ErrorKind::Other, "libx3hip: file I/O failed")),
            2 => Hound(hound::Error::FormatError("libx3hip: not a 16-bit PCM RIFF/WAVE file")),
            3 => BitPack(BitPackError::ArrayEndReached),
            4 => InvalidEncodingThresh,
            5 => OutOfBoundsInverse,
            6 => MoreThanOneChannel,
            7 => ArchiveHeaderXMLInvalid,
            8 => ArchiveHeaderXMLRiceCode,
            9 => ArchiveHeaderXMLInvalidKey,
            10 => FrameLength,
            11 => FrameHeaderInvalidKey,
            12 => FrameHeaderInvalidPayloadLen,
            13 => FrameHeaderInvalidHeaderCRC,
            14 => FrameHeaderInvalidPayloadCRC,
            15 => FrameDecodeInvalidBlockLength,
            16 => FrameDecodeInvalidIndex,
            17 => FrameDecodeInvalidNTOGO,
            18 => FrameDecodeInvalidFType,
            19 => FrameDecodeInvalidRiceCode,
            20 => FrameDecodeInvalidBPF,
            21 => FrameDecodeUnexpectedEnd,
            22 => ByteWriterInsufficientMemory,
            23 => Hip,
            _ => BadArg,
        }
    }
    pub(crate) fn check(rc: i32) -> Result<()> {
        if rc == 0 { Ok(()) } else { Err(from_status(rc)) }
    }
}

pub mod gpu {
    //! The device handle behind the reference-shaped functions.
    use crate::error::{self, X3Error};
    use crate::ffi;
    use std::sync::{Mutex, OnceLock};

    /// One GPU + stream + scratch (`x3_ctx`).
    pub struct Gpu(*mut ffi::x3_ctx);
    // the context is only ever used by one thread at a time (the default one sits behind a Mutex)
    unsafe impl Send for Gpu {}

    impl Gpu {
        pub fn new(device: i32) -> error::Result<Self> {
            let mut p = core::ptr::null_mut();
            error::check(unsafe { ffi::x3_ctx_create(device, &mut p) })?;
            Ok(Gpu(p))
        }
        pub fn raw(&self) -> *mut ffi::x3_ctx {
            self.0
        }
    }
    impl Drop for Gpu {
        fn drop(&mut self) {
            unsafe { ffi::x3_ctx_destroy(self.0) }
        }
    }

    static DEFAULT: OnceLock<Result<Mutex<Gpu>, i32>> = OnceLock::new();

    /// Run `f` on the process-wide context.  `Err(X3Error::Hip)` when there is no usable HIP device.
    pub fn with_default<T>(f: impl FnOnce(&Gpu) -> error::Result<T>) -> error::Result<T> {
        let slot = DEFAULT.get_or_init(|| {
            let device = std::env::var("X3HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut p = core::ptr::null_mut();
            match unsafe { ffi::x3_ctx_create(device, &mut p) } {
                0 => Ok(Mutex::new(Gpu(p))),
                rc => Err(rc),
            }
        });
        match slot {
            Ok(m) => f(&m.lock().unwrap_or_else(|e| e.into_inner())),
            Err(_) => Err(X3Error::Hip),
        }
    }
}

pub mod x3 {
    //! src/x3.rs
    use crate::error::{self, X3Error};
    use crate::ffi;
    use std::sync::OnceLock;

    /// src/x3.rs:24-27
    pub struct Decoder<'a> {
        pub channels: &'a [Channel<'a>],
        pub x3_inp: &'a mut [u8],
    }

    /// src/x3.rs:29-45
    pub struct Channel<'a> {
        pub id: u16,
        pub wav: &'a [i16],
        pub sample_rate: u32,
        pub params: Parameters,
    }
    impl<'a> Channel<'a> {
        pub fn new(id: u16, wav: &'a [i16], sample_rate: u32, params: Parameters) -> Self {
            Channel { id, wav, sample_rate, params }
        }
    }

    /// src/x3.rs:47-69
    pub struct IterChannel<I>
    where
        I: Iterator<Item = i16>,
    {
        pub id: u16,
        pub wav: I,
        pub sample_rate: u32,
        pub params: Parameters,
    }
    impl<I> IterChannel<I>
    where
        I: Iterator<Item = i16>,
    {
        pub fn new(id: u16, wav: impl IntoIterator<IntoIter = I>, sample_rate: u32, params: Parameters) -> Self {
            IterChannel { id, wav: wav.into_iter(), sample_rate, params }
        }
    }

    /// src/x3.rs:70-79
    pub struct X3aSpec {
        pub sample_rate: u32,
        pub params: Parameters,
        pub channels: u8,
    }

    /// src/x3.rs:81-88
    pub struct Parameters {
        pub block_len: usize,
        pub blocks_per_frame: usize,
        pub codes: [usize; 3],
        pub thresholds: [usize; 3],
        pub rice_codes: [&'static RiceCode; 3],
    }
    impl Parameters {
        pub const MAX_BLOCK_LENGTH: usize = 60;
        pub const WAV_BIT_SIZE: usize = 16;

        pub const DEFAULT_BLOCK_LENGTH: usize = 20;
        pub const DEFAULT_RICE_CODES: [usize; 3] = [0, 1, 3];
        pub const DEFAULT_THRESHOLDS: [usize; 3] = [3, 8, 20];
        pub const DEFAULT_BLOCKS_PER_FRAME: usize = 500;

        /// src/x3.rs:98-122: `InvalidEncodingThresh` if thresholds[k] > offset of code k for k = 0, 1; a code
        /// number > 3 panics on the table index, as in the reference
        pub fn new(block_len: usize, blocks_per_frame: usize, codes: [usize; 3], thresholds: [usize; 3])
                   -> Result<Self, X3Error> {
            let rice_codes = RiceCodes::get(codes);
            for k in 0..2 {
                if thresholds[k] > rice_codes[k].offset {
                    return Err(X3Error::InvalidEncodingThresh);
                }
            }
            Ok(Parameters { block_len, blocks_per_frame, codes, thresholds, rice_codes })
        }

        /// the C ABI's x3_params; values that do not fit 32 bits are `BadArg` (the kernels index with 32 bits)
        pub(crate) fn c(&self) -> error::Result<ffi::x3_params> {
            let n = |v: usize| u32::try_from(v).map_err(|_| X3Error::BadArg);
            Ok(ffi::x3_params {
                block_len: n(self.block_len)?,
                blocks_per_frame: n(self.blocks_per_frame)?,
                codes: [n(self.codes[0])?, n(self.codes[1])?, n(self.codes[2])?],
                thresholds: [n(self.thresholds[0])?, n(self.thresholds[1])?, n(self.thresholds[2])?],
            })
        }
        pub(crate) fn from_c(c: &ffi::x3_params) -> Self {
            let codes = [c.codes[0] as usize, c.codes[1] as usize, c.codes[2] as usize];
            Parameters {
                block_len: c.block_len as usize,
                blocks_per_frame: c.blocks_per_frame as usize,
                codes,
                thresholds: [c.thresholds[0] as usize, c.thresholds[1] as usize, c.thresholds[2] as usize],
                rice_codes: RiceCodes::get(codes),
            }
        }
    }
    impl Default for Parameters {
        fn default() -> Self {
            Parameters {
                block_len: Self::DEFAULT_BLOCK_LENGTH,
                blocks_per_frame: Self::DEFAULT_BLOCKS_PER_FRAME,
                codes: Self::DEFAULT_RICE_CODES,
                thresholds: Self::DEFAULT_THRESHOLDS,
                rice_codes: RiceCodes::get(Self::DEFAULT_RICE_CODES),
            }
        }
    }

    /// src/x3.rs:136-141
    pub struct Archive {}
    impl Archive {
        pub const ID: &'static [u8] = b"X3ARCHIV";
        pub const ID_LEN: usize = 8;
    }
    /// src/x3.rs:143-146
    pub struct Frame {}
    impl Frame {
        pub const MAX_LENGTH: usize = 0x7fe0;
    }
    /// src/x3.rs:148-184
    #[derive(Debug)]
    pub struct FrameHeader {
        pub source_id: u8,
        pub samples: u16,
        pub channels: u8,
        pub payload_len: usize,
        pub payload_crc: u16,
    }
    impl FrameHeader {
        pub const LENGTH: usize = 20;
        pub const KEY: u16 = 30771;
        pub const KEY_BUF: &'static [u8] = b"x3";
        pub const P_KEY: usize = 0;
        pub const P_SOURCE_ID: usize = 2;
        pub const P_CHANNELS: usize = 3;
        pub const P_SAMPLES: usize = 4;
        pub const P_PAYLOAD_SIZE: usize = 6;
        pub const P_TIME: usize = 8;
        pub const P_HEADER_CRC: usize = 16;
        pub const P_PAYLOAD_CRC: usize = 18;
    }

    /// src/x3.rs:187-194; the tables are the library's (x3_rice_code_get), static for the life of the process
    #[allow(dead_code)]
    pub struct RiceCode {
        pub nsubs: usize,
        pub offset: usize,
        pub code: &'static [usize],
        pub num_bits: &'static [usize],
        pub inv: &'static [i16],
        pub inv_len: usize,
    }
    /// src/x3.rs:196-260
    pub struct RiceCodes {}
    impl RiceCodes {
        fn table() -> &'static [RiceCode; 4] {
            static CODE: OnceLock<[RiceCode; 4]> = OnceLock::new();
            CODE.get_or_init(|| {
                core::array::from_fn(|k| {
                    let mut c = core::mem::MaybeUninit::<ffi::x3_rice_code>::uninit();
                    let rc = unsafe { ffi::x3_rice_code_get(k as u32, c.as_mut_ptr()) };
                    assert_eq!(rc, 0);
                    let c = unsafe { c.assume_init() };
                    let widen = |p: *const u32| -> &'static [usize] {
                        let s = unsafe { core::slice::from_raw_parts(p, c.len as usize) };
                        Box::leak(s.iter().map(|&v| v as usize).collect::<Vec<usize>>().into_boxed_slice())
                    };
                    RiceCode {
                        nsubs: c.nsubs as usize,
                        offset: c.offset as usize,
                        code: widen(c.code),
                        num_bits: widen(c.num_bits),
                        inv: unsafe { core::slice::from_raw_parts(c.inv, 60) },
                        inv_len: c.inv_len as usize,
                    }
                })
            })
        }
        pub fn get(code_list: [usize; 3]) -> [&'static RiceCode; 3] {
            let t = Self::table();
            [&t[code_list[0]], &t[code_list[1]], &t[code_list[2]]]
        }
    }
}

pub mod bytewriter {
    //! The crate's byte sinks.  The `ByteWriter` trait is the crate's public interface (five methods, src/bytewriter.rs:
    //! 14-22) and is repeated here so that this mirror compiles on its own; a maintainer who wires the FFI into the crate
    //! keeps the crate's own `src/bytewriter.rs` and drops this module.  The two sinks below are written
    //! for this mirror -- same observable behaviour: zero padding to an absolute position, `ByteWriterInsufficientMemory`
    //! when a slice is too short -- with one hidden, provided method added, through which the encoder hands a slice sink's
    //! memory to the library without an intermediate copy.
    use crate::error::{Result, X3Error};
    pub use std::io::{Seek, SeekFrom, Write};

    pub trait ByteWriter {
        fn align<const N: usize>(&mut self) -> Result<usize>;
        fn write_all(&mut self, value: impl AsRef<[u8]>) -> Result<()>;
        fn flush(&mut self) -> Result<()>;
        fn seek(&mut self, pos: SeekFrom) -> Result<u64>;
        fn stream_position(&mut self) -> Result<u64>;

        /// the whole backing memory of a slice sink (None for everything else)
        #[doc(hidden)]
        fn __x3hip_backing_slice(&mut self) -> Option<&mut [u8]> {
            None
        }
    }

    /// bytes missing from `at` to the next multiple of `n`
    fn gap(at: u64, n: usize) -> usize {
        let n = n as u64;
        ((n - at % n) % n) as usize
    }

    /// A sink over caller-owned memory: a cursor and the furthest byte ever written.
    pub struct SliceByteWriter<'a> {
        buf: &'a mut [u8],
        cursor: usize,
        high_water: usize,
    }

    impl<'a> SliceByteWriter<'a> {
        pub fn new(slice: &'a mut [u8]) -> Self {
            Self { buf: slice, cursor: 0, high_water: 0 }
        }

        /// the window [cursor, cursor + n) if the memory reaches that far
        fn claim(&mut self, n: usize) -> Result<&mut [u8]> {
            let end = self.cursor.checked_add(n).filter(|&e| e <= self.buf.len());
            match end {
                None => Err(X3Error::ByteWriterInsufficientMemory),
                Some(end) => {
                    let w = &mut self.buf[self.cursor..end];
                    self.cursor = end;
                    self.high_water = self.high_water.max(end);
                    Ok(w)
                }
            }
        }
    }

    impl<'a> ByteWriter for SliceByteWriter<'a> {
        fn align<const N: usize>(&mut self) -> Result<usize> {
            let pad = gap(self.cursor as u64, N);
            self.claim(pad)?.fill(0);
            Ok(pad)
        }

        fn write_all(&mut self, value: impl AsRef<[u8]>) -> Result<()> {
            let bytes = value.as_ref();
            self.claim(bytes.len())?.copy_from_slice(bytes);
            Ok(())
        }

        fn flush(&mut self) -> Result<()> {
            Ok(())
        }

        fn seek(&mut self, pos: SeekFrom) -> Result<u64> {
            // (a target in front of the memory wraps to a huge number, as `as usize` does in the crate: "insufficient")
            let target = match pos {
                SeekFrom::Start(to) => to as usize,
                SeekFrom::Current(by) => (self.cursor as i64).wrapping_add(by) as usize,
                SeekFrom::End(by) => (self.high_water as i64).wrapping_add(by) as usize,
            };
            if target > self.buf.len() {
                return Err(X3Error::ByteWriterInsufficientMemory);
            }
            self.cursor = target;
            self.high_water = self.high_water.max(target);
            Ok(target as u64)
        }

        fn stream_position(&mut self) -> Result<u64> {
            Ok(self.cursor as u64)
        }

        fn __x3hip_backing_slice(&mut self) -> Option<&mut [u8]> {
            Some(&mut *self.buf)
        }
    }

    /// A sink over anything that is `Write + Seek` (a file, a cursor): every call is the inner writer's.
    pub struct StreamByteWriter<'a, W: Write + Seek> {
        inner: &'a mut W,
    }

    impl<'a, W: Write + Seek> StreamByteWriter<'a, W> {
        pub fn new(writer: &'a mut W) -> Self {
            Self { inner: writer }
        }
    }

    impl<'a, W: Write + Seek> ByteWriter for StreamByteWriter<'a, W> {
        fn align<const N: usize>(&mut self) -> Result<usize> {
            let pad = gap(self.inner.stream_position()?, N);
            for _ in 0..pad {
                self.inner.write_all(&[0u8])?;
            }
            Ok(pad)
        }

        fn write_all(&mut self, value: impl AsRef<[u8]>) -> Result<()> {
            Ok(self.inner.write_all(value.as_ref())?)
        }

        fn flush(&mut self) -> Result<()> {
            Ok(self.inner.flush()?)
        }

        fn seek(&mut self, pos: SeekFrom) -> Result<u64> {
            Ok(self.inner.seek(pos)?)
        }

        fn stream_position(&mut self) -> Result<u64> {
            Ok(self.inner.stream_position()?)
        }
    }
}

pub mod crc {
    //! src/crc.rs
    use crate::{error, ffi, gpu};

    /// src/crc.rs:44-47
    pub fn update_crc16(crc: u16, data: &u8) -> u16 {
        unsafe { ffi::x3_crc16_update(crc, *data) }
    }
    /// src/crc.rs:49-58, computed on the GPU (segmented reduction).  The reference's signature has no error
    /// path: a HIP failure panics.
    pub fn crc16(data: &[u8]) -> u16 {
        gpu::with_default(|g| crc16_on(g, data)).expect("x3::crc::crc16: no usable HIP device")
    }
    pub fn crc16_on(gpu: &gpu::Gpu, data: &[u8]) -> error::Result<u16> {
        let mut c = 0u16;
        error::check(unsafe { ffi::x3_crc16(gpu.raw(), data.as_ptr(), data.len() as u64, &mut c) })?;
        Ok(c)
    }
}

pub mod bitpacker {
    //! src/bitpacker.rs:36-190.  Fields are recorded in the library and packed by a kernel (scan of the widths,
    //! fields OR-ed into place) when the packer flushes: on `word_align`, `inc_counter_n_bytes` and Drop, where the
    //! reference has handed every complete byte to the writer too.  `len()` / `crc()` are the reference's values
    //! between writes.  The frame encoder does not go through this type.
    use crate::bytewriter::{ByteWriter, SeekFrom};
    use crate::error::{self, Result, X3Error};
    use crate::{ffi, gpu};

    #[derive(Debug)]
    pub enum BitPackError {
        NotByteAligned,
        BoundaryReached,
        ArrayEndReached,
        ExceededBitBoundary,
    }

    pub struct BitPacker<'a, W: ByteWriter> {
        writer: &'a mut W,
        bp: *mut ffi::x3_bitpacker, // unbound packer (out = NULL): bytes come back through x3_bitpacker_take
        bits: usize,                // bits written since new()
        handed: usize,              // bytes the writer already has
    }

    impl<'a, W: ByteWriter> Drop for BitPacker<'a, W> {
        fn drop(&mut self) {
            if self.bits % 8 != 0 || self.bits / 8 > self.handed {
                let _ = self.flush();
            }
            unsafe { ffi::x3_bitpacker_free(self.bp) };
        }
    }

    impl<'a, W: ByteWriter> BitPacker<'a, W> {
        pub fn new(writer: &'a mut W) -> BitPacker<'a, W> {
            // word_align pads to an even ABSOLUTE writer position: the library needs the position at new()
            let start = writer.stream_position().unwrap_or(0);
            let mut bp = core::ptr::null_mut();
            gpu::with_default(|g| {
                error::check(unsafe { ffi::x3_bitpacker_new(g.raw(), core::ptr::null_mut(), 0, start, &mut bp) })
            })
            .expect("x3::bitpacker::BitPacker::new: no usable HIP device");
            BitPacker { writer, bp, bits: 0, handed: 0 }
        }

        pub fn crc(&self) -> u16 {
            let mut c = 0xffffu16;
            gpu::with_default(|_| error::check(unsafe { ffi::x3_bitpacker_peek(self.bp, core::ptr::null_mut(), &mut c) }))
                .expect("x3::bitpacker::BitPacker::crc: HIP error");
            c
        }

        /// flush (src/bitpacker.rs:79-86): a partial byte is zero-padded; every byte the writer does not have yet is
        /// packed on the GPU and written
        fn flush(&mut self) -> Result<()> {
            let pending = (self.bits + 7) / 8 - self.handed;
            let mut buf = vec![0u8; pending];
            let mut n_new = 0u64;
            gpu::with_default(|_| {
                error::check(unsafe {
                    ffi::x3_bitpacker_take(self.bp, buf.as_mut_ptr(), buf.len() as u64, &mut n_new,
                                           core::ptr::null_mut(), core::ptr::null_mut())
                })
            })?;
            self.bits = (self.bits + 7) & !7;
            self.handed += n_new as usize;
            self.writer.write_all(&buf[..n_new as usize])
        }

        pub fn len(&self) -> usize {
            self.bits / 8
        }

        pub fn write_bytes(&mut self, array: &[u8]) -> Result<()> {
            // src/bitpacker.rs:95-102: the array reaches the writer at once, in front of a partial byte that is still
            // in the packer -- x3_bitpacker_write_bytes keeps that order
            error::check(unsafe { ffi::x3_bitpacker_write_bytes(self.bp, array.as_ptr(), array.len() as u64) })?;
            self.bits += 8 * array.len();
            Ok(())
        }

        pub fn inc_counter_n_bytes(&mut self, n_bytes: usize) -> Result<()> {
            if self.bits % 8 != 0 {
                return Err(X3Error::BitPack(BitPackError::NotByteAligned));
            }
            self.flush()?;
            self.writer.seek(SeekFrom::Current(n_bytes as i64))?;
            Ok(())
        }

        pub fn word_align(&mut self) -> Result<()> {
            error::check(unsafe { ffi::x3_bitpacker_word_align(self.bp) })?;
            let mut len = 0u64;
            error::check(unsafe { ffi::x3_bitpacker_peek(self.bp, &mut len, core::ptr::null_mut()) })?;
            self.bits = len as usize * 8;
            self.flush()
        }

        pub fn write_bits(&mut self, value: usize, num_bits: usize) -> Result<()> {
            let n = u32::try_from(num_bits).map_err(|_| X3Error::BadArg)?;
            error::check(unsafe { ffi::x3_bitpacker_write_bits(self.bp, value as u64, n) })?;
            self.bits += num_bits;
            Ok(())
        }

        pub fn write_packed_zeros(&mut self, num_zeros: usize) -> Result<()> {
            let n = u32::try_from(num_zeros).map_err(|_| X3Error::BadArg)?;
            error::check(unsafe { ffi::x3_bitpacker_write_packed_zeros(self.bp, n) })?;
            self.bits += num_zeros;
            Ok(())
        }
    }
}

pub mod bitreader {
    //! src/bitreader.rs:51-176.  The array and the reader's state (idx, leading_word, rem_bit) live in device
    //! memory; each call runs the reference-exact reader there, reads past the end and the one-word peek of
    //! `count_zero_bits` included.  The frame decoders do not go through this type (they keep a bit window per
    //! lane in registers); it exists for callers of the reference's building blocks.
    use crate::{error, ffi, gpu};
    use core::marker::PhantomData;

    pub struct BitReader<'a> {
        pub(crate) br: *mut ffi::x3_bitreader,
        _array: PhantomData<&'a [u8]>,
    }

    impl<'a> Drop for BitReader<'a> {
        fn drop(&mut self) {
            unsafe { ffi::x3_bitreader_free(self.br) }
        }
    }

    impl<'a> BitReader<'a> {
        pub fn new(array: &'a [u8]) -> Self {
            let mut br = core::ptr::null_mut();
            gpu::with_default(|g| {
                error::check(unsafe { ffi::x3_bitreader_new(g.raw(), array.as_ptr(), array.len() as u64, &mut br) })
            })
            .expect("x3::bitreader::BitReader::new: no usable HIP device");
            BitReader { br, _array: PhantomData }
        }

        pub fn inc_bits(&mut self, n: usize) {
            gpu::with_default(|_| error::check(unsafe { ffi::x3_bitreader_inc_bits(self.br, n as u32) }))
                .expect("x3::bitreader::BitReader::inc_bits");
        }

        pub fn read_nbits(&mut self, n: usize) -> u32 {
            let mut v = 0u32;
            gpu::with_default(|_| error::check(unsafe { ffi::x3_bitreader_read_nbits(self.br, n as u32, &mut v) }))
                .expect("x3::bitreader::BitReader::read_nbits");
            v
        }

        pub fn count_zero_bits(&mut self) -> usize {
            let mut v = 0u32;
            gpu::with_default(|_| error::check(unsafe { ffi::x3_bitreader_count_zero_bits(self.br, &mut v) }))
                .expect("x3::bitreader::BitReader::count_zero_bits");
            v as usize
        }
    }
}

pub mod encoder {
    //! src/encoder.rs
    use crate::bytewriter::ByteWriter;
    use crate::error::{self, X3Error};
    use crate::{ffi, gpu, x3};
    use std::cell::Cell;

    thread_local! {
        static LAST_STATS: Cell<[usize; 6]> = const { Cell::new([0; 6]) };
    }
    /// Block counts by type (Rice-0, Rice-1, Rice-2, Rice-3, BFP, pass-through) of this thread's last `encode`:
    /// what the reference prints under `std` (src/encoder.rs:96-108) and otherwise drops.
    pub fn last_statistics() -> [usize; 6] {
        LAST_STATS.with(|s| s.get())
    }

    /// `encoder::encode` (src/encoder.rs:51-111).  The sample iterator is drained, the whole stream is produced
    /// by one GPU dispatch; frames, padding and bytes are those of the reference's frame loop.
    pub fn encode<'a, I, W: ByteWriter>(channels: &mut [&mut x3::IterChannel<I>], writer: &mut W) -> Result<(), X3Error>
    where
        I: Iterator<Item = i16>,
    {
        if channels.len() > 1 {
            return Err(X3Error::MoreThanOneChannel);
        }
        let ch = &mut channels[0];
        let wav: Vec<i16> = ch.wav.by_ref().collect();
        let mut stats = [0usize; 6];
        gpu::with_default(|g| encode_samples(g, &wav, writer, &ch.params, &mut stats, false))?;
        LAST_STATS.with(|s| s.set(stats));
        #[cfg(feature = "std")]
        {
            let total: f32 = stats.iter().sum::<usize>() as f32;
            println!("");
            println!("Statistics:");
            println!("  Rice-0: {:.4}%", (stats[0] as f32 / total) * 100.0);
            println!("  Rice-1: {:.4}%", (stats[1] as f32 / total) * 100.0);
            println!("  Rice-2: {:.4}%", (stats[2] as f32 / total) * 100.0);
            println!("  Rice-3: {:.4}%", (stats[3] as f32 / total) * 100.0);
            println!("  BFP: {:.4}%", (stats[4] as f32 / total) * 100.0);
            println!("  Pass-through {:.4}%", (stats[5] as f32 / total) * 100.0);
            println!("");
        }
        Ok(())
    }

    /// src/encoder.rs:122-162
    pub fn write_frame_header(num_samples: usize, id: u8, payload_len: usize, payload_crc: u16) -> [u8; x3::FrameHeader::LENGTH] {
        let mut h = [0u8; x3::FrameHeader::LENGTH];
        unsafe { ffi::x3_write_frame_header(num_samples as u64, id, payload_len as u64, payload_crc, h.as_mut_ptr()) };
        h
    }

    /// `encoder::encode_frame` (src/encoder.rs:175-214): one frame from `wav`, header included, at the writer's
    /// position padded to even; `stats` is added to.
    pub fn encode_frame<W: ByteWriter>(wav: &[i16], writer: &mut W, params: &x3::Parameters, stats: &mut [usize; 6])
                                       -> Result<(), X3Error> {
        gpu::with_default(|g| encode_samples(g, wav, writer, params, stats, true))
    }

    /// the same two on a context of the caller's
    pub fn encode_samples<W: ByteWriter>(gpu: &gpu::Gpu, wav: &[i16], writer: &mut W, params: &x3::Parameters,
                                         stats: &mut [usize; 6], one_frame: bool) -> Result<(), X3Error> {
        let p = params.c()?;
        let mut st = [0u64; 6];
        let mut pos = 0u64;
        let n = wav.len() as u64;
        let start = writer.stream_position()?;
        let call = |out: *mut u8, cap: u64, at: u64, pos: &mut u64, st: &mut [u64; 6]| unsafe {
            if one_frame {
                ffi::x3_encode_frame(gpu.raw(), wav.as_ptr(), n, &p, out, cap, at, pos, st.as_mut_ptr())
            } else {
                ffi::x3_encode(gpu.raw(), wav.as_ptr(), n, 1, &p, out, cap, at, pos, st.as_mut_ptr())
            }
        };
        let direct = match writer.__x3hip_backing_slice() {
            // a slice writer: the library writes into the caller's memory, behind the writer's position
            Some(slice) => Some(call(slice.as_mut_ptr(), slice.len() as u64, start, &mut pos, &mut st)),
            None => None,
        };
        match direct {
            Some(rc) => {
                // On ByteWriterInsufficientMemory the slice holds every frame that fits and *out_pos stands behind the last
                // of them (include/x3hip.h, x3_encode): the writer must stand there too before the error goes up, as the
                // reference's SliceByteWriter does after the frames it has taken (src/bytewriter.rs:86-99)
                if rc == 22 {
                    writer.seek(crate::bytewriter::SeekFrom::Start(pos))?;
                }
                error::check(rc)?;
                writer.seek(crate::bytewriter::SeekFrom::Start(pos))?;
            }
            None => {
                // any other writer: encode into a buffer that starts at the same parity, hand the bytes over
                let parity = start & 1;
                let slack = if one_frame { 3 * n + 64 } else { 64 };
                let mut buf = vec![0u8; (parity + unsafe { ffi::x3_encode_bound(n, &p) } + slack) as usize];
                error::check(call(buf.as_mut_ptr(), buf.len() as u64, parity, &mut pos, &mut st))?;
                writer.write_all(&buf[parity as usize..pos as usize])?;
            }
        }
        for k in 0..6 {
            stats[k] += st[k] as usize;
        }
        Ok(())
    }
}

pub mod decoder {
    //! src/decoder.rs
    use crate::bitreader::BitReader;
    use crate::error::{self, X3Error};
    use crate::x3::{self, FrameHeader};
    use crate::{ffi, gpu};

    /// src/decoder.rs:30-34
    pub enum FrameTest {
        IsFrame,
        EndOfBuffer,
        NotFrame,
    }

    /// `decoder::decode_frame` (src/decoder.rs:36-58): `x3_bytes` is one frame's payload, `samples` its header's
    /// sample count; `Ok(Some(samples written))`.  Loops over this function go through a frame cache in the
    /// library; whole streams belong to [`decode_stream`].
    pub fn decode_frame(x3_bytes: &mut [u8], wav_buf: &mut [i16], params: &x3::Parameters, samples: usize)
                        -> Result<Option<usize>, X3Error> {
        let p = params.c()?;
        let mut n = 0u64;
        gpu::with_default(|g| {
            error::check(unsafe {
                ffi::x3_decode_frame(g.raw(), x3_bytes.as_ptr(), x3_bytes.len() as u64, wav_buf.as_mut_ptr(),
                                     wav_buf.len() as u64, &p, samples as u64, &mut n)
            })
        })?;
        Ok(Some(n as usize))
    }

    /// Not in the reference: announce the frame stream a loop over [`decode_frame`] is about to walk.  While the
    /// guard lives, calls whose payload lies in `x3` are served from windows of frames that are checked and decoded
    /// ahead on the GPU (a header parse and a memcpy per call) instead of one dispatch per call; the results are
    /// those of `decode_frame` alone.
    ///
    /// # Safety
    /// `decode_frame` wants `&mut [u8]` payloads, so the guard cannot hold a borrow of the buffer: the caller keeps
    /// `x3` alive and unchanged until the guard is dropped.
    pub unsafe fn prefetch(x3: &[u8], params: &x3::Parameters) -> Result<Prefetch, X3Error> {
        let p = params.c()?;
        gpu::with_default(|g| error::check(ffi::x3_decode_prefetch(g.raw(), x3.as_ptr(), x3.len() as u64, &p)))?;
        Ok(Prefetch(()))
    }
    pub struct Prefetch(());
    impl Drop for Prefetch {
        fn drop(&mut self) {
            let _ = gpu::with_default(|g| {
                error::check(unsafe { ffi::x3_decode_prefetch(g.raw(), core::ptr::null(), 0, core::ptr::null()) })
            });
        }
    }

    /// src/decoder.rs:69-118
    pub fn read_frame_header(bytes: &[u8]) -> Result<FrameHeader, X3Error> {
        let mut h = ffi::x3_frame_header::default();
        error::check(unsafe { ffi::x3_read_frame_header(bytes.as_ptr(), bytes.len() as u64, &mut h) })?;
        Ok(FrameHeader { source_id: h.source_id, samples: h.samples, channels: h.channels,
                         payload_len: h.payload_len as usize, payload_crc: h.payload_crc })
    }

    /// `decoder::decode_block` (src/decoder.rs:132-145): `wav.len()` samples from the reader's position
    pub fn decode_block(br: &mut BitReader, wav: &mut [i16], last_wav: &mut i16, params: &x3::Parameters)
                        -> Result<(), X3Error> {
        let p = params.c()?;
        let n = u32::try_from(wav.len()).map_err(|_| X3Error::BadArg)?;
        gpu::with_default(|_| error::check(unsafe { ffi::x3_decode_block(br.br, wav.as_mut_ptr(), n, last_wav, &p) }))
    }

    /// Not in the reference: the `X3aReader::decode_next_frame` loop (src/decodefile.rs:105-136, 200-209) over a
    /// frame stream in memory in one call -- (samples decoded, good frames, counted frame errors).
    pub fn decode_stream(x3: &[u8], params: &x3::Parameters, wav: &mut [i16]) -> Result<(usize, usize, usize), X3Error> {
        let p = params.c()?;
        let (mut n, mut ok, mut bad) = (0u64, 0u64, 0u64);
        gpu::with_default(|g| {
            error::check(unsafe {
                ffi::x3_decode_stream(g.raw(), x3.as_ptr(), x3.len() as u64, &p, wav.as_mut_ptr(), wav.len() as u64,
                                      &mut n, &mut ok, &mut bad)
            })
        })?;
        Ok((n as usize, ok as usize, bad as usize))
    }
}

pub mod multichannel {
    //! Not in the reference crate, whose `encoder::encode` returns `MoreThanOneChannel` for more than one channel
    //! (src/encoder.rs:55-57) and whose reader refuses such frames (src/decoder.rs:90-94) -- as `encoder::encode` and
    //! `decoder::decode_stream` of this crate do.  The layout is the one the frame header's `<Num Channels>` and
    //! "pack the data block for each channel" (src/encoder.rs:197) foresee; include/x3hip.h, x3_mc.h.
    use crate::error::{self, X3Error};
    use crate::{ffi, gpu, x3};

    /// channels of equal length into `out[start_pos..]`; -> the writer's position behind the last frame
    pub fn encode(channels: &[&[i16]], params: &x3::Parameters, out: &mut [u8], start_pos: usize) -> Result<usize, X3Error> {
        let n = channels.first().ok_or(X3Error::BadArg)?.len();
        if channels.iter().any(|c| c.len() != n) {
            return Err(X3Error::BadArg);
        }
        let p = params.c()?;
        let ptrs: Vec<*const i16> = channels.iter().map(|c| c.as_ptr()).collect();
        let n_ch = u32::try_from(ptrs.len()).map_err(|_| X3Error::BadArg)?;
        let mut pos = 0u64;
        gpu::with_default(|g| {
            error::check(unsafe {
                ffi::x3_encode_mc(g.raw(), ptrs.as_ptr(), n_ch, n as u64, &p, out.as_mut_ptr(), out.len() as u64,
                                  start_pos as u64, &mut pos, std::ptr::null_mut())
            })
        })?;
        Ok(pos as usize)
    }

    /// -> (samples per channel, good frames, counted frame errors); every channel slice must hold `wav_cap` samples
    pub fn decode_stream(x3: &[u8], params: &x3::Parameters, channels: &mut [&mut [i16]]) -> Result<(usize, usize, usize), X3Error> {
        let cap = channels.iter().map(|c| c.len()).min().ok_or(X3Error::BadArg)?;
        let p = params.c()?;
        let ptrs: Vec<*mut i16> = channels.iter_mut().map(|c| c.as_mut_ptr()).collect();
        let n_ch = u32::try_from(ptrs.len()).map_err(|_| X3Error::BadArg)?;
        let (mut n, mut ok, mut bad) = (0u64, 0u64, 0u64);
        gpu::with_default(|g| {
            error::check(unsafe {
                ffi::x3_decode_stream_mc(g.raw(), x3.as_ptr(), x3.len() as u64, n_ch, &p, ptrs.as_ptr(), cap as u64, &mut n,
                                         &mut ok, &mut bad)
            })
        })?;
        Ok((n as usize, ok as usize, bad as usize))
    }
}

pub mod device {
    //! Not in the reference crate, which knows no device: samples and stream stay in HBM between encode and decode
    //! (`x3_encode_dev` / `x3_decode_dev`), optionally with the SEGMENT INDEX that lets a short stream decode on as many lanes
    //! as fill the GPU (`x3_encode_dev_seg` / `x3_decode_dev_seg`; a hint the decoder proves entry by entry).  The C++ mirror
    //! has the same module (host/x3.hpp, `x3::device`), tested in tests/host_cpp/test_x3_hpp.cpp.
    use crate::error::{self, X3Error};
    use crate::ffi;
    use crate::gpu::Gpu;
    use crate::x3;
    use std::os::raw::c_void;

    /// device memory of a context (`x3_dev_alloc`), freed with the value
    pub struct Buffer<'g> {
        gpu: &'g Gpu,
        p: *mut c_void,
        bytes: usize,
    }
    impl<'g> Buffer<'g> {
        pub fn new(gpu: &'g Gpu, bytes: usize) -> error::Result<Self> {
            let mut p = core::ptr::null_mut();
            error::check(unsafe { ffi::x3_dev_alloc(gpu.raw(), bytes as u64, &mut p) })?;
            Ok(Buffer { gpu, p, bytes })
        }
        pub fn len(&self) -> usize {
            self.bytes
        }
        pub fn is_empty(&self) -> bool {
            self.bytes == 0
        }
        pub fn as_ptr<T>(&self) -> *mut T {
            self.p as *mut T
        }
        pub fn upload<T: Copy>(&mut self, src: &[T]) -> error::Result<()> {
            let n = std::mem::size_of_val(src);
            if n > self.bytes {
                return Err(X3Error::BadArg);
            }
            error::check(unsafe { ffi::x3_dev_upload(self.gpu.raw(), self.p, src.as_ptr() as *const c_void, n as u64) })
        }
        pub fn download<T: Copy>(&self, dst: &mut [T]) -> error::Result<()> {
            let n = std::mem::size_of_val(dst);
            if n > self.bytes {
                return Err(X3Error::BadArg);
            }
            error::check(unsafe { ffi::x3_dev_download(self.gpu.raw(), dst.as_mut_ptr() as *mut c_void, self.p, n as u64) })
        }
    }
    impl Drop for Buffer<'_> {
        fn drop(&mut self) {
            unsafe { ffi::x3_dev_free(self.gpu.raw(), self.p) };
        }
    }

    /// an encoded batch in device memory: the stream, where its frames begin, and (`seg_blocks != 0`) the segment index
    pub struct EncodedStream<'g> {
        pub bytes: Buffer<'g>,
        pub frame_offsets: Buffer<'g>,
        pub seg_index: Option<Buffer<'g>>,
        pub len: usize,
        pub n_frames: usize,
        pub n_per_clip: usize,
        pub n_clips: usize,
        pub seg_blocks: u32,
        pub stats: [u64; 6],
    }

    /// `n_clips` clips of `n_per_clip` samples back to back in `d_wav`, each encoded as `encoder::encode` encodes a channel.
    /// `seg_blocks`: 0 = no index; a power of two >= 4 (32 for the default frames) = also leave the segment index.
    pub fn encode<'g>(gpu: &'g Gpu, d_wav: &Buffer<'g>, n_per_clip: usize, n_clips: usize, params: &x3::Parameters,
                      seg_blocks: u32) -> error::Result<EncodedStream<'g>> {
        if n_per_clip == 0 || n_clips == 0 || d_wav.len() < 2 * n_per_clip * n_clips {
            return Err(X3Error::BadArg);
        }
        let p = params.c()?;
        let n_frames = unsafe { ffi::x3_num_frames(n_per_clip as u64, &p) } as usize * n_clips;
        let cap = unsafe { ffi::x3_encode_bound(n_per_clip as u64, &p) } as usize * n_clips;
        let bytes = Buffer::new(gpu, cap + 16)?;
        let frame_offsets = Buffer::new(gpu, 8 * (n_frames + 1))?;
        let n_idx = if seg_blocks != 0 { unsafe { ffi::x3_seg_index_entries(n_frames as u64, &p, seg_blocks) } as usize } else { 0 };
        let seg_index = if n_idx != 0 { Some(Buffer::new(gpu, 8 * n_idx)?) } else { None };
        let b = ffi::x3_batch { n_per_clip: n_per_clip as u64, clip_stride: n_per_clip as u64, n_clips: n_clips as u64 };
        error::check(unsafe {
            match &seg_index {
                Some(idx) => ffi::x3_encode_dev_seg(gpu.raw(), d_wav.as_ptr::<i16>(), &b, &p, bytes.as_ptr::<u8>(), cap as u64, 0,
                                                    frame_offsets.as_ptr::<u64>(), idx.as_ptr::<u64>(), seg_blocks),
                None => ffi::x3_encode_dev(gpu.raw(), d_wav.as_ptr::<i16>(), &b, &p, bytes.as_ptr::<u8>(), cap as u64, 0, frame_offsets.as_ptr::<u64>()),
            }
        })?;
        let mut pos = 0u64;
        let mut stats = [0u64; 6];
        error::check(unsafe { ffi::x3_encode_result(gpu.raw(), &mut pos, stats.as_mut_ptr()) })?;
        let seg_blocks = if seg_index.is_some() { seg_blocks } else { 0 };
        Ok(EncodedStream { bytes, frame_offsets, seg_index, len: pos as usize, n_frames, n_per_clip, n_clips, seg_blocks, stats })
    }

    /// ... and back: every clip's samples at `d_wav + clip * n_per_clip`; by the segment index when the stream has one.
    /// -> samples in front of the first frame that failed (all of them if none did); `Err` = that frame's status
    pub fn decode<'g>(gpu: &'g Gpu, s: &EncodedStream<'g>, params: &x3::Parameters, d_wav: &mut Buffer<'g>) -> error::Result<usize> {
        let p = params.c()?;
        let b = ffi::x3_batch { n_per_clip: s.n_per_clip as u64, clip_stride: s.n_per_clip as u64, n_clips: s.n_clips as u64 };
        let wav_cap = (d_wav.len() / 2) as u64;
        error::check(unsafe {
            match &s.seg_index {
                Some(idx) => ffi::x3_decode_dev_seg(gpu.raw(), s.bytes.as_ptr::<u8>(), s.len as u64, s.frame_offsets.as_ptr::<u64>(), s.n_frames as u64,
                                                    &b, core::ptr::null(), &p, d_wav.as_ptr::<i16>(), wav_cap, core::ptr::null_mut(),
                                                    idx.as_ptr::<u64>(), s.seg_blocks, 0),
                None => ffi::x3_decode_dev(gpu.raw(), s.bytes.as_ptr::<u8>(), s.len as u64, s.frame_offsets.as_ptr::<u64>(), s.n_frames as u64, &b,
                                           core::ptr::null(), &p, d_wav.as_ptr::<i16>(), wav_cap, core::ptr::null_mut()),
            }
        })?;
        let (mut first_bad, mut before, mut st) = (0u64, 0u64, 0);
        error::check(unsafe { ffi::x3_decode_result(gpu.raw(), &mut first_bad, &mut st, &mut before) })?;
        error::check(st)?;
        Ok(before as usize)
    }

    /// Placement (`x3_place_buffers`; profiles/r6/decoder_modes.txt): the round trip timed on every pair of candidate buffers --
    /// `ms[i * backs.len() + j]` for `(streams[i], backs[j])`, `cap` bytes of room in every stream buffer.  A pipeline that keeps its
    /// buffers calls this once and keeps the pair that runs best.
    pub fn place_buffers<'g>(gpu: &'g Gpu, d_wav: &Buffer<'g>, n: usize, params: &x3::Parameters, streams: &[&Buffer<'g>], cap: usize,
                             frame_offsets: &mut Buffer<'g>, backs: &[&Buffer<'g>], warm: u32, steps: u32) -> error::Result<Vec<f64>> {
        let p = params.c()?;
        let s: Vec<*mut u8> = streams.iter().map(|b| b.as_ptr::<u8>()).collect();
        let k: Vec<*mut i16> = backs.iter().map(|b| b.as_ptr::<i16>()).collect();
        let mut ms = vec![0f64; s.len() * k.len()];
        error::check(unsafe {
            ffi::x3_place_buffers(gpu.raw(), d_wav.as_ptr::<i16>(), n as u64, &p, s.as_ptr(), s.len() as u32, cap as u64,
                                  frame_offsets.as_ptr::<u64>(), k.as_ptr(), k.len() as u32, warm, steps, ms.as_mut_ptr())
        })?;
        Ok(ms)
    }
}

pub mod encodefile {
    //! src/encodefile.rs:48-77, streamed through the GPU in chunks
    use crate::error::{self, X3Error};
    use crate::{ffi, gpu};
    use std::{ffi::CString, path};

    pub(crate) fn c_path<P: AsRef<path::Path>>(p: P) -> Result<CString, X3Error> {
        let bad = || X3Error::Io(std::io::Error::new(std::io::ErrorKind::InvalidInput, "path is not UTF-8 / has a NUL"));
        CString::new(p.as_ref().to_str().ok_or_else(bad)?).map_err(|_| bad())
    }

    pub fn wav_to_x3a<P: AsRef<path::Path>>(wav_filename: P, x3a_filename: P) -> Result<(), X3Error> {
        let (a, b) = (c_path(wav_filename)?, c_path(x3a_filename)?);
        let mut stats = [0u64; 6];
        gpu::with_default(|g| error::check(unsafe { ffi::x3_wav_to_x3a(g.raw(), a.as_ptr(), b.as_ptr(), stats.as_mut_ptr()) }))?;
        // the block `encoder::encode` prints under `std` (src/encoder.rs:96-108)
        let t = stats.iter().sum::<u64>() as f32;
        let pc = |k: usize| (stats[k] as f32 / t) * 100.0;
        println!("\nStatistics:\n  Rice-0: {:.4}%\n  Rice-1: {:.4}%\n  Rice-2: {:.4}%\n  Rice-3: {:.4}%\n  BFP: {:.4}%\n  Pass-through {:.4}%\n",
                 pc(0), pc(1), pc(2), pc(3), pc(4), pc(5));
        Ok(())
    }
}

pub mod decodefile {
    //! src/decodefile.rs
    use crate::encodefile::c_path;
    use crate::error::{self, X3Error};
    use crate::x3::{Parameters, X3aSpec};
    use crate::{ffi, gpu};
    use std::path;

    pub const X3_READ_BUFFER_SIZE: usize = 1024 * 24;
    pub const X3_WRITE_BUFFER_SIZE: usize = X3_READ_BUFFER_SIZE * 8;

    /// src/decodefile.rs:47-136.  Behind `decode_next_frame` the library decodes windows of frames ahead on the
    /// GPU and hands them out one per call, with the reference's per-call results.
    pub struct X3aReader {
        reader: *mut ffi::x3_reader,
        spec: X3aSpec,
    }
    impl Drop for X3aReader {
        fn drop(&mut self) {
            unsafe { ffi::x3_reader_close(self.reader) }
        }
    }
    impl X3aReader {
        pub fn open<P: AsRef<path::Path>>(filename: P) -> Result<Self, X3Error> {
            let name = c_path(filename)?;
            let mut reader = core::ptr::null_mut();
            gpu::with_default(|g| error::check(unsafe { ffi::x3_reader_open(g.raw(), name.as_ptr(), &mut reader) }))?;
            let mut p = ffi::x3_params { block_len: 0, blocks_per_frame: 0, codes: [0; 3], thresholds: [0; 3] };
            let (mut sample_rate, mut channels) = (0u32, 0u8);
            unsafe { ffi::x3_reader_spec(reader, &mut sample_rate, &mut p, &mut channels) };
            Ok(X3aReader { reader, spec: X3aSpec { sample_rate, params: Parameters::from_c(&p), channels } })
        }

        pub fn spec(&self) -> &X3aSpec {
            &self.spec
        }

        /// src/decodefile.rs:105-136: `Ok(Some(samples))`, `Ok(None)` at the end of the data / for a payload that
        /// runs past it / for a frame that fails to decode (counted), or the frame's error
        pub fn decode_next_frame(&mut self, wav_buf: &mut [i16; X3_WRITE_BUFFER_SIZE]) -> Result<Option<usize>, X3Error> {
            let mut n = 0u64;
            gpu::with_default(|_| {
                error::check(unsafe {
                    ffi::x3_reader_next_frame(self.reader, wav_buf.as_mut_ptr(), X3_WRITE_BUFFER_SIZE as u64, &mut n)
                })
            })?;
            Ok(if n == 0 { None } else { Some(n as usize) })
        }

        /// frames that failed to decode so far (the reference keeps this private and prints each one)
        pub fn frame_errors(&self) -> usize {
            unsafe { ffi::x3_reader_frame_errors(self.reader) as usize }
        }
    }

    /// src/decodefile.rs:189-212
    pub fn x3a_to_wav<P: AsRef<path::Path>>(x3a_filename: P, wav_filename: P) -> Result<(), X3Error> {
        let (a, b) = (c_path(x3a_filename)?, c_path(wav_filename)?);
        let (mut n, mut bad) = (0u64, 0u64);
        gpu::with_default(|g| error::check(unsafe { ffi::x3_x3a_to_wav(g.raw(), a.as_ptr(), b.as_ptr(), &mut n, &mut bad) }))
    }
}
