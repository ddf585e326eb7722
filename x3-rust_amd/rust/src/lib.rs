//! Drop-in for the encode/decode path of the `x3` crate, backed by libx3hip.so (HIP kernels on
//! an MI355X).  Module and item names follow the reference crate (`x3`, `encoder`, `decoder`,
//! `bytewriter`, `crc`, `error`); `bitpacker` / `bitreader` have no host-side counterpart because
//! on the GPU they are per-block LDS scratch + a wavefront scan and a per-lane register bit window.
//!
//! SOURCE ONLY -- not compiled in the build container (no Rust toolchain there).

pub mod ffi {
    //! `extern "C"` declarations of include/x3hip.h (the entry points this shim uses).
    use std::os::raw::{c_char, c_int, c_void};

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
    pub struct x3_ctx {
        _private: [u8; 0],
    }
    extern "C" {
        pub fn x3_strerror(status: c_int) -> *const c_char;
        pub fn x3_ctx_create(device: c_int, ctx: *mut *mut x3_ctx) -> c_int;
        pub fn x3_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, ctx: *mut *mut x3_ctx) -> c_int;
        pub fn x3_ctx_destroy(ctx: *mut x3_ctx);
        pub fn x3_params_default(p: *mut x3_params);
        pub fn x3_params_validate(p: *const x3_params) -> c_int;
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
        pub fn x3_decode_stream(ctx: *mut x3_ctx, x3: *const u8, len: u64, p: *const x3_params, wav: *mut i16,
                                wav_cap: u64, n_out: *mut u64, frames_ok: *mut u64, frame_errors: *mut u64) -> c_int;
        pub fn x3_index_dev(ctx: *mut x3_ctx, d_x3: *const u8, len: u64, max_frames: u64, d_frame_offsets: *mut u64,
                            d_wav_offsets: *mut u64, n_frames: *mut u64, n_samples: *mut u64, terminal: *mut c_int) -> c_int;
        pub fn x3_decode_stream_dev(ctx: *mut x3_ctx, d_x3: *const u8, len: u64, p: *const x3_params, d_wav: *mut i16,
                                    wav_cap: u64, n_out: *mut u64, frames_ok: *mut u64, frame_errors: *mut u64) -> c_int;
        pub fn x3_archive_header_write(sample_rate: u32, p: *const x3_params, out: *mut u8, out_cap: u64,
                                       out_len: *mut u64) -> c_int;
        pub fn x3_archive_header_read(bytes: *const u8, len: u64, sample_rate: *mut u32, p: *mut x3_params,
                                      channels: *mut u8, header_size: *mut u64) -> c_int;
        pub fn x3_x3a_encode(ctx: *mut x3_ctx, wav: *const i16, n: u64, sample_rate: u32, out: *mut u8, out_cap: u64,
                             out_len: *mut u64, stats: *mut u64) -> c_int;
        pub fn x3_x3a_decode(ctx: *mut x3_ctx, x3a: *const u8, len: u64, wav: *mut i16, wav_cap: u64, n_out: *mut u64,
                             sample_rate: *mut u32, frames_ok: *mut u64, frame_errors: *mut u64) -> c_int;
        pub fn x3_wav_to_x3a(ctx: *mut x3_ctx, wav_path: *const c_char, x3a_path: *const c_char, stats: *mut u64) -> c_int;
        pub fn x3_x3a_to_wav(ctx: *mut x3_ctx, x3a_path: *const c_char, wav_path: *const c_char, n_samples: *mut u64,
                             frame_errors: *mut u64) -> c_int;
    }
}

pub mod error {
    /// Same variants, same order as the reference's `error::X3Error` (src/error.rs:27-62);
    /// `Hip` / `BadArg` are the two extra codes of the C ABI (BadArg = the reference would panic).
    #[derive(Debug, Clone, Copy, PartialEq, Eq)]
    #[repr(i32)]
    pub enum X3Error {
        Io = 1, Hound, BitPack, InvalidEncodingThresh, OutOfBoundsInverse, MoreThanOneChannel,
        ArchiveHeaderXMLInvalid, ArchiveHeaderXMLRiceCode, ArchiveHeaderXMLInvalidKey, FrameLength,
        FrameHeaderInvalidKey, FrameHeaderInvalidPayloadLen, FrameHeaderInvalidHeaderCRC,
        FrameHeaderInvalidPayloadCRC, FrameDecodeInvalidBlockLength, FrameDecodeInvalidIndex,
        FrameDecodeInvalidNTOGO, FrameDecodeInvalidFType, FrameDecodeInvalidRiceCode, FrameDecodeInvalidBPF,
        FrameDecodeUnexpectedEnd, ByteWriterInsufficientMemory, Hip, BadArg,
    }
    pub type Result<T> = core::result::Result<T, X3Error>;
    pub(crate) fn check(rc: i32) -> Result<()> {
        if rc == 0 { Ok(()) } else { Err(unsafe { core::mem::transmute::<i32, X3Error>(rc.clamp(1, 24)) }) }
    }
}

/// One GPU + stream + scratch (`x3_ctx`).  `Gpu::new` fails when there is no HIP device: no CPU path.
pub struct Gpu(*mut ffi::x3_ctx);
impl Gpu {
    pub fn new(device: i32) -> error::Result<Self> {
        let mut p = core::ptr::null_mut();
        error::check(unsafe { ffi::x3_ctx_create(device, &mut p) })?;
        Ok(Gpu(p))
    }
    pub(crate) fn raw(&self) -> *mut ffi::x3_ctx { self.0 }
}
impl Drop for Gpu {
    fn drop(&mut self) { unsafe { ffi::x3_ctx_destroy(self.0) } }
}

pub mod x3 {
    use crate::error::{self, X3Error};
    use crate::ffi;

    /// src/x3.rs:81-134 (rice_codes is derived from `codes` inside the library)
    #[derive(Clone, Copy, Debug)]
    pub struct Parameters {
        pub block_len: usize,
        pub blocks_per_frame: usize,
        pub codes: [usize; 3],
        pub thresholds: [usize; 3],
    }
    impl Parameters {
        pub const MAX_BLOCK_LENGTH: usize = 60;
        pub const WAV_BIT_SIZE: usize = 16;
        pub const DEFAULT_BLOCK_LENGTH: usize = 20;
        pub const DEFAULT_RICE_CODES: [usize; 3] = [0, 1, 3];
        pub const DEFAULT_THRESHOLDS: [usize; 3] = [3, 8, 20];
        pub const DEFAULT_BLOCKS_PER_FRAME: usize = 500;
        pub fn new(block_len: usize, blocks_per_frame: usize, codes: [usize; 3], thresholds: [usize; 3])
                   -> Result<Self, X3Error> {
            let p = Parameters { block_len, blocks_per_frame, codes, thresholds };
            error::check(unsafe { ffi::x3_params_validate(&p.c()) })?;
            Ok(p)
        }
        pub(crate) fn c(&self) -> ffi::x3_params {
            ffi::x3_params {
                block_len: self.block_len as u32,
                blocks_per_frame: self.blocks_per_frame as u32,
                codes: [self.codes[0] as u32, self.codes[1] as u32, self.codes[2] as u32],
                thresholds: [self.thresholds[0] as u32, self.thresholds[1] as u32, self.thresholds[2] as u32],
            }
        }
    }
    impl Default for Parameters {
        fn default() -> Self {
            Parameters { block_len: 20, blocks_per_frame: 500, codes: [0, 1, 3], thresholds: [3, 8, 20] }
        }
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
    pub struct IterChannel<I: Iterator<Item = i16>> {
        pub id: u16,
        pub wav: I,
        pub sample_rate: u32,
        pub params: Parameters,
    }
    impl<I: Iterator<Item = i16>> IterChannel<I> {
        pub fn new(id: u16, wav: impl IntoIterator<IntoIter = I>, sample_rate: u32, params: Parameters) -> Self {
            IterChannel { id, wav: wav.into_iter(), sample_rate, params }
        }
    }
    /// src/x3.rs:148-184
    #[derive(Debug, Default, Clone, Copy)]
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
    }
    pub struct Frame {}
    impl Frame {
        pub const MAX_LENGTH: usize = 0x7fe0;
    }
}

pub mod bytewriter {
    //! The reference's `ByteWriter` trait and `SliceByteWriter` (src/bytewriter.rs:14-100), unchanged
    //! in meaning; the encoder below fills a slice in one call instead of byte by byte.
    use crate::error::{Result, X3Error};
    pub trait ByteWriter {
        fn write_all(&mut self, value: &[u8]) -> Result<()>;
        fn stream_position(&mut self) -> Result<u64>;
        /// hand out the whole underlying slice when there is one (zero-copy path)
        fn as_slice_mut(&mut self) -> Option<(&mut [u8], usize)> { None }
        fn set_position(&mut self, _pos: usize) {}
    }
    pub struct SliceByteWriter<'a> {
        slice: &'a mut [u8],
        p_byte: usize,
    }
    impl<'a> SliceByteWriter<'a> {
        pub fn new(slice: &'a mut [u8]) -> Self { SliceByteWriter { slice, p_byte: 0 } }
    }
    impl<'a> ByteWriter for SliceByteWriter<'a> {
        fn write_all(&mut self, value: &[u8]) -> Result<()> {
            if value.len() > self.slice.len() - self.p_byte { return Err(X3Error::ByteWriterInsufficientMemory); }
            self.slice[self.p_byte..self.p_byte + value.len()].copy_from_slice(value);
            self.p_byte += value.len();
            Ok(())
        }
        fn stream_position(&mut self) -> Result<u64> { Ok(self.p_byte as u64) }
        fn as_slice_mut(&mut self) -> Option<(&mut [u8], usize)> { let p = self.p_byte; Some((self.slice, p)) }
        fn set_position(&mut self, pos: usize) { self.p_byte = pos; }
    }
}

pub mod crc {
    use crate::{error, ffi, Gpu};
    /// src/crc.rs:44-47
    pub fn update_crc16(crc: u16, data: &u8) -> u16 { unsafe { ffi::x3_crc16_update(crc, *data) } }
    /// src/crc.rs:49-58, computed on the GPU
    pub fn crc16(gpu: &Gpu, data: &[u8]) -> error::Result<u16> {
        let mut c = 0u16;
        error::check(unsafe { ffi::x3_crc16(gpu.raw(), data.as_ptr(), data.len() as u64, &mut c) })?;
        Ok(c)
    }
}

pub mod encoder {
    use crate::bytewriter::ByteWriter;
    use crate::error::{self, X3Error};
    use crate::{ffi, x3, Gpu};

    /// `encoder::encode` (src/encoder.rs:51-111): same meaning, the sample iterator is collected
    /// and the whole stream is produced by one GPU dispatch.
    pub fn encode<I, W>(gpu: &Gpu, channels: &mut [&mut x3::IterChannel<I>], writer: &mut W) -> Result<[u64; 6], X3Error>
    where
        I: Iterator<Item = i16>,
        W: ByteWriter,
    {
        if channels.len() > 1 { return Err(X3Error::MoreThanOneChannel); }
        let ch = &mut channels[0];
        let wav: Vec<i16> = ch.wav.by_ref().collect();
        encode_slice(gpu, &wav, &ch.params, writer)
    }

    /// the README's slice shape: `x3::Channel` + a writer
    pub fn encode_channel<W: ByteWriter>(gpu: &Gpu, ch: &x3::Channel, writer: &mut W) -> Result<[u64; 6], X3Error> {
        encode_slice(gpu, ch.wav, &ch.params, writer)
    }

    fn encode_slice<W: ByteWriter>(gpu: &Gpu, wav: &[i16], params: &x3::Parameters, writer: &mut W)
                                   -> Result<[u64; 6], X3Error> {
        let p = params.c();
        let mut stats = [0u64; 6];
        let mut pos = 0u64;
        if let Some((slice, start)) = writer.as_slice_mut() {
            error::check(unsafe {
                ffi::x3_encode(gpu.raw(), wav.as_ptr(), wav.len() as u64, 1, &p, slice.as_mut_ptr(), slice.len() as u64,
                               start as u64, &mut pos, stats.as_mut_ptr())
            })?;
            writer.set_position(pos as usize);
            return Ok(stats);
        }
        let parity = (writer.stream_position()? & 1) as usize;
        let mut buf = vec![0u8; parity + unsafe { ffi::x3_encode_bound(wav.len() as u64, &p) } as usize + 64];
        error::check(unsafe {
            ffi::x3_encode(gpu.raw(), wav.as_ptr(), wav.len() as u64, 1, &p, buf.as_mut_ptr(), buf.len() as u64,
                           parity as u64, &mut pos, stats.as_mut_ptr())
        })?;
        writer.write_all(&buf[parity..pos as usize])?;
        Ok(stats)
    }

    /// src/encoder.rs:122-162
    pub fn write_frame_header(num_samples: usize, id: u8, payload_len: usize, payload_crc: u16) -> [u8; 20] {
        let mut h = [0u8; 20];
        unsafe { ffi::x3_write_frame_header(num_samples as u64, id, payload_len as u64, payload_crc, h.as_mut_ptr()) };
        h
    }
}

pub mod decoder {
    use crate::error::{self, X3Error};
    use crate::{ffi, x3, Gpu};

    /// src/decoder.rs:69-118
    pub fn read_frame_header(bytes: &[u8]) -> Result<x3::FrameHeader, X3Error> {
        let mut h = ffi::x3_frame_header::default();
        error::check(unsafe { ffi::x3_read_frame_header(bytes.as_ptr(), bytes.len() as u64, &mut h) })?;
        Ok(x3::FrameHeader { source_id: h.source_id, samples: h.samples, channels: h.channels,
                             payload_len: h.payload_len as usize, payload_crc: h.payload_crc })
    }

    /// src/decoder.rs:36-58
    pub fn decode_frame(gpu: &Gpu, x3_bytes: &mut [u8], wav_buf: &mut [i16], params: &x3::Parameters, samples: usize)
                        -> Result<Option<usize>, X3Error> {
        let mut n = 0u64;
        error::check(unsafe {
            ffi::x3_decode_frame(gpu.raw(), x3_bytes.as_ptr(), x3_bytes.len() as u64, wav_buf.as_mut_ptr(),
                                 wav_buf.len() as u64, &params.c(), samples as u64, &mut n)
        })?;
        Ok(Some(n as usize))
    }

    /// the `X3aReader::decode_next_frame` loop (src/decodefile.rs:105-136,200-209) over memory:
    /// (samples decoded, good frames, counted frame errors)
    pub fn decode_stream(gpu: &Gpu, x3: &[u8], params: &x3::Parameters, wav: &mut [i16])
                         -> Result<(usize, usize, usize), X3Error> {
        let (mut n, mut ok, mut bad) = (0u64, 0u64, 0u64);
        error::check(unsafe {
            ffi::x3_decode_stream(gpu.raw(), x3.as_ptr(), x3.len() as u64, &params.c(), wav.as_mut_ptr(),
                                  wav.len() as u64, &mut n, &mut ok, &mut bad)
        })?;
        Ok((n as usize, ok as usize, bad as usize))
    }
}

/// encodefile.rs / decodefile.rs on buffers: what `wav_to_x3a` (encodefile.rs:48-77) and `x3a_to_wav`
/// (decodefile.rs:189-212) do between their file reads and writes.
pub mod archive {
    use super::{error::{self, X3Error}, ffi, Gpu};

    pub fn wav_to_x3a(gpu: &Gpu, wav: &[i16], sample_rate: u32, out: &mut [u8]) -> Result<usize, X3Error> {
        let mut len = 0u64;
        error::check(unsafe {
            ffi::x3_x3a_encode(gpu.raw(), wav.as_ptr(), wav.len() as u64, sample_rate, out.as_mut_ptr(),
                               out.len() as u64, &mut len, std::ptr::null_mut())
        })?;
        Ok(len as usize)
    }

    /// Returns (samples, sample_rate, frame_errors).
    pub fn x3a_to_wav(gpu: &Gpu, x3a: &[u8], wav: &mut [i16]) -> Result<(usize, u32, usize), X3Error> {
        let (mut n, mut ok, mut bad, mut rate) = (0u64, 0u64, 0u64, 0u32);
        error::check(unsafe {
            ffi::x3_x3a_decode(gpu.raw(), x3a.as_ptr(), x3a.len() as u64, wav.as_mut_ptr(), wav.len() as u64,
                               &mut n, &mut rate, &mut ok, &mut bad)
        })?;
        Ok((n as usize, rate, bad as usize))
    }
}

/// src/encodefile.rs:48-77 on files, streamed through the GPU in chunks (same name, same arguments plus
/// the device handle).
pub mod encodefile {
    use super::{error::{self, X3Error}, ffi, Gpu};
    use std::{ffi::CString, path::Path};

    pub fn wav_to_x3a<P: AsRef<Path>>(gpu: &Gpu, wav_filename: P, x3a_filename: P) -> Result<(), X3Error> {
        let a = CString::new(wav_filename.as_ref().to_str().ok_or(X3Error::Io)?).map_err(|_| X3Error::Io)?;
        let b = CString::new(x3a_filename.as_ref().to_str().ok_or(X3Error::Io)?).map_err(|_| X3Error::Io)?;
        let mut stats = [0u64; 6];
        error::check(unsafe { ffi::x3_wav_to_x3a(gpu.raw(), a.as_ptr(), b.as_ptr(), stats.as_mut_ptr()) })?;
        // the block `encoder::encode` prints under `std` (src/encoder.rs:96-108)
        let t = stats.iter().sum::<u64>() as f32;
        let pc = |k: usize| (stats[k] as f32 / t) * 100.0;
        println!("\nStatistics:\n  Rice-0: {:.4}%\n  Rice-1: {:.4}%\n  Rice-2: {:.4}%\n  Rice-3: {:.4}%\n  BFP: {:.4}%\n  Pass-through {:.4}%\n",
                 pc(0), pc(1), pc(2), pc(3), pc(4), pc(5));
        Ok(())
    }
}

/// src/decodefile.rs:189-227 on files.
pub mod decodefile {
    use super::{error::{self, X3Error}, ffi, Gpu};
    use std::{ffi::CString, path::Path};

    pub fn x3a_to_wav<P: AsRef<Path>>(gpu: &Gpu, x3a_filename: P, wav_filename: P) -> Result<(), X3Error> {
        let a = CString::new(x3a_filename.as_ref().to_str().ok_or(X3Error::Io)?).map_err(|_| X3Error::Io)?;
        let b = CString::new(wav_filename.as_ref().to_str().ok_or(X3Error::Io)?).map_err(|_| X3Error::Io)?;
        let (mut n, mut bad) = (0u64, 0u64);
        error::check(unsafe { ffi::x3_x3a_to_wav(gpu.raw(), a.as_ptr(), b.as_ptr(), &mut n, &mut bad) })
    }
}
