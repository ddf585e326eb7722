// x3_bits.h -- the reference's small public items behind the C ABI, for callers (and known-answer tests) written
// against them: `bitreader::BitReader` (src/bitreader.rs:51-176), `decoder::decode_block` (src/decoder.rs:132-145)
// and `bitpacker::BitPacker` (src/bitpacker.rs:46-177).
//
// They are not how the bulk paths work (there the reader is a per-lane register window and the packer a prefix scan
// over block bit lengths), and a call per field is not how a GPU should be driven; they exist so that code and test
// vectors written against the reference's modules reach THIS library, not a CPU re-implementation:
//   * BitReader: the reader's state {idx, leading_word, rem_bit} lives in device memory next to a copy of the
//     array; every call runs the reference-exact reader of x3_decode_replay.h (one thread) and brings the result back;
//   * decode_block: the same reader state, one block through x3_replay_block;
//   * BitPacker: write_bits / write_packed_zeros / word_align are RECORDED on the host (a field list; the alignment
//     arithmetic of word_align is position bookkeeping, bitpacker.rs:124-132); finish packs all fields at once on the
//     GPU -- exclusive scan of the field widths, every field OR-ed into place -- and computes the running CRC-16
//     the reference keeps per flushed byte as one reduction over the bytes.
#pragma once

struct X3ReaderState {  // device mirror of BitReader's fields
  uint32_t idx, word, rem, pad;
};

struct x3_bitreader {
  x3_ctx* c = nullptr;
  uint8_t* d_array = nullptr;
  uint64_t len = 0;
  X3ReaderState* d_state = nullptr;
  uint32_t* d_out = nullptr;    // [0] value / count / status, then samples of decode_block
  uint32_t* h_out = nullptr;    // pinned
  X3ReaderState* h_state = nullptr;
};

// op: 0 new, 1 read_nbits(n), 2 count_zero_bits, 3 inc_bits(n), 4 decode_block(n samples; out[1] = last_wav in/out)
__global__ void x3_bitreader_op_kernel(const uint8_t* __restrict__ array, uint32_t len, X3ReaderState* __restrict__ st,
                                       uint32_t op, uint32_t n, X3DevParams p, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  X3RefReader br;
  br.a = array;
  br.len = len;
  if (op == 0) {
    br.open(array, len);
  } else {
    br.idx = st->idx;
    br.word = st->word;
    br.rem = st->rem;
  }
  uint32_t r = 0;
  if (op == 1) r = br.bits(n);
  else if (op == 2) r = br.zeros();
  else if (op == 3) br.skip(n);
  else if (op == 4) {
    uint32_t last = out[1] & 0xFFFFu;
    r = (uint32_t)x3_replay_block(br, n, p, last, reinterpret_cast<int16_t*>(out + 2));
    out[1] = last;
  }
  out[0] = r;
  st->idx = br.idx;
  st->word = br.word;
  st->rem = br.rem;
}

extern "C" void x3_bitreader_free(x3_bitreader* b) {
  if (!b) return;
  if (b->c) (void)hipSetDevice(b->c->device);
  if (b->d_array) (void)x3_dfree(b->d_array);
  if (b->d_state) (void)x3_dfree(b->d_state);
  if (b->d_out) (void)x3_dfree(b->d_out);
  if (b->h_out) (void)hipHostFree(b->h_out);
  if (b->h_state) (void)hipHostFree(b->h_state);
  delete b;
}

static int bitreader_op(x3_bitreader* b, uint32_t op, uint32_t n, const X3DevParams& dp, uint32_t fetch_dw) {
  x3_ctx* c = b->c;
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(x3_bitreader_op_kernel, dim3(1), dim3(64), 0, c->stream, (const uint8_t*)b->d_array, (uint32_t)b->len,
                     b->d_state, op, n, dp, b->d_out);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(b->h_out, b->d_out, fetch_dw * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(b->h_state, b->d_state, sizeof(X3ReaderState), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return X3_OK;
}

// BitReader::new (src/bitreader.rs:65-74)
extern "C" int x3_bitreader_new(x3_ctx* c, const uint8_t* array, uint64_t len, x3_bitreader** out) {
  if (!c || (!array && len) || !out || len > 0x7FFFFFFFull) return X3_ERR_BAD_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  x3_bitreader* b = new x3_bitreader();
  b->c = c;
  b->len = len;
  if (x3_dmalloc(&b->d_array, len + 16) != hipSuccess || x3_dmalloc(&b->d_state, sizeof(X3ReaderState)) != hipSuccess ||
      x3_dmalloc(&b->d_out, (2 + 64) * sizeof(uint32_t)) != hipSuccess ||
      hipHostMalloc(&b->h_out, (2 + 64) * sizeof(uint32_t)) != hipSuccess ||
      hipHostMalloc(&b->h_state, sizeof(X3ReaderState)) != hipSuccess) {
    c->last_error = "x3_bitreader_new: out of memory";
    x3_bitreader_free(b);
    return X3_ERR_HIP;
  }
  if (len && hipMemcpyAsync(b->d_array, array, len, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
    x3_bitreader_free(b);
    return X3_ERR_HIP;
  }
  X3DevParams dp{};
  int rc = bitreader_op(b, 0, 0, dp, 1);
  if (rc) { x3_bitreader_free(b); return rc; }
  *out = b;
  return X3_OK;
}

// BitReader::read_nbits (:105-119), n <= 31 as in the reference (debug_assert in inc_bits)
extern "C" int x3_bitreader_read_nbits(x3_bitreader* b, uint32_t n, uint32_t* value) {
  if (!b || !value || n > 32) return X3_ERR_BAD_ARG;
  X3DevParams dp{};
  int rc = bitreader_op(b, 1, n, dp, 1);
  if (!rc) *value = b->h_out[0];
  return rc;
}
// BitReader::count_zero_bits (:128-139)
extern "C" int x3_bitreader_count_zero_bits(x3_bitreader* b, uint32_t* count) {
  if (!b || !count) return X3_ERR_BAD_ARG;
  X3DevParams dp{};
  int rc = bitreader_op(b, 2, 0, dp, 1);
  if (!rc) *count = b->h_out[0];
  return rc;
}
// BitReader::inc_bits (:76-92)
extern "C" int x3_bitreader_inc_bits(x3_bitreader* b, uint32_t n) {
  if (!b) return X3_ERR_BAD_ARG;
  X3DevParams dp{};
  return bitreader_op(b, 3, n, dp, 1);
}
// the private fields, for tests that follow the reference's own (idx, leading_word, rem_bit assertions)
extern "C" int x3_bitreader_state(const x3_bitreader* b, uint64_t* idx, uint32_t* leading_word, uint32_t* rem_bit) {
  if (!b) return X3_ERR_BAD_ARG;
  if (idx) *idx = b->h_state->idx;
  if (leading_word) *leading_word = b->h_state->word;
  if (rem_bit) *rem_bit = b->h_state->rem;
  return X3_OK;
}

// decoder::decode_block (src/decoder.rs:132-145): wav[0..n) from the reader's position; *last_wav in and out
extern "C" int x3_decode_block(x3_bitreader* b, int16_t* wav, uint32_t n, int16_t* last_wav, const x3_params* p) {
  // n <= MAX_BLOCK_LENGTH (x3.rs:90: what an encoder can have produced).  n == 0 is an empty slice: the block's type
  // bits are read, a Rice block is then done and a BFP block fails or panics (x3_replay_block)
  if (!b || !wav || !last_wav || !p || n > 60) return X3_ERR_BAD_ARG;
  x3_ctx* c = b->c;
  X3DevParams dp;
  int rc = derive(p, spf_of(p) > 0xFFFFFFFFull ? 0 : spf_of(p), &dp);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  b->h_out[1] = (uint32_t)(uint16_t)*last_wav;
  HIPCHK(c, hipMemcpyAsync(b->d_out + 1, b->h_out + 1, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  if ((rc = bitreader_op(b, 4, n, dp, 2 + 32))) return rc;
  const int st = (int)b->h_out[0];
  if (st != X3_OK) return st;
  std::memcpy(wav, b->h_out + 2, n * sizeof(int16_t));
  *last_wav = (int16_t)(uint16_t)b->h_out[1];
  return X3_OK;
}

// ------------------------------------------------------------------------------------------------ BitPacker
struct x3_bitpacker {
  x3_ctx* c = nullptr;
  uint8_t* out = nullptr;
  uint64_t out_cap = 0, start_pos = 0;  // start_pos: the writer's position at new()
  uint64_t written = 0;                 // bytes behind start_pos already handed to `out`
  std::vector<uint32_t> val, nb;        // every field recorded since new() (each <= 32 bits)
  // inc_counter_n_bytes: (index in the packed bytes, bytes the writer skipped in front of that byte)
  std::vector<std::pair<uint64_t, uint64_t>> gaps;
  uint64_t gap_total = 0;
};

// pack fields [0, n): field i = the low nb[i] bits of val[i], MSB first, at the exclusive prefix sum of nb.  One
// workgroup; tiles of 256 fields (wave scan + cross-wave partials + running base); dst is zeroed, dword-granular.
__global__ void __launch_bounds__(256)
x3_pack_fields_kernel(const uint32_t* __restrict__ val, const uint32_t* __restrict__ nb, uint32_t n, uint32_t* __restrict__ dst) {
  __shared__ uint32_t part[4];
  __shared__ uint32_t s_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
    const uint32_t i = i0 + tid;
    const uint32_t w = i < n ? nb[i] : 0u;
    const uint32_t incl = x3_wave_incl_scan_dpp(w);
    if (lane == 63) part[wid] = incl;
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t k = 0; k < wid; ++k) base += part[k];
    const uint32_t pos = base + incl - w;
    if (w) {
      const uint32_t v = w >= 32u ? val[i] : (val[i] & ((1u << w) - 1u));  // write_bits masks the value (bitpacker.rs:145-146)
      const uint64_t field = (uint64_t)v << (64u - w - (pos & 31u));        // left-aligned in two dwords at pos / 32
      const uint32_t hi = (uint32_t)(field >> 32), lo = (uint32_t)field;
      if (hi) atomicOr(&dst[pos >> 5], x3_bswap32(hi));
      if (lo) atomicOr(&dst[(pos >> 5) + 1u], x3_bswap32(lo));
    }
    __syncthreads();
    if (tid == 0) s_base += part[0] + part[1] + part[2] + part[3];
    __syncthreads();
  }
}

// BitPacker::new over a SliceByteWriter positioned at start_pos (src/bitpacker.rs:65-73)
extern "C" int x3_bitpacker_new(x3_ctx* c, uint8_t* out, uint64_t out_cap, uint64_t start_pos, x3_bitpacker** bp) {
  if (!c || (!out && out_cap) || !bp || (out && start_pos > out_cap)) return X3_ERR_BAD_ARG;
  x3_bitpacker* b = new x3_bitpacker();
  b->c = c;
  b->out = out;
  b->out_cap = out_cap;
  b->start_pos = start_pos;
  *bp = b;
  return X3_OK;
}
extern "C" void x3_bitpacker_free(x3_bitpacker* b) { delete b; }

// BitPacker::write_bits (:143-163): the low num_bits bits of value, MSB first; num_bits == 0 writes nothing
extern "C" int x3_bitpacker_write_bits(x3_bitpacker* b, uint64_t value, uint32_t num_bits) {
  if (!b || num_bits > 64) return X3_ERR_BAD_ARG;
  if (num_bits > 32) {  // two fields
    b->val.push_back((uint32_t)(value >> 32));
    b->nb.push_back(num_bits - 32);
    num_bits = 32;
  }
  if (num_bits) {
    b->val.push_back((uint32_t)value);
    b->nb.push_back(num_bits);
  }
  return X3_OK;
}
// BitPacker::write_packed_zeros (:174-176)
extern "C" int x3_bitpacker_write_packed_zeros(x3_bitpacker* b, uint32_t num_zeros) {
  if (!b) return X3_ERR_BAD_ARG;
  while (num_zeros) {
    const uint32_t k = num_zeros > 32 ? 32 : num_zeros;
    b->val.push_back(0);
    b->nb.push_back(k);
    num_zeros -= k;
  }
  return X3_OK;
}
static uint64_t bitpacker_bits(const x3_bitpacker* b) {
  uint64_t t = 0;
  for (uint32_t w : b->nb) t += w;
  return t;
}
// BitPacker::write_bytes (:95-102): the bytes go to the writer AT ONCE -- in front of a partial byte that is still in
// the packer's scratch -- and count in len() and crc() from then on.  In the field list that is an insertion in front of
// the last (bits mod 8) bits.
extern "C" int x3_bitpacker_write_bytes(x3_bitpacker* b, const uint8_t* array, uint64_t n) {
  if (!b || (!array && n)) return X3_ERR_BAD_ARG;
  uint32_t need = (uint32_t)(bitpacker_bits(b) & 7u);
  std::vector<std::pair<uint32_t, uint32_t>> tail;  // the pending bits, last field first
  while (need) {
    uint32_t v = b->val.back(), w = b->nb.back();
    if (w < 32u) v &= (1u << w) - 1u;
    if (w <= need) {
      tail.emplace_back(v, w);
      need -= w;
      b->val.pop_back();
      b->nb.pop_back();
    } else {  // the field straddles the byte boundary: its upper part stays, its low `need` bits wait
      tail.emplace_back(v & ((1u << need) - 1u), need);
      b->val.back() = v >> need;
      b->nb.back() = w - need;
      need = 0;
    }
  }
  for (uint64_t i = 0; i < n; ++i) {
    b->val.push_back(array[i]);
    b->nb.push_back(8);
  }
  for (auto it = tail.rbegin(); it != tail.rend(); ++it) {
    b->val.push_back(it->first);
    b->nb.push_back(it->second);
  }
  return X3_OK;
}
// BitPacker::inc_counter_n_bytes (:112-118): the writer skips n bytes (SeekFrom::Current), len() and crc() do not
// move; only on a byte boundary (BitPackError::NotByteAligned), only for a packer bound to a slice
extern "C" int x3_bitpacker_inc_counter_n_bytes(x3_bitpacker* b, uint64_t n_bytes) {
  if (!b || !b->out) return X3_ERR_BAD_ARG;
  const uint64_t bits = bitpacker_bits(b);
  if (bits & 7u) return X3_ERR_BITPACK;
  if (b->start_pos + bits / 8 + b->gap_total + n_bytes > b->out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;  // bytewriter.rs:72-74
  if (n_bytes) {
    b->gaps.emplace_back(bits / 8, n_bytes);
    b->gap_total += n_bytes;
  }
  return X3_OK;
}
// BitPacker::word_align (:124-132): zero bits to the byte boundary, then zero bytes until the ABSOLUTE writer
// position is even
extern "C" int x3_bitpacker_word_align(x3_bitpacker* b) {
  if (!b) return X3_ERR_BAD_ARG;
  const uint64_t bits = bitpacker_bits(b);
  const uint32_t to_byte = (uint32_t)((8 - (bits & 7)) & 7);
  x3_bitpacker_write_packed_zeros(b, to_byte);
  if ((b->start_pos + (bits + to_byte) / 8 + b->gap_total) & 1) x3_bitpacker_write_packed_zeros(b, 8);
  return X3_OK;
}
// pack every field recorded since new() into the context's scratch (c->out) and take the CRC-16 (init 0xFFFF) of its
// first nbytes bytes
static int bitpacker_pack(const x3_bitpacker* b, uint64_t nbytes, uint16_t* crc) {
  x3_ctx* c = b->c;
  const uint32_t n = (uint32_t)b->val.size();
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t dst_bytes = (((bitpacker_bits(b) + 7) / 8 + 3) & ~(size_t)3) + 8;
  if ((rc = ensure(c, c->out, dst_bytes + 16))) return rc;
  if ((rc = ensure(c, c->in, (size_t)n * 8 + 16))) return rc;
  HIPCHK(c, hipMemsetAsync(c->out.p, 0, dst_bytes, c->stream));
  if (n) {
    HIPCHK(c, hipMemcpyAsync(c->in.p, b->val.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((uint8_t*)c->in.p + (size_t)n * 4, b->nb.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(x3_pack_fields_kernel, dim3(1), dim3(256), 0, c->stream, (const uint32_t*)c->in.p,
                       (const uint32_t*)((uint8_t*)c->in.p + (size_t)n * 4), n, (uint32_t*)c->out.p);
    HIPCHK(c, hipGetLastError());
  }
  uint16_t v = 0xFFFF;
  if ((rc = x3_crc16_dev(c, (const uint8_t*)c->out.p, nbytes, &v))) return rc;  // syncs
  if (crc) *crc = v;
  return X3_OK;
}

// len() (:88-90) and crc() (:75-77) as the reference reports them between writes: the count of COMPLETE bytes so far
// and the CRC-16 of those bytes.  Nothing is written to `out`.
extern "C" int x3_bitpacker_peek(const x3_bitpacker* b, uint64_t* len, uint16_t* crc) {
  if (!b) return X3_ERR_BAD_ARG;
  const uint64_t nbytes = bitpacker_bits(b) / 8;
  if (len) *len = nbytes;
  return crc ? bitpacker_pack(b, nbytes, crc) : X3_OK;
}

// flush (:79-86), what Drop does: a trailing partial byte is zero-padded and counted; every field recorded since new()
// is packed on the GPU, the bytes not handed over yet are written behind the writer's position.  *len and *crc are the
// reference's len() / crc() at that point (cumulative since new()), *out_pos the writer's position.  Writing may go on
// afterwards, from the next byte.
static int bitpacker_flush(x3_bitpacker* b, uint8_t* dst, uint64_t dst_cap, uint64_t* len, uint16_t* crc, uint64_t* n_new) {
  x3_ctx* c = b->c;
  const uint64_t bits = bitpacker_bits(b);
  const uint64_t nbytes = (bits + 7) / 8, fresh = nbytes - b->written;
  if (len) *len = nbytes;
  if (n_new) *n_new = fresh;
  if (fresh > dst_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  x3_bitpacker_write_packed_zeros(b, (uint32_t)((8 - (bits & 7)) & 7));
  int rc;
  if ((rc = bitpacker_pack(b, nbytes, crc))) return rc;
  if (fresh) {
    HIPCHK(c, hipMemcpyAsync(dst, (const uint8_t*)c->out.p + b->written, fresh, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b->written = nbytes;
  }
  return X3_OK;
}
extern "C" int x3_bitpacker_finish(x3_bitpacker* b, uint64_t* len, uint16_t* crc, uint64_t* out_pos) {
  if (!b || !b->out) return X3_ERR_BAD_ARG;
  uint64_t total = 0;
  if (b->gaps.empty()) {
    const uint64_t at = b->start_pos + b->written;
    const int rc = bitpacker_flush(b, b->out + at, b->out_cap - at, &total, crc, nullptr);
    if (len) *len = total;
    if (out_pos) *out_pos = b->start_pos + total;
    return rc;
  }
  // the writer has skipped bytes (inc_counter_n_bytes): the packed bytes land in pieces around the gaps
  x3_ctx* c = b->c;
  const uint64_t bits = bitpacker_bits(b);
  const uint64_t nbytes = (bits + 7) / 8;
  if (len) *len = nbytes;
  if (out_pos) *out_pos = b->start_pos + nbytes + b->gap_total;
  if (b->start_pos + nbytes + b->gap_total > b->out_cap) return X3_ERR_BYTE_WRITER_INSUFFICIENT_MEMORY;
  x3_bitpacker_write_packed_zeros(b, (uint32_t)((8 - (bits & 7)) & 7));
  int rc;
  if ((rc = bitpacker_pack(b, nbytes, crc))) return rc;
  uint64_t i = b->written, shift = 0;
  size_t g = 0;
  while (g < b->gaps.size() && b->gaps[g].first <= i) shift += b->gaps[g++].second;  // (gaps at or in front of the first new byte)
  while (i < nbytes) {
    const uint64_t end = g < b->gaps.size() ? std::min<uint64_t>(nbytes, b->gaps[g].first) : nbytes;
    if (end > i)
      HIPCHK(c, hipMemcpyAsync(b->out + b->start_pos + i + shift, (const uint8_t*)c->out.p + i, end - i, hipMemcpyDeviceToHost, c->stream));
    i = end;
    while (g < b->gaps.size() && b->gaps[g].first <= i) shift += b->gaps[g++].second;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  b->written = nbytes;
  return X3_OK;
}
// the same flush for a packer that is not bound to a slice (out == NULL at new(): any other ByteWriter): the bytes not
// handed over yet go to dst[0, *n_new) and the caller passes them to its writer.
extern "C" int x3_bitpacker_take(x3_bitpacker* b, uint8_t* dst, uint64_t dst_cap, uint64_t* n_new, uint64_t* len, uint16_t* crc) {
  if (!b || (!dst && dst_cap) || !b->gaps.empty()) return X3_ERR_BAD_ARG;
  return bitpacker_flush(b, dst, dst_cap, len, crc, n_new);
}
