"""CPU-only checks of the product library's host logic: it loads, exports every symbol that
include/x3hip.h declares, and its 20-byte header / parameter / bound helpers agree with the oracle.
No kernel is launched here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np

import oracle_lib as O
import x3hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "x3hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(x3_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(x3hip.SYMBOLS), declared ^ set(x3hip.SYMBOLS)
    L = x3hip.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return
    try:
        x3hip.Context(0)
    except x3hip.X3Error as e:
        assert e.rc == x3hip.ERR_HIP
    else:
        raise AssertionError("context creation must fail without a HIP device")


def test_status_codes_match_oracle_numbering():
    hdr = open(os.path.join(ROOT, "include", "x3hip.h")).read()
    prod = dict((k, int(v)) for k, v in re.findall(r"X3_(?:ERR_)?([A-Z_0-9]+)\s*=\s*(\d+)", hdr))
    ohdr = open(os.path.join(ROOT, "oracle", "x3_oracle.h")).read()
    orac = dict((k, int(v)) for k, v in re.findall(r"X3O_([A-Z_0-9]+)\s*=\s*(\d+)", ohdr))
    assert prod == orac
    assert x3hip.strerror(14) == "FrameHeaderInvalidPayloadCRC" and x3hip.strerror(22) == "ByteWriterInsufficientMemory"


def test_params_default_and_validate():
    p = x3hip.Params.default()
    assert (p.block_len, p.blocks_per_frame, list(p.codes), list(p.thresholds)) == (20, 500, [0, 1, 3], [3, 8, 20])
    L, OL = x3hip.lib(), O.lib()
    rng = np.random.default_rng(0)
    for _ in range(300):
        codes = tuple(int(v) for v in rng.integers(0, 5, size=3))
        thr = tuple(int(v) for v in rng.integers(0, 32, size=3))
        pp = x3hip.Params.make(20, 500, codes, thr)
        po = O.Params.make(20, 500, codes, thr)
        assert L.x3_params_validate(C.byref(pp)) == OL.x3o_params_new(C.byref(po)), (codes, thr)


def test_frame_header_helpers_match_oracle():
    OL = O.lib()
    rng = np.random.default_rng(1)
    for _ in range(200):
        ns, ident, plen, pcrc = (int(rng.integers(0, 70000)), int(rng.integers(0, 256)), int(rng.integers(0, 70000)),
                                 int(rng.integers(0, 65536)))
        h = x3hip.write_frame_header(ns, ident, plen, pcrc)
        ho = np.zeros(20, dtype=np.uint8)
        OL.x3o_write_frame_header(ns, ident, plen, pcrc, ho.ctypes.data)
        assert np.array_equal(h, ho)
        for tamper in (None, 0, 3, 5, 6, 16, 19):
            b = h.copy()
            if tamper is not None:
                b[tamper] ^= int(rng.integers(1, 256))
                if rng.integers(0, 2):  # keep the header CRC valid so that the later checks are reached
                    c = O.crc16(b[:16]); b[16] = c >> 8; b[17] = c & 0xFF
            rc, fh = x3hip.read_frame_header(b)
            fo = O.FrameHeader()
            rco = OL.x3o_read_frame_header(b.ctypes.data, 20, C.byref(fo))
            assert rc == rco
            if rc == 0:
                assert (fh.source_id, fh.channels, fh.samples, fh.payload_len, fh.payload_crc) == \
                       (fo.source_id, fo.channels, fo.samples, fo.payload_len, fo.payload_crc)
    assert x3hip.read_frame_header(np.zeros(19, dtype=np.uint8))[0] == x3hip.ERR_FRAME_DECODE_UNEXPECTED_END


def test_crc16_update_matches_oracle():
    OL = O.lib()
    L = x3hip.lib()
    rng = np.random.default_rng(2)
    for _ in range(2000):
        c, b = int(rng.integers(0, 65536)), int(rng.integers(0, 256))
        assert L.x3_crc16_update(c, b) == OL.x3o_update_crc16(c, b)


def test_encode_bound_covers_oracle_worst_case():
    L = x3hip.lib()
    for bl, bpf in [(20, 500), (1, 10), (60, 100), (7, 33)]:
        p = x3hip.Params.make(bl, bpf)
        po = O.Params.make(bl, bpf)
        for n in [1, 2, bl * bpf - 1, bl * bpf, bl * bpf + 1, 3 * bl * bpf + 5]:
            wav = x3hip.synth(x3hip.SYNTH_WHITE, 9, 0, n)  # all-literal = worst case
            rc, out, _ = O.encode(wav, po)
            assert rc == 0
            bound = L.x3_encode_bound(n, C.byref(p))
            assert out.size <= bound, (bl, bpf, n, out.size, bound)
            if bl >= 20 and n > 1000:  # white noise makes (nearly) every long block literal: the bound is tight
                assert bound <= out.size * 1.01 + 1
            assert L.x3_num_frames(n, C.byref(p)) == (n + bl * bpf - 1) // (bl * bpf)


def test_synth_is_position_independent_and_seeded():
    for kind in range(5):
        a = x3hip.synth(kind, 42, 0, 20000)
        assert np.array_equal(a[4000:9000], x3hip.synth(kind, 42, 4000, 5000))
        if kind in (1, 2, 4):
            assert not np.array_equal(a, x3hip.synth(kind, 43, 0, 20000))


def test_oracle_roundtrip_on_synthetic_kinds():
    """multi-frame concatenation is unpinned by reference tests: at least it must round-trip"""
    for kind in range(5):
        wav = x3hip.synth(kind, 7, 0, 45678)
        rc, stream, stats = O.encode(wav)
        assert rc == 0 and int(stats.sum()) == wav.size - 5
        rc, back, fok, ferr = O.decode_stream(stream, wav_cap=wav.size)
        assert (rc, fok, ferr) == (0, 5, 0) and np.array_equal(back, wav)


def test_shard_arithmetic():
    """x3_shard_frame_range / sample_range / offsets (C ABI, host arithmetic): contiguous ranges of whole frames that
    cover everything once, the remainder one frame each to the first ranks, offsets = exclusive scan of the lengths"""
    p = x3hip.Params.default()
    for F in [0, 1, 2, 7, 8, 9, 69120, 552960]:
        for world in [1, 2, 3, 4, 8]:
            base, rem = divmod(F, world)
            got = [x3hip.shard_frame_range(F, r, world) for r in range(world)]
            assert got == [(r * base + min(r, rem), base + (1 if r < rem else 0)) for r in range(world)]
            assert got[0][0] == 0 and got[-1][0] + got[-1][1] == F
    for n in [0, 1, 5, 9999, 10000, 10001, 123457, 691_200_000, 5_529_600_000]:
        for world in [1, 2, 3, 8]:
            F = (n + 9999) // 10000
            got = [x3hip.shard_sample_range(n, p, r, world) for r in range(world)]
            for r, (lo, cnt) in enumerate(got):
                f_lo, f_n = x3hip.shard_frame_range(F, r, world)
                # (a rank behind the last frame starts at the end of the samples: the ranges tile [0, n) for any arguments)
                assert lo == min(n, f_lo * 10000) and cnt == max(0, min(n, (f_lo + f_n) * 10000) - lo)
            assert sum(c for _, c in got) == n
            assert got[0][0] == 0 and all(got[r][0] + got[r][1] == got[r + 1][0] for r in range(world - 1))
    assert x3hip.shard_offsets([10, 0, 22, 4]) == [0, 10, 10, 32, 36]


def test_rice_code_tables_match_reference_literals():
    """x3_rice_code_get (RiceCodes::get, src/x3.rs:206-260) against the reference's literal tables"""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rice_tables.json")))
    L = x3hip.lib()
    for k, t in enumerate(g["tables"]):
        rc = x3hip.RiceCode()
        assert L.x3_rice_code_get(k, C.byref(rc)) == 0
        assert (rc.nsubs, rc.offset, rc.inv_len, rc.len) == (t["nsubs"], t["offset"], t["inv_len"], len(t["code"]))
        assert [rc.code[i] for i in range(rc.len)] == t["code"]
        assert [rc.num_bits[i] for i in range(rc.len)] == t["num_bits"]
        assert [rc.inv[i] for i in range(60)] == g["inv"]
    assert L.x3_rice_code_get(4, C.byref(x3hip.RiceCode())) == 24


def test_rust_ffi_block_matches_the_header():
    """The Rust mirror (x3-rust_amd/rust/src/lib.rs) cannot be compiled in this image: its `extern "C"` block is held
    against include/x3hip.h here instead -- every function it declares exists in the header with the same number of
    arguments, and pointer arguments stand where the header has pointers (the signatures could drift unseen)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "x3hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    cproto = {}
    for m in re.finditer(r"\b(x3_\w+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        if args == ["void"] or args == [""]:
            args = []
        cproto[m.group(1)] = args
    rs = open(os.path.join(ROOT, "x3-rust_amd", "rust", "src", "lib.rs")).read()
    blocks = re.findall(r'extern\s+"C"\s*\{(.*?)\n    \}', rs, flags=re.S)
    assert blocks, "no extern \"C\" block found"
    seen = 0
    for blk in blocks:
        blk = re.sub(r"//[^\n]*", " ", blk)
        for m in re.finditer(r"pub\s+fn\s+(x3_\w+)\s*\(([^)]*)\)", blk, flags=re.S):
            name = m.group(1)
            rargs = [a.strip() for a in m.group(2).replace("\n", " ").split(",") if a.strip()]
            assert name in cproto, "%s is declared in lib.rs but not in include/x3hip.h" % name
            cargs = cproto[name]
            assert len(rargs) == len(cargs), "%s: %d arguments in lib.rs, %d in x3hip.h" % (name, len(rargs), len(cargs))
            for i, (ra, ca) in enumerate(zip(rargs, cargs)):
                r_ptr = "*const" in ra or "*mut" in ra
                c_ptr = "*" in ca or "[" in ca
                assert r_ptr == c_ptr, "%s, argument %d: `%s` in lib.rs against `%s` in x3hip.h" % (name, i, ra, ca)
                # ... and the TYPES: every C scalar / pointee type against its Rust spelling, const-ness of pointers included
                # (a widened or narrowed argument must fail here, not on the first call of a compiled crate)
                assert _rust_type(ra.split(":", 1)[1]) == _c_type_as_rust(ca), \
                    "%s, argument %d: `%s` in lib.rs against `%s` in x3hip.h" % (name, i, ra, ca)
            seen += 1
        # return types
        for m in re.finditer(r"pub\s+fn\s+(x3_\w+)\s*\([^)]*\)\s*(->\s*([^;]+))?;", blk, flags=re.S):
            cm = re.search(r"([\w\s\*]+?)\b%s\s*\(" % m.group(1), hdr)
            assert cm, m.group(1)
            c_ret = cm.group(1).strip()
            r_ret = _rust_type(m.group(3)) if m.group(3) else "()"
            assert r_ret == ("()" if c_ret == "void" else _c_type_as_rust(c_ret + " x")), (m.group(1), c_ret, m.group(3))
    assert seen >= 30, seen
    # the #[repr(C)] structs: field order, names and types against the header's
    for st in ("x3_params", "x3_frame_header", "x3_batch", "x3_rice_code"):
        cm = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (st, st), hdr, flags=re.S)
        assert cm, st
        cfields = []
        for decl in cm.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            base, names = re.match(r"((?:const\s+)?\w+\s*\**)\s*(.*)", decl, flags=re.S).groups()
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                ctype = _c_type_as_rust(base + " x")
                cfields.append((arr.group(1), "[%s; %s]" % (ctype, arr.group(2))) if arr else (nm.lstrip("* "), ctype))
        rm = re.search(r"#\[repr\(C\)\][^{]*?pub\s+struct\s+%s\s*\{(.*?)\}" % st, rs, flags=re.S)
        assert rm, "%s: no #[repr(C)] struct in lib.rs" % st
        rfields = [(f.split(":")[0].replace("pub", "").strip(), _rust_type(f.split(":", 1)[1]))
                   for f in re.sub(r"//[^\n]*", " ", rm.group(1)).split(",") if ":" in f]
        assert rfields == cfields, (st, rfields, cfields)


_C2RUST = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16",
           "int32_t": "i32", "int64_t": "i64", "int": "c_int", "unsigned": "c_uint", "char": "c_char", "void": "c_void",
           "long long": "c_longlong", "unsigned long long": "c_ulonglong", "size_t": "usize", "double": "f64", "float": "f32"}


def _c_type_as_rust(decl):
    """`const uint8_t* data`, `uint64_t stats[6]`, `const int16_t* const* wavs`, `x3_ctx** ctx` -> the Rust FFI spelling"""
    import re
    d = decl.strip()
    is_arr = "[" in d
    d = re.sub(r"\[[^\]]*\]", "", d)
    parts = d.split("*")
    last = parts[-1].split()
    # the parameter's name: the last word of the last part, unless that part is the type itself (`int`, `long long`)
    if len(parts) > 1:
        parts[-1] = " ".join(t for t in last if t == "const")
    else:
        toks = [t for t in last if t not in ("const", "struct")]
        if len(toks) > 1 and " ".join(toks) not in _C2RUST:
            last = last[:-1]
        parts[-1] = " ".join(last)
    base_toks = parts[0].split()
    pointee_const = ["const" in base_toks] + ["const" in p_.split() for p_ in parts[1:-1]]
    base = " ".join(t for t in base_toks if t not in ("const", "struct"))
    t = _C2RUST.get(base, base)
    levels = len(parts) - 1
    if is_arr:
        levels += 1
        pointee_const = pointee_const + [False] if levels > len(pointee_const) else pointee_const
    for k in range(levels):
        t = ("*const " if pointee_const[k] else "*mut ") + t
    return t


def _rust_type(t):
    import re
    return re.sub(r"\s+", " ", t.strip().rstrip(",")).strip()


def test_rust_ffi_drift_test_catches_a_widened_argument():
    """the type mapping above is not vacuous: a C `uint32_t n_channels` against a Rust `u64` differs, a `const uint8_t*` against
    `*mut u8` too"""
    assert _c_type_as_rust("uint32_t n_channels") == "u32" != _rust_type("u64")
    assert _c_type_as_rust("const uint8_t* data") == "*const u8" != "*mut u8"
    assert _c_type_as_rust("uint64_t stats[6]") == "*mut u64"
    assert _c_type_as_rust("x3_ctx** ctx") == "*mut *mut x3_ctx"
    assert _c_type_as_rust("const x3_params* p") == "*const x3_params"
    assert _c_type_as_rust("long long value") == "c_longlong"
    assert _c_type_as_rust("int") == "c_int" and _c_type_as_rust("void* hip_stream") == "*mut c_void"


def test_decoder_ring_requests_are_not_waited_for_right_behind_their_issue():
    """Round 5: with code next to the ring service changed, the register allocator copied part of a request's destination
    registers directly behind the request -- a wait for a load that had just been issued, in every service of every group,
    0.69 -> 0.75 ms, with the results unchanged and every test green.  tools/check_decoder_isa.py compiles the decoder's
    translation unit to assembly (device only, ~6 s) and looks for that pattern in x3_decode_split_kernel."""
    import importlib.util
    import shutil
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not found")
    spec = importlib.util.spec_from_file_location("check_decoder_isa", os.path.join(ROOT, "tools", "check_decoder_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.findings([]) == []
    # ... and the budgets the decode phase's occupancy rests on: five decoder groups per CU by LDS, four waves per SIMD by
    # registers, three check-kernel waves beside them, no scratch (a round-5 variant of the check kernel took 151 registers
    # with nothing failing)
    res = m.resources([])
    assert set(res) == set(m.BUDGET) and m.over_budget([]) == [], res


def test_encoder_instantiations_keep_their_register_budgets():
    """Round 6 added block lengths 10 and 40 as instantiations of the two single-pass encoders: every one of them must fit the
    registers its occupancy rests on -- 128 for the wave encoder (four waves per SIMD), 80 for the second generation (six) --
    and use no scratch (tools/check_encoder_isa.py; compiles the encoder's translation unit to assembly, ~10 s)."""
    import importlib.util
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not found")
    spec = importlib.util.spec_from_file_location("check_encoder_isa", os.path.join(ROOT, "tools", "check_encoder_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    res = m.resources()
    assert set(res) == set(m.BUDGET), sorted(set(m.BUDGET) - set(res))
    assert m.over_budget(res) == [], res
