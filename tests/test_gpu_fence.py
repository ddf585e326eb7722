"""Guard pages around every device buffer (x3-rust_amd/csrc/x3_fence.h, X3HIP_FENCE): the GPU sanitizer is not available on
the test pool, so this is how a kernel that reads or writes behind a buffer is caught -- at once, not once in a soak when a
buffer happens to end where its mapping does (round 5: the decoders read 128 bytes behind a stream whose last frame is a
single sample ending on a 16-byte boundary; found by the soak as a memory fault, pinned by this fence, fixed).

The fence is chosen when the library is loaded, so each test is a CHILD process with X3HIP_FENCE=16 (buffers end, rounded up
to 16 bytes, at the last byte of their mapping) and X3HIP_FENCE_FILL=165 (fresh buffers are not zero).  A fault kills the
child; the parent sees the signal."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _child(code, timeout=600):
    env = dict(os.environ, X3HIP_FENCE="16", X3HIP_FENCE_FILL="165")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "x3-rust_amd"), HERE, os.path.join(ROOT, "tools"), env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, "child under the fence ended with %d:\n%s" % (r.returncode, tail)
    return r.stdout


def test_last_frame_of_one_sample_ending_on_a_chunk_boundary():
    """streams of k frames + 1 sample whose length is a multiple of 16, in a buffer of exactly that length: the frame-per-lane
    decoders (three-wave with and without a recorded index, the single-wave kernel through per-frame offsets) and the stream entry"""
    out = _child("""
        import ctypes as C
        import numpy as np
        import x3hip, oracle_lib as O
        ctx = x3hip.Context(0)
        L = x3hip.lib()
        done = 0
        for bpf in (100, 500, 7):
            p = x3hip.Params.make(20, bpf)
            po = O.Params.make(20, bpf)
            spf = 20 * bpf
            for seed in range(200):
                n = spf * (1 + seed % 3) + 1
                wav = x3hip.synth(2, 7000 + seed, 0, n)
                rc, ref, _ = O.encode(wav, po)
                assert rc == 0
                if ref.size % 16:
                    continue
                offs, pos = [], 0
                while pos < ref.size:
                    offs.append(pos); pos += 20 + (int(ref[pos + 6]) << 8 | int(ref[pos + 7]))
                F = len(offs)
                assert ref.size - offs[-1] == 22          # the last frame: a header and one raw sample
                d_x3 = ctx.alloc(ref.size); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n)
                d_wo = ctx.alloc(8 * F)
                ctx.upload(d_x3, ref); ctx.upload(d_off, np.array(offs + [ref.size], dtype=np.uint64))
                ctx.upload(d_wo, np.arange(F, dtype=np.uint64) * spf)
                # the three-wave decoder
                ctx.upload(d_back, np.zeros(n, dtype=np.int16))
                assert ctx.decode_dev(d_x3, ref.size, d_off, F, p, d_back, n, n_per_clip=n) == 0
                assert ctx.decode_result() == (0, F, 0, n)
                assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
                # ... recording a segment index, and decoding by it
                ne = L.x3_seg_index_entries(F, C.byref(p), 4)
                if ne:
                    d_seg = ctx.alloc(8 * ne)
                    ctx.upload(d_back, np.zeros(n, dtype=np.int16))
                    assert ctx.decode_dev_seg(d_x3, ref.size, d_off, F, p, d_back, n, d_seg, 4, record=True, n_per_clip=n) == 0
                    assert ctx.decode_result() == (0, F, 0, n)
                    for want in (0, 2, 25):
                        ctx.set_option("seg_stretches", want)
                        ctx.upload(d_back, np.zeros(n, dtype=np.int16))
                        assert ctx.decode_dev_seg(d_x3, ref.size, d_off, F, p, d_back, n, d_seg, 4, n_per_clip=n) == 0
                        assert ctx.decode_result() == (0, F, 0, n)
                        assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
                    ctx.set_option("seg_stretches", 0)
                    ctx.free(d_seg)
                # per-frame sample offsets that are not promised to be multiples of four: the single-wave decoder (x3_decode_fast_kernel)
                ctx.upload(d_back, np.zeros(n, dtype=np.int16))
                assert ctx.decode_dev(d_x3, ref.size, d_off, F, p, d_back, n, d_wav_offsets=d_wo) == 0
                assert ctx.decode_result()[:3] == (0, F, 0)
                assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
                # the stream entry (frame walk on the GPU and on the host)
                for hw in (0, 1):
                    ctx.set_option("host_walk", hw)
                    r = ctx.decode_stream(ref, p, wav_cap=n)
                    assert r[0] == 0 and np.array_equal(r[1], wav)
                ctx.set_option("host_walk", -1)
                for d in (d_x3, d_off, d_back, d_wo):
                    ctx.free(d)
                done += 1
                if done % 3 == 0:
                    break
        ctx.close()
        print("streams", done)
        """)
    assert int(out.split("streams")[1]) >= 3


def test_fuzz_families_under_the_fence():
    """a few hundred trials of every device-buffer family of tools/fuzz_parity.py (e g d b a m s) with buffers that end at
    an unmapped page and start out non-zero"""
    out = _child("""
        import fuzz_parity as FZ
        c = FZ.run(seed=9, trials=1400, families="egdbams")
        print("trials", sum(c.values()))
        """, timeout=900)
    assert int(out.split("trials")[1]) == 1400


def test_segment_index_of_frames_longer_than_the_parameters_say():
    """a frame whose HEADER holds more blocks than the parameters' blocks_per_frame (a stream encoded with 500
    blocks a frame, decoded with parameters that say 100) has no room in the index for its later stretches: recording must
    not write into the next frames' rows or behind the index (guard pages: the index buffer ends at its mapping), and a
    decode by that index must still give the stream's samples -- the last stretch decodes to the frame's end"""
    _child("""
        import ctypes as C
        import numpy as np
        import x3hip, oracle_lib as O
        ctx = x3hip.Context(0)
        ctx.set_option("wav_offsets_x4", 1)
        L = x3hip.lib()
        p_enc = x3hip.Params.make(20, 500)
        for bpf_dec, sb in ((100, 4), (48, 8), (499, 4), (20, 4)):
            p = x3hip.Params.make(20, bpf_dec)
            n = 10000 * 37 + 4321
            wav = x3hip.synth(2, 4400 + bpf_dec, 0, n)
            rc, ref, _ = O.encode(wav, O.Params.make(20, 500))
            assert rc == 0
            offs, pos = [], 0
            while pos < ref.size:
                offs.append(pos); pos += 20 + (int(ref[pos + 6]) << 8 | int(ref[pos + 7]))
            F = len(offs)
            d_x3 = ctx.alloc(ref.size); d_off = ctx.alloc(8 * (F + 1)); d_back = ctx.alloc(2 * n); d_wo = ctx.alloc(8 * F)
            ctx.upload(d_x3, ref); ctx.upload(d_off, np.array(offs + [ref.size], dtype=np.uint64))
            ctx.upload(d_wo, np.arange(F, dtype=np.uint64) * 10000)
            ne = L.x3_seg_index_entries(F, C.byref(p), sb)
            assert ne > 0
            d_seg = ctx.alloc(8 * ne)                     # exactly the entries the parameters ask for
            ctx.upload(d_back, np.zeros(n, dtype=np.int16))
            assert ctx.decode_dev_seg(d_x3, ref.size, d_off, F, p, d_back, n, d_seg, sb, record=True, d_wav_offsets=d_wo) == 0
            assert ctx.decode_result() == (0, F, 0, n)
            assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav)
            for want in (0, 2, 5, 25):
                ctx.set_option("seg_stretches", want)
                ctx.upload(d_back, np.zeros(n, dtype=np.int16))
                assert ctx.decode_dev_seg(d_x3, ref.size, d_off, F, p, d_back, n, d_seg, sb, d_wav_offsets=d_wo) == 0
                assert ctx.decode_result() == (0, F, 0, n), (bpf_dec, sb, want)
                assert np.array_equal(ctx.download(d_back, 2 * n, np.int16), wav), (bpf_dec, sb, want)
            ctx.set_option("seg_stretches", 0)
            for d in (d_x3, d_off, d_back, d_wo, d_seg):
                ctx.free(d)
        print("ok")
    """)
