"""x3hip.place_buffers (round 6; profiles/r6/decoder_modes.txt): the probe that times the round trip on every pair of
candidate stream / sample buffers.  Here only that it does what it says -- every pair timed, the buffers usable afterwards,
the last pair's decode the identity -- on a small input; what it buys is measured by tools/r6/bench_placement_stats.sh."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_every_pair_is_timed_and_the_buffers_hold_a_round_trip():
    import x3hip
    ctx = x3hip.Context(0)
    try:
        p = x3hip.Params.default()
        n = 2_000_000
        L = x3hip.lib()
        F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
        wav = x3hip.synth(2, 99, 0, n)
        d_wav = ctx.alloc(2 * n); d_off = ctx.alloc(8 * (F + 1))
        outs = [ctx.alloc(cap + 16) for _ in range(3)]
        backs = [ctx.alloc(2 * n) for _ in range(2)]
        ctx.upload(d_wav, wav)
        ms = x3hip.place_buffers(ctx, p, d_wav, n, outs, cap, d_off, backs, warm=1, steps=2)
        assert len(ms) == 3 and all(len(r) == 2 and all(0 < v < 1000 for v in r) for r in ms), ms
        # the last pair probed holds the last round trip
        assert np.array_equal(ctx.download(backs[-1], 2 * n, np.int16), wav)
        # ... and the context is where a caller expects it: nothing pending, the next call goes through
        assert ctx.encode_dev(d_wav, n, p, outs[0], cap, 0, d_off) == 0
        rc, pos, st = ctx.encode_result()
        assert rc == 0 and pos > 0 and int(st.sum()) == n - F
        # a stream that does not decode is an error, not a number
        with pytest.raises(x3hip.X3Error):
            x3hip.place_buffers(ctx, p, d_wav, n, outs[:1], 64, d_off, backs[:1], warm=1, steps=1)
    finally:
        ctx.close()


def test_kernel_timing_mask_chooses_the_kernels_that_carry_events():
    """option kernel_timing_mask (round 6; bench.py's timed region carries the decode phase's events only): bit k = kernel id k"""
    import x3hip
    ctx = x3hip.Context(0)
    try:
        p = x3hip.Params.default()
        n = 1_000_000
        L = x3hip.lib()
        F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
        d_wav = ctx.alloc(2 * n); d_off = ctx.alloc(8 * (F + 1)); d_out = ctx.alloc(cap + 16); d_back = ctx.alloc(2 * n)
        ctx.upload(d_wav, x3hip.synth(2, 5, 0, n))
        for mask, want in (((1 << 1) | (1 << 4), {0: 0, 1: 3, 4: 3}), (0xFFFFFFFF, {0: 3, 1: 3, 4: 3}), (1 << 0, {0: 3, 1: 0, 4: 0})):
            ctx.set_option("kernel_timing_mask", mask)
            ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
            for _ in range(3):
                assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
                assert ctx.decode_dev(d_out, cap, d_off, F, p, d_back, n, n_per_clip=n) == 0
            assert ctx.encode_result()[0] == 0 and ctx.decode_result()[:3] == (0, F, 0)
            for which, cnt in want.items():
                ms, got = ctx.kernel_time(which)
                assert got == cnt and (ms > 0) == (cnt > 0), (hex(mask), which, ms, got)
            ctx.enable_kernel_timing(False)
        ctx.set_option("kernel_timing_mask", 0xFFFFFFFF)
    finally:
        ctx.close()
