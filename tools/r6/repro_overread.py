import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, x3hip, oracle_lib as O
d = np.load(os.path.join(ROOT, "tools/r6/data/fail_701_120377.npz"))
s = d["stream"]; want = d["want"]
ctx = x3hip.Context(0)
p = x3hip.Params.make(20, 3)
offs = [0, 32]
F = 2; ln = 96; n = 121
for fill in (0x00, 0xFF, 0x80, 0x55, 0x78, 0x01):
    buf = np.full(256, fill, dtype=np.uint8); buf[:ln] = s[:ln]
    d_x3 = ctx.alloc(256); d_off = ctx.alloc(8 * 3); d_wo = ctx.alloc(16); d_back = ctx.alloc(2 * 256); d_st = ctx.alloc(4 * 2)
    ctx.upload(d_x3, buf); ctx.upload(d_off, np.array(offs + [ln], dtype=np.uint64)); ctx.upload(d_wo, np.array([0, 60], dtype=np.uint64))
    ctx.set_option("wav_offsets_x4", 1)
    for blocks in (1, 0):
        ctx.set_option("decode_blocks", blocks)
        ctx.upload(d_back, np.zeros(256, dtype=np.int16))
        rc = ctx.decode_dev(d_x3, ln, d_off, F, p, d_back, 200, d_wav_offsets=d_wo, d_status=d_st)
        r = ctx.decode_result()
        got = ctx.download(d_back, 2 * n, np.int16)
        st = ctx.download(d_st, 8, np.int32)
        dd = np.nonzero(got != want[:n])[0]
        print("fill %02x blocks %d kernel %d rc %d result %s status %s diffs %s got[120]=%d want %d" % (fill, blocks, ctx.get_option("decode_kernel_in_use"), rc, r, st.tolist(), dd[:5].tolist(), got[120], want[120]))
