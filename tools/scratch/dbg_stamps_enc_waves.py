"""Per-wave phase times of x3_encode_stream_kernel (build with -DX3_DBG_STAMPS -DX3_DBG_ALLWAVES, X3HIP_LIB=...):
which wave of a workgroup is late at which barrier?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import numpy as np, x3hip
x3hip.LIB_PATH = os.environ["X3HIP_LIB"]
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
d_wav = ctx.alloc(2*n); d_out = ctx.alloc(cap+16); d_off = ctx.alloc(8*(F+1))
ctx.synth_dev(2, 0x58330003, 0, n, d_wav)
for _ in range(2):
    assert ctx.encode_dev(d_wav, n, p, d_out, cap, 0, d_off) == 0
    ctx.encode_result()
out = np.zeros(8*9*128, dtype=np.uint64)
L.x3_dbg_read_enc.argtypes=[C.c_void_p, C.c_void_p, C.c_uint64]
L.x3_dbg_read_enc(ctx._h, out.ctypes.data, out.size)
a = out.reshape(128, 9, 8).astype(np.float64) / 135.0   # per frame
names = "analyze,scan,wait B1,wait B3,emit,crc,copy-out,wait B4+B4b".split(",")
print("per frame; waves 0..7 compute, 8 helper (its slots mean other things)")
A_all = a
for label, a in (("workgroups 0..63", A_all[:64]), ("workgroups 256..319", A_all[64:])):
  print(label)
  print("wave  " + "  ".join("%12s" % nm for k, nm in enumerate(names) if nm != "-") + "         total")
  for w in range(9):
      row = a[:, w, :].mean(axis=0)
      print("%4d  " % w + "  ".join("%12.0f" % row[k] for k, nm in enumerate(names) if nm != "-") + "  %12.0f" % row[:8].sum())
  busy = a[:, :8, [0, 1, 4, 5, 6]].sum(axis=2)
  print("compute waves: busy (non-barrier) ticks per frame, mean by wave:", busy.mean(axis=0).round(0))
  print("slowest wave's busy / mean busy per workgroup: %.3f" % (busy.max(axis=1) / busy.mean(axis=1)).mean())
