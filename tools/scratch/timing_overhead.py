#!/usr/bin/env python3
"""What the per-kernel HIP events of bench.py's timed region cost a step: config 3, 200 steps back to back, kernel timing
on (events on the dispatch packets, x3_ctx_enable_kernel_timing) against off."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "x3-rust_amd"))
import torch
import x3hip

n = 691_200_000
dev = torch.device("cuda:0")
ctx = x3hip.Context(0)
p = x3hip.Params.default()
L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
wav = torch.empty(n, dtype=torch.int16, device=dev)
ctx.synth_dev(2, 0x58330003, 0, n, wav.data_ptr())
out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
off = torch.empty(F + 1, dtype=torch.int64, device=dev)
back = torch.empty(n, dtype=torch.int16, device=dev)

def steps(k):
    for _ in range(k):
        assert ctx.encode_dev(wav.data_ptr(), n, p, out.data_ptr(), cap, 0, off.data_ptr()) == 0
        assert ctx.decode_dev(out.data_ptr(), cap, off.data_ptr(), F, p, back.data_ptr(), n, n_per_clip=n) == 0
    ctx.sync()

steps(60)
for rep in range(3):
    for on in (False, True):
        ctx.enable_kernel_timing(on); ctx.reset_kernel_time()
        steps(10)
        t0 = time.perf_counter(); steps(200); dt = (time.perf_counter() - t0) / 200
        print("rep %d timing %-3s %.4f ms per step" % (rep, "on" if on else "off", dt * 1e3), flush=True)
        ctx.enable_kernel_timing(False)
assert ctx.encode_result()[0] == 0 and ctx.decode_result()[:3] == (0, F, 0)
