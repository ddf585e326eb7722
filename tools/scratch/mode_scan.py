"""which buffer's placement selects the decoder's timing mode: re-allocate one buffer at a time"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "x3-rust_amd"))
import x3hip
ctx = x3hip.Context(0); p = x3hip.Params.default(); n = 691_200_000; L = x3hip.lib()
F = L.x3_num_frames(n, C.byref(p)); cap = L.x3_encode_bound(n, C.byref(p))
bufs = {"wav": ctx.alloc(2 * n), "out": ctx.alloc(cap + 16), "off": ctx.alloc(8 * (F + 1)), "back": ctx.alloc(2 * n)}
sizes = {"wav": 2 * n, "out": cap + 16, "off": 8 * (F + 1), "back": 2 * n}
ctx.synth_dev(2, 0x58330003, 0, n, bufs["wav"])
def run(steps=int(os.environ.get("MODE_STEPS", "20"))):
    ctx.enable_kernel_timing(False)
    for _ in range(8):
        assert ctx.encode_dev(bufs["wav"], n, p, bufs["out"], cap, 0, bufs["off"]) == 0
        assert ctx.decode_dev(bufs["out"], cap, bufs["off"], F, p, bufs["back"], n, n_per_clip=n) == 0
    ctx.encode_result(); ctx.decode_result()
    ctx.enable_kernel_timing(True); ctx.reset_kernel_time()
    for _ in range(steps):
        assert ctx.encode_dev(bufs["wav"], n, p, bufs["out"], cap, 0, bufs["off"]) == 0
        assert ctx.decode_dev(bufs["out"], cap, bufs["off"], F, p, bufs["back"], n, n_per_clip=n) == 0
    ctx.encode_result(); r = ctx.decode_result(); assert r[:3] == (0, F, 0)
    return [ctx.kernel_time(i)[0] / steps for i in (0, 1, 4)]
print("start:", " ".join("%s=%x" % kv for kv in bufs.items()), "enc %.3f dec %.3f chk %.3f" % tuple(run()))
keep = []
for which in ("back", "out", "wav", "back", "out", "wav"):
    for rep in range(5):
        keep.append(ctx.alloc((rep + 1) * 3 * 1024 * 1024))   # perturb the allocator
        new = ctx.alloc(sizes[which]); ctx.free(bufs[which]); bufs[which] = new
        if which == "wav": ctx.synth_dev(2, 0x58330003, 0, n, bufs["wav"])
        e, d, c = run()
        print("realloc %-4s -> %x: enc %.3f dec %.3f chk %.3f" % (which, new, e, d, c), flush=True)
