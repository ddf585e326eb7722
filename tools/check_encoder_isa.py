#!/usr/bin/env python3
"""The register and scratch budgets the encoders' occupancy rests on, for every instantiation (block lengths 10 / 20 / 40,
uniform layout / frame table, whole call / dense pass).  The wave encoder runs sixteen waves per CU = four per SIMD: 128
registers and NO scratch (a reload from scratch is a VMEM load, and the kernel's waits are counted by hand); the
second-generation kernel three workgroups of eight waves = six per SIMD: 80 registers, no scratch.
Compiles x3_encode.hip to assembly (device only, ~10 s).
   python tools/check_encoder_isa.py        -> prints what it found, exit 1 if something is over budget"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET = {}
for tab in ("0", "1"):
    for bl in ("10", "20", "40"):
        BUDGET["_Z21x3_encode_wave_kernelILb%sELj%sEE" % (tab, bl)] = 128
        for lst in ("0", "1"):
            BUDGET["_Z24x3_encode_stream2_kernelILb%sELb%sELj%sEE" % (lst, tab, bl)] = 80
def resources(flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-Wno-unused-function", "--cuda-device-only", "-S", "-o", out] + list(flags) +
                       [os.path.join(ROOT, "x3-rust_amd", "csrc", "x3_encode.hip")], check=True, capture_output=True)
        lines = open(out).read().split("\n")
    res, cur = {}, None
    for l in lines:
        t = l.strip()
        if t.startswith(".amdhsa_kernel"):
            cur = next((k for k in BUDGET if t.split()[1].startswith(k)), None)
            if cur: res[cur] = {}
        elif t.startswith(".end_amdhsa_kernel"):
            cur = None
        elif cur:
            for key, name in ((".amdhsa_private_segment_fixed_size", "scratch"), (".amdhsa_next_free_vgpr", "vgpr")):
                if t.startswith(key + " "):
                    res[cur][name] = int(t.split()[1])
    return res
def over_budget(res):
    return [(k, r, BUDGET[k]) for k, r in res.items() if r.get("vgpr", 0) > BUDGET[k] or r.get("scratch", 0) != 0]
if __name__ == "__main__":
    r = resources(sys.argv[1:])
    for k in sorted(r):
        print(k, r[k])
    missing = sorted(set(BUDGET) - set(r))
    bad = over_budget(r)
    print("missing:", missing, "over budget:", bad)
    sys.exit(1 if missing or bad else 0)
