#!/usr/bin/env python3
"""Does the register allocator wait for a ring request right behind it?  (round 5: it did, when code next to the service
lambda changed -- part of a request's destination registers was copied directly behind the load: 0.69 -> 0.75 ms.)
Compiles x3_decode.hip to assembly (device only) and looks, in x3_decode_split_kernel, for an `s_waitcnt vmcnt(n)` within
four instructions behind a run of global_load_dwordx4 that leaves fewer loads in flight than the run issued.
Also: the LDS, register and scratch budgets the decode phase's occupancy rests on (BUDGET below).
   python tools/check_decoder_isa.py [-DFLAG ...]        -> prints the findings, exit 1 if any"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_asm_cache = {}
def asm_lines(flags):
    key = tuple(flags)
    if key not in _asm_cache:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "x.s")
            subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                            "-Wno-unused-function", "--cuda-device-only", "-S", "-o", out] + list(flags) +
                           [os.path.join(ROOT, "x3-rust_amd", "csrc", "x3_decode.hip")], check=True, capture_output=True)
            _asm_cache[key] = open(out).read().split("\n")
    return _asm_cache[key]
# What the decode phase's occupancy rests on: five decoder groups of 31 744 bytes of LDS per CU, four
# waves per SIMD (<= 128 registers, arch + accumulation), and beside them three waves of the check kernel (<= 96) with its
# 6 KB of tables; no scratch in either.  {kernel prefix: (LDS bytes max, registers max)}
# Round 6's block-per-lane decoder: five groups of <= 32 768 bytes per CU and exactly five (more than a sixth of the CU's 160 KB
# each: the host sizes the grid for five), four waves per group = five per SIMD (<= 96 registers), in all three instantiations.
BUDGET = {"_Z22x3_decode_split_kernel": (32768, 128), "_Z21x3_frame_check_kernel": (6144, 96),
          "_Z23x3_decode_blocks_kernelILj20ELj1E": (32768, 96), "_Z23x3_decode_blocks_kernelILj20ELj2E": (32768, 96),
          "_Z23x3_decode_blocks_kernelILj10ELj1E": (32768, 96)}
LDS_MIN = {"_Z23x3_decode_blocks_kernelILj20ELj1E": 27308, "_Z23x3_decode_blocks_kernelILj20ELj2E": 27308, "_Z23x3_decode_blocks_kernelILj10ELj1E": 27308}
def resources(flags):
    """-> {kernel prefix: {"lds", "vgpr", "scratch"}} from the .amdhsa_ directives of the compiled unit"""
    lines, res, cur = asm_lines(flags), {}, None
    for l in lines:
        t = l.strip()
        if t.startswith(".amdhsa_kernel"):
            cur = next((k for k in BUDGET if t.split()[1].startswith(k)), None)
            if cur: res[cur] = {}
        elif t.startswith(".end_amdhsa_kernel"):
            cur = None
        elif cur:
            for key, name in ((".amdhsa_group_segment_fixed_size", "lds"), (".amdhsa_private_segment_fixed_size", "scratch"), (".amdhsa_next_free_vgpr", "vgpr")):
                if t.startswith(key + " "):
                    res[cur][name] = int(t.split()[1])
    return res
def over_budget(flags):
    out = []
    for k, r in resources(flags).items():
        lds, vg = BUDGET[k]
        if r.get("lds", 0) > lds or r.get("vgpr", 0) > vg or r.get("scratch", 0) != 0 or r.get("lds", 0) < LDS_MIN.get(k, 0):
            out.append((k, r, BUDGET[k]))
    return out
def kernel_asm(flags):
    lines = asm_lines(flags)
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z22x3_decode_split_kernel") and l.rstrip().endswith(("kernelPKhmPKmm6X3GeomS2_11X3DevParamsPsmPiP11X3FrameMetaPjj9X3SegArgs", ":")) or (l.startswith("_Z22x3_decode_split_kernel") and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith(";") and not l.strip().endswith(":")]
def findings(flags):
    ins = kernel_asm(flags)
    bad = []
    i = 0
    while i < len(ins):
        if ins[i].startswith("global_load_dwordx4"):
            j = i
            run = 0
            while j < len(ins) and (ins[j].startswith("global_load_dwordx4") or (run and not ins[j].startswith(("s_waitcnt", "ds_", "global_", "s_cbranch", "s_branch")) and j - i < run + 6 and any(x.startswith("global_load_dwordx4") for x in ins[j:j + 3]))):
                run += ins[j].startswith("global_load_dwordx4")
                j += 1
            for k in range(j, min(j + 4, len(ins))):
                m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins[k])
                if m and int(m.group(1)) < run and run >= 2:
                    bad.append((i, run, ins[k], ins[k + 1] if k + 1 < len(ins) else ""))
                    break
                if ins[k].startswith(("s_cbranch", "s_branch", "ds_", "global_")):
                    break
            i = j
        else:
            i += 1
    return bad
if __name__ == "__main__":
    b = findings(sys.argv[1:])
    for at, run, w, nxt in b:
        print("instruction %d: %d loads, then `%s` / `%s`" % (at, run, w, nxt))
    print("%d finding(s)" % len(b))
    ob = over_budget(sys.argv[1:])
    for k, r, bud in ob:
        print("%s: %s exceeds LDS %d / registers %d / scratch 0" % (k, r, bud[0], bud[1]))
    print("resources:", resources(sys.argv[1:]))
    sys.exit(1 if b or ob else 0)
