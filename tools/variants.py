#!/usr/bin/env python3
"""Build experiment variants of libx3hip.so side by side (CPU container; hipcc cross-compiles gfx950):

    tools/variants.py name1="-DFLAG=1 -DOTHER=2" name2="" ...

writes x3-rust_amd/lib/variants/libx3hip_<name>.so (git-ignored, travels with gpurun).  tools/kbench.py and the
tools/scratch scripts take a variant through X3HIP_LIB=<path>.  tools/run_variants.sh runs kbench over all of them."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "x3-rust_amd")
OUT = os.path.join(PKG, "lib", "variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-pthread"]


def build(arg):
    name, _, flags = arg.partition("=")
    lib = os.path.join(OUT, "libx3hip_%s.so" % name)
    import importlib.util
    spec = importlib.util.spec_from_file_location("x3_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    try:
        b.build_lib(lib, extra=flags.split(), objdir=os.path.join(OUT, "obj_" + name), verbose=False)
        return name, 0, ""
    except subprocess.CalledProcessError as e:
        return name, 1, str(e)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "6"))) as ex:
        for name, rc, err in ex.map(build, sys.argv[1:]):
            print("%-24s %s" % (name, "ok" if rc == 0 else "FAILED\n" + err), flush=True)
