#!/usr/bin/env python3
"""Build libx3hip.so (gfx950 HIP kernels + the C ABI of include/x3hip.h) in-tree.

    python x3-rust_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container;
the built .so (git-ignored) travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# the translation units of the library (x3_internal.h says what each one holds)
UNITS = ["x3_ctx.hip", "x3_encode.hip", "x3_decode.hip", "x3_files.hip", "x3_mgpu.hip"]
DEPS = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "x3hip.h")]
LIB = os.path.join(HERE, "lib", "libx3hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
CLI = os.path.join(HERE, "bin", "x3")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-pthread"]


def stale():
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(CLI))
    return any(os.path.getmtime(d) > t for d in DEPS + [os.path.join(HERE, "cli", "x3.cpp"), os.path.join(HERE, "host", "x3.hpp")])


def build_lib(lib=LIB, extra=(), objdir=OBJ, verbose=True):
    """hipcc -c every unit (side by side), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    os.makedirs(objdir, exist_ok=True)

    def cc(u):
        o = os.path.join(objdir, u.replace(".hip", ".o"))
        cmd = [HIPCC] + CFLAGS + list(extra) + ["-c", "-o", o, os.path.join(CSRC, u)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return o
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(cc, UNITS))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    build_lib(verbose=verbose)
    build_cli(verbose)
    return LIB


def build_cli(verbose=True):
    """the reference's `x3 -i/-o` command line on top of the library (cli/x3.cpp)"""
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(HERE, "..", "include"), "-o", CLI,
           os.path.join(HERE, "cli", "x3.cpp"), "-L", os.path.dirname(LIB), "-lx3hip", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
