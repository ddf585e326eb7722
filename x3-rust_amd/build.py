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
SRC = os.path.join(HERE, "csrc", "x3_api.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(HERE, "..", "include", "x3hip.h")]
LIB = os.path.join(HERE, "lib", "libx3hip.so")
CLI = os.path.join(HERE, "bin", "x3")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-pthread"]


def stale():
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(CLI))
    return any(os.path.getmtime(d) > t for d in DEPS + [os.path.join(HERE, "cli", "x3.cpp"), os.path.join(HERE, "host", "x3.hpp")])


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
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
