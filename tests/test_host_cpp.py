"""Builds and runs tests/host_cpp/test_x3_hpp.cpp: the C++ mirror of the reference's Rust API
(x3-rust_amd/host/x3.hpp) against the reference's small KATs and the oracle."""
import os
import subprocess

import pytest

import oracle_lib as O
import x3hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_cpp", "test_x3_hpp")


def build():
    O.lib()
    x3hip.lib()
    src = os.path.join(ROOT, "tests", "host_cpp", "test_x3_hpp.cpp")
    deps = [src, os.path.join(ROOT, "x3-rust_amd", "host", "x3.hpp"), os.path.join(ROOT, "include", "x3hip.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) > os.path.getmtime(d) for d in deps):
        return
    libdir = os.path.dirname(x3hip.LIB_PATH)
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-o", EXE, src, "-L" + libdir, "-lx3hip", "-L" + odir, "-lx3oracle",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_shard_offset_logic_two_to_eight_host_threads():
    """tests/host_cpp/test_shard_logic.cpp: host threads play the ranks of the multi-GPU path (no GPU)"""
    O.lib()
    x3hip.lib()
    exe = os.path.join(ROOT, "tests", "host_cpp", "test_shard_logic")
    src = exe + ".cpp"
    libdir = os.path.dirname(x3hip.LIB_PATH)
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", "-o", exe, src, "-L" + libdir, "-lx3hip", "-L" + odir,
                    "-lx3oracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    subprocess.run([exe], check=True, timeout=120)


def test_x3_hpp_host_only():
    build()
    subprocess.run([EXE, "--host-only"], check=True, timeout=60)


@pytest.mark.gpu
def test_x3_hpp_on_gpu():
    build()
    subprocess.run([EXE], check=True, timeout=300)
