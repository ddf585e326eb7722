"""The library's HOST code that reads bytes somebody else wrote -- archive headers, WAV headers, frame headers, the host's
frame walk, the shard arithmetic -- under AddressSanitizer + UndefinedBehaviorSanitizer, fuzzed, against the oracle
(tests/host_cpp/fuzz_host_parsers.cpp).  CPU only: the library's five translation units are compiled
with the HOST side sanitised (the device side as usual: the registration of the kernels needs their code objects; nothing
here creates a context or launches a kernel) and linked with the driver and a sanitised oracle.  GPU sanitizers are not
available on the pool; the host side is where untrusted files are parsed."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "x3-rust_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
UNITS = ["x3_ctx", "x3_encode", "x3_decode", "x3_files", "x3_mgpu"]
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]


def _newest_source():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "x3hip.h"),
            os.path.join(ROOT, "oracle", "x3_oracle.c"), os.path.join(ROOT, "oracle", "x3_oracle.h"),
            os.path.join(ROOT, "tests", "host_cpp", "fuzz_host_parsers.cpp")]
    return max(os.path.getmtime(s) for s in srcs)


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("gcc") is None, reason="hipcc / gcc not found")
def test_host_parsers_are_clean_under_asan_and_ubsan_and_agree_with_the_oracle():
    out = os.path.join(ROOT, "tests", "host_cpp", "_san")     # (git-ignored; rebuilt when a source is newer)
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "fuzz_host_parsers")
    if not os.path.exists(exe) or os.path.getmtime(exe) < _newest_source():
        def cc(u):
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-g0", "-std=c++17", "-fPIC", "-Wno-unused-function", "-pthread"] + SAN +
                           ["-c", "-o", os.path.join(out, u + ".o"), os.path.join(CSRC, u + ".hip")], check=True, capture_output=True)
        with ThreadPoolExecutor(max_workers=5) as ex:
            list(ex.map(cc, UNITS))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-x", "hip", "-O1", "-g", "-std=c++17", "-Wno-unused-function"] + SAN +
                       ["-I", CSRC, "-I", os.path.join(ROOT, "include"), "-c", "-o", os.path.join(out, "fuzz.o"),
                        os.path.join(ROOT, "tests", "host_cpp", "fuzz_host_parsers.cpp")], check=True, capture_output=True)
        subprocess.run(["gcc", "-O1", "-g"] + SAN + ["-c", "-o", os.path.join(out, "oracle.o"), os.path.join(ROOT, "oracle", "x3_oracle.c")],
                       check=True, capture_output=True)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-pthread"] + SAN + ["-o", exe, os.path.join(out, "fuzz.o"), os.path.join(out, "oracle.o")] +
                       [os.path.join(out, u + ".o") for u in UNITS], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, "100000"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok archives="), (r.stdout[-3000:], r.stderr[-4000:])
    # 10^5 mutations each of archive, WAV and frame headers, every prefix of the valid ones, 25 000 damaged streams
    counts = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(counts["archives"]) >= 100000 and int(counts["wavs"]) >= 100000 and int(counts["walks"]) >= 25000, counts
