"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: the checker the parity tests lean on must itself be
free of out-of-bounds accesses and undefined arithmetic on every input it is given, damaged streams included
(tests/host_c/oracle_sanitize.c; CPU only -- sanitizers are not available for the GPU code on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "host_c", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "x3_oracle.c"),
                    "-lm"], check=True)
    r = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "trials clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
