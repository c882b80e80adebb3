"""
The host-side packing code (csrc/cf_pack.h: both factor packings, the threaded long-double inversion, the create-time probes)
under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU (GPU sanitizers are not available on the pool).  The driver
(tools/pack_sanitize.cpp) packs factors of sizes on and off every tile / row-block / 256-row-block boundary, with a leading
dimension larger than n and NaN above the diagonal (only L[i][j <= i] may be read: sn/pantheon.py:14), replays both fragment streams
on the host and compares with row-by-row substitution.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "pack_sanitize.cpp")


@pytest.mark.parametrize("flags,sizes", [
    (["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], []),   # the driver's default size list
    (["-fsanitize=thread"], ["64", "257", "513"]),                          # the inversion's worker threads
])
def test_packing_code_is_clean_under_sanitizers(tmp_path, flags, sizes):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "pack_sanitize")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", *flags, "-pthread", SRC, "-o", exe], check=True, cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, *sizes], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "probe" in r.stdout and "Sanitizer" not in r.stderr
