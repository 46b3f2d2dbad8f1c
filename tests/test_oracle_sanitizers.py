"""The oracle judges the HIP path, so it must itself be free of out-of-bounds accesses and UB: build it with
AddressSanitizer + UBSan and run it over exact-size heap buffers (tests/check/oracle_asan.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_asan")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=gnu11", "-Wall",
           "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "check", "oracle_asan.c"),
           os.path.join(ROOT, "oracle", "bayer2rgb_oracle.c"), os.path.join(ROOT, "oracle", "bayer2rgb_simd.c"),
           "-o", exe, "-ldl", "-lpthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and "sanitize" in res.stderr:
        pytest.skip("sanitizer runtime not available: " + res.stderr[-200:])
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-2000:]
    assert "conversions ok" in run.stdout
