"""The boundary is a C ABI: a C99 translation unit including include/mibayer.h compiles with -pedantic -Werror,
links against libmibayer.so alone and behaves (refused without a GPU; the SURVEY B.4 known answer with one)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def consumer(pkg, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cabi") / "abi_c_consumer")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "check", "abi_c_consumer.c"), "-o", exe,
           "-L", os.path.dirname(pkg.LIB_PATH), "-lmibayer", "-Wl,-rpath," + os.path.dirname(pkg.LIB_PATH)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c99_consumer_without_gpu_is_refused(pkg, consumer):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    res = subprocess.run([consumer], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "no device" in res.stdout


@pytest.mark.gpu
def test_c99_consumer_known_answer_on_gpu(gpu_pkg, consumer):
    res = subprocess.run([consumer], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "known answer ok" in res.stdout
