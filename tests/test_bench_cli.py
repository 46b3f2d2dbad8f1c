"""bench.py contract checks that need no GPU: defaults, the JSON keys the driver reads, and the loud failure
without a device (the hot path has no CPU fallback, so neither has the benchmark)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_defaults_and_workload_constants():
    assert (bench.WIDTH, bench.HEIGHT, bench.BATCH) == (3840, 2160, 64)      # BASELINE.json configs[2]
    assert bench.BYTES_PER_PIXEL == 5 and bench.HBM_PEAK_GBPS == 8000.0
    assert set(bench.ORDERS) == {"bggr", "rggb", "grbg", "gbrg"}


def test_without_a_gpu_the_benchmark_refuses(pkg):
    if pkg.device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode != 0
    assert "no MI355X visible" in (res.stderr + res.stdout)
    assert "{" not in res.stdout           # no JSON line is fabricated
