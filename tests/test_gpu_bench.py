"""bench.py on the GPU box: the RCCL control plane of the N > 1 flow brought up by a single rank (--force-dist), and
`roofline.traffic` measured inside the bench run itself (the profiled child under rocprofv3).  CPU part: the parser of
the PMC csv on a committed profile."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def run_bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=timeout, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-1500:] + res.stderr[-3000:]
    return json.loads(lines[0]), res.stderr


def test_pmc_csv_parser_on_a_committed_profile():
    """The parser behind `roofline.traffic` on the round-3 PMC passes of the bench kernel (band 1): read 1.249 x,
    write 1.000 x the algorithmic bytes, the figures profiles/r03_summary.md reports."""
    alg_r, alg_w = 3840 * 2160 * 64, 4 * 3840 * 2160 * 64
    name, fetch, n = bench.parse_pmc_csv(os.path.join(ROOT, "profiles", "r03_pmc_bench_band1_FETCH_SIZE.csv"),
                                         "FETCH_SIZE", last=8)
    assert n == 8 and "bayer2rgb_lds_kernel<4, 2, 4, 0, 1, true, false>" in name
    assert abs(fetch * bench.PMC_TO_BYTES["FETCH_SIZE"] / alg_r - 1.249) < 0.005
    name, write, n = bench.parse_pmc_csv(os.path.join(ROOT, "profiles", "r03_pmc_bench_band1_WRITE_SIZE.csv"),
                                         "WRITE_SIZE", last=8)
    assert n == 8 and abs(write * bench.PMC_TO_BYTES["WRITE_SIZE"] / alg_w - 1.0) < 0.002
    # a counter the file does not hold, a kernel it does not hold
    assert bench.parse_pmc_csv(os.path.join(ROOT, "profiles", "r03_pmc_bench_band1_WRITE_SIZE.csv"),
                               "FETCH_SIZE") == (None, None, 0)
    assert bench.parse_pmc_csv(os.path.join(ROOT, "profiles", "r03_pmc_bench_band1_WRITE_SIZE.csv"),
                               "WRITE_SIZE", kernel_substr="no_such_kernel") == (None, None, 0)


@pytest.mark.gpu
def test_force_dist_brings_rccl_up_on_one_rank(gpu_pkg):
    """the process-group set-up of the N > 1 flow -- gloo bootstrap, an RCCL group, the probe
    all-reduce, both barriers of the timed region and the MAX reduce on a device tensor -- executed on real hardware
    by one rank, so the driver's SCALE run is not the first process that ever takes the RCCL branch."""
    common = ["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu", "--no-host-path", "--no-traffic"]
    plain, _ = run_bench(common)
    dist, err = run_bench(common + ["--force-dist"])
    assert dist["control_plane"] == "nccl" and dist["control_plane_fallback"] is None, err[-2000:]
    assert dist["config"]["control_plane"] == "nccl"
    assert plain["control_plane"] == "single process" and plain["barrier_ms"] == 0.0
    assert "bit-exact" in dist["config"]["parity"] and "bit-exact" in plain["config"]["parity"]
    assert dist["barrier_ms"] >= 0.0 and dist["value_kernel_only"] >= dist["value"] * 0.999
    # the kernel-only rate is what must agree (the wall-clock value carries the barrier, reported beside it); two
    # processes may land on different block orders of the allocation lottery: up to ~3 %
    assert abs(dist["value_kernel_only"] - plain["value_kernel_only"]) / plain["value_kernel_only"] < 0.06, (dist, plain)
    # 5 steps = 2 ms: one RCCL barrier may cost a few per cent of that; it must not cost more
    assert dist["value"] > 0.85 * plain["value"], (dist["value"], plain["value"], dist["barrier_ms"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_single_rank.json"), "w") as f:
        json.dump({"force_dist": dist, "plain": plain}, f)


@pytest.mark.gpu
def test_traffic_is_measured_inside_the_bench_run(gpu_pkg):
    """`roofline.traffic` is a measurement of THIS run (a profiled child of bench.py under
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE after the timed region), for the very plan the timed region ran."""
    if bench.find_rocprofv3() is None:
        pytest.fail("rocprofv3 is not installed on the GPU box")
    j, err = run_bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu", "--no-host-path"])
    t = j["roofline"]["traffic"]
    assert t is not None, j["roofline"].get("traffic_note")
    assert 0.99 <= t["ratio"] <= 1.30 and 0.99 <= t["write_ratio"] <= 1.02 and 0.99 <= t["read_ratio"] <= 2.1, t
    assert t["total"] == t["read"] + t["write"] and t["launches_averaged"] == 8
    # the profiled kernel is the template the timed region ran
    shape = j["config"]["kernel_variant"].split("_")[1]             # "4x2"
    wx, wy = shape.split("x")
    assert "bayer2rgb_lds" in t["kernel"] and "<%s, %s, 4," % (wx, wy) in t["kernel"], (t["kernel"], j["config"])
    assert t["child_kernel_variant"] == j["config"]["kernel_variant"]
    assert t["seconds"] < 120
