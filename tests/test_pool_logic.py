"""The frame-sharding pool's own logic (gst-plugins-bad_amd/csrc/mibayer_pool.cpp: round-robin order, device failover,
helper threads for pageable buffers) on a machine WITHOUT a GPU.

The pool is pure host code above the per-device context ABI, so the REAL source file is compiled against the test double
of the contexts (tests/check/mock_mibayer.c) and driven by tests/check/pool_logic.cpp -- once under AddressSanitizer,
once under ThreadSanitizer (the helper threads share frame states with the streaming thread).  Exit codes of the driver
encode which invariant broke (order, source identity, capacity accounting)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gst-plugins-bad_amd", "csrc")
CHECK = os.path.join(ROOT, "tests", "check")


@pytest.fixture(scope="module", params=["address,undefined", "thread"])
def driver(request, tmp_path_factory):
    d = str(tmp_path_factory.mktemp("poollogic_" + request.param.replace(",", "_")))
    san = ["-O1", "-g", "-fsanitize=" + request.param, "-fno-omit-frame-pointer", "-Wall"]
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    obj = os.path.join(d, "mock.o")
    res = subprocess.run(["gcc"] + san + inc + ["-c", os.path.join(CHECK, "mock_mibayer.c"), "-o", obj],
                         capture_output=True, text=True)
    if res.returncode != 0 and "sanitize" in res.stderr:
        pytest.skip("sanitizer runtime not available: " + res.stderr[-200:])
    assert res.returncode == 0, res.stderr[-2000:]
    exe = os.path.join(d, "pool_logic")
    res = subprocess.run(["g++", "-std=c++17"] + san + inc
                         + [os.path.join(CHECK, "pool_logic.cpp"), os.path.join(CSRC, "mibayer_pool.cpp"), obj,
                            "-o", exe, "-lpthread"], capture_output=True, text=True)
    if res.returncode != 0 and "sanitize" in res.stderr:
        pytest.skip("sanitizer runtime not available: " + res.stderr[-200:])
    assert res.returncode == 0, res.stderr[-2000:]
    # probe: some kernels refuse TSan's memory layout
    probe = subprocess.run([exe, "1", "1", "1", "0", "-", "ok"], capture_output=True, text=True)
    if probe.returncode != 0 and "unexpected memory mapping" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot run here: " + probe.stderr[-200:])
    return exe


def run(exe, shards, inflight, frames, pageable, faults, expect, env=None):
    res = subprocess.run([exe, str(shards), str(inflight), str(frames), str(int(pageable)), faults, expect],
                         capture_output=True, text=True, timeout=120, env=dict(os.environ, **(env or {})))
    out = res.stdout + res.stderr
    assert "Sanitizer" not in out, out[-4000:]
    assert res.returncode == 0, (res.returncode, out[-2000:])
    return dict(kv.split("=") for kv in res.stdout.split() if "=" in kv), res.stderr


@pytest.mark.parametrize("pageable", [False, True], ids=["pinned", "pageable"])
def test_round_robin_order_without_faults(driver, pageable):
    for shards, inflight, frames in ((1, 1, 5), (1, 3, 17), (4, 2, 50), (8, 1, 33), (16, 2, 70)):
        kv, _ = run(driver, shards, inflight, frames, pageable, "-", "ok")
        assert kv["delivered"] == str(frames) and kv["dropped_devices"] == "0"
        assert kv["alive"] == str(shards) and kv["capacity"] == str(shards * inflight)


@pytest.mark.parametrize("pageable", [False, True], ids=["pinned", "pageable"])
def test_a_failed_device_is_dropped_and_its_frames_are_redone(driver, pageable):
    # one of four devices fails after 3 frames: every frame still comes out once, in order, from its own source
    kv, err = run(driver, 4, 2, 60, pageable, "1:3", "ok")
    assert kv["delivered"] == "60" and kv["dropped_devices"] == "1" and kv["alive"] == "3" and kv["capacity"] == "6"
    assert "dropped from the rotation" in err and err.count("note:") == 1          # reported exactly once
    # it fails at once; two fail at different times; all but one fail
    for faults, alive in (("2:0", 3), ("0:5,3:9", 2), ("0:1,1:2,2:3", 1)):
        kv, _ = run(driver, 4, 3, 80, pageable, faults, "ok")
        assert kv["delivered"] == "80" and kv["alive"] == str(alive), (faults, kv)
        assert kv["dropped_devices"] == str(4 - alive)


@pytest.mark.parametrize("pageable", [False, True], ids=["pinned", "pageable"])
def test_the_stream_fails_only_when_no_device_is_left(driver, pageable):
    kv, _ = run(driver, 2, 2, 40, pageable, "0:3,1:6", "dead")
    assert kv["alive"] == "0" and kv["capacity"] == "0" and kv["rc"] == "-5"        # MIBAYER_ERR_HIP
    assert 3 <= int(kv["delivered"]) <= 9           # what completed before the oldest undeliverable frame came out
    kv, _ = run(driver, 1, 2, 10, pageable, "0:4", "dead")
    assert kv["delivered"] in ("3", "4") and kv["rc"] == "-5"   # 4 completed; the last may still be undelivered


def test_environment_fault_injection_matches_the_api(driver):
    kv, _ = run(driver, 3, 2, 30, False, "-", "ok", env={"MIBAYER_INJECT_FAULT": "2:4"})
    assert kv["delivered"] == "30" and kv["alive"] == "2"


def test_helpers_can_be_switched_off(driver):
    # MIBAYER_POOL_HELPERS=0: pageable frames take the direct path on the calling thread (A/B knob)
    kv, _ = run(driver, 4, 2, 40, True, "1:2", "ok", env={"MIBAYER_POOL_HELPERS": "0"})
    assert kv["delivered"] == "40" and kv["alive"] == "3"


def test_randomised_scenarios(driver):
    """Seeded random pools: 1-8 shards, 1-3 frames in flight each, 5-90 frames, pinned or pageable, a random subset of
    the shards failing at random points.  Every scenario either delivers every frame once, in order, from its own source
    (exit 0), or -- only if EVERY shard was given a fault -- may end with the stream dead (exit 21); no sanitizer report."""
    import random
    rng = random.Random(20260927)
    for _ in range(40):
        shards, inflight, frames, pageable = rng.randint(1, 8), rng.randint(1, 3), rng.randint(5, 90), rng.randint(0, 1)
        which = rng.sample(range(shards), rng.randint(0, shards))
        faults = ",".join("%d:%d" % (s, rng.randint(0, 12)) for s in which) or "-"
        res = subprocess.run([driver, str(shards), str(inflight), str(frames), str(pageable), faults, "ok"],
                             capture_output=True, text=True, timeout=120)
        out = res.stdout + res.stderr
        assert "Sanitizer" not in out, out[-3000:]
        assert res.returncode in (0, 21), (res.returncode, shards, inflight, frames, pageable, faults, out[-1000:])
        if res.returncode == 21:
            assert len(which) == shards, (shards, faults, out[-500:])


@pytest.mark.parametrize("pageable", [False, True], ids=["pinned", "pageable"])
def test_a_device_that_stops_answering_is_dropped_after_the_deadline(driver, pageable):
    """MOCK_MIBAYER_HANG: the context's waits run into the deadline (MIBAYER_ERR_TIMEOUT).  The shard leaves the
    rotation, ONE note, and the pool never waits for that context again (the double aborts on a wait without deadline,
    and on the destroy of a hung context that was not abandoned).  What the stalled device had IN FLIGHT is not
    converted again behind its back: those frames come back LOST, in order, at most `inflight` of them per
    stalled shard, their buffers stay quarantined until mibayer_pool_reclaim hands the tags back -- the double's
    device resumes after a few polls and writes them late, into buffers the driver must still hold -- and every
    other frame is delivered."""
    env = {"MOCK_MIBAYER_HANG": "1:3", "POOL_LOGIC_TIMEOUT_MS": "30"}
    kv, err = run(driver, 4, 2, 60, pageable, "-", "ok", env=env)
    assert int(kv["delivered"]) + int(kv["lost"]) == 60 and 1 <= int(kv["lost"]) <= 2, kv
    assert kv["reclaimed"] == kv["lost"] and kv["dropped_devices"] == "1" and kv["alive"] == "3"
    assert err.count("note:") == 1 and "dropped from the rotation" in err and "in flight on it lost" in err
    # two of three stop answering at different times; hang and error mixed
    kv, _ = run(driver, 3, 2, 50, pageable, "-", "ok", env={"MOCK_MIBAYER_HANG": "0:2,2:7", "POOL_LOGIC_TIMEOUT_MS": "20"})
    assert int(kv["delivered"]) + int(kv["lost"]) == 50 and 2 <= int(kv["lost"]) <= 4 and kv["alive"] == "1", kv
    kv, _ = run(driver, 4, 3, 70, pageable, "3:4", "ok", env={"MOCK_MIBAYER_HANG": "1:5", "POOL_LOGIC_TIMEOUT_MS": "20"})
    assert int(kv["delivered"]) + int(kv["lost"]) == 70 and 1 <= int(kv["lost"]) <= 3 and kv["alive"] == "2", kv
    # a device that never comes back: the lost buffers stay quarantined to the end (released after the pool is gone)
    kv, _ = run(driver, 4, 2, 40, pageable, "-", "ok",
                env={"MOCK_MIBAYER_HANG": "2:4", "POOL_LOGIC_TIMEOUT_MS": "20", "MOCK_MIBAYER_RESUME_POLLS": "-1"})
    assert int(kv["delivered"]) + int(kv["lost"]) == 40 and int(kv["lost"]) >= 1 and kv["reclaimed"] == "0", kv
    # every device hangs: the stream ends with the timeout status, nothing blocks
    kv, _ = run(driver, 2, 2, 30, pageable, "-", "dead", env={"MOCK_MIBAYER_HANG": "0:3,1:5", "POOL_LOGIC_TIMEOUT_MS": "20"})
    assert kv["alive"] == "0" and kv["rc"] in ("-9", "-5")


def test_stall_drill_through_the_pool_api(driver):
    kv, err = run(driver, 3, 2, 40, False, "-", "ok", env={"POOL_LOGIC_STALL": "2", "POOL_LOGIC_TIMEOUT_MS": "25"})
    assert int(kv["delivered"]) + int(kv["lost"]) == 40 and int(kv["lost"]) <= 2, kv
    assert kv["alive"] == "2" and err.count("note:") == 1


@pytest.mark.parametrize("threads", ["0", "1"], ids=["streaming_thread", "thread_per_shard"])
def test_numa_local_routing_keeps_order_and_balance(driver, threads):
    """Two fake NUMA nodes, four devices (device d on node d % 2), frame f's buffers allocated next to
    devices[f % 4] as the element's pinned pool does: every frame is converted on the node that holds its buffer,
    in order; with a device gone the survivors on the right node take over as far as the balance allows, and the
    stream still delivers everything.  MIBAYER_POOL_THREADS=1: a submit thread per shard (pinned frames too)."""
    env = {"MOCK_MIBAYER_DEVICES": "4", "MOCK_MIBAYER_NUMA_NODES": "2", "POOL_LOGIC_DISTINCT": "1",
           "POOL_LOGIC_NEAR": "1", "MIBAYER_POOL_THREADS": threads}
    kv, _ = run(driver, 4, 2, 96, False, "-", "ok", env=env)
    assert kv["delivered"] == "96" and int(kv["local"]) >= 0.9 * 96, kv
    # buffers handed out in an order that does not match the rotation: frame f next to devices[(f * 3 + 1) % 4]
    # is not what the driver does, but a dead device gives the same effect -- its node-mates take its frames
    kv, _ = run(driver, 4, 2, 96, False, "1:5", "ok", env=env)
    assert kv["delivered"] == "96" and kv["alive"] == "3" and int(kv["local"]) >= 0.6 * 96, kv
    # routing off: strictly g mod N (same result here, since the buffers follow the rotation)
    kv, _ = run(driver, 4, 2, 40, False, "-", "ok", env=dict(env, MIBAYER_POOL_NUMA="0"))
    assert kv["delivered"] == "40"
    # one node only: nothing to route
    kv, _ = run(driver, 4, 2, 40, False, "-", "ok", env=dict(env, MOCK_MIBAYER_NUMA_NODES="1"))
    assert kv["delivered"] == "40" and kv["local"] == "40"


def test_thread_per_shard_with_faults(driver):
    for faults in ("-", "2:4", "0:1,3:6"):
        kv, _ = run(driver, 4, 2, 70, False, faults, "ok", env={"MIBAYER_POOL_THREADS": "1"})
        assert kv["delivered"] == "70", (faults, kv)
    # pools over six or more DISTINCT devices start with a submit thread per shard by default (one streaming thread
    # saturates at what eight PCIe links carry): eight fake GPUs, pinned frames, with and without a failing / a
    # stalling device; MIBAYER_POOL_THREADS=0 keeps the single enqueue-only thread
    env8 = {"MOCK_MIBAYER_DEVICES": "8", "POOL_LOGIC_DISTINCT": "1"}
    for faults, extra in (("-", {}), ("5:3", {}), ("-", {"MIBAYER_POOL_THREADS": "0"}),
                          ("-", {"MOCK_MIBAYER_HANG": "2:3", "POOL_LOGIC_TIMEOUT_MS": "20"})):
        kv, _ = run(driver, 8, 2, 120, False, faults, "ok", env=dict(env8, **extra))
        assert int(kv["delivered"]) + int(kv["lost"]) == 120, (faults, extra, kv)
        assert kv["alive"] == ("8" if faults == "-" and "MOCK_MIBAYER_HANG" not in extra else "7"), kv


def test_a_redone_frame_is_not_handed_back_before_its_dead_context_is_abandoned(driver):
    """a helper thread sees its device fail near the end of the stream; the streaming thread, which is
    only draining, re-does the frames on another shard.  Before they are handed back, the context they came from
    must have been taken out and abandoned -- the double lets a failed, un-abandoned context carry out its queued
    writes late, into buffers the driver frees on delivery."""
    for spec in ((4, 2, 11), (4, 2, 10), (3, 3, 9), (4, 3, 14), (2, 2, 7)):
        for fail in ("1:2", "2:1", "3:2", "0:3"):
            kv, _ = run(driver, spec[0], spec[1], spec[2], True, "-", "ok", env={"MOCK_MIBAYER_FAIL": fail})
            assert kv["delivered"] == str(spec[2]), (spec, fail, kv)
