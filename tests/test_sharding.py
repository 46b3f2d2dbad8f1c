"""N > 1 path on CPU: bench.py's harness (round-robin frame sharding, barrier-bracketed timed region,
max-over-ranks) with world_size 2 over gloo.  The data path itself has no collective."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_round_robin_sharding_covers_every_frame_once():
    for total, world in [(64, 1), (64, 2), (512, 8), (10, 4), (3, 8)]:
        shards = [bench.shard_frames(total, world, r) for r in range(world)]
        flat = sorted(f for s in shards for f in s)
        assert flat == list(range(total))
        for r, s in enumerate(shards):
            assert all(f % world == r for f in s)


def test_aggregate_is_whole_job_throughput():
    # 2 ranks x 8.2944 Mpix x 64 frames x 10 steps in 1 s
    v = bench.aggregate_mpix_per_s(3840 * 2160 * 64, 2, 10, 1.0)
    assert abs(v - 2 * 3840 * 2160 * 64 * 10 / 1e6) < 1e-6


def test_timed_region_single_process():
    calls = []
    el = bench.timed_region(lambda i: calls.append(i), steps=4, warmup=3, sync_fn=lambda: None)
    assert calls == list(range(7)) and el >= 0.0


def test_gloo_world_size_2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    r = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert r[0]["frames"] == [0, 2, 4, 6, 8] and r[1]["frames"] == [1, 3, 5, 7, 9]
    # 2 warm-up + exactly 5 timed, then 1 + 3 through the plane with the shared-memory barrier
    assert r[0]["calls"] == r[1]["calls"] == list(range(7)) + list(range(4))
    # both ranks report the same, slowest-rank time: rank 1 sleeps 40 ms/step
    assert abs(r[0]["elapsed"] - r[1]["elapsed"]) < 1e-9
    assert r[0]["elapsed"] >= 5 * 0.04 * 0.9
    # per-GPU breakdown: every rank sees every rank's entry, in rank order
    want = [{"rank": 0, "kernel_ms": 0.4}, {"rank": 1, "kernel_ms": 1.4}]
    assert r[0]["gathered"] == want and r[1]["gathered"] == want
    # the shared-memory barrier of a gloo plane on one host: it holds the early rank, every time, and costs microseconds
    assert r[0]["transport"] == r[1]["transport"] == "shm"
    assert all(h >= 0.12 for h in r[0]["held"]) and all(h < 0.1 for h in r[1]["held"]), (r[0]["held"], r[1]["held"])
    assert abs(r[0]["elapsed_shm"] - r[1]["elapsed_shm"]) < 1e-9 and r[0]["elapsed_shm"] >= 3 * 0.04 * 0.9
    assert max(r[0]["per_barrier_us"], r[1]["per_barrier_us"]) < 5000.0      # a gloo barrier of two local ranks: ~300 us
    assert not os.path.exists(r[0]["shm_path"])                               # rank 0 removed the file
