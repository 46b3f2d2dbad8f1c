"""Worker for tests/test_sharding.py: runs bench.py's distributed harness on CPU (gloo)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    frames = bench.shard_frames(10, world, rank)
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.02 * (rank + 1))       # rank 1 is the slow one: max-over-ranks must see it

    elapsed = bench.timed_region(step, steps=5, warmup=2, sync_fn=lambda: None, dist=dist)
    # the per-GPU breakdown travels over the bootstrap group, as in bench.run()
    plane = bench.ControlPlane(dist, None, "gloo", 0)
    gathered = plane.gather_objects({"rank": rank, "kernel_ms": 0.4 + rank})
    # the gloo plane's barriers go over shared memory on one host (bench.ShmBarrier): rank 1 arrives 150 ms late at
    # each of 3 barriers, rank 0 must be held that long every time; then the timed region through the plane
    shm_plane = bench.ControlPlane(dist, None, "gloo", 0, shm=bench.ShmBarrier(dist, rank, world))
    held = []
    for _ in range(3):
        if rank == 1:
            time.sleep(0.15)
        t0 = time.perf_counter()
        shm_plane.barrier()
        held.append(time.perf_counter() - t0)
    info = {}
    elapsed_shm = bench.timed_region(step, steps=3, warmup=1, sync_fn=lambda: None, dist=shm_plane, info=info)
    t0 = time.perf_counter()
    for _ in range(50):
        shm_plane.barrier()
    per_barrier_us = (time.perf_counter() - t0) / 50 * 1e6
    shm_path = shm_plane._shm.path
    shm_plane._shm.close()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "frames": frames, "calls": calls, "elapsed": elapsed,
                   "gathered": gathered, "held": held, "elapsed_shm": elapsed_shm, "barrier_ms": info["barrier_ms"],
                   "transport": shm_plane.barrier_transport, "per_barrier_us": per_barrier_us,
                   "shm_path": shm_path}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
