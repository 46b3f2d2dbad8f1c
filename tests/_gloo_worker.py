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
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "frames": frames, "calls": calls, "elapsed": elapsed,
                   "gathered": gathered}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
