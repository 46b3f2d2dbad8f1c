#!/usr/bin/env python3
"""Host-path throughput of mibayer_pool with PAGEABLE frame buffers (an upstream that ignores the proposed pinned
pool, e.g. filesrc): per-shard helper threads (default) against everything on the calling thread
(MIBAYER_POOL_HELPERS=0), and pinned buffers beside it.  N logical shards on the visible GPU(s).
Usage (GPU box): python tools/pool_pageable_bench.py [shards] [frames]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
L = pkg.lib()
W, H = 3840, 2160
SHARDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ndev = pkg.device_count()


def pinned(nbytes, shape):
    p = L.mibayer_host_alloc(nbytes)
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape)


def run(memory, helpers, inflight=2, threads="0"):
    os.environ["MIBAYER_POOL_HELPERS"] = helpers
    os.environ["MIBAYER_POOL_THREADS"] = threads          # "1": a submit thread per shard, pinned frames included
    with pkg.Pool([i % ndev for i in range(SHARDS)], W, H, "rggb", "BGRx", inflight=inflight) as pool:
        cap = pool.capacity
        if memory == "pinned":
            bufs = [(pinned(W * H, (H, W)), pinned(4 * W * H, (H, 4 * W))) for _ in range(cap)]
        else:
            bufs = [((0, np.full((H, W), 0x55, np.uint8)), (0, np.zeros((H, 4 * W), np.uint8))) for _ in range(cap)]
        for (_, s), (_, d) in bufs:
            s[:] = 0x55
            d[:] = 0
        for phase, n in (("warm", 2 * cap), ("timed", FRAMES)):
            t0 = time.perf_counter()
            for i in range(n):
                if pool.pending() == cap:
                    pool.wait()
                (_, s), (_, d) = bufs[i % cap]
                pool.submit(s, d, tag=i + 1)
            while pool.pending():
                pool.wait()
            el = time.perf_counter() - t0
        if memory == "pinned":
            for (ps, _), (pd, _) in bufs:
                L.mibayer_host_free(ps)
                L.mibayer_host_free(pd)
    return FRAMES / el


print("# 4K bayer2rgb host path through mibayer_pool, %d logical shard(s) on %d GPU(s), 2 frames in flight per shard, "
      "%d frames" % (SHARDS, ndev, FRAMES))
for memory, helpers, threads, label in (
        ("pinned", "1", "0", "pinned buffers, the calling thread enqueues for every shard"),
        ("pinned", "1", "1", "pinned buffers, a submit thread per shard (MIBAYER_POOL_THREADS=1)"),
        ("pageable", "0", "0", "pageable buffers, copies on the calling thread (helpers off)"),
        ("pageable", "1", "0", "pageable buffers, one helper thread per shard"),
        ("pageable", "1", "1", "pageable buffers, a submit thread per shard from the start")):
    fps = run(memory, helpers, threads=threads)
    print("%-64s %7.1f fps  %8.1f Mpix/s  D2H %5.1f GB/s" % (label, fps, fps * W * H / 1e6, fps * 4 * W * H / 1e9),
          flush=True)
