#!/usr/bin/env python3
"""Development tool: ONE launch over a contiguous 64-frame 4K batch against the same batch cut into S slices that are
launched concurrently on the device's frame queues (hardware queues of their own), for the production block orders.
Wall time per pass incl. launch issue (python ctypes caller).   Usage (GPU box): python tools/split_batch_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
W, H, N, REPS = 3840, 2160, 64, 30
INT32_MIN = -2 ** 31
with pkg.Context(W, H, "rggb", "BGRx") as ctx:
    d_src, d_dst = ctx.device_alloc(N * ctx.src_bytes), ctx.device_alloc(N * ctx.dst_bytes)
    ctx.fill_synthetic(d_src, N, seed=2)
    ctx.sync()
    fq = ctx.frame_queues

    def one_pass(slices, queues):
        per = N // slices
        for s in range(slices):
            ctx.process_device(d_src + s * per * ctx.src_bytes, d_dst + s * per * ctx.dst_bytes, per,
                               stream=("ctx" if queues == 0 else fq[s % queues]))

    def timed(slices, queues):
        for _ in range(5):
            one_pass(slices, queues)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(REPS):
            one_pass(slices, queues)
        ctx.sync()
        return (time.perf_counter() - t0) / REPS

    names = pkg.variant_names()
    print("# 3840x2160 x 64 frames resident in HBM, %d passes per arm, %% of 8 TB/s at 5 B/px; slices x queues "
          "(queues 0 = the context's stream, every launch behind the previous one)" % REPS)
    for rnd in range(2):
        for (v, band) in ((1, 1), (1, -1), (1, 0), (3, 0), (3, 1), (2, 1)):
            ctx.set_plan(v, band, 0)
            row = []
            for (slices, queues) in ((1, 0), (4, 0), (2, 2), (4, 4), (8, 4), (16, 4), (4, 2)):
                t = timed(slices, queues)
                row.append("%dx%d %.1f" % (slices, queues, 5.0 * N * W * H / t / 1e9 / 80))
            print("round %d  %-18s band %2d   %s" % (rnd, names[v], band, "   ".join(row)), flush=True)
    ctx.device_free(d_src)
    ctx.device_free(d_dst)
