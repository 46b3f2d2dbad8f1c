#!/usr/bin/env python3
"""Development tool: times every kernel variant on the 4K x 64 batch (HIP events, C ABI) and prints
achieved algorithmic GB/s.  Usage: python tools/sweep.py [W H N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
W, H, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3840, 2160, 64)
names = pkg.variant_names()
only = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "").split(",") if v] or range(len(names))
with pkg.Context(W, H, "rggb", "BGRx") as c0:
    d_src = c0.device_alloc(N * c0.src_bytes)
    d_dst = c0.device_alloc(N * c0.dst_bytes)
    c0.fill_synthetic(d_src, N, 2)
    c0.sync()
    for v in only:
        with pkg.Context(W, H, "rggb", "BGRx", variant=v) as c:
            best = min(c.time_device(d_src, d_dst, N, warmup=3, reps=20) for _ in range(3))
            gbs = 5.0 * W * H * N / (best * 1e-3) / 1e9
            g = c.launch_geometry(N)
            print("variant %2d %-24s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s  %9.0f Mpix/s  tile %dx%d band %d grid %d"
                  % (v, names[v], best, gbs, gbs / 80.0, W * H * N / best / 1e3, g["tile_w"], g["tile_h"],
                     g["band"], g["grid_blocks"]), flush=True)
    c0.device_free(d_src)
    c0.device_free(d_dst)
