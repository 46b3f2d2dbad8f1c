#!/usr/bin/env python3
"""Runs the rgb2bayer kernel (the context's default launch shape, or whatever MIBAYER_R2B_* select) a few times on a
device-resident 4K x 64 batch: the process that tools/r2b_counters.sh wraps in rocprofv3.   Usage: r2b_run.py [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package(lab=True)    # the tuning knobs exist in the lab build only (make lab)
W, H, N = 3840, 2160, 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER) as ctx:
    d_src = ctx.device_alloc(N * ctx.src_bytes)
    d_dst = ctx.device_alloc(N * ctx.dst_bytes)
    ms = ctx.time_device(d_src, d_dst, N, warmup=2, reps=n)
    print("rgb2bayer 4K x 64: %.4f ms per launch, %.1f GB/s" % (ms, 5.0 * W * H * N / ms / 1e6))
