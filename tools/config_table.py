#!/usr/bin/env python3
"""Development tool: device-resident throughput of every BASELINE.json geometry on one MI355X, after
mibayer_autotune, as a markdown table (HIP events through mibayer_time_device).  Run on the GPU box."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
CONFIGS = [
    ("configs[0] geometry", 640, 480, 1, "bggr", "RGBx", 7),
    ("configs[0] geometry, batched", 640, 480, 2048, "bggr", "RGBx", 7),
    ("configs[1]", 1920, 1080, 1, "rggb", "BGRx", 1),
    ("configs[1] geometry, batched", 1920, 1080, 256, "rggb", "BGRx", 1),
    ("configs[2] (headline)", 3840, 2160, 64, "rggb", "BGRx", 2),
    ("configs[2], single frame", 3840, 2160, 1, "rggb", "BGRx", 2),
    ("configs[3] per-GPU share (512/8)", 7680, 4320, 64, "bggr", "RGBx", 3),
]
print("| config | W×H × frames | working set | plan after autotune | ms / launch | Mpix/s | GB/s (5 B/px) | % of 8 TB/s |")
print("|---|---|---:|---|---:|---:|---:|---:|")
for name, w, h, n, pat, fmt, seed in CONFIGS:
    with pkg.Context(w, h, pat, fmt) as c:
        d_src = c.device_alloc(n * c.src_bytes)
        d_dst = c.device_alloc(n * c.dst_bytes)
        c.fill_synthetic(d_src, n, seed)
        c.sync()
        c.autotune(d_src, d_dst, n)
        g = c.launch_geometry(n)
        ts = [c.time_device(d_src, d_dst, n, warmup=2, reps=20 if n * w * h > 4e6 else 200) for _ in range(5)]
        t = statistics.median(ts)
        px = w * h * n
        print("| %s | %d×%d × %d | %.2f GB | %s, band %d | %.4f | %.0f | %.1f | %.1f |" % (
            name, w, h, n, 5 * px / 1e9, c.variant_name_for(n), g["band"], t, px / t / 1e3, 5 * px / t / 1e6,
            5 * px / t / 1e6 / 80), flush=True)
        c.device_free(d_src)
        c.device_free(d_dst)
print()
print("Working sets below ~0.5 GB are (partly) resident in the 256 MB Infinity Cache, so their GB/s is not an HBM "
      "figure; single-frame launches are launch-latency-bound (a 4K frame is ~7 us of kernel).")
