#!/usr/bin/env python3
"""Development tool: interleaved (shuffled) A/B of rgb2bayer launch knobs.  arm = rows:band:sleep"""
import os, random, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, N = 3840, 2160, 64
arms = []
for spec in sys.argv[1:]:
    rows, band, sleep = spec.split(":")
    # rows is read once per process (static): only one rows value per run is meaningful
    os.environ.update(MIBAYER_R2B_ROWS=rows, MIBAYER_XCD_BAND=band, MIBAYER_START_SLEEP=sleep)
    arms.append((spec, pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER), []))
c0 = arms[0][1]
d_src = c0.device_alloc(N * c0.src_bytes); d_dst = c0.device_alloc(N * c0.dst_bytes)
rng = random.Random(7)
for r in range(9):
    order = list(arms); rng.shuffle(order)
    for spec, ctx, ts in order:
        t = ctx.time_device(d_src, d_dst, N, warmup=2, reps=10)
        if r: ts.append(t)
for spec, ctx, ts in sorted(arms, key=lambda a: statistics.median(a[2])):
    t = statistics.median(ts)
    print("rgb2bayer rows:band:sleep %-10s median %.4f ms %7.1f GB/s %5.1f%%" % (spec, t, 5.0 * W * H * N / t / 1e6, 5.0 * W * H * N / t / 1e6 / 80), flush=True)
