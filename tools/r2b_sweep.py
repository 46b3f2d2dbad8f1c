#!/usr/bin/env python3
"""Development tool: interleaved (shuffled) A/B of rgb2bayer launch shapes on one box.
arm = flat:px:ldnt:band:sleep  (flat = groups per thread of the flat kernel, 0 = the tile kernel with `px` meaning
rows per block); every knob is read per context at mibayer_create(), so all arms live in one process.
Usage: python tools/r2b_sweep.py [WxHxN] arm arm ...      (no arms: the default grid)"""
import itertools
import os
import random
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package(lab=True)    # the tuning knobs exist in the lab build only (make lab)
W, H, N = 3840, 2160, 64
args = sys.argv[1:]
if args and "x" in args[0]:
    W, H, N = (int(v) for v in args.pop(0).split("x"))
if not args:
    args = ["%d:%d:%d:%d:0" % (k, px, ld, band) for k, px, ld, band in
            itertools.product((1, 2, 3, 4), (4, 8), (0, 1), (-1, 0))]
    args += ["0:2:0:-1:0", "0:4:0:-1:0", "0:2:0:0:0"]
arms = []
for spec in args:
    flat, px, ld, band, sleep = spec.split(":")
    env = {"MIBAYER_R2B_FLAT": flat, "MIBAYER_R2B_LDNT": ld, "MIBAYER_XCD_BAND": band, "MIBAYER_START_SLEEP": sleep}
    env["MIBAYER_R2B_ROWS" if flat == "0" else "MIBAYER_R2B_PX"] = px
    os.environ.update(env)
    arms.append((spec, pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER), []))
    for k in env:
        del os.environ[k]
c0 = arms[0][1]
d_src = c0.device_alloc(N * c0.src_bytes)
d_dst = c0.device_alloc(N * c0.dst_bytes)
for _ in range(6):          # clock up
    c0.time_device(d_src, d_dst, N, warmup=0, reps=40)
rng = random.Random(7)
for r in range(9):
    order = list(arms)
    rng.shuffle(order)
    for spec, ctx, ts in order:
        t = ctx.time_device(d_src, d_dst, N, warmup=2, reps=10)
        if r:
            ts.append(t)
print("# rgb2bayer %dx%d x %d, flat:px|rows:ldnt:band:sleep, median of 8 shuffled rounds x 10 launches" % (W, H, N))
for spec, ctx, ts in sorted(arms, key=lambda a: statistics.median(a[2])):
    t = statistics.median(ts)
    gbs = 5.0 * W * H * N / t / 1e6
    print("rgb2bayer %-14s median %.4f ms %7.1f GB/s %5.1f%%  (min %.4f max %.4f)" % (
        spec, t, gbs, gbs / 80, min(ts), max(ts)), flush=True)
