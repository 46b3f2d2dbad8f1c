#!/usr/bin/env python3
"""Development tool: interleaved A/B timing of (kernel variant, XCD band) arms.

All arms are created up front and timed for several rounds in ONE process, in a freshly shuffled order each
round, so slow drifts of the box (clock / thermal state) and "who ran before me" effects hit every arm alike;
reports min and median per arm.
Usage: python tools/sweep2.py W H N rounds arm [arm ...]   with arm = variant_name[:band[:start_sleep[:align]]]
(align = MIBAYER_ALIGN_STORES: 0 | 64 | 128, the store-alignment arm for generic geometries)
Environment: SWEEP_SRC_STRIDE / SWEEP_DST_STRIDE (row pitches), SWEEP_DST_OFFSET (bytes added to the destination),
MIBAYER_FORCE_GENERIC=1 (sector-aligned geometries through the generic arm) -- to separate code path, read
misalignment and write misalignment.
(the logs under profiles/ were taken with earlier builds of this tool that also carried knobs for the block->XCD
rotation, an occupancy throttle, a staggered / repositioned delay; those lost and were removed from the product)"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package(lab=True)    # the tuning knobs exist in the lab build only (make lab)
W, H, N, ROUNDS = (int(v) for v in sys.argv[1:5])
names = pkg.variant_names()
arms = []
for spec in sys.argv[5:]:
    name, _, rest = spec.partition(":")
    parts = (rest.split(":") + ["", "", ""])[:3]      # band : start_sleep : align
    for key, val in zip(("MIBAYER_XCD_BAND", "MIBAYER_START_SLEEP", "MIBAYER_ALIGN_STORES"), parts):
        if val:
            os.environ[key] = val
        else:
            os.environ.pop(key, None)
    ctx = pkg.Context(W, H, "rggb", "BGRx", variant=names.index(name),
                      src_stride=int(os.environ.get("SWEEP_SRC_STRIDE", "0")),
                      dst_stride=int(os.environ.get("SWEEP_DST_STRIDE", "0")))
    arms.append((spec, ctx, []))
os.environ.pop("MIBAYER_XCD_BAND", None)
os.environ.pop("MIBAYER_START_SLEEP", None)
os.environ.pop("MIBAYER_ALIGN_STORES", None)
c0 = arms[0][1]
d_src = c0.device_alloc(N * c0.src_bytes)
DST_OFF = int(os.environ.get("SWEEP_DST_OFFSET", "0"))     # bytes: moves the destination off its 256-byte grid
d_dst = c0.device_alloc(N * c0.dst_bytes + 256) + DST_OFF
print("d_src %#x d_dst %#x" % (d_src, d_dst))
c0.fill_synthetic(d_src, N, 2)
c0.sync()
import random
rng = random.Random(1234)
for r in range(ROUNDS + 1):
    order = list(arms)
    rng.shuffle(order)          # a new arm order every round: no systematic "who ran before me" bias
    for spec, ctx, ts in order:
        t = ctx.time_device(d_src, d_dst, N, warmup=2, reps=10)
        if r > 0:
            ts.append(t)
print("%dx%d x %d frames, %d interleaved rounds x 10 launches" % (W, H, N, ROUNDS))
for spec, ctx, ts in sorted(arms, key=lambda a: statistics.median(a[2])):
    g = ctx.launch_geometry(N)
    med, best = statistics.median(ts), min(ts)
    print("%-34s band %4d  median %.4f ms %7.1f GB/s %5.1f%%   best %.4f ms %7.1f GB/s %5.1f%%"
          % (spec, g["band"], med, 5.0 * W * H * N / med / 1e6, 5.0 * W * H * N / med / 1e6 / 80,
             best, 5.0 * W * H * N / best / 1e6, 5.0 * W * H * N / best / 1e6 / 80), flush=True)
