#!/usr/bin/env python3
"""Development tool: does the block order that wins on small batches (the chunk order, tools/plan_sweep.py at 133-265
Mpixel) still win when the SOURCE is cold, i.e. not left in the 256 MB Infinity Cache by the previous launch?  K source
and K destination batches are rotated per launch (K x source bytes >> 256 MB), against K = 1 (warm).  Wall-clock timing
around back-to-back launches (the launches of a rotation use different pointers, so mibayer_time_device does not apply).
Usage (GPU box): python tools/cold_source_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
names = pkg.variant_names()
SHAPES = [names.index(n) for n in ("lds_4x2_r4_dpp_nt", "lds_2x4_r4_dpp_nt", "lds_1x8_r4_dpp_nt")]
print("# %% of 8 TB/s (5 B/pixel / wall time of back-to-back launches); warm = one source batch re-read every launch, "
      "cold = K batches rotated (K x source >= 1.2 GB)", flush=True)
for (w, h, n) in ((3840, 2160, 16), (3840, 2160, 8), (3840, 2160, 4), (3840, 2160, 32), (1920, 1080, 64), (1920, 1080, 16),
                  (7680, 4320, 4), (4096, 2160, 16), (2592, 1944, 16), (4112, 3008, 8)):
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        src_b, dst_b = n * ctx.src_bytes, n * ctx.dst_bytes
        K = max(2, int(1.2e9 / src_b) + 1)
        K = min(K, 24)
        srcs = [ctx.device_alloc(src_b) for _ in range(K)]
        dsts = [ctx.device_alloc(dst_b) for _ in range(min(K, 6))]
        for s in srcs:
            ctx.fill_synthetic(s, n, 2)
        ctx.sync()
        dv, db, _ = ctx.get_plan()
        plans = [("default", dv, db)] + [("%s/%d" % (names[v].split("_")[1], b), v, b) for v in SHAPES for b in (1, -1, 0)]
        reps = max(12, 2 * K)
        res = {}
        for mode in ("warm", "cold"):
            for rnd in range(3):
                for (label, v, b) in plans:
                    ctx.set_plan(v, b, 0)
                    for i in range(4):
                        ctx.process_device(srcs[i % K if mode == "cold" else 0], dsts[i % len(dsts)], n)
                    ctx.sync()
                    t0 = time.perf_counter()
                    for i in range(reps):
                        ctx.process_device(srcs[i % K if mode == "cold" else 0], dsts[i % len(dsts)], n)
                    ctx.sync()
                    res.setdefault((mode, label), []).append((time.perf_counter() - t0) / reps)
        pct = lambda t: 5.0 * w * h * n / t / 1e9 / 80   # noqa: E731
        for mode in ("warm", "cold"):
            print("%4dx%-4d x %2d (src %4.0f MB, K=%2d) %s: %s" % (
                w, h, n, src_b / 1e6, K if mode == "cold" else 1, mode,
                "  ".join("%s %.1f" % (label, pct(sorted(res[(mode, label)])[1])) for (label, _, _) in plans)), flush=True)
        for p in srcs + dsts:
            ctx.device_free(p)
