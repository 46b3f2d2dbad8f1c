#!/usr/bin/env python3
"""Development tool: default plan vs measured plan vs the plan a LATER context of the same geometry gets from the
process-wide plan cache (no measurement), for common camera geometries, device-resident batches of ~530 Mpixel.
Usage (GPU box): python tools/common_geometries.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
SRC = {0: "default", 1: "measured", 2: "cached", 3: "set"}
print("# default plan (variant 0), measured plan (mibayer_autotune) and what the NEXT context of that geometry starts from "
      "(process-wide plan cache), device-resident batches, % of 8 TB/s")
for (w, h) in ((640, 480), (1280, 720), (1280, 960), (1600, 1200), (2048, 1536), (2560, 1440), (2592, 1944), (3264, 2448),
               (4000, 3000), (4096, 2160), (5120, 2880), (1936, 1216), (2448, 2048), (4112, 3008), (3838, 2160), (4056, 3040)):
    n = max(4, int(530e6 / (w * h)))
    pct = lambda t: 5.0 * w * h * n / t / 1e6 / 80
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes); d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, 2); ctx.sync()
        for _ in range(4): ctx.time_device(d_src, d_dst, n, warmup=0, reps=40)
        t = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        name, g = ctx.variant_name, ctx.launch_geometry(n)
        ctx.autotune(d_src, d_dst, n)
        t2 = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        with pkg.Context(w, h, "gbrg", "BGRx") as nxt:          # another order of the same camera: same kernel, same plan
            t3 = sorted(nxt.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
            src3, name3, band3 = SRC[nxt.plan_source], nxt.variant_name, nxt.launch_geometry(n)["band"]
        print("%4dx%-4d x %4d  default %-18s band %5d %5.1f %%   measured %-18s band %5d %5.1f %%   next context: %-8s %-18s band %5d %5.1f %%" % (
            w, h, n, name, g["band"], pct(t), ctx.variant_name, ctx.launch_geometry(n)["band"], pct(t2), src3, name3, band3, pct(t3)),
            flush=True)
        ctx.device_free(d_src); ctx.device_free(d_dst)
