#!/usr/bin/env python3
"""Development tool: default plan vs autotuned plan for common camera geometries (sector-aligned widths), device-resident
batches of ~530 Mpixel.  Usage (GPU box): python tools/common_geometries.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
print("# default plan (variant 0, no autotune) and autotuned plan, device-resident batches, common camera geometries")
for (w, h) in ((640, 480), (1280, 720), (1280, 960), (1600, 1200), (2048, 1536), (2560, 1440), (2592, 1944), (3264, 2448),
               (4000, 3000), (4096, 2160), (5120, 2880), (1936, 1216), (2448, 2048), (4112, 3008)):
    n = max(4, int(530e6 / (w * h)))
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes); d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, 2); ctx.sync()
        for _ in range(4): ctx.time_device(d_src, d_dst, n, warmup=0, reps=40)
        t = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        name, g = ctx.variant_name, ctx.launch_geometry(n)
        rep = ctx.autotune(d_src, d_dst, n)
        t2 = sorted(ctx.time_device(d_src, d_dst, n, warmup=2, reps=10) for _ in range(7))[3]
        print("%4dx%-4d x %4d  default %-18s band %5d  %.4f ms %5.1f %%   autotuned %-18s band %5d  %.4f ms %5.1f %%" % (
            w, h, n, name, g["band"], t, 5.0 * w * h * n / t / 1e6 / 80, ctx.variant_name, ctx.launch_geometry(n)["band"], t2, 5.0 * w * h * n / t2 / 1e6 / 80), flush=True)
        ctx.device_free(d_src); ctx.device_free(d_dst)
