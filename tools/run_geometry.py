#!/usr/bin/env python3
"""Development tool: REPS device-resident launches of the context's DEFAULT plan for one geometry (no autotune), for
counter passes under rocprofv3 (tools/geometry_counters.sh).  The plan is whatever mibayer_create picks under the
current environment (MIBAYER_ALIGN_STORES, MIBAYER_XCD_BAND ...), or --variant NAME.
Usage: python tools/run_geometry.py W H N [--reps 8] [--variant NAME] [--pattern rggb] [--fmt BGRx] [--inverse]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("w", type=int)
ap.add_argument("h", type=int)
ap.add_argument("n", type=int)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--variant", default="auto")
ap.add_argument("--pattern", default="rggb")
ap.add_argument("--fmt", default="BGRx")
a = ap.parse_args()
pkg = entry.load_package(lab=True)    # the tuning knobs exist in the lab build only (make lab)
with pkg.Context(a.w, a.h, a.pattern, a.fmt, variant=pkg.variant_names().index(a.variant)) as ctx:
    d_src = ctx.device_alloc(a.n * ctx.src_bytes)
    d_dst = ctx.device_alloc(a.n * ctx.dst_bytes)
    ctx.fill_synthetic(d_src, a.n, 2)
    ctx.sync()
    t_warm = ctx.time_device(d_src, d_dst, a.n, warmup=0, reps=3)
    t = ctx.time_device(d_src, d_dst, a.n, warmup=0, reps=a.reps)
    g = ctx.launch_geometry(a.n)
    alg = 5.0 * a.w * a.h * a.n
    print("%dx%d x %d  %s band %d  align=%s  %.4f ms  %.1f GB/s  %.1f %% of 8 TB/s" % (
        a.w, a.h, a.n, ctx.variant_name, g["band"], os.environ.get("MIBAYER_ALIGN_STORES", "default"),
        t, alg / t / 1e6, alg / t / 1e6 / 80))
    ctx.device_free(d_src)
    ctx.device_free(d_dst)
