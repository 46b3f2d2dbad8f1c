#!/usr/bin/env python3
"""Development tool: every production plan -- 3 tile shapes x {band 1, chunk, identity} -- on sector-aligned frame
widths, interleaved rounds, through mibayer_set_plan (product build).  What the static default of mibayer_create
(resolve_variant + the band rules of plan_launch) should pick per width is read off this table.
Usage (GPU box): python tools/plan_sweep.py [json-out] [Mpixel per batch, default 265] [common]
(a 265-Mpixel batch keeps the SOURCE resident in the 256 MB Infinity Cache between launches -- what a streaming pipeline
sees; 530 Mpixel is the bench's regime, where the source comes from HBM every time)"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
names = pkg.variant_names()
SHAPES = [names.index(n) for n in ("lds_4x2_r4_dpp_nt", "lds_2x4_r4_dpp_nt", "lds_1x8_r4_dpp_nt")]
BANDS = (1, -1, 0)
widths = sorted(set(list(range(1088, 8192 + 1, 128)) + [
    1280, 1296, 1440, 1600, 1920, 1936, 2048, 2304, 2448, 2560, 2592, 2688, 2704, 3072, 3264, 3280, 3840, 4000, 4032,
    4064, 4096, 4112, 4208, 4608, 4656, 5120, 5472, 6000, 6144, 7680, 8192]))
MPIX = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 265e6
if len(sys.argv) > 3:
    widths = [1280, 1600, 1920, 1936, 2048, 2304, 2448, 2560, 2592, 2688, 3072, 3264, 3280, 3840, 4000, 4032, 4096, 4112, 4208,
              4608, 4656, 5120, 5472, 6000, 6144, 7680, 8192]
widths = [w for w in widths if w % 16 == 0]
out = {}
print("# width: default plan %%  |  best plan %%  |  per plan (shape/band): %% of 8 TB/s; ~%d Mpixel batches" % (MPIX / 1e6) + ", median of 3 "
      "interleaved rounds x 6 launches", flush=True)
for w in widths:
    h = 1080 if w <= 2600 else (2160 if w <= 5200 else 4320)
    n = max(4, int(MPIX / (w * h)))
    pct = lambda t: 5.0 * w * h * n / t / 1e6 / 80   # noqa: E731
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        d_src = ctx.device_alloc(n * ctx.src_bytes)
        d_dst = ctx.device_alloc(n * ctx.dst_bytes)
        ctx.fill_synthetic(d_src, n, 2)
        ctx.sync()
        dv, db, _ = ctx.get_plan()
        dgeo = ctx.launch_geometry(n)
        for _ in range(3):
            ctx.time_device(d_src, d_dst, n, warmup=0, reps=40)
        plans = [(v, b) for v in SHAPES for b in BANDS]
        t = {p: [] for p in plans}
        tdef = []
        for _ in range(3):
            ctx.set_plan(dv, db, 0)
            tdef.append(ctx.time_device(d_src, d_dst, n, warmup=1, reps=6))
            for (v, b) in plans:
                ctx.set_plan(v, b, 0)
                t[(v, b)].append(ctx.time_device(d_src, d_dst, n, warmup=1, reps=6))
        med = {p: sorted(ts)[1] for p, ts in t.items()}
        mdef = sorted(tdef)[1]
        best = min(med, key=med.get)
        row = {"height": h, "frames": n, "default": {"variant": names[dv], "band": dgeo["band"], "pct": round(pct(mdef), 2)},
               "best": {"variant": names[best[0]], "band": best[1], "pct": round(pct(med[best]), 2)},
               "plans": {"%s/%d" % (names[v].split("_")[1], b): round(pct(med[(v, b)]), 2) for (v, b) in plans}}
        out[str(w)] = row
        print("%5d x %4d x %3d  default %s/%-5d %5.1f | best %s/%-3d %5.1f | %s" % (
            w, h, n, names[dv].split("_")[1], dgeo["band"], pct(mdef), names[best[0]].split("_")[1], best[1], pct(med[best]),
            "  ".join("%s/%d %.1f" % (names[v].split("_")[1], b, pct(med[(v, b)])) for (v, b) in plans)), flush=True)
        ctx.device_free(d_src)
        ctx.device_free(d_dst)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
