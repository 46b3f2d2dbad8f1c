#!/usr/bin/env python3
"""gpurun_out/<tag>_geom/ (tools/geometry_counters.sh) -> profiles/<tag>_geometry_counters.md and the per-geometry,
per-plan entries of profiles/traffic_latest.json.  FETCH_SIZE is doubled (the gfx950 correction of
MI355X_MICROARCH.md "HBM", re-checked with tools/hbm_probe.hip in profiles/r02_summary.md); WRITE_SIZE is exact.
Usage: python tools/summarize_geometry_counters.py r03"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag + "_geom")


def means(d, last=8):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None, {}
    rows = [r for r in csv.DictReader(open(f[0])) if "bayer2rgb" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last:]
    agg = collections.defaultdict(list)
    name = None
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    return name, {k: sum(v) / len(v) for k, v in agg.items()}


runs = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(src, "g*"))):
    m = re.match(r"g(\d+)x(\d+)x(\d+)_(.+?)_(FETCH_SIZE|WRITE_SIZE|TCC.*)$", os.path.basename(d))
    if not m:
        continue
    w, h, n, plan, counter = int(m[1]), int(m[2]), int(m[3]), m[4], m[5]
    name, mv = means(d)
    e = runs.setdefault((w, h, n, plan), {"kernel": name})
    e.update(mv)
    if name:
        e["kernel"] = name
lines = ["# profiles/%s — HBM-side traffic per geometry and plan" % tag, "",
         "`rocprofv3 --kernel-trace --pmc <counter>` in separate passes over `tools/run_geometry.py` (8 timed launches of "
         "the context's default plan, device-resident batch).  read = FETCH_SIZE x 1024 x 2 (gfx950 correction), write = "
         "WRITE_SIZE x 1024; algorithmic = 1 B read + 4 B written per pixel.", "",
         "| geometry | plan | kernel | read / alg | write / alg | total / alg | 64-B share of write requests |",
         "|---|---|---|---:|---:|---:|---:|"]
out = {}
for (w, h, n, plan), e in runs.items():
    ar, aw = w * h * n, 4 * w * h * n
    rd = e.get("FETCH_SIZE", 0) * 1024 * 2
    wr = e.get("WRITE_SIZE", 0) * 1024
    req, req64 = e.get("TCC_EA0_WRREQ_sum"), e.get("TCC_EA0_WRREQ_64B_sum")
    share = "%.3f" % (req64 / req) if req and req64 is not None else "n/a"
    kern = (e.get("kernel") or "?")
    kern = re.sub(r"^void mibayer::", "", kern)[:58]
    lines.append("| %dx%d x %d | %s | `%s` | %.4f | %.4f | %.4f | %s |" % (
        w, h, n, plan, kern, rd / ar, wr / aw, (rd + wr) / (ar + aw), share))
    out["%dx%dx%d/%s" % (w, h, n, plan)] = {
        "read_bytes": round(rd), "write_bytes": round(wr), "hbm_bytes_per_launch": round(rd + wr),
        "algorithmic_bytes_per_launch": ar + aw, "kernel": e.get("kernel")}
open(os.path.join(ROOT, "profiles", "%s_geometry_counters.md" % tag), "w").write("\n".join(lines) + "\n")
tl = os.path.join(ROOT, "profiles", "traffic_latest.json")
cur = json.load(open(tl)) if os.path.exists(tl) else {}
cur["by_geometry_and_plan"] = out
cur["by_geometry_source"] = "profiles/%s_geometry_counters.md" % tag
for key, name in (("by_geometry_box_serial", "box.txt"), ("by_geometry_build", "build_hash.txt")):
    try:
        cur[key] = open(os.path.join(src, name)).read().split()[-1]
    except (OSError, IndexError):
        cur[key] = None
json.dump(cur, open(tl, "w"), indent=1)
print("\n".join(lines))
