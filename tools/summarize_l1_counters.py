#!/usr/bin/env python3
"""gpurun_out/<tag>_l1/ (tools/generic_l1_counters.sh) -> profiles/<tag>_generic_l1.md: per-pixel counter values of the
generic geometry divided by those of its aligned neighbour, same plan on both.
Usage: python tools/summarize_l1_counters.py r04"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag + "_l1")


def means(d, last=4):
    out, name, dur = {}, None, None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "bayer2rgb" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last:]
        agg, durs = collections.defaultdict(lambda: collections.defaultdict(float)), {}
        for r in rows:
            did = int(r["Dispatch_Id"])
            if did in ids:
                agg[r["Counter_Name"]][did] += float(r["Counter_Value"])
                durs[did] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                name = r["Kernel_Name"]
        for k, v in agg.items():
            out[k] = sum(v.values()) / len(v)
        if durs:
            dur = sum(durs.values()) / len(durs)
    return name, out, dur


runs = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(src, "g*_p*"))):
    m = re.match(r"g(\d+)x(\d+)x(\d+)_(\w+)_p(\d+)$", os.path.basename(d))
    if not m:
        continue
    key = (int(m[1]), int(m[2]), int(m[3]), m[4])
    name, mv, dur = means(d)
    e = runs.setdefault(key, {"kernel": name, "counters": {}, "ns": []})
    e["counters"].update(mv)
    if name:
        e["kernel"] = name
    if dur:
        e["ns"].append(dur)

PAIRS = ((3838, 3840, 2160, 64), (4056, 4064, 3040, 32), (2590, 2592, 1942, 64))
lines = ["# profiles/%s_generic_l1 — counters between the wave and the L2, generic geometry vs its aligned neighbour" % tag, "",
         "`rocprofv3 --kernel-trace --pmc <counters>` in separate passes over `tools/run_geometry.py` (4 launches of a "
         "device-resident batch, mean).  `same` = shape 1024x8, nt stores, band 1 forced on BOTH geometries (the generic "
         "one runs the GENERIC arm of that kernel, the aligned one the 16-byte arm); `default` = the generic geometry's "
         "own default plan.  Values are PER PIXEL; the ratio columns divide by the aligned neighbour's per-pixel value. "
         "Kernel time (ns per launch, under the profiler, i.e. slower than unprofiled) is the mean over the passes.  A "
         "TA_* pass (TA_TA_BUSY, TA_*_STALLED_BY_TC, TA_FLAT_WRITE_WAVEFRONTS) aborts rocprofv3 7.2 on gfx950 (signal 6 at "
         "finalisation) and is missing; SQ_ACTIVE_INST_VMEM reads 0 on this image.", "",
         "**Reading.** `TCP_TCC_WRITE_REQ` is exactly 1 request per 64 bytes for the aligned geometry (0.0625 per pixel) and "
         "1.03-1.07 x that for the generic one: the 16-byte lane stores that straddle a 16-byte boundary (width % 4 == 2) "
         "are NOT taken apart into two requests each -- the texture path coalesces a wave-store into 64-byte segments "
         "either way, and misalignment only adds the one partial segment at the end of each 1 KiB wave-store (17/16 = "
         "1.0625).  The L2 sees 3-6 % more requests, the fabric 0-3 % more writes; VALU instructions are 5 % up (the "
         "generic arm's per-lane guards).  Nothing between the wave and the L2 is 1.5 x or 2 x: there is no request "
         "amplification for an LDS transposition of the finished pixels to remove (VERDICT r03 next #3: item closed, "
         "DESIGN.md section 5).", ""]
counters = []
for e in runs.values():
    for c in e["counters"]:
        if c not in counters:
            counters.append(c)
for (wg, wa, h, n) in PAIRS:
    a = runs.get((wa, h, n, "same"))
    g = runs.get((wg, h, n, "same"))
    dflt = runs.get((wg, h, n, "default"))
    if not a or not g:
        continue
    pa, pg = wa * h * n, wg * h * n
    lines += ["## %dx%d vs %dx%d, %d frames" % (wg, h, wa, h, n), "",
              "kernels: aligned `%s`; generic-same `%s`; generic-default `%s`" % (
                  re.sub(r"^void mibayer::", "", a["kernel"] or "?"), re.sub(r"^void mibayer::", "", g["kernel"] or "?"),
                  re.sub(r"^void mibayer::", "", (dflt or {}).get("kernel") or "?")), "",
              "| counter | aligned / px | generic same-plan / px | ratio | generic default / px | ratio |",
              "|---|---:|---:|---:|---:|---:|"]

    def ns(e):
        return sum(e["ns"]) / len(e["ns"]) if e and e["ns"] else float("nan")
    rows = [("kernel ns (profiled)", ns(a) / pa, ns(g) / pg, ns(dflt) / pg if dflt else float("nan"))]
    for c in counters:
        va, vg = a["counters"].get(c), g["counters"].get(c)
        vd = dflt["counters"].get(c) if dflt else None
        if va is None or vg is None:
            continue
        rows.append((c, va / pa, vg / pg, vd / pg if vd is not None else float("nan")))
    for c, va, vg, vd in rows:
        lines.append("| %s | %.5g | %.5g | %.3f | %.5g | %.3f |" % (
            c, va, vg, vg / va if va else float("nan"), vd, vd / va if va else float("nan")))
    lines.append("")
open(os.path.join(ROOT, "profiles", "%s_generic_l1.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
