#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written on the GPU box by tools/gpu_profile.sh) into the tracked evidence under
profiles/: the rocprofv3 --kernel-trace --stats table, the PMC traffic summary with the calibration applied, the
bench line and the box description; also writes profiles/traffic_latest.json, which bench.py quotes as
roofline.traffic.   Usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

W, H, N = 3840, 2160, 64
ALG_R, ALG_W = W * H * N, 4 * W * H * N


def counter_means(d, last_bayer=None):
    """Mean counter value per kernel.  With last_bayer=K only the LAST K bayer2rgb dispatches are kept: those
    are the timed steps of bench.py (earlier ones are mibayer_autotune candidates, the parity check and warm-up)."""
    f = glob.glob(os.path.join(src, d, "*counter_collection.csv"))
    if not f:
        return {}
    rows = list(csv.DictReader(open(f[0])))
    if last_bayer:
        bayer = [r for r in rows if "bayer2rgb" in r["Kernel_Name"]]
        bayer.sort(key=lambda r: int(r["Dispatch_Id"]))
        rows = bayer[-last_bayer:]
    agg = collections.defaultdict(list)
    for r in rows:
        name = "bayer2rgb timed steps: " + r["Kernel_Name"] if last_bayer else r["Kernel_Name"]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def pick(means, needle):
    for (k, c), v in means.items():
        if needle in k:
            return k, v
    return None, (0.0, 0)


lines = ["# profiles/%s — rocprofv3 evidence for the bench kernel" % tag, ""]
box = os.path.join(src, "box.txt")
if os.path.exists(box):
    shutil.copy(box, os.path.join(dst, "%s_box.txt" % tag))
bench = os.path.join(src, "bench.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, "%s_bench.json" % tag))
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    lines += ["## bench line (unprofiled run, same box)", "",
              "* value %.0f Mpix/s, ms_per_step %.4f, kernel_ms (HIP events) %.4f, achieved %.1f GB/s = %.1f %% of 8 TB/s"
              % (b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["achieved"],
                 100 * b["roofline"]["frac"]),
              "* plan: %s, autotune: %s" % (b["config"]["kernel_variant"], b["config"].get("autotune")), ""]
stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu --no-host-path`", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---:|---:|---:|---:|---:|"]
    for r in csv.DictReader(open(stats[0])):
        lines.append("| `%s` | %s | %.0f | %s | %s | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]),
                                                             r["MinNs"], r["MaxNs"], r["Percentage"]))
    lines.append("")
    trace = glob.glob(os.path.join(src, "stats", "*kernel_trace.csv"))
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "bayer2rgb" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        timed = rows[-100:]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
        lines.append("Timed region only (the last 100 bayer2rgb dispatches of the trace = the 100 timed steps, kernel `%s`): "
                     "**avg %.0f ns**, min %d, max %d." % (timed[-1]["Kernel_Name"][:60], sum(d) / len(d), min(d), max(d)))
        lines.append("")
    lines.append("(calls include the autotune launches of `mibayer_autotune`, which try both tile shapes and both XCD maps; "
                 "profiled runs clock ~2 % lower than unprofiled ones, MI355X_MICROARCH.md \"DVFS\")")
    lines.append("")

fetch_b, write_b = counter_means("pmc_bench_FETCH_SIZE", 8), counter_means("pmc_bench_WRITE_SIZE", 8)
fetch_p, write_p = counter_means("pmc_probe_FETCH_SIZE"), counter_means("pmc_probe_WRITE_SIZE")
if fetch_b and write_b:
    lines += ["## HBM-side traffic (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, counters in KiB)", "",
              "Calibration on kernels with known byte counts (tools/hbm_probe.hip, 2 123 366 400 B streams, same passes):", "",
              "| probe kernel | true bytes | counter x 1024 | ratio |", "|---|---:|---:|---:|"]
    cal_f = cal_w = None
    for needle, true, means, name in (("k_read", 4 * ALG_R, fetch_p, "FETCH_SIZE"),
                                      ("k_mix14<true>", ALG_R, fetch_p, "FETCH_SIZE"),
                                      ("k_fill<true>", 4 * ALG_R, write_p, "WRITE_SIZE"),
                                      ("k_mix14<true>", 4 * ALG_R, write_p, "WRITE_SIZE")):
        k, (m, n) = pick(means, needle)
        if k:
            ratio = m * 1024 / true
            lines.append("| `%s` %s | %d | %.0f | %.4f |" % (needle, name, true, m * 1024, ratio))
            if name == "FETCH_SIZE":
                cal_f = ratio
            else:
                cal_w = ratio
    lines += ["", "=> FETCH_SIZE reports exactly 1/2 of the bytes fetched (16 B/lane and 4 B/lane loads alike; the gfx950 "
                  "correction of MI355X_MICROARCH.md \"HBM\": double it); WRITE_SIZE is exact.", ""]
    kf, (mf, nf) = pick(fetch_b, "bayer2rgb")
    kw, (mw, nw) = pick(write_b, "bayer2rgb")
    rd, wr = mf * 1024 * 2, mw * 1024
    lines += ["| bench kernel `%s` | per launch | algorithmic | ratio |" % kf[:70], "|---|---:|---:|---:|",
              "| read  (FETCH_SIZE x 1024 x 2, n=%d) | %.0f | %d | %.4f |" % (nf, rd, ALG_R, rd / ALG_R),
              "| write (WRITE_SIZE x 1024, n=%d) | %.0f | %d | %.4f |" % (nw, wr, ALG_W, wr / ALG_W),
              "| total | %.0f | %d | %.5f |" % (rd + wr, ALG_R + ALG_W, (rd + wr) / (ALG_R + ALG_W)), ""]
    with open(os.path.join(dst, "traffic_latest.json"), "w") as f:
        json.dump({"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
                   "algorithmic_bytes_per_launch": ALG_R + ALG_W,
                   "source": "profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + --pmc "
                             "WRITE_SIZE, separate passes, mean over the bench launches)" % tag}, f, indent=1)
    for d in ("pmc_bench_FETCH_SIZE", "pmc_bench_WRITE_SIZE"):
        f = glob.glob(os.path.join(src, d, "*counter_collection.csv"))
        if f:
            rows = [r for r in csv.DictReader(open(f[0])) if "bayer2rgb" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            rows = rows[-8:]
            with open(os.path.join(dst, "%s_%s.csv" % (tag, d)), "w", newline="") as out:
                w = csv.DictWriter(out, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size",
                                                    "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name",
                                                    "Counter_Value", "Start_Timestamp", "End_Timestamp"],
                                   extrasaction="ignore")
                w.writeheader()
                w.writerows(rows)
with open(os.path.join(dst, "%s_summary.md" % tag), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
