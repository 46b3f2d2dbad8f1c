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
    t = b["roofline"].get("traffic")
    if t:
        lines += ["## HBM-side traffic measured INSIDE that bench run (`roofline.traffic`)", "",
                  "bench.py re-launched itself after the timed region as a profiled child (`--traffic-pass`, the same batch "
                  "and plan, %d launches averaged) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
                  "(separate passes); FETCH_SIZE KiB x 2 (gfx950 correction), WRITE_SIZE KiB x 1:" % t["launches_averaged"], "",
                  "* kernel `%s`, plan %s" % (t["kernel"], t["plan"]),
                  "* read %d B (%.4f x), write %d B (%.4f x), total %d B = **%.4f x** the algorithmic 2 654 208 000 B; "
                  "the pass took %.1f s" % (t["read"], t["read_ratio"], t["write"], t["write_ratio"], t["total"], t["ratio"],
                                            t["seconds"]), ""]
    for k in ("control_plane", "barrier_ms", "value_kernel_only"):
        if k in b:
            lines.append("* %s: %s" % (k, b[k]))
    lines.append("")
stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-host-path --no-traffic "
              "--plan <config.plan of the line above>` (the driver's arguments, pinned to the plan the unprofiled line ran, so "
              "the dominant kernel of this table IS that line's kernel)", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---:|---:|---:|---:|---:|"]
    for r in csv.DictReader(open(stats[0])):
        lines.append("| `%s` | %s | %.0f | %s | %s | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]),
                                                             r["MinNs"], r["MaxNs"], r["Percentage"]))
    lines.append("")
    trace = glob.glob(os.path.join(src, "stats", "*kernel_trace.csv"))
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "bayer2rgb" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        timed = rows[-21:-1]        # the very last dispatch is the parity spot check after the timed region
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
        lines.append("Timed region only (the 20 bayer2rgb dispatches before the final parity launch = the 20 timed steps, kernel `%s`): "
                     "**avg %.0f ns**, min %d, max %d." % (timed[-1]["Kernel_Name"][:60], sum(d) / len(d), min(d), max(d)))
        lines.append("")
    lines.append("(calls include the ~150 ms time-based pre-warm and the warm-up steps; "
                 "profiled runs clock ~2 % lower than unprofiled ones, MI355X_MICROARCH.md \"DVFS\")")
    lines.append("")

# one --stats pass per production plan: whichever plan the driver's box measures, its kernel's average is on file
per_plan = sorted(glob.glob(os.path.join(src, "stats_lds_*")))
if per_plan:
    lines += ["## `rocprofv3 --kernel-trace --stats`, one pass per production plan (`bench.py --gpus 1 --steps 20 --warmup 5 "
              "--no-cpu --no-host-path --no-traffic --plan <plan>`)", "",
              "Every plan `mibayer_autotune` has returned for 4K x 64 on some box, each pinned: the driver's box measures its own "
              "plan, and the kernel behind its `roofline` is one of these rows.  Timed = the 20 dispatches of the timed region; "
              "GB/s = 2 654 208 000 B / avg; the bench line is the same process's own HIP-event figure.", "",
              "| plan | kernel | calls | avg ns (all calls) | timed avg ns | timed min | timed max | GB/s (timed) | frac of 8 TB/s | "
              "bench line kernel_ms | bench line frac |", "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for d in per_plan:
        st = glob.glob(os.path.join(d, "*kernel_stats.csv"))
        tr = glob.glob(os.path.join(d, "*kernel_trace.csv"))
        if not st or not tr:
            continue
        vname, band, align = os.path.basename(d)[len("stats_"):].rsplit("_", 2)
        plan = "%s:%s:%s" % (vname, band.replace("m", "-"), align)
        rows = [r for r in csv.DictReader(open(st[0])) if "bayer2rgb" in r["Name"]]
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        trace = [r for r in csv.DictReader(open(tr[0])) if "bayer2rgb" in r["Kernel_Name"]]
        trace.sort(key=lambda r: int(r["Start_Timestamp"]))
        timed = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace[-21:-1]]
        avg = sum(timed) / len(timed)
        try:
            bl = json.loads(open(os.path.join(d, "bench_line.json")).read().strip().splitlines()[-1])["roofline"]
            bl = ("%.4f" % bl["kernel_ms"], "%.4f" % bl["frac"])
        except Exception:
            bl = ("?", "?")
        lines.append("| `%s` | `%s` | %s | %.0f | %.0f | %d | %d | %.0f | %.4f | %s | %s |" % (
            plan, top["Name"].replace("void mibayer::", "").replace("(mibayer::KParams)", ""), top["Calls"],
            float(top["AverageNs"]), avg, min(timed), max(timed), (ALG_R + ALG_W) / avg, (ALG_R + ALG_W) / avg / 8000.0, bl[0], bl[1]))
        shutil.copy(st[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, os.path.basename(d)[len("stats_"):])))
    lines.append("")

fetch_p, write_p = counter_means("pmc_probe_FETCH_SIZE"), counter_means("pmc_probe_WRITE_SIZE")
plans = {}
for plan in ("band1", "chunk", "identity"):
    fb, wb = counter_means("pmc_bench_%s_FETCH_SIZE" % plan, 8), counter_means("pmc_bench_%s_WRITE_SIZE" % plan, 8)
    if fb and wb:
        plans[plan] = (fb, wb)
if plans:
    lines += ["## HBM-side traffic (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, counters in KiB)", "",
              "Calibration on kernels with known byte counts (tools/hbm_probe.hip, 2 123 366 400 B streams, same passes):", "",
              "| probe kernel | true bytes | counter x 1024 | ratio |", "|---|---:|---:|---:|"]
    for needle, true, means, name in (("k_read", 4 * ALG_R, fetch_p, "FETCH_SIZE"),
                                      ("k_mix14<true>", ALG_R, fetch_p, "FETCH_SIZE"),
                                      ("k_fill<true>", 4 * ALG_R, write_p, "WRITE_SIZE"),
                                      ("k_mix14<true>", 4 * ALG_R, write_p, "WRITE_SIZE")):
        k, (m, n) = pick(means, needle)
        if k:
            lines.append("| `%s` %s | %d | %.0f | %.4f |" % (needle, name, true, m * 1024, m * 1024 / true))
    lines += ["", "=> FETCH_SIZE reports exactly 1/2 of the bytes fetched (16 B/lane and 4 B/lane loads alike; the gfx950 "
                  "correction of MI355X_MICROARCH.md \"HBM\": double it); WRITE_SIZE is exact.", "",
              "Bench kernel, timed steps only, for the three block orders `mibayer_autotune` chooses between "
              "(pinned with `bench.py --plan lds_4x2_r4_dpp_nt:<band>:0`, product build):", "",
              "| block order | read = FETCH_SIZE x 1024 x 2 | write = WRITE_SIZE x 1024 | total per launch | / algorithmic 2 654 208 000 |",
              "|---|---:|---:|---:|---:|"]
    out = {}
    for plan, (fb, wb) in plans.items():
        kf, (mf, nf) = pick(fb, "bayer2rgb")
        kw, (mw, nw) = pick(wb, "bayer2rgb")
        rd, wr = mf * 1024 * 2, mw * 1024
        lines.append("| %s | %.0f (%.4fx) | %.0f (%.4fx) | %.0f | %.5f |" % (
            {"band1": "band 1: one full-width tile row per XCD at a time (+ start delay) -- the default",
             "chunk": "one chunk of the batch per XCD (+ start delay)", "identity": "identity"}[plan],
            rd, rd / ALG_R, wr, wr / ALG_W,
            rd + wr, (rd + wr) / (ALG_R + ALG_W)))
        out[plan] = {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr)}
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(os.path.join(src, "pmc_bench_%s_%s" % (plan, cname), "*counter_collection.csv"))
            if f:
                rows = [r for r in csv.DictReader(open(f[0])) if "bayer2rgb" in r["Kernel_Name"]]
                rows.sort(key=lambda r: int(r["Dispatch_Id"]))
                rows = rows[-8:]
                with open(os.path.join(dst, "%s_pmc_bench_%s_%s.csv" % (tag, plan, cname)), "w", newline="") as o:
                    w = csv.DictWriter(o, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size",
                                                      "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name",
                                                      "Counter_Value", "Start_Timestamp", "End_Timestamp"],
                                       extrasaction="ignore")
                    w.writeheader()
                    w.writerows(rows)
    lines += ["", "Halo lines that live in another XCD's L2 are fetched through the fabric again (the 256 MB Infinity Cache "
                  "absorbs them before HBM): all of them for the identity order, the rows above/below a tile row for "
                  "band 1, none for the chunk order.", ""]
    serial, build = None, None
    if os.path.exists(box):
        for ln in open(box):
            if "Serial Number:" in ln:
                serial = ln.split()[-1]
    bh = os.path.join(src, "build_hash.txt")
    if os.path.exists(bh):
        build = open(bh).read().strip()
    tl = os.path.join(dst, "traffic_latest.json")
    cur = {}
    if os.path.exists(tl):        # the per-geometry entries (tools/summarize_geometry_counters.py) live in the same file
        try:
            cur = json.load(open(tl))
        except ValueError:
            cur = {}
    cur.update({"plans": out, "algorithmic_bytes_per_launch": ALG_R + ALG_W, "box_serial": serial, "build": build,
                "source": "profiles/%s_summary.md (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + --pmc "
                          "WRITE_SIZE, separate passes, mean over the timed bench launches, per block order)" % tag})
    with open(tl, "w") as f:
        json.dump(cur, f, indent=1)
with open(os.path.join(dst, "%s_summary.md" % tag), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
