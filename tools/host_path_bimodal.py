#!/usr/bin/env python3
"""Development tool: why does the host path (pinned ring, streams + events -- the element's default mode) run at
13.1 Gpix/s in most runs and at 5.7 in some (BENCH_r05.json, profiles/r04_bench_head.json; VERDICT r05 Weak 3)?

`worker` runs ONE arm in a fresh process: `frames` 4K frames through mibayer_submit / mibayer_wait with `inflight` in
flight, pinned buffers next to the GPU, and prints one JSON line: rate, per-frame completion intervals (p50 / p99 / max;
the time between consecutive frames leaving mibayer_wait, i.e. what a downstream element sees), per-frame latency
(submit -> wait returns), the library's own host statistics (polls, naps, wait wall / CPU time per frame).
The driver runs every arm `--reps` times in fresh processes, interleaved, and prints a table.

Arms: wait policy {auto (spin only when the frame waited for is alone, else naps 20..250 us), spin (always), nap (never
spin)} x launch mode {events, graph}.  Run on the GPU box:  python tools/host_path_bimodal.py --reps 6
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H = 3840, 2160


def worker(a):
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    L = pkg.lib()
    flags = pkg.FLAG_HIPGRAPH if a.mode == "graph" else 0
    with pkg.Context(W, H, "rggb", "BGRx", device=0, inflight=a.inflight, flags=flags) as ctx:
        if a.policy == "spin":
            ctx.set_wait_spin(10_000_000)
        elif a.policy == "nap":
            ctx.set_wait_spin(0)
        bufs = []
        for _ in range(a.inflight):
            ps = L.mibayer_host_alloc_near(0, ctx.src_bytes)
            pd = L.mibayer_host_alloc_near(0, ctx.dst_bytes)
            s = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), (ctx.src_bytes,))
            d = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_uint8)), (ctx.dst_bytes,))
            s[:] = 0x55
            bufs.append((ps, pd, s, d))
        # where everything lives: the card's NUMA node, the node of every pinned block, the CPU this thread runs on
        L.mibayer_host_numa_node.argtypes = [ctypes.c_void_p]
        placement = {"device_node": L.mibayer_device_numa_node(0),
                     "src_nodes": [L.mibayer_host_numa_node(b[0]) for b in bufs],
                     "dst_nodes": [L.mibayer_host_numa_node(b[1]) for b in bufs],
                     "cpu": ctypes.CDLL(None).sched_getcpu()}
        out = {}
        for phase, n in (("warm", 2 * a.inflight + a.warm), ("timed", a.frames)):
            if phase == "timed" and a.idle > 0:
                # the GPU sits idle, as it does in bench.py while the PMC child or the CPU baseline runs
                time.sleep(a.idle)
            if phase == "timed" and a.foreign:
                # ANOTHER process uses the GPU meanwhile (what bench.py's PMC child is): does this process pay for getting
                # its hardware queues back when it resumes?
                subprocess.run([sys.executable, "-c",
                                "import sys; sys.path.insert(0, %r)\n"
                                "import __graft_entry__ as g\n"
                                "p = g.load_package()\n"
                                "c = p.Context(3840, 2160, 'rggb', 'BGRx')\n"
                                "s = c.device_alloc(64 * c.src_bytes); d = c.device_alloc(64 * c.dst_bytes)\n"
                                "[c.time_device(s, d, 64, warmup=1, reps=200) for _ in range(%d)]\n" % (ROOT, a.foreign)],
                               timeout=120)
            before = ctx.host_stats()
            t_submit, t_done = {}, []
            submit_wall, wait_wall = [], []
            t0 = time.perf_counter()
            for i in range(n):
                if ctx.pending() == a.inflight:
                    tw = time.perf_counter()
                    tag = ctx.wait()
                    now = time.perf_counter()
                    wait_wall.append(now - tw)
                    t_done.append((now, now - t_submit[tag]))
                t_submit[i + 1] = time.perf_counter()
                ctx.submit(bufs[i % a.inflight][2], bufs[i % a.inflight][3], tag=i + 1)
                submit_wall.append(time.perf_counter() - t_submit[i + 1])
            while ctx.pending():
                tag = ctx.wait()
                now = time.perf_counter()
                t_done.append((now, now - t_submit[tag]))
            el = time.perf_counter() - t0
            after = ctx.host_stats()
            if phase != "timed":
                continue
            gaps = np.diff(np.array([t for t, _ in t_done])) * 1e6
            lat = np.array([l for _, l in t_done]) * 1e6
            out = {"mode": a.mode, "policy": a.policy, "inflight": a.inflight, "frames": n,
                   "mpix_s": round(W * H * n / el / 1e6, 1), "us_per_frame": round(el / n * 1e6, 1),
                   "gap_us": {"p50": round(float(np.percentile(gaps, 50)), 1), "p99": round(float(np.percentile(gaps, 99)), 1),
                              "max": round(float(gaps.max()), 1)},
                   "latency_us": {"p50": round(float(np.percentile(lat, 50)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                                  "max": round(float(lat.max()), 1)},
                   "polls_per_frame": round((after["polls"] - before["polls"]) / n, 1),
                   "naps_per_frame": round((after["naps"] - before["naps"]) / n, 2),
                   "wait_wall_us": round((after["wait_wall_ms"] - before["wait_wall_ms"]) * 1e3 / n, 1),
                   "wait_cpu_us": round((after["wait_cpu_ms"] - before["wait_cpu_ms"]) * 1e3 / n, 1),
                   "submit_cpu_us": round((after["submit_cpu_ms"] - before["submit_cpu_ms"]) * 1e3 / n, 1),
                   # the slow state, if it shows, as a time line: mean gap of each tenth of the run
                   "gap_by_decile_us": [round(float(x.mean()), 0) for x in np.array_split(gaps, 10)],
                   "idle_s": a.idle, "placement": placement,
                   "stalls_over_2ms": [round(float(g)) for g in gaps[gaps > 2000.0]][:20],
                   # which call a stall sits in: the longest mibayer_submit and the longest mibayer_wait of the run
                   "submit_wall_us": {"p50": round(float(np.percentile(submit_wall, 50)) * 1e6, 1),
                                      "max": round(max(submit_wall) * 1e6, 1), "argmax": int(np.argmax(submit_wall))},
                   "wait_wall_max_us": {"max": round(max(wait_wall) * 1e6, 1), "argmax": int(np.argmax(wait_wall))},
                   # the first 48 frames in groups of 6: a slow start after an idle period shows here
                   "first_gaps_us": [round(float(x.mean()), 0) for x in np.array_split(gaps[:48], 8)]}
        for ps, pd, _, _ in bufs:
            L.mibayer_host_free(ps)
            L.mibayer_host_free(pd)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--mode", default="events")
    ap.add_argument("--policy", default="auto")
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--warm", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--idle", type=float, default=0.0, help="seconds the GPU idles between the warm-up and the timed frames")
    ap.add_argument("--foreign", type=int, default=0,
                    help="N > 0: another process runs N x 200 batch launches on the GPU between the warm-up and the timed frames")
    ap.add_argument("--arms", default="events:auto,graph:auto,events:spin,events:nap,graph:spin")
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    rows = []
    for rep in range(a.reps):
        for arm in a.arms.split(","):
            mode, policy = arm.split(":")
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--mode", mode, "--policy", policy,
                                  "--inflight", str(a.inflight), "--frames", str(a.frames), "--warm", str(a.warm),
                                  "--idle", str(a.idle), "--foreign", str(a.foreign)],
                                 capture_output=True, text=True, timeout=300)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("arm %s rep %d failed: %s" % (arm, rep, (res.stdout + res.stderr)[-400:]))
                continue
            r = json.loads(line[-1])
            r["rep"] = rep
            rows.append(r)
            print("%-12s rep %d  %8.1f Mpix/s  %7.1f us/frame  gap p50 %7.1f p99 %7.1f max %8.1f  latency p50 %7.1f max %8.1f  "
                  "polls %7.1f naps %5.2f  wait wall %7.1f cpu %6.1f  deciles %s  first48 %s  placement %s  submit wall %s  longest wait %s  stalls>2ms %s"
                  % (arm, rep, r["mpix_s"], r["us_per_frame"], r["gap_us"]["p50"], r["gap_us"]["p99"], r["gap_us"]["max"],
                     r["latency_us"]["p50"], r["latency_us"]["max"], r["polls_per_frame"], r["naps_per_frame"],
                     r["wait_wall_us"], r["wait_cpu_us"], r["gap_by_decile_us"], r["first_gaps_us"], json.dumps(r["placement"]), json.dumps(r["submit_wall_us"]), json.dumps(r["wait_wall_max_us"]), r["stalls_over_2ms"]), flush=True)
    print("== by arm: min / median / max Mpix/s over %d fresh processes" % a.reps)
    for arm in a.arms.split(","):
        mode, policy = arm.split(":")
        v = sorted(r["mpix_s"] for r in rows if r["mode"] == mode and r["policy"] == policy)
        if v:
            print("%-12s %8.1f %8.1f %8.1f" % (arm, v[0], v[len(v) // 2], v[-1]))


if __name__ == "__main__":
    main()
