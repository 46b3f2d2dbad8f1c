#!/usr/bin/env python3
"""The launch the elements actually issue: ONE frame per launch (hipbayer2rgb batch=1, every
mibayer_process_device(..., nframes=1) caller).  64 device-resident 4K frames, each its own allocation.

  python tools/single_frame_bench.py                 table: shape x block order x number of frame queues the frames
                                                     are dealt over (1 = the context's stream, every launch behind
                                                     the previous one; 2-4 = mibayer_ctx_frame_queue)
  python tools/single_frame_bench.py trace ARM       a few passes of one arm, for rocprofv3 --kernel-trace
                                                     (ARM = variant:band:queues, e.g. 1:d:1; band d = the plan's default)
  python tools/single_frame_bench.py gaps DIR        kernel durations and gaps out of a rocprofv3 kernel trace

Wall time per pass incl. launch issue (python ctypes caller); GB/s = 5 B/px algorithmic."""
import csv
import ctypes
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H = (int(v) for v in os.environ.get("SFB_GEOMETRY", "3840x2160").split("x"))
N = max(8, min(256, int(64 * 3840 * 2160 / (W * H))))
REPS = 20
INT32_MIN = -2 ** 31


def gaps(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + d)
    rows = [r for r in csv.DictReader(open(files[0])) if "bayer2rgb" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 3:]            # warm-up passes out
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
    pitch = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rows, rows[1:])]
    # gaps between passes (host sync) are not launch boundaries
    inner = [(g, p) for g, p in zip(gap, pitch) if p < 100000]
    gap, pitch = [g for g, _ in inner], [p for _, p in inner]

    def med(v):
        v = sorted(v)
        return v[len(v) // 2]

    def mean(v):
        return sum(v) / max(1, len(v))
    queues = sorted(set(r.get("Queue_Id", "?") for r in rows))
    print("kernel %s" % rows[0]["Kernel_Name"][:70])
    print("dispatches %d on queues %s" % (len(rows), ",".join(queues)))
    print("kernel duration   mean %7.0f ns  median %7.0f ns  min %7.0f ns" % (mean(dur), med(dur), min(dur)))
    print("end -> next start mean %7.0f ns  median %7.0f ns   (negative = the next kernel started before this one ended)"
          % (mean(gap), med(gap)))
    print("start -> start    mean %7.0f ns  median %7.0f ns   = %.1f %% of 8 TB/s at 5 B/px" % (
        mean(pitch), med(pitch), 5.0 * W * H / mean(pitch) / 80))


def main():
    import __graft_entry__ as entry
    pkg = entry.load_package()
    L = pkg.lib()
    names = [L.mibayer_variant_name(v).decode() for v in range(L.mibayer_variant_count())]
    with pkg.Context(W, H, "rggb", "BGRx") as ctx:
        srcs = [ctx.device_alloc(ctx.src_bytes) for _ in range(N)]
        dsts = [ctx.device_alloc(ctx.dst_bytes) for _ in range(N)]
        for p in srcs:
            ctx.fill_synthetic(p, 1, seed=2)
        ctx.sync()
        fq = ctx.frame_queues         # hardware queues of their own (mibayer_ctx_frame_queue); ctx.sync covers them

        def sync_all(nq):
            ctx.sync()

        def one_pass(nq):
            qs = ["ctx"] if nq == 1 else fq[:nq]
            for i, (s, d) in enumerate(zip(srcs, dsts)):
                ctx.process_device(s, d, 1, stream=qs[i % len(qs)])

        def timed(nq, reps=REPS):
            for _ in range(3):
                one_pass(nq)
            sync_all(nq)
            t0 = time.perf_counter()
            for _ in range(reps):
                one_pass(nq)
            sync_all(nq)
            return (time.perf_counter() - t0) / reps

        def arm(variant, band, nq):
            ctx.set_plan(variant, band, 0)
            return timed(nq)

        if len(sys.argv) > 2 and sys.argv[1] == "trace":
            v, b, nq = sys.argv[2].split(":")
            if v == "auto":
                v = ctx.get_plan_for(1)[0]
            ctx.set_plan(int(v), INT32_MIN if b == "d" else int(b), 0)
            t = timed(int(nq), reps=6)
            print("arm %s (%s): %.3f ms per pass of %d frames" % (sys.argv[2], ctx.variant_name, t * 1e3, N))
            return
        v0, b0, _, _ = ctx.get_plan_for(1)
        print("# %d device-resident %dx%d frames, ONE launch per frame, wall time per pass incl. launch issue, %d passes"
              % (N, W, H, REPS))
        print("# the context's default plan: %s band %s; grid per frame: %s" % (
            names[v0], "default" if b0 == INT32_MIN else b0, ctx.launch_geometry(1)))
        variants = [int(v) for v in os.environ.get("SFB_VARIANTS", "1,2,3").split(",")]
        queues = [int(v) for v in os.environ.get("SFB_QUEUES", "1,2,3,4").split(",")]
        bands = [INT32_MIN if v == "d" else int(v) for v in os.environ.get("SFB_BANDS", "d,0").split(",")]
        # store alignment of the generic arm (0 = unshifted; 64 / 128 = the sector-aligned arm): geometries off the grid only
        aligns = [int(v) for v in os.environ.get("SFB_ALIGNS", "0").split(",")]
        best = {}
        for rnd in range(2):
            for variant in variants:
                for band in bands:
                    for al in aligns:
                        ctx.set_plan(variant, band, al)
                        g = ctx.launch_geometry(1)
                        for nq in queues:
                            t = timed(nq)
                            key = (names[variant], "default" if band == INT32_MIN else band, al, nq)
                            best[key] = min(best.get(key, 1e9), t)
                            print("round %d  %-22s band %-7s align %3d  grid %5d  queues %d  %8.3f ms  %8.1f fps  %6.1f GB/s (%4.1f %% of 8 TB/s)" % (
                                rnd, names[variant], "default" if band == INT32_MIN else band, al, g["grid_blocks"], nq, t * 1e3,
                                N / t, 5.0 * N * W * H / t / 1e9, 5.0 * N * W * H / t / 1e9 / 80), flush=True)
        print("# best of two rounds, fastest first (queues: %s)" % queues)
        for key, t in sorted(best.items(), key=lambda kv: kv[1])[:12]:
            print("#   %-22s band %-7s align %3d queues %d  %5.1f %% of 8 TB/s" % (key + (5.0 * N * W * H / t / 1e9 / 80,)))
        # what the context would have run by default
        with pkg.Context(W, H, "rggb", "BGRx") as fresh:
            v, b, al, _ = fresh.get_plan_for(1)
            key = (names[v], "default" if b == INT32_MIN else b, al, queues[0])
            if key in best:
                print("#   DEFAULT plan of this geometry: %-22s band %-7s align %3d -> %5.1f %% of 8 TB/s" % (
                    key[:3] + (5.0 * N * W * H / best[key] / 1e9 / 80,)))
            else:
                print("#   DEFAULT plan of this geometry (not among the arms): %s" % (key,))
        for p in srcs + dsts:
            ctx.device_free(p)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "gaps":
        gaps(sys.argv[2])
    else:
        main()
