#!/usr/bin/env python3
"""What hipbayer2rgb batch=N buys: 64 device-resident 4K frames, each its own allocation, converted
  (a) with one launch per frame (hipbayer2rgb batch=1),  (b) with list launches of 2 / 4 / 8 / 16 frames
  (mibayer_process_device_list),  (c) for reference, as one contiguous 64-frame batch (mibayer_process_device).
Wall time per pass incl. launch issue.   Usage (GPU box): python tools/list_launch_bench.py [inverse]
`inverse`: the sibling direction (rgb2bayer, MIBAYER_FLAG_RGB2BAYER), 4 B/px in, 1 B/px out."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
L = pkg.lib()
W, H, N, REPS = 3840, 2160, 64, 20
vp = ctypes.c_void_p
INVERSE = len(sys.argv) > 1 and sys.argv[1] == "inverse"
with (pkg.Context(W, H, "rggb", (1, 2, 3), flags=pkg.FLAG_RGB2BAYER) if INVERSE
      else pkg.Context(W, H, "rggb", "BGRx")) as ctx:
    srcs = [ctx.device_alloc(ctx.src_bytes) for _ in range(N)]
    dsts = [ctx.device_alloc(ctx.dst_bytes) for _ in range(N)]
    big_src, big_dst = ctx.device_alloc(N * ctx.src_bytes), ctx.device_alloc(N * ctx.dst_bytes)
    if not INVERSE:             # the content does not matter for the timing; the mosaic generator is bayer2rgb's
        for p in srcs:
            ctx.fill_synthetic(p, 1, seed=2)
        ctx.fill_synthetic(big_src, N, seed=2)
    ctx.sync()
    ev0, ev1 = L.mibayer_dev_event_create(0), L.mibayer_dev_event_create(0)

    def timed(fn):
        for _ in range(3):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(REPS):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / REPS

    def per_frame():
        for s, d in zip(srcs, dsts):
            ctx.process_device(s, d, 1)

    fq = ctx.frame_queues

    def per_frame_frame_queues():       # what hipbayer2rgb does by default (property overlap=true); ctx.sync covers them
        for i, (s, d) in enumerate(zip(srcs, dsts)):
            ctx.process_device(s, d, 1, stream=fq[i % len(fq)])

    def per_frame_two_queues():
        for i, (s, d) in enumerate(zip(srcs, dsts)):
            ctx.process_device(s, d, 1, stream=fq[i & 1])

    def per_frame_batch_plan():         # the rounds-unaware shape of rounds 1-4 (the batch-class plan on one frame)
        for s, d in zip(srcs, dsts):
            ctx.process_device(s, d, 1)

    def lists(k):
        def fn():
            for i in range(0, N, k):
                ctx.process_device_list(srcs[i:i + k], dsts[i:i + k])
        return fn

    def lists_fq(k):
        def fn():
            for j, i in enumerate(range(0, N, k)):
                ctx.process_device_list(srcs[i:i + k], dsts[i:i + k], stream=fq[j % len(fq)])
        return fn

    print("# direction: %s" % ("rgb2bayer (inverse)" if INVERSE else "bayer2rgb"))
    print("# 64 device-resident 4K frames per pass, wall time per pass incl. launch issue (python ctypes caller), %d passes" % REPS)
    rows = [("one launch per frame (batch=1), one queue", per_frame),
            ("one launch per frame, over two of the frame queues", per_frame_two_queues),
            ("one launch per frame, round-robin over the 4 frame queues", per_frame_frame_queues)]
    if not INVERSE:
        frame_plan, batch_plan = ctx.get_plan_for(1)[:3], ctx.get_plan_for(N)[:3]
        print("# frame-class plan %s, batch-class plan %s" % (frame_plan, batch_plan))
        if frame_plan != batch_plan:
            ctx.set_plan_for(1, *batch_plan)
            t = timed(per_frame_batch_plan)
            print("%-56s %8.3f ms  %8.1f fps  %9.1f Mpix/s  %6.1f GB/s (%4.1f %% of 8 TB/s)" % (
                "one launch per frame, rounds 1-4 shape (batch plan)", t * 1e3, N / t, N * W * H / t / 1e6,
                5.0 * N * W * H / t / 1e9, 5.0 * N * W * H / t / 1e9 / 80), flush=True)
            ctx.set_plan_for(1, *frame_plan)
    for label, fn in rows + \
                     [("list launches of %2d separately allocated frames" % k, lists(k)) for k in (2, 4, 8, 16)] + \
                     [("list launches of %2d frames, dealt over the 4 frame queues" % k, lists_fq(k)) for k in (2, 4, 8, 16)] + \
                     [("one launch over a contiguous 64-frame batch", lambda: ctx.process_device(big_src, big_dst, N))]:
        t = timed(fn)
        print("%-56s %8.3f ms  %8.1f fps  %9.1f Mpix/s  %6.1f GB/s (%4.1f %% of 8 TB/s)" % (
            label, t * 1e3, N / t, N * W * H / t / 1e6, 5.0 * N * W * H / t / 1e9, 5.0 * N * W * H / t / 1e9 / 80), flush=True)
