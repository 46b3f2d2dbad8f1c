#!/usr/bin/env python3
"""Is the per-process state of the chunk-per-XCD order (DESIGN.md "per-process state") a property of the QUEUE the
launches go through?  One process, one source / destination pair, eight contexts with PRIVATE queues
(MIBAYER_SHARED_QUEUES=0: each context creates its own hipStream, i.e. may land on another hardware queue) per block
order; every context is timed on the same buffers in shuffled rounds.  If the chunk order is fast through some
queues and slow through others, a measured queue choice would fix the state; if all eight agree, the state is not the
queue's.   Usage (GPU box): MIBAYER_SHARED_QUEUES=0 python tools/queue_state_probe.py [contexts]"""
import os
import random
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIBAYER_SHARED_QUEUES"] = "0"
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
W, H, N = 3840, 2160, 64
NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PLANS = [("band1", "1"), ("chunk", "-1")]
ctxs = []
for name, band in PLANS:
    os.environ["MIBAYER_XCD_BAND"] = band
    for q in range(NCTX):
        ctxs.append((name, q, pkg.Context(W, H, "rggb", "BGRx", variant=1)))
del os.environ["MIBAYER_XCD_BAND"]
c0 = ctxs[0][2]
d_src = c0.device_alloc(N * c0.src_bytes)
d_dst = c0.device_alloc(N * c0.dst_bytes)
c0.fill_synthetic(d_src, N, seed=2)
c0.sync()
for _ in range(8):
    c0.time_device(d_src, d_dst, N, warmup=0, reps=40)
rng = random.Random(5)
times = {(n, q): [] for n, q, _ in ctxs}
for r in range(7):
    order = list(ctxs)
    rng.shuffle(order)
    for n, q, c in order:
        t = c.time_device(d_src, d_dst, N, warmup=1, reps=8)
        if r:
            times[(n, q)].append(t)
alg = 5.0 * W * H * N
print("# pid %d: 4K x 64, lds_4x2, ms per launch by private queue (median of 6 shuffled rounds x 8 launches)" % os.getpid())
for name, _ in PLANS:
    row = [statistics.median(times[(name, q)]) for q in range(NCTX)]
    print("%-6s " % name + "  ".join("%.4f" % t for t in row)
          + "   spread %.1f %%  best %.1f %% of peak" % (100 * (max(row) / min(row) - 1), 100 * alg / (min(row) * 1e-3) / 8e12))
