import os, sys, random, statistics
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, N = 3840, 2160, 64
ctxs = []
for band in ("1", "-1"):
    for sl in ("0", "6", "12", "18", "24", "32", "48", "64", "96"):
        os.environ["MIBAYER_XCD_BAND"] = band; os.environ["MIBAYER_START_SLEEP"] = sl
        ctxs.append(("band%s/sleep%s" % (band, sl), pkg.Context(W, H, "rggb", "BGRx", variant=1)))
del os.environ["MIBAYER_XCD_BAND"]; del os.environ["MIBAYER_START_SLEEP"]
c0 = ctxs[0][1]
d_src = c0.device_alloc(N * c0.src_bytes); d_dst = c0.device_alloc(N * c0.dst_bytes)
c0.fill_synthetic(d_src, N, seed=2); c0.sync()
for _ in range(8): c0.time_device(d_src, d_dst, N, warmup=0, reps=40)
rng = random.Random(9); times = {n: [] for n, _ in ctxs}
for r in range(7):
    order = list(ctxs); rng.shuffle(order)
    for n, c in order:
        t = c.time_device(d_src, d_dst, N, warmup=1, reps=8)
        if r: times[n].append(t)
alg = 5.0 * W * H * N
for n, _ in ctxs:
    m = statistics.median(times[n]); print("%-18s %.4f ms  %.1f %%" % (n, m, 100 * alg / (m * 1e-3) / 8e12))
