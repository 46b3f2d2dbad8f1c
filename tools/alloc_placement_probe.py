#!/usr/bin/env python3
"""Is the block-order spread (the chunk-per-XCD order runs at 75-84 % of peak "from box to box")
a property of the BOX or of where this process's buffers landed?  Round 2 saw the same box give 0.4329 ms and 0.4077 ms
for the chunk order in two consecutive processes, so: allocate several source / destination pairs in ONE process
(keeping all of them alive, so each pair sits on different physical pages), time every plan on every pair in
shuffled rounds, and additionally time the same pair at shifted destination offsets.
Usage (GPU box): python tools/alloc_placement_probe.py [pairs]"""
import os
import random
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
W, H, N = 3840, 2160, 64
NPAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# "frag": before the pairs are allocated, 96 x 256 MiB of junk are allocated and every other block is freed again, so
# the pairs are assembled from holes all over the 288 GB instead of one fresh run of physical pages
FRAG = len(sys.argv) > 2 and sys.argv[2] == "frag"
PLANS = [("4x2/band1", 1, "1"), ("4x2/chunk", 1, "-1"), ("1x8/chunk", 3, "-1"), ("1x8/identity", 3, "0")]
ctxs = {}
for name, variant, band in PLANS:
    os.environ["MIBAYER_XCD_BAND"] = band
    ctxs[name] = pkg.Context(W, H, "rggb", "BGRx", variant=variant)
del os.environ["MIBAYER_XCD_BAND"]
c0 = ctxs[PLANS[0][0]]
SHIFT = 64 << 20            # room to slide the destination inside its allocation
if FRAG:
    junk = [c0.device_alloc(256 << 20) for _ in range(96)]
    for j in junk[1::2]:
        c0.device_free(j)
    print("# fragmented first: 96 x 256 MiB allocated, every other block freed")
pairs = []
for k in range(NPAIRS):
    d_src = c0.device_alloc(N * c0.src_bytes)
    d_dst = c0.device_alloc(N * c0.dst_bytes + SHIFT)
    c0.fill_synthetic(d_src, N, seed=2)
    pairs.append((d_src, d_dst))
c0.sync()
for _ in range(8):          # clock up
    c0.time_device(pairs[0][0], pairs[0][1], N, warmup=0, reps=40)
print("# 4K x 64, ms per launch (median of 6 shuffled rounds x 8 launches); %% of 8 TB/s in brackets")
print("# virtual addresses: " + "  ".join("pair%d src %#x dst %#x" % (k, s, d) for k, (s, d) in enumerate(pairs)))
rng = random.Random(3)
cells = [(k, name, 0) for k in range(NPAIRS) for name, _, _ in PLANS]
cells += [(0, name, off) for name in ("4x2/chunk", "1x8/chunk", "4x2/band1")
          for off in (4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 33 << 20)]
times = {c: [] for c in cells}
for r in range(7):
    order = list(cells)
    rng.shuffle(order)
    for cell in order:
        k, name, off = cell
        t = ctxs[name].time_device(pairs[k][0], pairs[k][1] + off, N, warmup=1, reps=8)
        if r:
            times[cell].append(t)
alg = 5.0 * W * H * N
print("%-8s" % "pair" + "".join("%22s" % name for name, _, _ in PLANS))
for k in range(NPAIRS):
    row = "%-8d" % k
    for name, _, _ in PLANS:
        t = statistics.median(times[(k, name, 0)])
        row += "%14.4f (%4.1f%%)" % (t, alg / t / 1e6 / 80)
    print(row)
print("# pair 0, destination shifted inside its allocation")
for name in ("4x2/chunk", "1x8/chunk", "4x2/band1"):
    row = "%-14s" % name
    for off in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 33 << 20):
        t = statistics.median(times[(0, name, off)])
        row += "  +%-9s %.4f" % ("%dK" % (off >> 10) if off < (1 << 20) else "%dM" % (off >> 20), t)
    print(row)
