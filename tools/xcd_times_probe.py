#!/usr/bin/env python3
"""Per-XCD progress of one launch under the chunk-per-XCD order vs band 1, on several destination allocations.
Needs the instrumented build tools/libmibayer_xcdtimes.so (csrc + one 8-byte store per workgroup: wall_clock64() and
HW_REG_XCC_ID into a per-block table; tools/xcdtimes.patch applied to a COPY of csrc/ and built like the Makefile builds
libmibayer.so; not part of the product): every workgroup records its START time, so per XCD
`last start - first start of the launch` is how long that XCD needed to get through its share of the grid.
The chunk order gives every XCD its own eighth of the batch and nothing rebalances: the launch lasts as long as the
slowest XCD.   Usage (GPU box): MIBAYER_LIB_PATH=$PWD/tools/libmibayer_xcdtimes.so python tools/xcd_times_probe.py"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
raw = ctypes.CDLL(pkg.LIB_PATH)
raw.mibayer_dbg_reset.restype = None
raw.mibayer_dbg_read.restype = None
hip = ctypes.CDLL("libamdhip64.so")
W, H, N = 3840, 2160, 64
PLANS = [("4x2/band1", 1, "1"), ("4x2/chunk", 1, "-1"), ("1x8/chunk", 3, "-1")]
ctxs = {}
for name, variant, band in PLANS:
    os.environ["MIBAYER_XCD_BAND"] = band
    ctxs[name] = pkg.Context(W, H, "rggb", "BGRx", variant=variant)
del os.environ["MIBAYER_XCD_BAND"]
c0 = ctxs[PLANS[0][0]]
d_src = c0.device_alloc(N * c0.src_bytes)
c0.fill_synthetic(d_src, N, seed=2)
bufs = [("hipMalloc#%d" % k, c0.device_alloc(N * c0.dst_bytes)) for k in range(5)]
p = ctypes.c_void_p()
if hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(N * c0.dst_bytes), 4) == 0:
    bufs.append(("contiguous", p.value))
c0.sync()
for _ in range(8):
    c0.time_device(d_src, bufs[0][1], N, warmup=0, reps=40)
import numpy as np
out = np.zeros(1 << 17, dtype=np.uint64)
out_p = out.ctypes.data_as(ctypes.c_void_p)
print("# 4K x 64; per XCD: microseconds from the launch's first workgroup start to that XCD's LAST workgroup start "
      "(median of 5 launches); wall_clock64 = 100 MHz")
for bname, d_dst in bufs:
    for name, _, _ in PLANS:
        c = ctxs[name]
        ms = c.time_device(d_src, d_dst, N, warmup=2, reps=8)
        per = [[] for _ in range(8)]
        counts = None
        for _ in range(5):
            raw.mibayer_dbg_reset()
            c.time_device(d_src, d_dst, N, warmup=0, reps=1)
            raw.mibayer_dbg_read(out_p)
            live = out[out != 0]
            xcc = (live & np.uint64(7)).astype(int)
            t = (live >> np.uint64(3)).astype(np.int64)
            t0 = t.min()
            for k in range(8):
                per[k].append((t[xcc == k].max() - t0) / 100.0)
            counts = [int((xcc == k).sum()) for k in range(8)]
        med = [statistics.median(v) for v in per]
        print("%-12s %-10s %.4f ms | last start per XCD (us): %s | spread %.1f us | wgs/XCD %s"
              % (bname, name, ms, " ".join("%6.1f" % m for m in med), max(med) - min(med),
                 "all %d" % counts[0] if len(set(counts)) == 1 else counts))
