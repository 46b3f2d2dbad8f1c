import os, sys, statistics
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
pkg = e.load_package()
for (w, h) in ((3840, 2160), (1920, 1080), (640, 480)):
    with pkg.Context(w, h, "rggb", "BGRx") as c:
        d_src = c.device_alloc(c.src_bytes); d_dst = c.device_alloc(c.dst_bytes)
        c.fill_synthetic(d_src, 1, 2); c.sync()
        for _ in range(3): c.time_device(d_src, d_dst, 1, warmup=0, reps=2000)
        ts = sorted(c.time_device(d_src, d_dst, 1, warmup=10, reps=2000) for _ in range(9))
        print("%s %dx%d single-frame launches back to back: median %.3f us  min %.3f us" % (os.path.basename(pkg.LIB_PATH), w, h, ts[4] * 1e3, ts[0] * 1e3))
