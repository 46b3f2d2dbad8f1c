#!/usr/bin/env python3
"""Development tool: synchronous host-path rate (mibayer_process_host, pinned buffers) for different numbers of
intra-frame bands (MIBAYER_HOST_BANDS), and the queued rate (3 frames in flight).  Run on the GPU box."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package(lab=True)    # the tuning knobs exist in the lab build only (make lab)
L = pkg.lib()


def pinned(n):
    p = L.mibayer_host_alloc(n)
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (n,))


for (w, h) in ((3840, 2160), (1920, 1080), (7680, 4320)):
    for bands in (1, 2, 4, 8):
        os.environ["MIBAYER_HOST_BANDS"] = str(bands)
        with pkg.Context(w, h, "rggb", "BGRx", inflight=3) as ctx:
            bufs = [(pinned(ctx.src_bytes), pinned(ctx.dst_bytes)) for _ in range(3)]
            for (ps, s), _ in bufs:
                s[:] = 0x55
            n = 300 if w < 7000 else 80
            for _ in range(10):
                ctx.process_host(bufs[0][0][1], bufs[0][1][1])
            t0 = time.perf_counter()
            for i in range(n):
                ctx.process_host(bufs[0][0][1], bufs[0][1][1])
            sync_fps = n / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            for i in range(n):
                if ctx.pending() == 3:
                    ctx.wait()
                ctx.submit(bufs[i % 3][0][1], bufs[i % 3][1][1], tag=i + 1)
            while ctx.pending():
                ctx.wait()
            q_fps = n / (time.perf_counter() - t0)
            print("%dx%d bands %d: synchronous %7.1f fps (%6.2f Gpix/s)   3 in flight %7.1f fps (%6.2f Gpix/s)"
                  % (w, h, bands, sync_fps, sync_fps * w * h / 1e9, q_fps, q_fps * w * h / 1e9))
            for (ps, _), (pd, _) in bufs:
                L.mibayer_host_free(ps)
                L.mibayer_host_free(pd)
