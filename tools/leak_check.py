#!/usr/bin/env python3
"""Development tool: do contexts, pools, rings, graphs and events give their device and pinned memory back?
Creates / uses / destroys them many times and compares hipMemGetInfo before and after.  Run on the GPU box."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package(lab=True)      # the wedge drill (Context.stall) is part of the pool's seam: lab build only
L = pkg.lib()
hip = ctypes.CDLL("libamdhip64.so.7")       # already loaded by libmibayer.so: same runtime


def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def rss_kb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1])


rng = np.random.default_rng(1)
w, h = 1920, 1080
src = rng.integers(0, 256, (h, w), dtype=np.uint8)


def one_round(i):
    flags = (0, pkg.FLAG_HIPGRAPH, pkg.FLAG_RGB2BAYER)[i % 3]
    if flags == pkg.FLAG_RGB2BAYER:
        with pkg.Context(w, h, "rggb", "ARGB", inflight=3, flags=flags) as ctx:
            ctx.process_host(np.zeros((h, 4 * w), np.uint8))
        return
    with pkg.Context(w, h, "rggb", "BGRx", inflight=3, flags=flags) as ctx:
        ctx.process_host(src)
        outs = [np.empty((h, 4 * w), np.uint8) for _ in range(3)]
        for k, o in enumerate(outs):
            ctx.submit(src, o, tag=k + 1)
        while ctx.pending():
            ctx.wait()
        d = ctx.device_alloc(ctx.dst_bytes)
        ctx.device_free(d)
    with pkg.Pool([0, 0], w, h, "bggr", "RGBx", inflight=2) as pool:
        o = np.empty((h, 4 * w), np.uint8)
        pool.submit(src, o, tag=1)
        pool.wait()
    # round 2: failover (a shard dies and its frames are redone), helper threads (pageable buffers), private
    # queues, NUMA-local pinned memory, list launches, copy queue + asynchronous upload
    with pkg.Pool([0, 0, 0], w, h, "bggr", "RGBx", inflight=2) as pool:
        pool.inject_fault(1, 1)
        outs = [np.empty((h, 4 * w), np.uint8) for _ in range(6)]
        for k, o in enumerate(outs):
            if pool.pending() >= L.mibayer_pool_capacity(pool._h):
                pool.wait()
            pool.submit(src, o, tag=k + 1)
        while pool.pending():
            pool.wait()
        pool.take_failure()
    pn = L.mibayer_host_alloc_near(0, 4 * w * h)
    L.mibayer_host_free(pn)
    with pkg.Context(w, h, "rggb", "BGRx") as ctx:
        ds = [ctx.device_alloc(ctx.src_bytes) for _ in range(3)]
        dd = [ctx.device_alloc(ctx.dst_bytes) for _ in range(3)]
        ctx.process_device_list(ds, dd)
        ctx.sync()
        st = L.mibayer_dev_stream_create(0)
        e2 = L.mibayer_dev_event_create(0)
        L.mibayer_dev_upload_async(0, ctypes.c_void_p(ds[0]), ctypes.c_void_p(src.ctypes.data), src.nbytes, ctypes.c_void_p(st))
        L.mibayer_dev_event_record(0, ctypes.c_void_p(e2), ctypes.c_void_p(st))
        L.mibayer_dev_event_wait(0, ctypes.c_void_p(e2))
        L.mibayer_dev_event_destroy(0, ctypes.c_void_p(e2))
        L.mibayer_dev_stream_destroy(0, ctypes.c_void_p(st))
        for q in ds + dd:
            ctx.device_free(q)
    ev = L.mibayer_dev_event_create(0)
    L.mibayer_dev_event_record(0, ev, None)
    L.mibayer_dev_event_wait(0, ev)
    L.mibayer_dev_event_destroy(0, ev)
    p = L.mibayer_host_alloc(1 << 20)
    L.mibayer_host_free(p)


for i in range(6):
    one_round(i)                # warm-up: runtime pools, code objects
f0, r0 = free_bytes(), rss_kb()
N = 150
trail = []
for i in range(N):
    one_round(i)
    if (i + 1) % (N // 5) == 0:         # a leak grows with the rounds; a pool the runtime fills once does not
        trail.append((i + 1, free_bytes() - f0, rss_kb() - r0))
f1, r1 = free_bytes(), rss_kb()
print("device memory free: before %d B, after %d rounds %d B, delta %+d B" % (f0, N, f1, f1 - f0))
print("host RSS: before %d kB, after %d kB, delta %+d kB" % (r0, r1, r1 - r0))
print("trail (rounds: device delta B, RSS delta kB): " + "  ".join("%d: %+d, %+d" % t for t in trail))


# round 4: the wedge registry.  A context that runs into its wait deadline and is destroyed while the device has not
# caught up hands its ring (device frames, events, queues) to the registry, and pinned blocks freed meanwhile are parked;
# once the stall is over everything must have gone back to the runtime -- by polling alone.
def wedge_round():
    import time
    ctx = pkg.Context(w, h, "rggb", "BGRx", inflight=2)
    ctx.process_host(src)
    ctx.set_wait_timeout(40)
    ctx.stall(250)
    ps, pd = L.mibayer_host_alloc(w * h), L.mibayer_host_alloc(4 * w * h)
    hs = np.ctypeslib.as_array(ctypes.cast(ps, ctypes.POINTER(ctypes.c_uint8)), (h, w))
    hd = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_uint8)), (h, 4 * w))
    hs[...] = src
    ctx.submit(hs, hd, 1)
    try:
        ctx.wait()
        raise SystemExit("the stalled frame came back before the deadline?")
    except pkg.MibayerError as exc:
        assert exc.status == pkg.ERR_TIMEOUT
    ctx.close()                                 # orphaned: the registry owns the ring now
    extra = L.mibayer_host_alloc(1 << 20)
    L.mibayer_host_free(extra)                  # parked
    assert L.mibayer_wedged_contexts() == 1 and L.mibayer_deferred_frees() == 1
    time.sleep(0.35)
    assert L.mibayer_wedged_contexts() == 0 and L.mibayer_deferred_frees() == 0
    L.mibayer_host_free(ps)
    L.mibayer_host_free(pd)


for i in range(3):
    wedge_round()
f2 = free_bytes()
M = 25
for i in range(M):
    wedge_round()
with pkg.Context(w, h, "rggb", "BGRx") as ctx:      # (the last context of the device trims the frame cache)
    ctx.process_host(src)
f3 = free_bytes()
print("wedge registry: device memory free before %d B, after %d timed-out-and-destroyed contexts %d B, delta %+d B" % (
    f2, M, f3, f3 - f2))
