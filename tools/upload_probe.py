#!/usr/bin/env python3
"""Why was `hipupload` (asynchronous, deferred release) slower than `hipupload async=false` in the first pipeline
measurement?  Times the raw copies: blocking hipMemcpy, hipMemcpyAsync + wait per copy, hipMemcpyAsync four deep with an
event per copy (what the element does), from pinned memory (plain and NUMA-local) and from pageable memory.
Usage (GPU box): python tools/upload_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
L = pkg.lib()
NB = 3840 * 2160           # one 4K mosaic
N = 300
vp = ctypes.c_void_p
d = [L.mibayer_dev_alloc(0, NB) for _ in range(4)]
stream = L.mibayer_dev_stream_create(0)
events = [L.mibayer_dev_event_create(0) for _ in range(8)]


def rate(fn, label):
    fn(20)
    t0 = time.perf_counter()
    fn(N)
    el = time.perf_counter() - t0
    print("%-72s %7.1f copies/s  %6.2f GB/s  %6.1f us per copy" % (label, N / el, N * NB / el / 1e9, el / N * 1e6), flush=True)


for kind in ("pinned", "pinned_near", "pageable"):
    if kind == "pageable":
        host = [np.full(NB, 7, np.uint8) for _ in range(4)]
        hp = [h.ctypes.data for h in host]
    else:
        alloc = L.mibayer_host_alloc if kind == "pinned" else (lambda n: L.mibayer_host_alloc_near(0, n))
        hp = [alloc(NB) for _ in range(4)]
        for p in hp:
            ctypes.memset(p, 7, NB)

    def sync_copy(n):
        for i in range(n):
            L.mibayer_dev_upload(0, vp(d[i % 4]), vp(hp[i % 4]), NB)

    def async_wait_each(n):
        for i in range(n):
            L.mibayer_dev_upload_async(0, vp(d[i % 4]), vp(hp[i % 4]), NB, vp(stream))
            L.mibayer_dev_event_record(0, vp(events[0]), vp(stream))
            L.mibayer_dev_event_wait(0, vp(events[0]))

    def async_four_deep(n):
        pend = []
        for i in range(n):
            if len(pend) == 4:
                L.mibayer_dev_event_wait(0, vp(pend.pop(0)))
            L.mibayer_dev_upload_async(0, vp(d[i % 4]), vp(hp[i % 4]), NB, vp(stream))
            ev = events[i % 8]
            L.mibayer_dev_event_record(0, vp(ev), vp(stream))
            pend.append(ev)
        for ev in pend:
            L.mibayer_dev_event_wait(0, vp(ev))

    def async_four_deep_fresh_events(n):
        pend = []
        for i in range(n):
            if len(pend) == 4:
                ev = pend.pop(0)
                L.mibayer_dev_event_wait(0, vp(ev))
                L.mibayer_dev_event_destroy(0, vp(ev))
            L.mibayer_dev_upload_async(0, vp(d[i % 4]), vp(hp[i % 4]), NB, vp(stream))
            ev = L.mibayer_dev_event_create(0)
            L.mibayer_dev_event_record(0, vp(ev), vp(stream))
            pend.append(ev)
        for ev in pend:
            L.mibayer_dev_event_wait(0, vp(ev))
            L.mibayer_dev_event_destroy(0, vp(ev))

    rate(sync_copy, "%s: blocking hipMemcpy" % kind)
    rate(async_wait_each, "%s: hipMemcpyAsync + event wait per copy" % kind)
    rate(async_four_deep, "%s: hipMemcpyAsync four deep, recycled events" % kind)
    rate(async_four_deep_fresh_events, "%s: hipMemcpyAsync four deep, event created + destroyed per copy" % kind)
    if kind != "pageable":
        print("    NUMA node of the buffer: %d (device node %d)" % (L.mibayer_host_numa_node(vp(hp[0])), L.mibayer_device_numa_node(0)))
        for p in hp:
            L.mibayer_host_free(vp(p))
