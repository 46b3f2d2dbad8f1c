/* Internal seam between the frame-sharding pool (mibayer_pool.cpp: pure host
 * logic, no HIP call in it) and the per-device stream context
 * (mibayer_abi.hip).  The pool only ever talks to a device through the public
 * per-context ABI of include/mibayer.h plus the entry points below, so the
 * whole failover / ordering / helper-thread logic can be built against a test
 * double of the contexts and run on a machine without a GPU, under
 * AddressSanitizer and ThreadSanitizer (tests/check/mock_mibayer.c,
 * tests/test_pool_logic.py).  Exported, but not part of the public ABI. */
#ifndef MIBAYER_HOOKS_H
#define MIBAYER_HOOKS_H

#include "../../include/mibayer.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One whole frame, synchronously, through a spare device-side slot of the
 * context that is NOT part of its submit/wait ring: upload, kernel and download
 * are queued behind whatever the context's streams already hold, and the call
 * returns when `dst` is complete.  Safe to call from one other thread while the
 * owner of the context uses mibayer_submit / mibayer_wait (the ring state is not
 * touched; HIP streams are thread-safe); two concurrent calls on one context are
 * not allowed.  This is what re-does, on a surviving device, the frames of a
 * device that failed (the pool's helper threads drive their context's ordinary
 * submit / wait ring). */
int mibayer_internal_run_spare (mibayer_ctx *ctx, const uint8_t *src,
    uint8_t *dst);

/* 1 when `p` is ordinary pageable host memory (an asynchronous copy from / to it
 * blocks the calling thread while the runtime stages it), 0 for pinned /
 * registered / device memory. */
int mibayer_internal_is_pageable (const void *p);

/* Give the context host-path queues of its own instead of its device's shared
 * set (csrc/mibayer_abi.hip, DeviceQueues).  Shared queues keep PINNED copies of
 * several contexts back to back on the one DMA engine per direction; a context
 * whose frames are PAGEABLE is driven by a helper thread whose copies block
 * while the runtime stages them, and several such threads stage concurrently
 * only if each has its own queue (4 shards, pageable: 1483 fps on shared queues,
 * 1627 fps on private ones; pinned: 1587 vs 870 -- profiles/r02_pool_queues.log).
 * Frames already queued finish where they are; safe at any time. */
void mibayer_internal_private_queues (mibayer_ctx *ctx);

/* Quiesce of a context whose device is being dropped: waits -- with the context's
 * deadline -- for the frames it still has in flight and for what a half-failed
 * submit left on its queues.  MIBAYER_OK: nothing of the context can touch the
 * callers' buffers any more (its work completed, or the device reported an error
 * and its queues are dead).  MIBAYER_ERR_TIMEOUT: the device did not answer within
 * the deadline (or had not before -- a context that already ran into a deadline is
 * not waited for again, the call returns at once): whatever it had queued may still
 * run later, the buffers of its in-flight frames remain the device's. */
int mibayer_internal_abandon (mibayer_ctx *ctx);

/* 1 when nothing this context queued is outstanding as far as a wait deadline is
 * concerned: it never ran into one, or the device has since caught up with
 * everything the context had queued at that moment; 0 otherwise.  Never blocks. */
int mibayer_internal_settled (mibayer_ctx *ctx);

/* Drill: occupy the context's compute queue for `ms` milliseconds (1 .. 5000) with
 * a kernel that does nothing but wait -- what a wedged GPU looks like to the host,
 * except that it ends by itself.  Everything queued behind it on that queue
 * (frames of this context; of its neighbours too while the device's queues are
 * shared) completes only afterwards.  Tests and mibayer_pool_inject_stall(). */
int mibayer_internal_stall (mibayer_ctx *ctx, int ms);

#ifdef __cplusplus
}
#endif
#endif
