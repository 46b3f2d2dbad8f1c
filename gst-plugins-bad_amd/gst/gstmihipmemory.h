/* Device-resident GstMemory for MI355X: allocator, buffer pool and the
 * `memory:HIPMemory` caps feature.
 *
 * Same role as the reference tree's CUDA memory for its NVIDIA path:
 * sys/nvcodec/gstcudamemory.c:258-325 (allocator, host staging for CPU maps),
 * sys/nvcodec/gstcudabufferpool.c:56-207 (pool).  Written against HIP through
 * the context-free helpers of mibayer.h. */
#ifndef MI_GST_HIP_MEMORY_H
#define MI_GST_HIP_MEMORY_H

#include <gst/gst.h>

G_BEGIN_DECLS

#define GST_MI_HIP_MEMORY_TYPE "MiHIPMemory"
#define GST_CAPS_FEATURE_MEMORY_HIP "memory:HIPMemory"

/* map flag: gst_memory_map (mem, &info, GST_MAP_READ | GST_MAP_HIP) yields the
 * DEVICE pointer in info.data; without it the memory is staged through pinned
 * host memory (download on map for READ, upload on unmap after WRITE) */
#define GST_MAP_HIP (GST_MAP_FLAG_LAST << 1)
/* A memory carries a "last access" event (see gst_mi_hip_memory_mark_access).  Any map waits for it on the host, so a
 * consumer that knows nothing about it is always safe; a consumer that orders its own stream after the event with
 * gst_mi_hip_memory_order_after () adds this flag to GST_MAP_HIP to skip the host wait. */
#define GST_MAP_HIP_ASYNC (GST_MAP_FLAG_LAST << 2)

typedef struct _GstMiHipMemory GstMiHipMemory;

struct _GstMiHipMemory
{
  GstMemory mem;
  gpointer d_ptr;               /* device pointer */
  gint device;                  /* HIP ordinal */
  gpointer staging;             /* pinned host mirror, allocated on first CPU map */
  GMutex lock;
  gint cpu_maps;                /* outstanding CPU maps */
  gboolean cpu_dirty;           /* a CPU WRITE map is outstanding */
  gboolean device_defined;      /* something has written the device copy (upload, GPU work): a
                                   WRITE-only CPU map must start from it, not from a stale mirror */
  gpointer access_event;        /* HIP event of the last GPU access queued on this memory, created on first use */
  gboolean access_pending;      /* that event has not been waited for on the host yet */
  gpointer access_stream;       /* the stream that event was recorded on: a user on the SAME stream is ordered after
                                   the access by the stream itself and needs no wait (round 5) */
};

GType gst_mi_hip_allocator_get_type (void);
GType gst_mi_hip_pool_get_type (void);

gboolean gst_is_mi_hip_memory (GstMemory * mem);
/* one device allocation of `size` bytes wrapped in a GstMemory, or NULL */
GstMemory *gst_mi_hip_memory_new (gint device, gsize size);
/* buffers of one GstMiHipMemory each; size comes from the pool config */
GstBufferPool *gst_mi_hip_pool_new (gint device);

/* Stream-ordered hand-over.  A user that queues GPU work touching `mem` on `hip_stream` and does not wait for it:
 *   gst_mi_hip_memory_order_after (mem, stream);     before queueing: the work starts after the last queued access
 *   ... queue the work on stream ...
 *   gst_mi_hip_memory_mark_access (mem, stream);     after queueing: later users will be ordered after this work
 * mark_access returns FALSE if the event could not be recorded; the caller must then synchronise the stream itself. */
gboolean gst_mi_hip_memory_order_after (GstMiHipMemory * mem, gpointer hip_stream);
gboolean gst_mi_hip_memory_mark_access (GstMiHipMemory * mem, gpointer hip_stream);
/* host waits until the last queued access has completed */
gboolean gst_mi_hip_memory_wait (GstMiHipMemory * mem);

G_END_DECLS
#endif
