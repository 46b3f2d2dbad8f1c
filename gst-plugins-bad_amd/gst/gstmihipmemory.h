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
/* A memory remembers its queued GPU accesses (see gst_mi_hip_memory_mark_access).  Any map waits for them on the host,
 * so a consumer that knows nothing about them is always safe; a consumer that orders its own stream after them with
 * gst_mi_hip_memory_order_after () adds this flag to GST_MAP_HIP to skip the host wait. */
#define GST_MAP_HIP_ASYNC (GST_MAP_FLAG_LAST << 2)

typedef struct _GstMiHipMemory GstMiHipMemory;
typedef struct _GstMiHipTimeline GstMiHipTimeline;

/* How GPU accesses to a memory are ordered (round 6: no runtime call per access).
 *
 * Rounds 2-5 recorded one HIP event per memory per access -- two hipEventRecord per converted frame, 2 N per list
 * launch of N frames, each of them also a barrier packet in the hardware queue -- and that bookkeeping, not the
 * kernel, bounded a device-resident pipeline (17 % of HBM peak at 4K where the launch itself reaches 54 %).  A queue
 * is in order, so "this access has completed" is implied by "the queue has got past ANY point behind it": the fence
 * can be placed later, by whoever needs one, and most accesses never need one (the next user launches on the same
 * queue, or the buffer just goes back to its pool).
 *
 * A TIMELINE belongs to one HIP stream and counts the accesses marked on it (`submitted`, a host-side counter).  A
 * memory remembers (timeline, sequence number) of its last accesses -- marking is an atomic increment, no runtime
 * call.  Somebody who must be ordered after an access
 *   - on the same stream: nothing to do;
 *   - on another stream: records ONE event on the timeline's stream now (it covers every access marked so far, so the
 *     frames of a whole list launch, and both memories of a frame, share it) and makes its stream wait for it;
 *   - on the host: records an event now and waits for it; `completed` remembers how far the queue is known to have
 *     got, so the other memories of that launch do not wait again.
 * A fence recorded later than the access waits for whatever else was queued in between: a few more kernels of the
 * same pipeline, microseconds each.  Pattern in the reference tree: sys/nvcodec/gstcudamemory.c:258-325 keeps no
 * per-memory events at all and synchronises the whole stream on every map; this keeps its cost model for the common
 * case (no call) without its blocking. */
#define GST_MI_HIP_MEMORY_MAX_ACCESSES 4

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
  /* GPU accesses not known to have completed: at most one per stream (a later access on a stream subsumes the earlier
   * ones on it).  Several entries = several queues touched the memory (tee branches, a copy queue and a compute
   * queue): a user is ordered after ALL of them (one slot lost the first of two concurrent readers). */
  struct
  {
    GstMiHipTimeline *timeline; /* a reference */
    guint64 seq;
  } access[GST_MI_HIP_MEMORY_MAX_ACCESSES];
  guint n_access;
};

GType gst_mi_hip_allocator_get_type (void);
GType gst_mi_hip_pool_get_type (void);

gboolean gst_is_mi_hip_memory (GstMemory * mem);
/* one device allocation of `size` bytes wrapped in a GstMemory, or NULL */
GstMemory *gst_mi_hip_memory_new (gint device, gsize size);
/* buffers of one GstMiHipMemory each; size comes from the pool config */
GstBufferPool *gst_mi_hip_pool_new (gint device);

/* Stream-ordered hand-over.  A user that queues GPU work touching `mem` on `hip_stream` and does not wait for it:
 *   gst_mi_hip_memory_order_after (mem, stream);     before queueing: the work starts after every queued access
 *   ... queue the work on stream ...
 *   gst_mi_hip_memory_mark_access (mem, stream);     after queueing: later users will be ordered after this work
 * order_after returns FALSE if the ordering could not be queued (the caller then waits on the host:
 * gst_mi_hip_memory_wait); mark_access cannot fail (it makes no runtime call) and returns TRUE.
 * The _tl forms take the stream's timeline, for elements that launch on one stream frame after frame and look it up
 * once (gst_mi_hip_timeline_for); the stream forms look it up per call. */
gboolean gst_mi_hip_memory_order_after (GstMiHipMemory * mem, gpointer hip_stream);
gboolean gst_mi_hip_memory_mark_access (GstMiHipMemory * mem, gpointer hip_stream);
gboolean gst_mi_hip_memory_order_after_tl (GstMiHipMemory * mem, GstMiHipTimeline * tl);
void gst_mi_hip_memory_mark_access_tl (GstMiHipMemory * mem, GstMiHipTimeline * tl);
/* host waits until every queued access has completed */
gboolean gst_mi_hip_memory_wait (GstMiHipMemory * mem);

/* The timeline of a stream (a new reference; one per (device, stream) per process). */
GstMiHipTimeline *gst_mi_hip_timeline_for (gint device, gpointer hip_stream);
void gst_mi_hip_timeline_unref (GstMiHipTimeline * tl);
/* The caller has just waited for everything IT queued on the timeline's stream (mibayer_sync of its context) after
 * reading `upto` = gst_mi_hip_timeline_submitted (tl): accesses up to `upto` are complete.  An element calls this
 * before it gives up the context whose stream the timeline follows -- nothing it marked will ask that stream for a
 * fence afterwards. */
guint64 gst_mi_hip_timeline_submitted (GstMiHipTimeline * tl);
void gst_mi_hip_timeline_settle (GstMiHipTimeline * tl, guint64 upto);
/* Destroys a stream made by mibayer_dev_stream_create (synchronises it first) and retires its timeline: every access
 * marked on it is complete from then on.  Use instead of mibayer_dev_stream_destroy for streams accesses were marked on. */
void gst_mi_hip_stream_destroy (gint device, gpointer hip_stream);
/* diagnostics: events recorded for fences since the process started (what the per-access scheme paid per frame) */
guint64 gst_mi_hip_timeline_fences_recorded (void);

G_END_DECLS
#endif
