/* GstMiHostPool -- a GstBufferPool whose buffers wrap hipHostMalloc-pinned host
 * memory (mibayer_host_alloc), so that the element's H2D / D2H copies are true
 * asynchronous DMA.  The memory is ordinary CPU-addressable memory: any element
 * can map it like system memory.
 *
 * GstMiHostAllocator -- the same memory as a GstAllocator, proposed in the
 * ALLOCATION query next to the pool: an upstream element that builds a pool of
 * its own (or allocates buffer by buffer) around the proposed allocator gets
 * pinned memory too.  Both honour the prefix / padding / alignment of the
 * GstAllocationParams they are given.
 *
 * This file is compiled into BOTH plugins (`bayer` and `mihip`), and GStreamer
 * loads plugins RTLD_LOCAL: each copy registers its own GType, under its own
 * name (MI_HOST_POOL_TYPE_NAME, set per plugin by the Makefile) -- two
 * registrations of one name would make the second plugin's pool unusable.
 *
 * SURVEY.md section 8(f) rank 1.  Pattern the reference tree uses for the same
 * job: sys/nvcodec/gstcudabufferpool.c:56-207 (pool) and
 * sys/nvcodec/gstcudamemory.c:258-325 (CuMemAllocHost staging). */
#ifndef MI_GST_HOST_POOL_H
#define MI_GST_HOST_POOL_H

#include <gst/gst.h>

G_BEGIN_DECLS

#define GST_TYPE_MI_HOST_POOL (gst_mi_host_pool_get_type ())
#define GST_MI_HOST_POOL(obj) \
  (G_TYPE_CHECK_INSTANCE_CAST ((obj), GST_TYPE_MI_HOST_POOL, GstMiHostPool))
#define GST_IS_MI_HOST_POOL(obj) \
  (G_TYPE_CHECK_INSTANCE_TYPE ((obj), GST_TYPE_MI_HOST_POOL))

#define GST_MI_HOST_POOL_MAX_DEVICES 16
typedef struct _GstMiHostPool GstMiHostPool;
typedef struct _GstMiHostPoolClass GstMiHostPoolClass;

struct _GstMiHostPool
{
  GstBufferPool parent;
  guint size;                   /* bytes per buffer, from the pool config */
  /* HIP ordinals the buffers are placed next to (NUMA), in rotation: buffer k of the pool sits next to
   * devices[k % ndevices] -- the k-th frame of a `devices=` list goes to that GPU (mibayer_pool routes a frame to
   * the device next to its buffer when the rotation drifts).  ndevices == 0: no preference */
  gint devices[GST_MI_HOST_POOL_MAX_DEVICES];
  guint ndevices;
  volatile gint next;           /* buffers allocated so far */
};

struct _GstMiHostPoolClass
{
  GstBufferPoolClass parent_class;
};

GType gst_mi_host_pool_get_type (void);
/* `device`: the HIP ordinal that will read / write the buffers: they are pinned on
 * the NUMA node next to it (mibayer_host_alloc_near); -1 = no preference */
GstBufferPool *gst_mi_host_pool_new (gint device);
/* the same for a list of ordinals: buffer k next to devices[k % n] (per-GPU NUMA-local staging for a stream that
 * is sharded over the GPUs of a two-socket node; reference pattern of binding an element's memory to its device:
 * sys/nvcodec/gstcudabasetransform.c:301-329) */
GstBufferPool *gst_mi_host_pool_new_for_devices (const gint * devices, guint n);

/* a GstAllocator of pinned host memory near `device` (-1 = no preference); the
 * memories it returns are plain wrapped system memory (any element maps them),
 * released with mibayer_host_free when the last reference goes */
GType gst_mi_host_allocator_get_type (void);
GstAllocator *gst_mi_host_allocator_new (gint device);
GstAllocator *gst_mi_host_allocator_new_for_devices (const gint * devices, guint n);

G_END_DECLS
#endif
