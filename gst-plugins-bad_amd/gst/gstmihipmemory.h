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
};

GType gst_mi_hip_allocator_get_type (void);
GType gst_mi_hip_pool_get_type (void);

gboolean gst_is_mi_hip_memory (GstMemory * mem);
/* one device allocation of `size` bytes wrapped in a GstMemory, or NULL */
GstMemory *gst_mi_hip_memory_new (gint device, gsize size);
/* buffers of one GstMiHipMemory each; size comes from the pool config */
GstBufferPool *gst_mi_hip_pool_new (gint device);

G_END_DECLS
#endif
