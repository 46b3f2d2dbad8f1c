/* GstMiHostPool -- a GstBufferPool whose buffers wrap hipHostMalloc-pinned host
 * memory (mibayer_host_alloc), so that the element's H2D / D2H copies are true
 * asynchronous DMA.  The memory is ordinary CPU-addressable memory: any element
 * can map it like system memory.
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

typedef struct _GstMiHostPool GstMiHostPool;
typedef struct _GstMiHostPoolClass GstMiHostPoolClass;

struct _GstMiHostPool
{
  GstBufferPool parent;
  guint size;                   /* bytes per buffer, from the pool config */
};

struct _GstMiHostPoolClass
{
  GstBufferPoolClass parent_class;
};

GType gst_mi_host_pool_get_type (void);
GstBufferPool *gst_mi_host_pool_new (void);

G_END_DECLS
#endif
