/* GstMiHipMemory / allocator / pool.  See the header. */
#include "gstmihipmemory.h"

#include "mibayer.h"

GST_DEBUG_CATEGORY_STATIC (gst_mi_hip_memory_debug);
#define GST_CAT_DEFAULT gst_mi_hip_memory_debug

/* ---- allocator ------------------------------------------------------------------ */

typedef struct
{
  GstAllocator parent;
} GstMiHipAllocator;

typedef struct
{
  GstAllocatorClass parent_class;
} GstMiHipAllocatorClass;

G_DEFINE_TYPE (GstMiHipAllocator, gst_mi_hip_allocator, GST_TYPE_ALLOCATOR);

static GstAllocator *
gst_mi_hip_allocator_obtain (void)
{
  static GstAllocator *singleton = NULL;
  static gsize once = 0;

  if (g_once_init_enter (&once)) {
    singleton = g_object_new (gst_mi_hip_allocator_get_type (), NULL);
    gst_object_ref_sink (singleton);
    /* process-lifetime singleton: not a leak for GST_TRACERS=leaks */
    GST_OBJECT_FLAG_SET (singleton, GST_OBJECT_FLAG_MAY_BE_LEAKED);
    g_once_init_leave (&once, 1);
  }
  return singleton;
}

static GstMemory *
gst_mi_hip_allocator_dummy_alloc (GstAllocator * allocator, gsize size,
    GstAllocationParams * params)
{
  /* a device ordinal is needed: use gst_mi_hip_memory_new () */
  g_return_val_if_reached (NULL);
}

static void
gst_mi_hip_allocator_free (GstAllocator * allocator, GstMemory * memory)
{
  GstMiHipMemory *m = (GstMiHipMemory *) memory;

  gst_mi_hip_memory_wait (m);   /* no GPU work may outlive the allocation */
  if (m->access_event)
    mibayer_dev_event_destroy (m->device, m->access_event);
  if (m->staging)
    mibayer_host_free (m->staging);
  mibayer_dev_free (m->device, m->d_ptr);
  g_mutex_clear (&m->lock);
  g_free (m);
}

static gpointer
gst_mi_hip_mem_map_full (GstMemory * memory, GstMapInfo * info, gsize maxsize)
{
  GstMiHipMemory *m = (GstMiHipMemory *) memory;
  gpointer ret = NULL;

  /* GPU work queued by an earlier user must have finished, unless the caller
   * orders its own stream after it (gst_mi_hip_memory_order_after) */
  if (!(info->flags & GST_MAP_HIP_ASYNC) && !gst_mi_hip_memory_wait (m))
    return NULL;
  if (info->flags & GST_MAP_HIP) {
    if (info->flags & GST_MAP_WRITE) {
      g_mutex_lock (&m->lock);
      m->device_defined = TRUE; /* GPU work is about to define the contents */
      g_mutex_unlock (&m->lock);
    }
    return m->d_ptr;            /* device access: the caller orders its own GPU work */
  }

  g_mutex_lock (&m->lock);
  if (m->staging == NULL)
    m->staging = mibayer_host_alloc (memory->maxsize);
  if (m->staging != NULL) {
    gboolean ok = TRUE;

    /* Bring the host mirror up to date unless another CPU map already did.  A
     * WRITE-only map needs it too: unmap uploads the WHOLE mirror, so a writer
     * that touches part of the buffer must find the rest of the frame in it --
     * unless nothing has ever defined the device contents (a fresh buffer). */
    if (m->cpu_maps == 0
        && ((info->flags & GST_MAP_READ) || m->device_defined))
      ok = mibayer_dev_download (m->device, m->staging, m->d_ptr,
          memory->maxsize) == MIBAYER_OK;
    if (ok) {
      m->cpu_maps++;
      if (info->flags & GST_MAP_WRITE)
        m->cpu_dirty = TRUE;
      ret = m->staging;
    }
  }
  g_mutex_unlock (&m->lock);
  if (ret == NULL)
    GST_ERROR ("cannot stage HIP memory for a CPU map: %s",
        mibayer_last_hip_error ());
  return ret;
}

static void
gst_mi_hip_mem_unmap_full (GstMemory * memory, GstMapInfo * info)
{
  GstMiHipMemory *m = (GstMiHipMemory *) memory;

  if (info->flags & GST_MAP_HIP)
    return;
  g_mutex_lock (&m->lock);
  if (--m->cpu_maps == 0 && m->cpu_dirty) {
    if (mibayer_dev_upload (m->device, m->d_ptr, m->staging,
            memory->maxsize) != MIBAYER_OK)
      GST_ERROR ("upload after CPU write failed: %s",
          mibayer_last_hip_error ());
    else
      m->device_defined = TRUE;
    m->cpu_dirty = FALSE;
  }
  g_mutex_unlock (&m->lock);
}

gboolean
gst_mi_hip_memory_wait (GstMiHipMemory * m)
{
  gboolean ok = TRUE;

  g_mutex_lock (&m->lock);
  if (m->access_pending) {
    ok = mibayer_dev_event_wait (m->device, m->access_event) == MIBAYER_OK;
    if (ok)
      m->access_pending = FALSE;
    else
      GST_ERROR ("waiting for queued GPU work failed: %s",
          mibayer_last_hip_error ());
  }
  g_mutex_unlock (&m->lock);
  return ok;
}

gboolean
gst_mi_hip_memory_order_after (GstMiHipMemory * m, gpointer hip_stream)
{
  gboolean ok = TRUE;

  g_mutex_lock (&m->lock);
  /* Work queued on the stream the last access was queued on runs after it anyway (a stream is in order): no
   * cross-queue wait to insert.  In a device-resident pipeline every element's context launches on the device's ONE
   * shared compute queue, so this is the common case, and it saves two runtime calls per frame (hipbayersrc !
   * hipbayer2rgb at 4K: 33.7 k -> see profiles/r05_gst_device_source.log). */
  if (m->access_pending && !(hip_stream != NULL && m->access_stream == hip_stream))
    ok = mibayer_dev_stream_wait_event (m->device, hip_stream,
        m->access_event) == MIBAYER_OK;
  g_mutex_unlock (&m->lock);
  return ok;
}

gboolean
gst_mi_hip_memory_mark_access (GstMiHipMemory * m, gpointer hip_stream)
{
  gboolean ok;

  g_mutex_lock (&m->lock);
  if (m->access_event == NULL)
    m->access_event = mibayer_dev_event_create (m->device);
  ok = m->access_event != NULL
      && mibayer_dev_event_record (m->device, m->access_event,
      hip_stream) == MIBAYER_OK;
  if (ok) {
    m->access_pending = TRUE;
    m->access_stream = hip_stream;
  }
  m->device_defined = TRUE;
  g_mutex_unlock (&m->lock);
  return ok;
}

static GstMemory *
gst_mi_hip_mem_share (GstMemory * mem, gssize offset, gssize size)
{
  return NULL;                  /* no sub-memories: callers fall back to a copy */
}

static void
gst_mi_hip_allocator_class_init (GstMiHipAllocatorClass * klass)
{
  GstAllocatorClass *allocator_class = GST_ALLOCATOR_CLASS (klass);

  allocator_class->alloc = gst_mi_hip_allocator_dummy_alloc;
  allocator_class->free = gst_mi_hip_allocator_free;
  GST_DEBUG_CATEGORY_INIT (gst_mi_hip_memory_debug, "mihipmemory", 0,
      "HIP device memory");
}

static void
gst_mi_hip_allocator_init (GstMiHipAllocator * self)
{
  GstAllocator *alloc = GST_ALLOCATOR_CAST (self);

  alloc->mem_type = GST_MI_HIP_MEMORY_TYPE;
  alloc->mem_map_full = gst_mi_hip_mem_map_full;
  alloc->mem_unmap_full = gst_mi_hip_mem_unmap_full;
  alloc->mem_share = gst_mi_hip_mem_share;
  GST_OBJECT_FLAG_SET (self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

gboolean
gst_is_mi_hip_memory (GstMemory * mem)
{
  return mem != NULL && mem->allocator != NULL
      && G_TYPE_CHECK_INSTANCE_TYPE (mem->allocator,
      gst_mi_hip_allocator_get_type ());
}

GstMemory *
gst_mi_hip_memory_new (gint device, gsize size)
{
  GstMiHipMemory *m;
  gpointer d_ptr = mibayer_dev_alloc (device, size);

  if (d_ptr == NULL) {
    GST_ERROR ("hipMalloc of %" G_GSIZE_FORMAT " bytes on device %d failed: %s",
        size, device, mibayer_last_hip_error ());
    return NULL;
  }
  m = g_new0 (GstMiHipMemory, 1);
  /* no sub-memories (mem_share returns NULL): say so, so that gst_memory_share
   * callers copy instead of failing */
  gst_memory_init (GST_MEMORY_CAST (m), GST_MEMORY_FLAG_NO_SHARE,
      gst_mi_hip_allocator_obtain (), NULL, size, 0, 0, size);
  m->d_ptr = d_ptr;
  m->device = device;
  g_mutex_init (&m->lock);
  return GST_MEMORY_CAST (m);
}

/* ---- pool ------------------------------------------------------------------------- */

typedef struct
{
  GstBufferPool parent;
  gint device;
  guint size;
} GstMiHipPool;

typedef struct
{
  GstBufferPoolClass parent_class;
} GstMiHipPoolClass;

G_DEFINE_TYPE (GstMiHipPool, gst_mi_hip_pool, GST_TYPE_BUFFER_POOL);

static gboolean
gst_mi_hip_pool_set_config (GstBufferPool * pool, GstStructure * config)
{
  GstMiHipPool *self = (GstMiHipPool *) pool;
  GstCaps *caps = NULL;
  guint size = 0, min = 0, max = 0;

  if (!gst_buffer_pool_config_get_params (config, &caps, &size, &min, &max)
      || size == 0)
    return FALSE;
  self->size = size;
  return GST_BUFFER_POOL_CLASS (gst_mi_hip_pool_parent_class)->set_config (pool,
      config);
}

static GstFlowReturn
gst_mi_hip_pool_alloc_buffer (GstBufferPool * pool, GstBuffer ** buffer,
    GstBufferPoolAcquireParams * params)
{
  GstMiHipPool *self = (GstMiHipPool *) pool;
  GstMemory *mem = gst_mi_hip_memory_new (self->device, self->size);

  if (mem == NULL)
    return GST_FLOW_ERROR;
  *buffer = gst_buffer_new ();
  gst_buffer_append_memory (*buffer, mem);
  return GST_FLOW_OK;
}

/* A buffer that comes back to the pool carries a frame nobody will look at again: the next user defines the
 * contents afresh.  Without this, every WRITE-only CPU map of a recycled buffer first downloaded the stale frame
 * (a whole frame over PCIe and a host sync) only to have it overwritten.  GPU work still queued on the memory
 * stays ordered: the "last access" event is not touched. */
static void
gst_mi_hip_pool_reset_buffer (GstBufferPool * pool, GstBuffer * buffer)
{
  guint i, n = gst_buffer_n_memory (buffer);

  for (i = 0; i < n; i++) {
    GstMemory *mem = gst_buffer_peek_memory (buffer, i);

    if (gst_is_mi_hip_memory (mem)) {
      GstMiHipMemory *m = (GstMiHipMemory *) mem;

      g_mutex_lock (&m->lock);
      if (m->cpu_maps == 0) {
        m->device_defined = FALSE;
        m->cpu_dirty = FALSE;
      }
      g_mutex_unlock (&m->lock);
    }
  }
  GST_BUFFER_POOL_CLASS (gst_mi_hip_pool_parent_class)->reset_buffer (pool,
      buffer);
}

static void
gst_mi_hip_pool_class_init (GstMiHipPoolClass * klass)
{
  GstBufferPoolClass *pool_class = GST_BUFFER_POOL_CLASS (klass);

  pool_class->set_config = gst_mi_hip_pool_set_config;
  pool_class->alloc_buffer = gst_mi_hip_pool_alloc_buffer;
  pool_class->reset_buffer = gst_mi_hip_pool_reset_buffer;
}

static void
gst_mi_hip_pool_init (GstMiHipPool * self)
{
  self->device = 0;
  self->size = 0;
}

GstBufferPool *
gst_mi_hip_pool_new (gint device)
{
  GstMiHipPool *pool = g_object_new (gst_mi_hip_pool_get_type (), NULL);

  gst_object_ref_sink (pool);
  pool->device = device;
  return GST_BUFFER_POOL_CAST (pool);
}
