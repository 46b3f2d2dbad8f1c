/* GstMiHipMemory / allocator / pool.  See the header. */
#include "gstmihipmemory.h"

#include <string.h>

#include "mibayer.h"

GST_DEBUG_CATEGORY_STATIC (gst_mi_hip_memory_debug);
#define GST_CAT_DEFAULT gst_mi_hip_memory_debug

/* ---- timelines: lazy fences (see the header) --------------------------------------- */

#define TIMELINE_SPARE_EVENTS 4

struct _GstMiHipTimeline
{
  gint refcount;
  gint device;
  gpointer stream;
  gboolean retired;             /* the stream is gone (gst_mi_hip_stream_destroy): everything has completed */
  guint64 submitted;            /* accesses marked so far (atomic) */
  guint64 completed;            /* every access up to here is known to have completed (atomic, monotonic) */
  GMutex lock;                  /* the fields below, and `stream` going away */
  gpointer fence;               /* the event other streams are made to wait for ... */
  guint64 fence_seq;            /* ... recorded when `submitted` was this: it covers every access up to it */
  gpointer spare[TIMELINE_SPARE_EVENTS];        /* events of host waits, kept for the next one */
  guint n_spare;
};

static GMutex registry_lock;
static GHashTable *registry;    /* hip_stream -> GstMiHipTimeline * (owns a reference) */
static guint64 fences_recorded; /* atomic */

static GstMiHipTimeline *
timeline_ref (GstMiHipTimeline * tl)
{
  g_atomic_int_inc (&tl->refcount);
  return tl;
}

void
gst_mi_hip_timeline_unref (GstMiHipTimeline * tl)
{
  guint i;

  if (tl == NULL || !g_atomic_int_dec_and_test (&tl->refcount))
    return;
  if (tl->fence)
    mibayer_dev_event_destroy (tl->device, tl->fence);
  for (i = 0; i < tl->n_spare; i++)
    mibayer_dev_event_destroy (tl->device, tl->spare[i]);
  g_mutex_clear (&tl->lock);
  g_free (tl);
}

GstMiHipTimeline *
gst_mi_hip_timeline_for (gint device, gpointer hip_stream)
{
  GstMiHipTimeline *tl;

  g_mutex_lock (&registry_lock);
  if (registry == NULL)
    registry = g_hash_table_new (g_direct_hash, g_direct_equal);
  tl = g_hash_table_lookup (registry, hip_stream);
  /* An entry may outlive its stream (a context's queue went with the context) and the runtime may hand the address
   * out again: the old stream's work was waited for before it went (gst_mi_hip_timeline_settle), so carrying the
   * counters on is right -- every old access is behind whatever the new stream is asked to fence. */
  if (tl == NULL) {
    tl = g_new0 (GstMiHipTimeline, 1);
    tl->refcount = 1;           /* the registry's */
    tl->stream = hip_stream;
    g_mutex_init (&tl->lock);
    g_hash_table_insert (registry, hip_stream, tl);
  }
  tl->device = device;
  timeline_ref (tl);
  g_mutex_unlock (&registry_lock);
  return tl;
}

guint64
gst_mi_hip_timeline_submitted (GstMiHipTimeline * tl)
{
  return __atomic_load_n (&tl->submitted, __ATOMIC_SEQ_CST);
}

guint64
gst_mi_hip_timeline_fences_recorded (void)
{
  return __atomic_load_n (&fences_recorded, __ATOMIC_RELAXED);
}

static inline gboolean
timeline_reached (GstMiHipTimeline * tl, guint64 seq)
{
  return __atomic_load_n (&tl->completed, __ATOMIC_ACQUIRE) >= seq;
}

/* monotonic: never moves backwards whoever calls in whatever order */
void
gst_mi_hip_timeline_settle (GstMiHipTimeline * tl, guint64 upto)
{
  guint64 seen = __atomic_load_n (&tl->completed, __ATOMIC_RELAXED);

  while (seen < upto
      && !__atomic_compare_exchange_n (&tl->completed, &seen, upto, TRUE,
          __ATOMIC_RELEASE, __ATOMIC_RELAXED));
}

/* The host waits until the stream has got past access `seq`.  The event is recorded NOW: behind `seq` and behind
 * everything else marked so far, all of which is complete when it fires. */
static gboolean
timeline_wait_host (GstMiHipTimeline * tl, guint64 seq)
{
  gpointer ev;
  guint64 upto;
  gboolean ok;

  if (timeline_reached (tl, seq))
    return TRUE;
  g_mutex_lock (&tl->lock);
  if (tl->retired || timeline_reached (tl, seq)) {
    g_mutex_unlock (&tl->lock);
    return TRUE;
  }
  ev = tl->n_spare > 0 ? tl->spare[--tl->n_spare]
      : mibayer_dev_event_create (tl->device);
  upto = gst_mi_hip_timeline_submitted (tl);
  ok = ev != NULL
      && mibayer_dev_event_record (tl->device, ev, tl->stream) == MIBAYER_OK;
  g_mutex_unlock (&tl->lock);   /* not held while waiting: others may fence the stream meanwhile */
  if (ok) {
    __atomic_add_fetch (&fences_recorded, 1, __ATOMIC_RELAXED);
    ok = mibayer_dev_event_wait (tl->device, ev) == MIBAYER_OK;
  }
  if (ok)
    gst_mi_hip_timeline_settle (tl, upto);
  if (ev != NULL) {
    g_mutex_lock (&tl->lock);
    if (tl->n_spare < TIMELINE_SPARE_EVENTS) {
      tl->spare[tl->n_spare++] = ev;
      ev = NULL;
    }
    g_mutex_unlock (&tl->lock);
    if (ev != NULL)
      mibayer_dev_event_destroy (tl->device, ev);
  }
  return ok;
}

/* `target`'s stream starts what is queued on it from now on after access `seq` of `tl`.  One event per timeline
 * serves every waiter until an access newer than it is asked for: the memories of one launch (both buffers of a
 * frame, all 2 N of a list launch) cost one record between them, and a second waiting stream costs none. */
static gboolean
timeline_order_stream (GstMiHipTimeline * tl, guint64 seq,
    GstMiHipTimeline * target)
{
  gboolean ok = TRUE;

  g_mutex_lock (&tl->lock);
  /* a fence that covers the access and has fired: the access is done, and so is everything before that fence -- a
   * query instead of a wait queued frame after frame for something that completed long ago (a prefilled source's
   * frames, read for ever by a converter on another queue) */
  if (!tl->retired && tl->fence != NULL && tl->fence_seq >= seq
      && mibayer_dev_event_query (tl->device, tl->fence) == 1)
    gst_mi_hip_timeline_settle (tl, tl->fence_seq);
  if (!tl->retired && !timeline_reached (tl, seq)) {
    if (tl->fence == NULL)
      tl->fence = mibayer_dev_event_create (tl->device);
    if (tl->fence == NULL) {
      ok = FALSE;
    } else if (tl->fence_seq < seq) {
      const guint64 upto = gst_mi_hip_timeline_submitted (tl);

      ok = mibayer_dev_event_record (tl->device, tl->fence,
          tl->stream) == MIBAYER_OK;
      if (ok) {
        tl->fence_seq = upto;
        __atomic_add_fetch (&fences_recorded, 1, __ATOMIC_RELAXED);
      }
    }
    /* (the wait is queued against what the event stands for at this moment; recording it again later, for a newer
     * access, does not move a wait that is already queued) */
    ok = ok && mibayer_dev_stream_wait_event (target->device, target->stream,
        tl->fence) == MIBAYER_OK;
  }
  g_mutex_unlock (&tl->lock);
  return ok;
}

void
gst_mi_hip_stream_destroy (gint device, gpointer hip_stream)
{
  GstMiHipTimeline *tl = NULL;

  if (hip_stream == NULL)
    return;
  g_mutex_lock (&registry_lock);
  if (registry != NULL) {
    tl = g_hash_table_lookup (registry, hip_stream);
    if (tl != NULL)
      g_hash_table_remove (registry, hip_stream);       /* the registry's reference is ours now */
  }
  g_mutex_unlock (&registry_lock);
  if (tl != NULL)
    g_mutex_lock (&tl->lock);   /* nobody records on the stream while it goes */
  mibayer_dev_stream_destroy (device, hip_stream);      /* synchronises */
  if (tl != NULL) {
    tl->retired = TRUE;
    gst_mi_hip_timeline_settle (tl, gst_mi_hip_timeline_submitted (tl));
    g_mutex_unlock (&tl->lock);
    gst_mi_hip_timeline_unref (tl);
  }
}

static void
memory_drop_accesses (GstMiHipMemory * m)
{
  guint i;

  for (i = 0; i < m->n_access; i++)
    gst_mi_hip_timeline_unref (m->access[i].timeline);
  m->n_access = 0;
}

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
  memory_drop_accesses (m);     /* (what a failed wait left behind) */
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
  guint i, kept = 0;

  g_mutex_lock (&m->lock);
  for (i = 0; i < m->n_access; i++) {
    GstMiHipTimeline *tl = m->access[i].timeline;

    if (timeline_wait_host (tl, m->access[i].seq)) {
      gst_mi_hip_timeline_unref (tl);
    } else {
      ok = FALSE;
      GST_ERROR ("waiting for queued GPU work failed: %s",
          mibayer_last_hip_error ());
      m->access[kept++] = m->access[i];
    }
  }
  m->n_access = kept;
  g_mutex_unlock (&m->lock);
  return ok;
}

gboolean
gst_mi_hip_memory_order_after_tl (GstMiHipMemory * m, GstMiHipTimeline * target)
{
  gboolean ok = TRUE;
  guint i, kept = 0;

  g_mutex_lock (&m->lock);
  for (i = 0; i < m->n_access; i++) {
    GstMiHipTimeline *tl = m->access[i].timeline;
    const guint64 seq = m->access[i].seq;

    /* Work queued on the stream the access was queued on runs after it anyway (a stream is in order).  In a
     * device-resident pipeline every element's context launches on the device's ONE shared compute queue, so this is
     * the common case: no runtime call at all. */
    if (tl != target) {
      if (timeline_reached (tl, seq)) {
        gst_mi_hip_timeline_unref (tl); /* long done: forget it */
        continue;
      }
      if (!timeline_order_stream (tl, seq, target))
        ok = FALSE;
    }
    m->access[kept++] = m->access[i];
  }
  m->n_access = kept;
  g_mutex_unlock (&m->lock);
  return ok;
}

void
gst_mi_hip_memory_mark_access_tl (GstMiHipMemory * m, GstMiHipTimeline * tl)
{
  const guint64 seq = __atomic_add_fetch (&tl->submitted, 1, __ATOMIC_SEQ_CST);
  guint i;

  g_mutex_lock (&m->lock);
  m->device_defined = TRUE;
  for (i = 0; i < m->n_access; i++)
    if (m->access[i].timeline == tl) {
      m->access[i].seq = seq;   /* in order: the later access completes last */
      g_mutex_unlock (&m->lock);
      return;
    }
  if (m->n_access == GST_MI_HIP_MEMORY_MAX_ACCESSES) {
    /* more queues than slots have touched this memory: make room by dropping entries that have completed, or by
     * waiting for the oldest one (never seen outside tests) */
    guint kept = 0;

    for (i = 0; i < m->n_access; i++) {
      if (timeline_reached (m->access[i].timeline, m->access[i].seq))
        gst_mi_hip_timeline_unref (m->access[i].timeline);
      else
        m->access[kept++] = m->access[i];
    }
    m->n_access = kept;
    if (m->n_access == GST_MI_HIP_MEMORY_MAX_ACCESSES) {
      (void) timeline_wait_host (m->access[0].timeline, m->access[0].seq);
      gst_mi_hip_timeline_unref (m->access[0].timeline);
      memmove (&m->access[0], &m->access[1],
          (GST_MI_HIP_MEMORY_MAX_ACCESSES - 1) * sizeof m->access[0]);
      m->n_access--;
    }
  }
  m->access[m->n_access].timeline = timeline_ref (tl);
  m->access[m->n_access].seq = seq;
  m->n_access++;
  g_mutex_unlock (&m->lock);
}

gboolean
gst_mi_hip_memory_order_after (GstMiHipMemory * m, gpointer hip_stream)
{
  GstMiHipTimeline *tl = gst_mi_hip_timeline_for (m->device, hip_stream);
  gboolean ok = gst_mi_hip_memory_order_after_tl (m, tl);

  gst_mi_hip_timeline_unref (tl);
  return ok;
}

gboolean
gst_mi_hip_memory_mark_access (GstMiHipMemory * m, gpointer hip_stream)
{
  GstMiHipTimeline *tl = gst_mi_hip_timeline_for (m->device, hip_stream);

  gst_mi_hip_memory_mark_access_tl (m, tl);
  gst_mi_hip_timeline_unref (tl);
  return TRUE;
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
 * stays ordered: the memory's queued accesses are not touched. */
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
