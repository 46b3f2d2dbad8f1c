/* GstMiHostPool: buffer pool over hipHostMalloc-pinned memory.  See the header. */
#include "gstmihostpool.h"

#include <string.h>

#include "mibayer.h"

GST_DEBUG_CATEGORY_STATIC (gst_mi_host_pool_debug);
#define GST_CAT_DEFAULT gst_mi_host_pool_debug

#ifndef MI_HOST_POOL_TYPE_NAME
#define MI_HOST_POOL_TYPE_NAME "GstMiHostPool"
#endif

static void gst_mi_host_pool_class_init (GstMiHostPoolClass * klass);
static void gst_mi_host_pool_init (GstMiHostPool * self);
static gpointer gst_mi_host_pool_parent_class = NULL;

static void
gst_mi_host_pool_class_intern_init (gpointer klass, gpointer data)
{
  gst_mi_host_pool_parent_class = g_type_class_peek_parent (klass);
  gst_mi_host_pool_class_init ((GstMiHostPoolClass *) klass);
}

/* Registered under a per-plugin name (see the header).  Should the name be
 * taken all the same -- the same plugin file loaded from two paths -- a
 * numbered one is used instead of failing. */
GType
gst_mi_host_pool_get_type (void)
{
  static gsize type_id = 0;

  if (g_once_init_enter (&type_id)) {
    gchar *name = g_strdup (MI_HOST_POOL_TYPE_NAME);
    GType t;
    guint n = 1;

    while (g_type_from_name (name) != 0) {
      g_free (name);
      name = g_strdup_printf ("%s%u", MI_HOST_POOL_TYPE_NAME, ++n);
    }
    t = g_type_register_static_simple (GST_TYPE_BUFFER_POOL,
        g_intern_string (name), sizeof (GstMiHostPoolClass),
        (GClassInitFunc) gst_mi_host_pool_class_intern_init,
        sizeof (GstMiHostPool), (GInstanceInitFunc) gst_mi_host_pool_init, 0);
    g_free (name);
    g_once_init_leave (&type_id, t);
  }
  return type_id;
}

static gboolean
gst_mi_host_pool_set_config (GstBufferPool * pool, GstStructure * config)
{
  GstMiHostPool *self = GST_MI_HOST_POOL (pool);
  GstCaps *caps = NULL;
  guint size = 0, min = 0, max = 0;

  if (!gst_buffer_pool_config_get_params (config, &caps, &size, &min, &max)
      || size == 0) {
    GST_WARNING_OBJECT (pool, "invalid pool configuration");
    return FALSE;
  }
  self->size = size;
  return GST_BUFFER_POOL_CLASS (gst_mi_host_pool_parent_class)->set_config
      (pool, config);
}

static void
pinned_memory_free (gpointer data)
{
  mibayer_host_free (data);
}

/* `size` usable bytes of pinned memory with the prefix, padding and alignment
 * of `params` (NULL = none), as one wrapped system memory */
static GstMemory *
pinned_memory_new (gint device, gsize size, const GstAllocationParams * params)
{
  const gsize align = params ? params->align : 0;       /* a mask: 2^n - 1 */
  const gsize prefix = params ? params->prefix : 0;
  const gsize padding = params ? params->padding : 0;
  const GstMemoryFlags flags = params ? params->flags : 0;
  const gsize maxsize = size + prefix + padding + align;
  gsize offset = prefix, mis;
  guint8 *data = device >= 0 ? mibayer_host_alloc_near (device, maxsize)
      : mibayer_host_alloc (maxsize);

  if (data == NULL)
    return NULL;
  /* hipHostMalloc memory starts on a page; the first byte after the prefix goes
   * onto the requested boundary */
  mis = ((gsize) (data + offset)) & align;
  if (mis)
    offset += align + 1 - mis;
  if (offset && (flags & GST_MEMORY_FLAG_ZERO_PREFIXED))
    memset (data, 0, offset);
  if (maxsize > offset + size && (flags & GST_MEMORY_FLAG_ZERO_PADDED))
    memset (data + offset + size, 0, maxsize - offset - size);
  return gst_memory_new_wrapped (flags, data, maxsize, offset, size, data,
      pinned_memory_free);
}

static GstFlowReturn
gst_mi_host_pool_alloc_buffer (GstBufferPool * pool, GstBuffer ** buffer,
    GstBufferPoolAcquireParams * params)
{
  GstMiHostPool *self = GST_MI_HOST_POOL (pool);
  GstStructure *config = gst_buffer_pool_get_config (pool);
  GstAllocationParams aparams;
  GstAllocator *ignored = NULL;
  GstMemory *mem;

  /* whoever configured the pool may have asked for a prefix / padding /
   * alignment (GstBaseSrc and GstBaseTransform copy the query's allocation
   * params into the pool they were offered); the allocator itself is ours */
  gst_allocation_params_init (&aparams);
  if (config != NULL) {
    (void) gst_buffer_pool_config_get_allocator (config, &ignored, &aparams);
    gst_structure_free (config);
  }
  mem = pinned_memory_new (self->ndevices > 0
      ? self->devices[(guint) g_atomic_int_add (&self->next, 1) % self->ndevices] : -1, self->size, &aparams);
  if (mem == NULL) {
    GST_ERROR_OBJECT (pool, "hipHostMalloc of %u bytes failed: %s",
        self->size, mibayer_last_hip_error ());
    return GST_FLOW_ERROR;
  }
  *buffer = gst_buffer_new ();
  gst_buffer_append_memory (*buffer, mem);
  return GST_FLOW_OK;
}

static void
gst_mi_host_pool_class_init (GstMiHostPoolClass * klass)
{
  GstBufferPoolClass *pool_class = GST_BUFFER_POOL_CLASS (klass);

  pool_class->set_config = gst_mi_host_pool_set_config;
  pool_class->alloc_buffer = gst_mi_host_pool_alloc_buffer;
  GST_DEBUG_CATEGORY_INIT (gst_mi_host_pool_debug, "mihostpool", 0,
      "pinned host memory buffer pool");
}

static void
gst_mi_host_pool_init (GstMiHostPool * self)
{
  self->size = 0;
  self->ndevices = 0;
  self->next = 0;
}

GstBufferPool *
gst_mi_host_pool_new_for_devices (const gint * devices, guint n)
{
  GstBufferPool *pool = g_object_new (GST_TYPE_MI_HOST_POOL, NULL);
  GstMiHostPool *self = GST_MI_HOST_POOL (pool);
  guint i;

  for (i = 0; i < n && i < GST_MI_HOST_POOL_MAX_DEVICES; i++)
    if (devices[i] >= 0)
      self->devices[self->ndevices++] = devices[i];
  gst_object_ref_sink (pool);
  return pool;
}

GstBufferPool *
gst_mi_host_pool_new (gint device)
{
  return gst_mi_host_pool_new_for_devices (&device, 1);
}

/* ---- the same memory as a GstAllocator ---------------------------------------------- */

typedef struct
{
  GstAllocator parent;
  gint devices[GST_MI_HOST_POOL_MAX_DEVICES];
  guint ndevices;
  volatile gint next;
} GstMiHostAllocator;

typedef struct
{
  GstAllocatorClass parent_class;
} GstMiHostAllocatorClass;

static GstMemory *
gst_mi_host_allocator_alloc (GstAllocator * allocator, gsize size,
    GstAllocationParams * params)
{
  GstMiHostAllocator *self = (GstMiHostAllocator *) allocator;
  GstMemory *mem = pinned_memory_new (self->ndevices > 0
      ? self->devices[(guint) g_atomic_int_add (&self->next, 1) % self->ndevices] : -1, size, params);

  if (mem == NULL)
    GST_ERROR_OBJECT (allocator, "hipHostMalloc of %" G_GSIZE_FORMAT
        " bytes failed: %s", size, mibayer_last_hip_error ());
  return mem;
}

/* the memories are wrapped system memory: the system allocator frees them
 * (through pinned_memory_free), never this one */
static void
gst_mi_host_allocator_free (GstAllocator * allocator, GstMemory * memory)
{
  g_warn_if_reached ();
}

static void
gst_mi_host_allocator_class_init (gpointer klass, gpointer data)
{
  GstAllocatorClass *allocator_class = GST_ALLOCATOR_CLASS (klass);

  allocator_class->alloc = gst_mi_host_allocator_alloc;
  allocator_class->free = gst_mi_host_allocator_free;
}

static void
gst_mi_host_allocator_init (GTypeInstance * instance, gpointer klass)
{
  ((GstMiHostAllocator *) instance)->ndevices = 0;
  ((GstMiHostAllocator *) instance)->next = 0;
}

/* per-plugin type name, like the pool's */
GType
gst_mi_host_allocator_get_type (void)
{
  static gsize type_id = 0;

  if (g_once_init_enter (&type_id)) {
    gchar *name = g_strdup (MI_HOST_POOL_TYPE_NAME "Allocator");
    GType t;
    guint n = 1;

    while (g_type_from_name (name) != 0) {
      g_free (name);
      name = g_strdup_printf ("%sAllocator%u", MI_HOST_POOL_TYPE_NAME, ++n);
    }
    t = g_type_register_static_simple (GST_TYPE_ALLOCATOR,
        g_intern_string (name), sizeof (GstMiHostAllocatorClass),
        gst_mi_host_allocator_class_init, sizeof (GstMiHostAllocator),
        gst_mi_host_allocator_init, 0);
    g_free (name);
    g_once_init_leave (&type_id, t);
  }
  return type_id;
}

GstAllocator *
gst_mi_host_allocator_new_for_devices (const gint * devices, guint n)
{
  GstMiHostAllocator *self =
      g_object_new (gst_mi_host_allocator_get_type (), NULL);
  guint i;

  for (i = 0; i < n && i < GST_MI_HOST_POOL_MAX_DEVICES; i++)
    if (devices[i] >= 0)
      self->devices[self->ndevices++] = devices[i];
  gst_object_ref_sink (self);
  return GST_ALLOCATOR_CAST (self);
}

GstAllocator *
gst_mi_host_allocator_new (gint device)
{
  return gst_mi_host_allocator_new_for_devices (&device, 1);
}
