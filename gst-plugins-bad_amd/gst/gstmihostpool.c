/* GstMiHostPool: buffer pool over hipHostMalloc-pinned memory.  See the header. */
#include "gstmihostpool.h"

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

static GstFlowReturn
gst_mi_host_pool_alloc_buffer (GstBufferPool * pool, GstBuffer ** buffer,
    GstBufferPoolAcquireParams * params)
{
  GstMiHostPool *self = GST_MI_HOST_POOL (pool);
  gpointer data = self->device >= 0
      ? mibayer_host_alloc_near (self->device, self->size)
      : mibayer_host_alloc (self->size);
  GstBuffer *buf;

  if (data == NULL) {
    GST_ERROR_OBJECT (pool, "hipHostMalloc of %u bytes failed: %s",
        self->size, mibayer_last_hip_error ());
    return GST_FLOW_ERROR;
  }
  buf = gst_buffer_new ();
  gst_buffer_append_memory (buf, gst_memory_new_wrapped (0, data, self->size,
          0, self->size, data, pinned_memory_free));
  *buffer = buf;
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
  self->device = -1;
}

GstBufferPool *
gst_mi_host_pool_new (gint device)
{
  GstBufferPool *pool = g_object_new (GST_TYPE_MI_HOST_POOL, NULL);

  GST_MI_HOST_POOL (pool)->device = device;
  gst_object_ref_sink (pool);
  return pool;
}
