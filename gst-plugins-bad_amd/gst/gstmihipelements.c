/* Plugin `mihip`: elements that keep video frames in MI355X device memory
 * (caps feature memory:HIPMemory), SURVEY.md section 8(f) rank 4:
 *
 *   hipupload      video/x-raw | video/x-bayer            -> same caps (memory:HIPMemory)
 *   hipdownload    same caps (memory:HIPMemory)           -> system memory
 *   hipbayer2rgb   video/x-bayer(memory:HIPMemory)        -> video/x-raw(memory:HIPMemory)
 *   hiprgb2bayer   video/x-raw(memory:HIPMemory), ARGB    -> video/x-bayer(memory:HIPMemory)
 *   hipbayersrc    (gstmihipbayersrc.c) synthetic video/x-bayer(memory:HIPMemory) frames generated in device memory
 *
 * so that   ... ! hipupload ! hipbayer2rgb ! <GPU consumer>   never moves the
 * 4 B/pixel output over PCIe -- the only way a *pipeline* gets near the HBM
 * roofline of the kernel.  They live in their own plugin so that plugin `bayer`
 * keeps exactly the reference's two element factories.
 *
 * Pattern in the reference tree (its NVIDIA path): sys/nvcodec/gstcudaupload.c,
 * gstcudadownload.c, gstcudabasetransform.c:436-587 (allocation queries).
 * hipbayer2rgb's caps logic is that of bayer2rgb (gst/bayer/gstbayer2rgb.c:237-352)
 * with the caps feature carried through.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <string.h>

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "gstmihipmemory.h"
#include "gstmihostpool.h"
#include "mibayer.h"

GST_DEBUG_CATEGORY_STATIC (gst_mi_hip_debug);
#define GST_CAT_DEFAULT gst_mi_hip_debug

#define HIP_CAPS(media) media "(" GST_CAPS_FEATURE_MEMORY_HIP ")"

enum
{
  PROP_0,
  PROP_DEVICE_ID,
  PROP_ASYNC,
  PROP_BATCH,
  PROP_AUTOTUNE,
  PROP_PLAN,
  PROP_OVERLAP
};

/* bytes of one frame for either media type; same rules as bayer2rgb's
 * get_unit_size (reference gstbayer2rgb.c:324-352) for the two it knows */
static gboolean
frame_size_from_caps (GstCaps * caps, gsize * size)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w, h;

  if (gst_structure_has_name (s, "video/x-bayer")) {
    if (!gst_structure_get_int (s, "width", &w)
        || !gst_structure_get_int (s, "height", &h))
      return FALSE;
    *size = (gsize) GST_ROUND_UP_4 (w) * h;
    return TRUE;
  } else {
    GstVideoInfo info;

    if (!gst_video_info_from_caps (&info, caps))
      return FALSE;
    *size = GST_VIDEO_INFO_SIZE (&info);
    return TRUE;
  }
}

static GstCaps *
caps_with_feature (GstCaps * caps, const gchar * feature)
{
  GstCaps *out = gst_caps_copy (caps);
  guint i, n = gst_caps_get_size (out);

  for (i = 0; i < n; i++)
    gst_caps_set_features (out, i, feature ? gst_caps_features_new (feature,
            NULL) : gst_caps_features_new_empty ());
  return out;
}

static GstBufferPool *
configured_pool (GstBufferPool * pool, GstCaps * caps, guint size, guint min)
{
  GstStructure *config = gst_buffer_pool_get_config (pool);

  gst_buffer_pool_config_set_params (config, caps, size, min, 0);
  if (!gst_buffer_pool_set_config (pool, config)) {
    gst_object_unref (pool);
    return NULL;
  }
  return pool;
}

/* the single GstMiHipMemory of a buffer, or NULL */
static GstMemory *
buffer_hip_memory (GstBuffer * buf)
{
  GstMemory *mem;

  if (gst_buffer_n_memory (buf) != 1)
    return NULL;
  mem = gst_buffer_peek_memory (buf, 0);
  return gst_is_mi_hip_memory (mem) ? mem : NULL;
}

/* ======================================================================== */
/* hipupload / hipdownload                                                   */
/* ======================================================================== */

/* one host buffer whose DMA is still in flight: hipupload's input (kept until the copy has read
 * it), hipdownload's output (pushed once the copy has written it) */
typedef struct
{
  GstBuffer *host_buf;          /* our reference keeps the memory (and its pool slot) alive */
  GstMapInfo map;
  gpointer event;               /* recorded on the copy queue right after the copy */
  GstBuffer *dev_buf;           /* hipdownload: the device buffer being read (released with the entry) */
} PendingUpload;

#define MAX_PENDING_UPLOADS 4

typedef struct
{
  GstBaseTransform parent;
  gint device_id;               /* properties: g_atomic_int_* (set from any thread) */
  gint async;                   /* do not wait for the DMA (property "async") */
  gboolean prerolled;           /* hipdownload: a frame has left since start / flush */
  /* hipupload, async: the copy queue and the host buffers it still reads */
  gpointer stream;
  gint stream_device;
  GQueue pending;               /* PendingUpload*, oldest first */
} GstMiHipXfer;

typedef struct
{
  GstBaseTransformClass parent_class;
  gboolean to_device;           /* TRUE: hipupload, FALSE: hipdownload */
} GstMiHipXferClass;

#define GST_MI_HIP_XFER(obj) ((GstMiHipXfer *) (obj))
#define XFER_ASYNC(self) (g_atomic_int_get (&(self)->async) != 0)
#define XFER_DEVICE(self) (g_atomic_int_get (&(self)->device_id))
#define GST_MI_HIP_XFER_GET_CLASS(obj) \
  ((GstMiHipXferClass *) G_OBJECT_GET_CLASS (obj))

G_DEFINE_ABSTRACT_TYPE (GstMiHipXfer, gst_mi_hip_xfer, GST_TYPE_BASE_TRANSFORM);

static void
xfer_set_property (GObject * object, guint prop_id, const GValue * value,
    GParamSpec * pspec)
{
  /* read by the streaming thread: plain aligned ints, set atomically */
  if (prop_id == PROP_DEVICE_ID)
    g_atomic_int_set (&GST_MI_HIP_XFER (object)->device_id,
        g_value_get_int (value));
  else if (prop_id == PROP_ASYNC)
    g_atomic_int_set (&GST_MI_HIP_XFER (object)->async,
        g_value_get_boolean (value));
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

static void
xfer_get_property (GObject * object, guint prop_id, GValue * value,
    GParamSpec * pspec)
{
  if (prop_id == PROP_DEVICE_ID)
    g_value_set_int (value,
        g_atomic_int_get (&GST_MI_HIP_XFER (object)->device_id));
  else if (prop_id == PROP_ASYNC)
    g_value_set_boolean (value,
        g_atomic_int_get (&GST_MI_HIP_XFER (object)->async));
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

/* ---- asynchronous upload: deferred release of the host buffers --------------- */
/* The synchronous uploader waits for every DMA, so upstream cannot fill the next
 * frame while this one crosses PCIe.  Asynchronously, the copy is queued on the
 * element's own copy queue, the device memory is marked with an event recorded
 * after it (downstream orders its GPU work after that event, or waits for it on
 * a CPU map -- the protocol of gstmihipmemory.h), and the element returns at
 * once.  What has to outlive the call is the HOST buffer: the DMA engine still
 * reads it, and handing it back to its pool would let upstream overwrite a frame
 * that has not left yet.  So each upload keeps a mapped reference to its input
 * buffer in `pending`, released when its own event has fired: polled at the
 * next buffer, waited for when more than MAX_PENDING_UPLOADS pile up and
 * before EOS / flush / stop. */

static void
pending_upload_free (GstMiHipXfer * self, PendingUpload * p)
{
  gst_buffer_unmap (p->host_buf, &p->map);
  gst_buffer_unref (p->host_buf);
  if (p->dev_buf)
    gst_buffer_unref (p->dev_buf);
  if (p->event)
    mibayer_dev_event_destroy (self->stream_device, p->event);
  g_free (p);
}

/* release what has completed; wait == TRUE: everything */
static void
xfer_reap_uploads (GstMiHipXfer * self, gboolean wait, guint keep)
{
  PendingUpload *p;

  while ((p = g_queue_peek_head (&self->pending)) != NULL) {
    gboolean must = wait || g_queue_get_length (&self->pending) > keep;

    if (p->event != NULL) {
      if (must)
        mibayer_dev_event_wait (self->stream_device, p->event);
      else if (mibayer_dev_event_query (self->stream_device, p->event) == 0)
        break;                  /* still copying: younger ones are too */
    }
    g_queue_pop_head (&self->pending);
    pending_upload_free (self, p);
  }
}

static void
xfer_drop_stream (GstMiHipXfer * self)
{
  xfer_reap_uploads (self, TRUE, 0);
  if (self->stream) {
    gst_mi_hip_stream_destroy (self->stream_device, self->stream);      /* synchronises; retires the stream's timeline */
    self->stream = NULL;
  }
}

static gboolean
xfer_upload_async (GstMiHipXfer * self, GstBuffer * inbuf, GstMemory * dev_mem)
{
  GstMiHipMemory *m = (GstMiHipMemory *) dev_mem;
  PendingUpload *p;
  GstMapInfo dev_map;
  gsize n;

  if (self->stream != NULL && self->stream_device != m->device)
    xfer_drop_stream (self);
  if (self->stream == NULL) {
    self->stream = mibayer_dev_stream_create (m->device);
    self->stream_device = m->device;
    if (self->stream == NULL)
      return FALSE;
  }
  xfer_reap_uploads (self, FALSE, MAX_PENDING_UPLOADS - 1);

  p = g_new0 (PendingUpload, 1);
  if (!gst_buffer_map (inbuf, &p->map, GST_MAP_READ)) {
    g_free (p);
    return FALSE;
  }
  /* From pageable memory (a source that ignored the proposed pinned pool, e.g.
   * fakesrc, filesrc) hipMemcpyAsync blocks for the whole copy anyway -- 159 us per
   * 4K mosaic in the HIP API trace, profiles/r02_async_upload_hip_api_trace.log --
   * and the deferred release only adds calls: take the plain blocking copy. */
  if (!mibayer_host_is_pinned (p->map.data)) {
    gst_buffer_unmap (inbuf, &p->map);
    g_free (p);
    return FALSE;
  }
  if (!gst_memory_map (dev_mem, &dev_map,
          GST_MAP_WRITE | GST_MAP_HIP | GST_MAP_HIP_ASYNC)) {
    gst_buffer_unmap (inbuf, &p->map);
    g_free (p);
    return FALSE;
  }
  n = MIN (p->map.size, dev_map.size);
  /* the copy starts after whatever the device memory's last user queued */
  if (!gst_mi_hip_memory_order_after (m, self->stream))
    gst_mi_hip_memory_wait (m);
  if (mibayer_dev_upload_async (m->device, dev_map.data, p->map.data, n,
          self->stream) != MIBAYER_OK) {
    gst_memory_unmap (dev_mem, &dev_map);
    gst_buffer_unmap (inbuf, &p->map);
    g_free (p);
    return FALSE;
  }
  /* two markers behind the copy: the memory's, for whoever touches it next, and
   * ours, for the release of the host buffer */
  p->event = mibayer_dev_event_create (m->device);
  if (!gst_mi_hip_memory_mark_access (m, self->stream) || p->event == NULL
      || mibayer_dev_event_record (m->device, p->event,
          self->stream) != MIBAYER_OK) {
    /* no event to defer on: finish now */
    gst_mi_hip_stream_destroy (m->device, self->stream);        /* synchronises */
    self->stream = NULL;
    gst_memory_unmap (dev_mem, &dev_map);
    gst_buffer_unmap (inbuf, &p->map);
    if (p->event)
      mibayer_dev_event_destroy (m->device, p->event);
    g_free (p);
    return TRUE;
  }
  gst_memory_unmap (dev_mem, &dev_map);
  p->host_buf = gst_buffer_ref (inbuf);
  g_queue_push_tail (&self->pending, p);
  return TRUE;
}

/* ---- asynchronous download: the queued mode of hipdownload ---------------------- */
/* The synchronous downloader holds the streaming thread for the whole 4 B/pixel
 * copy (0.63 ms per 4K frame), during which nothing upstream on that thread --
 * the next upload, the next launch -- is queued: `hipupload ! hipbayer2rgb !
 * hipdownload` runs at 1285 fps where the same three stages overlapped inside
 * `bayer2rgb inflight=2` reach 1604.  Asynchronously, the copy is queued on the
 * element's copy queue (ordered after the device memory's last access, and the
 * memory is marked so that it is not recycled under the copy), the output buffer
 * waits in `pending` with its own event, and generate_output hands on the
 * oldest output once its copy has finished -- immediately if it has, at the
 * latest when more than MAX_PENDING_UPLOADS are waiting; events that must not
 * overtake buffers push what is waiting first, a flush drops it.  Pageable
 * output memory (a downstream pool that is not pinned) takes the blocking copy:
 * hipMemcpyAsync would block anyway.  The first frame after a start / flush is
 * not held back (preroll). */

/* the oldest output whose copy is done (or, wait == TRUE, the oldest one) */
static GstBuffer *
xfer_pop_download (GstMiHipXfer * self, gboolean wait)
{
  PendingUpload *p = g_queue_peek_head (&self->pending);
  GstBuffer *out;

  if (p == NULL)
    return NULL;
  if (p->event != NULL) {
    if (wait)
      mibayer_dev_event_wait (self->stream_device, p->event);
    else if (mibayer_dev_event_query (self->stream_device, p->event) == 0)
      return NULL;
  }
  g_queue_pop_head (&self->pending);
  out = gst_buffer_ref (p->host_buf);
  pending_upload_free (self, p);        /* unmaps; drops our references */
  return out;
}

static gboolean
xfer_download_async (GstMiHipXfer * self, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstMemory *dev_mem = buffer_hip_memory (inbuf);
  GstMiHipMemory *m = (GstMiHipMemory *) dev_mem;
  PendingUpload *p;
  GstMapInfo dev_map;
  gsize n;

  if (dev_mem == NULL)
    return FALSE;
  if (self->stream != NULL && self->stream_device != m->device)
    return FALSE;               /* frames in flight on another device: blocking copy */
  if (self->stream == NULL) {
    self->stream = mibayer_dev_stream_create (m->device);
    self->stream_device = m->device;
    if (self->stream == NULL)
      return FALSE;
  }
  p = g_new0 (PendingUpload, 1);
  if (!gst_buffer_map (outbuf, &p->map, GST_MAP_WRITE)) {
    g_free (p);
    return FALSE;
  }
  if (!mibayer_host_is_pinned (p->map.data)
      || !gst_memory_map (dev_mem, &dev_map,
          GST_MAP_READ | GST_MAP_HIP | GST_MAP_HIP_ASYNC)) {
    gst_buffer_unmap (outbuf, &p->map);
    g_free (p);
    return FALSE;
  }
  n = MIN (p->map.size, dev_map.size);
  if (!gst_mi_hip_memory_order_after (m, self->stream))
    gst_mi_hip_memory_wait (m);
  p->event = mibayer_dev_event_create (m->device);
  if (p->event == NULL
      || mibayer_dev_download_async (m->device, p->map.data, dev_map.data, n,
          self->stream) != MIBAYER_OK
      || !gst_mi_hip_memory_mark_access (m, self->stream)
      || mibayer_dev_event_record (m->device, p->event,
          self->stream) != MIBAYER_OK) {
    /* whatever was queued must not outlive this call */
    gst_mi_hip_stream_destroy (m->device, self->stream);        /* synchronises */
    self->stream = NULL;
    gst_memory_unmap (dev_mem, &dev_map);
    gst_buffer_unmap (outbuf, &p->map);
    if (p->event)
      mibayer_dev_event_destroy (m->device, p->event);
    g_free (p);
    return FALSE;
  }
  gst_memory_unmap (dev_mem, &dev_map);
  p->host_buf = outbuf;         /* takes the caller's reference */
  p->dev_buf = gst_buffer_ref (inbuf);
  g_queue_push_tail (&self->pending, p);
  return TRUE;
}

static GstFlowReturn xfer_transform (GstBaseTransform * trans,
    GstBuffer * inbuf, GstBuffer * outbuf);

static GstFlowReturn
xfer_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstMiHipXfer *self = GST_MI_HIP_XFER (trans);
  GstBaseTransformClass *klass = GST_BASE_TRANSFORM_GET_CLASS (trans);
  GstBuffer *inbuf;

  if (GST_MI_HIP_XFER_GET_CLASS (trans)->to_device
      || (!XFER_ASYNC (self) && g_queue_is_empty (&self->pending)))
    return GST_BASE_TRANSFORM_CLASS (gst_mi_hip_xfer_parent_class)->generate_output
        (trans, outbuf);

  *outbuf = NULL;
  inbuf = trans->queued_buf;
  trans->queued_buf = NULL;
  if (inbuf != NULL) {
    GstBuffer *out = NULL;
    GstFlowReturn ret = klass->prepare_output_buffer (trans, inbuf, &out);

    if (ret != GST_FLOW_OK || out == NULL) {
      gst_buffer_unref (inbuf);
      return ret == GST_FLOW_OK ? GST_FLOW_ERROR : ret;
    }
    if (!XFER_ASYNC (self) || !xfer_download_async (self, inbuf, out)) {
      /* blocking copy; what is already waiting leaves first, in order */
      GQueue done = G_QUEUE_INIT;
      GstBuffer *b;

      while ((b = xfer_pop_download (self, TRUE)) != NULL)
        g_queue_push_tail (&done, b);
      ret = xfer_transform (trans, inbuf, out);
      gst_buffer_unref (inbuf);
      if (ret != GST_FLOW_OK) {
        gst_buffer_unref (out);
        while ((b = g_queue_pop_head (&done)) != NULL)
          gst_buffer_unref (b);
        return ret;
      }
      g_queue_push_tail (&done, out);
      *outbuf = g_queue_pop_head (&done);
      while ((b = g_queue_pop_head (&done)) != NULL) {  /* rare: keep the order */
        PendingUpload *q = g_new0 (PendingUpload, 1);

        q->host_buf = b;
        if (!gst_buffer_map (b, &q->map, GST_MAP_READ)) {
          gst_buffer_unref (b);
          g_free (q);
          continue;
        }
        g_queue_push_tail (&self->pending, q);  /* no event: ready at once */
      }
      self->prerolled = TRUE;
      return GST_FLOW_OK;
    }
    gst_buffer_unref (inbuf);           /* the pending entry holds its own ref */
  }
  /* the base class calls again for as long as a buffer comes out */
  *outbuf = xfer_pop_download (self, !self->prerolled
      || g_queue_get_length (&self->pending) > MAX_PENDING_UPLOADS);
  if (*outbuf != NULL)
    self->prerolled = TRUE;
  return GST_FLOW_OK;
}

static gboolean
xfer_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  GstMiHipXfer *self = GST_MI_HIP_XFER (trans);
  const gboolean to_device = GST_MI_HIP_XFER_GET_CLASS (trans)->to_device;
  GstBuffer *buf;

  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_EOS:
    case GST_EVENT_CAPS:
    case GST_EVENT_SEGMENT:
    case GST_EVENT_GAP:
      /* serialised with the streaming thread.  hipupload: hand every host buffer
       * back; hipdownload: what is waiting precedes the event */
      if (to_device) {
        if (GST_EVENT_TYPE (event) == GST_EVENT_EOS
            || GST_EVENT_TYPE (event) == GST_EVENT_CAPS)
          xfer_reap_uploads (self, TRUE, 0);
      } else {
        while ((buf = xfer_pop_download (self, TRUE)) != NULL)
          gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (trans), buf);
      }
      break;
    case GST_EVENT_FLUSH_STOP:
      xfer_reap_uploads (self, TRUE, 0);        /* either direction: drop */
      self->prerolled = FALSE;
      break;
    default:
      break;
  }
  return GST_BASE_TRANSFORM_CLASS (gst_mi_hip_xfer_parent_class)->sink_event
      (trans, event);
}

static gboolean
xfer_stop (GstBaseTransform * trans)
{
  xfer_drop_stream (GST_MI_HIP_XFER (trans));
  GST_MI_HIP_XFER (trans)->prerolled = FALSE;
  return TRUE;
}

static void
xfer_finalize (GObject * object)
{
  xfer_drop_stream (GST_MI_HIP_XFER (object));
  G_OBJECT_CLASS (gst_mi_hip_xfer_parent_class)->finalize (object);
}

static GstCaps *
xfer_transform_caps (GstBaseTransform * trans, GstPadDirection direction,
    GstCaps * caps, GstCaps * filter)
{
  gboolean to_device = GST_MI_HIP_XFER_GET_CLASS (trans)->to_device;
  /* the pad we produce caps FOR is the other one */
  gboolean want_device = (direction == GST_PAD_SINK) ? to_device : !to_device;
  GstCaps *result = caps_with_feature (caps,
      want_device ? GST_CAPS_FEATURE_MEMORY_HIP : NULL);

  if (filter) {
    GstCaps *tmp = result;

    result = gst_caps_intersect_full (filter, tmp, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (tmp);
  }
  return result;
}

static gboolean
xfer_get_unit_size (GstBaseTransform * trans, GstCaps * caps, gsize * size)
{
  return frame_size_from_caps (caps, size);
}

/* upstream allocates system memory: offer it pinned memory (upload only) */
static gboolean
xfer_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query,
    GstQuery * query)
{
  GstCaps *caps = NULL;
  gsize size = 0;

  if (!GST_BASE_TRANSFORM_CLASS (gst_mi_hip_xfer_parent_class)->propose_allocation
      (trans, decide_query, query))
    return FALSE;
  if (!GST_MI_HIP_XFER_GET_CLASS (trans)->to_device
      || mibayer_device_count () <= 0)
    return TRUE;
  gst_query_parse_allocation (query, &caps, NULL);
  if (caps && frame_size_from_caps (caps, &size)) {
    GstBufferPool *pool =
        configured_pool (gst_mi_host_pool_new (XFER_DEVICE (GST_MI_HIP_XFER
                (trans))), caps, (guint) size, MAX_PENDING_UPLOADS + 2);

    if (pool) {
      GstAllocationParams params;
      GstAllocator *allocator =
          gst_mi_host_allocator_new (XFER_DEVICE (GST_MI_HIP_XFER (trans)));

      /* asynchronous uploads hold up to MAX_PENDING_UPLOADS input buffers */
      gst_query_add_allocation_pool (query, pool, (guint) size,
          MAX_PENDING_UPLOADS + 2, 0);
      gst_object_unref (pool);
      /* and the allocator behind it, for an upstream that builds its own pool */
      gst_allocation_params_init (&params);
      gst_query_add_allocation_param (query, allocator, &params);
      gst_object_unref (allocator);
    }
  }
  return TRUE;
}

/* our own output: device memory (upload) or pinned host memory (download) */
static gboolean
xfer_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  GstMiHipXfer *self = GST_MI_HIP_XFER (trans);
  gboolean to_device = GST_MI_HIP_XFER_GET_CLASS (trans)->to_device;
  GstCaps *caps = NULL;
  gsize size = 0;

  gst_query_parse_allocation (query, &caps, NULL);
  if (caps && frame_size_from_caps (caps, &size)
      && mibayer_device_count () > 0) {
    GstBufferPool *pool = NULL;

    if (to_device) {
      /* device memory is the only thing our src caps allow: replace whatever
       * downstream proposed */
      while (gst_query_get_n_allocation_pools (query) > 0)
        gst_query_remove_nth_allocation_pool (query, 0);
      pool = configured_pool (gst_mi_hip_pool_new (XFER_DEVICE (self)), caps,
          (guint) size, 2);
    } else if (gst_query_get_n_allocation_pools (query) == 0) {
      pool = configured_pool (gst_mi_host_pool_new (XFER_DEVICE (self)), caps,
          (guint) size, MAX_PENDING_UPLOADS + 2);
    }
    if (pool) {
      gst_query_add_allocation_pool (query, pool, (guint) size,
          to_device ? 2 : MAX_PENDING_UPLOADS + 2, 0);
      gst_object_unref (pool);
    }
  }
  return GST_BASE_TRANSFORM_CLASS (gst_mi_hip_xfer_parent_class)->decide_allocation
      (trans, query);
}

static GstFlowReturn
xfer_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstMiHipXfer *self = GST_MI_HIP_XFER (trans);
  gboolean to_device = GST_MI_HIP_XFER_GET_CLASS (trans)->to_device;
  GstBuffer *host_buf = to_device ? inbuf : outbuf;
  GstBuffer *dev_buf = to_device ? outbuf : inbuf;
  GstMemory *dev_mem = buffer_hip_memory (dev_buf);
  GstMapInfo host_map, dev_map;
  GstFlowReturn ret = GST_FLOW_OK;
  gsize n;
  int rc;

  if (dev_mem == NULL) {
    GST_ELEMENT_ERROR (self, CORE, NEGOTIATION,
        ("%s buffer does not hold HIP device memory", to_device ? "output"
            : "input"), (NULL));
    return GST_FLOW_ERROR;
  }
  if (to_device && XFER_ASYNC (self)) {
    if (xfer_upload_async (self, inbuf, dev_mem))
      return GST_FLOW_OK;
    GST_LOG_OBJECT (self, "pageable input or no copy queue: blocking copy");
  }
  if (!gst_buffer_map (host_buf, &host_map,
          to_device ? GST_MAP_READ : GST_MAP_WRITE))
    return GST_FLOW_ERROR;
  if (!gst_memory_map (dev_mem, &dev_map,
          (to_device ? GST_MAP_WRITE : GST_MAP_READ) | GST_MAP_HIP)) {
    gst_buffer_unmap (host_buf, &host_map);
    return GST_FLOW_ERROR;
  }
  n = MIN (host_map.size, dev_map.size);
  if (to_device)
    rc = mibayer_dev_upload (((GstMiHipMemory *) dev_mem)->device,
        dev_map.data, host_map.data, n);
  else
    rc = mibayer_dev_download (((GstMiHipMemory *) dev_mem)->device,
        host_map.data, dev_map.data, n);
  if (rc != MIBAYER_OK) {
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED, ("HIP copy failed"),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    ret = GST_FLOW_ERROR;
  }
  gst_memory_unmap (dev_mem, &dev_map);
  gst_buffer_unmap (host_buf, &host_map);
  return ret;
}

static void
gst_mi_hip_xfer_class_init (GstMiHipXferClass * klass)
{
  GObjectClass *object_class = G_OBJECT_CLASS (klass);
  GstBaseTransformClass *transform_class = GST_BASE_TRANSFORM_CLASS (klass);

  object_class->set_property = xfer_set_property;
  object_class->get_property = xfer_get_property;
  object_class->finalize = xfer_finalize;
  g_object_class_install_property (object_class, PROP_DEVICE_ID,
      g_param_spec_int ("device-id", "Device ID", "HIP ordinal of the MI355X",
          0, G_MAXINT, 0, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_ASYNC,
      g_param_spec_boolean ("async", "Asynchronous copy",
          "Queue the copy and return.  hipupload: the host buffer is released "
          "when the copy has completed, downstream GPU work is ordered after "
          "it.  hipdownload: the output buffer is pushed when its copy has "
          "completed (up to 4 frames wait; the first frame after a start or "
          "flush is not held back).  Applies to pinned host memory (the pools "
          "these elements propose / use); pageable memory is copied blocking, "
          "as the runtime would do anyway", TRUE,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  transform_class->sink_event = GST_DEBUG_FUNCPTR (xfer_sink_event);
  transform_class->generate_output = GST_DEBUG_FUNCPTR (xfer_generate_output);
  transform_class->stop = GST_DEBUG_FUNCPTR (xfer_stop);
  transform_class->passthrough_on_same_caps = FALSE;
  transform_class->transform_caps = GST_DEBUG_FUNCPTR (xfer_transform_caps);
  transform_class->get_unit_size = GST_DEBUG_FUNCPTR (xfer_get_unit_size);
  transform_class->propose_allocation =
      GST_DEBUG_FUNCPTR (xfer_propose_allocation);
  transform_class->decide_allocation =
      GST_DEBUG_FUNCPTR (xfer_decide_allocation);
  transform_class->transform = GST_DEBUG_FUNCPTR (xfer_transform);
}

static void
gst_mi_hip_xfer_init (GstMiHipXfer * self)
{
  self->device_id = 0;
  self->async = TRUE;
  self->prerolled = FALSE;
  self->stream = NULL;
  self->stream_device = 0;
  g_queue_init (&self->pending);
}

#define SYS_CAPS "video/x-raw; video/x-bayer"
#define DEV_CAPS HIP_CAPS ("video/x-raw") "; " HIP_CAPS ("video/x-bayer")

typedef GstMiHipXfer GstMiHipUpload;
typedef GstMiHipXferClass GstMiHipUploadClass;
typedef GstMiHipXfer GstMiHipDownload;
typedef GstMiHipXferClass GstMiHipDownloadClass;

GType gst_mi_hip_upload_get_type (void);
GType gst_mi_hip_download_get_type (void);
G_DEFINE_TYPE (GstMiHipUpload, gst_mi_hip_upload, gst_mi_hip_xfer_get_type ());
G_DEFINE_TYPE (GstMiHipDownload, gst_mi_hip_download,
    gst_mi_hip_xfer_get_type ());

static void
xfer_add_templates (GstElementClass * element_class, const gchar * sink_caps,
    const gchar * src_caps)
{
  gst_element_class_add_pad_template (element_class,
      gst_pad_template_new ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
          gst_caps_from_string (sink_caps)));
  gst_element_class_add_pad_template (element_class,
      gst_pad_template_new ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
          gst_caps_from_string (src_caps)));
}

static void
gst_mi_hip_upload_class_init (GstMiHipUploadClass * klass)
{
  klass->to_device = TRUE;
  xfer_add_templates (GST_ELEMENT_CLASS (klass), SYS_CAPS, DEV_CAPS);
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass),
      "HIP uploader", "Filter/Video",
      "Copies video frames into MI355X device memory (memory:HIPMemory)",
      "gst-plugins-bad_amd");
}

static void
gst_mi_hip_upload_init (GstMiHipUpload * self)
{
}

static void
gst_mi_hip_download_class_init (GstMiHipDownloadClass * klass)
{
  klass->to_device = FALSE;
  xfer_add_templates (GST_ELEMENT_CLASS (klass), DEV_CAPS, SYS_CAPS);
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass),
      "HIP downloader", "Filter/Video",
      "Copies video frames from MI355X device memory to (pinned) system memory",
      "gst-plugins-bad_amd");
}

static void
gst_mi_hip_download_init (GstMiHipDownload * self)
{
}

/* ======================================================================== */
/* hipbayer2rgb                                                              */
/* ======================================================================== */

typedef struct
{
  GstBaseTransform parent;
  GstVideoInfo info;
  gint width, height, r_off, g_off, b_off, format;
  gint device_id;               /* properties device-id / batch: g_atomic_int_*; -1 = follow the frames */
  mibayer_ctx *ctx;             /* created at the first buffer, on the device its memory lives on */
  gint ctx_device;              /* the device the context was created on */
  GstMiHipTimeline *tl;         /* the timeline of the context's stream: every launch marks its buffers on it */
  gpointer tl_stream;           /* ... and the stream it was looked up for (hb2r_ctx_timeline) */
  GstBufferPool *out_pool;      /* output frames, on the same device */
  gint out_pool_device;
  /* batch mode (property "batch" > 1): input / output buffer pairs waiting for ONE launch over all of
   * them, and converted outputs waiting to be handed to the base class one by one */
  gint batch;
  gint autotune;                /* property "autotune": measure the launch plan on the first frames (g_atomic_int_*);
                                   -1 = not set by anybody: on for batch >= 4, off below (HB2R_AUTOTUNE_ON) */
  gboolean tuned;               /* the context's plan has been settled: measured, or taken from the process cache */
  gint overlap;                 /* property "overlap": consecutive frames go round-robin over the device's frame queues */
  guint frame_no;               /* frames dealt over the frame queues so far: picks the queue */
  gpointer launch_ev;           /* recorded behind every frame-by-frame launch: "is the previous conversion still running?" */
  gint launch_ev_device;
  guint busy_run;               /* consecutive frames that arrived while the previous conversion was still running */
  gchar plan[160];              /* property "plan" (read-only): the context's launch plan and where it came from */
  gboolean prerolled;           /* a frame has left since start / flush: batching may begin */
  GQueue waiting;               /* Hb2rPair* */
  GQueue ready;                 /* GstBuffer* */
} GstMiHipBayer2RGB;

typedef struct
{
  GstBuffer *in, *out;
} Hb2rPair;

#define HB2R_MAX_BATCH 16       /* frames of one list launch (mibayer_process_device_list) */
/* Does this instance measure its launch plan?  Whoever set the property decides; otherwise batch mode from 4 frames per
 * launch on measures (the static default is up to 5 points off on some widths, profiles/
 * r04_plan_sweep_530Mpix.log, a measurement costs one first batch per geometry per PROCESS thanks to the plan cache,
 * and a stream that asked for batch >= 4 has traded latency for throughput already), frame-by-frame mode does not
 * (its first frame would wait 0.1-0.2 s). */
#define HB2R_AUTOTUNE_MIN_BATCH 4
#define HB2R_AUTOTUNE_ON(self) (g_atomic_int_get (&(self)->autotune) >= 0 \
    ? g_atomic_int_get (&(self)->autotune) != 0 \
    : g_atomic_int_get (&(self)->batch) >= HB2R_AUTOTUNE_MIN_BATCH)

/* hiprgb2bayer, the sibling direction (the plugin's second element, reference gst/bayer/gstrgb2bayer.c), is the
 * same element with the pad roles swapped: a subclass whose class carries `inverse` -- exactly how plugin `bayer`
 * shares gstmibayerelement.c between bayer2rgb and rgb2bayer */
typedef struct
{
  GstBaseTransformClass parent_class;
  gboolean inverse;             /* FALSE: hipbayer2rgb, TRUE: hiprgb2bayer */
  const gchar *label;           /* element name used in messages */
} GstMiHipBayer2RGBClass;

#define HB2R_CLASS_OF(obj) ((GstMiHipBayer2RGBClass *) G_OBJECT_GET_CLASS (obj))
#define HB2R_INVERSE(obj) (HB2R_CLASS_OF (obj)->inverse)
#define HB2R_LABEL(obj) (HB2R_CLASS_OF (obj)->label)
/* bytes of the element's input / output frame */
#define HB2R_MOSAIC_BYTES(self) ((gsize) GST_ROUND_UP_4 ((self)->width) * (self)->height)
#define HB2R_VIDEO_BYTES(self) ((gsize) 4 * (self)->width * (self)->height)
#define HB2R_IN_BYTES(self) (HB2R_INVERSE (self) ? HB2R_VIDEO_BYTES (self) : HB2R_MOSAIC_BYTES (self))
#define HB2R_OUT_BYTES(self) (HB2R_INVERSE (self) ? HB2R_MOSAIC_BYTES (self) : HB2R_VIDEO_BYTES (self))

GType gst_mi_hip_bayer2rgb_get_type (void);
G_DEFINE_TYPE (GstMiHipBayer2RGB, gst_mi_hip_bayer2rgb, GST_TYPE_BASE_TRANSFORM);

#define HB2R_SINK_CAPS HIP_CAPS ("video/x-bayer") \
  ",format=(string){bggr,grbg,gbrg,rggb}," \
  "width=(int)[1,MAX],height=(int)[1,MAX],framerate=(fraction)[0/1,MAX]"
#define HB2R_SRC_CAPS GST_VIDEO_CAPS_MAKE_WITH_FEATURES ( \
    GST_CAPS_FEATURE_MEMORY_HIP, \
    "{ RGBx, xRGB, BGRx, xBGR, RGBA, ARGB, BGRA, ABGR }")

static void
hb2r_drop_ctx (GstMiHipBayer2RGB * self)
{
  if (self->launch_ev) {
    mibayer_dev_event_destroy (self->launch_ev_device, self->launch_ev);
    self->launch_ev = NULL;
  }
  self->busy_run = 0;
  if (self->ctx) {
    /* Nothing this element marked may ask the context's stream for a fence once the context (and, if it was the last
     * one on the device, the stream) is gone: wait for what the context queued -- microseconds of kernels -- and tell
     * the timeline how far that was.  (Rounds 2-5 had an event per memory instead, which outlived the stream.) */
    if (self->tl != NULL) {
      const guint64 upto = gst_mi_hip_timeline_submitted (self->tl);

      if (mibayer_sync (self->ctx) == MIBAYER_OK)
        gst_mi_hip_timeline_settle (self->tl, upto);
      gst_mi_hip_timeline_unref (self->tl);
      self->tl = NULL;
    }
    mibayer_destroy (self->ctx);
    self->ctx = NULL;
  }
}

static void
hb2r_drop_out_pool (GstMiHipBayer2RGB * self)
{
  if (self->out_pool) {
    gst_buffer_pool_set_active (self->out_pool, FALSE);
    gst_object_unref (self->out_pool);
    self->out_pool = NULL;
  }
}

static void
hb2r_set_property (GObject * object, guint prop_id, const GValue * value,
    GParamSpec * pspec)
{
  if (prop_id == PROP_DEVICE_ID)
    g_atomic_int_set (&((GstMiHipBayer2RGB *) object)->device_id,
        g_value_get_int (value));
  else if (prop_id == PROP_BATCH)
    g_atomic_int_set (&((GstMiHipBayer2RGB *) object)->batch,
        g_value_get_int (value));
  else if (prop_id == PROP_AUTOTUNE)
    g_atomic_int_set (&((GstMiHipBayer2RGB *) object)->autotune,
        g_value_get_boolean (value) ? 1 : 0);
  else if (prop_id == PROP_OVERLAP)
    g_atomic_int_set (&((GstMiHipBayer2RGB *) object)->overlap,
        g_value_get_boolean (value) ? 1 : 0);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

static void
hb2r_get_property (GObject * object, guint prop_id, GValue * value,
    GParamSpec * pspec)
{
  if (prop_id == PROP_DEVICE_ID)
    g_value_set_int (value,
        g_atomic_int_get (&((GstMiHipBayer2RGB *) object)->device_id));
  else if (prop_id == PROP_BATCH)
    g_value_set_int (value,
        g_atomic_int_get (&((GstMiHipBayer2RGB *) object)->batch));
  else if (prop_id == PROP_AUTOTUNE)
    g_value_set_boolean (value, HB2R_AUTOTUNE_ON ((GstMiHipBayer2RGB *) object));
  else if (prop_id == PROP_OVERLAP)
    g_value_set_boolean (value,
        g_atomic_int_get (&((GstMiHipBayer2RGB *) object)->overlap) != 0);
  else if (prop_id == PROP_PLAN) {
    GST_OBJECT_LOCK (object);
    g_value_set_string (value, ((GstMiHipBayer2RGB *) object)->plan);
    GST_OBJECT_UNLOCK (object);
  } else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

static void
hb2r_drop_queued (GstMiHipBayer2RGB * self)
{
  Hb2rPair *pair;
  GstBuffer *buf;

  while ((pair = g_queue_pop_head (&self->waiting)) != NULL) {
    gst_buffer_unref (pair->in);
    gst_buffer_unref (pair->out);
    g_free (pair);
  }
  while ((buf = g_queue_pop_head (&self->ready)) != NULL)
    gst_buffer_unref (buf);
}

static void
hb2r_finalize (GObject * object)
{
  hb2r_drop_queued ((GstMiHipBayer2RGB *) object);
  hb2r_drop_ctx ((GstMiHipBayer2RGB *) object);
  hb2r_drop_out_pool ((GstMiHipBayer2RGB *) object);
  G_OBJECT_CLASS (gst_mi_hip_bayer2rgb_parent_class)->finalize (object);
}

/* bayer2rgb's caps transform (reference :289-322); caps features survive the
 * structure rename because they are stored beside the structure */
static GstCaps *
hb2r_transform_caps (GstBaseTransform * trans, GstPadDirection direction,
    GstCaps * caps, GstCaps * filter)
{
  GstCaps *result = gst_caps_copy (caps);
  guint i, n = gst_caps_get_size (result);

  for (i = 0; i < n; i++) {
    GstStructure *s = gst_caps_get_structure (result, i);

    if ((direction == GST_PAD_SINK) != HB2R_INVERSE (trans)) {
      gst_structure_set_name (s, "video/x-raw");
      gst_structure_remove_field (s, "format");
    } else {
      gst_structure_set_name (s, "video/x-bayer");
      gst_structure_remove_fields (s, "format", "colorimetry", "chroma-site",
          NULL);
    }
  }
  if (filter) {
    GstCaps *tmp = result;

    result = gst_caps_intersect_full (filter, tmp, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (tmp);
  }
  return result;
}

static gboolean
hb2r_get_unit_size (GstBaseTransform * trans, GstCaps * caps, gsize * size)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w, h;

  if (!gst_structure_get_int (s, "width", &w)
      || !gst_structure_get_int (s, "height", &h))
    return FALSE;
  *size = gst_structure_has_name (s, "video/x-raw") ? (gsize) w * h * 4
      : (gsize) GST_ROUND_UP_4 (w) * h;
  return TRUE;
}

static gboolean
hb2r_set_caps (GstBaseTransform * trans, GstCaps * incaps, GstCaps * outcaps)
{
  static const gchar *orders[] = { "bggr", "gbrg", "grbg", "rggb" };     /* mibayer_pattern order */
  GstMiHipBayer2RGB *self = (GstMiHipBayer2RGB *) trans;
  const gboolean inverse = HB2R_INVERSE (self);
  GstStructure *s = gst_caps_get_structure (inverse ? outcaps : incaps, 0);     /* the mosaic side */
  const gchar *order = gst_structure_get_string (s, "format");
  GstVideoInfo info;
  gint i;

  if (!gst_structure_get_int (s, "width", &self->width)
      || !gst_structure_get_int (s, "height", &self->height) || !order)
    return FALSE;
  for (i = 0; i < 4 && !g_str_equal (order, orders[i]); i++);
  if (i == 4)
    return FALSE;
  self->format = i;
  if (!gst_video_info_from_caps (&info, inverse ? incaps : outcaps))
    return FALSE;
  self->info = info;
  self->r_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 0);
  self->g_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 1);
  self->b_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 2);

  /* outside the domain in which the reference is well defined (see
   * gstmibayerelement.c: set_caps): refuse the caps.  rgb2bayer has no
   * neighbourhood and takes any size */
  if (!inverse && (self->width < 4 || (self->width & 1) || self->height < 3)) {
    GST_WARNING_OBJECT (self, "refusing %dx%d: needs an even width >= 4 and a "
        "height >= 3", self->width, self->height);
    return FALSE;
  }

  /* the context and the output pool are (re)built at the next buffer, on the device
   * that buffer lives on: the frames decide where the conversion runs */
  hb2r_drop_ctx (self);
  hb2r_drop_out_pool (self);
  return TRUE;
}

/* The element works where its input lives: `hipupload device-id=N` (or any other
 * producer of HIPMemory) chooses the GPU, hipbayer2rgb follows -- its context and
 * its output pool are created on the device of the first frame.  device-id >= 0
 * pins the element instead: frames from another GPU are then an error, not a
 * silent cross-device access (the C ABI takes bare device pointers). */
static gboolean
hb2r_device_of (GstMiHipBayer2RGB * self, GstBuffer * inbuf, gint * device)
{
  GstMemory *mem = buffer_hip_memory (inbuf);
  const gint pinned = g_atomic_int_get (&self->device_id);

  if (mem == NULL) {
    GST_ELEMENT_ERROR (self, CORE, NEGOTIATION,
        ("%s needs HIP device memory on both pads", HB2R_LABEL (self)), (NULL));
    return FALSE;
  }
  *device = ((GstMiHipMemory *) mem)->device;
  if (pinned >= 0 && pinned != *device) {
    GST_ELEMENT_ERROR (self, CORE, NEGOTIATION,
        ("%s: buffers live on another GPU than device-id=%d", HB2R_LABEL (self),
            pinned),
        ("input memory on HIP device %d; leave device-id at -1 to follow the "
            "frames, or set the same device-id on hipupload", *device));
    return FALSE;
  }
  return TRUE;
}

/* the context's plan, for the read-only "plan" property and the debug log */
/* frames per launch in the steady state: the launch class the plan is looked up, measured and reported for */
static int
hb2r_launch_frames (GstMiHipBayer2RGB * self)
{
  return MIN (MAX (g_atomic_int_get (&self->batch), 1), HB2R_MAX_BATCH);
}

static void
hb2r_note_plan (GstMiHipBayer2RGB * self)
{
  static const char *const source[] = { "default", "measured", "cached", "set" };
  int variant = 0, band = 0, align = 0, src = -1;
  const char *name;

  /* the plan of the launches this element issues: `batch` frames each (ABI v5: one plan per launch class) */
  (void) mibayer_get_plan_for (self->ctx, hb2r_launch_frames (self), &variant, &band, &align, &src);
  name = mibayer_variant_name (variant);
  GST_OBJECT_LOCK (self);
  g_snprintf (self->plan, sizeof self->plan, "%s band=%d align=%d source=%s",
      name ? name : "?", band == G_MININT32 ? -999 : band, align,
      (src >= 0 && src < 4) ? source[src] : "?");
  GST_OBJECT_UNLOCK (self);
  GST_INFO_OBJECT (self, "launch plan: %s", self->plan);
}

static gboolean
hb2r_ensure_ctx (GstMiHipBayer2RGB * self, gint device)
{
  mibayer_cfg cfg;
  int rc;

  if (self->ctx != NULL && self->ctx_device == device)
    return TRUE;
  hb2r_drop_ctx (self);
  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = self->width;
  cfg.height = self->height;
  cfg.pattern = self->format;
  cfg.r_off = self->r_off;
  cfg.g_off = self->g_off;
  cfg.b_off = self->b_off;
  cfg.device = device;
  cfg.flags = HB2R_INVERSE (self) ? MIBAYER_FLAG_RGB2BAYER : 0;
  rc = mibayer_create (&cfg, &self->ctx);
  if (rc != MIBAYER_OK) {
    self->ctx = NULL;
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
        ("%s: cannot create GPU context on device %d", HB2R_LABEL (self), device),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    return FALSE;
  }
  self->ctx_device = device;
  self->tl_stream = mibayer_ctx_stream (self->ctx);
  self->tl = gst_mi_hip_timeline_for (device, self->tl_stream);
  self->tuned = FALSE;
  hb2r_note_plan (self);
  return TRUE;
}

/* Property "autotune" (additive, default off): the launch plan of this stream geometry is measured once, on the
 * first frame(s) the element converts -- mibayer_autotune_list over the very device buffers it holds (the kernel is
 * idempotent, so the outputs are right whichever candidate ran last) -- and recorded in the library's process-wide
 * plan cache; every later context of that geometry on that device, in this element or another, starts from the
 * measured plan without measuring (the reference sets its ORC programs up once per process the same way,
 * gst/bayer/gstbayerorc-dist.c:321-397).  Costs ~0.1-0.2 s of the first frame's latency.  The caller has ordered
 * the context's stream after the buffers' last accesses. */
static void
hb2r_autotune_once (GstMiHipBayer2RGB * self, const void *const *srcs,
    void *const *dsts, guint n)
{
  char report[1024] = "";
  int src = MIBAYER_PLAN_DEFAULT;
  int rc;

  if (self->tuned)
    return;
  self->tuned = TRUE;           /* from here on the launches may leave the context's stream (hb2r_next_stream) */
  if (HB2R_INVERSE (self))
    return;                     /* rgb2bayer has one launch shape: nothing to measure */
  (void) mibayer_get_plan_for (self->ctx, (int) n, NULL, NULL, NULL, &src);
  if (src != MIBAYER_PLAN_DEFAULT)
    return;                     /* the process cache had a plan for this launch class when the context was created */
  if (mibayer_plan_from_cache (self->ctx) == 1) {
    /* ... or has one now: another element measured since.  The cache is keyed by launch class and reports a hit for
     * EITHER class: only a plan for the class THIS element launches settles the matter -- a batch=16
     * element's measurement says nothing about one frame per launch, and the other way round. */
    (void) mibayer_get_plan_for (self->ctx, (int) n, NULL, NULL, NULL, &src);
    if (src != MIBAYER_PLAN_DEFAULT) {
      hb2r_note_plan (self);
      return;
    }
  }
  if (!HB2R_AUTOTUNE_ON (self))
    return;                     /* nobody asked, and not a batch mode that measures by default */
  rc = mibayer_autotune_list (self->ctx, srcs, dsts, (int) n, report, sizeof report);
  if (rc != MIBAYER_OK) {
    GST_WARNING_OBJECT (self, "plan measurement failed (%s); the default plan stays", mibayer_strerror (rc));
    return;
  }
  hb2r_note_plan (self);
  GST_INFO_OBJECT (self, "plan measured on %u frame(s): %s [%s]", n, self->plan, report);
}

/* output frames come from the element's own device-memory pool on the device of
 * the input (downstream sees the memory's device in the memory itself) */
static GstFlowReturn
hb2r_prepare_output_buffer (GstBaseTransform * trans, GstBuffer * inbuf,
    GstBuffer ** outbuf)
{
  GstMiHipBayer2RGB *self = (GstMiHipBayer2RGB *) trans;
  GstBaseTransformClass *klass = GST_BASE_TRANSFORM_GET_CLASS (trans);
  GstFlowReturn ret;
  gint device;

  *outbuf = NULL;
  if (!hb2r_device_of (self, inbuf, &device))
    return GST_FLOW_ERROR;
  if (self->out_pool == NULL || self->out_pool_device != device) {
    GstCaps *caps = gst_pad_get_current_caps (GST_BASE_TRANSFORM_SRC_PAD (trans));

    hb2r_drop_out_pool (self);
    if (caps == NULL)
      return GST_FLOW_NOT_NEGOTIATED;
    self->out_pool = configured_pool (gst_mi_hip_pool_new (device), caps,
        (guint) HB2R_OUT_BYTES (self), 2);
    gst_caps_unref (caps);
    if (self->out_pool == NULL
        || !gst_buffer_pool_set_active (self->out_pool, TRUE)) {
      hb2r_drop_out_pool (self);
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("%s: cannot set up a device-memory pool on device %d",
              HB2R_LABEL (self), device), ("%s", mibayer_last_hip_error ()));
      return GST_FLOW_ERROR;
    }
    self->out_pool_device = device;
  }
  ret = gst_buffer_pool_acquire_buffer (self->out_pool, outbuf, NULL);
  if (ret != GST_FLOW_OK)
    return ret;
  if (klass->copy_metadata != NULL
      && !klass->copy_metadata (trans, inbuf, *outbuf))
    GST_WARNING_OBJECT (self, "could not copy the buffer metadata");
  return GST_FLOW_OK;
}

/* The C ABI takes bare device pointers: a frame that lives on another GPU, or a
 * buffer smaller than the negotiated frame, would fault on the device.  Maps both
 * memories for device access (no host wait: the caller orders its launch after
 * their queued accesses: gstmihipmemory.h). */
static gboolean
hb2r_map_pair (GstMiHipBayer2RGB * self, GstBuffer * inbuf, GstBuffer * outbuf,
    GstMemory ** in_mem, GstMemory ** out_mem, GstMapInfo * in_map,
    GstMapInfo * out_map)
{
  gint device;

  *in_mem = buffer_hip_memory (inbuf);
  *out_mem = buffer_hip_memory (outbuf);
  if (!hb2r_device_of (self, inbuf, &device))
    return FALSE;
  if (!*out_mem || ((GstMiHipMemory *) * out_mem)->device != device) {
    GST_ELEMENT_ERROR (self, CORE, NEGOTIATION,
        ("%s: input and output frames must live on one GPU", HB2R_LABEL (self)),
        ("input memory on HIP device %d, output %s", device,
            *out_mem ? "on another device" : "not in HIP device memory"));
    return FALSE;
  }
  if (!hb2r_ensure_ctx (self, device))
    return FALSE;
  if (!gst_memory_map (*in_mem, in_map,
          GST_MAP_READ | GST_MAP_HIP | GST_MAP_HIP_ASYNC))
    return FALSE;
  if (!gst_memory_map (*out_mem, out_map,
          GST_MAP_WRITE | GST_MAP_HIP | GST_MAP_HIP_ASYNC)) {
    gst_memory_unmap (*in_mem, in_map);
    return FALSE;
  }
  if (in_map->size < HB2R_IN_BYTES (self) || out_map->size < HB2R_OUT_BYTES (self)) {
    GST_ELEMENT_ERROR (self, STREAM, FORMAT,
        ("%s: device buffer smaller than a %dx%d frame", HB2R_LABEL (self),
            self->width, self->height),
        ("input %" G_GSIZE_FORMAT " bytes (need %" G_GSIZE_FORMAT "), output %"
            G_GSIZE_FORMAT " bytes (need %" G_GSIZE_FORMAT ")", in_map->size,
            HB2R_IN_BYTES (self), out_map->size, HB2R_OUT_BYTES (self)));
    gst_memory_unmap (*out_mem, out_map);
    gst_memory_unmap (*in_mem, in_map);
    return FALSE;
  }
  return TRUE;
}

/* The compute stream of the next launch.  Frames (and list launches) are independent and every buffer is handed over
 * by its own queued accesses (gstmihipmemory.h), so consecutive launches may go round-robin over the device's frame queues (hardware
 * queues of their own): a one-frame launch is a single round of workgroups, and on the other queues the ramp-up of the
 * next launches overlaps the drain of launch n (one 4K frame per launch: 54 -> 66 % of HBM peak, rgb2bayer 55 -> 77 %;
 * list launches of 4: 65 -> 74 %, of 16: 78 -> 81 %; profiles/r05_single_frame.md) -- property `overlap`, OFF BY
 * DEFAULT: those figures are launches without dependencies; in a pipeline every frame waits for an event of its
 * producer's queue and is waited for by its consumer's, and with the frame queues these become cross-queue
 * dependencies per frame where on the device's one shared compute queue they are plain stream order (4K frames from
 * hipbayersrc: 14.5 k fps with the frame queues, 36 k fps without; 1080p 11.6 k vs 53 k; profiles/
 * r05_gst_device_source.log).  When it is on, it acts only UNDER BACK-PRESSURE:
 * while the previous conversion is still running when the next one is issued (a device-resident producer that is
 * faster than one kernel per frame).  A stream whose frames arrive slower than they are converted -- anything fed
 * over PCIe: `hipupload ! hipbayer2rgb` is bound by the 1 B/px upload -- gains nothing from overlapping kernels that
 * never meet, and pays for waking an idle hardware queue per frame (6050 fps on the context's stream against 4900 fps
 * dealt over the frame queues, profiles/r05_gst_pipeline_overlap_ab.log).  The launch that settles the plan (it may run
 * mibayer_autotune_list on the context's stream) stays on the context's stream. */
/* The timeline of the context's stream.  Looked up once per context; a context whose device call failed moves to
 * queues of its own (csrc: mibayer_internal_private_queues), so the stream is compared, not assumed. */
static GstMiHipTimeline *
hb2r_ctx_timeline (GstMiHipBayer2RGB * self, gpointer ctx_stream)
{
  if (G_UNLIKELY (ctx_stream != self->tl_stream || self->tl == NULL)) {
    gst_mi_hip_timeline_unref (self->tl);
    self->tl_stream = ctx_stream;
    self->tl = gst_mi_hip_timeline_for (self->ctx_device, ctx_stream);
  }
  return self->tl;
}

static gpointer
hb2r_next_stream (GstMiHipBayer2RGB * self)
{
  gpointer stream = mibayer_ctx_stream (self->ctx);

  if (self->tuned && g_atomic_int_get (&self->overlap)) {
    const gboolean busy = self->launch_ev != NULL
        && mibayer_dev_event_query (self->launch_ev_device, self->launch_ev) == 0;

    self->busy_run = busy ? self->busy_run + 1 : 0;
    if (self->busy_run >= 2) {
      gpointer fq = mibayer_ctx_frame_queue (self->ctx, (int) (self->frame_no++ % MIBAYER_FRAME_QUEUES));

      if (fq != NULL)
        stream = fq;
    }
  }
  return stream;
}

/* ... and the event that tells the next call whether this launch is still running */
static void
hb2r_note_launch (GstMiHipBayer2RGB * self, gpointer stream)
{
  if (!g_atomic_int_get (&self->overlap))
    return;
  if (self->launch_ev == NULL) {
    self->launch_ev_device = self->ctx_device;
    self->launch_ev = mibayer_dev_event_create (self->ctx_device);
  }
  if (self->launch_ev != NULL
      && mibayer_dev_event_record (self->launch_ev_device, self->launch_ev, stream) != MIBAYER_OK) {
    mibayer_dev_event_destroy (self->launch_ev_device, self->launch_ev);
    self->launch_ev = NULL;
  }
}

static GstFlowReturn
hb2r_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstMiHipBayer2RGB *self = (GstMiHipBayer2RGB *) trans;
  GstMemory *in_mem, *out_mem;
  GstMapInfo in_map, out_map;
  GstFlowReturn ret = GST_FLOW_OK;
  GstMiHipTimeline *tl;
  gpointer stream;
  int rc;

  if (!hb2r_map_pair (self, inbuf, outbuf, &in_mem, &out_mem, &in_map, &out_map))
    return GST_FLOW_ERROR;
  /* Stream-ordered, no host round trip and -- on the context's own stream, the default -- no runtime call besides the
   * launch: the launch is ordered after whatever was last queued on the two memories (nothing to do when that was
   * queued on this stream too), and both are marked on the stream's timeline after it (a counter, gstmihipmemory.h).
   * The next user either orders its own stream after that access or -- any plain map, hipdownload, a CPU map --
   * waits for it on the host; the fence is recorded then, by them. */
  stream = hb2r_next_stream (self);
  tl = stream == mibayer_ctx_stream (self->ctx) ? hb2r_ctx_timeline (self, stream) : NULL;
  if (!(tl ? gst_mi_hip_memory_order_after_tl ((GstMiHipMemory *) in_mem, tl)
          && gst_mi_hip_memory_order_after_tl ((GstMiHipMemory *) out_mem, tl)
          : gst_mi_hip_memory_order_after ((GstMiHipMemory *) in_mem, stream)
          && gst_mi_hip_memory_order_after ((GstMiHipMemory *) out_mem,
              stream))) {
    /* could not order on the device: fall back to waiting on the host */
    gst_mi_hip_memory_wait ((GstMiHipMemory *) in_mem);
    gst_mi_hip_memory_wait ((GstMiHipMemory *) out_mem);
  }
  {
    const void *one_src = in_map.data;
    void *one_dst = out_map.data;

    hb2r_autotune_once (self, &one_src, &one_dst, 1);
  }
  /* device-resident call: no PCIe traffic at all */
  rc = mibayer_process_device (self->ctx, in_map.data, 0, out_map.data, 0, 1,
      stream);
  if (rc == MIBAYER_OK) {
    if (tl) {
      gst_mi_hip_memory_mark_access_tl ((GstMiHipMemory *) in_mem, tl);
      gst_mi_hip_memory_mark_access_tl ((GstMiHipMemory *) out_mem, tl);
    } else {
      gst_mi_hip_memory_mark_access ((GstMiHipMemory *) in_mem, stream);
      gst_mi_hip_memory_mark_access ((GstMiHipMemory *) out_mem, stream);
    }
    hb2r_note_launch (self, stream);
  }
  if (rc != MIBAYER_OK) {
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
        ("%s: GPU conversion failed", HB2R_LABEL (self)),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    ret = GST_FLOW_ERROR;
  }
  gst_memory_unmap (out_mem, &out_map);
  gst_memory_unmap (in_mem, &in_map);
  return ret;
}

/* ---- batch mode: one launch over up to `batch` queued frames ------------------ */
/* A 4K frame is ~7 us of kernel -- about what one launch costs to issue -- so a
 * device-resident pipeline that converts buffer by buffer is bound by launch
 * latency, not by HBM.  With batch=N the element parks N input buffers (each
 * with its output buffer already allocated), converts them with ONE list launch
 * (mibayer_process_device_list: every frame is its own allocation) and hands
 * the outputs on in order.  The price is latency: up to N-1 frames wait; events
 * that must not overtake buffers (caps, segment, gap, EOS) convert and push what
 * is waiting first, a flush drops it.  Same stream-ordered hand-over as the
 * frame-by-frame mode: the launch is ordered after every memory's last access
 * and every memory is marked after it. */
static GstFlowReturn
hb2r_convert_waiting (GstMiHipBayer2RGB * self)
{
  const void *srcs[HB2R_MAX_BATCH];
  void *dsts[HB2R_MAX_BATCH];
  GstMemory *in_mem[HB2R_MAX_BATCH], *out_mem[HB2R_MAX_BATCH];
  GstMapInfo in_map[HB2R_MAX_BATCH], out_map[HB2R_MAX_BATCH];
  Hb2rPair *pairs[HB2R_MAX_BATCH];
  GstFlowReturn ret = GST_FLOW_OK;
  GstMiHipTimeline *tl;
  gpointer stream;
  guint n = 0, i;
  int rc;

  while (n < HB2R_MAX_BATCH && !g_queue_is_empty (&self->waiting)) {
    Hb2rPair *pair = g_queue_pop_head (&self->waiting);

    if (ret == GST_FLOW_OK && hb2r_map_pair (self, pair->in, pair->out,
            &in_mem[n], &out_mem[n], &in_map[n], &out_map[n])) {
      pairs[n++] = pair;
    } else {
      ret = GST_FLOW_ERROR;
      gst_buffer_unref (pair->in);
      gst_buffer_unref (pair->out);
      g_free (pair);
    }
  }
  if (n == 0)
    return ret;
  stream = hb2r_next_stream (self);     /* list launches, too, go over the frame queues under back-pressure */
  tl = gst_mi_hip_timeline_for (self->ctx_device, stream);       /* one look-up per launch */
  for (i = 0; i < n; i++) {
    if (!gst_mi_hip_memory_order_after_tl ((GstMiHipMemory *) in_mem[i], tl)
        || !gst_mi_hip_memory_order_after_tl ((GstMiHipMemory *) out_mem[i], tl)) {
      gst_mi_hip_memory_wait ((GstMiHipMemory *) in_mem[i]);
      gst_mi_hip_memory_wait ((GstMiHipMemory *) out_mem[i]);
    }
    srcs[i] = in_map[i].data;
    dsts[i] = out_map[i].data;
  }
  if (ret == GST_FLOW_OK && n >= (guint) MIN (MAX (g_atomic_int_get (&self->batch), 1), HB2R_MAX_BATCH))
    hb2r_autotune_once (self, srcs, dsts, n);   /* on a full batch: what the steady state launches */
  rc = ret == GST_FLOW_OK
      ? mibayer_process_device_list (self->ctx, srcs, dsts, (int) n, stream)
      : MIBAYER_OK;
  /* 2 n counter increments; whoever needs a fence behind this launch records ONE for all 2 n memories */
  for (i = 0; i < n && rc == MIBAYER_OK; i++) {
    gst_mi_hip_memory_mark_access_tl ((GstMiHipMemory *) in_mem[i], tl);
    gst_mi_hip_memory_mark_access_tl ((GstMiHipMemory *) out_mem[i], tl);
  }
  gst_mi_hip_timeline_unref (tl);
  if (rc == MIBAYER_OK && ret == GST_FLOW_OK)
    hb2r_note_launch (self, stream);
  if (rc != MIBAYER_OK) {
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
        ("%s: GPU conversion failed", HB2R_LABEL (self)),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    ret = GST_FLOW_ERROR;
  }
  for (i = 0; i < n; i++) {
    gst_memory_unmap (out_mem[i], &out_map[i]);
    gst_memory_unmap (in_mem[i], &in_map[i]);
    gst_buffer_unref (pairs[i]->in);
    if (ret == GST_FLOW_OK)
      g_queue_push_tail (&self->ready, pairs[i]->out);
    else
      gst_buffer_unref (pairs[i]->out);
    g_free (pairs[i]);
  }
  return ret;
}

static GstFlowReturn
hb2r_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstMiHipBayer2RGB *self = (GstMiHipBayer2RGB *) trans;
  GstBaseTransformClass *klass = GST_BASE_TRANSFORM_GET_CLASS (trans);
  GstBuffer *inbuf;
  GstFlowReturn ret = GST_FLOW_OK;
  const gint batch = g_atomic_int_get (&self->batch);

  if (batch <= 1 && g_queue_is_empty (&self->waiting)
      && g_queue_is_empty (&self->ready))
    return GST_BASE_TRANSFORM_CLASS (gst_mi_hip_bayer2rgb_parent_class)->generate_output
        (trans, outbuf);

  *outbuf = NULL;
  inbuf = trans->queued_buf;
  trans->queued_buf = NULL;
  if (inbuf != NULL) {
    GstBuffer *out = NULL;
    Hb2rPair *pair;

    ret = klass->prepare_output_buffer (trans, inbuf, &out);
    if (ret != GST_FLOW_OK || out == NULL) {
      gst_buffer_unref (inbuf);
      return ret == GST_FLOW_OK ? GST_FLOW_ERROR : ret;
    }
    /* frames of one list launch share a context: a frame from another GPU
     * (upstream switched devices) first converts what is waiting */
    {
      gint device = 0;

      if (hb2r_device_of (self, inbuf, &device) && self->ctx != NULL
          && self->ctx_device != device && !g_queue_is_empty (&self->waiting))
        ret = hb2r_convert_waiting (self);
      if (ret != GST_FLOW_OK) {
        gst_buffer_unref (inbuf);
        gst_buffer_unref (out);
        return ret;
      }
    }
    pair = g_new0 (Hb2rPair, 1);
    pair->in = inbuf;
    pair->out = out;
    g_queue_push_tail (&self->waiting, pair);
    /* The first frame after a start or a flush goes out alone: sinks preroll on
     * it, and a pipeline does not reach PLAYING before they have -- parking it
     * until N-1 more arrive can deadlock against upstream queues that fill up
     * while another branch's sink sits prerolled (tee ! queue ! ...). */
    if ((gint) g_queue_get_length (&self->waiting)
        >= (self->prerolled ? MIN (MAX (batch, 1), HB2R_MAX_BATCH) : 1))
      ret = hb2r_convert_waiting (self);
  }
  /* the base class calls again for as long as a buffer comes out */
  if (ret == GST_FLOW_OK) {
    *outbuf = g_queue_pop_head (&self->ready);
    if (*outbuf != NULL)
      self->prerolled = TRUE;
  }
  return ret;
}

static gboolean
hb2r_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  GstMiHipBayer2RGB *self = (GstMiHipBayer2RGB *) trans;
  GstBuffer *buf;

  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_CAPS:
    case GST_EVENT_SEGMENT:
    case GST_EVENT_GAP:
    case GST_EVENT_EOS:
      /* waiting frames precede the event */
      while (!g_queue_is_empty (&self->waiting)
          && hb2r_convert_waiting (self) == GST_FLOW_OK);
      while ((buf = g_queue_pop_head (&self->ready)) != NULL)
        gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (trans), buf);
      break;
    case GST_EVENT_FLUSH_STOP:
      hb2r_drop_queued (self);
      self->prerolled = FALSE;
      break;
    default:
      break;
  }
  return GST_BASE_TRANSFORM_CLASS (gst_mi_hip_bayer2rgb_parent_class)->sink_event
      (trans, event);
}

static gboolean
hb2r_stop (GstBaseTransform * trans)
{
  hb2r_drop_queued ((GstMiHipBayer2RGB *) trans);
  hb2r_drop_ctx ((GstMiHipBayer2RGB *) trans);
  hb2r_drop_out_pool ((GstMiHipBayer2RGB *) trans);
  ((GstMiHipBayer2RGB *) trans)->prerolled = FALSE;
  return TRUE;
}

static void
gst_mi_hip_bayer2rgb_class_init (GstMiHipBayer2RGBClass * klass)
{
  GObjectClass *object_class = G_OBJECT_CLASS (klass);
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *transform_class = GST_BASE_TRANSFORM_CLASS (klass);

  klass->inverse = FALSE;
  klass->label = "hipbayer2rgb";
  object_class->set_property = hb2r_set_property;
  object_class->get_property = hb2r_get_property;
  object_class->finalize = hb2r_finalize;
  g_object_class_install_property (object_class, PROP_DEVICE_ID,
      g_param_spec_int ("device-id", "Device ID",
          "HIP ordinal of the MI355X; -1 = convert on whichever GPU the incoming "
          "frames live on (hipupload's device-id decides).  A value >= 0 pins the "
          "element: frames from another GPU are then refused",
          -1, G_MAXINT, -1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_BATCH,
      g_param_spec_int ("batch", "Frames per launch",
          "Convert this many queued frames with ONE kernel launch (each frame "
          "stays its own buffer); 1 = a launch per frame, no added latency.  A "
          "launch per 4K frame costs about as much as the kernel runs.  The "
          "first frame after a start or flush is never held back (preroll)",
          1, HB2R_MAX_BATCH, 1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_AUTOTUNE,
      g_param_spec_boolean ("autotune", "Measure the launch plan",
          "Measure the kernel launch plan (tile shape, block order, store policy) "
          "once on the first frame(s) of a stream geometry and record it in the "
          "process-wide plan cache; later contexts of that geometry on that GPU "
          "-- in any element -- take the measured plan without measuring.  Off: "
          "the static default plan, or the cached one if some element measured "
          "before.  Unless set, on for batch >= 4 (the first full batch measures) "
          "and off below.  No effect on hiprgb2bayer",
          FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_OVERLAP,
      g_param_spec_boolean ("overlap", "Overlap consecutive frames",
          "While frames arrive faster than they are converted, deal consecutive "
          "launches (one frame each, or one batch each) round-robin over four compute "
          "queues (hardware queues of their own), so that the start of the next "
          "launches overlaps the tail of launch n (a one-frame launch never reaches "
          "a steady state by itself).  Frames are handed over by per-buffer events "
          "either way; off (the default) = every launch behind the previous one on "
          "the device's shared compute queue.  Off by default because every frame of "
          "an element has dependencies on other queues (its producer's and its "
          "consumer's events), and those cost more than the overlap gains: 4K from a "
          "device-resident source 14.5 k fps with, 36 k fps without",
          FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_PLAN,
      g_param_spec_string ("plan", "Launch plan",
          "The launch plan of the current stream and where it came from "
          "(default / measured / cached)", "",
          G_PARAM_READABLE | G_PARAM_STATIC_STRINGS));
  xfer_add_templates (element_class, HB2R_SINK_CAPS, HB2R_SRC_CAPS);
  gst_element_class_set_static_metadata (element_class,
      "Bayer to RGB decoder (HIP device memory)", "Filter/Converter/Video",
      "Converts video/x-bayer to video/x-raw without leaving MI355X memory",
      "gst-plugins-bad_amd");
  transform_class->transform_caps = GST_DEBUG_FUNCPTR (hb2r_transform_caps);
  transform_class->get_unit_size = GST_DEBUG_FUNCPTR (hb2r_get_unit_size);
  transform_class->set_caps = GST_DEBUG_FUNCPTR (hb2r_set_caps);
  transform_class->prepare_output_buffer =
      GST_DEBUG_FUNCPTR (hb2r_prepare_output_buffer);
  transform_class->transform = GST_DEBUG_FUNCPTR (hb2r_transform);
  transform_class->generate_output = GST_DEBUG_FUNCPTR (hb2r_generate_output);
  transform_class->sink_event = GST_DEBUG_FUNCPTR (hb2r_sink_event);
  transform_class->stop = GST_DEBUG_FUNCPTR (hb2r_stop);
}

static void
gst_mi_hip_bayer2rgb_init (GstMiHipBayer2RGB * self)
{
  gst_video_info_init (&self->info);
  self->device_id = -1;
  self->ctx = NULL;
  self->ctx_device = 0;
  self->tl = NULL;
  self->tl_stream = NULL;
  self->out_pool = NULL;
  self->out_pool_device = 0;
  self->batch = 1;
  self->autotune = -1;
  self->overlap = 0;
  self->frame_no = 0;
  self->launch_ev = NULL;
  self->launch_ev_device = 0;
  self->busy_run = 0;
  self->tuned = FALSE;
  self->plan[0] = '\0';
  self->prerolled = FALSE;
  g_queue_init (&self->waiting);
  g_queue_init (&self->ready);
}

/* ---- hiprgb2bayer: the same element, pad roles swapped ------------------------------ */

typedef GstMiHipBayer2RGB GstMiHipRGB2Bayer;
typedef GstMiHipBayer2RGBClass GstMiHipRGB2BayerClass;
GType gst_mi_hip_rgb2bayer_get_type (void);
G_DEFINE_TYPE (GstMiHipRGB2Bayer, gst_mi_hip_rgb2bayer,
    gst_mi_hip_bayer2rgb_get_type ());

/* the reference's rgb2bayer takes ARGB only (gst/bayer/gstrgb2bayer.c:59-63) */
#define HR2B_SINK_CAPS GST_VIDEO_CAPS_MAKE_WITH_FEATURES ( \
    GST_CAPS_FEATURE_MEMORY_HIP, "ARGB")
#define HR2B_SRC_CAPS HIP_CAPS ("video/x-bayer") \
  ",format=(string){bggr,grbg,gbrg,rggb}," \
  "width=(int)[1,MAX],height=(int)[1,MAX],framerate=(fraction)[0/1,MAX]"

static void
gst_mi_hip_rgb2bayer_class_init (GstMiHipRGB2BayerClass * klass)
{
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);

  klass->inverse = TRUE;
  klass->label = "hiprgb2bayer";
  xfer_add_templates (element_class, HR2B_SINK_CAPS, HR2B_SRC_CAPS);    /* replace the parent's */
  gst_element_class_set_static_metadata (element_class,
      "RGB to Bayer converter (HIP device memory)", "Filter/Converter/Video",
      "Converts video/x-raw to video/x-bayer without leaving MI355X memory; "
      "batch=N converts N buffers with one launch",
      "gst-plugins-bad_amd");
}

static void
gst_mi_hip_rgb2bayer_init (GstMiHipRGB2Bayer * self)
{
}

/* ======================================================================== */

#ifndef PACKAGE
#define PACKAGE "gst-plugins-bad_amd"
#endif
#ifndef VERSION
#define VERSION "0.1.0"
#endif

GType gst_mi_hip_bayer_src_get_type (void);    /* gstmihipbayersrc.c */

static gboolean
plugin_init (GstPlugin * plugin)
{
  GST_DEBUG_CATEGORY_INIT (gst_mi_hip_debug, "mihip", 0,
      "MI355X device-memory elements");
  return gst_element_register (plugin, "hipupload", GST_RANK_NONE,
      gst_mi_hip_upload_get_type ())
      && gst_element_register (plugin, "hipdownload", GST_RANK_NONE,
      gst_mi_hip_download_get_type ())
      && gst_element_register (plugin, "hipbayer2rgb", GST_RANK_NONE,
      gst_mi_hip_bayer2rgb_get_type ())
      && gst_element_register (plugin, "hiprgb2bayer", GST_RANK_NONE,
      gst_mi_hip_rgb2bayer_get_type ())
      && gst_element_register (plugin, "hipbayersrc", GST_RANK_NONE,
      gst_mi_hip_bayer_src_get_type ());
}

GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, mihip,
    "MI355X device-memory video elements", plugin_init, VERSION, "LGPL",
    "gst-plugins-bad_amd (MI355X-native bayer2rgb)",
    "https://gstreamer.freedesktop.org/")
