/* bayer2rgb -- MI355X-native element.
 *
 * Everything a neighbouring element can observe is kept identical to the
 * reference element (gst-plugins-bad 1.19.2, gst/bayer/gstbayer2rgb.c):
 *   factory name / rank / GType name            :148-150
 *   element metadata (long name, klass, ...)    :180-183
 *   pad templates                               :134-138, :185-190
 *   transform_caps / get_unit_size / set_caps   :289-322 / :324-352 / :237-276
 *   1-in/1-out synchronous transform            :456-487  (the default mode)
 * What changes is below the transform vfunc: the reference calls its CPU/ORC
 * frame loop gst_bayer2rgb_process (:475-477); this element hands the mapped
 * pointers and strides to the HIP path through the C ABI of mibayer.h.  There
 * is no CPU fallback: without a usable MI355X the element posts a RESOURCE
 * error instead of converting on the host.
 *
 * Additive, optional behaviour:
 *   - hipHostMalloc-pinned buffer pools are proposed upstream and used
 *     downstream when nobody offers a pool (propose/decide_allocation), so the
 *     H2D/D2H copies are asynchronous DMA;
 *   - `inflight` > 1 and/or `devices` switch to a queued mode: input buffers
 *     are submitted to a round-robin pool of GPUs (frame g -> devices[g % N])
 *     and outputs are pushed in order as they complete; pending frames are
 *     drained before EOS / caps / segment events and dropped on flush.
 *     In-tree precedent for queueing in submit_input_buffer/generate_output:
 *     sys/va/gstvadeinterlace.c:186-232, :467-531.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <stdlib.h>
#include <string.h>

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "mibayer.h"
#include "gstmibayer.h"
#include "gstmihostpool.h"

/* ---- type (private to this file) ------------------------------------------------ */

#define GST_TYPE_BAYER2RGB (gst_bayer2rgb_get_type ())
#define GST_BAYER2RGB(obj) \
  (G_TYPE_CHECK_INSTANCE_CAST ((obj), GST_TYPE_BAYER2RGB, GstBayer2RGB))
#define GST_IS_BAYER2RGB(obj) \
  (G_TYPE_CHECK_INSTANCE_TYPE ((obj), GST_TYPE_BAYER2RGB))

typedef struct _GstBayer2RGB GstBayer2RGB;
typedef struct _GstBayer2RGBClass GstBayer2RGBClass;

/* Same negotiated state as the reference's struct _GstBayer2RGB
 * (gstbayer2rgb.c:115-127) plus the handle of the GPU context that replaces
 * gst_bayer2rgb_process. */
struct _GstBayer2RGB
{
  GstBaseTransform basetransform;

  GstVideoInfo info;            /* output video info */
  gint width;
  gint height;
  gint r_off;                   /* byte offset of red in an output pixel */
  gint g_off;
  gint b_off;
  gint format;                  /* mibayer_pattern == reference enum :95-101 */

  /* additive, optional properties (the reference has none); the defaults give
   * the reference's behaviour: one device, strictly 1-in/1-out synchronous */
  gint device_id;
  gchar *devices;               /* "0,1,2,..." round-robin frame sharding; NULL = device-id */
  gint inflight;                /* frames in flight per device; 1 = synchronous */
  gboolean use_hipgraph;
  gboolean pinned_pool;

  /* GPU side: one shard (mibayer_ctx) per device behind a round-robin pool;
   * (re)created when caps or the mapped output stride change */
  mibayer_pool *pool;
  gint pool_dst_stride;
  gint capacity;                /* frames the pool may hold in flight */
  GQueue pending;               /* Bayer2RGBPending*, oldest first */
};

struct _GstBayer2RGBClass
{
  GstBaseTransformClass parent;
};

GType gst_bayer2rgb_get_type (void);

GST_DEBUG_CATEGORY_STATIC (gst_bayer2rgb_debug);
#define GST_CAT_DEFAULT gst_bayer2rgb_debug

/* identical strings to the reference, order matters: the first src format
 * (RGBx) is what default negotiation fixates to */
#define BAYER2RGB_SRC_CAPS \
  GST_VIDEO_CAPS_MAKE ("{ RGBx, xRGB, BGRx, xBGR, RGBA, ARGB, BGRA, ABGR }")
#define BAYER2RGB_SINK_CAPS \
  "video/x-bayer,format=(string){bggr,grbg,gbrg,rggb}," \
  "width=(int)[1,MAX],height=(int)[1,MAX],framerate=(fraction)[0/1,MAX]"

enum
{
  PROP_0,
  PROP_DEVICE_ID,
  PROP_DEVICES,
  PROP_INFLIGHT,
  PROP_HIPGRAPH,
  PROP_PINNED_POOL
};

#define DEFAULT_DEVICE_ID 0
#define DEFAULT_INFLIGHT 1
#define DEFAULT_HIPGRAPH FALSE
#define DEFAULT_PINNED_POOL TRUE

/* one frame between submit and wait: both buffers stay mapped until the GPU
 * has written the output */
typedef struct
{
  GstBuffer *inbuf;
  GstBuffer *outbuf;
  GstMapInfo in_map;
  GstVideoFrame out_frame;
  gboolean owns_outbuf;         /* queued mode: we hold the only reference to outbuf;
                                   synchronous mode: the base class owns it */
} Bayer2RGBPending;

G_DEFINE_TYPE (GstBayer2RGB, gst_bayer2rgb, GST_TYPE_BASE_TRANSFORM);

/* ---- GPU pool --------------------------------------------------------------- */

/* unmap and free the bookkeeping; the output buffer is unreffed only if this
 * entry owns it and the caller does not take it over */
static void
pending_release (Bayer2RGBPending * p, gboolean caller_takes_outbuf)
{
  gst_video_frame_unmap (&p->out_frame);
  gst_buffer_unmap (p->inbuf, &p->in_map);
  gst_buffer_unref (p->inbuf);
  if (p->owns_outbuf && !caller_takes_outbuf)
    gst_buffer_unref (p->outbuf);
  g_free (p);
}

/* wait for everything in flight; push it downstream (push == TRUE) or drop it */
static GstFlowReturn
bayer2rgb_drain (GstBayer2RGB * self, gboolean push)
{
  GstFlowReturn ret = GST_FLOW_OK;
  Bayer2RGBPending *p;

  while ((p = g_queue_pop_head (&self->pending)) != NULL) {
    GstBuffer *out = p->outbuf;
    gboolean owned = p->owns_outbuf;
    int rc = self->pool ? mibayer_pool_wait (self->pool, NULL) : MIBAYER_OK;
    gboolean do_push = push && owned && rc == MIBAYER_OK && ret == GST_FLOW_OK;

    pending_release (p, do_push);
    if (rc != MIBAYER_OK) {
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("bayer2rgb: GPU conversion failed"),
          ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
      ret = GST_FLOW_ERROR;
    } else if (do_push) {
      ret = gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (self), out);
    }
  }
  return ret;
}

static void
bayer2rgb_drop_pool (GstBayer2RGB * self)
{
  bayer2rgb_drain (self, FALSE);
  if (self->pool) {
    mibayer_pool_destroy (self->pool);
    self->pool = NULL;
  }
  self->pool_dst_stride = 0;
  self->capacity = 0;
}

/* reference gst_bayer2rgb_reset, :278-287 */
static void
bayer2rgb_clear_negotiation (GstBayer2RGB * self)
{
  self->width = 0;
  self->height = 0;
  self->r_off = 0;
  self->g_off = 0;
  self->b_off = 0;
  self->format = MIBAYER_BGGR;
  gst_video_info_init (&self->info);
}

static gboolean
bayer2rgb_parse_devices (GstBayer2RGB * self, mibayer_pool_cfg * pc)
{
  gchar **tok, **t;

  pc->ndevices = 0;
  if (self->devices == NULL || self->devices[0] == '\0') {
    pc->devices[pc->ndevices++] = self->device_id;
    return TRUE;
  }
  tok = g_strsplit_set (self->devices, ",;: ", -1);
  for (t = tok; *t != NULL; t++) {
    gchar *end = NULL;
    glong v;

    if (**t == '\0')
      continue;
    v = strtol (*t, &end, 10);
    if (end == *t || *end != '\0' || v < 0 || pc->ndevices >= MIBAYER_MAX_SHARDS) {
      g_strfreev (tok);
      return FALSE;
    }
    pc->devices[pc->ndevices++] = (int32_t) v;
  }
  g_strfreev (tok);
  return pc->ndevices > 0;
}

static gboolean
bayer2rgb_ensure_pool (GstBayer2RGB * self, gint dst_stride)
{
  mibayer_pool_cfg pc;
  int rc;

  if (self->pool && self->pool_dst_stride == dst_stride)
    return TRUE;
  bayer2rgb_drop_pool (self);

  memset (&pc, 0, sizeof pc);
  pc.struct_size = sizeof pc;
  pc.stream.struct_size = sizeof pc.stream;
  pc.stream.width = self->width;
  pc.stream.height = self->height;
  pc.stream.src_stride = GST_ROUND_UP_4 (self->width);  /* reference :477 */
  pc.stream.dst_stride = dst_stride;                    /* reference :476 */
  pc.stream.pattern = self->format;
  pc.stream.r_off = self->r_off;
  pc.stream.g_off = self->g_off;
  pc.stream.b_off = self->b_off;
  pc.stream.inflight = self->inflight;
  pc.stream.flags = self->use_hipgraph ? MIBAYER_FLAG_HIPGRAPH : 0;
  if (!bayer2rgb_parse_devices (self, &pc)) {
    GST_ELEMENT_ERROR (self, LIBRARY, SETTINGS,
        ("bayer2rgb: cannot parse devices=\"%s\"", self->devices), (NULL));
    return FALSE;
  }

  rc = mibayer_pool_create (&pc, &self->pool);
  if (rc != MIBAYER_OK) {
    self->pool = NULL;
    if (rc == MIBAYER_ERR_NO_DEVICE) {
      GST_ELEMENT_ERROR (self, RESOURCE, NOT_FOUND,
          ("bayer2rgb: no usable MI355X / HIP device (device-id=%d devices=%s)",
              self->device_id, self->devices ? self->devices : ""),
          ("%s; this element has no CPU path", mibayer_strerror (rc)));
    } else if (rc == MIBAYER_ERR_GEOMETRY) {
      GST_ELEMENT_ERROR (self, STREAM, FORMAT,
          ("bayer2rgb: unsupported frame geometry %dx%d", self->width,
              self->height), ("%s", mibayer_strerror (rc)));
    } else {
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("bayer2rgb: cannot create GPU context"),
          ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    }
    return FALSE;
  }
  self->pool_dst_stride = dst_stride;
  self->capacity = mibayer_pool_capacity (self->pool);
  GST_DEBUG_OBJECT (self, "GPU pool: %d device(s), %d frame(s) in flight, "
      "%dx%d pattern %d stride %d%s", pc.ndevices, self->capacity, self->width,
      self->height, self->format, dst_stride,
      self->use_hipgraph ? ", hipGraph per frame" : "");
  return TRUE;
}

static inline gboolean
bayer2rgb_is_queued_mode (GstBayer2RGB * self)
{
  return self->inflight > 1
      || (self->devices != NULL && strchr (self->devices, ',') != NULL);
}

/* ---- GObject ----------------------------------------------------------------- */

static void
gst_bayer2rgb_set_property (GObject * object, guint prop_id,
    const GValue * value, GParamSpec * pspec)
{
  GstBayer2RGB *self = GST_BAYER2RGB (object);

  GST_OBJECT_LOCK (self);
  switch (prop_id) {
    case PROP_DEVICE_ID:
      self->device_id = g_value_get_int (value);
      break;
    case PROP_DEVICES:
      g_free (self->devices);
      self->devices = g_value_dup_string (value);
      break;
    case PROP_INFLIGHT:
      self->inflight = g_value_get_int (value);
      break;
    case PROP_HIPGRAPH:
      self->use_hipgraph = g_value_get_boolean (value);
      break;
    case PROP_PINNED_POOL:
      self->pinned_pool = g_value_get_boolean (value);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
  GST_OBJECT_UNLOCK (self);
}

static void
gst_bayer2rgb_get_property (GObject * object, guint prop_id, GValue * value,
    GParamSpec * pspec)
{
  GstBayer2RGB *self = GST_BAYER2RGB (object);

  switch (prop_id) {
    case PROP_DEVICE_ID:
      g_value_set_int (value, self->device_id);
      break;
    case PROP_DEVICES:
      g_value_set_string (value, self->devices);
      break;
    case PROP_INFLIGHT:
      g_value_set_int (value, self->inflight);
      break;
    case PROP_HIPGRAPH:
      g_value_set_boolean (value, self->use_hipgraph);
      break;
    case PROP_PINNED_POOL:
      g_value_set_boolean (value, self->pinned_pool);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
}

static void
gst_bayer2rgb_finalize (GObject * object)
{
  GstBayer2RGB *self = GST_BAYER2RGB (object);

  bayer2rgb_drop_pool (self);
  g_free (self->devices);
  self->devices = NULL;
  G_OBJECT_CLASS (gst_bayer2rgb_parent_class)->finalize (object);
}

/* ---- caps (identical semantics to the reference) ------------------------------ */

/* reference :289-322 -- the bayer side and the raw side differ only in the
 * media type name and in the fields that describe the pixel encoding */
static GstCaps *
gst_bayer2rgb_transform_caps (GstBaseTransform * base,
    GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  GstCaps *result = gst_caps_copy (caps);
  guint i, n = gst_caps_get_size (result);

  for (i = 0; i < n; i++) {
    GstStructure *s = gst_caps_get_structure (result, i);

    if (direction == GST_PAD_SINK) {
      gst_structure_set_name (s, "video/x-raw");
      gst_structure_remove_field (s, "format");
    } else {
      gst_structure_set_name (s, "video/x-bayer");
      gst_structure_remove_fields (s, "format", "colorimetry", "chroma-site",
          NULL);
    }
  }
  if (filter) {
    GstCaps *unfiltered = result;

    result = gst_caps_intersect_full (filter, unfiltered,
        GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (unfiltered);
  }
  GST_DEBUG_OBJECT (base, "transformed %" GST_PTR_FORMAT " into %"
      GST_PTR_FORMAT, caps, result);
  return result;
}

/* reference :324-352 */
static gboolean
gst_bayer2rgb_get_unit_size (GstBaseTransform * base, GstCaps * caps,
    gsize * size)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w, h;

  if (!gst_structure_get_int (s, "width", &w)
      || !gst_structure_get_int (s, "height", &h)) {
    GST_ELEMENT_ERROR (base, CORE, NEGOTIATION, (NULL),
        ("Incomplete caps, some required field missing"));
    return FALSE;
  }
  if (gst_structure_has_name (s, "video/x-raw"))
    *size = (gsize) w * h * 4;            /* always 32 bits per pixel */
  else
    *size = (gsize) GST_ROUND_UP_4 (w) * h;     /* 8-bit mosaic, rows padded to 4 */
  return TRUE;
}

/* reference :237-276 */
static gboolean
gst_bayer2rgb_set_caps (GstBaseTransform * base, GstCaps * incaps,
    GstCaps * outcaps)
{
  static const struct
  {
    const gchar *name;
    gint pattern;
  } orders[] = {
    {"bggr", MIBAYER_BGGR}, {"gbrg", MIBAYER_GBRG},
    {"grbg", MIBAYER_GRBG}, {"rggb", MIBAYER_RGGB}
  };
  GstBayer2RGB *self = GST_BAYER2RGB (base);
  GstStructure *s = gst_caps_get_structure (incaps, 0);
  const gchar *order;
  GstVideoInfo info;
  guint i;

  GST_DEBUG_OBJECT (self, "in caps %" GST_PTR_FORMAT " out caps %"
      GST_PTR_FORMAT, incaps, outcaps);

  gst_structure_get_int (s, "width", &self->width);
  gst_structure_get_int (s, "height", &self->height);

  order = gst_structure_get_string (s, "format");
  if (order == NULL)
    return FALSE;
  for (i = 0; i < G_N_ELEMENTS (orders); i++) {
    if (g_str_equal (order, orders[i].name))
      break;
  }
  if (i == G_N_ELEMENTS (orders))
    return FALSE;
  self->format = orders[i].pattern;

  /* where R, G and B live inside the 4-byte output pixel */
  if (!gst_video_info_from_caps (&info, outcaps))
    return FALSE;
  self->r_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 0);
  self->g_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 1);
  self->b_off = GST_VIDEO_INFO_COMP_OFFSET (&info, 2);
  self->info = info;

  /* geometry changed: the GPU pool is rebuilt on the next buffer, once the
   * mapped output stride is known */
  bayer2rgb_drop_pool (self);
  return TRUE;
}

/* ---- allocation: pinned pools -------------------------------------------------- */

static GstBufferPool *
bayer2rgb_make_pinned_pool (GstBayer2RGB * self, GstCaps * caps, guint size,
    guint min)
{
  GstBufferPool *pool;
  GstStructure *config;

  if (mibayer_device_count () <= 0)
    return NULL;
  pool = gst_mi_host_pool_new ();
  config = gst_buffer_pool_get_config (pool);
  gst_buffer_pool_config_set_params (config, caps, size, min, 0);
  if (!gst_buffer_pool_set_config (pool, config)) {
    gst_object_unref (pool);
    return NULL;
  }
  GST_DEBUG_OBJECT (self, "pinned pool: %u bytes per buffer, min %u", size,
      min);
  return pool;
}

/* upstream asks how to allocate the mosaic buffers it will send us */
static gboolean
gst_bayer2rgb_propose_allocation (GstBaseTransform * base,
    GstQuery * decide_query, GstQuery * query)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);
  GstCaps *caps = NULL;
  gboolean need_pool = FALSE;
  gsize size = 0;

  if (!GST_BASE_TRANSFORM_CLASS (gst_bayer2rgb_parent_class)->propose_allocation
      (base, decide_query, query))
    return FALSE;
  if (!self->pinned_pool)
    return TRUE;
  gst_query_parse_allocation (query, &caps, &need_pool);
  if (caps == NULL || !gst_bayer2rgb_get_unit_size (base, caps, &size))
    return TRUE;
  {
    mibayer_pool_cfg pc;
    guint min;
    GstBufferPool *pool;

    if (!bayer2rgb_parse_devices (self, &pc))
      pc.ndevices = 1;
    /* every frame in flight keeps its input buffer mapped */
    min = (guint) (MAX (self->inflight, 1) * pc.ndevices + 2);
    pool =
        bayer2rgb_make_pinned_pool (self, caps, (guint) size, min);

    if (pool) {
      gst_query_add_allocation_pool (query, pool, (guint) size, min, 0);
      gst_object_unref (pool);
      GST_DEBUG_OBJECT (self, "proposed a pinned input pool upstream");
    }
  }
  return TRUE;
}

/* downstream answered our allocation query: if it brought no pool of its own,
 * allocate the RGB buffers from pinned memory */
static gboolean
gst_bayer2rgb_decide_allocation (GstBaseTransform * base, GstQuery * query)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);

  if (self->pinned_pool && gst_query_get_n_allocation_pools (query) == 0) {
    GstCaps *caps = NULL;
    gsize size = 0;

    gst_query_parse_allocation (query, &caps, NULL);
    if (caps != NULL && gst_bayer2rgb_get_unit_size (base, caps, &size)) {
      guint min = (guint) (self->capacity > 0 ? self->capacity + 1 : 2);
      GstBufferPool *pool =
          bayer2rgb_make_pinned_pool (self, caps, (guint) size, min);

      if (pool) {
        gst_query_add_allocation_pool (query, pool, (guint) size, min, 0);
        gst_object_unref (pool);
        GST_DEBUG_OBJECT (self, "using a pinned output pool");
      }
    }
  }
  return GST_BASE_TRANSFORM_CLASS (gst_bayer2rgb_parent_class)->decide_allocation
      (base, query);
}

/* ---- data flow -------------------------------------------------------------------- */

/* map both buffers and hand the frame to the GPU pool; on success the mapped
 * frame is appended to self->pending */
static GstFlowReturn
bayer2rgb_submit (GstBayer2RGB * self, GstBuffer * inbuf, GstBuffer * outbuf,
    gboolean owns_outbuf)
{
  Bayer2RGBPending *p = g_new0 (Bayer2RGBPending, 1);
  int rc;

  if (!gst_buffer_map (inbuf, &p->in_map, GST_MAP_READ)) {
    g_free (p);
    return GST_FLOW_CUSTOM_ERROR;       /* map failure: see callers */
  }
  if (!gst_video_frame_map (&p->out_frame, &self->info, outbuf, GST_MAP_WRITE)) {
    gst_buffer_unmap (inbuf, &p->in_map);
    g_free (p);
    return GST_FLOW_CUSTOM_ERROR;
  }
  if (p->in_map.size < (gsize) GST_ROUND_UP_4 (self->width) * self->height) {
    GST_ELEMENT_ERROR (self, STREAM, FORMAT, ("bayer2rgb: short input buffer"),
        ("%" G_GSIZE_FORMAT " bytes for %dx%d", p->in_map.size, self->width,
            self->height));
    goto fail;
  }
  if (!bayer2rgb_ensure_pool (self,
          GST_VIDEO_FRAME_PLANE_STRIDE (&p->out_frame, 0)))
    goto fail;

  /* the call that replaces gst_bayer2rgb_process (reference :475-477) */
  rc = mibayer_pool_submit (self->pool, p->in_map.data,
      GST_VIDEO_FRAME_PLANE_DATA (&p->out_frame, 0), p);
  if (rc != MIBAYER_OK) {
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
        ("bayer2rgb: GPU conversion failed"),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    goto fail;
  }
  p->inbuf = gst_buffer_ref (inbuf);
  p->outbuf = outbuf;
  p->owns_outbuf = owns_outbuf;
  g_queue_push_tail (&self->pending, p);
  return GST_FLOW_OK;

fail:
  gst_video_frame_unmap (&p->out_frame);
  gst_buffer_unmap (inbuf, &p->in_map);
  g_free (p);
  return GST_FLOW_ERROR;
}

/* oldest frame: wait for the GPU, unmap, hand the output buffer back */
static GstFlowReturn
bayer2rgb_collect (GstBayer2RGB * self, GstBuffer ** outbuf)
{
  Bayer2RGBPending *p = g_queue_pop_head (&self->pending);
  int rc;

  *outbuf = NULL;
  if (p == NULL)
    return GST_FLOW_OK;
  rc = mibayer_pool_wait (self->pool, NULL);
  if (rc != MIBAYER_OK) {
    pending_release (p, FALSE);
    GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
        ("bayer2rgb: GPU conversion failed"),
        ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    return GST_FLOW_ERROR;
  }
  *outbuf = p->outbuf;
  pending_release (p, TRUE);
  return GST_FLOW_OK;
}

/* reference :456-487 -- synchronous mode, the default */
static GstFlowReturn
gst_bayer2rgb_transform (GstBaseTransform * base, GstBuffer * inbuf,
    GstBuffer * outbuf)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);
  GstFlowReturn ret;
  GstBuffer *done = NULL;

  GST_DEBUG_OBJECT (self, "transforming buffer");

  ret = bayer2rgb_submit (self, inbuf, outbuf, FALSE);
  if (ret == GST_FLOW_CUSTOM_ERROR) {
    /* same as the reference: warn and skip (:484-486) */
    GST_WARNING_OBJECT (self, "Could not map buffer, skipping");
    return GST_FLOW_OK;
  }
  if (ret != GST_FLOW_OK)
    return ret;
  return bayer2rgb_collect (self, &done);       /* done == outbuf, still owned by the base class */
}

/* queued mode: take the input the base class parked in queued_buf, submit it,
 * and release the oldest frame once the pool is full */
static GstFlowReturn
gst_bayer2rgb_generate_output (GstBaseTransform * base, GstBuffer ** outbuf)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);
  GstBaseTransformClass *klass = GST_BASE_TRANSFORM_GET_CLASS (base);
  GstBuffer *inbuf;

  if (!bayer2rgb_is_queued_mode (self))
    return GST_BASE_TRANSFORM_CLASS (gst_bayer2rgb_parent_class)->generate_output
        (base, outbuf);

  *outbuf = NULL;
  inbuf = base->queued_buf;
  base->queued_buf = NULL;
  if (inbuf != NULL) {
    GstBuffer *out = NULL;
    GstFlowReturn ret = klass->prepare_output_buffer (base, inbuf, &out);

    if (ret != GST_FLOW_OK || out == NULL) {
      gst_buffer_unref (inbuf);
      return ret == GST_FLOW_OK ? GST_FLOW_ERROR : ret;
    }
    ret = bayer2rgb_submit (self, inbuf, out, TRUE);
    gst_buffer_unref (inbuf);           /* the pending entry holds its own ref */
    if (ret == GST_FLOW_CUSTOM_ERROR) {
      GST_WARNING_OBJECT (self, "Could not map buffer, skipping");
      gst_buffer_unref (out);
      return GST_FLOW_OK;
    }
    if (ret != GST_FLOW_OK) {
      gst_buffer_unref (out);
      return ret;
    }
  }
  if (self->capacity > 0
      && (gint) g_queue_get_length (&self->pending) >= self->capacity)
    return bayer2rgb_collect (self, outbuf);
  return GST_FLOW_OK;
}

static gboolean
gst_bayer2rgb_sink_event (GstBaseTransform * base, GstEvent * event)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);

  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_CAPS:
    case GST_EVENT_EOS:
    case GST_EVENT_SEGMENT:
    case GST_EVENT_GAP:
      /* frames in flight precede the event */
      bayer2rgb_drain (self, TRUE);
      break;
    case GST_EVENT_FLUSH_STOP:
      bayer2rgb_drain (self, FALSE);
      break;
    default:
      break;
  }
  return GST_BASE_TRANSFORM_CLASS (gst_bayer2rgb_parent_class)->sink_event
      (base, event);
}

/* queued mode holds up to `capacity` frames back: report that to live pipelines */
static gboolean
gst_bayer2rgb_query (GstBaseTransform * base, GstPadDirection direction,
    GstQuery * query)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);

  if (direction == GST_PAD_SRC && GST_QUERY_TYPE (query) == GST_QUERY_LATENCY
      && bayer2rgb_is_queued_mode (self)) {
    gboolean live = FALSE;
    GstClockTime min = 0, max = GST_CLOCK_TIME_NONE;
    gint fps_n = GST_VIDEO_INFO_FPS_N (&self->info);
    gint fps_d = GST_VIDEO_INFO_FPS_D (&self->info);

    if (!gst_pad_peer_query (GST_BASE_TRANSFORM_SINK_PAD (base), query))
      return FALSE;
    gst_query_parse_latency (query, &live, &min, &max);
    if (fps_n > 0 && fps_d > 0) {
      mibayer_pool_cfg pc;
      GstClockTime held;

      if (!bayer2rgb_parse_devices (self, &pc))
        pc.ndevices = 1;
      held = gst_util_uint64_scale_int (GST_SECOND * (guint64) (self->inflight
              * pc.ndevices), fps_d, fps_n);
      min += held;
      if (GST_CLOCK_TIME_IS_VALID (max))
        max += held;
      GST_DEBUG_OBJECT (self, "queued mode adds %" GST_TIME_FORMAT " latency",
          GST_TIME_ARGS (held));
    }
    gst_query_set_latency (query, live, min, max);
    return TRUE;
  }
  return GST_BASE_TRANSFORM_CLASS (gst_bayer2rgb_parent_class)->query (base,
      direction, query);
}

static gboolean
gst_bayer2rgb_stop (GstBaseTransform * base)
{
  bayer2rgb_drop_pool (GST_BAYER2RGB (base));
  return TRUE;
}

/* ---- type ------------------------------------------------------------------------ */

static void
gst_bayer2rgb_class_init (GstBayer2RGBClass * klass)
{
  GObjectClass *object_class = G_OBJECT_CLASS (klass);
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *transform_class = GST_BASE_TRANSFORM_CLASS (klass);

  object_class->set_property = gst_bayer2rgb_set_property;
  object_class->get_property = gst_bayer2rgb_get_property;
  object_class->finalize = gst_bayer2rgb_finalize;

  g_object_class_install_property (object_class, PROP_DEVICE_ID,
      g_param_spec_int ("device-id", "Device ID",
          "HIP ordinal of the MI355X that converts this stream", 0, G_MAXINT,
          DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_DEVICES,
      g_param_spec_string ("devices", "Devices",
          "Comma-separated HIP ordinals; frames are sharded round-robin over "
          "them (frame g -> devices[g % N]); empty = device-id only", NULL,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_INFLIGHT,
      g_param_spec_int ("inflight", "Frames in flight",
          "Frames in flight per device; 1 = strictly synchronous 1-in/1-out "
          "like the stock element, more = queued mode (adds latency)", 1, 16,
          DEFAULT_INFLIGHT, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_HIPGRAPH,
      g_param_spec_boolean ("hipgraph", "hipGraph per frame",
          "Run each frame's upload/kernel/download chain as one instantiated "
          "hipGraph", DEFAULT_HIPGRAPH,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (object_class, PROP_PINNED_POOL,
      g_param_spec_boolean ("pinned-pool", "Pinned buffer pools",
          "Propose hipHostMalloc-pinned buffer pools upstream and use them "
          "downstream when no other pool is offered", DEFAULT_PINNED_POOL,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  gst_element_class_set_static_metadata (element_class,
      "Bayer to RGB decoder for cameras", "Filter/Converter/Video",
      "Converts video/x-bayer to video/x-raw",
      "William Brack <wbrack@mmm.com.hk>");

  gst_element_class_add_pad_template (element_class,
      gst_pad_template_new ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
          gst_caps_from_string (BAYER2RGB_SRC_CAPS)));
  gst_element_class_add_pad_template (element_class,
      gst_pad_template_new ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
          gst_caps_from_string (BAYER2RGB_SINK_CAPS)));

  transform_class->transform_caps =
      GST_DEBUG_FUNCPTR (gst_bayer2rgb_transform_caps);
  transform_class->get_unit_size =
      GST_DEBUG_FUNCPTR (gst_bayer2rgb_get_unit_size);
  transform_class->set_caps = GST_DEBUG_FUNCPTR (gst_bayer2rgb_set_caps);
  transform_class->transform = GST_DEBUG_FUNCPTR (gst_bayer2rgb_transform);
  transform_class->generate_output =
      GST_DEBUG_FUNCPTR (gst_bayer2rgb_generate_output);
  transform_class->sink_event = GST_DEBUG_FUNCPTR (gst_bayer2rgb_sink_event);
  transform_class->query = GST_DEBUG_FUNCPTR (gst_bayer2rgb_query);
  transform_class->propose_allocation =
      GST_DEBUG_FUNCPTR (gst_bayer2rgb_propose_allocation);
  transform_class->decide_allocation =
      GST_DEBUG_FUNCPTR (gst_bayer2rgb_decide_allocation);
  transform_class->stop = GST_DEBUG_FUNCPTR (gst_bayer2rgb_stop);

  GST_DEBUG_CATEGORY_INIT (gst_bayer2rgb_debug, "bayer2rgb", 0,
      "bayer2rgb element");
}

static void
gst_bayer2rgb_init (GstBayer2RGB * self)
{
  bayer2rgb_clear_negotiation (self);
  self->device_id = DEFAULT_DEVICE_ID;
  self->devices = NULL;
  self->inflight = DEFAULT_INFLIGHT;
  self->use_hipgraph = DEFAULT_HIPGRAPH;
  self->pinned_pool = DEFAULT_PINNED_POOL;
  self->pool = NULL;
  self->pool_dst_stride = 0;
  self->capacity = 0;
  g_queue_init (&self->pending);
  /* the reference asks for in-place operation (:209) although no transform_ip
   * exists; kept so that base-class behaviour is the same */
  gst_base_transform_set_in_place (GST_BASE_TRANSFORM (self), TRUE);
}

gboolean
gst_bayer2rgb_register (GstPlugin * plugin)
{
  /* GST_ELEMENT_REGISTER (bayer2rgb, plugin) in the reference (:149-150,
   * gstbayer.c:33); spelled out so that it also builds against GStreamer < 1.20 */
  return gst_element_register (plugin, "bayer2rgb", GST_RANK_NONE,
      GST_TYPE_BAYER2RGB);
}
