/* bayer2rgb -- MI355X-native element.
 *
 * Everything a neighbouring element can observe is kept identical to the
 * reference element (gst-plugins-bad 1.19.2, gst/bayer/gstbayer2rgb.c):
 *   factory name / rank / GType name            :148-150
 *   element metadata (long name, klass, ...)    :180-183
 *   pad templates                               :134-138, :185-190
 *   transform_caps / get_unit_size / set_caps   :289-322 / :324-352 / :237-276
 *   1-in/1-out synchronous transform            :456-487
 * What changes is below the transform vfunc: the reference calls its CPU/ORC
 * frame loop gst_bayer2rgb_process (:475-477); this element hands the mapped
 * pointers and strides to the HIP path through the C ABI of mibayer.h.  There
 * is no CPU fallback: without a usable MI355X the element posts a RESOURCE
 * error instead of converting on the host.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <string.h>

#include "gstbayer2rgb.h"

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
  PROP_DEVICE_ID
};

#define DEFAULT_DEVICE_ID 0

G_DEFINE_TYPE (GstBayer2RGB, gst_bayer2rgb, GST_TYPE_BASE_TRANSFORM);

/* ---- helpers -------------------------------------------------------------- */

static void
bayer2rgb_drop_context (GstBayer2RGB * self)
{
  if (self->ctx) {
    mibayer_destroy (self->ctx);
    self->ctx = NULL;
  }
  self->ctx_dst_stride = 0;
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
bayer2rgb_ensure_context (GstBayer2RGB * self, gint dst_stride)
{
  mibayer_cfg cfg;
  int rc;

  if (self->ctx && self->ctx_dst_stride == dst_stride)
    return TRUE;
  bayer2rgb_drop_context (self);

  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = self->width;
  cfg.height = self->height;
  cfg.src_stride = GST_ROUND_UP_4 (self->width);        /* reference :477 */
  cfg.dst_stride = dst_stride;                          /* reference :476 */
  cfg.pattern = self->format;
  cfg.r_off = self->r_off;
  cfg.g_off = self->g_off;
  cfg.b_off = self->b_off;
  cfg.device = self->device_id;
  cfg.inflight = 2;

  rc = mibayer_create (&cfg, &self->ctx);
  if (rc != MIBAYER_OK) {
    self->ctx = NULL;
    if (rc == MIBAYER_ERR_NO_DEVICE) {
      GST_ELEMENT_ERROR (self, RESOURCE, NOT_FOUND,
          ("bayer2rgb: no usable MI355X / HIP device (device-id=%d)",
              self->device_id),
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
  self->ctx_dst_stride = dst_stride;
  GST_DEBUG_OBJECT (self, "GPU context for %dx%d pattern %d stride %d",
      self->width, self->height, self->format, dst_stride);
  return TRUE;
}

/* ---- GObject ----------------------------------------------------------------- */

static void
gst_bayer2rgb_set_property (GObject * object, guint prop_id,
    const GValue * value, GParamSpec * pspec)
{
  GstBayer2RGB *self = GST_BAYER2RGB (object);

  switch (prop_id) {
    case PROP_DEVICE_ID:
      GST_OBJECT_LOCK (self);
      self->device_id = g_value_get_int (value);
      GST_OBJECT_UNLOCK (self);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
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
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
      break;
  }
}

static void
gst_bayer2rgb_finalize (GObject * object)
{
  bayer2rgb_drop_context (GST_BAYER2RGB (object));
  G_OBJECT_CLASS (gst_bayer2rgb_parent_class)->finalize (object);
}

/* ---- GstBaseTransform vfuncs ------------------------------------------------- */

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

  /* geometry changed: the context is rebuilt on the next buffer, once the
   * mapped output stride is known */
  bayer2rgb_drop_context (self);
  return TRUE;
}

/* reference :456-487 */
static GstFlowReturn
gst_bayer2rgb_transform (GstBaseTransform * base, GstBuffer * inbuf,
    GstBuffer * outbuf)
{
  GstBayer2RGB *self = GST_BAYER2RGB (base);
  GstMapInfo in_map;
  GstVideoFrame out_frame;
  GstFlowReturn ret = GST_FLOW_OK;
  int rc;

  GST_DEBUG_OBJECT (self, "transforming buffer");

  if (!gst_buffer_map (inbuf, &in_map, GST_MAP_READ))
    goto map_failed;
  if (!gst_video_frame_map (&out_frame, &self->info, outbuf, GST_MAP_WRITE)) {
    gst_buffer_unmap (inbuf, &in_map);
    goto map_failed;
  }

  if (in_map.size < (gsize) GST_ROUND_UP_4 (self->width) * self->height) {
    GST_ELEMENT_ERROR (self, STREAM, FORMAT, ("bayer2rgb: short input buffer"),
        ("%" G_GSIZE_FORMAT " bytes for %dx%d", in_map.size, self->width,
            self->height));
    ret = GST_FLOW_ERROR;
  } else if (!bayer2rgb_ensure_context (self,
          GST_VIDEO_FRAME_PLANE_STRIDE (&out_frame, 0))) {
    ret = GST_FLOW_ERROR;
  } else {
    /* the call that replaces gst_bayer2rgb_process (reference :475-477) */
    rc = mibayer_process_host (self->ctx, in_map.data,
        GST_VIDEO_FRAME_PLANE_DATA (&out_frame, 0));
    if (rc != MIBAYER_OK) {
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("bayer2rgb: GPU conversion failed"),
          ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
      ret = GST_FLOW_ERROR;
    }
  }

  gst_video_frame_unmap (&out_frame);
  gst_buffer_unmap (inbuf, &in_map);
  return ret;

map_failed:
  GST_WARNING_OBJECT (self, "Could not map buffer, skipping");
  return GST_FLOW_OK;
}

static gboolean
gst_bayer2rgb_stop (GstBaseTransform * base)
{
  bayer2rgb_drop_context (GST_BAYER2RGB (base));
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
  transform_class->stop = GST_DEBUG_FUNCPTR (gst_bayer2rgb_stop);

  GST_DEBUG_CATEGORY_INIT (gst_bayer2rgb_debug, "bayer2rgb", 0,
      "bayer2rgb element");
}

static void
gst_bayer2rgb_init (GstBayer2RGB * self)
{
  bayer2rgb_clear_negotiation (self);
  self->device_id = DEFAULT_DEVICE_ID;
  self->ctx = NULL;
  self->ctx_dst_stride = 0;
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
