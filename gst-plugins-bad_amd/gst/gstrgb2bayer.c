/* rgb2bayer -- MI355X-native element (the bayer plugin's second element).
 *
 * Observable behaviour follows reference gst/bayer/gstrgb2bayer.c: pad
 * templates :47-74, metadata :98-102, transform_caps :128-159, get_unit_size
 * :161-188, set_caps :190-228.  The per-pixel double loop of
 * gst_rgb2bayer_transform (:254-268) is replaced by one HIP kernel launch through
 * the C ABI (mibayer.h, MIBAYER_FLAG_RGB2BAYER).  No CPU fallback.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <string.h>

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "mibayer.h"
#include "gstmibayer.h"

/* ---- type (private to this file) ------------------------------------------------ */

#define GST_TYPE_RGB_2_BAYER (gst_rgb2bayer_get_type ())
#define GST_RGB_2_BAYER(obj) \
  (G_TYPE_CHECK_INSTANCE_CAST ((obj), GST_TYPE_RGB_2_BAYER, GstRGB2Bayer))

typedef struct _GstRGB2Bayer GstRGB2Bayer;
typedef struct _GstRGB2BayerClass GstRGB2BayerClass;

struct _GstRGB2Bayer
{
  GstBaseTransform base_rgb2bayer;

  GstVideoInfo info;            /* input video info */
  gint width, height;
  gint format;                  /* mibayer_pattern == reference enum, gstrgb2bayer.h:36-41 */

  gint device_id;               /* additive property */
  mibayer_ctx *ctx;
  gint ctx_src_stride;
};

struct _GstRGB2BayerClass
{
  GstBaseTransformClass base_rgb2bayer_class;
};

GType gst_rgb2bayer_get_type (void);

GST_DEBUG_CATEGORY_STATIC (gst_rgb2bayer_debug);
#define GST_CAT_DEFAULT gst_rgb2bayer_debug

enum
{
  PROP_0,
  PROP_DEVICE_ID
};

static GstStaticPadTemplate rgb2bayer_sink_template =
GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE ("ARGB")));

static GstStaticPadTemplate rgb2bayer_src_template =
GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS ("video/x-bayer,format=(string){bggr,gbrg,grbg,rggb},"
        "width=[1,MAX],height=[1,MAX],framerate=(fraction)[0/1,MAX]"));

G_DEFINE_TYPE (GstRGB2Bayer, gst_rgb2bayer, GST_TYPE_BASE_TRANSFORM);

static void
rgb2bayer_drop_context (GstRGB2Bayer * self)
{
  if (self->ctx) {
    mibayer_destroy (self->ctx);
    self->ctx = NULL;
  }
  self->ctx_src_stride = 0;
}

static gboolean
rgb2bayer_ensure_context (GstRGB2Bayer * self, gint src_stride)
{
  mibayer_cfg cfg;
  int rc;

  if (self->ctx && self->ctx_src_stride == src_stride)
    return TRUE;
  rgb2bayer_drop_context (self);
  memset (&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.width = self->width;
  cfg.height = self->height;
  cfg.src_stride = src_stride;                          /* frame.info.stride[0], :256 */
  cfg.dst_stride = GST_ROUND_UP_4 (self->width);        /* :255 */
  cfg.pattern = self->format;
  cfg.r_off = GST_VIDEO_INFO_COMP_OFFSET (&self->info, 0);      /* ARGB: 1, :263 */
  cfg.g_off = GST_VIDEO_INFO_COMP_OFFSET (&self->info, 1);      /* 2, :265 */
  cfg.b_off = GST_VIDEO_INFO_COMP_OFFSET (&self->info, 2);      /* 3, :261 */
  cfg.device = self->device_id;
  cfg.flags = MIBAYER_FLAG_RGB2BAYER;
  rc = mibayer_create (&cfg, &self->ctx);
  if (rc != MIBAYER_OK) {
    self->ctx = NULL;
    if (rc == MIBAYER_ERR_NO_DEVICE)
      GST_ELEMENT_ERROR (self, RESOURCE, NOT_FOUND,
          ("rgb2bayer: no usable MI355X / HIP device (device-id=%d)",
              self->device_id),
          ("%s; this element has no CPU path", mibayer_strerror (rc)));
    else
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("rgb2bayer: cannot create GPU context"),
          ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
    return FALSE;
  }
  self->ctx_src_stride = src_stride;
  return TRUE;
}

static void
gst_rgb2bayer_set_property (GObject * object, guint prop_id,
    const GValue * value, GParamSpec * pspec)
{
  GstRGB2Bayer *self = GST_RGB_2_BAYER (object);

  if (prop_id == PROP_DEVICE_ID)
    self->device_id = g_value_get_int (value);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

static void
gst_rgb2bayer_get_property (GObject * object, guint prop_id, GValue * value,
    GParamSpec * pspec)
{
  GstRGB2Bayer *self = GST_RGB_2_BAYER (object);

  if (prop_id == PROP_DEVICE_ID)
    g_value_set_int (value, self->device_id);
  else
    G_OBJECT_WARN_INVALID_PROPERTY_ID (object, prop_id, pspec);
}

static void
gst_rgb2bayer_finalize (GObject * object)
{
  rgb2bayer_drop_context (GST_RGB_2_BAYER (object));
  G_OBJECT_CLASS (gst_rgb2bayer_parent_class)->finalize (object);
}

/* reference :128-159: the mirror image of bayer2rgb's caps transform */
static GstCaps *
gst_rgb2bayer_transform_caps (GstBaseTransform * trans,
    GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  GstCaps *result = gst_caps_copy (caps);
  guint i, n = gst_caps_get_size (result);

  for (i = 0; i < n; i++) {
    GstStructure *s = gst_caps_get_structure (result, i);

    if (direction == GST_PAD_SRC) {
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
  GST_DEBUG_OBJECT (trans, "transformed %" GST_PTR_FORMAT " into %"
      GST_PTR_FORMAT, caps, result);
  return result;
}

/* reference :161-188 */
static gboolean
gst_rgb2bayer_get_unit_size (GstBaseTransform * trans, GstCaps * caps,
    gsize * size)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w, h;

  if (!gst_structure_get_int (s, "width", &w)
      || !gst_structure_get_int (s, "height", &h))
    return FALSE;
  if (gst_structure_has_name (s, "video/x-bayer"))
    *size = (gsize) GST_ROUND_UP_4 (w) * h;
  else
    *size = (gsize) w * h * 4;
  return TRUE;
}

/* reference :190-228 */
static gboolean
gst_rgb2bayer_set_caps (GstBaseTransform * trans, GstCaps * incaps,
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
  GstRGB2Bayer *self = GST_RGB_2_BAYER (trans);
  GstStructure *s;
  const gchar *order;
  GstVideoInfo info;
  guint i;

  GST_DEBUG_OBJECT (self, "in caps %" GST_PTR_FORMAT " out caps %"
      GST_PTR_FORMAT, incaps, outcaps);
  if (!gst_video_info_from_caps (&info, incaps))
    return FALSE;
  self->info = info;

  s = gst_caps_get_structure (outcaps, 0);
  gst_structure_get_int (s, "width", &self->width);
  gst_structure_get_int (s, "height", &self->height);
  order = gst_structure_get_string (s, "format");
  if (order == NULL)
    return FALSE;
  for (i = 0; i < G_N_ELEMENTS (orders); i++)
    if (g_str_equal (order, orders[i].name))
      break;
  if (i == G_N_ELEMENTS (orders))
    return FALSE;
  self->format = orders[i].pattern;
  rgb2bayer_drop_context (self);
  return TRUE;
}

/* reference :230-278 */
static GstFlowReturn
gst_rgb2bayer_transform (GstBaseTransform * trans, GstBuffer * inbuf,
    GstBuffer * outbuf)
{
  GstRGB2Bayer *self = GST_RGB_2_BAYER (trans);
  GstVideoFrame in_frame;
  GstMapInfo out_map;
  GstFlowReturn ret = GST_FLOW_OK;

  if (!gst_video_frame_map (&in_frame, &self->info, inbuf, GST_MAP_READ))
    goto map_failed;
  if (!gst_buffer_map (outbuf, &out_map, GST_MAP_WRITE)) {
    gst_video_frame_unmap (&in_frame);
    goto map_failed;
  }

  if (out_map.size < (gsize) GST_ROUND_UP_4 (self->width) * self->height) {
    GST_ELEMENT_ERROR (self, STREAM, FORMAT, ("rgb2bayer: short output buffer"),
        (NULL));
    ret = GST_FLOW_ERROR;
  } else if (!rgb2bayer_ensure_context (self,
          GST_VIDEO_FRAME_PLANE_STRIDE (&in_frame, 0))) {
    ret = GST_FLOW_ERROR;
  } else {
    int rc = mibayer_process_host (self->ctx,
        GST_VIDEO_FRAME_PLANE_DATA (&in_frame, 0), out_map.data);

    if (rc != MIBAYER_OK) {
      GST_ELEMENT_ERROR (self, RESOURCE, FAILED,
          ("rgb2bayer: GPU conversion failed"),
          ("%s %s", mibayer_strerror (rc), mibayer_last_hip_error ()));
      ret = GST_FLOW_ERROR;
    }
  }
  gst_buffer_unmap (outbuf, &out_map);
  gst_video_frame_unmap (&in_frame);
  return ret;

map_failed:
  GST_WARNING_OBJECT (trans, "Could not map buffer, skipping");
  return GST_FLOW_OK;
}

static gboolean
gst_rgb2bayer_stop (GstBaseTransform * trans)
{
  rgb2bayer_drop_context (GST_RGB_2_BAYER (trans));
  return TRUE;
}

static void
gst_rgb2bayer_class_init (GstRGB2BayerClass * klass)
{
  GObjectClass *object_class = G_OBJECT_CLASS (klass);
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *transform_class = GST_BASE_TRANSFORM_CLASS (klass);

  object_class->set_property = gst_rgb2bayer_set_property;
  object_class->get_property = gst_rgb2bayer_get_property;
  object_class->finalize = gst_rgb2bayer_finalize;
  g_object_class_install_property (object_class, PROP_DEVICE_ID,
      g_param_spec_int ("device-id", "Device ID",
          "HIP ordinal of the MI355X that converts this stream", 0, G_MAXINT, 0,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  gst_element_class_add_static_pad_template (element_class,
      &rgb2bayer_src_template);
  gst_element_class_add_static_pad_template (element_class,
      &rgb2bayer_sink_template);
  gst_element_class_set_static_metadata (element_class,
      "RGB to Bayer converter", "Filter/Converter/Video",
      "Converts video/x-raw to video/x-bayer",
      "David Schleef <ds@entropywave.com>");

  transform_class->transform_caps =
      GST_DEBUG_FUNCPTR (gst_rgb2bayer_transform_caps);
  transform_class->get_unit_size =
      GST_DEBUG_FUNCPTR (gst_rgb2bayer_get_unit_size);
  transform_class->set_caps = GST_DEBUG_FUNCPTR (gst_rgb2bayer_set_caps);
  transform_class->transform = GST_DEBUG_FUNCPTR (gst_rgb2bayer_transform);
  transform_class->stop = GST_DEBUG_FUNCPTR (gst_rgb2bayer_stop);

  GST_DEBUG_CATEGORY_INIT (gst_rgb2bayer_debug, "rgb2bayer", 0,
      "rgb2bayer element");
}

static void
gst_rgb2bayer_init (GstRGB2Bayer * self)
{
  gst_video_info_init (&self->info);
  self->width = self->height = 0;
  self->format = MIBAYER_BGGR;
  self->device_id = 0;
  self->ctx = NULL;
  self->ctx_src_stride = 0;
}

gboolean
gst_rgb2bayer_register (GstPlugin * plugin)
{
  /* reference gstrgb2bayer.c:80-81, gstbayer.c:34 */
  return gst_element_register (plugin, "rgb2bayer", GST_RANK_NONE,
      GST_TYPE_RGB_2_BAYER);
}
