/* rgb2bayer -- MI355X-native element (the bayer plugin's second element): the
 * concrete GType.
 *
 * Observable identity follows reference gst/bayer/gstrgb2bayer.c: factory and
 * GType name :80-81, pad templates :47-74, metadata :98-102, parent type
 * GstBaseTransform.  The behaviour -- the mirror image of bayer2rgb's caps
 * functions (:128-228) and the HIP kernel launch that replaces the per-pixel
 * double loop of gst_rgb2bayer_transform (:254-268) -- is shared with bayer2rgb
 * and lives in gstmibayerelement.c.  No CPU fallback.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "gstmibayerelement.h"
#include "gstmibayer.h"

typedef GstMiBayerElement GstRGB2Bayer;
typedef GstMiBayerElementClass GstRGB2BayerClass;

GType gst_rgb2bayer_get_type (void);

static GstStaticPadTemplate rgb2bayer_sink_template =
GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VIDEO_CAPS_MAKE ("ARGB")));

static GstStaticPadTemplate rgb2bayer_src_template =
GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS ("video/x-bayer,format=(string){bggr,gbrg,grbg,rggb},"
        "width=[1,MAX],height=[1,MAX],framerate=(fraction)[0/1,MAX]"));

MI_DEFINE_ELEMENT_TYPE (GstRGB2Bayer, gst_rgb2bayer, MIBAYER_TYPE_NAME ("RGB2Bayer"));

static void
gst_rgb2bayer_class_init (GstRGB2BayerClass * klass)
{
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);

  gst_mi_bayer_element_class_setup (klass, TRUE, "rgb2bayer");

  gst_element_class_add_static_pad_template (element_class,
      &rgb2bayer_src_template);
  gst_element_class_add_static_pad_template (element_class,
      &rgb2bayer_sink_template);
  gst_element_class_set_static_metadata (element_class,
      "RGB to Bayer converter", "Filter/Converter/Video",
      "Converts video/x-raw to video/x-bayer",
      "David Schleef <ds@entropywave.com>");
}

static void
gst_rgb2bayer_init (GstRGB2Bayer * self)
{
  gst_mi_bayer_element_instance_setup (self);
}

gboolean
gst_rgb2bayer_register (GstPlugin * plugin)
{
  /* reference gstrgb2bayer.c:80-81, gstbayer.c:34 */
  return gst_element_register (plugin, MIBAYER_FACTORY ("rgb2bayer"), GST_RANK_NONE,
      gst_rgb2bayer_get_type ());
}
