/* bayer2rgb -- MI355X-native element: the concrete GType.
 *
 * What a neighbouring element can observe is kept identical to the reference
 * element (gst-plugins-bad 1.19.2, gst/bayer/gstbayer2rgb.c):
 *   factory name / rank / GType name            :148-150
 *   element metadata (long name, klass, ...)    :180-183
 *   pad templates                               :134-138, :185-190
 *   parent type GstBaseTransform                :148
 * The behaviour (caps functions, data flow, the call into the HIP path that
 * replaces gst_bayer2rgb_process, :475-477) is shared with rgb2bayer and lives
 * in gstmibayerelement.c.
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include "gstmibayerelement.h"
#include "gstmibayer.h"

typedef GstMiBayerElement GstBayer2RGB;
typedef GstMiBayerElementClass GstBayer2RGBClass;

GType gst_bayer2rgb_get_type (void);

/* identical strings to the reference, order matters: the first src format
 * (RGBx) is what default negotiation fixates to */
#define BAYER2RGB_SRC_CAPS \
  GST_VIDEO_CAPS_MAKE ("{ RGBx, xRGB, BGRx, xBGR, RGBA, ARGB, BGRA, ABGR }")
#define BAYER2RGB_SINK_CAPS \
  "video/x-bayer,format=(string){bggr,grbg,gbrg,rggb}," \
  "width=(int)[1,MAX],height=(int)[1,MAX],framerate=(fraction)[0/1,MAX]"

MI_DEFINE_ELEMENT_TYPE (GstBayer2RGB, gst_bayer2rgb, MIBAYER_TYPE_NAME ("Bayer2RGB"));

static void
gst_bayer2rgb_class_init (GstBayer2RGBClass * klass)
{
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);

  gst_mi_bayer_element_class_setup (klass, FALSE, "bayer2rgb");

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
}

static void
gst_bayer2rgb_init (GstBayer2RGB * self)
{
  gst_mi_bayer_element_instance_setup (self);
  /* the reference asks for in-place operation (:209) although no transform_ip
   * exists; kept so that base-class behaviour is the same */
  gst_base_transform_set_in_place (GST_BASE_TRANSFORM (self), TRUE);
}

gboolean
gst_bayer2rgb_register (GstPlugin * plugin)
{
  /* GST_ELEMENT_REGISTER (bayer2rgb, plugin) in the reference (:149-150,
   * gstbayer.c:33); spelled out so that it also builds against GStreamer < 1.20 */
  return gst_element_register (plugin, MIBAYER_FACTORY ("bayer2rgb"), GST_RANK_NONE,
      gst_bayer2rgb_get_type ());
}
