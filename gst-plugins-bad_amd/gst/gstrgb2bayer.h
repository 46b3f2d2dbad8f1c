/* MI355X-native rgb2bayer element: type declaration (reference
 * gst/bayer/gstrgb2bayer.h:27-58). */
#ifndef MI_GST_RGB2BAYER_H
#define MI_GST_RGB2BAYER_H

#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <gst/video/video.h>

#include "mibayer.h"

G_BEGIN_DECLS

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
gboolean gst_rgb2bayer_register (GstPlugin * plugin);

G_END_DECLS
#endif
