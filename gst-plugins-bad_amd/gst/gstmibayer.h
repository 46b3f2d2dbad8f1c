/* Registration entry points of the two elements of plugin `bayer`
 * (reference: GST_ELEMENT_REGISTER_DECLARE in gst/bayer/gstbayerelements.h:30-31). */
#ifndef MI_GST_BAYER_H
#define MI_GST_BAYER_H

#include <gst/gst.h>

G_BEGIN_DECLS

gboolean gst_bayer2rgb_register (GstPlugin * plugin);
gboolean gst_rgb2bayer_register (GstPlugin * plugin);

G_END_DECLS
#endif
