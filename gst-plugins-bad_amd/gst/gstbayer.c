/* Plugin `bayer`: entry point.
 *
 * Drop-in for reference gst/bayer/gstbayer.c:28-43: same plugin name ("bayer"),
 * same description, same element factory name ("bayer2rgb", GST_RANK_NONE).
 * The sibling element rgb2bayer (reference gstrgb2bayer.c) is not on the
 * accelerated path and is not provided by this build (DESIGN.md "Out of
 * scope"); install this plugin ahead of the stock one with GST_PLUGIN_PATH.
 */
#include <gst/gst.h>

#include "gstbayer2rgb.h"

#ifndef PACKAGE
#define PACKAGE "gst-plugins-bad_amd"
#endif
#ifndef VERSION
#define VERSION "0.1.0"
#endif

static gboolean
plugin_init (GstPlugin * plugin)
{
  return gst_bayer2rgb_register (plugin);
}

GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, bayer,
    "Elements to convert Bayer images", plugin_init, VERSION, "LGPL",
    "gst-plugins-bad_amd (MI355X-native bayer2rgb)",
    "https://gstreamer.freedesktop.org/")
