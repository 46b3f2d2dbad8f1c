/* Plugin `bayer`: entry point.
 *
 * Drop-in for reference gst/bayer/gstbayer.c:28-43: same plugin name ("bayer"),
 * same description, same two element factories ("bayer2rgb" and "rgb2bayer",
 * both GST_RANK_NONE).  Install ahead of the stock plugin with GST_PLUGIN_PATH.
 */
#include <gst/gst.h>

#include "gstmibayer.h"

#ifndef PACKAGE
#define PACKAGE "gst-plugins-bad_amd"
#endif
#ifndef VERSION
#define VERSION "0.1.0"
#endif

static gboolean
plugin_init (GstPlugin * plugin)
{
  gboolean ok = FALSE;

  ok |= gst_bayer2rgb_register (plugin);
  ok |= gst_rgb2bayer_register (plugin);
  return ok;
}

GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, MIBAYER_PLUGIN_NAME,
    "Elements to convert Bayer images", plugin_init, VERSION, "LGPL",
    "gst-plugins-bad_amd (MI355X-native bayer2rgb)",
    "https://gstreamer.freedesktop.org/")
